#!/bin/bash
# round 6, call i: which tests of the -m gpu suite take the time (--durations); the default bench line once more (gz leg)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06i; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -A45 "slowest" $O/pytest.log | head -50
timeout 1500 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06i/bench_northstar_default.json").read().strip().split("\n")[-1])
t2=d.get("t2") or d.get("extra",{}).get("t2")
print(d["value"], d["ms_per_step"], d["roofline"]["frac"]); print("gz", t2.get("gz")); print("bgzf", t2["bgzf"]["seconds"]["total_s"], "whole", t2["bgzf_whole_workload"].get("seconds"))
PY
