#!/bin/bash
# round 2, call r: the device tokenizer as the default -- full GPU suite, T2 timing, default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02r
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02r/pytest.log 2>&1; tail -5 gpurun_out/r02r/pytest.log
timeout 900 python tools/t2_bench.py 10000000 100 > gpurun_out/r02r/t2.log 2>&1; grep -v "^wrote\|^packed /" gpurun_out/r02r/t2.log | cut -c1-420
timeout 900 python bench.py > gpurun_out/r02r/bench.json 2> gpurun_out/r02r/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02r/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d.get("tiers", {}).get("t2"))
PY
