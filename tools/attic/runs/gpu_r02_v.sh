#!/bin/bash
# round 2, call v: whole GPU suite, smoke(), default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02v
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02v/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r02v/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r02v/bench.json 2> gpurun_out/r02v/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02v/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["kernel_ms_per_step"], d.get("placement_trials"), d["cpu_baseline"]["value"], d["t2"]["seconds"], d["ms_per_step_pipelined_sub_batches"])
PY
