#!/bin/bash
# round 2, call v: pair counts on the matrix cores as the default -- whole GPU suite, default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02v
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02v/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r02v/pytest.log | tail -3
timeout 900 python bench.py > gpurun_out/r02v/bench.json 2> gpurun_out/r02v/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02v/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"], d["kernel_ms_per_step"], d["pair_kernels"], d["cpu_baseline"]["value"])
PY
