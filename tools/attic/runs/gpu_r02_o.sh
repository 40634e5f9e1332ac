#!/bin/bash
# GPU call O: placement trials of the pack kernel's scratch -- distribution of the pack time over fresh processes, with and without.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "pairwise or popdist or pipelines" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
show='
import sys, json
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln); print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"].get("frac"))
    elif "rror" in ln: print(ln.strip())
'
for rep in 1 2 3 4; do
  for v in PG_NONE=1 PG_PLACE_TRIALS=1; do
    echo "== northstar $v"
    env $v timeout 300 python bench.py --workload northstar --steps 5 --warmup 2 --no-cpu-baseline --no-tiers 2>&1 | tee -a $O/place_northstar.log | python -c "$show"
  done
done
for rep in 1 2 3; do
  for v in PG_NONE=1 PG_PLACE_TRIALS=1; do
    echo "== c2 $v"
    env $v timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-tiers 2>&1 | tee -a $O/place_c2.log | python -c "$show"
  done
done
