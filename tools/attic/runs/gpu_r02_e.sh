#!/bin/bash
# GPU call E: k_pack3 with the lane-parallel allele table -- parity of the pairwise paths, then same-box A/B against k_pack2.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "pairwise or pipelines or popdist" ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
show='
import sys, json
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln); print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d.get("kernel_ms_per_step"))
    elif "rror" in ln: print(ln.strip())
'
for rep in 1 2 3; do
  for v in PG_NONE=1 PG_PACK2=1; do
    echo "== c2 $v"
    env $v timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-tiers 2>&1 | tee -a $O/ab_c2.log | python -c "$show"
  done
done
for rep in 1 2; do
  for v in PG_PACK3=1 PG_NONE=1; do
    echo "== northstar $v"
    env $v timeout 300 python bench.py --workload northstar --steps 6 --warmup 2 --no-cpu-baseline --no-tiers 2>&1 | tee -a $O/ab_northstar.log | python -c "$show"
  done
done
