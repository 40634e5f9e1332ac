#!/bin/bash
# round 5: deferred copy-back of the distMat-shaped result table (C4): its test, then bench c4 with and without it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05c4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "deferred or indpair or distmat" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for k in 1 2; do
timeout 300 python bench.py --workload c4 --steps 20 --warmup 3 --no-tiers --no-cpu-baseline > $O/bench_c4_deferred_$k.json 2> $O/bench_c4.err; python - $O/bench_c4_deferred_$k.json <<'P'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("deferred:", b["value"], b["ms_per_step"], b.get("kernel_ms_per_step"))
P
PG_BENCH_DEFER=0 timeout 300 python bench.py --workload c4 --steps 20 --warmup 3 --no-tiers --no-cpu-baseline > $O/bench_c4_immediate_$k.json 2>> $O/bench_c4.err; python - $O/bench_c4_immediate_$k.json <<'P'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("immediate:", b["value"], b["ms_per_step"], b.get("kernel_ms_per_step"))
P
done
timeout 300 python bench.py --workload c4 --steps 20 --warmup 3 --no-tiers > $O/bench_c4.json 2>> $O/bench_c4.err; tail -c 600 $O/bench_c4.json
