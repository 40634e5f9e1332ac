#!/bin/bash
# round 4, call v: pi / dxy / Fst with the sums in NumPy's order (k_popdist_np): the kernel tests with ==, then timing against the older finisher
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04v; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "group_dist or half_missing or popdist or dense_poly or c2_popdist" ) 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
for t in 1 0 default; do
  env $( [ $t = default ] || echo PG_POPDIST_TREE=$t ) python bench.py --workload northstar --steps 10 --warmup 2 --no-cpu-baseline --no-tiers > $O/bench_tree$t.json 2> $O/err$t.txt
  python -c "
import json; d=json.loads(open('$O/bench_tree$t.json').read().strip().splitlines()[-1]); print('PG_POPDIST_TREE=$t', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
  env $( [ $t = default ] || echo PG_POPDIST_TREE=$t ) python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-tiers > $O/bench_c2_tree$t.json 2> $O/errc$t.txt
  python -c "
import json; d=json.loads(open('$O/bench_c2_tree$t.json').read().strip().splitlines()[-1]); print('c2 PG_POPDIST_TREE=$t', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
done
