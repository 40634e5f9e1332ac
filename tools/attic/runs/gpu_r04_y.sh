#!/bin/bash
# round 4, call y: k_popdist_np variants: 2 kb windows of C2 and the north-star shape forced into NumPy's order
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "group_dist or numpy_order or window_by_window or half_missing or last_bit or through_the_api" 2>&1 | tail -1
python bench.py --workload c2_w2k --steps 20 --warmup 3 --no-tiers --no-cpu-baseline > $O/a.json 2> $O/a.err
python -c "
import json; d=json.loads(open('$O/a.json').read().strip().splitlines()[-1]); print('c2_w2k', d['ms_per_step'], d['kernel_ms_per_step'])"
PG_POPDIST_TREE=1 python bench.py --workload northstar --steps 5 --warmup 2 --no-tiers --no-cpu-baseline > $O/b.json 2> $O/b.err
python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('northstar forced', d['ms_per_step'], d['kernel_ms_per_step'])"
