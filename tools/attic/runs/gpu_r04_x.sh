#!/bin/bash
# round 4, call x: kernel stats of the workloads whose kernels changed late in the round (popFreq: k_popfreq_ordered; C3: k_abba_q with
# its flag argument; 2 kb windows: k_popdist_np), their un-profiled bench lines, and the large-population test added last
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r04x; mkdir -p $O/prof_stats
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "too_large or window_by_window" 2>&1 | tail -2
export PG_PLACE_TRIALS=1
for wl in popfreq c3 c2_w2k; do
  B="python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-200
done
unset PG_PLACE_TRIALS
for wl in popfreq c3 c2 c2_w5k c2_w2k; do
  timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-tiers --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "
import json; d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['ms_per_step'], d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])"
done
PG_POPDIST_TREE=0 timeout 300 python bench.py --workload c2_w2k --steps 20 --warmup 3 --no-tiers --no-cpu-baseline > $O/bench_c2_w2k_fixed_tree.json 2> $O/e.err
python -c "
import json; d=json.loads(open('$O/bench_c2_w2k_fixed_tree.json').read().strip().splitlines()[-1]); print('c2_w2k fixed tree', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"
ls $O/prof_stats
