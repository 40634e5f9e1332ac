#!/bin/bash
# round 4, call x2: kernel stats of the BASELINE workloads at the end of the round (the finishers' signatures changed late; the pack and
# pair kernels did not): rocprofv3 --kernel-trace --stats of the same bench commands as tools/runs/gpu_r04_prof.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r04x2; mkdir -p $O/prof_stats
export PG_PLACE_TRIALS=1
for wl in northstar c2 c4; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  head -6 $O/prof_stats/${wl}_kernel_stats.csv | cut -c1-60,200-330
done
