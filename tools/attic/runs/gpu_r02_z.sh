#!/bin/bash
# round 2, call z: rocprofv3 stats + PMC passes of the final kernel selection (c2, northstar), bench lines of the other workloads
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z
mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
for wl in c2 northstar; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-100
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM -d $O/pmc_mfma -o $wl --output-format csv -- $B > $O/pmc_mfma_$wl.log 2>&1
done
rm -f $O/*/*agent_info.csv $O/prof_stats/*kernel_trace.csv
for wl in c2 c3 c4; do
  ST=20; [ $wl = c4 ] && ST=5
  timeout 600 python bench.py --workload $wl --steps $ST --warmup 3 > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "
import json,sys
d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['ms_per_step'], d['value'], d['roofline'].get('kernel'), d['roofline'].get('frac'), d.get('kernel_ms_per_step'))"
done
