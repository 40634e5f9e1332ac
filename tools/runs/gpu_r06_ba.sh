#!/bin/bash
# round 6, call ba: the driver's command under rocprofv3 --kernel-trace at the commit of pg_tune_planes: the launches of the timed region
# cut out of the trace (behind the fill, the plane probes and the warm-up) against the line's own HIP events
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ba; mkdir -p $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o default --output-format csv -- python bench.py --no-tiers --no-cpu-baseline > $O/bench_default_under_rocprof.json 2> $O/bench_default_under_rocprof.err
python tools/prof_timed_region.py $O/prof/default_kernel_trace.csv 10 2 k_pack3 $O/bench_default_under_rocprof.json > $O/northstar_default_timed_region_kernel_stats.csv; head -6 $O/northstar_default_timed_region_kernel_stats.csv
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06ba/bench_default_under_rocprof.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['placement_trials'].get('planes_probe_ms'))
PY
head -8 $O/prof/default_kernel_stats.csv | cut -c1-60,200-300
find $O -name "*kernel_trace.csv" -size +20M -delete
