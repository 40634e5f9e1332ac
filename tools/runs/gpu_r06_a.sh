#!/bin/bash
# round 6, call a: the -m gpu suite at HEAD; the placement experiments (block order, slices); the CU-partition sweep; the DRIVER'S
# command under rocprofv3 with placement trials on (VERDICT round 5 #1a), and PG_PLACE_TRIALS=1 four times for the spread
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06a; mkdir -p $O/prof
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -3
timeout 600 python tools/pack_placement.py 8 > $O/pack_placement.txt 2> $O/pack_placement.err; cat $O/pack_placement.txt | cut -c1-420; tail -3 $O/pack_placement.err
timeout 900 python tools/cu_split_sweep.py > $O/cu_split_sweep.txt 2> $O/cu_split_sweep.err; cat $O/cu_split_sweep.txt | cut -c1-330; tail -3 $O/cu_split_sweep.err
# the driver's own command, placement trials on
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o default --output-format csv -- python bench.py > $O/bench_default_under_rocprof.json 2> $O/bench_default_under_rocprof.err
tail -c 1500 $O/bench_default_under_rocprof.json | cut -c1-1500; echo
python tools/prof_timed_region.py $O/prof/default_kernel_trace.csv 10 > $O/northstar_timed_region_kernel_stats.csv; head -8 $O/northstar_timed_region_kernel_stats.csv
for k in 1 2 3 4; do
  PG_PLACE_TRIALS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o place$k --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/bench_place$k.json 2> $O/bench_place$k.err
  python tools/prof_timed_region.py $O/prof/place${k}_kernel_trace.csv 5 | head -3
done
ls -la $O/prof | head; du -sh $O
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
