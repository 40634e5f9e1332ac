#!/bin/bash
# round 4, call m: kernel trace of a tier-T2 run (8.1 GB of `.geno` text through popgenWindows.py): what the ingestion path launches
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04m; mkdir -p $O/prof
CMD=$(python tools/t2_write_sample.py /tmp/t2prof.geno 10000000 200 2> $O/sample.txt); cat $O/sample.txt
PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2> $O/timing_plain.txt; grep PG_TIMING $O/timing_plain.txt | cut -c1-600
PG_TIMING=1 PG_PLACE_TRIALS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o t2 --output-format csv -- $CMD > $O/rocprof.log 2>&1; grep PG_TIMING $O/rocprof.log | cut -c1-400
head -25 $O/prof/t2_kernel_stats.csv
