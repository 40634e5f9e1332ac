#!/bin/bash
# round 5: the whole north star as one bgzipped .geno.gz through popgenWindows.py under the kernel trace (what the GPU does for the
# 76 blocks of 1 GiB of text), after a plain run of the tool
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05whole; mkdir -p $O/prof
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 2 > $O/whole.json 2> $O/whole.err; tail -c 600 $O/whole.json; echo
CMD=$(cat /tmp/pg_ns_cmd.txt)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o whole --output-format csv -- $CMD > $O/prof.log 2>&1
PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | cut -c1-900 > $O/timing_after.txt
head -14 $O/prof/whole_kernel_stats.csv | cut -c1-160
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
