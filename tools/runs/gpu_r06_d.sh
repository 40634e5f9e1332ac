#!/bin/bash
# round 6, call d: the --inferPloidy goldens and the BGZF tests on the real engine (line feeds listed by k_inflate); the whole north
# star as one bgzipped .geno.gz through popgenWindows.py under the kernel trace, with the passes over the text (PG_BGZF_NL=0) and
# without them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06d; mkdir -p $O/prof
timeout 1500 python -m pytest tests -m gpu -x -q -k "ploidy or inflate or bgzf or line_feeds or Bgzf" > $O/pytest_sel.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest_sel.log | tail -8
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 2 > $O/whole.json 2> $O/whole.err; tail -c 700 $O/whole.json; echo
CMD=$(cat /tmp/pg_ns_cmd.txt)
for k in 1 2 3; do PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | cut -c1-1000; done > $O/timing_lists.txt
for k in 1 2 3; do PG_BGZF_NL=0 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | cut -c1-1000; done > $O/timing_passes.txt
cut -c1-330 $O/timing_lists.txt; cut -c1-330 $O/timing_passes.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o whole_lists --output-format csv -- $CMD > $O/prof_lists.log 2>&1
PG_BGZF_NL=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o whole_passes --output-format csv -- $CMD > $O/prof_passes.log 2>&1
head -16 $O/prof/whole_lists_kernel_stats.csv | cut -c1-60,200-290
head -16 $O/prof/whole_passes_kernel_stats.csv | cut -c1-60,200-290
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
