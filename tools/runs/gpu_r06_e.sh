#!/bin/bash
# round 6, call e: the whole -m gpu suite at HEAD (per-window --inferPloidy, line feeds listed by k_inflate, k_crc32 on its own stream,
# the native gzip reader, the runtime's start-up beside the imports); the default bench line (T2 legs now timed at the reference's
# default rounding); the whole north star again under the kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06e; mkdir -p $O/prof
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest.log | tail -8
timeout 1500 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; tail -c 3000 $O/bench_northstar_default.json | cut -c1-3000; echo
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 3 > $O/whole.json 2> $O/whole.err; tail -c 1500 $O/whole.json; echo
CMD=$(cat /tmp/pg_ns_cmd.txt)
for k in 1 2 3; do PG_EARLY_INIT=0 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"compute_and_write_s": [0-9.]*' | paste - - - -; done > $O/timing_no_early_init.txt
for k in 1 2 3; do PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"compute_and_write_s": [0-9.]*' | paste - - - -; done > $O/timing_early_init.txt
for k in 1 2 3; do PG_BGZF_CRC_STREAM=0 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"compute_and_write_s": [0-9.]*' | paste - - - -; done > $O/timing_crc_on_the_chain.txt
echo no_early; cat $O/timing_no_early_init.txt; echo early; cat $O/timing_early_init.txt; echo crc_chain; cat $O/timing_crc_on_the_chain.txt
for k in 1 2 3; do /usr/bin/time -f "%e s wall" $CMD 2>&1 | tail -1; done > $O/wall_early.txt; for k in 1 2 3; do PG_EARLY_INIT=0 /usr/bin/time -f "%e s wall" $CMD 2>&1 | tail -1; done > $O/wall_no_early.txt; paste $O/wall_early.txt $O/wall_no_early.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o whole --output-format csv -- $CMD > $O/prof.log 2>&1
head -14 $O/prof/whole_kernel_stats.csv | cut -c1-60,200-290
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
