#!/bin/bash
# round 5: a BGZF block's copy-in / inflate / checksum / line count on a stream of their own (beside the parse kernels of the block before)
# against everything on one stream (PG_TOK_ONE_STREAM=1): the whole north star at --roundTo 12 (nearly every window recomputed in
# NumPy's order: the main thread's kernels are many) and at the drivers' default rounding (4 places)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05two; mkdir -p $O
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 1 > $O/whole.json 2> $O/whole.err
CMD=$(cat /tmp/pg_ns_cmd.txt)
CMD4=$(echo "$CMD" | sed "s/--roundTo 12/--roundTo 4/")
G='"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"main_stats_s": [0-9.]*\|"prep_wait_s": [0-9.]*'
for k in 1 2 3 4; do
  PG_TOK_ONE_STREAM=1 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD4 2>&1 | grep PG_TIMING | grep -o "$G" | tr '\n' ' '; echo " (roundTo 4, one stream)"
  PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD4 2>&1 | grep PG_TIMING | grep -o "$G" | tr '\n' ' '; echo " (roundTo 4, two streams)"
done | tee $O/round4_ab.txt
for k in 1 2; do
  PG_TOK_ONE_STREAM=1 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o "$G" | tr '\n' ' '; echo " (roundTo 12, one stream)"
  PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o "$G" | tr '\n' ' '; echo " (roundTo 12, two streams)"
done | tee $O/round12_ab.txt
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
