#!/bin/bash
# round 6, call ap: the positions' copy to the host on the small stream as soon as k_tok_heads has them (PG_TOK_POS_STREAM=0: on the copy
# stream behind the cell kernel) -- tests, alternating runs of the whole north star, the copy stream's kernels by rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ap; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_e2e.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -n 6 --timeout=300 2>&1 | tail -2
PG_NS_KEEP=/tmp/ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 1 > $O/t2_northstar_first.json 2> $O/err.txt; cut -c1-400 $O/t2_northstar_first.json; echo
CMD=$(cat /tmp/ns_cmd.txt)
for k in 1 2 3 4 5 6; do for v in 0 1; do
  PG_TOK_POS_STREAM=$v PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 >/dev/null | grep PG_TIMING | python -c "
import sys, json
t = json.loads(sys.stdin.read().split('PG_TIMING ', 1)[1])
print('pos_on_small_stream=$v', {k: round(t[k], 4) for k in ('total_s', 'context_s', 'tokenize_s', 'prep_wait_s', 'main_stats_s', 'tokenizer_kernels_s') if k in t}, 'without context', round(t['total_s'] - t['context_s'], 4))"
done; done | tee $O/t2_whole_pos_stream_ab.txt
