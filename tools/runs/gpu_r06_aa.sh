#!/bin/bash
# round 6, call aa: a block's parse kernels on a stream of their own beside the next block's k_inflate (PG_TOK_PARSE_STREAM=1) against the
# copy stream: the whole north star as one bgzipped .geno.gz, alternating runs; the BGZF / golden / e2e GPU tests with it on
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06aa; mkdir -p $O
PG_TOK_PARSE_STREAM=1 timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_e2e.py tests/test_gpu_inflate.py -q -n 6 2>&1 | tail -2
PG_NS_KEEP=/tmp/ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 1 > $O/t2_northstar_first.json 2> $O/err.txt; cut -c1-600 $O/t2_northstar_first.json
CMD=$(cat /tmp/ns_cmd.txt)
for k in 1 2 3 4 5; do for v in 0 1; do
  PG_TOK_PARSE_STREAM=$v PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 >/dev/null | grep PG_TIMING | python -c "
import sys, json
t = json.loads(sys.stdin.read().split('PG_TIMING ', 1)[1])
print('parse_stream=$v', {k: round(t[k], 4) for k in ('total_s', 'context_s', 'tokenize_s', 'prep_wait_s', 'main_stats_s', 'tokenizer_kernels_s') if k in t})"
done; done | tee $O/t2_whole_parse_stream_ab.txt
