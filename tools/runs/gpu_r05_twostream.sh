#!/bin/bash
# round 5: a BGZF block's copy-in / inflate / checksum / line count on a stream of their own (beside the parse kernels of the block before)
# against everything on one stream (PG_TOK_ONE_STREAM=1): the whole north star, and the tests that use the path
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05two; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_inflate.py tests/test_gpu_e2e.py -m gpu -x -q -n 4 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 3 > $O/whole_two_streams.json 2> $O/whole.err
python - $O/whole_two_streams.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print("two streams:", [(r["total_s"], r["tokenize_s"], r["context_s"]) for r in d["runs"]], d["csv_matches_t0"])
P
CMD=$(cat /tmp/pg_ns_cmd.txt)
for k in 1 2 3; do PG_TOK_ONE_STREAM=1 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*' | tr '\n' ' '; echo " (one stream)"; done | tee $O/whole_one_stream.txt
for k in 1 2 3; do PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*' | tr '\n' ' '; echo " (two streams)"; done | tee $O/whole_two_streams.txt
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
