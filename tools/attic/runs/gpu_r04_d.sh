#!/bin/bash
# round 4, call d: what the FIRST call of the device tokenizer pays (buffers, page-locked staging, streams) and how many copy
# streams the staging threads need
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04d; mkdir -p $O
for ns in 8 4 2 1; do
  echo "== PG_TOK_STREAMS=$ns" >> $O/tok_first_call.txt
  PG_TOK_DEBUG=1 PG_TOK_STREAMS=$ns timeout 300 python tools/tok_bench2.py 1300000 200 2>&1 | grep -E "^tok:|^file \(pread\), call" | head -24 >> $O/tok_first_call.txt
done
cat $O/tok_first_call.txt
