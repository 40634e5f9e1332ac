#!/bin/bash
# round 5: block size of the streaming drivers on bgzipped and plain text (20.3 GB sample): start-up (allocations, pipeline fill) against per-block costs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05blk; mkdir -p $O
S=/tmp/pg_r05_sample; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 25000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> /dev/null
CMDT=$(cat $S/cmd.txt)
CMDZ=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/sample.geno.gz #; s#$S/sample.geno.csv#$S/out_gz.csv#")
for mb in 1024 512 256 128; do
  for k in 1 2 3; do
    echo -n "bgzf block ${mb} MiB: "; PG_STREAM_BYTES=$((mb << 20)) PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDZ 2>&1 | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"chunks": [0-9]*' | tr '\n' ' '; echo
  done
done | tee $O/block_size_bgzf.txt
for mb in 1024 512 256; do
  for k in 1 2; do
    echo -n "text block ${mb} MiB: "; PG_STREAM_BYTES=$((mb << 20)) PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDT 2>&1 | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"chunks": [0-9]*' | tr '\n' ' '; echo
  done
done | tee $O/block_size_text.txt
rm -rf $S
