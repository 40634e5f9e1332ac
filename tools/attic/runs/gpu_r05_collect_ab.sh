#!/bin/bash
# round 5: collect() waiting for the block's own event (tree) against waiting for the whole stream (gpurun_variants/libpopgen_streamsync.so):
# T2 on 8.1 GB of text, bgzipped and plain, + the golden / inflate / end-to-end tests with the new library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05ab; mkdir -p $O
S=/tmp/pg_r05_sample; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> /dev/null
CMDT=$(cat $S/cmd.txt)
CMDZ=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/sample.geno.gz #; s#$S/sample.geno.csv#$S/out_gz.csv#")
for rep in 1 2 3 4; do
  for lib in tree gpurun_variants/libpopgen_streamsync.so; do
    L="PG_X=1"; [ $lib != tree ] && L="PG_LIBRARY=$R/$lib"
    echo -n "bgzf $lib: "; env $L PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDZ 2>&1 | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"prep_wait_s": [0-9.]*' | tr '\n' ' '; echo
  done
done | tee $O/collect_ab_bgzf.txt
for rep in 1 2; do
  for lib in tree gpurun_variants/libpopgen_streamsync.so; do
    L="PG_X=1"; [ $lib != tree ] && L="PG_LIBRARY=$R/$lib"
    echo -n "text $lib: "; env $L PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDT 2>&1 | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"prep_wait_s": [0-9.]*' | tr '\n' ' '; echo
  done
done | tee $O/collect_ab_text.txt
cmp $S/sample.geno.csv $S/out_gz.csv && echo "csv of the bgzf run == csv of the text run"
rm -rf $S
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_inflate.py tests/test_gpu_e2e.py -m gpu -x -q -n 4 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
