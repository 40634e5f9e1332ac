#!/bin/bash
# round 5: k_tok_parse2 (several lines per wavefront, column tables in LDS) -- the goldens through the device tokenizer, then the kernels
# of a T2 run on 8.1 GB of bgzipped text under the kernel trace, with the first form (PG_TOK_PARSE=1) beside it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05tok; mkdir -p $O/prof
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_inflate.py tests/test_gpu_e2e.py -m gpu -x -q -n 4 > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -4
PG_TOK_PARSE=1 timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -n 4 -k "streaming_in_small_blocks" > $O/pytest_first_form.log 2>&1; tail -2 $O/pytest_first_form.log
S=/tmp/pg_r05_sample; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> /dev/null
CMD=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/sample.geno.gz #; s#$S/sample.geno.csv#$S/out_gz.csv#")
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t2_bgzf_tok2 --output-format csv -- $CMD > $O/prof_tok2.log 2>&1
PG_TOK_PARSE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t2_bgzf_tok1 --output-format csv -- $CMD > $O/prof_tok1.log 2>&1
for f in tok2 tok1; do echo $f; grep -E "k_tok_parse|k_inflate|k_nl_" $O/prof/t2_bgzf_${f}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120; done
for k in 1 2 3; do PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*' | tr '\n' ' '; echo; done | tee $O/t2_bgzf_tok2_timing.txt
$(cat $S/cmd.txt) > /dev/null 2>&1; cmp $S/sample.geno.csv $S/out_gz.csv && echo "csv equal"
rm -rf $S
