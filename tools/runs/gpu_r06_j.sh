#!/bin/bash
# round 6, call j: the members' CRC-32 taken inside k_inflate's flush against k_crc32 (PG_BGZF_CRC_FOLD=0): the inflate and BGZF tests,
# 20 000 random members, the kernels alone, the whole north star (timing + kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06j; mkdir -p $O/prof
timeout 1500 python -m pytest tests -m gpu -x -q -k "inflate or bgzf or Bgzf or vcf or line_feeds or damaged or golden" > $O/pytest_sel.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest_sel.log | tail -6
timeout 900 python tools/inflate_fuzz.py 20000 11 > $O/inflate_fuzz_20000.txt 2>&1; tail -1 $O/inflate_fuzz_20000.txt
S=/tmp/pg_r06j; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> /dev/null
for k in 1 2; do echo -n "fold "; python tools/inflate_bench.py --file $S/sample.geno.gz | tail -1; echo -n "k_crc32 "; PG_BGZF_CRC_FOLD=0 python tools/inflate_bench.py --file $S/sample.geno.gz | tail -1; done | tee $O/inflate_bench_crc_fold_ab.txt
rm -rf $S
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 3 > $O/whole_fold.json 2> $O/whole.err; tail -c 500 $O/whole_fold.json; echo
CMD=$(cat /tmp/pg_ns_cmd.txt)
for k in 1 2 3 4; do
  echo -n "fold    "; PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"compute_and_write_s": [0-9.]*' | paste - - - -
  echo -n "k_crc32 "; PG_BGZF_CRC_FOLD=0 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*\|"compute_and_write_s": [0-9.]*' | paste - - - -
done | tee $O/t2_whole_crc_fold_ab.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o whole_fold --output-format csv -- $CMD > $O/prof_fold.log 2>&1
head -12 $O/prof/whole_fold_kernel_stats.csv | cut -c1-60,200-290
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
