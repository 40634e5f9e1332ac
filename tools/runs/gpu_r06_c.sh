#!/bin/bash
# round 6, call c: the new --inferPloidy goldens on the real engine; where the start-up of a run goes (tools/ctx_time.py, fresh
# processes, with and without deferred code-object loading); PG_PACK_PERM as a same-process A/B; c5_share (150 GB resident) at HEAD;
# the inflate kernels' counter passes for profiles/r06
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06c; mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout 900 python -m pytest tests -m gpu -x -q -k "ploidy" > $O/pytest_ploidy.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest_ploidy.log | tail -8
for k in 1 2 3; do timeout 120 python tools/ctx_time.py 2>/dev/null | tail -1; done > $O/ctx_time.txt
for k in 1 2; do HIP_ENABLE_DEFERRED_LOADING=0 timeout 120 python tools/ctx_time.py 2>/dev/null | tail -1; done >> $O/ctx_time.txt
for k in 1 2; do GPU_MAX_HW_QUEUES=2 timeout 120 python tools/ctx_time.py 2>/dev/null | tail -1; done >> $O/ctx_time.txt
cut -c1-900 $O/ctx_time.txt
timeout 300 python tools/ab_env.py PG_PACK_PERM=8 northstar 5 > $O/ab_pack_perm8_northstar.txt 2>&1; tail -6 $O/ab_pack_perm8_northstar.txt
timeout 300 python tools/ab_env.py PG_PACK_PERM=8 c2 8 > $O/ab_pack_perm8_c2.txt 2>&1; tail -6 $O/ab_pack_perm8_c2.txt
timeout 300 python tools/ab_env.py PG_PACK_PERM=32 northstar 5 > $O/ab_pack_perm32_northstar.txt 2>&1; tail -4 $O/ab_pack_perm32_northstar.txt
timeout 900 python tools/c5_share.py 3 > $O/c5_share.txt 2>&1; tail -5 $O/c5_share.txt
S=/tmp/pg_r06_sample; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> $O/bgzip.txt; cat $O/bgzip.txt
B="python tools/inflate_bench.py --file $S/sample.geno.gz"
$B > $O/inflate_bench.json 2>&1; cat $O/inflate_bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o inflate --output-format csv -- $B > $O/bench_prof_inflate.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o inflate --output-format csv -- $B > $O/pmc_fetch_inflate.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o inflate --output-format csv -- $B > $O/pmc_write_inflate.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o inflate --output-format csv -- $B > $O/pmc_sq_inflate.log 2>&1
rm -rf $S
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
