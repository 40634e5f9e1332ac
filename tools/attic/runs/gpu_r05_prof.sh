#!/bin/bash
# round 5: the whole -m gpu suite, the profile passes behind profiles/r05 (kernel stats + three counter passes per workload;
# PG_PLACE_TRIALS=1: no placement probes, so every launch of a kernel in a pass is a real step), the inflate kernels alone and inside a
# T2 run on bgzip-compressed text, then the un-profiled bench lines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05prof; mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -4
export PG_PLACE_TRIALS=1
for wl in northstar c2 c3 c4; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  tail -1 $O/bench_prof_$wl.log | cut -c1-160
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
done
unset PG_PLACE_TRIALS
# ---- the inflate kernels: 1 GiB of north-star text, bgzipped at level 6 (workload name: inflate) ----
S=/tmp/pg_r05_sample; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> $O/bgzip.txt; cat $O/bgzip.txt
B="python tools/inflate_bench.py --file $S/sample.geno.gz"
$B > $O/inflate_bench.json 2>&1; cat $O/inflate_bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o inflate --output-format csv -- $B > $O/bench_prof_inflate.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o inflate --output-format csv -- $B > $O/pmc_fetch_inflate.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o inflate --output-format csv -- $B > $O/pmc_write_inflate.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o inflate --output-format csv -- $B > $O/pmc_sq_inflate.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU -d $O/pmc_sq2 -o inflate --output-format csv -- $B > $O/pmc_sq2_inflate.log 2>&1
# ---- a T2 run on the bgzipped text (8.1 GB of text, 0.32 GB compressed) under the kernel trace, and its timing without the profiler ----
CMD=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/sample.geno.gz #; s#$S/sample.geno.csv#$S/out_gz.csv#")
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o t2_bgzf --output-format csv -- $CMD > $O/t2_bgzf_prof.log 2>&1
for k in 1 2 3; do PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 | grep PG_TIMING | cut -c1-1200; done > $O/t2_bgzf_8GB_timing.txt; cat $O/t2_bgzf_8GB_timing.txt | cut -c1-400
CMDT=$(cat $S/cmd.txt)
for k in 1 2; do PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDT 2>&1 | grep PG_TIMING | cut -c1-1200; done > $O/t2_text_8GB_timing.txt
cmp $S/sample.geno.csv $S/out_gz.csv && echo "csv of the bgzf run == csv of the text run" | tee -a $O/t2_bgzf_8GB_timing.txt
rm -rf $S
# ---- the un-profiled lines ----
for wl in c2 c3 c4; do
  timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-tiers > $O/bench_$wl.json 2> $O/bench_$wl.err
  tail -c 300 $O/bench_$wl.json; echo
done
timeout 1200 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; tail -c 400 $O/bench_northstar_default.json; echo
ls $O/prof_stats | head -30; du -sh $O
