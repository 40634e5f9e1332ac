#!/bin/bash
# round 6, call h: the evidence at the round's HEAD -- the whole -m gpu suite and smoke(); the driver's command (python bench.py), plain
# and under rocprofv3 (timed region cut out by tools/prof_timed_region.py); kernel statistics + FETCH / WRITE / SQ counter passes of
# the four workloads and of the inflate kernels; the whole north star under the kernel trace; the other workloads' bench lines;
# c5_share; the 1000-seed GPU fuzz and 60 000 random members through k_inflate; the drivers and the VCF drop-in end to end
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06h; mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/prof
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; tail -c 600 $O/bench_northstar_default.json; echo
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o default --output-format csv -- python bench.py > $O/bench_default_under_rocprof.json 2> $O/bench_default_under_rocprof.err
python tools/prof_timed_region.py $O/prof/default_kernel_trace.csv 10 2 > $O/northstar_default_timed_region_kernel_stats.csv; head -6 $O/northstar_default_timed_region_kernel_stats.csv
export PG_PLACE_TRIALS=1
for wl in northstar c2 c3 c4; do
  ST=5; [ $wl = northstar ] && ST=3
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline --no-tiers"
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o $wl --output-format csv -- $B > $O/bench_prof_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $wl --output-format csv -- $B > $O/pmc_fetch_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $wl --output-format csv -- $B > $O/pmc_write_$wl.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o $wl --output-format csv -- $B > $O/pmc_sq_$wl.log 2>&1
done
unset PG_PLACE_TRIALS
S=/tmp/pg_r06h; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> $O/bgzip.txt
B="python tools/inflate_bench.py --file $S/sample.geno.gz"
$B > $O/inflate_bench.json 2>&1; cat $O/inflate_bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o inflate --output-format csv -- $B > $O/bench_prof_inflate.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o inflate --output-format csv -- $B > $O/pmc_fetch_inflate.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o inflate --output-format csv -- $B > $O/pmc_write_inflate.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o inflate --output-format csv -- $B > $O/pmc_sq_inflate.log 2>&1
rm -rf $S
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 4 > $O/t2_northstar_whole_bgzf.json 2> $O/whole.err; tail -c 900 $O/t2_northstar_whole_bgzf.json; echo
CMD=$(cat /tmp/pg_ns_cmd.txt)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o whole --output-format csv -- $CMD > $O/prof_whole.log 2>&1
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
for wl in c2 c3 c4; do
  timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-tiers > $O/bench_$wl.json 2> $O/bench_$wl.err; tail -c 250 $O/bench_$wl.json; echo
done
timeout 900 python tools/c5_share.py 3 > $O/c5_share.txt 2>&1; tail -3 $O/c5_share.txt
PG_FUZZ_SEEDS=1000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 8 > $O/gpu_fuzz_1000_seeds.txt 2>&1; tail -2 $O/gpu_fuzz_1000_seeds.txt
timeout 1200 python tools/inflate_fuzz.py 60000 7 > $O/inflate_fuzz_60000_members.txt 2>&1; tail -2 $O/inflate_fuzz_60000_members.txt
timeout 900 python tools/drivers_bench.py 5000000 200 > $O/drivers_bench_gpu.json 2> $O/drivers_bench.err; tail -c 700 $O/drivers_bench_gpu.json; echo
VCF_LEGS=0,1 VCF_REPS=2 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2> $O/vcf_bench.err; tail -c 900 $O/vcf_bench_6GB.json; echo
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
