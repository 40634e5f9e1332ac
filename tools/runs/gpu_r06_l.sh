#!/bin/bash
# round 6, call l: the evidence at the round's last HEAD (after the CRC-32 inside k_inflate and its direct table): the whole -m gpu
# suite, smoke(), 60 000 random members, the driver's command plain and under rocprofv3, the inflate kernels' counter passes, the whole
# north star under the kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06l; mkdir -p $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/prof
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 | tee $O/smoke.txt
timeout 1200 python tools/inflate_fuzz.py 60000 7 > $O/inflate_fuzz_60000_members.txt 2>&1; tail -1 $O/inflate_fuzz_60000_members.txt | cut -c1-150
timeout 1500 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; tail -c 400 $O/bench_northstar_default.json; echo
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o default --output-format csv -- python bench.py > $O/bench_default_under_rocprof.json 2> $O/bench_default_under_rocprof.err
python tools/prof_timed_region.py $O/prof/default_kernel_trace.csv 10 2 > $O/northstar_default_timed_region_kernel_stats.csv; head -6 $O/northstar_default_timed_region_kernel_stats.csv
S=/tmp/pg_r06l; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> $O/bgzip.txt
B="python tools/inflate_bench.py --file $S/sample.geno.gz"
$B > $O/inflate_bench.json 2>&1; cat $O/inflate_bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o inflate --output-format csv -- $B > $O/bench_prof_inflate.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o inflate --output-format csv -- $B > $O/pmc_fetch_inflate.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o inflate --output-format csv -- $B > $O/pmc_write_inflate.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o inflate --output-format csv -- $B > $O/pmc_sq_inflate.log 2>&1
rm -rf $S
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 4 > $O/t2_northstar_whole_bgzf.json 2> $O/whole.err; tail -c 600 $O/t2_northstar_whole_bgzf.json; echo
CMD=$(cat /tmp/pg_ns_cmd.txt)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o whole --output-format csv -- $CMD > $O/prof_whole.log 2>&1
head -10 $O/prof/whole_kernel_stats.csv | cut -c1-60,200-290
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
