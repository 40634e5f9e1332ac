#!/bin/bash
# round 6, call m: the last HEAD (BGZF writer on the library's own compressor): the whole -m gpu suite, the VCF drop-in on 6 GB of VCF
# (its writer was the bound), the drivers end to end, the default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06m; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -1 | tee $O/smoke.txt
VCF_LEGS=0,1 VCF_REPS=2 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2> $O/vcf_bench.err; tail -c 1300 $O/vcf_bench_6GB.json; echo
PG_BGZF_ZLIB=1 VCF_LEGS=0 VCF_REPS=2 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB_zlib_writer.json 2>> $O/vcf_bench.err; tail -c 700 $O/vcf_bench_6GB_zlib_writer.json; echo
timeout 900 python tools/drivers_bench.py 5000000 200 > $O/drivers_bench_gpu.json 2> $O/drivers_bench.err; tail -c 500 $O/drivers_bench_gpu.json; echo
timeout 1500 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; tail -c 300 $O/bench_northstar_default.json; echo
du -sh $O
