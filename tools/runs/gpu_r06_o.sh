#!/bin/bash
# round 6, call o: k_deflate for the first time -- its tests, its rate, the VCF drop-in writing .geno.gz with it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_deflate.py -x -q > $O/pytest_deflate.log 2>&1; tail -25 $O/pytest_deflate.log
timeout 600 python tools/deflate_bench.py 400000 200 > $O/deflate_bench.json 2> $O/deflate_bench.err; cat $O/deflate_bench.json; tail -3 $O/deflate_bench.err
timeout 600 python -m pytest tests/test_gpu_vcf.py -x -q 2>&1 | tail -3
VCF_LEGS=0 VCF_REPS=2 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB_device_deflate.json 2> $O/vcf_bench.err; tail -c 1500 $O/vcf_bench_6GB_device_deflate.json; echo; tail -5 $O/vcf_bench.err
