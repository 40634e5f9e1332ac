#!/bin/bash
# round 6, call v: k_deflate with the lanes' group masks in LDS (no 64-step rank loop) and the tokens made by all lanes -- tests, rate,
# the VCF drop-in's wall-clock with the process ending at os._exit
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06v; mkdir -p $O
PG_DEFLATE_FUZZ_SEEDS=600 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 2>&1 | tail -2
timeout 600 python tools/deflate_bench.py 400000 200 > $O/deflate_bench.json 2> $O/err.txt; cat $O/deflate_bench.json
VCF_LEGS=0,2 VCF_REPS=3 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2> $O/vcf_bench.err; tail -c 1700 $O/vcf_bench_6GB.json; echo
timeout 600 python -m pytest tests/test_gpu_vcf.py -q -n 4 2>&1 | tail -2
