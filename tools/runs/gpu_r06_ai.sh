#!/bin/bash
# round 6, call ai: the VCF drop-in with the blocks submitted to the device by the reader thread
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ai; mkdir -p $O
PG_VCF_FUZZ_SEEDS=200 timeout 900 python -m pytest tests/test_gpu_vcf.py tests/test_gpu_deflate.py -q -n 6 2>&1 | tail -2
VCF_LEGS=0,2 VCF_REPS=3 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2> $O/vcf_bench.err; cut -c1-1700 $O/vcf_bench_6GB.json; echo
