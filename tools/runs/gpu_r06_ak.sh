#!/bin/bash
# round 6, call ak: the VCF drop-in with the blocks submitted to the device by the reader thread (after the shutdown fix)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ak; mkdir -p $O
PG_VCF_FUZZ_SEEDS=300 timeout 600 python -m pytest tests/test_gpu_vcf.py tests/test_gpu_deflate.py -q -n 6 --timeout=120 2>&1 | tail -3
VCF_LEGS=0,2 VCF_REPS=3 timeout 600 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2> $O/vcf_bench.err; cut -c1-1700 $O/vcf_bench_6GB.json; echo
