#!/bin/bash
# round 6, call n: the device's VCF parser for the first time -- its GPU tests, the VCF drop-in on 6 GB of VCF with it and without it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vcf.py tests/test_gpu_inflate.py -x -q > $O/pytest_vcf.log 2>&1; tail -15 $O/pytest_vcf.log
VCF_LEGS=0,2 VCF_REPS=2 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB_device_parser.json 2> $O/vcf_bench.err; tail -c 2500 $O/vcf_bench_6GB_device_parser.json; echo; tail -5 $O/vcf_bench.err
PG_VCF_DEVICE=0 VCF_LEGS=0,2 VCF_REPS=1 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB_host_parser.json 2>> $O/vcf_bench.err; tail -c 1500 $O/vcf_bench_6GB_host_parser.json; echo
