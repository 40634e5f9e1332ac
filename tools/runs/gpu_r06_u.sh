#!/bin/bash
# round 6, call u: the device's VCF parser against the host parser on 800 random files x option sets, k_deflate on 3000 random texts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06u; mkdir -p $O
PG_VCF_FUZZ_SEEDS=800 timeout 1500 python -m pytest tests/test_gpu_vcf.py -q -n 8 -k "random_files or damaged or gatk" 2>&1 | tail -3 | tee $O/vcf_device_parser_fuzz_800_seeds.txt
PG_DEFLATE_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_deflate.py -q -n 8 -k "random_texts" 2>&1 | tail -3 | tee $O/deflate_fuzz_3000_seeds.txt
