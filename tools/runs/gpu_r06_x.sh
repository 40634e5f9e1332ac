#!/bin/bash
# round 6, call x: --excludeDuplicates on the device (k_vcf_lastkey carries the key from block to block)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06x; mkdir -p $O
PG_VCF_FUZZ_SEEDS=400 timeout 1500 python -m pytest tests/test_gpu_vcf.py -q -n 8 2>&1 | tail -15 | tee $O/vcf_device_parser_tests_with_duplicates.txt
