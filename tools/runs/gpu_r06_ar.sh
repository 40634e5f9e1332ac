#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PG_VCF_FUZZ_SEEDS=200 timeout 600 python -m pytest tests/test_gpu_vcf.py -q -n 6 --timeout=120 2>&1 | tail -8
