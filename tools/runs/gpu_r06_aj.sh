#!/bin/bash
# round 6, call aj: which test hangs with the reader thread submitting (per-test timeout with the threads' stacks)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06aj; mkdir -p $O
timeout 420 python -m pytest tests/test_gpu_vcf.py tests/test_gpu_deflate.py -q -x --timeout=45 --timeout-method=thread -p no:cacheprovider > $O/pytest.log 2>&1; tail -80 $O/pytest.log | cut -c1-220
