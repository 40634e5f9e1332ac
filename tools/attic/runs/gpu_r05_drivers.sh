#!/bin/bash
# round 5: every drop-in driver end to end on one bgzipped sample (5e6 sites x 200 diploids: 4.1 GB of text), after the tests of the
# pieces that changed for it (pg_site_target, freq.py on BGZF spans, the matrix text of distMat.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05drv; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_inflate.py -m gpu -x -q -n 4 -k "site_target or site_counts or freq or distmat or vcf" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python tools/drivers_bench.py ${DRV_SITES:-5000000} 200 > $O/drivers_bench.json 2> $O/drivers_bench.err
cat $O/drivers_bench.json; tail -3 $O/drivers_bench.err
