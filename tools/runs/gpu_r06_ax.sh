#!/bin/bash
# round 6, call ax: pg_tune_planes -- the new test, and what the choice of the planes is worth on the north-star shape (three fresh processes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ax; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "other_planes" --timeout=200 2>&1 | tail -3
for k in 1 2 3; do timeout 400 python tools/plane_placement.py 6 2>&1 | tail -1 | tee -a $O/plane_placement_northstar.txt; done
