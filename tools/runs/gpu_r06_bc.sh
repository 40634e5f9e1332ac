#!/bin/bash
# round 6, call bc: the drivers' GPU fuzz (HIP engine == stand-in) once more at the round's last commit, after the host-side fixes the
# differentials against the reference asked for
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06bc; mkdir -p $O
PG_FUZZ_SEEDS=2000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 --timeout=600 2>&1 | tail -1 | tee $O/gpu_fuzz_2000_seeds_last_commit.txt
