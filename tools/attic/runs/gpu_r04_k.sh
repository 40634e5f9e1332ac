#!/bin/bash
# round 4, call k: where the compute thread's 0.1 s goes in a T2 run from packed input (first chunk against the others)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04k; mkdir -p $O
timeout 900 python tools/t2_pgeno_bench.py 25000000 200 2>&1 | grep -v zlib > $O/t2_pgeno.txt; cat $O/t2_pgeno.txt
