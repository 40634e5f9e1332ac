#!/bin/bash
# round 4, call h: tier T2 from `.pgeno` (raw and deflated cells) as it is today
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04h; mkdir -p $O
timeout 900 python tools/t2_pgeno_bench.py 25000000 200 > $O/t2_pgeno.txt 2>&1; cat $O/t2_pgeno.txt
