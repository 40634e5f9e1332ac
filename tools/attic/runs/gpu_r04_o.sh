#!/bin/bash
# round 4, call o: one rank's share of configs[4] at N = 8 (150 GB resident, 7500 windows) on one GPU, as in round 3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04o; mkdir -p $O
timeout 900 python tools/c5_share.py 3 > $O/c5_share.txt 2>&1; tail -15 $O/c5_share.txt
