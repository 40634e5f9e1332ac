#!/bin/bash
# round 3, call v: the text tier with the host threads a rank has at N = 8 (256 / 8 = 32) next to all of them
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03v; mkdir -p $O
export PG_PLACE_TRIALS=1
timeout 900 python tools/t2_bench.py 2000000 200 > $O/t2_all_threads.txt 2>&1; grep -E "device tokenizer run|windows/s end to end" $O/t2_all_threads.txt | cut -c1-420
PG_HOST_THREADS=32 timeout 900 python tools/t2_bench.py 2000000 200 > $O/t2_32_threads.txt 2>&1; grep -E "device tokenizer run|windows/s end to end" $O/t2_32_threads.txt | cut -c1-420
PG_HOST_THREADS=16 timeout 900 python tools/t2_bench.py 2000000 200 > $O/t2_16_threads.txt 2>&1; grep -E "device tokenizer run" $O/t2_16_threads.txt | cut -c1-420
