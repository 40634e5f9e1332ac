#!/bin/bash
# round 6, call an: k_tok_cells3 asking for the next line's bytes before it works on this line's -- tests and the kernel's time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06an; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_e2e.py tests/test_gpu_kernels.py tests/test_gpu_inflate.py tests/test_gpu_fuzz.py -q -n 6 --timeout=300 2>&1 | tail -2
PG_NS_KEEP=/tmp/ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 2 > $O/t2_northstar.json 2> $O/err.txt; cut -c1-900 $O/t2_northstar.json; echo
CMD=$(cat /tmp/ns_cmd.txt)
PG_PLACE_TRIALS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t2 --output-format csv -- $CMD > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/t2_whole_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.csv" -size +2M -delete
head -8 $O/t2_whole_kernel_stats.csv | cut -c1-60,150-330
