#!/bin/bash
# round 6, call am: k_tok_cells3 with a lane's column-table entries in registers and a line's loads side by side -- the tokenizer's tests,
# the kernel's time by rocprofv3 with and without it (PG_TOK_CELLS_REGS=0), the whole north star alternating
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06am; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_e2e.py tests/test_gpu_kernels.py tests/test_gpu_inflate.py -q -n 6 --timeout=300 2>&1 | tail -2
PG_NS_KEEP=/tmp/ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 1 > $O/t2_northstar_first.json 2> $O/err.txt; cut -c1-300 $O/t2_northstar_first.json; echo
CMD=$(cat /tmp/ns_cmd.txt)
for k in 1 2 3 4; do for v in 0 1; do
  PG_TOK_CELLS_REGS=$v PG_TIMING=1 PG_PLACE_TRIALS=1 $CMD 2>&1 >/dev/null | grep PG_TIMING | python -c "
import sys, json
t = json.loads(sys.stdin.read().split('PG_TIMING ', 1)[1])
print('cells_in_regs=$v', {k: round(t[k], 4) for k in ('total_s', 'context_s', 'tokenize_s', 'prep_wait_s', 'main_stats_s', 'tokenizer_kernels_s') if k in t})"
done; done | tee $O/t2_whole_cells_regs_ab.txt
for v in 0 1; do
  PG_TOK_CELLS_REGS=$v PG_PLACE_TRIALS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof$v -o t2 --output-format csv -- $CMD > $O/prof$v.log 2>&1
  find $O/prof$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/t2_whole_kernel_stats_cells_regs_$v.csv
  find $O/prof$v -name "*kernel_trace.csv" -delete; find $O/prof$v -name "*.csv" -size +2M -delete
  echo "== PG_TOK_CELLS_REGS=$v"; head -6 $O/t2_whole_kernel_stats_cells_regs_$v.csv | cut -c1-60,150-330
done
