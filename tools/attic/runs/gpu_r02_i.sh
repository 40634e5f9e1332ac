#!/bin/bash
# GPU call I: suite with the run-time compaction group and the unrolled k_popdist_fin; bench lines.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i
mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|Error" $O/pytest_gpu.log | tail -3
show='
import sys, json
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln); print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"].get("frac"), d.get("kernel_ms_per_step"))
    elif "rror" in ln: print(ln.strip())
'
for wl in northstar c2 c2_w5k c4; do
  for v in PG_NONE=1 PG_GROUP_WORDS=64; do
    echo "== $wl $v"
    env $v timeout 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-tiers 2>&1 | tee -a $O/ab_group_$wl.log | python -c "$show"
  done
done
