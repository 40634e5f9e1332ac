#!/bin/bash
# round 3, call y: a kernel that only moves the pack kernel's bytes (tools/ubench/pack_rw.hip) next to k_pack3 on the same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03y; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/ubench/pack_rw.hip -o /tmp/pack_rw 2>/dev/null || cp gpurun_variants/pack_rw /tmp/pack_rw
for k in 1 2 3; do
  timeout 250 /tmp/pack_rw | tee -a $O/pack_rw.txt
  PG_PLACE_TRIALS=1 timeout 200 python bench.py --workload northstar --steps 5 --warmup 2 --no-cpu-baseline --no-tiers 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  k_pack3 on this box (fresh allocation of the rows): %.3f ms (pass %.3f ms)' % (d['kernel_ms_per_step']['k_pack3'], d['ms_per_step']))" | tee -a $O/pack_rw.txt
done
