#!/bin/bash
# round 5: the pack kernel on the C2 shape (10^7 sites x 100 diploids: one-wave blocks, 52 of 64 lanes with data, a 0.44 ms launch):
# compaction group size x LDS cells per thread x burst, same box, PG_PLACE_TRIALS=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05c2; mkdir -p $O
export PG_PLACE_TRIALS=1
for rep in 1 2; do
for cells in 24 12; do
for grp in 16 24 32 48 64 128; do
  PG_PACK_CELLS=$cells PG_GROUP_WORDS=$grp python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline --no-tiers 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('cells $cells grp %3d rep $rep: k_pack3 %.4f ms frac %.3f  step %.4f ms' % ($grp, r['avg_launch_ms'], r['frac'], d['ms_per_step']))"
done; done
PG_PACK_BURST=0 python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline --no-tiers 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('no burst         rep $rep: k_pack3 %.4f ms frac %.3f  step %.4f ms' % (r['avg_launch_ms'], r['frac'], d['ms_per_step']))"
done | tee $O/c2_pack_grid.txt
