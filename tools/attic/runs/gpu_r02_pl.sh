#!/bin/bash
# round 2, call pl: placement trials at reserve(): alternating fresh processes with (default 4) and without (PG_PLACE_TRIALS=1)
cd $GRAFT_REPO_ROOT
WL=${1:-northstar}; ST=5; [ $WL = c2 ] && ST=20
for rep in 1 2 3 4; do
  for tr in 4 1; do
    PG_PLACE_TRIALS=$tr timeout 300 python bench.py --workload $WL --steps $ST --warmup 2 --no-cpu-baseline --no-tiers 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$WL trials=$tr rep $rep', d['ms_per_step'], d.get('kernel_ms_per_step'), d.get('placement_trials', {}).get('probe_ms'), d.get('placement_trials', {}).get('kept'))"
  done
done
