# round 2, call kp: k_pairC_fp4 with its word range cut into 1..16 parts (PG_MFMA_KPARTS_C was an experiment hook of that
# build; the rule it led to -- parts of at most 1 MiB of plane -- is in pg_launch_pairC_mfma, the hook is gone)
cd $GRAFT_REPO_ROOT
for kp in 1 2 4 8 16; do
  PG_MFMA_KPARTS_C=$kp timeout 300 python bench.py --workload northstar --steps 5 --warmup 2 --no-cpu-baseline --no-tiers 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('kparts_C=$kp', d['ms_per_step'], d.get('kernel_ms_per_step'))"
done
