#!/bin/bash
# round 3, call a: LDS-staged pair kernels (pg_pair_tile.hip) -- parity (whole GPU suite), then A/B against the one-wave kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-22s ms_per_step %.4f  kernels %s" % (sys.argv[2], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
for wl in northstar c2 c4; do
  run ${wl}_tile $wl PG_X=1
  run ${wl}_onewave $wl PG_PAIR_TILE=none
done
run northstar_tile2 northstar PG_X=1
