#!/bin/bash
# round 3, call z2: the called plane as narrow as the pack kernel writes it, A/B on one box (PG_VP_WIDE=1: whole tiles as before)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; mkdir -p $O
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-26s ms_per_step %.4f kernels %s" % (sys.argv[2], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
for k in 1 2 3 4; do
  run ns_narrow$k northstar PG_PLACE_TRIALS=1
  run ns_wide$k northstar PG_PLACE_TRIALS=1 PG_VP_WIDE=1
done
run c2_narrow c2
run c2_wide c2 PG_VP_WIDE=1
run c2_narrow2 c2
run c2_wide2 c2 PG_VP_WIDE=1
