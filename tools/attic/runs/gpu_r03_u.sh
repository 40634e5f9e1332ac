#!/bin/bash
# round 3, call u3: k_popdist_fin with one reciprocal per individual pair (same summation order with per-haplotype counts)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03u; mkdir -p $O
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
run ns northstar PG_PLACE_TRIALS=1
run ns_nodip northstar PG_PLACE_TRIALS=1 PG_NO_DIP=1
run c2 c2
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^E |FAILED" $O/pytest.log | head
