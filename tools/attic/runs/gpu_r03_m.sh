#!/bin/bash
# round 3, call m: k_pairC_big with the bookkeeping inside the products' shadow, ring of 8 pairs: kernel tests, then timings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -5
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-26s ms_per_step %.4f  kernels %s" % (sys.argv[2], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
run ns_b    northstar PG_PLACE_TRIALS=1
run ns_b2   northstar PG_PLACE_TRIALS=1
run c2_b    c2
run c4_b    c4
