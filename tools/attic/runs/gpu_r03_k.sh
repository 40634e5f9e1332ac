#!/bin/bash
# round 3, call k: k_pairC_big after the LDS-DMA offset fix: single-missing-site probes, parity, A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03k; mkdir -p $O
for args in "37 1250 33" "70 3000 40" "200 6000 150"; do echo "== $args"; PG_PAIR_TILE=b timeout 120 python tools/debug_pairc.py $args 2>&1 | grep -v " 0 wrong" | tail -5; done
PG_PAIR_TILE=b timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -k "pairwise or code_path or half_missing or c2 or additive or independent or northstar or three_and" 2>&1 | tail -8
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
run ns_b    northstar PG_PAIR_TILE=b PG_PLACE_TRIALS=1
run ns_c    northstar PG_PAIR_TILE=c PG_PLACE_TRIALS=1
run c2_b    c2 PG_PAIR_TILE=b
run c2_c    c2 PG_PAIR_TILE=c
run c3_b    c3 PG_PAIR_TILE=b
run c3_c    c3 PG_PAIR_TILE=c
