#!/bin/bash
# round 3, call j: k_pairC_big (one wave per SIMD, 14 tiles per wave, generated main loop): parity, then A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03j; mkdir -p $O
PG_PAIR_TILE=b timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "pairwise or code_path or half_missing" > $O/pytest_k.log 2>&1; echo "kernels: $(grep -E 'passed|failed' $O/pytest_k.log | tail -1)"; grep -E "^E |Error|assert" $O/pytest_k.log | head -8
PG_PAIR_TILE=b timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_f.log 2>&1; echo "fullsize: $(grep -E 'passed|failed' $O/pytest_f.log | tail -1)"; grep -E "^E |Error" $O/pytest_f.log | head -8
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
run ns_b2   northstar PG_PAIR_TILE=b PG_PLACE_TRIALS=1
run c2_b    c2 PG_PAIR_TILE=b
run c2_none c2 PG_PAIR_TILE=none
