#!/bin/bash
# round 3, call h: one-wave pair kernels with the unscaled matrix instruction and asymmetric fragment forms -- parity, A/B against the tile kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03h; mkdir -p $O
PG_PAIR_TILE=none timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_none.log 2>&1; echo "none: $(grep -E 'passed|failed' $O/pytest_none.log | tail -1)"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_c.log 2>&1; echo "default: $(grep -E 'passed|failed' $O/pytest_c.log | tail -1)"
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
for rep in 1 2; do
run ns_none_$rep northstar PG_PAIR_TILE=none PG_PLACE_TRIALS=1
run ns_c_$rep    northstar PG_PAIR_TILE=c PG_PLACE_TRIALS=1
done
run c2_none c2 PG_PAIR_TILE=none
run c2_c    c2 PG_PAIR_TILE=c
run c4      c4 PG_X=1
