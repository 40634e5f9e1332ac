#!/bin/bash
# round 3, call d: pruned tree (no v1 / int8 kernels), k_pack3 up to 4096 slots, bench.py's own launcher -- whole GPU suite, C4 A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03d; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
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
run c4_pack3   c4 PG_X=1
run c4_pack2   c4 PG_PACK2=1
run c4_pack3_b c4 PG_X=1
run ns         northstar PG_X=1
