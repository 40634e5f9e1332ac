#!/bin/bash
# round 3, call w: the extra workloads after the kernel changes of the round (5 kb windows, popFreq, C3, C4)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03w; mkdir -p $O
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-26s ms_per_step %.4f value %.1f kernels %s" % (sys.argv[2], d["ms_per_step"], d["value"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
run c2_w5k c2_w5k
run c2_w5k_tile c2_w5k PG_PAIR_TILE=c
run popfreq popfreq
run c3 c3
run c4 c4
run tiny tiny
