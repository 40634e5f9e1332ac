#!/bin/bash
# round 3, call t2: words per pack block chosen by the launcher (32 for small jobs): C2, C4, tiny; kernel tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03t; mkdir -p $O
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-26s ms_per_step %.4f kernels %s" % (sys.argv[2], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
run c2_auto c2
run c2_g64 c2 PG_GROUP_WORDS=64
run c2_auto2 c2
run c4_auto c4
run c4_g64 c4 PG_GROUP_WORDS=64
run c4_auto2 c4
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -2
