#!/bin/bash
# round 2, call s: pack kernel and pair kernels side by side on disjoint CU sets (PG_CU_PACK), north-star shape
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02s
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --workload northstar --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > gpurun_out/r02s/$tag.json 2> gpurun_out/r02s/$tag.err
  python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r02s/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s ms_per_step %.3f  kernels %s" % (sys.argv[1], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "failed", e, open('gpurun_out/r02s/%s.err' % sys.argv[1]).read()[-300:])
PY
}
run single_batch PG_X=1
run overlap8 PG_OVERLAP=1
run overlap16 PG_OVERLAP=1 PG_SUBBATCHES=16
for cu in 64 96 112 128 144 160; do
  run cu${cu}_sub8 PG_OVERLAP=1 PG_CU_PACK=$cu
  run cu${cu}_sub16 PG_OVERLAP=1 PG_CU_PACK=$cu PG_SUBBATCHES=16
done
run single_batch_again PG_X=1
