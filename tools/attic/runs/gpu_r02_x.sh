#!/bin/bash
# round 2, call x: fp4 pair kernels after a change: parity (kernel + full-size tests), timing on the three shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02x
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02x/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r02x/pytest.log | tail -2
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > gpurun_out/r02x/$tag.json 2> gpurun_out/r02x/$tag.err
  python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r02x/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s ms_per_step %.4f  kernels %s" % (sys.argv[1], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "failed", e, open('gpurun_out/r02x/%s.err' % sys.argv[1]).read()[-600:])
PY
}
for wl in northstar c2 c4 northstar; do run ${wl}_fp4 $wl PG_X=1; done
