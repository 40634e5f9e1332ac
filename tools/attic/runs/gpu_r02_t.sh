#!/bin/bash
# round 2, call t: pair counts on the matrix cores -- parity, then timing against the popcount kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02t
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02t/pytest.log 2>&1; tail -15 gpurun_out/r02t/pytest.log
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > gpurun_out/r02t/$tag.json 2> gpurun_out/r02t/$tag.err
  python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r02t/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s ms_per_step %.4f  kernels %s" % (sys.argv[1], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "failed", e, open('gpurun_out/r02t/%s.err' % sys.argv[1]).read()[-600:])
PY
}
for wl in northstar c2 c4; do
  run ${wl}_mfma $wl PG_X=1
  run ${wl}_valu $wl PG_PAIR_VALU=1
done
