#!/bin/bash
# round 2, call w: pair-count kernels on the matrix cores, int8 (default) and MX fp4 (PG_PAIR_FP4=1): parity, timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02w
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02w/pytest_i8.log 2>&1; tail -3 gpurun_out/r02w/pytest_i8.log | cut -c1-300
PG_PAIR_FP4=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02w/pytest_fp4.log 2>&1; tail -3 gpurun_out/r02w/pytest_fp4.log | cut -c1-300
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > gpurun_out/r02w/$tag.json 2> gpurun_out/r02w/$tag.err
  python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r02w/%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s ms_per_step %.4f  kernels %s" % (sys.argv[1], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "failed", e, open('gpurun_out/r02w/%s.err' % sys.argv[1]).read()[-600:])
PY
}
for wl in northstar c2 c4; do
  run ${wl}_i8 $wl PG_X=1
  run ${wl}_fp4 $wl PG_PAIR_FP4=1
done
run northstar_i8_again northstar PG_X=1
run northstar_fp4_again northstar PG_PAIR_FP4=1
