#!/bin/bash
# round 3, call p: pack beside the pair kernels again (PG_OVERLAP), now that the called-count kernel is bound by the matrix pipe;
# the micro-benchmark in its asm-only form
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
timeout 120 ./gpurun_variants/mfma_rate > $O/mfma_rate.txt 2>&1; cat $O/mfma_rate.txt
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-26s ms_per_step %.4f pipelined %s kernels %s" % (sys.argv[2], d["ms_per_step"], d.get("ms_per_step_pipelined_sub_batches"), d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
run ns      northstar PG_PLACE_TRIALS=1
run ns_ov   northstar PG_PLACE_TRIALS=1 PG_OVERLAP=1
run ns2     northstar PG_PLACE_TRIALS=1
run ns_ov2  northstar PG_PLACE_TRIALS=1 PG_OVERLAP=1
