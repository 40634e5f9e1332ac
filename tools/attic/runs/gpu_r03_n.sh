#!/bin/bash
# round 3, call n: where k_pairC_big's time goes -- builds with one class of loop instructions dropped (wrong results, timing only)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03n; mkdir -p $O
export PG_PLACE_TRIALS=1
for v in base B V BV C R X M BVCR; do
  lib=$PWD/gpurun_variants/lib_$v.so; [ $v = base ] && lib=$PWD/genomics_general_amd/libpopgen_hip.so
  PG_LIBRARY=$lib timeout 120 python bench.py --workload northstar --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/$v.json 2> $O/$v.err
  python - $O/$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-6s pairC %.4f  (step %.3f)" % (sys.argv[2], d["kernel_ms_per_step"]["k_pairC_big"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
