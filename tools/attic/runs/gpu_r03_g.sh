#!/bin/bash
# round 3, call g: ring shapes of k_pairC_tile (PG_TILE_CFG 0/1/2) and the barrier-free one-wave form k_pairC_wave (3: ring of 3 pairs, 4: of 4)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03g; mkdir -p $O
for cfg in 1 3 4; do
  PG_TILE_CFG=$cfg timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_$cfg.log 2>&1; echo "cfg $cfg: $(grep -E 'passed|failed' $O/pytest_$cfg.log | tail -1)"
done
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
for cfg in 0 1 2 3 4; do run ns_cfg$cfg northstar PG_TILE_CFG=$cfg PG_PLACE_TRIALS=1; done
run ns_none northstar PG_PAIR_TILE=none PG_PLACE_TRIALS=1
for cfg in 0 3; do run c2_cfg$cfg c2 PG_TILE_CFG=$cfg; done
run c2_none c2 PG_PAIR_TILE=none
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --workload northstar --steps 2 --warmup 1 --no-cpu-baseline --no-tiers"
for cfg in 3; do
PG_TILE_CFG=$cfg PG_PLACE_TRIALS=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $O/pmc_sq$cfg -o ns --output-format csv -- $B > $O/pmc_sq$cfg.log 2>&1
PG_TILE_CFG=$cfg PG_PLACE_TRIALS=1 timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch$cfg -o ns --output-format csv -- $B > $O/pmc_fetch$cfg.log 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in sorted(glob.glob("%s/pmc_*/**/*counter_collection.csv" % O, recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "pairC" in k:
            print(f.split("/")[-2], k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
