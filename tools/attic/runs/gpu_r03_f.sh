#!/bin/bash
# round 3, call f: producer / consumer form of the called-count kernel (PG_PAIR_TILE=s) -- parity, then A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03f; mkdir -p $O
PG_PAIR_TILE=s timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
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
run ns_s     northstar PG_PAIR_TILE=s PG_PLACE_TRIALS=1
run ns_c     northstar PG_PAIR_TILE=c PG_PLACE_TRIALS=1
run ns_none  northstar PG_PAIR_TILE=none PG_PLACE_TRIALS=1
run ns_s2    northstar PG_PAIR_TILE=s PG_PLACE_TRIALS=1
run c2_s     c2 PG_PAIR_TILE=s
run c2_none  c2 PG_PAIR_TILE=none
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --workload northstar --steps 2 --warmup 1 --no-cpu-baseline --no-tiers"
PG_PAIR_TILE=s PG_PLACE_TRIALS=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $O/pmc_sq -o ns --output-format csv -- $B > $O/pmc_sq.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in glob.glob("%s/pmc_sq/**/*counter_collection.csv" % O, recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "pair" in k:
            print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
