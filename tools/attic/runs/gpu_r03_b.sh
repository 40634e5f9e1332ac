#!/bin/bash
# round 3, call b: hand-scheduled LDS-staged pair kernels -- parity, A/B against the one-wave kernels, counters
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-22s ms_per_step %.4f  kernels %s" % (sys.argv[2], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
for wl in northstar c2; do
  run ${wl}_tile $wl PG_X=1
  run ${wl}_onewave $wl PG_PAIR_TILE=none
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --workload northstar --steps 2 --warmup 1 --no-cpu-baseline --no-tiers"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o ns --output-format csv -- $B > $O/prof_stats.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/pmc_sq -o ns --output-format csv -- $B > $O/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o ns --output-format csv -- $B > $O/pmc_fetch.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for sub in ("stats", "pmc_sq", "pmc_fetch"):
    for f in glob.glob("%s/%s/**/*.csv" % (O, sub), recursive=True):
        if f.endswith("kernel_stats.csv"):
            for r in list(csv.DictReader(open(f)))[:8]:
                print("stats", r.get("Name", "")[:50], r.get("Calls"), r.get("AverageNs"))
        if f.endswith("counter_collection.csv"):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, d in acc.items():
                if "pair" in k or "pack" in k:
                    print(sub, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
