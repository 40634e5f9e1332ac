#!/bin/bash
# round 4: counters of k_popdist_np on the C2 data set in 2 kb windows (where does its time go?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r04pmc; mkdir -p $O/a $O/b
export PG_PLACE_TRIALS=1
B="python bench.py --workload c2_w2k --steps 3 --warmup 1 --no-cpu-baseline --no-tiers"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/a -o np --output-format csv -- $B > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/b -o np --output-format csv -- $B > $O/b.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob("gpurun_out/r04pmc/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        for k, v in acc.items():
            if "popdist_np" in k or "mirror" in k or "pack3" in k:
                print(d, k[:40], {c: round(x / 1e6, 2) for c, x in v.items()})
PY
