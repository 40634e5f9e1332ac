#!/bin/bash
# round 2, call y: where the fp4 pair kernels' cycles go (SQ counters), north-star shape
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02y; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_VALU_[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z_0-9]*\|SQ_WAIT_INST_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/counters.txt; cat $O/counters.txt; echo
B="python bench.py --workload northstar --steps 2 --warmup 1 --no-cpu-baseline --no-tiers"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o ns --output-format csv -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $O/pmc_sq2 -o ns --output-format csv -- $B > $O/pmc_sq2.log 2>&1
python - <<'PY'
import csv, collections, glob
for d in ("gpurun_out/r02y/pmc_sq", "gpurun_out/r02y/pmc_sq2"):
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "pair" in k or "pack" in k:
                print(k[-40:], {c: "%.4g" % (sum(x) / len(x)) for c, x in v.items()})
PY
tail -3 $O/pmc_sq2.log | cut -c1-300
