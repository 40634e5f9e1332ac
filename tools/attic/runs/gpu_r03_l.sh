#!/bin/bash
# round 3, call l: k_pairC_big as the default: kernel tests, counters of the north-star step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -5
export PG_PLACE_TRIALS=1
B="python bench.py --workload northstar --steps 3 --warmup 2 --no-cpu-baseline --no-tiers"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o ns --output-format csv -- $B > $O/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_mfma -o ns --output-format csv -- $B > $O/pmc_mfma.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_BUSY_CU_CYCLES -d $O/pmc_x -o ns --output-format csv -- $B > $O/pmc_x.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("pmc_sq", "pmc_mfma", "pmc_x"):
    for f in glob.glob("gpurun_out/r03l/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "pair" in k:
                print(d, k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
tail -3 $O/pmc_x.log | cut -c1-300
