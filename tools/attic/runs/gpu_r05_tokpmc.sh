#!/bin/bash
# round 5: SQ counters of k_tok_parse2 on a 1.07 GB block of text (tools/tok_bench2.py 1320000 200)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05tokpmc; mkdir -p $O
B="python tools/tok_bench2.py 1320000 200"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o tok --output-format csv -- $B > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o tok --output-format csv -- $B > $O/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $O/sq2 -o tok --output-format csv -- $B > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o tok --output-format csv -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o tok --output-format csv -- $B > $O/write.log 2>&1
python - <<PY
import csv, collections, glob
for d in ("sq", "sq2", "fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "tok_parse" in k or "nl_" in k:
            print(d, k, {c: "%.3g" % (sum(x) / len(x)) for c, x in v.items()})
for r in csv.DictReader(open(glob.glob("$O/stats/*kernel_stats.csv")[0])):
    if "tok_parse" in r["Name"] or "nl_" in r["Name"]:
        print(r["Name"].split("(")[1][:30] if False else r["Name"][:50], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
