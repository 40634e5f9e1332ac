#!/usr/bin/env python
"""Turn a gpurun_out/ rocprofv3 capture into the committed summaries under profiles/<tag>/ and refresh
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

    python profiles/summarize.py <tag> <workload> [gpurun_out]

Expects (each from its own rocprofv3 run of `python bench.py --workload <workload> ...`):
    <out>/prof_stats/<workload>_kernel_stats.csv          --kernel-trace --stats
    <out>/pmc_fetch/<workload>_counter_collection.csv     --pmc FETCH_SIZE
    <out>/pmc_write/<workload>_counter_collection.csv     --pmc WRITE_SIZE
    <out>/pmc_sq/<workload>_counter_collection.csv        --pmc SQ_* (optional)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE
reports half of the bytes of coalesced streaming reads (/opt/skills/guides/MI355X_MICROARCH.md, section HBM; confirmed here on
k_pack, whose 2.08 GB input reads show as 1.04 GB while its 1.6 GB of writes show exactly).
"""
import collections
import csv
import json
import os
import shutil
import sys


def main():
    tag, wl = sys.argv[1], sys.argv[2]
    src = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, tag)
    os.makedirs(dst, exist_ok=True)
    for f in ("kernel_stats", "domain_stats"):
        p = os.path.join(src, "prof_stats", "%s_%s.csv" % (wl, f))
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, "%s_%s.csv" % (wl, f)))
    out = {}
    for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mfma"):
        p = os.path.join(src, name, "%s_counter_collection.csv" % wl)
        if not os.path.exists(p):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k.startswith("__amd"):
                continue
            out.setdefault(k, {}).update({c: {"mean_per_launch": sum(x) / len(x), "launches": len(x)} for c, x in v.items()})
    with open(os.path.join(dst, "%s_pmc_summary.json" % wl), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    tpath = os.path.join(here, "hbm_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    traffic.setdefault(wl, {})
    traffic["_source"] = traffic.get("_source", {})
    for k, v in out.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            name = k.split("<")[0]
            traffic[wl][name] = int((2 * v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]) * 1024)
            traffic["_source"][wl] = tag
    # VALU wave-instructions per launch (SQ_INSTS_VALU) of the pair kernels: bench.py prices a VALU-bound dominant kernel with them
    for k, v in out.items():
        if "SQ_INSTS_VALU" in v:
            traffic.setdefault("_valu", {}).setdefault(wl, {})[k.split("<")[0]] = int(v["SQ_INSTS_VALU"]["mean_per_launch"])
    with open(tpath, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    log = os.path.join(src, "bench_prof_%s.log" % wl)
    if not os.path.exists(log):
        log = os.path.join(src, "bench_prof.log")
    if os.path.exists(log):
        with open(log) as f, open(os.path.join(dst, "%s_bench_line_under_rocprof.json" % wl), "w") as g:
            g.writelines(ln for ln in f if '"metric"' in ln)
    print(json.dumps(traffic.get(wl), indent=1))


if __name__ == "__main__":
    main()
