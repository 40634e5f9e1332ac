#!/usr/bin/env python
"""A rocprofv3 kernel trace of `python bench.py` holds more launches of the pack kernel than the timed steps: reserve()'s placement
probes (passes over each candidate allocation, on empty rows) come first, then k_synth fills the rows, then pg_tune_planes tries its
sets of planes on the filled rows (three passes a set), then the warm-up passes.  The timed region is the `steps` launches of the
dominant kernel that follow the `probes` + `warmup` ones behind the fill of the rows; this script writes the per-kernel statistics
of exactly those passes, next to rocprofv3's own --stats table over all launches.  `probes` = 3 x the number of sets in the bench
line's placement_trials.planes_probe_ms (0 under PG_PLANE_TRIALS=1): pass the bench line's file and it is read from there.

    python tools/prof_timed_region.py <kernel_trace.csv> <steps> [warmup = 2] [dominant kernel prefix = k_pack3] [bench line .json | probes = 0] > summary.csv"""
import csv
import statistics
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dom = sys.argv[4] if len(sys.argv) > 4 else "k_pack3"
    probes = 0
    if len(sys.argv) > 5:
        if sys.argv[5].isdigit():
            probes = int(sys.argv[5])
        else:
            import json
            line = json.loads(open(sys.argv[5]).read().strip().splitlines()[-1])
            probes = 3 * len((line.get("placement_trials") or {}).get("planes_probe_ms") or [])
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # bench.py's T0 section: reserve() (its probe passes run on empty rows) -> k_synth fills the rows -> `probes` passes of
    # pg_tune_planes -> `warmup` passes -> `steps` timed passes -> the other tiers.  So: the dominant kernel's launches behind the
    # first k_synth launch, the first `probes + warmup` of them dropped, the next `steps` kept.
    warmup += probes
    synth = [i for i, r in enumerate(rows) if r[2].startswith("k_synth")]
    start = synth[0] if synth else 0
    idx = [i for i in range(start, len(rows)) if rows[i][2].startswith(dom)]
    timed = idx[warmup:warmup + steps]
    assert len(timed) == steps, "the trace holds %d launches of %s behind the fill, %d + %d wanted" % (len(idx), dom, warmup, steps)
    first, last = timed[0], timed[-1]
    end_t = rows[idx[warmup + steps]][0] if len(idx) > warmup + steps else None
    per = {}
    for i in range(first, len(rows)):
        if end_t is not None and rows[i][0] >= end_t:
            break
        if rows[i][0] - rows[last][1] > 20_000_000:
            break
        per.setdefault(rows[i][2], []).append((rows[i][1] - rows[i][0]) / 1e6)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches_in_timed_region", "avg_ms", "min_ms", "median_ms", "max_ms", "launches_in_whole_trace", "avg_ms_whole_trace"])
    allk = {}
    for r in rows:
        allk.setdefault(r[2], []).append((r[1] - r[0]) / 1e6)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), "%.4f" % statistics.mean(v), "%.4f" % min(v), "%.4f" % statistics.median(v), "%.4f" % max(v),
                    len(allk[k]), "%.4f" % statistics.mean(allk[k])])


if __name__ == "__main__":
    main()
