#!/usr/bin/env python
"""Benchmark of the hot path: popgenWindows pi / dxy / Fst over 50 kb coordinate windows (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload northstar|c2|c3|c4|c5|popfreq|c2_w5k|tiny]

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher).  A step is one pass of the whole per-window
statistics path over the rank's resident synthetic data set: pack -> pairwise D/C -> population sums on the GPU, the float64
finalisation (pi, dxy, Fst) on the GPU and D2H of the result table.  Measurement tier T0: the inputs are
generated on the device before the timed region (counter-based generator, genomics_general_amd/synth.py) and stay resident
in HBM.  The data path has no collective (windows are independent): at N>1 the ranks meet in an RCCL barrier on both sides
of the timed region, and every step ends in the one exchange a driver job has -- the all-gather of the finished per-window rows
(`result_allgather_ms_per_step`, inside the reported time).  `python bench.py --gpus N` without a launcher starts its own N
ranks.  Weak scaling (default): every rank owns a full-size data set (different scaffolds), `value` = windows of all ranks /
max-over-ranks time; the shape is the same at every N (north-star per rank), and from 8 ranks on the rank's share of BASELINE.json
configs[4] is measured behind the headline and reported as `c5_share`.  `--strong`: ONE data set cut into window ranges by the
drivers' multi-GPU plan (genomics_general_amd/shardplan.py), `scaling: "strong"`.

Default workload = the north-star single-GPU shape (BASELINE.json `north_star` "Target": 10^8 sites x 200 diploids, 4
populations, 50 kb windows = the first 10^8 sites of configs[4]); `--workload c2` is configs[1], c3 configs[2], c4 configs[3],
c5 the rank's share of configs[4] (3*10^9 sites / N, needs N >= 8).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the kernel family with the most GPU time), timed with HIP
events on the stream it runs on: an HBM-bound family (k_pack3, k_abba_q, k_popfreq_q) is priced in algorithmic bytes against
8 TB/s; the pair-count kernels run on the matrix cores (k_pairC_tile / k_pairD_fp4: algorithmic multiply-accumulates against the
dense MX fp4 peak; the popcount kernels of PG_PAIR_VALU in VALU wave-instructions against
the guide's issue ceiling and the measured ceiling of their instruction mix, profiles/: tools/valu_rate.hip).  `cpu_baseline` is the CPU oracle's restatement of
the reference's FULL path (.geno text -> parse -> windows -> alignment -> pair-by-pair loop -> statistics) run on all host
cores, one window per worker process (N=1, rank 0 only).  No torch anywhere: barriers and the gather go through RCCL in
libpopgen_hip.so.
"""
import argparse
import json
import os

# the benchmark's bgzipped samples are what htslib's bgzip writes (zlib, level 6) -- the reference's default input --, not what this
# library's own, faster compressor would write (csrc/pg_fast_deflate.h: shorter matches, i.e. more symbols for k_inflate to decode)
os.environ.setdefault("PG_BGZF_ZLIB", "1")
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # north-star single-GPU shape: first 10^8 sites of config 5 (200 diploids)
    "northstar": dict(n_sites=100_000_000, n_scaf=4, n_dip=200, n_pops=4, wind=50_000, min_sites=100, tool="popgen",
                      desc="popgenWindows pi/Fst/Dxy: 1e8 sites x 200 diploids (400 haplotypes), 4 pops, 50 kb windows"),
    # BASELINE.json configs[1]: 10^7 sites x 100 diploids, 4 pops, 50 kb windows
    "c2": dict(n_sites=10_000_000, n_scaf=4, n_dip=100, n_pops=4, wind=50_000, min_sites=100, tool="popgen",
               desc="popgenWindows pi/Fst/Dxy: 1e7 sites x 100 diploids (200 haplotypes), 4 pops, 50 kb windows"),
    # the C2 data set in 5 kb windows (2000 windows of 157 words): per-window fixed costs of the pair kernels; not a BASELINE.json config
    "c2_w5k": dict(n_sites=10_000_000, n_scaf=4, n_dip=100, n_pops=4, wind=5_000, min_sites=100, tool="popgen",
                   desc="popgenWindows pi/Fst/Dxy: 1e7 sites x 100 diploids (200 haplotypes), 4 pops, 5 kb windows"),
    # the C2 data set in 2 kb windows (5000 windows): windows of up to 4096 sites get their float64 sums in NumPy's order (k_popdist_np),
    # what that costs at tier T0 (PG_POPDIST_TREE=0 for the fixed-tree finisher on the same windows); not a BASELINE.json config
    "c2_w2k": dict(n_sites=10_000_000, n_scaf=4, n_dip=100, n_pops=4, wind=2_000, min_sites=100, tool="popgen",
                   desc="popgenWindows pi/Fst/Dxy: 1e7 sites x 100 diploids (200 haplotypes), 4 pops, 2 kb windows"),
    # BASELINE.json configs[4]: 3*10^9 sites x 200 diploids sharded over the ranks (n_sites is per rank, set in main)
    "c5": dict(n_sites=None, n_scaf=3, n_dip=200, n_pops=4, wind=50_000, min_sites=100, tool="popgen",
               desc="popgenWindows pi/Fst/Dxy: this rank's share of 3e9 sites x 200 diploids (400 haplotypes), 4 pops, 50 kb windows"),
    # BASELINE.json configs[2]: ABBA-BABA
    "c3": dict(n_sites=10_000_000, n_scaf=4, n_dip=100, n_pops=4, wind=50_000, min_sites=100, tool="abba",
               desc="ABBABABAwindows D/fd: 1e7 sites, P1/P2/P3/O x 25 diploids, 50 kb windows"),
    # BASELINE.json configs[3]: distMat pairwise-kernel stress
    "c4": dict(n_sites=1_000_000, n_scaf=1, n_dip=1000, n_pops=1, wind=100_000, min_sites=1, tool="distmat",
               desc="distMat full pairwise distance: 1e6 sites x 1000 diploids (2000 haplotypes), 100 kb windows"),
    # --analysis popFreq on the C2 data set (k_popfreq; not a BASELINE.json config)
    "popfreq": dict(n_sites=10_000_000, n_scaf=4, n_dip=100, n_pops=4, wind=50_000, min_sites=100, tool="popfreq",
                    desc="popgenWindows --analysis popFreq: 1e7 sites x 100 diploids (200 haplotypes), 4 pops, 50 kb windows"),
    # small variant for quick checks
    "tiny": dict(n_sites=400_000, n_scaf=2, n_dip=20, n_pops=4, wind=50_000, min_sites=100, tool="popgen",
                 desc="tiny smoke workload"),
}
CPU_DISTMAT_HAPS = 120          # distMat CPU sample: 7140 pairs x 100 kb ~ 5 s per window
T1_BLOCK_SITES = 1 << 20        # T1 sample: host blocks of about a million sites, eight of them
T1_BLOCKS = 8
T2_SITES = 25_000_000           # T2 sample: this many sites of the workload as `.geno` text (500 windows of 50 kb; 20.3 GB at 400 haplotypes),
                                # less when the temporary directory or the host memory is short of room for it (PG_BENCH_T2_SITES overrides)
CPU_WHOLE_WINDOWS = 8           # CPU sample: this many whole windows, one per worker on an otherwise idle host (~60 s at 400 haplotypes)
CPU_SLICE_SITES = 4_000         # CPU sample: the first 4000 sites of a window per worker (~4 s of CPU work at 400 haplotypes on an idle core,
                                # ~10x that with every hardware thread of a 256-thread host busy)
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GUIDE = 1024 * 2.4e9 / 2      # wave64 VALU instructions per second: 256 CUs x 4 SIMD-32, 2 cycles each, 2.4 GHz (guide)
MFMA_FP4_PEAK_TFLOPS = 10000.0  # MX fp4 MFMA, dense (guide: ~10 PF; measured ceiling 9099 with 32x32x64)
VALU_PAIRSITES_PEAK = 3.6e14    # 7 lane-ops per 32 pair-sites at 7.9e13 lane-ops/s


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline worker (spawned process: no GPU, no engine; only the oracle).  One window of the workload: rendered as
# `.geno` text (untimed), then the oracle's restatement of the reference's whole path on it (timed).
# ---------------------------------------------------------------------------------------------------------------
def _cpu_window_job(job):
    import tempfile
    sys.path.insert(0, ROOT)
    from genomics_general_amd import synth
    from oracle import popgen_oracle as orc
    codes, names, pops, wind, min_sites, tool, scaf, pos0 = job
    L = codes.shape[0]
    fd, path = tempfile.mkstemp(suffix=".geno", prefix="pg_cpu_baseline_")
    os.close(fd)
    try:
        synth.write_geno(path, [scaf], np.zeros(L, dtype=np.int64), np.arange(pos0, pos0 + L), codes, names, sep="/", fmt="phased")
        t0 = time.time()
        if tool == "popgen":          # popgenWindows.py:28-75 on the file: parse, windows, genoToAlignment, pair loop, statistics
            csv = orc.popgen_windows_csv(path, "phased", pops, wind, min_sites=min_sites, min_data=0.01, round_to=12,
                                         counts_fn=orc.pair_counts_loop, write_failed=True)
        elif tool == "popfreq":
            csv = orc.popgen_windows_csv(path, "phased", pops, wind, min_sites=min_sites, analysis=("popFreq",), round_to=12,
                                         write_failed=True)
        else:                         # ABBABABAwindows.py:27-52
            csv = orc.abbababa_windows_csv(path, "phased", pops, wind, min_sites=min_sites, min_data=0.01, write_failed=True)
        t1 = time.time()
    finally:
        os.unlink(path)
    return t0, t1, csv


def _cpu_leg(eng, lay, wl, names, scaf_len, s_lo, s_hi, workers):
    """The oracle's restatement of the reference's whole path on the site ranges [s_lo[k], s_hi[k]) (each rendered as its own
    one-window `.geno` file), `workers` processes at once.  Returns (wall seconds, summed busy seconds, CSVs, GPU statistics of the
    same ranges)."""
    import multiprocessing as mp
    per = len(names) // lay.n_pops
    pops = [(p, names[k * per:(k + 1) * per]) for k, p in enumerate(lay.sampleData.popNames)]
    # generator order of the columns: sample d owns columns (2d, 2d+1)
    col_of_slot = np.array([2 * names.index(lay.hap_sample_name[s]) + (s - lay.ind_slots[lay.hap_sample_name[s]][0])
                            for s in range(lay.n_hap)])
    wb = eng.batch(s_lo, s_hi)
    if wl["tool"] == "popgen":
        stats = wb.groupDistStats(doPairs=True, minSites=wl["min_sites"], minData=0.01)
    elif wl["tool"] == "popfreq":
        stats = wb.groupFreqStats()
    else:
        stats = wb.ABBABABA("pop0", "pop1", "pop2", "pop3", 0.01)
    jobs = []
    for a, b in zip(s_lo, s_hi):
        slot_codes = eng.download(int(a), int(b - a))
        codes = np.zeros_like(slot_codes)
        codes[:, col_of_slot] = slot_codes
        # the range is rendered with positions 1.. on one scaffold: exactly one window of its length for the tool
        jobs.append((codes, names, pops, int(b - a), wl["min_sites"], wl["tool"], "chr%d" % (int(a) // scaf_len + 1), 1))
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_window_job, jobs, chunksize=1)
    wall = max(r[1] for r in res) - min(r[0] for r in res)
    busy = sum(r[1] - r[0] for r in res)
    return wall, busy, [r[2] for r in res], stats


def _cpu_matches(wl, csvs, stats):
    ok = True
    for k, csv in enumerate(csvs):
        rows = csv.strip().split("\n")
        head, vals = rows[0].split(","), rows[1].split(",")
        for name, v in zip(head, vals):
            if name in ("scaffold", "start", "end", "mid", "sites"):
                continue
            v = float(v)
            if name == "sitesUsed":
                ok = ok and (int(stats[name][k]) == int(v) if v == v else True)
                continue
            if name not in stats:
                continue
            g = float(stats[name][k])
            # the ABBA-BABA driver rounds to 4 decimals (ABBABABAwindows.py:42), the popgenWindows leg is run with --roundTo 12
            tol = 0.5e-4 + 1e-6 if wl["tool"] == "abba" else 1e-6 * max(1.0, abs(v))
            ok = ok and (abs(g - v) <= tol or (g != g and v != v))
    return bool(ok)


def effective_cpus():
    """How many CPUs this process may really use, and where the number comes from: the logical CPUs of the host, cut by the
    scheduler affinity mask and by the cgroup's CPU quota (a container on a 256-thread host may be limited to a handful of CPUs'
    worth of time: more busy processes than that share the quota and gain nothing)."""
    logical = os.cpu_count() or 1
    info = {"logical": logical}
    n = float(logical)
    try:
        aff = len(os.sched_getaffinity(0))
        info["affinity"] = aff
        n = min(n, aff)
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt and txt[0] != "max":
                    info["cgroup_quota_cpus"] = round(int(txt[0]) / int(txt[1]), 2)
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        info["cgroup_quota_cpus"] = round(q / int(f.read()), 2)
        except Exception:
            continue
    if "cgroup_quota_cpus" in info:
        n = min(n, info["cgroup_quota_cpus"])
    try:
        with open("/proc/pressure/cpu") as f:
            info["pressure_cpu"] = f.readline().strip()
    except Exception:
        pass
    info["usable"] = max(1, int(n + 0.5))
    return info


REF_WRAP = ("import sys, runpy, numpy as np; np.NaN = np.nan; sys.path.insert(0, %r); "
            "sys.argv = sys.argv[1:]; runpy.run_path(sys.argv[0], run_name='__main__')")


def cpu_baseline_reference(wl, names, n_pops, ref_dir, cores, budget_s=420.0):
    """The UNMODIFIED reference (/root/reference/popgenWindows.py:386-460: reader -> window queue -> -T worker
    processes -> sorter -> writer) on a W-window prefix of the workload rendered as `.geno` text by the host statement of the
    generator (no GPU involved), `-T cores` and `-T 1`, wall clock around the whole process, under `timeout`.  Only where the
    reference is present (the build container; it does not travel to the GPU box)."""
    import subprocess
    import tempfile
    from genomics_general_amd import synth
    tool = {"popgen": "popgenWindows.py", "abba": "ABBABABAwindows.py"}[wl["tool"]]
    wind, n_dip = wl["wind"], wl["n_dip"]
    per = n_dip // n_pops
    W = int(max(2, min(cores, 16)))                            # one window per worker process
    tmp = tempfile.mkdtemp(prefix="pg_ref_baseline_")
    geno = os.path.join(tmp, "prefix.geno")
    sid, pos = np.zeros(W * wind + 1, dtype=np.int64), np.arange(1, W * wind + 2)      # one site beyond: the last window is full
    with open(geno, "w") as f:
        step = 50_000
        for a in range(0, len(pos), step):
            codes = synth.gen_codes(synth.SEED_DEFAULT, sid[a:a + step], pos[a:a + step], n_dip, n_pops)
            part = geno + ".part"
            synth.write_geno(part, ["chr1"], sid[a:a + step], pos[a:a + step], codes, names, sep="/", fmt="phased")
            with open(part) as g:
                if a:
                    g.readline()
                f.write(g.read())
            os.remove(part)
    argv = ["-g", geno, "-f", "phased", "-w", str(wind), "-m", str(wl["min_sites"]), "--roundTo", "12"]
    if tool == "popgenWindows.py":
        for k in range(n_pops):
            argv += ["-p", "pop%d" % k, ",".join(names[k * per:(k + 1) * per])]
    else:
        for flag, k in (("-P1", 0), ("-P2", 1), ("-P3", 2), ("-O", 3)):
            argv += [flag, "pop%d" % k, ",".join(names[k * per:(k + 1) * per])]
    runs = {}
    try:
        for label, T, n_win in (("T_all", cores, W), ("T_1", 1, 2)):
            out = os.path.join(tmp, label + ".csv")
            a = list(argv)
            if n_win < W:                                       # -T 1 on the first two windows only
                short = os.path.join(tmp, "two.geno")
                with open(geno) as f, open(short, "w") as g:
                    for i, ln in enumerate(f):
                        if i > n_win * wind + 1:
                            break
                        g.write(ln)
                a[1] = short
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-c", REF_WRAP % ref_dir, os.path.join(ref_dir, tool)] + a + ["-o", out, "-T", str(T)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget_s, cwd=tmp)
            dt = time.perf_counter() - t0
            with open(out) as f:
                rows = max(sum(1 for _ in f) - 1, 0)
            runs[label] = {"threads": T, "windows": rows, "wall_seconds": round(dt, 2), "windows_per_sec": round(rows / dt, 5),
                           "sites_per_sec": round(rows * wind / dt, 1), "returncode": r.returncode}
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    best = max(runs.values(), key=lambda x: x["windows_per_sec"])
    return {"value": best["windows_per_sec"], "unit": "windows/s", "sites_per_sec": best["sites_per_sec"], "cores": best["threads"],
            "kind": "reference", "runs": runs,
            "sample": "the unmodified %s/%s on the first %d windows of the workload (%d sites x %d haplotypes each) as `.geno` text, -T %d and "
                      "-T 1 (first 2 windows), wall clock around the process (interpreter start, the serial reader and the %s of the "
                      "reference included), under a %d s timeout" % (ref_dir, tool, W, wind, 2 * n_dip, cores,
                                                                     "sorter / writer threads" if tool == "popgenWindows.py" else "5 s polling sleeps",
                                                                     int(budget_s))}


def cpu_baseline_full_path(eng, lay, wl, names, lo, hi, scaf_len, max_workers):
    """Three legs of the oracle's restatement of the reference's whole path (text parse -> window -> genoToAlignment -> pair-by-pair
    loop -> statistics), each compared with the GPU statistics of the same sites:
      A  one slice of CPU_SLICE_SITES sites (the head of a window) per worker, one worker per hardware thread of the host, all at
         once: the host's throughput in sites/s;
      B  a few WHOLE windows, one per worker, the rest of the host idle;   C  the heads of the same windows as slices, same workers:
         B against C is the measured cost of a whole window relative to a slice (every stage of the reference is linear in the
         sites of a window; this shows it instead of asserting it).
    windows/s of the host = A's sites/s / sites per window / (B's seconds per site / C's seconds per site)."""
    cpus = effective_cpus()
    cores = cpus["logical"]
    phys = cores
    workers = cores
    slice_sites = int(min(CPU_SLICE_SITES, wl["wind"]))
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or cores
        # a worker holds its slice as Python strings + int64 alignment: ~160 B per genotype cell, plus the interpreter
        per_worker = 160 * slice_sites * lay.n_hap // 2 + (1 << 28)
        workers = max(1, min(workers, int(psutil.virtual_memory().available * 0.6 // per_worker)))
    except Exception:
        pass
    workers = max(1, min(workers, max_workers, len(lo)))
    # Leg A is a sweep over the number of worker processes: the host's best is what counts, and one process per LOGICAL CPU is not
    # it when the process may only use a few CPUs' worth of time (cgroup quota, affinity) or when SMT siblings share a core.
    # Every point runs one slice per worker, all at once; value = the best point's sites/s.
    points = sorted(set(w for w in (cpus["usable"], 8, 16, 32, 64, 128, 256, phys, workers) if 1 <= w <= workers))
    sweep, ok, stale = [], True, 0
    for wk in points:
        sel = np.linspace(0, len(lo) - 1, wk).astype(int) if wk > 1 else np.array([0])
        s_lo = lo[sel].copy()
        s_hi = np.minimum(s_lo + slice_sites, hi[sel])
        wall, busy, csvs, stats = _cpu_leg(eng, lay, wl, names, scaf_len, s_lo, s_hi, wk)
        ok = ok and _cpu_matches(wl, csvs, stats)
        sweep.append({"workers": wk, "wall_seconds": round(wall, 2), "busy_seconds": round(busy, 2),
                      "sites_per_sec": round(len(sel) * slice_sites / wall, 1),
                      "busy_seconds_per_site_per_worker": busy / float(len(sel) * slice_sites)})
        rate = sweep[-1]["sites_per_sec"]
        prev_best = max([x["sites_per_sec"] for x in sweep[:-1]] or [0.0])
        stale = stale + 1 if rate < 1.1 * prev_best else 0
        if stale >= 2 or wall > 60:                              # two points in a row without a 10 % gain: more workers only share the same CPUs
            break
    best = max(sweep, key=lambda x: x["sites_per_sec"])
    workers, wall, busy, n_slices, sites_s = best["workers"], best["wall_seconds"], best["busy_seconds"], best["workers"], best["sites_per_sec"]
    # legs B and C: whole windows against their heads, on an otherwise idle host
    n_whole = int(max(1, min(CPU_WHOLE_WINDOWS, phys, max_workers, len(lo))))
    wsel = np.linspace(0, len(lo) - 1, n_whole).astype(int) if n_whole > 1 else np.array([0])
    w_lo, w_hi = lo[wsel].copy(), hi[wsel].copy()
    wall_b, busy_b, csv_b, stats_b = _cpu_leg(eng, lay, wl, names, scaf_len, w_lo, w_hi, n_whole)
    wall_c, busy_c, csv_c, stats_c = _cpu_leg(eng, lay, wl, names, scaf_len, w_lo, np.minimum(w_lo + slice_sites, w_hi), n_whole)
    ok = ok and _cpu_matches(wl, csv_b, stats_b) and _cpu_matches(wl, csv_c, stats_c)
    per_site_whole = busy_b / float((w_hi - w_lo).sum())
    per_site_slice = busy_c / float(n_whole * slice_sites)
    ratio = per_site_whole / per_site_slice
    return {"value": round(sites_s / wl["wind"] / ratio, 5), "unit": "windows/s", "sites_per_sec": round(sites_s / ratio, 1),
            "cores": workers, "host_cores": cores, "host_cores_physical": phys, "host_cpus": cpus, "kind": "port",
            "worker_sweep": sweep,
            "worker_sweep_note": "busy_seconds = summed wall time of the workers' timed regions: when it grows with the worker count "
                                 "at constant work per worker, the workers are sharing CPUs (quota / SMT / memory), not computing more",
            "sample": "A (best point of a sweep over the worker count): %d slices of %d sites x %d haplotypes (the heads of %d evenly spaced windows of the workload), each rendered as "
                      ".geno text and run through the oracle's restatement of the reference's whole path (text parse -> window -> "
                      "genoToAlignment -> pair-by-pair loop -> statistics; popgenWindows.py:28-75, genomics.py:1884-1945, 1101-1127, "
                      "903-916, 956-995), one slice per worker process, %d worker processes at once.  "
                      "B: %d WHOLE windows of %d sites, one per worker, the rest of the host idle; C: the heads of the same windows as "
                      "slices, same workers.  value = A's sites/s / %d sites per window / (B's seconds per site / C's seconds per "
                      "site)" % (n_slices, slice_sites, lay.n_hap, n_slices, workers, n_whole, wl["wind"], wl["wind"]),
            "all_threads_slices": {"workers": workers, "wall_seconds": round(wall, 2), "cpu_seconds": round(busy, 2),
                                   "sites_per_sec": round(sites_s, 1)},
            "whole_windows": {"workers": n_whole, "windows": n_whole, "wall_seconds": round(wall_b, 2), "cpu_seconds": round(busy_b, 2),
                              "windows_per_sec": round(n_whole / wall_b, 5), "seconds_per_site_per_worker": per_site_whole},
            "slices_same_workers": {"workers": n_whole, "wall_seconds": round(wall_c, 2), "cpu_seconds": round(busy_c, 2),
                                    "seconds_per_site_per_worker": per_site_slice},
            "whole_window_cost_relative_to_slices": round(ratio, 4),
            "wall_seconds": round(wall + wall_b + wall_c, 2), "cpu_seconds": round(busy + busy_b + busy_c, 2),
            "gpu_matches_oracle_on_sample": bool(ok)}


def cpu_baseline_distmat(eng, lay, wl, lo, hi):
    """distMat (2000 haplotypes): the pair loop is quadratic in haplotypes, a window costs ~20 minutes on one core; time the first
    CPU_DISTMAT_HAPS haplotypes of one window and scale by the pair count (numeric core only)."""
    from oracle import popgen_oracle as orc
    n_hap = lay.n_hap
    codes = eng.download(int(lo[0]), int(hi[0] - lo[0]))
    scale = (n_hap * (n_hap - 1) / 2) / (CPU_DISTMAT_HAPS * (CPU_DISTMAT_HAPS - 1) / 2)
    aln, _ = orc.aln_from_codes(codes[:, :CPU_DISTMAT_HAPS], lay.hap_names[:CPU_DISTMAT_HAPS],
                                lay.hap_sample_name[:CPU_DISTMAT_HAPS], lay.hap_group[:CPU_DISTMAT_HAPS])
    c0 = time.perf_counter()
    orc.pair_counts_loop(aln)
    dt = (time.perf_counter() - c0) * scale
    return {"value": round(1.0 / dt, 6), "unit": "windows/s", "cores": 1, "kind": "port",
            "sample": "numeric core only (no text parsing / alignment build): the pair-by-pair loop (genomics.py:903-916) of one window "
                      "(%d sites) timed on its first %d haplotypes and scaled by the pair count to %d haplotypes" % (
                          wl["wind"], CPU_DISTMAT_HAPS, n_hap),
            "seconds": round(dt, 2), "gpu_matches_oracle_on_sample": None}


def tier_samples(eng, lay, wl, names, slot_gen, scaf_len, n_sites, t1_block, t0_table):
    """Tiers T1 and T2 of SURVEY.md 8d on bounded samples of the workload (the T0 value above is the headline):
    T1  host int8 blocks (page-locked, at the engine's row pitch) -> asynchronous H2D into alternating halves of a device buffer
        -> kernels -> result table D2H, block k+1 uploading while block k computes (Engine.upload_async / upload_wait);
    T2  `.geno` text on disk -> the drop-in popgenWindows.py (reader thread, tokenizer thread, uploads, kernels) -> CSV,
        timed inside the driver process (PG_TIMING total_s: from opening the input to the last row), and its rows compared
        with the T0 statistics of the same windows."""
    import subprocess
    import tempfile
    out = {}
    wind = wl["wind"]
    pitch = eng.row_pitch
    n_blocks = int(min(T1_BLOCKS, n_sites // t1_block))
    host = eng.pinned.empty((n_blocks * t1_block, pitch), np.int8)
    host[:, lay.n_hap:] = 0
    for k in range(n_blocks):                                  # untimed: the sample's rows, from the resident data set
        host[k * t1_block:(k + 1) * t1_block, :lay.n_hap] = eng.download(k * t1_block, t1_block)
    base = [n_sites, n_sites + t1_block]
    lo = np.arange(0, t1_block, wind, dtype=np.int64)
    for rep in range(2):                                       # second pass is the measurement (first: allocations)
        t0 = time.perf_counter()
        eng.upload_async(host[0:t1_block], base[0])
        for k in range(n_blocks):
            eng.upload_wait()
            if k + 1 < n_blocks:
                eng.upload_async(host[(k + 1) * t1_block:(k + 2) * t1_block], base[(k + 1) % 2])
            tab, _ = eng.batch(lo + base[k % 2], lo + base[k % 2] + wind).groupDistTable(True, wl["min_sites"], 0.01)
        dt = time.perf_counter() - t0
    ok = bool(np.array_equal(tab, t0_table[(n_blocks - 1) * len(lo):n_blocks * len(lo)], equal_nan=True))
    # the same blocks as packed cells (the `.pgeno` payload: one byte per diploid genotype, expanded on the device by k_unpack)
    t1p = None
    if lay.n_hap == 2 * lay.n_samp:
        cells = eng.pinned.empty((n_blocks * t1_block, lay.n_samp), np.uint8)
        slot0 = np.array([lay.ind_slots[nm][0] for nm in lay.ind_order])
        np.bitwise_or(host[:, slot0].view(np.uint8), host[:, slot0 + 1].view(np.uint8) << 4, out=cells)
        slot_src = np.empty(lay.n_hap, dtype=np.int32)
        slot_src[slot0], slot_src[slot0 + 1] = 2 * np.arange(lay.n_samp), 2 * np.arange(lay.n_samp) + 1
        for rep in range(2):
            t0 = time.perf_counter()
            eng.upload_packed_async(cells[0:t1_block], base[0], slot_src)
            for k in range(n_blocks):
                eng.upload_wait()
                if k + 1 < n_blocks:
                    eng.upload_packed_async(cells[(k + 1) * t1_block:(k + 2) * t1_block], base[(k + 1) % 2], slot_src)
                tabp, _ = eng.batch(lo + base[k % 2], lo + base[k % 2] + wind).groupDistTable(True, wl["min_sites"], 0.01)
            dtp = time.perf_counter() - t0
        t1p = {"sites_per_sec": round(n_blocks * t1_block / dtp, 1), "windows_per_sec": round(n_blocks * len(lo) / dtp, 2),
               "h2d_GBps": round(n_blocks * t1_block * lay.n_samp / dtp / 1e9, 2),
               "matches_t0": bool(np.array_equal(tabp, t0_table[(n_blocks - 1) * len(lo):n_blocks * len(lo)], equal_nan=True)),
               "sample": "the same blocks as packed cells (1 byte per diploid genotype = 0.5 byte per call over PCIe), expanded into "
                         "resident rows on the device (pg_upload_packed_async / k_unpack)"}
        del cells
    out["t1_packed"] = t1p
    out["t1"] = {"sites_per_sec": round(n_blocks * t1_block / dt, 1), "windows_per_sec": round(n_blocks * len(lo) / dt, 2),
                 "h2d_GBps": round(n_blocks * t1_block * pitch / dt / 1e9, 2), "matches_t0": ok,
                 "sample": "%d page-locked host blocks of %d sites x %d haplotypes (int8, 1 byte per call), uploaded into alternating "
                           "halves of a device buffer while the previous block's windows are computed; PCIe-bound" % (
                               n_blocks, t1_block, lay.n_hap)}
    del host
    # ---- T2 ----
    out["t2"] = t2_sample(eng, lay, wl, names, scaf_len, t0_table)
    out["vcf"] = vcf_sample()
    return out


def vcf_sample(n_sites=400000, n_samples=200):
    """The upstream producer of the engine's input: VCF_processing/parseVCF.py's drop-in on a synthetic GATK-style
    VCF (tools/vcf_bench.py: GT:AD:DP:GQ, indels, tri-allelic sites, missing calls; bgzipped by zlib at level 6 as htslib does) --
    lines parsed on the device (k_vcf_heads / k_vcf_cells), `.geno.gz` rows deflated on the device (k_deflate) --, beside the same
    command with the host parser and the host's deflate (PG_VCF_DEVICE=0).  Seconds are process wall-clock, device context included."""
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "vcf_bench.py")
    out = {}
    try:
        for key, env in (("device_parser", {"VCF_LEGS": "0,2", "VCF_REPS": "2"}),
                         ("host_parser", {"VCF_LEGS": "0", "VCF_REPS": "1", "PG_VCF_DEVICE": "0"})):
            r = subprocess.run([sys.executable, tool, str(n_sites), str(n_samples), "--ref-sites", "0"], env=dict(os.environ, **env),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if r.returncode != 0:
                out[key] = {"error": r.stderr.decode()[-300:]}
                continue
            res = json.loads(r.stdout.decode().strip().splitlines()[-1])
            out.setdefault("sample", {k: res[k] for k in ("sites", "samples", "vcf_bytes", "vcf_gz_bytes", "options")})
            out[key] = {name: {"seconds": leg["seconds"], "sites_per_sec": leg["sites_per_sec"], "vcf_text_MBps": leg["vcf_text_MBps"],
                               "blocks_parsed_on_device": leg["timing"].get("blocks_parsed_on_device"),
                               "blocks": leg["timing"].get("blocks")} for name, leg in res["legs"].items()}
        a = out.get("device_parser", {}).get("vcf.gz -> geno.gz"), out.get("host_parser", {}).get("vcf.gz -> geno.gz")
        if a[0] and a[1]:
            out["device_over_host_parser"] = round(a[1]["seconds"] / a[0]["seconds"], 2)
    except Exception as exc:                                    # side information: never lose the main line
        out["error"] = repr(exc)[:300]
    return out


def write_geno_resident(path, eng, lay, names, n_rows, scaf="chr1", workers=0):
    """The first n_rows resident rows of the engine as `.geno` text (phased cells `A/C`, one scaffold, positions 1..): rows are
    fetched from the device in pieces and rendered by a pool of threads, every piece written at its own offset (lines of positions
    with the same number of digits have the same length, so the offsets follow from the row numbers)."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from genomics_general_amd import synth
    n = len(names)
    s0 = np.array([lay.ind_slots[nm][0] for nm in names])          # file column d <- slots s0[d], s0[d] + 1
    head = (scaf + "\t").encode()
    hdr = ("#CHROM\tPOS\t" + "\t".join(names) + "\n").encode()
    tasks, off, a, step = [], len(hdr), 0, 250_000
    while a < n_rows:
        nd = len(str(a + 1))
        b = min(n_rows, a + step, 10 ** nd - 1)
        tasks.append((a, b, nd, off))
        off += (b - a) * (len(head) + nd + 1 + 4 * n)
        a = b
    lock = threading.Lock()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.pwrite(fd, hdr, 0)

    def render(task):
        a, b, nd, off = task
        with lock:                                                  # one copy out of the device at a time (a context is not thread-safe)
            rows = eng.download(a, b - a)
        letters = synth.codes_to_letters(rows)
        line = np.empty((b - a, len(head) + nd + 1 + 4 * n), dtype=np.uint8)
        line[:, :len(head)] = np.frombuffer(head, dtype=np.uint8)
        pos = np.arange(a + 1, b + 1, dtype=np.int64)
        for k in range(nd):
            line[:, len(head) + nd - 1 - k] = (pos // 10 ** k % 10 + ord("0")).astype(np.uint8)
        line[:, len(head) + nd] = ord("\t")
        cell = line[:, len(head) + nd + 1:].reshape(b - a, n, 4)
        cell[:, :, 0] = letters[:, s0]
        cell[:, :, 1] = ord("/")
        cell[:, :, 2] = letters[:, s0 + 1]
        cell[:, :, 3] = ord("\t")
        cell[:, -1, 3] = ord("\n")
        buf, at = memoryview(line).cast("B"), 0
        while at < len(buf):
            at += os.pwrite(fd, buf[at:at + (1 << 30)], off + at)
    try:
        with ThreadPoolExecutor(workers or min(32, os.cpu_count() or 1)) as ex:
            list(ex.map(render, tasks))
    finally:
        os.close(fd)
    return off


def write_pgeno_resident(path, eng, lay, names, n_rows, scaf="chr1", workers=0):
    """The first n_rows resident rows as a `.pgeno` file with raw cells (codec none: one byte per diploid genotype, first allele's
    one-hot code | second << 4; genomics_general_amd/genoio.py, tools/geno_pack.py): blocks of a million rows rendered by a pool of
    threads, every block written at its own offset."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from genomics_general_amd import genoio
    n = len(names)
    s0 = np.array([lay.ind_slots[nm][0] for nm in names])
    head = json.dumps({"names": list(names), "ploidy": [2] * n, "codec": "none"}).encode()
    pre = genoio.PGENO_MAGIC + len(head).to_bytes(4, "little") + head
    sb = scaf.encode()
    tasks, off, step = [], len(pre), 1_000_000
    for a in range(0, n_rows, step):
        b = min(n_rows, a + step)
        tasks.append((a, b, off))
        off += 8 + 4 + 8 + 2 + len(sb) + (b - a) * (4 + n)
    lock = threading.Lock()
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.pwrite(fd, pre, 0)
    os.pwrite(fd, (0).to_bytes(8, "little"), off)                   # the terminating block

    def render(task):
        a, b, at = task
        with lock:
            rows = eng.download(a, b - a).view(np.uint8)
        blk = ((b - a).to_bytes(8, "little") + (1).to_bytes(4, "little") + (0).to_bytes(8, "little") + len(sb).to_bytes(2, "little") + sb +
               np.arange(a + 1, b + 1, dtype="<i4").tobytes())
        os.pwrite(fd, blk, at)
        cells = rows[:, s0] | (rows[:, s0 + 1] << 4)
        buf, done = memoryview(np.ascontiguousarray(cells)).cast("B"), 0
        while done < len(buf):
            done += os.pwrite(fd, buf[done:done + (1 << 30)], at + len(blk) + done)
    try:
        with ThreadPoolExecutor(workers or min(32, os.cpu_count() or 1)) as ex:
            list(ex.map(render, tasks))
    finally:
        os.close(fd)
    return off + 8


def t2_sample(eng, lay, wl, names, scaf_len, t0_table):
    """Tier T2 (the only tier whose work is the reference's: `.geno` text in, CSV out): the head of the workload's
    first scaffold as text on disk (written before the clock starts, so it sits in the page cache like a file a pipeline has just
    produced) through the drop-in popgenWindows.py in a process of its own, timed inside the driver (PG_TIMING), its rows compared
    with the T0 statistics of the same windows."""
    import shutil
    import subprocess
    import tempfile
    wind = wl["wind"]
    want = int(os.environ.get("PG_BENCH_T2_SITES", T2_SITES))
    tmp = tempfile.mkdtemp(prefix="pg_bench_t2_")
    geno, csv = os.path.join(tmp, "sample.geno"), os.path.join(tmp, "out.csv")
    line_bytes = 5 * len(names) + 20                                  # the text + the same sites once more as packed cells
    room = shutil.disk_usage(tmp).free * 0.4
    try:
        import psutil
        room = min(room, psutil.virtual_memory().available * 0.3)       # the file should stay in the page cache
    except Exception:
        pass
    n_txt = int(min(want, scaf_len, room // line_bytes) // wind * wind)
    per = len(names) // lay.n_pops
    cmd = [sys.executable, os.path.join(ROOT, "popgenWindows.py"), "-g", geno, "-o", csv, "-f", "phased", "-w", str(wind),
           "-m", str(wl["min_sites"])]
    for k, p in enumerate(lay.sampleData.popNames):
        cmd += ["-p", p, ",".join(names[k * per:(k + 1) * per])]
    # The timed runs are the reference's own command line, i.e. its default --roundTo 4 (popgenWindows.py:198); every cell is held against
    # the T0 statistics to half a unit of the fourth decimal.  One more run of the text leg (and of the whole workload) prints twelve
    # decimals and is held against T0 to 1e-9: at twelve digits every value is within reach of a rounding tie of its last digit, so the
    # driver computes EVERY window a second time in NumPy's summation order (cli._refine_long_windows) -- twice the statistics work,
    # reported beside the rate, not as the rate (rounds 3 - 5 timed that command).
    DEEP = ["--roundTo", "12"]

    def agrees(v, g, digits):
        v = float(v)
        if g != g or v != v:
            return g != g and v != v
        return abs(v - g) <= (0.5 * 10.0 ** -digits if digits < 10 else 0.0) + 1e-9 * max(1.0, abs(g))
    try:
        if n_txt < wind:
            raise RuntimeError("no room for a T2 sample in %s" % tmp)
        w0 = time.perf_counter()
        size = write_geno_resident(geno, eng, lay, names, n_txt)
        write_s = time.perf_counter() - w0
        # (PG_PLACE_TRIALS=1: the driver reserves its rows once, without the placement probes of a long-lived resident data set)
        r = subprocess.run(cmd, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"), stderr=subprocess.PIPE, stdout=subprocess.PIPE,
                           timeout=900)
        line = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
        if not line:
            raise RuntimeError("popgenWindows.py: " + r.stderr.decode()[-300:])
        tm = json.loads(line[-1][len("PG_TIMING "):])
        with open(csv) as f:
            rows = [ln.strip().split(",") for ln in f.readlines()]
        head, rows = rows[0], rows[1:]
        lo1 = np.zeros(1, dtype=np.int64)
        _, cols = eng.batch(lo1, lo1 + wind).groupDistTable(True, wl["min_sites"], 0.01)
        same = len(rows) == n_txt // wind
        for w, row in enumerate(rows):
            for name, v in zip(head[5:], row[5:]):
                same = same and agrees(v, t0_table[w, cols.index(name)], 4)
        # the validation run: twelve decimals against T0 to 1e-9
        deep = {}
        try:
            csvd = os.path.join(tmp, "out_deep.csv")
            rd = subprocess.run([csvd if c == csv else c for c in cmd] + DEEP, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"),
                                stderr=subprocess.PIPE, stdout=subprocess.PIPE, timeout=900)
            lined = [ln for ln in rd.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
            td = json.loads(lined[-1][len("PG_TIMING "):])
            with open(csvd) as f:
                rowsd = [ln.strip().split(",") for ln in f.readlines()]
            headd, rowsd = rowsd[0], rowsd[1:]
            samed = len(rowsd) == len(rows)
            for w, row in enumerate(rowsd):
                for name, v in zip(headd[5:], row[5:]):
                    samed = samed and agrees(v, t0_table[w, cols.index(name)], 12)
            deep = {"matches_t0_to_1e-9": bool(samed), "total_s": round(td["total_s"], 4),
                    "windows_computed_a_second_time_in_numpy_order": td.get("windows_recomputed_in_numpy_order", 0),
                    "note": "the same command with --roundTo 12: every window is computed twice (fixed trees, then NumPy's summation order for the "
                            "last digit), which is why the timed runs print the reference's default four decimals"}
        except Exception as exc:
            deep = {"error": repr(exc)[:200]}
        work_s = tm["total_s"] - tm.get("context_s", 0.0)
        stages = tm.get("tokenize_s", 0.0) + tm.get("windows_s", 0.0) + tm.get("compute_and_write_s", 0.0)
        t2 = {"sites_per_sec": round(n_txt / tm["total_s"], 1), "windows_per_sec": round(len(rows) / tm["total_s"], 3),
              "text_GBps": round(size / tm["total_s"] / 1e9, 2), "sites": n_txt, "windows": len(rows), "text_bytes": size,
              "matches_t0": bool(same), "round_to": 4, "run_at_roundTo_12": deep,
              "without_context_creation": {"seconds": round(work_s, 4), "sites_per_sec": round(n_txt / work_s, 1),
                                           "windows_per_sec": round(len(rows) / work_s, 3), "text_GBps": round(size / work_s / 1e9, 2)},
              "tokenizer_text_GBps": round(size / max(tm.get("tokenize_s", 0.0), 1e-9) / 1e9, 2),
              "tokenizer_h2d_GBps": round(size / max(tm.get("tokenizer_h2d_s", 0.0), 1e-9) / 1e9, 2) if tm.get("tokenizer_h2d_s") else None,
              "tokenizer": "device (pg_tokenize_file)" if tm.get("device_tokenizer") else "host threads (pg_encode_text)",
              "stages_overlap": bool(stages > work_s * 1.02),
              "seconds": {k: round(tm[k], 4) for k in ("total_s", "context_s", "context_create_s", "read_s", "tokenize_s", "tokenizer_h2d_s",
                                                        "tokenizer_kernels_s", "windows_s", "prep_wait_s", "engine_and_upload_s",
                                                        "compute_and_write_s", "compute_first_chunk_s", "compute_other_chunks_s",
                                                        "main_stats_s", "main_format_s") if k in tm},
              "seconds_note": "tokenize_s / windows_s: the ingestion thread; compute_first_chunk_s + compute_other_chunks_s: what the main thread "
                              "spends on the chunks (kernels, statistics, rows; main_stats_s / main_format_s inside); compute_and_write_s: the rest "
                              "of the main thread's time outside its waits, start-up of the library and the runtime included; prep_wait_s: what the main thread waited for the ingestion thread; they run beside each "
                              "other, so their sum exceeds total_s - context_s when the stages overlap",
              "sample": "the first %d sites of the workload as %.1f GB of `.geno` text (%d windows; written in %.1f s before the clock starts) "
                        "through popgenWindows.py, timed inside the driver (total_s: from opening the input to the last row; the device "
                        "context is created beside the opening of the input, context_s is what was still waited for)" % (
                            n_txt, size / 1e9, len(rows), write_s)}
        # the same sites as a `.pgeno` file with raw cells (what `parseVCF.py --packed` / tools/geno_pack.py keep instead of the text:
        # 1 byte per genotype instead of 4): the staging threads read the cells from the file, k_unpack expands them on the device
        try:
            pg, csv3 = os.path.join(tmp, "sample.pgeno"), os.path.join(tmp, "out3.csv")
            w0 = time.perf_counter()
            psize = write_pgeno_resident(pg, eng, lay, names, n_txt)
            pwrite_s = time.perf_counter() - w0
            cmd3 = [pg if c == geno else csv3 if c == csv else c for c in cmd]
            r3 = subprocess.run(cmd3, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"), stderr=subprocess.PIPE, stdout=subprocess.PIPE,
                                timeout=900)
            line3 = [ln for ln in r3.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
            if not line3:
                raise RuntimeError("popgenWindows.py: " + r3.stderr.decode()[-300:])
            tp = json.loads(line3[-1][len("PG_TIMING "):])
            with open(csv) as f, open(csv3) as g:
                same3 = f.read() == g.read()
            work3 = tp["total_s"] - tp.get("context_s", 0.0)
            t2["packed"] = {"sites_per_sec": round(n_txt / tp["total_s"], 1), "windows_per_sec": round(len(rows) / tp["total_s"], 3),
                            "file_bytes": psize, "cells_GBps": round(psize / tp["total_s"] / 1e9, 2), "csv_equals_text_run": bool(same3),
                            "from_file_to_device": bool(tp.get("packed_cells_from_file")),
                            "without_context_creation": {"seconds": round(work3, 4), "sites_per_sec": round(n_txt / work3, 1),
                                                         "cells_GBps": round(psize / work3 / 1e9, 2)},
                            "seconds": {k: round(tp[k], 4) for k in ("total_s", "context_s", "read_s", "tokenize_s", "tokenizer_h2d_s", "windows_s",
                                                                      "prep_wait_s", "compute_and_write_s") if k in tp},
                            "sample": "the same %d sites as %.1f GB of `.pgeno` (raw cells, written in %.1f s before the clock starts) through "
                                      "popgenWindows.py" % (n_txt, psize / 1e9, pwrite_s)}
        except Exception as exc:
            t2["packed"] = {"error": repr(exc)[:300]}
        # the same text the way the reference's users keep it -- `.geno.gz` (popgenWindows.py:313; parseVCF.py ... | bgzip): as BGZF
        # the members cross PCIe deflated and are inflated on the device (k_inflate, a wavefront per member); as a single gzip stream
        # (what `gzip` writes: one serial stream, nothing to parallelise) a hundredth of the sample through the gzip module
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bgzip
            gz, csv4 = os.path.join(tmp, "sample.geno.gz"), os.path.join(tmp, "out4.csv")
            w0 = time.perf_counter()
            n_in, n_gz = bgzip.bgzip_file(geno, gz)
            zip_s = time.perf_counter() - w0
            cmd4 = [gz if c == geno else csv4 if c == csv else c for c in cmd]
            r4 = subprocess.run(cmd4, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"), stderr=subprocess.PIPE, stdout=subprocess.PIPE,
                                timeout=900)
            line4 = [ln for ln in r4.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
            if not line4:
                raise RuntimeError("popgenWindows.py: " + r4.stderr.decode()[-300:])
            tb = json.loads(line4[-1][len("PG_TIMING "):])
            with open(csv) as f, open(csv4) as g:
                same4 = f.read() == g.read()
            work4 = tb["total_s"] - tb.get("context_s", 0.0)
            t2["bgzf"] = {"text_GBps": round(size / tb["total_s"] / 1e9, 2), "sites_per_sec": round(n_txt / tb["total_s"], 1),
                          "windows_per_sec": round(len(rows) / tb["total_s"], 3), "file_bytes": n_gz, "deflate_ratio": round(size / n_gz, 1),
                          "csv_equals_text_run": bool(same4),
                          "inflate": "device (k_inflate: members inflated, line feeds listed, CRC-32 checked in one kernel)" if tb.get("bgzf_blocks_inflated_on_device") else "host threads",
                          "blocks_inflated_on_device": tb.get("bgzf_blocks_inflated_on_device", 0),
                          "without_context_creation": {"seconds": round(work4, 4), "text_GBps": round(size / work4 / 1e9, 2),
                                                       "sites_per_sec": round(n_txt / work4, 1)},
                          "seconds": {k: round(tb[k], 4) for k in ("total_s", "context_s", "read_s", "first_block_prefetch_s", "tokenize_s",
                                                                    "tokenizer_h2d_s", "tokenizer_kernels_s", "windows_s", "prep_wait_s",
                                                                    "compute_and_write_s", "compute_first_chunk_s", "compute_other_chunks_s",
                                                                    "main_stats_s", "main_format_s") if k in tb},
                          "sample": "the same %d sites bgzipped (level 6, members of 65280 bytes: %.2f GB, written in %.1f s before the clock "
                                    "starts) through popgenWindows.py" % (n_txt, n_gz / 1e9, zip_s)}
            # ONE gzip stream (what `gzip` / `pigz` write: a single member, nothing that names its pieces): a tenth of the sample.  Written
            # the way pigz does it -- pieces deflated side by side, each primed with the 32 KiB in front of it (one continuous stream of
            # back references), joined by sync flushes -- because one thread of the gzip module would need a minute for it
            import zlib as _zlib
            from concurrent.futures import ThreadPoolExecutor
            n_ser = max(n_txt // 10 // wind, 2) * wind
            sgz, csv5 = os.path.join(tmp, "serial.geno.gz"), os.path.join(tmp, "out5.csv")
            w0 = time.perf_counter()
            with open(geno, "rb") as f:
                head_line = f.readline()
                body_len = 0
                lines_left = n_ser
                pos0 = f.tell()
                # the byte length of n_ser data lines: every line of the sample has its scaffold, position and cells, so count them
                while lines_left > 0:
                    chunk = f.read(64 << 20)
                    if not chunk:
                        break
                    k = chunk.count(b"\n")
                    if k <= lines_left:
                        body_len += len(chunk)
                        lines_left -= k
                    else:
                        at = -1
                        for _ in range(lines_left):
                            at = chunk.index(b"\n", at + 1)
                        body_len += at + 1
                        lines_left = 0
                f.seek(pos0)
                text5 = head_line + f.read(body_len)
            piece = 8 << 20

            def deflate_piece(a):
                zd = text5[max(a - 32768, 0):a]
                c5 = _zlib.compressobj(6, _zlib.DEFLATED, -15, 8, _zlib.Z_DEFAULT_STRATEGY, zd) if zd else _zlib.compressobj(6, _zlib.DEFLATED, -15)
                last = a + piece >= len(text5)
                return c5.compress(text5[a:a + piece]) + c5.flush(_zlib.Z_FINISH if last else _zlib.Z_SYNC_FLUSH)
            with ThreadPoolExecutor(max(2, min(16, effective_cpus()["usable"]))) as pool5:
                parts5 = list(pool5.map(deflate_piece, range(0, len(text5), piece)))
            with open(sgz, "wb") as g:
                g.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
                for p5 in parts5:
                    g.write(p5)
                g.write(struct.pack("<II", _zlib.crc32(text5) & 0xFFFFFFFF, len(text5) & 0xFFFFFFFF))
            del text5, parts5
            ser_zip_s = time.perf_counter() - w0
            cmd5 = [sgz if c == geno else csv5 if c == csv else c for c in cmd]
            r5 = subprocess.run(cmd5, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"), stderr=subprocess.PIPE, stdout=subprocess.PIPE,
                                timeout=900)
            line5 = [ln for ln in r5.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
            if not line5:
                raise RuntimeError("popgenWindows.py: " + r5.stderr.decode()[-300:])
            ts = json.loads(line5[-1][len("PG_TIMING "):])
            with open(csv) as f, open(csv5) as g:
                head_rows = f.readlines()[:1 + n_ser // wind]
                same5 = head_rows == g.readlines()
            works = ts["total_s"] - ts.get("context_s", 0.0)
            t2["gz"] = {"text_GBps": round(ts["text_bytes"] / ts["total_s"] / 1e9, 3), "sites": n_ser, "text_bytes": ts["text_bytes"],
                        "file_bytes": os.path.getsize(sgz), "csv_equals_text_run": bool(same5),
                        "without_context_creation": {"seconds": round(works, 4), "text_GBps": round(ts["text_bytes"] / works / 1e9, 3)},
                        "gzip_reader": ts.get("gzip_reader"),
                        "inflate": "host: the library's own deflate decoder, chunks of the ONE stream side by side (block starts found by trial, "
                                   "16-bit output with markers for the unknown window, chained and resolved; csrc/pg_par_gunzip.h) -- the device "
                                   "takes over at the tokenizer; `bgzip` the file to get the BGZF route, whose members the device inflates",
                        "sample": "the first %d sites as ONE gzip stream (one member, written pigz-style in %.1f s before the clock starts)" % (n_ser, ser_zip_s)}
        except Exception as exc:
            t2.setdefault("bgzf", {"error": repr(exc)[:300]})
            t2.setdefault("gz", {"error": repr(exc)[:300]})
        # the WHOLE workload in that format: every resident site as one bgzipped `.geno.gz` (the text itself is never on disk: pieces
        # of 250 000 rows are rendered from the resident rows and deflated one after the other, before the clock starts) through
        # popgenWindows.py, every cell of its CSV against the T0 table.  north star: 81 GB of text, 3.2 GB on disk, 2000 windows
        try:
            n_all = int(t0_table.shape[0]) * wind
            need = n_all * (4 * len(names) + 16) // 20
            if (os.environ.get("PG_BENCH_T2_WHOLE", "1") != "0" and n_all > n_txt and n_all % scaf_len == 0
                    and shutil.disk_usage(tmp).free > 2 * need):
                if os.path.join(ROOT, "tools") not in sys.path:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                import t2_northstar_bgzf as whole
                wgz, csv6 = os.path.join(tmp, "whole.geno.gz"), os.path.join(tmp, "out6.csv")
                w0 = time.perf_counter()
                wtext, wfile = whole.write_bgzf_resident(wgz, eng, lay, names, n_all, scaf_len)
                wwrite_s = time.perf_counter() - w0
                cmd6 = [wgz if c == geno else csv6 if c == csv else c for c in cmd]
                runs6 = []
                for _ in range(2):
                    r6 = subprocess.run(cmd6, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"), stderr=subprocess.PIPE,
                                        stdout=subprocess.PIPE, timeout=600)
                    line6 = [ln for ln in r6.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
                    if not line6:
                        raise RuntimeError("popgenWindows.py: " + r6.stderr.decode()[-300:])
                    runs6.append(json.loads(line6[-1][len("PG_TIMING "):]))
                tw = min(runs6, key=lambda x: x["total_s"])
                with open(csv6) as f:
                    rows6 = [ln.strip().split(",") for ln in f.readlines()]
                head6, rows6 = rows6[0], rows6[1:]
                same6 = len(rows6) == t0_table.shape[0]
                for w, row in enumerate(rows6):
                    k6, r6_ = divmod(w * wind, scaf_len)
                    same6 = same6 and int(row[1]) == r6_ + 1 and int(row[2]) == r6_ + wind and int(row[4]) == wind
                    for name, v in zip(head6[5:], row[5:]):
                        same6 = same6 and agrees(v, t0_table[w, cols.index(name)], 4)
                # (and once with twelve decimals, every cell against T0 to 1e-9)
                deep6 = {}
                try:
                    r6d = subprocess.run(cmd6 + DEEP, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"), stderr=subprocess.PIPE,
                                         stdout=subprocess.PIPE, timeout=600)
                    t6d = json.loads([ln for ln in r6d.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")][-1][len("PG_TIMING "):])
                    with open(csv6) as f:
                        rows6d = [ln.strip().split(",") for ln in f.readlines()]
                    head6d, rows6d = rows6d[0], rows6d[1:]
                    same6d = len(rows6d) == t0_table.shape[0]
                    for w, row in enumerate(rows6d):
                        for name, v in zip(head6d[5:], row[5:]):
                            same6d = same6d and agrees(v, t0_table[w, cols.index(name)], 12)
                    deep6 = {"matches_t0_to_1e-9": bool(same6d), "total_s": round(t6d["total_s"], 4),
                             "windows_per_sec": round(len(rows6d) / t6d["total_s"], 1),
                             "windows_computed_a_second_time_in_numpy_order": t6d.get("windows_recomputed_in_numpy_order", 0)}
                except Exception as exc:
                    deep6 = {"error": repr(exc)[:200]}
                workw = tw["total_s"] - tw.get("context_s", 0.0)
                t2["bgzf_whole_workload"] = {
                    "windows_per_sec": round(len(rows6) / tw["total_s"], 1), "sites_per_sec": round(n_all / tw["total_s"], 1),
                    "text_GBps": round(wtext / tw["total_s"] / 1e9, 2), "seconds": [round(x["total_s"], 4) for x in runs6],
                    "sites": n_all, "windows": len(rows6), "scaffolds": n_all // scaf_len, "text_bytes": wtext, "file_bytes": wfile,
                    "matches_t0": bool(same6), "round_to": 4, "run_at_roundTo_12": deep6,
                    "blocks_inflated_on_device": tw.get("bgzf_blocks_inflated_on_device", 0),
                    "without_context_creation": {"seconds": round(workw, 4), "windows_per_sec": round(len(rows6) / workw, 1),
                                                 "text_GBps": round(wtext / workw / 1e9, 2)},
                    "sample": "ALL %d sites of the workload as one bgzipped `.geno.gz` (%.1f GB of text, %.2f GB on disk, written from the "
                              "resident rows in %.0f s before the clock starts) through popgenWindows.py (the reference's default --roundTo 4), timed "
                              "inside the driver; every cell of the CSV against the T0 table (half a unit of the fourth decimal; "
                              "run_at_roundTo_12: 1e-9 relative)" % (n_all, wtext / 1e9, wfile / 1e9, wwrite_s)}
                os.remove(wgz)
        except Exception as exc:
            t2["bgzf_whole_workload"] = {"error": repr(exc)[:300]}
        # the same file on TWO ranks (both on this GPU, so the rows travel through files and the ranks share one PCIe link: not a
        # scaling number): the drivers' multi-GPU plan at the size of real data -- every rank reads, tokenises and computes its
        # window range of the ONE scaffold, the gathered CSV is the single-rank one
        try:
            csv2 = os.path.join(tmp, "out2.csv")
            cmd2 = [c if c != csv else csv2 for c in cmd]
            procs = []
            for r in range(2):
                env = dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1", PG_COMM="file", RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2",
                           MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", PG_RDZV_FILE=os.path.join(tmp, "rdzv"))
                procs.append(subprocess.Popen(cmd2, env=env, stderr=subprocess.PIPE, stdout=subprocess.PIPE))
            tms = []
            for pr in procs:
                _, err = pr.communicate(timeout=900)
                tms += [json.loads(ln[len("PG_TIMING "):]) for ln in err.decode().splitlines() if ln.startswith("PG_TIMING ")]
            with open(csv) as f, open(csv2) as g:
                same2 = f.read() == g.read()
            t2["two_ranks_one_gpu"] = {"csv_equals_single_rank": bool(same2), "window_ranges": all(t.get("window_ranges") for t in tms),
                                       "rank_bytes_share": [round(t["text_bytes"] / size, 4) for t in sorted(tms, key=lambda t: t["rank"])],
                                       "rank_sites": [t["sites"] for t in sorted(tms, key=lambda t: t["rank"])],
                                       "plan_scanned_bytes": [t.get("plan_scanned_bytes") for t in sorted(tms, key=lambda t: t["rank"])],
                                       "total_s": [round(t["total_s"], 4) for t in sorted(tms, key=lambda t: t["rank"])],
                                       "note": "both ranks on one GPU and one PCIe link, PG_COMM=file: shows the plan and the gather, not a speed-up"}
        except Exception as exc:
            t2["two_ranks_one_gpu"] = {"error": repr(exc)[:300]}
    except Exception as exc:                                    # the tiers are side information: never lose the main line
        t2 = {"error": repr(exc)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return t2


def spawn_ranks(n, script=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, the environment a launcher would
    set), pass rank 0's JSON line through and wait for all of them.  No torch anywhere."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PG_RDZV_FILE="/tmp/pg_rdzv_bench_%d_%d" % (os.getpid(), port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # all children are watched: the first one that ends with an error ends the launch (the others would wait for it at the next
    # exchange; the drivers stop by themselves when a peer leaves a failure marker, bench.py's ranks are stopped here)
    rc, live = 0, dict(enumerate(procs))
    while live and rc == 0:
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            if code != 0:
                rc = abs(code)
                sys.stderr.write("bench.py: rank %d ended with exit code %d; stopping the other %d ranks\n" % (r, code, len(live)))
        time.sleep(0.05)
    for p in live.values():
        p.terminate()
    for p in live.values():
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def main():
    from genomics_general_amd import _lib, dist, synth, windows
    from genomics_general_amd.engine import Engine
    from genomics_general_amd.samples import HapLayout, SampleData

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="northstar", choices=sorted(WORKLOADS),
                    help="default: northstar at every N (one shape per scaling curve); c5 = the rank's share of BASELINE.json configs[4], "
                         "3e9 sites over >= 8 ranks (also measured after the headline at N >= 8 and reported as `c5_share`)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: ONE data set of the workload's size, cut into window ranges by the drivers' multi-GPU plan "
                         "(genomics_general_amd.shardplan), every rank holds and computes only its range, the gather of the rows inside the time")
    ap.add_argument("--no-c5", action="store_true", help="at N >= 8: skip the c5_share measurement behind the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tiers", action="store_true", help="skip the T1 (host blocks -> H2D -> kernels) and T2 (text -> CSV) samples")
    ap.add_argument("--cpu-workers", type=int, default=1 << 30, help="upper bound of the CPU baseline's worker processes")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="no GPU: time the UNMODIFIED reference (BASELINE.json reference_path, where it exists) on a W-window prefix of the "
                         "workload with -T <usable CPUs> and -T 1 and print {\"cpu_baseline\": ...}")
    args = ap.parse_args()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        ref_dir = json.load(f).get("reference_path") or ""
    if args.cpu_baseline_only:
        wl = dict(WORKLOADS[args.workload])
        assert os.path.isdir(ref_dir), "the reference (%s) is not on this machine: the default run times the oracle's port instead" % ref_dir
        assert wl["tool"] in ("popgen", "abba"), "reference baseline: popgenWindows / ABBABABAwindows workloads"
        cpus = effective_cpus()
        names = ["s%d" % d for d in range(wl["n_dip"])]
        cpu = cpu_baseline_reference(wl, names, wl["n_pops"], ref_dir, cpus["usable"])
        cpu["host_cpus"] = cpus
        # the oracle's port (what the GPU box times, where the reference is absent) on the same windows and the same cores: how a
        # `kind: "port"` number translates into the reference's
        try:
            import multiprocessing as mp
            from genomics_general_amd import synth
            W = cpu["runs"]["T_all"]["windows"]
            per = wl["n_dip"] // wl["n_pops"]
            pops = [("pop%d" % k, names[k * per:(k + 1) * per]) for k in range(wl["n_pops"])]
            jobs = []
            for k in range(W):
                pos = np.arange(k * wl["wind"] + 1, (k + 1) * wl["wind"] + 1)
                codes = synth.gen_codes(synth.SEED_DEFAULT, np.zeros(len(pos), dtype=np.int64), pos, wl["n_dip"], wl["n_pops"])
                jobs.append((codes, names, pops, wl["wind"], wl["min_sites"], wl["tool"], "chr1", 1))
            with mp.get_context("spawn").Pool(min(W, cpus["usable"])) as pool:
                res = pool.map(_cpu_window_job, jobs, chunksize=1)
            wall = max(r[1] for r in res) - min(r[0] for r in res)
            cpu["port_same_windows"] = {"workers": min(W, cpus["usable"]), "windows": W, "wall_seconds": round(wall, 2),
                                        "windows_per_sec": round(W / wall, 5),
                                        "port_over_reference": round((W / wall) / cpu["runs"]["T_all"]["windows_per_sec"], 2),
                                        "note": "the oracle's restatement of the whole path (what `kind: port` times on the GPU box) on the same "
                                                "windows, one per worker; the reference's rate includes its serial reader and process start"}
        except Exception as exc:
            cpu["port_same_windows"] = {"error": repr(exc)[:300]}
        print(json.dumps({"cpu_baseline": cpu, "config": {"workload": wl["desc"], "name": args.workload}}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                  # a bare `python bench.py --gpus N`: this process becomes the launcher
    wl = dict(WORKLOADS[args.workload])
    world = dist.world_from_env()
    assert world.size == args.gpus, "WORLD_SIZE (%d) must equal --gpus (%d)" % (world.size, args.gpus)
    if args.workload == "c5":
        assert world.size >= 8, "the c5 workload is 3e9 sites sharded over the ranks: 150 GB per rank at 8 ranks, needs --gpus >= 8"
        wl["n_sites"] = 3_000_000_000 // world.size // (wl["n_scaf"] * wl["wind"]) * (wl["n_scaf"] * wl["wind"])

    # ---- setup (untimed) ---------------------------------------------------------------------------
    n_dip, n_pops = wl["n_dip"], wl["n_pops"]
    names = ["s%d" % d for d in range(n_dip)]
    per = n_dip // n_pops
    sd = SampleData(popNames=["pop%d" % k for k in range(n_pops)],
                    popInds=[names[k * per:(k + 1) * per] for k in range(n_pops)])
    lay = HapLayout(sd, names, "phased")
    slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
    eng = Engine(dist.device_for(world))
    eng.set_layout(lay)
    comm = dist.make_comm(eng, world)
    if world.size > 1 and isinstance(comm, dist.RcclComm):
        assert eng._comm_ranks() == world.size, "the RCCL communicator has %d ranks, expected %d" % (eng._comm_ranks(), world.size)
    tiers = world.size == 1 and not args.no_tiers and wl["tool"] == "popgen" and not args.strong

    def setup_data(wl, extra_rows_of=None):
        """Make the rank's data set resident and find its windows (untimed).  Weak scaling: rank r owns global sites
        [r*n, (r+1)*n) of an endless dense genome -- distinct scaffolds, the same shape on every rank.  --strong: ONE data set of
        n sites; the ranks cut it into window ranges with the drivers' plan (shardplan.shard_reader on synth.DenseRows, the
        `.pgeno` interface), every rank generates only its rows and streams only its windows (windows.CoordWindowStream with the
        plan's start / stop state)."""
        n_total = wl["n_sites"]
        scaf_len = n_total // wl["n_scaf"]
        d = {"scaf_len": scaf_len, "share": 1.0}
        if args.strong:
            from genomics_general_amd import shardplan
            src = synth.DenseRows(n_total, scaf_len, ["chr%d" % (k + 1) for k in range(wl["n_scaf"])])
            wp = dict(windType="coordinate", windSize=wl["wind"], stepSize=wl["wind"], overlap=0, maxDist=np.inf)
            plan = shardplan.shard_reader(src, world, comm, wp, lambda nm: True) if world.size > 1 else shardplan.Plan(None, None, 1.0)
            first, end = src._rows
            n_sites = end - first
            run_starts, run_names, positions = src.local_runs()
            stream = windows.CoordWindowStream(wl["wind"], wl["wind"], start=plan.start, stop=plan.stop)
            T, _ = stream.feed(run_starts, run_names, positions, final=True)
            d["share"] = n_sites / n_total
        else:
            n_sites, first = n_total, world.rank * n_total
            run_starts = np.arange(wl["n_scaf"], dtype=np.int64) * scaf_len
            run_names = ["chr%d" % (world.rank * wl["n_scaf"] + k + 1) for k in range(wl["n_scaf"])]
            positions = np.tile(np.arange(1, scaf_len + 1, dtype=np.int32), wl["n_scaf"])
            T = windows.coord_windows(run_starts, run_names, positions, wl["wind"], wl["wind"])
        del positions
        extra_rows = extra_rows_of(n_sites) if extra_rows_of else 0
        eng.reserve(max(n_sites, 1) + extra_rows)
        if n_sites:
            eng.synth_fill(0, n_sites, first, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
            if eng.plane_placement is not None and wl["tool"] in ("popgen", "distmat"):
                # a resident data set of 4 GiB and more (reserve() chose among placements of the rows and, on the empty rows, of the
                # planes): the planes chosen once more on the rows as the passes will read them (PG_PLANE_TRIALS=1: no such choices)
                d["planes_on_empty_rows"] = eng.plane_placement
                eng.tune_planes(n_sites, min(8, 2 * int(os.environ.get("PG_PLANE_TRIALS", "4"))), wl["wind"])
        good = T.sites >= wl["min_sites"]
        d.update(n_sites=n_sites, lo=T.lo[good], hi=T.hi[good], n_win=int(good.sum()), sites_per_step=int((T.hi[good] - T.lo[good]).sum()))
        return d

    t1_block = min(T1_BLOCK_SITES, wl["n_sites"] // wl["wind"] * wl["wind"]) // wl["wind"] * wl["wind"]
    data = setup_data(wl, (lambda n: 2 * t1_block) if tiers else None)   # the two halves of the T1 upload buffer sit behind the data set
    n_sites, scaf_len, lo, hi, n_win, sites_per_step = (data[k] for k in ("n_sites", "scaf_len", "lo", "hi", "n_win", "sites_per_step"))
    counts = comm.allgather(np.array([float(n_win), float(sites_per_step), data["share"]])) if world.size > 1 else np.array(
        [[float(n_win), float(sites_per_step), 1.0]])
    total_win, total_sites_step = int(counts[:, 0].sum()), int(counts[:, 1].sum())

    def step(lo=lo, hi=hi):
        if len(lo) == 0:                                   # (--strong with more ranks than windows: nothing of its own)
            return None, np.zeros((0, 1))
        wb = eng.batch(lo, hi)
        if wl["tool"] == "popgen":
            table, cols = wb.groupDistTable(doPairs=True, minSites=wl["min_sites"], minData=0.01)
            return None, table
        elif wl["tool"] == "popfreq":
            st = wb.groupFreqStats()
        elif wl["tool"] == "distmat":
            tab = wb.indPairTable()                  # finished individual-pair means, [window][pair]
            return {"d": tab}, tab
        else:
            st = wb.ABBABABA("pop0", "pop1", "pop2", "pop3", 0.01)
        keys = sorted(k for k in st if k != "sitesUsed")
        table = np.stack([st[k] for k in keys], axis=1)
        return st, table

    cols_seen = {}

    def gather_rows(tab, n_mine, n_of_rank):
        """the finished rows of every rank on every rank, in rank order = input order (the shares of --strong differ in size: padded
        to the largest, one all-gather)"""
        width = int(max(n_of_rank.max(), 1))
        a = np.asarray(tab, dtype=np.float64).reshape(n_mine, -1) if n_mine else None
        if "n" not in cols_seen:                           # (a rank without windows learns the row width from the others, once)
            cols_seen["n"] = int(comm.allgather(np.array([float(a.shape[1] if a is not None else 0)])).max())
        n_cols = cols_seen["n"]
        pad = np.zeros((width, n_cols))
        if a is not None:
            pad[:n_mine] = a
        allr = comm.allgather(pad.ravel()).reshape(world.size, width, n_cols)
        return np.concatenate([allr[r, :int(n_of_rank[r])] for r in range(world.size)], axis=0)

    # warm-up with every kernel family bracketed by events: it tells which family is the dominant one; the timed region then
    # brackets only that family (an event record between two kernels costs a few microseconds of GPU idle time), and the
    # per-family breakdown reported next to it comes from the warm-up pass
    cand = [_lib.K_PACK, _lib.K_PAIRWISE, _lib.K_PAIRD] if wl["tool"] in ("popgen", "distmat") else [_lib.K_SITESTATS]
    # (the finalisers and the copy of a large result table are bracketed in the breakdown pass, never "dominant": roofline is for kernels)
    # (the warm-up steps keep their result tables alive exactly as the timed loop does: a large table lands in page-locked memory
    # from a pool of two buffers, and the first DMA into a fresh page-locked buffer runs at a tenth of the PCIe rate)
    # distMat shape: a 40 MB table per pass -- its copy back runs beside the next pass's kernels (pg_set_deferred_results; the tables
    # are complete at the eng.sync() that ends the timed region).  Only where nothing reads a table between two passes.
    # The headline is the pass as distMat.py runs it -- every pass waits for its own table (no driver defers) --; the
    # deferred copy is timed behind it as an experiment (extra.deferred_result_copy_experiment), or as the line itself with PG_BENCH_DEFER=1.
    deferred = wl["tool"] == "distmat" and world.size == 1 and os.environ.get("PG_BENCH_DEFER", "0") == "1"
    if deferred:
        eng.set_deferred_results(True)
    st = _tab = None
    for _ in range(max(args.warmup - 1, 0)):
        st, _tab = step()
        if world.size > 1:
            gather_rows(_tab, n_win, counts[:, 0])
    eng.sync()
    eng.kernel_time_reset()
    n_warm = 0
    if args.warmup >= 1:                                   # the last warm-up step is the breakdown pass
        st, _tab = step()
        eng.sync()
        if world.size > 1:
            gather_rows(_tab, n_win, counts[:, 0])
        n_warm = 1
    kt = {name: eng.kernel_time(kid) for kid, name in _lib.KERNEL_NAMES.items()}
    dom_id = max(cand, key=lambda k: eng.kernel_time(k)[0]) if n_warm else None
    if dom_id is not None:
        eng.kernel_time_select([dom_id])
    eng.kernel_time_reset()
    comm.barrier()
    t0 = time.perf_counter()
    gather_s = 0.0
    for _ in range(args.steps):
        st, _tab = step()
        if world.size > 1:                                 # the drivers' one exchange per job: the finished rows meet on every rank
            g0 = time.perf_counter()
            full = gather_rows(_tab, n_win, counts[:, 0])
            gather_s += time.perf_counter() - g0
            assert full.shape[0] == total_win
    eng.sync()
    my_elapsed = time.perf_counter() - t0
    comm.barrier()
    elapsed = time.perf_counter() - t0
    if deferred:
        eng.set_deferred_results(False)                    # (what follows reads its tables at once)
    per_rank = comm.allgather(np.array([my_elapsed, elapsed])) if world.size > 1 else np.array([[my_elapsed, elapsed]])
    elapsed = float(np.max(per_rank[:, 1]))
    gather_ms = gather_s * 1e3 / args.steps if world.size > 1 else None

    # ---- per-kernel timing of the timed region (HIP events on the engine's stream) ---------------------
    # dominant kernel = the kernel family with the most GPU time (chosen in the warm-up pass, timed live here)
    if dom_id is None:                                     # --warmup 0: everything was bracketed in the timed region
        kt = {name: eng.kernel_time(kid) for kid, name in _lib.KERNEL_NAMES.items()}
        dom_id = max(cand, key=lambda k: eng.kernel_time(k)[0])
        n_warm = args.steps
    dom_ms, dom_n = eng.kernel_time(dom_id)
    eng.kernel_time_select(None)
    n_hap = lay.n_hap
    roofline = None
    extra = {}
    # rocprofv3's names of the kernel behind each family, for this workload (the library's own choices, restated)
    dip = lay.n_hap == 2 * lay.n_samp and not os.environ.get("PG_NO_DIP")
    pack_name = "k_pack2" if (os.environ.get("PG_PACK2") or n_hap > 4096) else "k_pack3"
    if os.environ.get("PG_PAIR_VALU"):
        pair_c, pair_d = "k_pairC", "k_pairD"
    else:
        # matrix cores: one wave per SIMD up to 224 units, LDS-staged block kernels where the plane fits their ring (PG_PAIR_TILE,
        # default "bc"), else one wave per block
        tile_sel = os.environ.get("PG_PAIR_TILE", "bc")
        np32 = (n_hap + 31) // 32 * 32
        npv = (n_hap // 2 + 31) // 32 * 32 if dip else np32
        pair_c = ("k_pairC_big" if ("b" in tile_sel and (n_hap // 2 if dip else n_hap) <= 224) else
                  "k_pairC_tile" if ("c" in tile_sel and 3 * 4 * npv * 16 <= 65536) else "k_pairC_fp4")
        pair_d = "k_pairD_fp4"
    rocprof_name = {_lib.K_PACK: pack_name, _lib.K_PAIRWISE: pair_c, _lib.K_PAIRD: pair_d,
                    _lib.K_SITESTATS: "k_popfreq_q" if wl["tool"] == "popfreq" else "k_abba_q"}
    pmc = {}
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                pmc = json.load(f)
        except Exception:
            pmc = {}
    if dom_n > 0:
        per_launch_s = dom_ms / dom_n / 1e3
        launches_per_step = dom_n / args.steps
        kname = rocprof_name.get(dom_id, _lib.KERNEL_NAMES[dom_id])
        alg_bytes_launch = n_hap * sites_per_step / launches_per_step       # 1 byte per haplotype allele call
        if dom_id in (_lib.K_PAIRWISE, _lib.K_PAIRD) and not os.environ.get("PG_PAIR_VALU"):
            # pair kernel on the matrix cores: algorithmic multiply-accumulates (called counts: unordered unit pairs incl. the
            # diagonal x sites; differences: two products per haplotype pair and virtual site, whose number only the device knows)
            units = n_hap // 2 if dip else n_hap
            macs = units * (units + 1) / 2 * sites_per_step / launches_per_step if dom_id == _lib.K_PAIRWISE else None
            peak = MFMA_FP4_PEAK_TFLOPS
            roofline = {"kernel": kname, "bound": "mfma", "unit": "TFLOP/s", "peak": peak,
                        "peak_source": "MX fp4 dense (MI355X_MICROARCH.md: ~10 PF; measured ceiling there 9099)",
                        "achieved": round(2 * macs / per_launch_s / 1e12, 2) if macs else None,
                        "frac": round(2 * macs / per_launch_s / 1e12 / peak, 5) if macs else None,
                        "algorithmic_macs_per_launch": macs, "traffic": pmc.get(args.workload, {}).get(kname),
                        "avg_launch_ms": round(dom_ms / dom_n, 4), "launches": int(dom_n)}
        elif dom_id in (_lib.K_PAIRWISE, _lib.K_PAIRD):
            # VALU-bound pair kernel: wave-instructions per launch from the committed PMC pass (SQ_INSTS_VALU), live launch time
            insts = pmc.get("_valu", {}).get(args.workload, {}).get(kname)
            peak_meas = pmc.get("_valu_peak_measured", {}).get(kname)
            roofline = {"kernel": kname, "bound": "valu", "unit": "wave-instr/s", "peak": VALU_PEAK_GUIDE,
                        "peak_source": "256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md)",
                        "achieved": (insts / per_launch_s) if insts else None,
                        "frac": round(insts / per_launch_s / VALU_PEAK_GUIDE, 5) if insts else None,
                        "peak_measured_mix": peak_meas,
                        "frac_of_measured_mix": round(insts / per_launch_s / peak_meas, 5) if insts and peak_meas else None,
                        "valu_wave_instructions_per_launch": insts, "traffic": pmc.get(args.workload, {}).get(kname),
                        "hbm_equivalent": {"achieved_GBps": round(alg_bytes_launch / per_launch_s / 1e9, 2),
                                           "frac_of_8TBps": round(alg_bytes_launch / per_launch_s / 1e9 / HBM_PEAK_GBS, 5),
                                           "note": "algorithmic bytes / launch time; this kernel is not HBM-bound"},
                        "avg_launch_ms": round(dom_ms / dom_n, 4), "launches": int(dom_n)}
        else:
            achieved = alg_bytes_launch / per_launch_s / 1e9
            roofline = {"kernel": kname, "bound": "hbm", "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": pmc.get(args.workload, {}).get(kname), "avg_launch_ms": round(dom_ms / dom_n, 4),
                        "launches": int(dom_n), "algorithmic_bytes_per_launch": int(alg_bytes_launch)}
        if wl["tool"] == "popgen":
            pair_ms = sum(kt[_lib.KERNEL_NAMES[k]][0] for k in (_lib.K_PAIRWISE, _lib.K_PAIRD)) / n_warm
            pair_sites = n_hap * (n_hap - 1) / 2 * sites_per_step
            extra["pair_kernels"] = {"ms_per_step": round(pair_ms, 4), "algorithmic_pair_sites_per_s": pair_sites / (pair_ms / 1e3),
                                     "naive_valu_bound": VALU_PAIRSITES_PEAK,
                                     "engine": ("VALU popcount (PG_PAIR_VALU)" if os.environ.get("PG_PAIR_VALU") else
                                                "MX fp4 MFMA on the bit planes (exact: parts below 2^23 sites, integer atomics between parts)"),
                                     "note": "pair-count kernels C + D together vs SURVEY 8d's 7-lane-op-per-32-pair-sites VALU bound; "
                                             "polymorphic-site compaction and per-individual called counts do less work than that, "
                                             "and the matrix cores are not bound by it"}
    if roofline is not None and roofline.get("traffic") is not None:
        roofline["traffic_source"] = ("profiles/hbm_traffic.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch from separate "
                                      "`rocprofv3 --pmc` passes of this command (profiles/%s/), not a live counter"
                                      % pmc.get("_source", {}).get(args.workload, "?"))
    if gather_ms is not None:
        extra["result_allgather_ms_per_step"] = round(gather_ms, 3)      # inside the timed region (rank 0's wait for the slowest rank included)
        extra["per_rank_ms_per_step"] = {"min": round(1e3 * float(per_rank[:, 0].min()) / args.steps, 4),
                                         "max": round(1e3 * float(per_rank[:, 0].max()) / args.steps, 4)}
        extra["comm_ranks"] = int(eng._comm_ranks()) if isinstance(comm, dist.RcclComm) else world.size
        extra["comm"] = "rccl" if isinstance(comm, dist.RcclComm) else "files (PG_COMM=file)"
    if getattr(eng, "placement", None):
        extra["placement_trials"] = {"probe_ms": eng.placement[0], "kept": eng.placement[1],
                                     "note": "reserve() tried these physical placements of the resident rows (empty) and kept the "
                                             "one the pack + pair path streams fastest from; PG_PLACE_TRIALS=1 takes the first"}
        if getattr(eng, "plane_placement", None):
            extra["placement_trials"]["planes_probe_ms"] = eng.plane_placement[0]
            extra["placement_trials"]["planes_kept"] = eng.plane_placement[1]
            if data.get("planes_on_empty_rows"):
                extra["placement_trials"]["planes_probe_ms_on_empty_rows"] = data["planes_on_empty_rows"][0]
            extra["placement_trials"]["planes_note"] = ("then, with the rows where they are, sets of the planes the pack kernel writes "
                                                        "(pg_tune_planes): by reserve() on the empty rows, and once more on the filled rows "
                                                        "(planes_probe_ms: ms per pass on each set; candidate 0 = the set chosen on the empty "
                                                        "rows); all of it before the warm-up, PG_PLANE_TRIALS=1 keeps the first set")
    extra["kernel_ms_per_step"] = {rocprof_name.get(kid, k): round(kt[k][0] / n_warm, 4)
                                   for kid, k in _lib.KERNEL_NAMES.items() if kt[k][1] > 0}
    if deferred:
        extra["result_table_copy"] = ("deferred (PG_BENCH_DEFER=1; no driver does this): the copy of pass k's table into page-locked memory runs on a "
                                      "stream of its own beside the kernels of pass k + 1 (pg_set_deferred_results); every table is complete at the "
                                      "synchronisation that ends the timed region")
    elif wl["tool"] == "distmat" and world.size == 1:
        # the experiment behind the headline: the same passes with the table's copy deferred
        eng.set_deferred_results(True)
        try:
            for _ in range(2):
                step()
            eng.sync()
            d0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            eng.sync()
            d_ms = (time.perf_counter() - d0) * 1e3 / args.steps
        finally:
            eng.set_deferred_results(False)
        extra["deferred_result_copy_experiment"] = {
            "ms_per_step": round(d_ms, 4),
            "note": "the copy of pass k's table runs beside the kernels of pass k + 1 (pg_set_deferred_results); not the headline: distMat.py "
                    "formats every table before it computes the next one, so no driver runs this"}
    extra["kernel_ms_per_step_source"] = ("last warm-up step (all families bracketed by events); roofline.avg_launch_ms is from the timed region"
                                           if args.warmup >= 1 else "timed region")
    extra["whole_step_hbm_frac"] = round(n_hap * sites_per_step / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)
    if wl["tool"] == "popgen" and n_hap * sites_per_step >= (8 << 30) and not os.environ.get("PG_OVERLAP") and not args.no_tiers:
        # side information: the same pass cut into >= 8 sub-batches whose pack kernel runs on a second stream beside the pair
        # kernels of the previous sub-batch (PG_OVERLAP=1).  Not the default: the kernels then share the GPU and a per-kernel
        # event bracket no longer times one kernel
        os.environ["PG_OVERLAP"] = "1"
        try:
            step()
            eng.sync()
            p0 = time.perf_counter()
            for _ in range(3):
                step()
            eng.sync()
            extra["ms_per_step_pipelined_sub_batches"] = round((time.perf_counter() - p0) / 3 * 1e3, 4)
        finally:
            del os.environ["PG_OVERLAP"]

    # ---- CPU baseline: the oracle's restatement of the reference's whole path on all host cores, bounded sample -------------
    cpu = None
    if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:
        if wl["tool"] == "distmat":
            cpu = cpu_baseline_distmat(eng, lay, wl, lo, hi)
        else:
            cpu = cpu_baseline_full_path(eng, lay, wl, names, lo, hi, scaf_len, args.cpu_workers)
            if os.path.isdir(ref_dir) and wl["tool"] in ("popgen", "abba"):
                # the reference itself is on this machine: it is the baseline, the port's legs stay beside it
                port = cpu
                try:
                    cpu = cpu_baseline_reference(wl, names, n_pops, ref_dir, port["host_cpus"]["usable"])
                    cpu["host_cpus"], cpu["port"] = port["host_cpus"], port
                except Exception as exc:
                    port["reference_error"] = repr(exc)[:300]
                    cpu = port

    if tiers:
        try:
            extra.update(tier_samples(eng, lay, wl, names, slot_gen, scaf_len, n_sites, t1_block, np.asarray(_tab)))
        except Exception as exc:
            extra["t1"] = {"error": repr(exc)[:300]}
    # ---- N >= 8: the rank's share of BASELINE.json configs[4] (3e9 sites / N: 150 GB resident at N = 8) behind the headline.  The
    # headline stays on ONE shape for N = 1, 2, 4, 8, so that a scaling curve compares like with like; this is the same pass at the
    # size config 5 asks for, a few steps, the gather inside the time
    c5_total = int(os.environ.get("PG_BENCH_C5_SITES", 3_000_000_000))        # (a smaller total lets a test run this section on one GPU)
    if world.size >= 8 and (args.workload == "northstar" or "PG_BENCH_C5_SITES" in os.environ) and not args.strong and not args.no_c5:
        # (every rank decides together after each phase: a rank that fails -- out of memory on its 150 GB -- must not leave the others
        # waiting in an exchange it never enters)
        def all_ranks_ok(fn):
            ok, out, msg = 1.0, None, ""
            try:
                out = fn()
            except Exception as exc:
                ok, msg = 0.0, repr(exc)[:300]
            good = bool(comm.allgather(np.array([ok])).min() > 0)
            return good, out, msg
        try:
            wl5 = dict(WORKLOADS["c5"])
            wl5["n_sites"] = c5_total // world.size // (wl5["n_scaf"] * wl5["wind"]) * (wl5["n_scaf"] * wl5["wind"])
            if wl["n_dip"] != wl5["n_dip"]:
                wl5.update(n_dip=wl["n_dip"], n_pops=wl["n_pops"])           # (the layout of the run is the headline workload's)
            good, d5, msg = all_ranks_ok(lambda: setup_data(wl5))
            if good:
                good, _, msg = all_ranks_ok(lambda: (step(d5["lo"], d5["hi"]), eng.sync()))
            if not good:
                raise RuntimeError("a rank could not set up its share: " + (msg or "(another rank)"))
            c5_counts = np.full(world.size, float(d5["n_win"]))
            cols_seen.clear()
            comm.barrier()
            c0 = time.perf_counter()
            for _ in range(3):
                _, tab5 = step(d5["lo"], d5["hi"])
                gather_rows(tab5, d5["n_win"], c5_counts)
            eng.sync()
            comm.barrier()
            dt5 = float(np.max(comm.allgather(np.array([time.perf_counter() - c0])))) / 3
            extra["c5_share"] = {"workload": wl5["desc"], "sites_per_gpu": d5["sites_per_step"], "windows_per_gpu": d5["n_win"],
                                 "resident_GB_per_gpu": round(d5["sites_per_step"] * lay.n_hap / 1e9, 1), "steps": 3,
                                 "ms_per_step": round(dt5 * 1e3, 3), "windows_per_sec": round(d5["n_win"] * world.size / dt5, 1),
                                 "sites_per_sec": round(d5["sites_per_step"] * world.size / dt5, 1)}
        except Exception as exc:                                # side information: never lose the main line
            extra["c5_share"] = {"error": repr(exc)[:300]}
    if cpu and cpu.get("kind") == "port" and args.workload in ("northstar", "c2"):
        # what the port's number means in units of the reference: both were timed on the same windows and the same cores where the
        # reference exists (bench.py --cpu-baseline-only in the build container; the committed line)
        try:
            cal_path = next(pth for pth in (os.path.join(ROOT, "profiles", rr, "cpu_baseline_reference_%s_build_container.json" % args.workload)
                                            for rr in ("r06", "r04")) if os.path.exists(pth))
            with open(cal_path) as f:
                cal = json.load(f)["cpu_baseline"]
            ratio = cal["port_same_windows"]["port_over_reference"]
            cpu["reference_calibration"] = {"port_over_reference": ratio, "reference_equivalent_windows_per_sec": round(cpu["value"] / ratio, 5),
                                            "source": os.path.relpath(cal_path, ROOT),
                                            "note": "the unmodified reference (-T 8) and this port on the same 8 windows and the same 8 vCPU of the "
                                                    "build container: the port is that many times faster; an estimate, not a measurement on this host.  "
                                                    "(The reference is Python: by the rules of this build it cannot travel to the GPU box in any form, "
                                                    "so `kind` is \"port\" there and \"reference\" only where /root/reference exists.)"}
        except Exception:
            pass
    t2 = extra.get("t2") or {}
    if cpu and "windows_per_sec" in t2 and cpu.get("unit") == "windows/s" and cpu.get("value"):
        # like for like: both read `.geno` text and write the CSV (T0's `value` has its inputs resident in HBM and is NOT comparable)
        extra["t2_vs_cpu"] = {"ratio": round(t2["windows_per_sec"] / cpu["value"], 1),
                              "ratio_without_context_creation": round(t2["without_context_creation"]["windows_per_sec"] / cpu["value"], 1),
                              "gpu_windows_per_sec": t2["windows_per_sec"], "cpu_windows_per_sec": cpu["value"], "cpu_kind": cpu.get("kind"),
                              "ratio_to_reference_equivalent": (round(t2["windows_per_sec"] / cpu["reference_calibration"]["reference_equivalent_windows_per_sec"], 1)
                                                                if "reference_calibration" in cpu else None),
                              "note": "tier T2 (text in, CSV out, one GPU + its host threads) against the CPU baseline's whole path on every host core"}
        whole = t2.get("bgzf_whole_workload") or {}
        if whole.get("windows_per_sec"):                      # the whole workload as the reference's default input (`.geno.gz`)
            extra["t2_vs_cpu"]["bgzf_whole_workload"] = {
                "ratio": round(whole["windows_per_sec"] / cpu["value"], 1), "gpu_windows_per_sec": whole["windows_per_sec"],
                "ratio_to_reference_equivalent": (round(whole["windows_per_sec"] / cpu["reference_calibration"]["reference_equivalent_windows_per_sec"], 1)
                                                  if "reference_calibration" in cpu else None)}
    if world.rank == 0:
        total_windows = total_win * args.steps
        total_sites = total_sites_step * args.steps
        if args.strong:
            extra["strong"] = {"rank_bytes_share": [round(float(x), 4) for x in counts[:, 2]],
                               "windows_per_rank": [int(x) for x in counts[:, 0]],
                               "plan": "genomics_general_amd.shardplan (window ranges inside scaffold runs) on the one data set; every rank "
                                       "generates, holds and computes only its range"}
        line = {
            "metric": "windows_per_sec", "value": round(total_windows / elapsed, 3), "unit": "windows/s",
            "sites_per_sec": round(total_sites / elapsed, 1),
            "n_gpus": world.size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u32 bit-planes / int32 counts / f64 statistics", "data": "synthetic",
            "config": {"workload": wl["desc"], "name": args.workload, "tier": "T0 (inputs resident in HBM; SURVEY.md 8d)",
                       "windows_per_gpu": n_win, "sites_per_gpu": sites_per_step, "haplotypes": n_hap,
                       "parallelism": ("one data set cut into %d window ranges" if args.strong else "windows sharded, dp%d") % world.size},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line))
    comm.barrier()
    comm.close()
    eng.close()


if __name__ == "__main__":
    main()
