#!/usr/bin/env python
"""Benchmark of the hot path: popgenWindows pi / dxy / Fst over 50 kb coordinate windows (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c2|northstar|c3]

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher).  A step is one pass of the whole per-window
statistics path over the rank's resident synthetic data set: pack -> pairwise D/C -> population sums on the GPU,
the float64 finalisation (pi, dxy, Fst) on the GPU and D2H of the result table.  The data path has no collective (windows are
independent): at N>1 the ranks meet in an RCCL barrier on both sides of the timed region, and the per-window tables are
all-gathered once after it (what the drivers do once per input block before rank 0 writes the CSV).  Inputs are generated on the device before the timed region (counter-based generator,
genomics_general_amd/synth.py) and stay resident in HBM.  Weak scaling: every rank owns a full-size data set
(different sites), `value` = windows of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, timed with HIP events on the
stream it runs on (the kernel family with the most GPU time in the timed region); `cpu_baseline` is the CPU oracle's faithful pair-loop port timed on a bounded sample
(N=1, rank 0 only).  No torch anywhere: barriers and the gather go through RCCL in libpopgen_hip.so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from genomics_general_amd import _lib, dist, synth, windows  # noqa: E402
from genomics_general_amd.engine import Engine  # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: 10^7 sites x 100 diploids, 4 pops, 50 kb windows
    "c2": dict(n_sites=10_000_000, n_scaf=4, n_dip=100, n_pops=4, wind=50_000, min_sites=100, tool="popgen",
               desc="popgenWindows pi/Fst/Dxy: 1e7 sites x 100 diploids (200 haplotypes), 4 pops, 50 kb windows"),
    # the C2 data set in 5 kb windows (2000 windows of 157 words): per-window fixed costs of the pair kernels; not a BASELINE.json config
    "c2_w5k": dict(n_sites=10_000_000, n_scaf=4, n_dip=100, n_pops=4, wind=5_000, min_sites=100, tool="popgen",
                   desc="popgenWindows pi/Fst/Dxy: 1e7 sites x 100 diploids (200 haplotypes), 4 pops, 5 kb windows"),
    # north-star single-GPU shape: first 10^8 sites of config 5 (200 diploids)
    "northstar": dict(n_sites=100_000_000, n_scaf=4, n_dip=200, n_pops=4, wind=50_000, min_sites=100, tool="popgen",
                      desc="popgenWindows pi/Fst/Dxy: 1e8 sites x 200 diploids (400 haplotypes), 4 pops, 50 kb windows"),
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
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PAIRSITES_PEAK = 3.6e14   # 7 lane-ops per 32 pair-sites at 7.9e13 lane-ops/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=2, help="windows of the workload timed on the CPU port")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    world = dist.world_from_env()
    assert world.size == args.gpus, "WORLD_SIZE (%d) must equal --gpus (%d)" % (world.size, args.gpus)

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
    comm = dist.RcclComm(eng, world) if world.size > 1 else dist.SoloComm()
    n_sites = wl["n_sites"]
    scaf_len = n_sites // wl["n_scaf"]
    eng.reserve(n_sites)
    # rank r owns global sites [r*n_sites, (r+1)*n_sites): distinct scaffolds, same shape
    eng.synth_fill(0, n_sites, world.rank * n_sites, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen,
                   synth.VAR_THR, synth.MISS_THR)
    run_starts = np.arange(wl["n_scaf"], dtype=np.int64) * scaf_len
    run_names = ["chr%d" % (world.rank * wl["n_scaf"] + k + 1) for k in range(wl["n_scaf"])]
    positions = np.tile(np.arange(1, scaf_len + 1, dtype=np.int32), wl["n_scaf"])
    T = windows.coord_windows(run_starts, run_names, positions, wl["wind"], wl["wind"])
    del positions
    good = T.sites >= wl["min_sites"]
    lo, hi = T.lo[good], T.hi[good]
    n_win = int(len(lo))
    sites_per_step = int((hi - lo).sum())

    def step():
        wb = eng.batch(lo, hi)
        if wl["tool"] == "popgen":
            table, cols = wb.groupDistTable(doPairs=True, minSites=wl["min_sites"], minData=0.01)
            return None, table                       # the named statistics for the oracle check: stats_for_check()
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

    def stats_for_check():
        """{statistic: array over windows} of one (untimed) pass, for the comparison with the CPU port"""
        if wl["tool"] == "popgen":
            return eng.batch(lo, hi).groupDistStats(doPairs=True, minSites=wl["min_sites"], minData=0.01)
        return step()[0]

    # warm-up with every kernel family bracketed by events: it tells which family is the dominant one; the timed region then
    # brackets only that family (an event record between two kernels costs a few microseconds of GPU idle time), and the
    # per-family breakdown reported next to it comes from the warm-up pass
    cand = [_lib.K_PACK, _lib.K_PAIRWISE, _lib.K_PAIRD] if wl["tool"] in ("popgen", "distmat") else [_lib.K_SITESTATS]
    for _ in range(max(args.warmup - 1, 0)):
        step()
    eng.sync()
    eng.kernel_time_reset()
    n_warm = 0
    if args.warmup >= 1:                                   # the last warm-up step is the breakdown pass
        step()
        eng.sync()
        n_warm = 1
    kt = {name: eng.kernel_time(kid) for kid, name in _lib.KERNEL_NAMES.items()}
    dom_id = max(cand, key=lambda k: eng.kernel_time(k)[0]) if n_warm else None
    if dom_id is not None:
        eng.kernel_time_select([dom_id])
    eng.kernel_time_reset()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, _tab = step()
    eng.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = float(np.max(comm.allgather(np.array([elapsed])))) if world.size > 1 else elapsed
    gather_ms = None
    if world.size > 1:                                     # the result exchange of the drivers, once, outside the timed region
        g0 = time.perf_counter()
        full = dist.gather_table(comm, np.asarray(_tab, dtype=np.float64).reshape(n_win, -1), n_win * world.size)
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert full.shape[0] == n_win * world.size

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
    if dom_n > 0:
        per_launch_s = dom_ms / dom_n / 1e3
        launches_per_step = dom_n / args.steps
        alg_bytes_launch = n_hap * sites_per_step / launches_per_step       # 1 byte per haplotype allele call
        achieved = alg_bytes_launch / per_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f).get(args.workload, {})
                # rocprof kernel names of the family (the v2 pack kernel is k_pack2, the ABBA kernel k_abba_q)
                alias = {_lib.K_PACK: ["k_pack2", "k_pack"], _lib.K_PAIRWISE: ["k_pairC", "k_pairwise"],
                         _lib.K_PAIRD: ["k_pairD"], _lib.K_SITESTATS: ["k_abba_q", "k_popfreq"]}
                traffic = next((tj[n] for n in alias.get(dom_id, []) if n in tj), None)
            except Exception:
                traffic = None
        roofline = {"kernel": _lib.KERNEL_NAMES[dom_id], "bound": "hbm", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "avg_launch_ms": round(dom_ms / dom_n, 4), "launches": int(dom_n),
                    "algorithmic_bytes_per_launch": int(alg_bytes_launch)}
        if wl["tool"] == "popgen":
            pair_ms = sum(kt[_lib.KERNEL_NAMES[k]][0] for k in (_lib.K_PAIRWISE, _lib.K_PAIRD)) / n_warm
            pair_sites = n_hap * (n_hap - 1) / 2 * sites_per_step
            extra["pair_kernels"] = {"ms_per_step": round(pair_ms, 4), "algorithmic_pair_sites_per_s": pair_sites / (pair_ms / 1e3),
                                     "naive_valu_bound": VALU_PAIRSITES_PEAK,
                                     "note": "k_pairC + k_pairD together vs SURVEY 8d's 7-lane-op-per-32-pair-sites bound; "
                                             "polymorphic-site compaction and per-individual called counts do less work than that"}
    if gather_ms is not None:
        extra["result_allgather_ms_once_untimed"] = round(gather_ms, 3)
    extra["kernel_ms_per_step"] = {k: round(v[0] / n_warm, 4) for k, v in kt.items() if v[1] > 0}
    extra["kernel_ms_per_step_source"] = ("last warm-up step (all families bracketed by events); roofline.avg_launch_ms is from the timed region"
                                           if args.warmup >= 1 else "timed region")

    # ---- CPU baseline: the oracle's faithful port of the reference algorithm, bounded sample -------------
    cpu = None
    if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:
        from oracle import popgen_oracle as orc
        st = stats_for_check()
        # at least --cpu-windows windows, then more until about 10 s of CPU work have been sampled (at most 8 windows)
        nw_min, nw_max = max(1, min(args.cpu_windows, n_win)), max(1, min(max(args.cpu_windows, 8), n_win))
        t_cpu, t_wall = 0.0, 0.0
        ok = True
        nw = 0
        for w in range(nw_max):
            if w >= nw_min and t_wall >= 10.0:
                break
            nw += 1
            codes = eng.download(int(lo[w]), int(hi[w] - lo[w]))
            scale = 1.0
            if wl["tool"] == "distmat" and n_hap > CPU_DISTMAT_HAPS:
                # the pair loop is quadratic in haplotypes: time the first CPU_DISTMAT_HAPS of them and scale by the pair count
                scale = (n_hap * (n_hap - 1) / 2) / (CPU_DISTMAT_HAPS * (CPU_DISTMAT_HAPS - 1) / 2)
                aln, _ = orc.aln_from_codes(codes[:, :CPU_DISTMAT_HAPS], lay.hap_names[:CPU_DISTMAT_HAPS],
                                            lay.hap_sample_name[:CPU_DISTMAT_HAPS], lay.hap_group[:CPU_DISTMAT_HAPS])
            else:
                aln, _ = orc.aln_from_codes(codes, lay.hap_names, lay.hap_sample_name, lay.hap_group)
            c0 = time.perf_counter()
            if wl["tool"] == "popgen":
                D, C = orc.pair_counts_loop(aln)                     # genomics.py:903-916 + 1042-1047, pair by pair
                so, _ = orc.group_dist_stats(aln, D, C, True, wl["min_sites"], 0.01)
            elif wl["tool"] == "popfreq":
                so = orc.group_freq_stats(aln)
            elif wl["tool"] == "distmat":
                D, C = orc.pair_counts_loop(aln)
                so = {}
            else:
                so = orc.abbababa(aln, "pop0", "pop1", "pop2", "pop3", 0.01)
            dt = time.perf_counter() - c0
            t_wall += dt
            t_cpu += dt * scale
            for k, v in so.items():
                if k == "sitesUsed":
                    ok = ok and int(st[k][w]) == int(v)
                    continue
                g = st[k][w]
                ok = ok and (abs(g - v) <= 1e-6 * max(1.0, abs(v)) or (g != g and v != v))
        cpu = {"value": round(nw / t_cpu, 5), "unit": "windows/s", "cores": 1, "kind": "port",
               "sample": "first %d windows (%d sites x %d haplotypes each) of the workload, numeric core only "
                         "(no text parsing / alignment build, which dominate the real reference)%s" % (
                             nw, wl["wind"], n_hap,
                             "; pair loop timed on the first %d haplotypes and scaled by the pair count" % CPU_DISTMAT_HAPS
                             if wl["tool"] == "distmat" and n_hap > CPU_DISTMAT_HAPS else ""),
               "seconds": round(t_cpu, 2), "gpu_matches_oracle_on_sample": bool(ok)}

    if world.rank == 0:
        total_windows = n_win * world.size * args.steps
        total_sites = sites_per_step * world.size * args.steps
        line = {
            "metric": "windows_per_sec", "value": round(total_windows / elapsed, 3), "unit": "windows/s",
            "sites_per_sec": round(total_sites / elapsed, 1),
            "n_gpus": world.size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 bit-planes / int32 counts / f64 statistics", "data": "synthetic",
            "config": {"workload": wl["desc"], "name": args.workload, "windows_per_gpu": n_win,
                       "sites_per_gpu": sites_per_step, "haplotypes": n_hap, "parallelism": "windows sharded, dp%d" % world.size},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line))
    comm.barrier()
    eng.close()


if __name__ == "__main__":
    main()
