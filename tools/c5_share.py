#!/usr/bin/env python
"""One rank's share of BASELINE configs[4] at N = 8 on ONE GPU: 3.75e8 sites x 200 diploids (150 GB of resident rows, 7500 windows
of 50 kb), the popgenWindows pi / dxy / Fst pass; two windows checked against the oracle.   python tools/c5_share.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genomics_general_amd import _lib, synth                                # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_dip, n_pops, wind = 200, 4, 50_000
n_sites = 3_000_000_000 // 8 // (3 * wind) * (3 * wind)                      # 3 scaffolds per rank
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // n_pops
sd = SampleData(popNames=["pop%d" % k for k in range(n_pops)], popInds=[names[k * per:(k + 1) * per] for k in range(n_pops)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
e = Engine(0)
e.set_layout(lay)
t0 = time.perf_counter()
e.reserve(n_sites)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, n_sites // 3, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
e.sync()
print("resident: %d sites x %d haplotypes = %.1f GB, generated in %.1f s (placement trials: %s)" % (
    n_sites, lay.n_hap, n_sites * e.row_pitch / 1e9, time.perf_counter() - t0, e.placement), flush=True)
lo = np.arange(0, n_sites, wind, dtype=np.int64)
hi = lo + wind
tab, cols = e.batch(lo, hi).groupDistTable(True, 100, 0.01)
e.sync()
e.kernel_time_reset()
t0 = time.perf_counter()
for _ in range(steps):
    tab, cols = e.batch(lo, hi).groupDistTable(True, 100, 0.01)
e.sync()
dt = (time.perf_counter() - t0) / steps
kt = {name: e.kernel_time(kid) for kid, name in _lib.KERNEL_NAMES.items()}
print("%d windows per pass: %.2f ms per pass = %.3e windows/s = %.3e sites/s; kernel ms per pass: %s" % (
    len(lo), dt * 1e3, len(lo) / dt, n_sites / dt, {k: round(v[0] / steps, 3) for k, v in kt.items() if v[1]}), flush=True)
assert np.all(np.isfinite(tab)), "non-finite statistics"
from oracle import popgen_oracle as orc                                      # noqa: E402  (checker only)
for w in (0, len(lo) - 1):
    codes = e.download(int(lo[w]), wind)
    aln, _ = orc.aln_from_codes(codes, lay.hap_names, lay.hap_sample_name, lay.hap_group)
    Do, Co = orc.pair_counts_gemm(aln)
    so, _ = orc.group_dist_stats(aln, Do, Co, True, 100, 0.01)
    for key, v in so.items():
        if key not in cols:                                              # (the oracle's dict holds both key orders of a pair)
            continue
        g = tab[w, cols.index(key)]
        assert abs(g - v) <= 1e-9 * max(1.0, abs(v)), (w, key, g, v)
print("windows 0 and %d match the oracle (1e-9)" % (len(lo) - 1))
