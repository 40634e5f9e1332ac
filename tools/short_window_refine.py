#!/usr/bin/env python
"""Short windows: what does the reference's last digit cost, and what would the long-window route cost instead?  
    python tools/short_window_refine.py [n_sites] [n_dip]
The C2 data set (10^7 sites x 100 diploids, 4 populations) in windows of 500 / 1000 / 2000 / 4000 sites:
  np_order      every window's sums in NumPy's order (k_popdist_np: today's route for windows of up to 4096 sites)
  fixed_tree    every window's sums in the fixed trees (k_popdist_fin)
  flagged       share of the windows in which a printed value (--roundTo 4) lies within reach of a rounding tie of the fixed-tree value
                (cli._near_rounding_tie: the windows the long-window route would compute again in NumPy's order)
  refine        fixed trees everywhere + the flagged windows again in NumPy's order (pack, pair kernels and finisher of those windows)
and whether the refined table prints like the NumPy-order table (it must: that is the route's claim)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomics_general_amd import cli, synth                                     # noqa: E402
from genomics_general_amd.engine import Engine                                 # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData                 # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_dip = int(sys.argv[2]) if len(sys.argv) > 2 else 100
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // 4
sd = SampleData(popNames=["pop%d" % k for k in range(4)], popInds=[names[k * per:(k + 1) * per] for k in range(4)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
os.environ.setdefault("PG_PLACE_TRIALS", "1")
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, n_sites, n_dip, 4, slot_gen, synth.VAR_THR, synth.MISS_THR)


def stats(lo, hi, mode):
    e.set_sum_order(mode)
    try:
        return e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    finally:
        e.set_sum_order(0)


def timed(fn, reps=5):
    fn()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    e.sync()
    return (time.perf_counter() - t0) / reps * 1e3, out


print("%6s %8s | %9s %10s | %8s %9s | %9s | %s" % ("window", "windows", "np_order", "fixed_tree", "flagged", "share", "refine", "refined text == NumPy-order text"))
for w in (500, 1000, 2000, 4000):
    lo = np.arange(0, n_sites - w + 1, w, dtype=np.int64)
    hi = lo + w
    t_np, s_np = timed(lambda: stats(lo, hi, 1))
    t_fx, s_fx = timed(lambda: stats(lo, hi, 2))
    near = np.zeros(len(lo), dtype=bool)
    for k, v in s_fx.items():
        if np.asarray(v).dtype.kind == "f":
            near |= cli._near_rounding_tie(v, 4, difference=k.startswith("Fst_"))

    def refine():
        s = stats(lo, hi, 2)
        if near.any():
            s2 = stats(lo[near], hi[near], 1)
            for k in s:
                a = np.array(s[k], copy=True)
                a[near] = s2[k]
                s[k] = a
        return s
    t_rf, s_rf = timed(refine)
    same = all(np.array_equal(np.round(s_rf[k], 4), np.round(s_np[k], 4), equal_nan=True) and
               np.array_equal(np.signbit(np.round(s_rf[k], 4)), np.signbit(np.round(s_np[k], 4))) for k in s_np)
    wrong = sum(int((np.round(s_fx[k], 4) != np.round(s_np[k], 4)).sum() - (np.isnan(s_fx[k]) & np.isnan(s_np[k])).sum()) for k in s_np)
    print("%6d %8d | %7.3f ms %7.3f ms | %8d %9.2e | %6.3f ms | %s (fixed tree alone: %d cells print differently)" % (
        w, len(lo), t_np, t_fx, int(near.sum()), near.mean(), t_rf, same, wrong), flush=True)
e.close()
