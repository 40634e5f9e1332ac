#!/usr/bin/env python
"""Does k_pack2's run-to-run spread (0.46 vs 0.50 ms on C2) come with the allocation?  Several engines in ONE process, each with
its own resident buffer and scratch, same data, same kernel: per-engine average k_pack2 time over 10 passes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomics_general_amd import _lib, synth, windows                       # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

n_dip, n_pops, n_sites, n_scaf, wind = 100, 4, 10_000_000, 4, 50_000
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // n_pops
sd = SampleData(popNames=["pop%d" % k for k in range(n_pops)], popInds=[names[k * per:(k + 1) * per] for k in range(n_pops)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
scaf_len = n_sites // n_scaf
run_starts = np.arange(n_scaf, dtype=np.int64) * scaf_len
positions = np.tile(np.arange(1, scaf_len + 1, dtype=np.int32), n_scaf)
T = windows.coord_windows(run_starts, ["chr%d" % (k + 1) for k in range(n_scaf)], positions, wind, wind)
engines = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(n_sites)
    e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
    engines.append(e)                                                       # kept alive: later engines get other addresses
    for rep in range(2):
        for _ in range(3):
            e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
        e.sync()
        e.kernel_time_reset()
        for _ in range(10):
            e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
        e.sync()
        ms, n = e.kernel_time(_lib.K_PACK)
        print("engine %d rep %d: k_pack2 %.4f ms (%d launches)" % (k, rep, ms / n, n), flush=True)
