#!/usr/bin/env python
"""Does the pack kernel's time on an EMPTY (zeroed) resident buffer predict its time on the same physical allocation once the
data is there?  The kernel's time moves by +-6 % with the physical pages behind the 40 GB of rows (tools/attic/pack_variance.py); if a
probe on the fresh allocation tells the two apart, reserve() could try a few allocations and keep a fast one.

    python tools/placement_probe.py [n_trials]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomics_general_amd import _lib, synth, windows                       # noqa: E402
from genomics_general_amd._lib import check                                  # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_dip, n_pops, n_sites, n_scaf, wind = 200, 4, 100_000_000, 4, 50_000
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // n_pops
sd = SampleData(popNames=["pop%d" % k for k in range(n_pops)], popInds=[names[k * per:(k + 1) * per] for k in range(n_pops)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
scaf_len = n_sites // n_scaf
run_starts = np.arange(n_scaf, dtype=np.int64) * scaf_len
positions = np.tile(np.arange(1, scaf_len + 1, dtype=np.int32), n_scaf)
T = windows.coord_windows(run_starts, ["chr%d" % (k + 1) for k in range(n_scaf)], positions, wind, wind)
del positions
L = _lib.lib()


def pack_ms(e, passes=3):
    e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    e.kernel_time_reset()
    for _ in range(passes):
        e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    ms, n = e.kernel_time(_lib.K_PACK)
    return ms / n


def addr(e):
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(L.pg_debug_address(e._h, 0, C.byref(a), C.byref(b)))
    return a.value


e = Engine(0)
e.set_layout(lay)
spacer = Engine(0)
spacer.set_layout(lay)
for trial in range(n_trials):
    check(L.pg_debug_place(e._h, 0, 0))                  # the rows are released ...
    spacer.reserve((trial % 4 + 1) * 3_000_000)           # ... something else takes a piece of what they leave (1.2 GB steps)
    if trial % 4 == 3:
        check(L.pg_debug_place(spacer._h, 0, 0))
    e.reserve(n_sites)                                     # ... and allocated again: zero rows
    t_zero = pack_ms(e)
    e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
    t_real = pack_ms(e)
    t_real2 = pack_ms(e)
    print("trial %d  rows at %#x   pack on zero rows %.3f ms   with data %.3f / %.3f ms" % (trial, addr(e), t_zero, t_real, t_real2), flush=True)
