#!/usr/bin/env python
"""The pack kernel's placement lottery, finer: is "slow" a property of the physical memory behind a stretch of the ROWS, whatever
the planes it writes to sit on?  For every fresh allocation of the north-star rows: the kernel's time over 64 slices of the rows
(31 windows = 625 MB each), then the same after the plane buffers (called plane, virtual-site planes) have been released and
allocated again.  If the pattern of fast and slow slices stays, it belongs to the rows' memory.

    python tools/pack_placement2.py [n_trials] [n_slices]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomics_general_amd import _lib, synth, windows                       # noqa: E402
from genomics_general_amd._lib import check                                  # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_slices = int(sys.argv[2]) if len(sys.argv) > 2 else 64
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
os.environ["PG_PLACE_TRIALS"] = "1"


def pack_ms(e, lo, hi, passes=5):
    e.batch(lo, hi).groupDistTable(True, 100, 0.01)
    e.sync()
    e.kernel_time_reset()
    for _ in range(passes):
        e.batch(lo, hi).groupDistTable(True, 100, 0.01)
    e.sync()
    ms, n = e.kernel_time(_lib.K_PACK)
    return round(ms / n, 4)


def slices(e):
    n_win = len(T.lo)
    edges = [n_win * k // n_slices for k in range(n_slices + 1)]
    return [pack_ms(e, T.lo[a:b].copy(), T.hi[a:b].copy()) for a, b in zip(edges, edges[1:])]


def addr(e, which):
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(L.pg_debug_address(e._h, which, C.byref(a), C.byref(b)))
    return hex(a.value)


e = Engine(0)
e.set_layout(lay)
spacer = Engine(0)
spacer.set_layout(lay)
for trial in range(n_trials):
    check(L.pg_debug_place(e._h, 0, 0))
    spacer.reserve((trial % 4 + 1) * 3_000_000)
    if trial % 4 == 3:
        check(L.pg_debug_place(spacer._h, 0, 0))
    e.reserve(n_sites)
    e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
    whole = pack_ms(e, T.lo, T.hi, 3)
    a = slices(e)
    planes_before = (addr(e, 1), addr(e, 2))
    check(L.pg_debug_place(e._h, 1, 0))
    check(L.pg_debug_place(e._h, 2, 0))
    whole2 = pack_ms(e, T.lo, T.hi, 3)
    b = slices(e)
    med = float(np.median(a))
    print(json.dumps({"trial": trial, "rows_at": addr(e, 0), "whole_ms": whole, "whole_ms_planes_allocated_again": whole2,
                      "planes_at": planes_before, "planes_again_at": (addr(e, 1), addr(e, 2)),
                      "slice_ms": a, "slice_ms_planes_allocated_again": b,
                      "slow_slices(>3% above the median)": [k for k, v in enumerate(a) if v > 1.03 * med],
                      "slow_slices_planes_allocated_again": [k for k, v in enumerate(b) if v > 1.03 * float(np.median(b))],
                      "correlation": round(float(np.corrcoef(a, b)[0, 1]), 3)}), flush=True)
