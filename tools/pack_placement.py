#!/usr/bin/env python
"""VERDICT round 5 #4b: what is the 12 % placement lottery of the pack kernel, and does the order of its blocks remove it?
For every fresh physical allocation of the north-star rows (released and reserved again with something else taking a piece of
what they leave, PG_PLACE_TRIALS=1 -- no probes), in one process:

  * the pack kernel's time per pass with the windows in order (PG_PACK_PERM unset) and with concurrently running blocks spread
    over the whole batch (PG_PACK_PERM = 8, 32, 128: consecutive blockIdx.y are n/k windows apart);
  * the same over eight slices of the rows (250 windows = 5 GB each): is a slow allocation slow everywhere or in places?

    python tools/pack_placement.py [n_trials]"""
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
os.environ["PG_PLACE_TRIALS"] = "1"


def pack_ms(e, lo, hi, passes=3):
    e.batch(lo, hi).groupDistTable(True, 100, 0.01)
    e.sync()
    e.kernel_time_reset()
    for _ in range(passes):
        e.batch(lo, hi).groupDistTable(True, 100, 0.01)
    e.sync()
    ms, n = e.kernel_time(_lib.K_PACK)
    return round(ms / n, 3)


def addr(e):
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(L.pg_debug_address(e._h, 0, C.byref(a), C.byref(b)))
    return a.value


e = Engine(0)
e.set_layout(lay)
spacer = Engine(0)
spacer.set_layout(lay)
n_win = len(T.lo)
for trial in range(n_trials):
    check(L.pg_debug_place(e._h, 0, 0))                  # the rows are released ...
    spacer.reserve((trial % 4 + 1) * 3_000_000)           # ... something else takes a piece of what they leave (1.2 GB steps)
    if trial % 4 == 3:
        check(L.pg_debug_place(spacer._h, 0, 0))
    e.reserve(n_sites)
    e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
    rec = {"trial": trial, "rows_at": hex(addr(e))}
    os.environ.pop("PG_PACK_PERM", None)
    rec["in_order_ms"] = [pack_ms(e, T.lo, T.hi), pack_ms(e, T.lo, T.hi)]
    for k in (8, 32, 128):
        os.environ["PG_PACK_PERM"] = str(k)
        rec["perm_%d_ms" % k] = pack_ms(e, T.lo, T.hi)
    os.environ.pop("PG_PACK_PERM", None)
    sl = n_win // 8
    rec["slices_in_order_ms"] = [pack_ms(e, T.lo[i * sl:(i + 1) * sl].copy(), T.hi[i * sl:(i + 1) * sl].copy()) for i in range(8)]
    rec["in_order_again_ms"] = pack_ms(e, T.lo, T.hi)
    print(json.dumps(rec), flush=True)
