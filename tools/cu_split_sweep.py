#!/usr/bin/env python
"""VERDICT round 5 #4a: the matrix-core kernels of sub-batch k-1 beside the pack kernel of sub-batch k on DISJOINT compute units
(streams created with hipExtStreamCreateWithCUMask, pg_debug_cu_split).  One process, one allocation of the north-star rows (the
pack kernel's time moves with the physical placement, so configurations are only comparable inside one allocation): for every
(sub-batches, CUs per XCD given to the pair stream) the time of a whole pass, the kernel families' event times, and whether the
result table is bit-identical to the one-stream pass.

    python tools/cu_split_sweep.py [n_sites] [passes]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomics_general_amd import _lib, synth, windows                       # noqa: E402
from genomics_general_amd._lib import check                                  # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_dip, n_pops, n_scaf, wind = 200, 4, 4, 50_000
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
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
FAM = {"pack": _lib.K_PACK, "pairC": _lib.K_PAIRWISE, "pairD": _lib.K_PAIRD}


def run(label):
    tab = e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    e.kernel_time_reset()
    t0 = time.perf_counter()
    for _ in range(passes):
        e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    ms = (time.perf_counter() - t0) * 1e3 / passes
    fam = {}
    for k, v in FAM.items():
        t, n = e.kernel_time(v)
        fam[k] = round(t / passes, 3)
    return ms, fam, tab


base_ms, fam, base = run("one stream")
base_bytes = base[0].tobytes()
print(json.dumps({"config": "one stream, one batch", "ms_per_pass": round(base_ms, 3), "event_ms_per_pass": fam}), flush=True)
for n_sub in (8, 16, 32):
    os.environ["PG_OVERLAP"] = str(n_sub)
    for cus in (0, 2, 4, 6, 8, 10, 12, 16):
        check(L.pg_debug_cu_split(e._h, cus))
        ms, fam, tab = run("")
        tb = tab[0].tobytes()
        print(json.dumps({"config": "two streams, %d sub-batches, pair stream on %s" % (n_sub, "%d CUs per XCD (%d), pack stream on the other %d" % (cus, 8 * cus, 256 - 8 * cus) if cus else "all CUs (no masks)"),
                          "ms_per_pass": round(ms, 3), "vs_one_stream": round(ms / base_ms, 4), "event_ms_per_pass_overlapping": fam,
                          "table_bit_identical": tb == base_bytes}), flush=True)
    check(L.pg_debug_cu_split(e._h, 0))
del os.environ["PG_OVERLAP"]
ms, fam, _ = run("")
print(json.dumps({"config": "one stream again", "ms_per_pass": round(ms, 3), "event_ms_per_pass": fam}), flush=True)
