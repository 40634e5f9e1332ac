#!/usr/bin/env python
"""Where a C4 (distMat, 10^6 sites x 1000 diploids, 100 kb windows) step spends its time: every kernel family and the copy of
the 40 MB result table bracketed by HIP events in the steady state, wall time per step next to their sum (the rest is host time).
    python tools/c4_host_time.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomics_general_amd import _lib, synth  # noqa: E402
from genomics_general_amd.engine import Engine  # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_dip, n_sites, wind = 1000, 1_000_000, 100_000
names = ["s%d" % d for d in range(n_dip)]
lay = HapLayout(SampleData(popNames=["pop0"], popInds=[names]), names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, n_sites, n_dip, 1, slot_gen, synth.VAR_THR, synth.MISS_THR)
lo = np.arange(0, n_sites, wind, dtype=np.int64)
hi = lo + wind
tab = None
for _ in range(4):
    tab = e.batch(lo, hi).indPairTable()
e.sync()
e.kernel_time_reset()
phases = {"batch()": 0.0, "indPairTable()": 0.0}
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    wb = e.batch(lo, hi)
    b = time.perf_counter()
    tab = wb.indPairTable()
    c = time.perf_counter()
    phases["batch()"] += b - a
    phases["indPairTable()"] += c - b
e.sync()
wall = (time.perf_counter() - t0) / steps * 1e3
fam = {name: round(e.kernel_time(k)[0] / steps, 4) for k, name in _lib.KERNEL_NAMES.items() if e.kernel_time(k)[1]}
print(json.dumps({"wall_ms_per_step": round(wall, 4), "event_brackets_ms_per_step": fam, "sum_of_brackets_ms": round(sum(fam.values()), 4),
                  "host_and_gaps_ms": round(wall - sum(fam.values()), 4),
                  "python_phases_ms_per_step": {k: round(v / steps * 1e3, 4) for k, v in phases.items()},
                  "table_MB": round(tab.nbytes / 1e6, 1)}))
e.close()
