#!/usr/bin/env python
"""Same-process A/B of an environment switch the library reads per call: one resident data set (north-star shape, or
10^7 sites x 100 diploids with `c2`), the popgenWindows pi / dxy / Fst pass alternately without and with the switch, kernel times
per pass.  Placement of the rows (which moves the pack kernel by +-5 % between processes) is the same for both.
    python tools/ab_env.py NAME[=VALUE] [northstar|c2] [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomics_general_amd import _lib, synth                                # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

name, _, value = sys.argv[1].partition("=")
shape = sys.argv[2] if len(sys.argv) > 2 else "northstar"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n_sites, n_dip = (100_000_000, 200) if shape == "northstar" else (10_000_000, 100)
n_pops, wind = 4, 50_000
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // n_pops
sd = SampleData(popNames=["pop%d" % k for k in range(n_pops)], popInds=[names[k * per:(k + 1) * per] for k in range(n_pops)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
os.environ.setdefault("PG_PLACE_TRIALS", "1")
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, n_sites // 4, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
lo = np.arange(0, n_sites, wind, dtype=np.int64)
hi = lo + wind
ref = None
for r in range(rounds):
    for on in (False, True):
        if on:
            os.environ[name] = value or "1"
        else:
            os.environ.pop(name, None)
        e.batch(lo, hi).groupDistTable(True, 100, 0.01)                       # (buffers of the other mode may be reallocated)
        e.sync()
        e.kernel_time_reset()
        for _ in range(3):
            tab, cols = e.batch(lo, hi).groupDistTable(True, 100, 0.01)
        e.sync()
        kt = {nm: e.kernel_time(kid) for kid, nm in _lib.KERNEL_NAMES.items()}
        print("%-24s %s" % ((sys.argv[1] if on else "(unset)"), {k: round(v[0] / 3, 3) for k, v in kt.items() if v[1]}), flush=True)
        ref = tab if ref is None else ref
        assert os.environ.get("AB_NOCHECK") or np.array_equal(tab, ref, equal_nan=True), "the switch changes the statistics"
