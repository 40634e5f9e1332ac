#!/usr/bin/env python
"""pg_tune_planes on the north-star shape: how far do sets of planes lie apart, and does the choice made on EMPTY rows (what
Engine.reserve does) hold once the rows are filled?  One fresh process per line:

  1. the rows reserved as bench.py does (PG_PLACE_TRIALS candidates), the planes left as the reservation's probe allocated them;
  2. `trials` sets of planes tried on the empty rows, the fastest kept              -> empty_rows_ms, kept
  3. the rows filled, the pack kernel's own time (HIP events, 5 passes) on that set -> pack_ms_on_kept
  4. `trials` sets tried again on the filled rows; candidate 0 = the set kept in (2), the others = what a run without
     the choice might have got                                                       -> filled_rows_ms

    python tools/plane_placement.py [trials]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomics_general_amd import _lib, synth, windows                       # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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


def pack_ms(e, passes=5):
    e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    e.kernel_time_reset()
    for _ in range(passes):
        e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    ms, n = e.kernel_time(_lib.K_PACK)
    return round(ms / n, 4)


os.environ["PG_PLANE_TRIALS"] = "1"
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites)
empty = e.tune_planes(n_sites, trials)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)
on_kept = pack_ms(e)
filled = e.tune_planes(n_sites, trials)
on_kept2 = pack_ms(e)
others = filled[0][1:]
print(json.dumps({"rows_probe_ms": e.placement[0], "rows_kept": e.placement[1], "empty_rows_ms": empty[0], "kept": empty[1],
                  "pack_ms_on_kept": on_kept, "filled_rows_ms": filled[0], "kept_on_filled_rows": filled[1],
                  "pack_ms_on_the_second_choice": on_kept2,
                  "first_choice_vs_median_of_the_others_pct": round(100.0 * (filled[0][0] / float(np.median(others)) - 1.0), 2),
                  "best_vs_median_of_the_others_pct": round(100.0 * (min(filled[0]) / float(np.median(others)) - 1.0), 2)}), flush=True)
e.close()
