#!/usr/bin/env python
"""Debug aid: called counts of one window in which exactly one unit is missing at exactly one site, for a sweep of sites; prints
which entries of C deviate from the expected L / L - 1 pattern.  python tools/debug_pairc.py n_dip L unit"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
from genomics_general_amd.engine import Engine  # noqa: E402

n_dip, L, unit = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
names, lay = G.make_layout(n_dip, 1)
H = lay.n_hap
e = Engine(0)
e.set_layout(lay)
e.reserve(L)
for s in list(range(0, 260, 13)) + [L - 1, L - 70, L // 2]:
    codes = np.ones((L, H), dtype=np.int8)
    codes[s, 2 * unit] = 0
    codes[s, 2 * unit + 1] = 0
    e.upload(codes, 0)
    D, C = e.batch([0], [L]).pairCounts(reference_order=False)
    C = C[0][::2, ::2]                     # individuals (both haplotypes of an individual share calledness)
    want = np.full((n_dip, n_dip), L)
    want[unit, :] = L - 1
    want[:, unit] = L - 1
    np.fill_diagonal(want, 0)
    Cc = C.copy(); np.fill_diagonal(Cc, 0)
    bad = np.argwhere(Cc != want)
    print("site %5d: %d wrong entries" % (s, len(bad)), [(int(i), int(j), int(Cc[i, j]) - int(want[i, j])) for i, j in bad[:12]])
e.close()
