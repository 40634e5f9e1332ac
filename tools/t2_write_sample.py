#!/usr/bin/env python
"""Write the head of the north-star data set as `.geno` text (and optionally `.pgeno`) from device-resident rows, for profile runs of
the drop-in drivers:   python tools/t2_write_sample.py OUT.geno [n_sites] [n_dip]   -> prints the popgenWindows.py command line"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                   # noqa: E402
from genomics_general_amd import synth                                         # noqa: E402
from genomics_general_amd.engine import Engine                                 # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData                 # noqa: E402

out = sys.argv[1]
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
n_dip = int(sys.argv[3]) if len(sys.argv) > 3 else 200
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // 4
sd = SampleData(popNames=["pop%d" % k for k in range(4)], popInds=[names[k * per:(k + 1) * per] for k in range(4)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, n_sites, n_dip, 4, slot_gen, synth.VAR_THR, synth.MISS_THR)
size = bench.write_geno_resident(out, e, lay, names, n_sites)
e.close()
cmd = ["python", os.path.join(ROOT, "popgenWindows.py"), "-g", out, "-o", out + ".csv", "-f", "phased", "-w", "50000", "-m", "100"]
for k in range(4):
    cmd += ["-p", "pop%d" % k, ",".join(names[k * per:(k + 1) * per])]
sys.stderr.write("%s: %.2f GB\n" % (out, size / 1e9))
print(" ".join(cmd))
