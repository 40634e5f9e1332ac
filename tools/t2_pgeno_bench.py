#!/usr/bin/env python
"""Tier T2 from packed input: the head of the north-star data set written as `.pgeno` (codec none = 1 byte per genotype, and zlib)
straight from the device-resident rows, then the drop-in popgenWindows.py on each file, per-phase times (PG_TIMING).
    python tools/t2_pgeno_bench.py [n_sites] [n_dip]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomics_general_amd import genoio, synth                                 # noqa: E402
from genomics_general_amd.engine import Engine                                 # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData                 # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
n_dip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // 4
sd = SampleData(popNames=["pop%d" % k for k in range(4)], popInds=[names[k * per:(k + 1) * per] for k in range(4)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites)
e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, n_sites, n_dip, 4, slot_gen, synth.VAR_THR, synth.MISS_THR)
s0 = np.array([lay.ind_slots[nm][0] for nm in names])
for codec in (("none",) if os.environ.get("PG_BENCH_PGENO_ONLY_NONE") else ("none", "zlib")):
    path = "/tmp/t2_%d_%d_%s.pgeno" % (n_sites, n_dip, codec)
    t0 = time.time()
    wr = genoio.PackedWriter(path, names, [2] * n_dip, codec)
    step = 1_000_000
    for a in range(0, n_sites, step):
        b = min(n_sites, a + step)
        rows = e.download(a, b - a).view(np.uint8)
        cells = rows[:, s0] | (rows[:, s0 + 1] << 4)
        wr.write_block(genoio.GenoData(None, np.arange(a + 1, b + 1, dtype=np.int32), np.zeros(1, dtype=np.int64), ["chr1"]), cells)
    wr.close()
    print("wrote %s: %.1f MB in %.1f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0), flush=True)
    cmd = [sys.executable, os.path.join(ROOT, "popgenWindows.py"), "-g", path, "-o", "/tmp/t2_pgeno_out.csv", "-f", "phased", "-w", "50000", "-m", "100"]
    for k in range(4):
        cmd += ["-p", "pop%d" % k, ",".join(names[k * per:(k + 1) * per])]
    for rep in range(2):
        r = subprocess.run(cmd, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1"), stderr=subprocess.PIPE)
        line = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING")]
        if not line:
            print(r.stderr.decode()[-600:])
            continue
        tm = json.loads(line[-1][len("PG_TIMING "):])
        print("%s run %d: total %.3f s (context %.3f) = %.2e sites/s | read %.3f tokenize(inflate) %.3f upload %.3f prep_wait %.3f compute+write %.3f" % (
            codec, rep, tm["total_s"], tm.get("context_s", 0), n_sites / tm["total_s"], tm["read_s"], tm["tokenize_s"], tm["upload_s"],
            tm["prep_wait_s"], tm["compute_and_write_s"]), "first chunk %.3f others %.3f" % (tm.get("compute_first_chunk_s", 0), tm.get("compute_other_chunks_s", 0)), flush=True)
    os.remove(path)
