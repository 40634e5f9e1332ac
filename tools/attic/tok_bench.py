#!/usr/bin/env python
"""Where the time of the device tokenizer goes: pg_tokenize_text on one block of synthetic `.geno` text, (a) from a fresh
memory mapping (first touch of the pages), (b) from the same mapping again, (c) from an anonymous copy, next to the host tokenizer
on the same block.  Run under `rocprofv3 --kernel-trace --stats` for the kernel times.   python tools/tok_bench.py [n_sites] [n_dip]"""
import mmap
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from genomics_general_amd import genoio                                       # noqa: E402
from genomics_general_amd.engine import Engine                                # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData                # noqa: E402
from t2_bench import write_fast                                               # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
n_dip = int(sys.argv[2]) if len(sys.argv) > 2 else 100
path = "/tmp/tok_%d_%d.geno" % (n_sites, n_dip)
names = write_fast(path, n_sites, n_dip, 4)
lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites + 1024)
size = os.path.getsize(path)
with open(path, "rb") as f:
    head = len(f.readline())


def timed(tag, body):
    t0 = time.perf_counter()
    got = e.tokenize_text(body, 0, n_sites)
    dt = time.perf_counter() - t0
    assert got is not None and got[0] == n_sites
    print("%-44s %.4f s   %.2f GB/s of text   %.2e sites/s" % (tag, dt, len(body) / dt / 1e9, n_sites / dt), flush=True)
    return got


for rep in range(2):
    f = open(path, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    body = memoryview(mm)[head:]
    timed("fresh mapping, first call", body)
    timed("same mapping, second call", body)
    timed("same mapping, third call", body)
    del body
    mm.close()
    f.close()
f = open(path, "rb")
mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
t0 = time.perf_counter()
mm.madvise(22, 0, size - size % mmap.PAGESIZE)          # MADV_POPULATE_READ
print("madvise(MADV_POPULATE_READ) on a fresh mapping: %.4f s" % (time.perf_counter() - t0))
body = memoryview(mm)[head:]
timed("populated mapping, first call", body)
anon = bytes(body)
got = timed("anonymous copy (bytes)", anon)
timed("anonymous copy again", anon)
t0 = time.perf_counter()
want = genoio.encode(body, lay)
print("host tokenizer (all threads) on the same block: %.4f s" % (time.perf_counter() - t0))
assert np.array_equal(e.download(0, 1000), want.gt[:1000]) and np.array_equal(got[1], want.pos) and got[3] == want.run_names
