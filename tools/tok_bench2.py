#!/usr/bin/env python
"""Where the time of the device tokenizer goes (round 4: self-paced staging threads, pread from the file): one block of synthetic
`.geno` text tokenised (a) from the file (pg_tokenize_file), (b) from a memory mapping, (c) from an anonymous copy; wall time per
call, and inside it the copies (PCIe) and the kernels (pg_tokenize_stats).   python tools/tok_bench2.py [n_sites] [n_dip]"""
import mmap
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from genomics_general_amd.engine import Engine                                # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData                # noqa: E402
from t2_bench import write_fast                                               # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
n_dip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
path = "/tmp/tok_%d_%d.geno" % (n_sites, n_dip)
names = write_fast(path, n_sites, n_dip, 4)
lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
e = Engine(0)
e.set_layout(lay)
e.reserve(n_sites + 1024)
size = os.path.getsize(path)
f = open(path, "rb")
head = len(f.readline())
mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
body = memoryview(mm)[head:]
bound = len(body) // (4 * n_dip + 4) + 1


def timed(tag, **kw):
    s0 = e.tokenize_stats()
    t0 = time.perf_counter()
    got = e.tokenize_text(kw.pop("buf", body), 0, bound, at_most=True, **kw)
    dt = time.perf_counter() - t0
    s1 = e.tokenize_stats()
    assert got is not None and got[0] == n_sites
    print("%-40s %.4f s  %6.2f GB/s of text | copies %.4f s = %6.2f GB/s, kernels + results %.4f s" % (
        tag, dt, len(body) / dt / 1e9, s1["h2d_s"] - s0["h2d_s"], len(body) / (s1["h2d_s"] - s0["h2d_s"]) / 1e9,
        s1["kernels_s"] - s0["kernels_s"]), flush=True)


for rep in range(3):
    timed("file (pread), call %d" % rep, file=(f.fileno(), head))
for rep in range(2):
    timed("mapping (memcpy), call %d" % rep)
anon = bytes(body)
for rep in range(2):
    timed("anonymous copy, call %d" % rep, buf=anon)
for nt in (4, 8, 32):
    os.environ["PG_HOST_THREADS"] = str(nt)
    timed("file (pread), PG_HOST_THREADS=%d" % nt, file=(f.fileno(), head))
os.remove(path)
