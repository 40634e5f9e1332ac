#!/usr/bin/env python
"""T2 on the reference's default input format:   python tools/t2_bgzf_bench.py [n_sites] [n_dip]
writes the head of the north-star data set as `.geno` text, bgzips it (tools/bgzip.py), and runs popgenWindows.py on the text and on
the `.geno.gz` (PG_TIMING lines, CSVs compared); then the inflate kernels alone on 1 GiB of that text (pg_inflate_device)."""
import ctypes as C
import json
import os

# the benchmark's bgzipped samples are what htslib's bgzip writes (zlib, level 6) -- the reference's default input --, not what this
# library's own, faster compressor would write (csrc/pg_fast_deflate.h: shorter matches, i.e. more symbols for k_inflate to decode)
os.environ.setdefault("PG_BGZF_ZLIB", "1")
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from genomics_general_amd import _lib, genoio                                   # noqa: E402
import bgzip                                                                    # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
n_dip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tmp = tempfile.mkdtemp(prefix="pg_bgzf_")
geno = os.path.join(tmp, "sample.geno")
cmd = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "t2_write_sample.py"), geno, str(n_sites), str(n_dip)],
                     stdout=subprocess.PIPE, check=True).stdout.decode().split()
t0 = time.perf_counter()
n_in, n_out = bgzip.bgzip_file(geno, geno + ".gz")
print("bgzip: %.2f GB -> %.3f GB (%.1f : 1) in %.1f s" % (n_in / 1e9, n_out / 1e9, n_in / n_out, time.perf_counter() - t0), flush=True)
res = {}
runs = [("text", geno, {}), ("bgzf_device", geno + ".gz", {})]
if os.environ.get("T2_HOST_POOL", "1") != "0":
    runs.append(("bgzf_host_pool", geno + ".gz", {"PG_BGZF_DEVICE": "0"}))
# what a rank gets at N = 8 under the GPU boxes' 16-CPU quota: two host threads (four at N = 4)
for nt in (os.environ.get("T2_HOST_THREADS") or "").split(","):
    if nt:
        runs += [("text_%s_host_threads" % nt, geno, {"PG_HOST_THREADS": nt}), ("bgzf_device_%s_host_threads" % nt, geno + ".gz", {"PG_HOST_THREADS": nt})]
summary = {}
for label, path, env in runs:
    c = [path if x == geno else (path + ".csv") if x == geno + ".csv" else x for x in cmd]
    for rep in range(2):
        r = subprocess.run(c, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1", **env), stderr=subprocess.PIPE, stdout=subprocess.PIPE)
        line = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
        if not line:
            print(label, "FAILED", r.stderr.decode()[-2000:])
            break
        for ln in r.stderr.decode().splitlines():
            if ln.startswith("PG_TOK_TRACE"):
                print("   ", ln)
            if ln.startswith("PG_TIMELINE ") and rep == 1:
                for th, lab, a, b in json.loads(ln[len("PG_TIMELINE "):]):
                    print("    TL %-12s %-11s %8.1f -> %8.1f ms  (%6.1f)" % (th, lab, a * 1e3, b * 1e3, (b - a) * 1e3))
        tm = json.loads(line[-1][len("PG_TIMING "):])
        res[label] = tm
        print(label, rep, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items()}), flush=True)
        print("   -> %.2f GB/s of text, %.2f without the context" % (n_in / tm["total_s"] / 1e9, n_in / (tm["total_s"] - tm.get("context_s", 0)) / 1e9))
        summary.setdefault(label, []).append({"total_s": round(tm["total_s"], 4), "context_s": round(tm.get("context_s", 0), 4),
                                              "text_GBps": round(n_in / tm["total_s"] / 1e9, 2),
                                              "text_GBps_without_context": round(n_in / (tm["total_s"] - tm.get("context_s", 0)) / 1e9, 2),
                                              "tokenize_s": round(tm.get("tokenize_s", 0), 4), "read_s": round(tm.get("read_s", 0), 4),
                                              "compute_and_write_s": round(tm.get("compute_and_write_s", 0), 4)})
print("SUMMARY " + json.dumps({"sites": n_sites, "diploids": n_dip, "text_bytes": n_in, "bgzf_bytes": n_out, "usable_cpus": _lib.usable_cpus(), "runs": summary}))
csvs = [open(p + ".csv").read() for p in (geno, geno + ".gz")]
print("csv equal:", csvs[0] == csvs[1], len(csvs[0]))
# the kernels alone
from genomics_general_amd.engine import Engine                                  # noqa: E402
e = Engine(0)
with open(geno + ".gz", "rb") as f:
    data = f.read(160 << 20)
tab, used, text = genoio.bgzf_walk(data, None, 1 << 30)
in_off, in_len, out_len, crc = tab
arr = np.frombuffer(data, dtype=np.uint8)
out = np.empty(text, dtype=np.uint8)
vp = lambda a: C.c_void_p(a.ctypes.data)                                        # noqa: E731
for crc_on in (True, False):
    for rep in range(3):
        ms = C.c_double(0)
        _lib.check(_lib.lib().pg_inflate_device(e._h, vp(arr), used, vp(in_off), vp(in_len), vp(out_len), vp(crc) if crc_on else None, len(in_off),
                                                vp(out), C.byref(ms)))
        print("k_inflate%s: %d members, %.1f MB -> %.1f MB of text in %.3f ms = %.1f GB/s of text" % (
            " + k_crc32" if crc_on else "", len(in_off), used / 1e6, text / 1e6, ms.value, text / ms.value / 1e6), flush=True)
with open(geno, "rb") as f:
    assert f.read(text) == out.tobytes()
print("inflated text == file")
import shutil                                                                   # noqa: E402
shutil.rmtree(tmp, ignore_errors=True)
