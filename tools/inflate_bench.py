#!/usr/bin/env python
"""The inflate kernels alone:   python tools/inflate_bench.py [--sites N] [--dip D] [--level L] [--libs a.so,b.so]
1 GiB of north-star `.geno` text (N sites x D diploids, written from device-resident rows), bgzipped at level L, inflated by
pg_inflate_device: device milliseconds of k_inflate (+ k_crc32), GB/s of text.  --libs: the same with other builds of the library
(PG_LIBRARY; A/B of kernel variants), each in a process of its own."""
import argparse
import ctypes as C
import json
import os

# the benchmark's bgzipped samples are what htslib's bgzip writes (zlib, level 6) -- the reference's default input --, not what this
# library's own, faster compressor would write (csrc/pg_fast_deflate.h: shorter matches, i.e. more symbols for k_inflate to decode)
os.environ.setdefault("PG_BGZF_ZLIB", "1")
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_one(path):
    from genomics_general_amd import _lib, genoio
    from genomics_general_amd.engine import Engine
    with open(path, "rb") as f:
        data = f.read()
    tab, used, text = genoio.bgzf_walk(data, None, 1 << 30)
    in_off, in_len, out_len, crc = tab
    arr = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(text, dtype=np.uint8)
    e = Engine(0)
    vp = lambda a: C.c_void_p(a.ctypes.data)                                    # noqa: E731
    res = {}
    for crc_on in (True, False):
        best = []
        for rep in range(5):
            ms = C.c_double(0)
            _lib.check(_lib.lib().pg_inflate_device(e._h, vp(arr), used, vp(in_off), vp(in_len), vp(out_len), vp(crc) if crc_on else None,
                                                    len(in_off), vp(out), C.byref(ms)))
            best.append(ms.value)
        res["inflate+crc32" if crc_on else "inflate"] = {"ms": round(min(best[1:]), 3), "text_GBps": round(text / min(best[1:]) / 1e6, 1)}
    res.update(members=int(len(in_off)), compressed_MB=round(used / 1e6, 1), text_MB=round(text / 1e6, 1), lib=os.path.basename(_lib.LIB_PATH))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=1_320_000)
    ap.add_argument("--dip", type=int, default=200)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--libs", default="")
    ap.add_argument("--file")
    a = ap.parse_args()
    if a.file:
        run_one(a.file)
        sys.exit(0)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bgzip
    tmp = tempfile.mkdtemp(prefix="pg_inflate_")
    geno = os.path.join(tmp, "sample.geno")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "t2_write_sample.py"), geno, str(a.sites), str(a.dip)], stdout=subprocess.DEVNULL, check=True)
    n_in, n_out = bgzip.bgzip_file(geno, geno + ".gz", a.level)
    print("sample: %.2f GB of text -> %.3f GB (%.1f : 1, level %d)" % (n_in / 1e9, n_out / 1e9, n_in / n_out, a.level), flush=True)
    for lib in [None] + [x for x in a.libs.split(",") if x]:
        env = dict(os.environ)
        if lib:
            env["PG_LIBRARY"] = os.path.abspath(lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--file", geno + ".gz"], env=env, check=False)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
