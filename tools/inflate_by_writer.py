#!/usr/bin/env python
"""k_inflate on the same `.geno` text as three writers leave it: zlib at level 6 (what htslib's bgzip writes), the library's host
compressor (pg_fast_deflate.h), and k_deflate -- the file `parseVCF.py -o out.geno.gz` writes is the window drivers' input.
    python tools/inflate_by_writer.py [n_sites] [n_dip]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from genomics_general_amd import genoio
    from genomics_general_amd.engine import Engine
    n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 1200000
    n_dip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    tmp = tempfile.mkdtemp(prefix="pg_ibw_", dir=os.environ.get("PG_BENCH_TMP", "/tmp"))
    geno = os.path.join(tmp, "s.geno")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "t2_write_sample.py"), geno, str(n_sites), str(n_dip)], stdout=subprocess.PIPE, check=True)
    with open(geno, "rb") as f:
        text = f.read()
    os.remove(geno)
    os.rmdir(tmp)
    eng = Engine(0)
    res = {"text_bytes": len(text), "writers": {}}
    os.environ["PG_BGZF_ZLIB"] = "1"
    by = {"zlib level 6 (htslib's bgzip)": genoio.bgzf_compress(text, 6, 65280, eof_marker=False).tobytes()}
    # (pg_bgzf_compress reads PG_BGZF_ZLIB once per process: the host compressor's members come from a child)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from genomics_general_amd import genoio; import os; "
                        "sys.stdout.buffer.write(genoio.bgzf_compress(sys.stdin.buffer.read(), 6, 65280, eof_marker=False).tobytes())" % ROOT],
                       input=text, stdout=subprocess.PIPE, env={k: v for k, v in os.environ.items() if k != "PG_BGZF_ZLIB"}, check=True)
    by["the library's host compressor"] = r.stdout
    by["k_deflate"] = eng.bgzf_compress(text)[0].tobytes()
    dst = eng.pinned.empty((len(text) + 64,), np.uint8)
    for name, comp in by.items():
        tab, used, n_text = genoio.bgzf_walk(comp, None, 1 << 40)
        assert n_text == len(text)
        best = None
        for _ in range(4):
            ms = eng.inflate_members(comp, tab, dst)
            best = ms if best is None else min(best, ms)
        assert dst[:len(text)].tobytes() == text
        res["writers"][name] = {"bytes": len(comp), "ratio": round(len(text) / len(comp), 2), "k_inflate_ms": round(best, 3),
                                "k_inflate_ms_per_GiB_of_text": round(best / (len(text) / 2**30), 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
