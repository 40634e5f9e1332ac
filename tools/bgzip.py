#!/usr/bin/env python
"""A stand-in for htslib's `bgzip` where that is not installed:   python tools/bgzip.py [-l LEVEL] [-@ THREADS] IN [OUT]
writes IN as BGZF (independent gzip members of 65280 bytes of text, `BC` size field, EOF member) to OUT (default IN + ".gz"), the
format `parseVCF.py ... | bgzip > out.geno.gz` produces (VCF_processing/README.md:33) and the drivers inflate on the device.
The members are deflated by the library's host threads (pg_bgzf_compress), 256 MiB of text at a time; --device: by k_deflate on
the GPU (pg_bgzf_compress_device; the text crosses PCIe, the members come back)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomics_general_amd import genoio                                         # noqa: E402


def bgzip_file(src, dst, level=6, threads=0, piece=255 * 65280 * 16, device=False):
    """-> (text bytes, compressed bytes)"""
    n_in = n_out = 0
    eng = None
    if device:
        from genomics_general_amd.engine import Engine
        eng = Engine(int(os.environ.get("PG_DEVICE", "0")))
    with (sys.stdin.buffer if src == "-" else open(src, "rb")) as f, open(dst, "wb") as g:
        while True:
            text = f.read(piece)                                                # (a multiple of the member size: only the last member is short)
            if not text:
                break
            comp = eng.bgzf_compress(text)[0] if eng is not None else genoio.bgzf_compress(text, level, eof_marker=False, n_threads=threads)
            g.write(memoryview(comp))
            n_in += len(text)
            n_out += len(comp)
        g.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return n_in, n_out + 28


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-l", "--level", type=int, default=6)
    ap.add_argument("-@", "--threads", type=int, default=0)
    ap.add_argument("--device", action="store_true", help="deflate on the GPU (k_deflate)")
    ap.add_argument("src")
    ap.add_argument("dst", nargs="?")
    a = ap.parse_args()
    n_in, n_out = bgzip_file(a.src, a.dst or a.src + ".gz", a.level, a.threads, device=a.device)
    sys.stderr.write("%d -> %d bytes (%.1f : 1)\n" % (n_in, n_out, n_in / max(n_out, 1)))
