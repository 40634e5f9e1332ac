"""`.geno` input for the drop-in drivers: header handling + bulk tokenisation through the C-ABI (K0).

Replaces the line-at-a-time GenoFileReader (genomics.py:1914-1945).  The file format is unchanged:
whitespace separated, first line `#CHROM POS name...` (or `--header` text), `.gz` by suffix, stdin if no file.
"""
import ctypes as C
import gzip
import sys

import numpy as np

from . import _lib
from ._lib import check
from .engine import encode_text


def read_all(path):
    """Whole input as bytes (gunzipped when the name ends in .gz; stdin when path is None)."""
    if path is None:
        return sys.stdin.buffer.read()
    if str(path).endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


class BgzfFile:
    """Read-only file object over a BGZF file (bgzip / htslib: a gzip file made of independent members of at most 64 KiB,
    each announcing its compressed size in a 'BC' extra field).  A plain gzip stream has to be inflated serially, which caps
    `.geno.gz` ingestion at a few hundred MB/s of text; BGZF members are inflated here by a pool of threads (zlib releases
    the GIL).  Offers read(n), readline() and close(), which is all BlockReader needs."""

    CHUNK = 32 << 20                    # compressed bytes fetched per refill

    def __init__(self, path, n_threads=0):
        import os
        from concurrent.futures import ThreadPoolExecutor
        self.raw = open(path, "rb")
        self.pool = ThreadPoolExecutor(max_workers=n_threads or min(16, os.cpu_count() or 1))
        self.pending = b""               # compressed bytes not yet split into whole members
        self.buf = bytearray()           # inflated bytes not yet handed out
        self.eof = False

    @staticmethod
    def is_bgzf(path):
        try:
            with open(path, "rb") as f:
                h = f.read(18)
        except OSError:
            return False
        return (len(h) == 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[10:12] == b"\x06\x00" and h[12:14] == b"BC"
                and h[14:16] == b"\x02\x00")

    @staticmethod
    def _inflate(member):
        import zlib
        return zlib.decompress(member[18:-8], wbits=-15)

    def _refill(self):
        new = self.raw.read(self.CHUNK)
        data = self.pending + new
        if not data:
            self.eof = True
            return
        members, off, n = [], 0, len(data)
        while off + 18 <= n:
            if data[off:off + 4] != b"\x1f\x8b\x08\x04" or data[off + 12:off + 14] != b"BC":
                raise ValueError("input stops being BGZF in the middle (bad member header)")
            size = int.from_bytes(data[off + 16:off + 18], "little") + 1
            if off + size > n:
                break
            members.append(data[off:off + size])
            off += size
        self.pending = data[off:]
        if not new:
            if self.pending:
                raise ValueError("truncated BGZF input")
            self.eof = True
        for part in self.pool.map(self._inflate, members):
            self.buf += part

    def read(self, n=-1):
        if n is None or n < 0:
            while not self.eof:
                self._refill()
            out = bytes(self.buf)
            self.buf = bytearray()
            return out
        while len(self.buf) < n and not self.eof:
            self._refill()
        out = bytes(self.buf[:n])
        del self.buf[:n]
        return out

    def readline(self):
        while True:
            k = self.buf.find(b"\n")
            if k >= 0:
                out = bytes(self.buf[:k + 1])
                del self.buf[:k + 1]
                return out
            if self.eof:
                out = bytes(self.buf)
                self.buf = bytearray()
                return out
            self._refill()

    def close(self):
        self.raw.close()
        self.pool.shutdown(wait=False)


class BlockReader:
    """The input as a sequence of byte blocks that end at line boundaries (gunzipped when the name ends in .gz -- in parallel
    when the file is BGZF, i.e. written by bgzip; stdin when path is None).  read_block(None) returns everything that is
    left."""

    def __init__(self, path):
        if path is None:
            self.f = sys.stdin.buffer
        elif str(path).endswith(".gz"):
            self.f = BgzfFile(path) if BgzfFile.is_bgzf(path) else gzip.open(path, "rb")
        else:
            self.f = open(path, "rb")
        self.bytes_read = 0

    def read_header(self):
        line = self.f.readline()
        self.bytes_read += len(line)
        return line

    def read_block(self, nbytes=None):
        if nbytes is None:
            data = self.f.read()
        else:
            data = self.f.read(nbytes)
            if data and not data.endswith(b"\n"):
                data += self.f.readline()
        self.bytes_read += len(data)
        return data

    def close(self):
        if self.f is not sys.stdin.buffer:
            self.f.close()


def read_header_names(path):
    """Sample names of the file's first line (popgenWindows.py:284-286, distMat.py:205-206)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as f:
        return f.readline().split()[2:]


def split_header(data, header_line=None):
    """(sample names, data bytes after the header).  With --header the file has no header line
    (GenoFileReader.__init__, genomics.py:1917-1919)."""
    if header_line:
        return header_line.split()[2:], data
    nl = data.find(b"\n")
    first = data if nl < 0 else data[:nl]
    rest = b"" if nl < 0 else data[nl + 1:]
    return first.decode("utf-8", "replace").split()[2:], rest


class GenoData:
    """Encoded input: one-hot int8 genotypes in device slot order, positions and scaffold runs."""

    def __init__(self, gt, pos, run_starts, run_names):
        self.gt, self.pos, self.run_starts, self.run_names = gt, pos, run_starts, run_names
        self.n_sites = len(pos)


def concat(a, b):
    """Rows of GenoData a followed by those of b (a run that continues across the seam is merged).  When b was encoded
    with head_rows = a.n_sites the rows of a are copied into the spare rows in front of b's arrays (no copy of b)."""
    if a is None or a.n_sites == 0:
        return b
    if b.n_sites == 0:
        return a
    merge = a.run_names[-1] == b.run_names[0]
    starts = np.concatenate([a.run_starts, (b.run_starts[1:] if merge else b.run_starts) + a.n_sites]).astype(np.int64)
    names = list(a.run_names) + list(b.run_names[1:] if merge else b.run_names)
    n = a.n_sites
    gbase, pbase = b.gt.base, b.pos.base
    if (gbase is not None and pbase is not None and gbase.ndim == 2 and gbase.shape[0] >= n + b.n_sites
            and b.gt.ctypes.data == gbase.ctypes.data + n * gbase.shape[1] and b.pos.ctypes.data == pbase.ctypes.data + 4 * n):
        gbase[:n] = a.gt
        pbase[:n] = a.pos
        return GenoData(gbase[:n + b.n_sites], pbase[:n + b.n_sites], starts, names)
    return GenoData(np.concatenate([a.gt, b.gt]), np.concatenate([a.pos, b.pos]), starts, names)


def tail(d, keep_from):
    """Rows [keep_from, n) of GenoData d."""
    if keep_from >= d.n_sites:
        return None
    r = int(np.searchsorted(d.run_starts, keep_from, side="right")) - 1
    starts = np.concatenate([[0], d.run_starts[r + 1:] - keep_from]).astype(np.int64)
    return GenoData(d.gt[keep_from:].copy(), d.pos[keep_from:].copy(), starts, list(d.run_names[r:]))


def encode(data, layout, n_threads=0, head_rows=0):
    gt, pos, soff, slen = encode_text(data, layout, n_threads, head_rows)
    n = len(pos)
    L = _lib.lib()
    cap = 1024
    while True:
        starts = np.zeros(cap, dtype=np.int64)
        nr = C.c_int64(0)
        rc = L.pg_scaffold_runs(data, soff, slen, n, starts, cap, C.byref(nr))
        if nr.value > cap:
            cap = int(nr.value)
            continue
        check(rc)
        break
    starts = starts[:nr.value]
    names = [data[int(soff[i]):int(soff[i]) + int(slen[i])].decode("utf-8", "replace") for i in starts]
    return GenoData(gt, pos, starts, names)
