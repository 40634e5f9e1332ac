#!/usr/bin/env python
"""k_inflate + k_crc32 against zlib on many random BGZF members:   python tools/inflate_fuzz.py [n_members] [seed]
Content families (uniform bytes, few-symbol alphabets, `.geno`-like and VCF-like lines, long runs, periodic patterns of every period
from 1 to 300, sparse edits of a repeated block, empty and one-byte members) x zlib levels 0-9 x strategies (default, filtered,
Huffman only, RLE, fixed) x memLevels; inflated on the device in batches of a few thousand members and compared byte for byte."""
import json
import sys
import zlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomics_general_amd import genoio                                        # noqa: E402
from genomics_general_amd.engine import Engine                                 # noqa: E402


def content(rng, kind, n):
    if n == 0:
        return b""
    if kind == 0:
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    if kind == 1:
        k = int(rng.integers(1, 6))
        return bytes(rng.choice(rng.integers(0, 256, size=k), size=n).astype(np.uint8))
    if kind == 2:                                                              # .geno-like lines
        ncell = int(rng.integers(3, 300))
        cells = rng.choice(list(b"ACGTN"), size=(n // (4 * ncell + 12) + 2, ncell, 2), p=[.3, .3, .19, .19, .02]).astype(np.uint8)
        lines = []
        for i, row in enumerate(cells):
            lines.append(b"chr1\t%d\t" % (1000 + i) + b"\t".join(bytes([a]) + b"/" + bytes([b]) for a, b in row))
        return (b"\n".join(lines) + b"\n")[:n]
    if kind == 3:                                                              # VCF-like cells
        out = []
        while sum(map(len, out)) < n:
            out.append(b"%d/%d:%d,%d:%d:%d\t" % (rng.integers(0, 2), rng.integers(0, 2), rng.integers(0, 40), rng.integers(0, 40), rng.integers(0, 80), rng.integers(0, 99)))
        return b"".join(out)[:n]
    if kind == 4:                                                              # long runs
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 5000))
        return bytes(out[:n])
    if kind == 5:                                                              # periodic, every period
        p = int(rng.integers(1, 301))
        unit = rng.integers(0, 256, size=p, dtype=np.uint8).tobytes()
        return (unit * (n // p + 1))[:n]
    if kind == 6:                                                              # a repeated block with sparse edits (far matches)
        p = int(rng.integers(300, 33000))
        unit = bytearray(rng.integers(65, 91, size=p, dtype=np.uint8).tobytes())
        out = bytearray()
        while len(out) < n:
            for _ in range(int(rng.integers(0, 4))):
                unit[int(rng.integers(0, p))] = int(rng.integers(65, 91))
            out += unit
        return bytes(out[:n])
    return bytes([int(rng.integers(0, 256))]) * n


def main():
    n_members = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    e = Engine(0)
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    done = bad = 0
    text_bytes = 0
    kinds = {}
    while done < n_members:
        batch = min(4000, n_members - done)
        parts, texts = [], []
        for _ in range(batch):
            kind = int(rng.integers(0, 8))
            n = int(rng.choice([0, 1, 2, int(rng.integers(3, 400)), int(rng.integers(400, 20000)), int(rng.integers(20000, 65281)), 65280],
                               p=[.02, .02, .02, .2, .3, .34, .1]))
            t = content(rng, kind, n)
            if kind == 0 and len(t) > 60000:
                t = t[:60000]                                                  # (incompressible bytes must still fit a BGZF member)
            lvl, strat, mem = int(rng.integers(0, 10)), strategies[int(rng.integers(0, 5))], int(rng.integers(1, 10))
            c = zlib.compressobj(lvl, zlib.DEFLATED, -15, mem, strat)
            comp = c.compress(t) + c.flush()
            if len(comp) + 26 > 65536:
                t = t[:30000]
                c = zlib.compressobj(lvl, zlib.DEFLATED, -15, mem, strat)
                comp = c.compress(t) + c.flush()
            parts.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (len(comp) + 25).to_bytes(2, "little") + comp +
                         (zlib.crc32(t) & 0xffffffff).to_bytes(4, "little") + len(t).to_bytes(4, "little"))
            texts.append(t)
            kinds[kind] = kinds.get(kind, 0) + 1
        blob = np.frombuffer(b"".join(parts), dtype=np.uint8)
        tab, used, total = genoio.bgzf_walk(blob, None, 1 << 40)
        want = b"".join(texts)
        assert used == blob.size and total == len(want), (used, blob.size, total, len(want))
        dst = np.full(total + 64, 0xEE, dtype=np.uint8)
        e.inflate_members(blob, tab, dst)
        if dst[:total].tobytes() != want or not (dst[total:] == 0xEE).all():
            bad += 1
            off = 0
            for k, t in enumerate(texts):                                      # name the first member that differs
                if dst[off:off + len(t)].tobytes() != t:
                    print("member %d of batch at %d differs (len %d)" % (k, done, len(t)), flush=True)
                    break
                off += len(t)
        done += batch
        text_bytes += total
    e.close()
    print(json.dumps({"members": done, "text_bytes": text_bytes, "batches_with_differences": bad, "seed": seed, "kinds": kinds}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
