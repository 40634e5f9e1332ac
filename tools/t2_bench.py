#!/usr/bin/env python
"""Tier T2 (text end to end): write a synthetic `.geno` file, run the drop-in popgenWindows.py on it, print the per-phase wall
times (PG_TIMING).  python tools/t2_bench.py [n_sites] [n_dip]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomics_general_amd import synth  # noqa: E402


def write_fast(path, n_sites, n_dip, n_pops, seed=7):
    """synthetic `.geno` text through the vectorised writer, 100 000 sites at a time"""
    names = ["s%d" % d for d in range(n_dip)]
    step = 100000
    with open(path, "wb") as f:
        for a in range(0, n_sites, step):
            b = min(n_sites, a + step)
            codes = synth.gen_codes(seed, np.zeros(b - a, dtype=np.int64), np.arange(a + 1, b + 1), n_dip, n_pops)
            tmp = path + ".part"
            synth.write_geno_fast(tmp, codes, names, "chr1", a + 1)
            with open(tmp, "rb") as g:
                if a > 0:
                    g.readline()
                f.write(g.read())
            os.remove(tmp)
    return names


def main():
    n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_dip = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    path = "/tmp/t2_%d_%d.geno" % (n_sites, n_dip)
    t0 = time.time()
    names = write_fast(path, n_sites, n_dip, 4)
    print("wrote %s: %.1f MB in %.1f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0))
    per = n_dip // 4
    cmd = [sys.executable, os.path.join(ROOT, "popgenWindows.py"), "-g", path, "-o", "/tmp/t2_out.csv", "-f", "phased", "-w", "50000",
           "-m", "100"]
    for k in range(4):
        cmd += ["-p", "pop%d" % k, ",".join(names[k * per:(k + 1) * per])]
    ref_rows = None
    for rep, block in enumerate([None, None, 64 << 20]):       # twice with the default block size, once in 64 MiB blocks
        env = dict(os.environ, PG_TIMING="1", PG_GPU_TOKENIZER="0")             # K0 on the host (reader || tokenizer || upload)
        if block:
            env["PG_STREAM_BYTES"] = str(block)
        t0 = time.time()
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
        wall = time.time() - t0
        line = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING")]
        print("host tokenizer run %d (%s): wall %.2f s (incl. interpreter start)  %s" % (
            rep, "blocks of %d MiB" % (block >> 20) if block else "default blocks", wall,
            line[-1] if line else r.stderr.decode()[-400:]))
        with open("/tmp/t2_out.csv") as f:
            rows = f.readlines()
        if ref_rows is None:
            ref_rows = rows
        print("   output identical to run 0:", rows == ref_rows)
    # K0 on the device (the default): the text goes down as it is, pg_tokenize_text writes the resident rows
    for rep, block in enumerate([None, None, 256 << 20]):
        env = dict(os.environ, PG_TIMING="1")
        if block:
            env["PG_STREAM_BYTES"] = str(block)
        t0 = time.time()
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
        dwall = time.time() - t0
        line = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING")]
        print("device tokenizer run %d (%s): wall %.2f s  %s" % (rep, "blocks of %d MiB" % (block >> 20) if block else "default blocks",
                                                                 dwall, line[-1] if line else r.stderr.decode()[-400:]))
        with open("/tmp/t2_out.csv") as f:
            print("   output identical to run 0:", f.readlines() == ref_rows)
    print("device tokenizer: windows/s end to end:", round((len(ref_rows) - 1) / dwall, 2), "| sites/s:", round(n_sites / dwall))
    # the same job from a packed .pgeno file (tokenised once by tools/geno_pack.py)
    packed = path[:-5] + ".pgeno"
    t0 = time.time()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "geno_pack.py"), "-g", path, "-o", packed, "-f", "phased"], check=True)
    print("packed %s: %.1f MB in %.1f s" % (packed, os.path.getsize(packed) / 1e6, time.time() - t0))
    pcmd = [packed if c == path else c for c in cmd]
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run(pcmd, env=dict(os.environ, PG_TIMING="1"), stderr=subprocess.PIPE)
        pwall = time.time() - t0
        line = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING")]
        print("packed run %d: wall %.2f s  %s" % (rep, pwall, line[-1] if line else r.stderr.decode()[-400:]))
        with open("/tmp/t2_out.csv") as f:
            print("   output identical to run 0:", f.readlines() == ref_rows)
    print("packed input: windows/s end to end:", round((len(rows) - 1) / pwall, 2), "| sites/s:", round(n_sites / pwall))
    print("rows:", len(rows) - 1, "| windows/s end to end:", round((len(rows) - 1) / wall, 2), "| sites/s:", round(n_sites / wall))


if __name__ == "__main__":
    main()
