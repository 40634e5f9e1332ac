#!/usr/bin/env python
"""Every drop-in driver end to end on one bgzipped `.geno.gz` (SURVEY 8 rows a, f1, f2: popgenWindows.py, ABBABABAwindows.py,
fourPopWindows.py, distMat.py, freq.py), timed inside the driver (PG_TIMING total_s: from opening the input to the last row):

    python tools/drivers_bench.py [n_sites] [n_dip]                 # on a GPU box: writes the sample from device-resident rows
    python tools/drivers_bench.py --reference [n_sites] [n_dip]     # where /root/reference is: the UNMODIFIED scripts, with as many
                                                                    # workers as the host has CPUs, on a small host-generated sample

One JSON line.  The reference's scripts are per-window Python loops: their sites/s do not depend on the file's length, so the two
modes together give the ratio (different hosts: stated in the output)."""
import gzip
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
REF = "/root/reference"


def driver_commands(geno, out, names, n_pops, wind):
    per = len(names) // n_pops
    pops = []
    for k in range(n_pops):
        pops += ["-p", "pop%d" % k, ",".join(names[k * per:(k + 1) * per])]
    p4 = ["-P1", "pop0", ",".join(names[0:per]), "-P2", "pop1", ",".join(names[per:2 * per]), "-P3", "pop2", ",".join(names[2 * per:3 * per]),
          "-O", "pop3", ",".join(names[3 * per:4 * per])]
    return {
        "popgenWindows.py": ["-g", geno, "-o", out + ".popgen.csv", "-f", "phased", "-w", str(wind), "-m", "100"] + pops,
        "popgenWindows.py --analysis popFreq": ["-g", geno, "-o", out + ".popfreq.csv", "-f", "phased", "-w", str(wind), "-m", "100",
                                                "--analysis", "popFreq"] + pops,
        "popgenWindows.py --analysis indHet": ["-g", geno, "-o", out + ".indhet.csv", "-f", "phased", "-w", str(wind), "-m", "100",
                                               "--analysis", "indHet"] + pops,
        "popgenWindows.py --analysis hapStats": ["-g", geno, "-o", out + ".hapstats.csv", "-f", "phased", "-w", str(wind), "-m", "100",
                                                 "--analysis", "hapStats"] + pops,
        "popgenWindows.py --analysis indPairDist": ["-g", geno, "-o", out + ".indpair.csv", "-f", "phased", "-w", str(wind), "-m", "100",
                                                    "--analysis", "indPairDist"] + pops,
        "ABBABABAwindows.py": ["-g", geno, "-o", out + ".abba.csv", "-f", "phased", "-w", str(wind), "-m", "100", "--minData", "0.5"] + p4,
        "fourPopWindows.py": ["-g", geno, "-o", out + ".fourpop.csv", "-f", "phased", "-w", str(wind), "-m", "100", "--minData", "0.5",
                              "--polarize"] + p4,
        "distMat.py": ["-g", geno, "-f", "phased", "--windType", "coordinate", "-w", str(wind), "-o", out + ".dist", "--outFormat", "raw"],
        "freq.py": ["-g", geno, "-o", out + ".freq.tsv", "-f", "phased", "--target", "derived"] + pops,
        # "If you add `.gz` it will be gzipped" (the reference's README): BGZF by the library's host threads here ...
        "freq.py -o out.tsv.gz": ["-g", geno, "-o", out + ".freq.tsv.gz", "-f", "phased", "--target", "derived"] + pops,
        # ... against what the reference does, gzip.open(path, "wt"): the gzip module at level 9 on the calling thread
        "freq.py -o out.tsv.gz (PG_OUT_GZIP_MODULE=1: the gzip module, as the reference writes it)":
            ["-g", geno, "-o", out + ".freq2.tsv.gz", "-f", "phased", "--target", "derived"] + pops,
    }


def run_timed(script, argv, env=None):
    t = time.perf_counter()
    r = subprocess.run([sys.executable, script] + argv, env=dict(os.environ, PG_TIMING="1", PG_PLACE_TRIALS="1", **(env or {})),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    wall = time.perf_counter() - t
    tm = None
    for ln in r.stderr.decode().splitlines():
        if ln.startswith("PG_TIMING "):
            tm = json.loads(ln[10:])
    return r.returncode, wall, tm, r.stderr.decode()[-1500:]


def gpu_mode(n_sites, n_dip):
    import bgzip
    tmp = tempfile.mkdtemp(prefix="pg_drivers_", dir=os.environ.get("PG_BENCH_TMP", "/tmp"))
    geno = os.path.join(tmp, "sample.geno")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "t2_write_sample.py"), geno, str(n_sites), str(n_dip)], stdout=subprocess.PIPE, check=True)
    n_in, n_gz = bgzip.bgzip_file(geno, geno + ".gz")
    os.remove(geno)
    names = ["s%d" % d for d in range(n_dip)]
    res = {"mode": "gpu", "sites": n_sites, "diploids": n_dip, "text_bytes": n_in, "file_bytes": n_gz, "drivers": {}}
    for tool, argv in driver_commands(geno + ".gz", os.path.join(tmp, "out"), names, 4, 50000).items():
        best = None
        for _ in range(2):
            rc, wall, tm, err = run_timed(os.path.join(ROOT, tool.split()[0]), argv, {"PG_OUT_GZIP_MODULE": "1"} if "PG_OUT_GZIP_MODULE" in tool else None)
            if rc != 0 or tm is None:
                best = {"error": err[-400:]}
                break
            tm.setdefault("total_s", wall)
            if best is None or tm["total_s"] < best["total_s"]:
                best = {"total_s": round(tm["total_s"], 4), "context_s": round(tm.get("context_s", 0.0), 4), "process_wall_s": round(wall, 3),
                        "sites_per_sec": round(n_sites / tm["total_s"], 1), "text_GBps": round(n_in / tm["total_s"] / 1e9, 2),
                        "tokenize_s": round(tm.get("tokenize_s", 0.0), 4), "prep_wait_s": round(tm.get("prep_wait_s", 0.0), 4),
                        "main_stats_s": round(tm.get("main_stats_s", 0.0), 4), "main_format_s": round(tm.get("main_format_s", 0.0), 4),
                        "windows": tm.get("windows"), "bgzf_blocks_inflated_on_device": tm.get("bgzf_blocks_inflated_on_device"),
                        "timing": {k: v for k, v in tm.items() if k.endswith("_s") and isinstance(v, float)}}
        res["drivers"][tool] = best
    for fn in os.listdir(tmp):
        os.remove(os.path.join(tmp, fn))
    os.rmdir(tmp)
    print(json.dumps(res))


def reference_mode(n_sites, n_dip):
    """the unmodified scripts (np.NaN restored for fourPopWindows.py as tests/golden/make_golden.py does: NumPy 2 dropped the alias);
    parity with them is the goldens' and the differential tools' business, this is their speed"""
    tmp = tempfile.mkdtemp(prefix="pg_drivers_ref_")
    rng = np.random.default_rng(1)
    names = ["s%d" % d for d in range(n_dip)]
    geno = os.path.join(tmp, "ref.geno.gz")
    # two alleles per site with population structure enough for non-trivial statistics; ~2 % missing
    hap = 2 * n_dip
    base = rng.integers(0, 4, size=n_sites)
    alt = (base + rng.integers(1, 4, size=n_sites)) % 4
    freq = rng.random((n_sites, 4)) ** 2
    per = hap // 4
    is_alt = rng.random((n_sites, hap)) < np.repeat(freq, per, axis=1)
    is_alt[rng.random(n_sites) < 0.5] = False
    letters = np.where(is_alt, np.array(list(b"ACGT"), dtype=np.uint8)[alt][:, None], np.array(list(b"ACGT"), dtype=np.uint8)[base][:, None])
    letters[rng.random((n_sites, hap)) < 0.02] = ord("N")
    with gzip.open(geno, "wb", compresslevel=4) as f:
        f.write(("#CHROM\tPOS\t" + "\t".join(names) + "\n").encode())
        for i in range(n_sites):
            row = letters[i]
            f.write(b"chr1\t%d\t" % (i + 1) + b"\t".join(bytes([row[2 * d]]) + b"/" + bytes([row[2 * d + 1]]) for d in range(n_dip)) + b"\n")
    shim = os.path.join(tmp, "run_ref.py")
    with open(shim, "w") as f:
        f.write("import sys, runpy\nimport numpy as np\nif not hasattr(np, 'NaN'):\n    np.NaN = np.nan\n"
                "sys.path.insert(0, %r)\nsys.argv = sys.argv[1:]\nrunpy.run_path(sys.argv[0], run_name='__main__')\n" % REF)
    res = {"mode": "reference", "sites": n_sites, "diploids": n_dip, "host_cpus": len(os.sched_getaffinity(0)), "drivers": {}}
    wind = max(n_sites // 8, 1000)
    cmds_ref = driver_commands(geno, os.path.join(tmp, "ref"), names, 4, wind)
    for tool in cmds_ref:
        ncpu = str(len(os.sched_getaffinity(0)))                     # the reference's own parallelism: worker processes per window / slice
        extra = ["-t", ncpu] if tool == "freq.py" else ["-T", ncpu]
        if (os.environ.get("DRV_ONLY") and os.environ["DRV_ONLY"] not in tool) or tool.startswith("freq.py -o"):
            continue
        t = time.perf_counter()
        r = subprocess.run([sys.executable, shim, os.path.join(REF, tool.split()[0])] + cmds_ref[tool] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t
        entry = {"reference_seconds": round(dt, 2), "reference_sites_per_sec": round(n_sites / dt, 1), "reference_rc": r.returncode,
                 "reference_workers": int(ncpu)}
        if r.returncode != 0:
            entry["reference_error"] = r.stderr.decode()[-300:]
        res["drivers"][tool] = entry
    print(json.dumps(res))
    for fn in os.listdir(tmp):
        os.remove(os.path.join(tmp, fn))
    os.rmdir(tmp)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--reference" in sys.argv:
        reference_mode(int(args[0]) if args else 20000, int(args[1]) if len(args) > 1 else 200)
    else:
        gpu_mode(int(args[0]) if args else 5_000_000, int(args[1]) if len(args) > 1 else 200)
