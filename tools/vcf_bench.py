#!/usr/bin/env python
"""VCF -> .geno: the parseVCF drop-in (genomics_general_amd/vcf.py on the native pg_encode_vcf) timed on a synthetic
GATK-style VCF (GT:AD:DP:GQ per sample, ~2 % indels, ~3 % multi-allelic sites, ~5 % missing calls), bgzipped, beside the
unmodified reference script when /root/reference is there (VCF_processing/parseVCF.py:49-191, 334-391).

    python tools/vcf_bench.py [n_sites] [n_samples] [--ref-sites N]

Writes one JSON line.  The reference is timed on the first N sites of the same file (it is a per-line Python loop: seconds per site
do not depend on the file's length) and both outputs of those sites are compared byte for byte.
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/VCF_processing/parseVCF.py"


def write_vcf(path, n_sites, n_samples, seed=5):
    """text of a VCF; returns its size.  The per-sample part of a line is made for one slab of 20 000 sites and used again by the
    later slabs under fresh CHROM / POS / QUAL columns (a BGZF member holds 64 KiB of text, a slab is 60 MB: the repetition is invisible
    to the compressor and to the parser)."""
    rng = np.random.default_rng(seed)
    bases = [b"A", b"C", b"G", b"T"]
    slab = min(20000, n_sites)
    ref = rng.integers(0, 4, size=slab)
    alt1 = (ref + rng.integers(1, 4, size=slab)) % 4
    alt2 = (ref + rng.integers(1, 4, size=slab)) % 4
    kind = rng.random(slab)                                       # < .02 indel, < .05 tri-allelic, < .55 invariant, else SNP
    gt = rng.random((slab, n_samples))
    dp = rng.integers(0, 60, size=(slab, n_samples))
    gq = rng.integers(0, 99, size=(slab, n_samples))
    mid, tails = [], []
    for i in range(slab):
        r = bases[ref[i]]
        if kind[i] < 0.02:
            refs, alts, na = r + b"TG", r, 1
        elif kind[i] < 0.05 and alt1[i] != alt2[i]:
            refs, alts, na = r, bases[alt1[i]] + b"," + bases[alt2[i]], 2
        elif kind[i] < 0.55:
            refs, alts, na = r, b".", 0
        else:
            refs, alts, na = r, bases[alt1[i]], 1
        cells = []
        g, d, q = gt[i], dp[i], gq[i]
        for s in range(n_samples):
            x = g[s]
            if x < 0.05:
                cells.append(b"./.:.:.:.")
                continue
            if na == 0 or x < 0.6:
                a, b = 0, 0
            elif x < 0.85:
                a, b = 0, 1
            elif x < 0.97 or na < 2:
                a, b = 1, 1
            else:
                a, b = 1, 2
            cells.append(b"%d/%d:%d,%d:%d:%d" % (a, b, d[s] // 2, d[s] - d[s] // 2, d[s], q[s]))
        mid.append(b"\t.\t" + refs + b"\t" + alts + b"\t")
        tails.append(b"\tDP=%d\tGT:AD:DP:GQ\t" % int(d.sum()) + b"\t".join(cells) + b"\n")
    with open(path, "wb") as f:
        f.write(b"##fileformat=VCFv4.2\n##source=vcf_bench\n##contig=<ID=chr1>\n##contig=<ID=chr2>\n")
        f.write(b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"ind%03d" % i for i in range(n_samples)) + b"\n")
        pos, done = 0, 0
        while done < n_sites:
            m = min(slab, n_sites - done)
            p = pos + np.cumsum(rng.integers(1, 40, size=m))
            pos = int(p[-1])
            qual = rng.integers(5, 5000, size=m)
            out = []
            for i in range(m):
                chrom = b"chr1\t" if done + i < n_sites // 2 else b"chr2\t"
                out.append(chrom + b"%d" % p[i] + mid[i] + b"%d" % qual[i] + (b"\tPASS" if qual[i] > 30 else b"\tLowQual") + tails[i])
            f.write(b"".join(out))
            done += m
    return os.path.getsize(path)


def timed(argv, stdout=None, env=None, info=None):
    t = time.perf_counter()
    r = subprocess.run(argv, stdout=stdout, stderr=subprocess.PIPE, env=dict(os.environ, PG_TIMING="1", **(env or {})))
    dt = time.perf_counter() - t
    if r.returncode != 0:
        raise SystemExit("%s failed:\n%s" % (" ".join(argv), r.stderr.decode()[-2000:]))
    if info is not None:
        for ln in r.stderr.decode().splitlines():
            if ln.startswith("PG_TIMING "):
                info.update(json.loads(ln[10:]))
    return dt


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_sites = int(args[0]) if args else 400000
    n_samples = int(args[1]) if len(args) > 1 else 200
    ref_sites = 20000
    if "--ref-sites" in sys.argv:
        ref_sites = int(sys.argv[sys.argv.index("--ref-sites") + 1])
    tmp = tempfile.mkdtemp(prefix="pg_vcfbench_", dir=os.environ.get("PG_BENCH_TMP", "/tmp"))
    vcf = os.path.join(tmp, "in.vcf")
    size = write_vcf(vcf, n_sites, n_samples)
    bgz = vcf + ".gz"
    # the input is what htslib's bgzip writes (zlib, level 6); the drop-in's own writer keeps its default (the library's compressor)
    timed([sys.executable, os.path.join(HERE, "bgzip.py"), vcf, bgz], env={"PG_BGZF_ZLIB": "1"})
    shim = os.path.join(ROOT, "VCF_processing", "parseVCF.py")
    opts = ["--skipIndels", "--minQual", "30", "--gtf", "flag=DP", "min=8", "--gtf", "flag=GQ", "min=20"]
    res = {"sites": n_sites, "samples": n_samples, "vcf_bytes": size, "vcf_gz_bytes": os.path.getsize(bgz), "options": " ".join(opts), "legs": {}}
    legs = [("vcf.gz -> geno.gz", bgz, os.path.join(tmp, "o1.geno.gz"), [], {}),
            ("vcf.gz -> geno.gz, members inflated by the host threads (PG_BGZF_DEVICE=0)", bgz, os.path.join(tmp, "o1h.geno.gz"), [], {"PG_BGZF_DEVICE": "0"}),
            ("vcf -> geno", vcf, os.path.join(tmp, "o2.geno"), [], {}),
            ("vcf.gz -> pgeno (raw cells)", bgz, None, ["--packed", os.path.join(tmp, "o3.pgeno"), "--packedCodec", "none"], {})]
    only = [int(x) for x in os.environ.get("VCF_LEGS", "0,1,2,3").split(",")]
    for name, src, dst, extra, env in [legs[k] for k in only]:
        best, binfo = None, {}
        for _ in range(int(os.environ.get("VCF_REPS", "3"))):
            info = {}
            dt = timed([sys.executable, shim, "-i", src] + (["-o", dst] if dst else []) + opts + extra, env=env, info=info)
            if best is None or dt < best:
                best, binfo = dt, info
        res["legs"][name] = {"seconds": round(best, 3), "sites_per_sec": round(n_sites / best), "vcf_text_MBps": round(size / best / 1e6, 1),
                             "timing": binfo}
    import gzip as _gz
    if only == [0, 1, 2, 3]:
        with _gz.open(os.path.join(tmp, "o1.geno.gz"), "rb") as f1, _gz.open(os.path.join(tmp, "o1h.geno.gz"), "rb") as f2, \
                open(os.path.join(tmp, "o2.geno"), "rb") as f3:
            t1 = f1.read()
            res["outputs_of_the_legs_identical"] = bool(t1 == f2.read() and t1 == f3.read())
            res["geno_text_bytes"] = len(t1)
    if os.path.exists(REF) and ref_sites:
        head = os.path.join(tmp, "head.vcf")
        with open(vcf, "rb") as f, open(head, "wb") as g:
            k = 0
            for line in f:
                g.write(line)
                if not line.startswith(b"#"):
                    k += 1
                    if k >= ref_sites:
                        break
        hsize = os.path.getsize(head)
        with gzip.open(head + ".gz", "wb", compresslevel=6) as g:
            g.write(open(head, "rb").read())
        o_ref, o_new = os.path.join(tmp, "ref.geno"), os.path.join(tmp, "new.geno")
        dt_ref = timed([sys.executable, REF, "-i", head + ".gz", "-o", o_ref] + opts)
        dt_new = timed([sys.executable, shim, "-i", head + ".gz", "-o", o_new] + opts)
        same = open(o_ref, "rb").read() == open(o_new, "rb").read()
        res["reference"] = {"sites": k, "seconds": round(dt_ref, 3), "sites_per_sec": round(k / dt_ref), "vcf_text_MBps": round(hsize / dt_ref / 1e6, 2),
                            "drop_in_seconds_same_file": round(dt_new, 3), "outputs_identical": same,
                            "note": "unmodified VCF_processing/parseVCF.py (one process; it has no thread option) on the first sites of the same file"}
        res["drop_in_over_reference"] = round(res["legs"]["vcf.gz -> geno.gz"]["sites_per_sec"] / (k / dt_ref), 1)
    res["host_cpus"] = len(os.sched_getaffinity(0))
    print(json.dumps(res))
    for fn in os.listdir(tmp):
        os.remove(os.path.join(tmp, fn))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
