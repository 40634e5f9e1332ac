#!/usr/bin/env python
"""Differential fuzz of the parseVCF.py drop-in against the UNMODIFIED reference VCF_processing/parseVCF.py (build container only):
random VCF files (tests/golden/make_golden_vcf.py's two generators with random seeds, sample counts, a haploid sample, genotypes of
the wrong ploidy) x random option sets (sample subsets and order, contig filters by list and by file, --minQual, --gtf filters of
every kind, --skipIndels, --excludeDuplicates, --simplifyALT / --expandMulti on the CIGAR files, --maxREFlen, --ploidy /
--ploidyFile / --ploidyMismatchToMissing, --keepPartial, --addRefTrack, --noHeader, --field, --missing, --outSep).  The outputs
must be byte-identical; where the reference stops with an error the drop-in must stop too.
    python tools/diff_reference_vcf.py [n_cases] [seed]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_vcf as MV                                                    # noqa: E402

REF = "/root/reference/VCF_processing/parseVCF.py"
OURS = os.path.join(ROOT, "VCF_processing", "parseVCF.py")


def pick(rng, seq):
    return seq[int(rng.integers(0, len(seq)))]


def make_case(tmp, case, rng):
    n = int(pick(rng, [1, 2, 4, 6, 9]))
    cigar = rng.random() < 0.35
    hap = int(rng.integers(0, n)) if rng.random() < 0.3 else None
    wrong = float(pick(rng, [0.0, 0.0, 0.03, 0.2]))
    vcf = os.path.join(tmp, "c%d.vcf.gz" % case)
    seed = int(rng.integers(1, 1 << 30))
    if cigar:
        MV.make_cigar_vcf(vcf, seed, n_samples=n, hap_sample=hap, wrong_ploidy=wrong)
    else:
        MV.make_vcf(vcf, seed, n_samples=n, hap_sample=hap, snps_only=rng.random() < 0.2, wrong_ploidy=wrong)
    if rng.random() < 0.3:                                                       # plain text input
        plain = vcf[:-3]
        import gzip
        with gzip.open(vcf, "rb") as f, open(plain, "wb") as g:
            g.write(f.read())
        vcf = plain
    elif rng.random() < 0.6:                                                     # what bgzip writes: members of a few hundred bytes to 64 KiB of text
        import gzip
        from genomics_general_amd import genoio
        with gzip.open(vcf, "rb") as f:
            text = f.read()
        with open(vcf, "wb") as g:
            g.write(genoio.bgzf_compress(text, 6, int(pick(rng, [300, 2000, 20000, 65280]))).tobytes())
    names = ["s%d" % k for k in range(n)]
    argv = []
    if rng.random() < 0.35:
        k = int(rng.integers(1, n + 1))
        argv += ["-s", ",".join(str(x) for x in rng.choice(names, size=k, replace=False))]
    r = rng.random()
    contigs = [c for c in ("chr1", "chr2", "chr3", "chrX") if rng.random() < 0.5] or ["chr2"]
    if r < 0.15:
        argv += ["--include", ",".join(contigs)]
    elif r < 0.3:
        argv += ["--exclude", ",".join(contigs)]
    elif r < 0.4:
        cf = os.path.join(tmp, "c%d.contigs" % case)
        with open(cf, "w") as f:
            f.write("\n".join(contigs) + "\n")
        argv += [pick(rng, ["--includeFile", "--excludeFile"]), cf]
    if rng.random() < 0.3:
        argv += ["--minQual", str(int(pick(rng, [0, 10, 30, 80])))]
    for _ in range(int(pick(rng, [0, 0, 1, 1, 2, 3]))):
        flag = pick(rng, ["DP", "GQ", "AD"] if not cigar else ["DP", "GQ"])
        g = ["--gtf", "flag=" + flag]
        if rng.random() < 0.8:
            g += ["min=" + str(int(pick(rng, [1, 5, 20, 50])))]
        if rng.random() < 0.3:
            g += ["max=" + str(int(pick(rng, [10, 30, 90])))]
        if rng.random() < 0.3:
            g += ["siteTypes=" + ",".join(t for t in ("SNP", "MONO", "INDEL") if rng.random() < 0.6 or t == "SNP")]
        if rng.random() < 0.3:
            g += ["gtTypes=" + ",".join(t for t in ("Het", "HomRef", "HomAlt") if rng.random() < 0.5 or t == "Het")]
        if rng.random() < 0.3:
            g += ["samples=" + ",".join(str(x) for x in rng.choice(names, size=int(rng.integers(1, n + 1)), replace=False))]
        argv += g
    if rng.random() < 0.6:
        argv += ["--skipIndels"]
    if rng.random() < 0.3:
        argv += ["--excludeDuplicates"]
    if cigar:
        r = rng.random()
        if r < 0.45:
            argv += ["--simplifyALT"]
        elif r < 0.9:
            argv += ["--expandMulti"] + (["--simplifyALT"] if rng.random() < 0.3 else [])
    if rng.random() < 0.25:
        argv += ["--maxREFlen", str(int(pick(rng, [1, 2, 3])))]
    if hap is not None and rng.random() < 0.8:
        pf = os.path.join(tmp, "c%d.ploidy" % case)
        with open(pf, "w") as f:
            f.write("s%d 1\n" % hap)
            if n > 1 and rng.random() < 0.5:
                f.write("s%d 2\n" % ((hap + 1) % n))
        argv += ["--ploidyFile", pf]
    elif rng.random() < 0.1:
        argv += ["--ploidy", str(int(pick(rng, [1, 2])))]
    if rng.random() < 0.5 and (hap is not None or wrong > 0):
        argv += ["--ploidyMismatchToMissing"]
    if rng.random() < 0.3:
        argv += ["--keepPartial"]
    if rng.random() < 0.3:
        argv += ["--addRefTrack"]
    if rng.random() < 0.2:
        argv += ["--noHeader"]
    if rng.random() < 0.2:
        argv += ["--field", pick(rng, ["DP", "GQ", "AD"] if not cigar else ["DP", "GQ"])]
    if rng.random() < 0.25:
        argv += ["--missing", pick(rng, ["NA", "X", "?", "."])]
    if rng.random() < 0.25:
        argv += ["--outSep", pick(rng, [" ", ",", ";", "::"])]
    return vcf, argv


def run(script, vcf, argv, out):
    try:
        r = subprocess.run([sys.executable, script, "-i", vcf, "-o", out] + argv, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    except subprocess.TimeoutExpired:
        return None, "timeout"
    lines = [ln for ln in r.stderr.decode().strip().splitlines() if ln.strip()]
    return r.returncode, (lines[-1][:160] if lines else "")


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260926
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp(prefix="pg_vcf_")
    bad = 0
    for case in range(n_cases):
        vcf, argv = make_case(tmp, case, rng)
        ref_out, got_out = os.path.join(tmp, "ref%d.geno" % case), os.path.join(tmp, "got%d.geno" % case)
        rc_ref, msg_ref = run(REF, vcf, argv, ref_out)
        rc, msg = run(OURS, vcf, argv, got_out)
        note = ""
        if rc_ref != 0:
            v = "both-stop" if rc != 0 else "ref-only-stop"
            note = "\n           reference: %s\n           drop-in:   %s" % (msg_ref, msg)
        elif rc != 0:
            v, bad, note = "FAILED", bad + 1, "\n           drop-in: " + msg
        else:
            want, got = open(ref_out, "rb").read(), open(got_out, "rb").read()
            v = "ok" if want == got else "DIFF"
            if want != got:
                bad += 1
                wl, gl = want.splitlines(), got.splitlines()
                first = next((k for k in range(min(len(wl), len(gl))) if wl[k] != gl[k]), min(len(wl), len(gl)))
                note = "\n           line %d of %d / %d:\n             ref %r\n             got %r" % (
                    first + 1, len(wl), len(gl), wl[first][:150] if first < len(wl) else None, gl[first][:150] if first < len(gl) else None)
        rows = open(ref_out, "rb").read().count(b"\n") if os.path.exists(ref_out) else 0
        print("case %3d  %-13s rows %5d  %s %s%s" % (case, v, rows, os.path.basename(vcf), " ".join(
            os.path.basename(a) if a.startswith(tmp) else a for a in argv), note), flush=True)
    print("differences: %d of %d cases (seed %d)   files kept in %s" % (bad, n_cases, seed, tmp))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
