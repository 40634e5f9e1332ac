#!/usr/bin/env python
"""Regenerate tests/golden/ by running the UNMODIFIED reference (/root/reference) on seeded synthetic
.geno fixtures.  Only runs in the build container (the reference does not travel to the GPU box);
the fixtures and outputs it writes are committed.

    python tests/golden/make_golden.py [case-name ...]
"""
import gzip
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from genomics_general_amd import synth  # noqa: E402
from cases import AUX_FILES, CASES, FIXTURES  # noqa: E402

REF = "/root/reference"
# np.NaN was removed in NumPy 2; distMat.py:50 uses it on failed windows.  Injected in the harness
# process only; the reference files are not edited.
WRAP = ("import sys, runpy, numpy as np; np.NaN = np.nan; sys.path.insert(0, %r); "
        "sys.argv = sys.argv[1:]; runpy.run_path(sys.argv[0], run_name='__main__')" % REF)


def fixture_sites(p):
    """positions present per scaffold: hash-thinned to `density`, optional empty gap on scaffold 0."""
    scaf_ids, poss = [], []
    for k, ln in enumerate(p["scaf_len"]):
        pos = np.arange(1, ln + 1, dtype=np.int64)
        if p["density"] < 1.0:
            key = synth.site_keys(p["seed"] + 77, np.full(ln, k), pos)
            keep = ((key >> np.uint64(5)) & np.uint64(0xFFFF)).astype(np.int64) < int(p["density"] * 65536)
            pos = pos[keep]
        if k == 0 and "gap" in p:
            pos = pos[(pos < p["gap"][0]) | (pos > p["gap"][1])]
        if "pos_offset" in p:
            pos = pos + int(p["pos_offset"][k])
        scaf_ids.append(np.full(len(pos), k, dtype=np.int64))
        poss.append(pos)
    return np.concatenate(scaf_ids), np.concatenate(poss)


def make_fixture(name):
    p = FIXTURES[name]
    path = os.path.join(HERE, name + ".geno.gz")
    sid, pos = fixture_sites(p)
    codes = synth.gen_codes(p["seed"], sid, pos, p["n_dip"], p["n_pops"], var_thr=p["var_thr"], miss_thr=p["miss_thr"])
    if p.get("multi_frac"):
        # a share of the sites gets alleles drawn uniformly from A,C,G,T (missing calls stay missing): three and four alleles per site
        rng = np.random.default_rng(p["seed"])
        pick = rng.random(len(pos)) < p["multi_frac"]
        rnd = (1 << rng.integers(0, 4, size=codes.shape)).astype(np.int8)
        codes = np.where(pick[:, None] & (codes != 0), rnd, codes).astype(np.int8)
    if p["fmt"] == "haplo":
        names = ["s%d_%s" % (d, ab) for d in range(p["n_dip"]) for ab in "AB"]
    else:
        names = ["s%d" % d for d in range(p["n_dip"])]
    scaf_names = ["chr%d" % (k + 1) for k in range(len(p["scaf_len"]))]
    with gzip.GzipFile(path, "wb", mtime=0) as raw:
        import io
        txt = io.TextIOWrapper(raw, newline="\n")
        if p.get("haploid_where"):
            # ploidy that changes along the file: the cells of some samples lose their second allele over ranges of positions
            mem = io.StringIO()
            synth.write_geno(mem, scaf_names, sid, pos, codes, names, sep=p["sep"], fmt=p["fmt"])
            lines = mem.getvalue().split("\n")
            for i in range(len(pos)):
                f = lines[1 + i].split("\t")
                for samples, k, a, b in p["haploid_where"]:
                    if int(sid[i]) == k and a <= int(pos[i]) <= b:
                        for d in samples:
                            f[2 + d] = f[2 + d][0]
                lines[1 + i] = "\t".join(f)
            txt.write("\n".join(lines))
        else:
            synth.write_geno(txt, scaf_names, sid, pos, codes, names, sep=p["sep"], fmt=p["fmt"], haploid=tuple(p.get("haploid", ())))
        txt.flush()
    return path


def run_case(case):
    geno = os.path.join(HERE, case["fixture"] + ".geno.gz")
    out = os.path.join(HERE, case["name"] + ".out")
    argv = [a.format(geno=geno, dir=HERE, out=out) for a in case["argv"]]
    cmd = [sys.executable, "-c", WRAP, os.path.join(REF, case["tool"])] + argv + ["-o", out]
    env = dict(os.environ, PYTHONHASHSEED="0")
    r = subprocess.run(cmd, cwd=HERE, env=env, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        sys.stderr.write(r.stderr.decode()[-2000:])
        raise SystemExit("reference failed on " + case["name"])
    return out


def main():
    want = set(sys.argv[1:])
    for fn, txt in AUX_FILES.items():
        with open(os.path.join(HERE, fn), "w") as f:
            f.write(txt)
    for name in FIXTURES:
        if not want or any(c["fixture"] == name and c["name"] in want for c in CASES):
            print("fixture", name, make_fixture(name))
    for case in CASES:
        if want and case["name"] not in want:
            continue
        out = run_case(case)
        with open(out) as f:
            n = sum(1 for _ in f)
        print("golden", case["name"], n, "lines")


if __name__ == "__main__":
    main()
