"""Adapter: reference-style argv -> oracle driver call (test infrastructure)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import popgen_oracle as orc  # noqa: E402


def _take(argv, flag, n=1, default=None):
    if flag not in argv:
        return default
    i = argv.index(flag)
    vals = argv[i + 1:i + 1 + n]
    return vals[0] if n == 1 else vals


def _pops(argv, flags):
    out = []
    i = 0
    while i < len(argv):
        if argv[i] in flags:
            out.append((argv[i + 1], argv[i + 2].split(",")))
            i += 3
        else:
            i += 1
    return out


def _pops_with_file(argv, flags):
    """-p/-P name [samples] plus --popsFile (popgenWindows.py:262-273, ABBABABAwindows.py:203-216)."""
    out = []
    i = 0
    while i < len(argv):
        if argv[i] in flags:
            name = argv[i + 1]
            members = []
            if i + 2 < len(argv) and not argv[i + 2].startswith("-"):
                members = argv[i + 2].split(",")
                i += 1
            out.append((name, members))
            i += 2
        else:
            i += 1
    pf = _take(argv, "--popsFile")
    if pf:
        names = [p[0] for p in out]
        with open(pf) as f:
            for ln in f:
                parts = ln.split()
                if len(parts) >= 2 and parts[1] in names:
                    out[names.index(parts[1])][1].append(parts[0])
    return out


def _multi(argv, flag):
    if flag not in argv:
        return None
    i = argv.index(flag) + 1
    vals = []
    while i < len(argv) and not argv[i].startswith("-"):
        vals.append(argv[i])
        i += 1
    return vals


def _ploidy(argv, haploid_is_list):
    """--haploid (comma list in popgenWindows / ABBA, space list in distMat / freq) and --ploidyFile."""
    out = {}
    pf = _take(argv, "--ploidyFile")
    if pf:
        with open(pf) as f:
            for ln in f:
                parts = ln.split()
                if len(parts) >= 2:
                    out[parts[0]] = int(parts[1])
    if "--inferPloidy" in argv:                       # popgenWindows.py:299-300: ploidy None for every sample
        return "infer"
    if "--haploid" in argv:
        names = _multi(argv, "--haploid") if haploid_is_list else _take(argv, "--haploid").split(",")
        for nm in names:
            out[nm] = 1
    return out or None


def _lines(path):
    with open(path) as f:
        return [ln.rstrip() for ln in f.readlines()]


def _common(argv):
    kw = dict(wind_type=_take(argv, "--windType", default="coordinate"))
    w = _take(argv, "-w")
    kw["wind_size"] = int(w) if w else None
    s = _take(argv, "-s")
    kw["step"] = int(s) if s else None
    m = _take(argv, "-m")
    kw["min_sites"] = int(m) if m is not None else 1
    d = _take(argv, "-D")
    kw["max_dist"] = int(d) if d else float("inf")
    c = _take(argv, "--windCoords")
    if c:
        kw["coords"] = [tuple([p[0], int(p[1]), int(p[2])] + p[3:4]) for p in (ln.split() for ln in _lines(c))]
    inc, exc = _take(argv, "--include"), _take(argv, "--exclude")
    kw["include"] = _lines(inc) if inc else None
    kw["exclude"] = _lines(exc) if exc else None
    kw["add_id"] = "--addWindowID" in argv
    kw["write_failed"] = "--writeFailedWindows" in argv
    md = _take(argv, "--minData")
    kw["min_data"] = float(md) if md else 0.01
    return kw


def run(tool, argv):
    geno = _take(argv, "-g")
    fmt = _take(argv, "-f")
    if tool == "popgenWindows.py":
        kw = _common(argv)
        if kw["wind_type"] == "predefined":
            kw["coords"] = [c[:3] for c in kw["coords"]]        # popgenWindows.py:240 keeps 3 columns
        o = _take(argv, "-O") or _take(argv, "--overlap")
        kw["overlap"] = int(o) if o else 0
        pops = (_pops_with_file(argv, ("-p",)) if "--popsFile" in argv else _pops(argv, ("-p",))) or None
        samples = _take(argv, "--samples")
        analysis = ("popDist", "popPairDist")
        if "--analysis" in argv:
            i = argv.index("--analysis") + 1
            analysis = []
            while i < len(argv) and not argv[i].startswith("-"):
                analysis.append(argv[i])
                i += 1
        r = _take(argv, "--roundTo")
        if pops is None and samples:
            pops = []
            kw["_samples"] = samples.split(",")
        hd = _take(argv, "--hapDist")
        kw["ploidy"] = _ploidy(argv, False)
        return orc.popgen_windows_csv(geno, fmt, pops, analysis=tuple(analysis), round_to=int(r) if r else 4,
                                      samples_only=kw.pop("_samples", None), hap_dist=float(hd) if hd else 0, **kw)
    if tool == "ABBABABAwindows.py":
        kw = _common(argv)
        o = _take(argv, "--overlap")
        kw["overlap"] = int(o) if o else 0
        pops4 = (_pops_with_file(argv, ("-P1", "-P2", "-P3", "-O")) if "--popsFile" in argv
                 else _pops(argv, ("-P1", "-P2", "-P3", "-O")))
        return orc.abbababa_windows_csv(geno, fmt, pops4, ploidy=_ploidy(argv, False), **kw)
    if tool == "fourPopWindows.py":
        kw = _common(argv)
        o = _take(argv, "--overlap")
        kw["overlap"] = int(o) if o else 0
        pops4 = _pops(argv, ("-P1", "-P2", "-P3", "-O"))
        return orc.fourpop_windows_csv(geno, fmt, pops4, polarize="--polarize" in argv, fixed="--fixed" in argv, ploidy=_ploidy(argv, False), **kw)
    if tool == "distMat.py":
        w, s, m = _take(argv, "-w"), _take(argv, "-s"), _take(argv, "-m")
        r = _take(argv, "--roundTo")
        mi, ov = _take(argv, "-Mi"), _take(argv, "-O")
        c = _take(argv, "--windCoords")
        coords = [tuple([p[0], int(p[1]), int(p[2])] + p[3:4]) for p in (ln.split() for ln in _lines(c))] if c else None
        return orc.distmat_text(geno, fmt, wind_size=int(w) if w else None, step=int(s) if s else None,
                                min_sites=int(m) if m is not None else 1,
                                wind_type=_take(argv, "--windType", default="coordinate"),
                                out_format=_take(argv, "--outFormat", default="phylip"),
                                round_to=int(r) if r else 4, include_same="--includeSameWithSame" in argv,
                                min_per_ind=int(mi) if mi else None, samples=_multi(argv, "--samples"),
                                overlap=int(ov) if ov else 0, ploidy=_ploidy(argv, True),
                                write_failed="--writeFailedWindows" in argv, coords=coords)
    if tool == "freq.py":
        pops = _pops(argv, ("-p",))
        if "--indFreqs" in argv:                                 # freq.py:250-253: every individual is its own population
            with orc.open_text(geno) as fh:
                names = fh.readline().split()[2:]
            pops = [(nm, [nm]) for nm in names]
        elif not pops:
            with orc.open_text(geno) as fh:
                names = fh.readline().split()[2:]
            pops = [("all", names)]
        md, th = _take(argv, "--minData"), _take(argv, "--threshold")
        return orc.freq_tsv(geno, fmt, pops, target=_take(argv, "--target"), as_counts="--asCounts" in argv,
                            min_data=float(md) if md else 0, threshold=float(th) if th else None,
                            keep_nan="--keepNanLines" in argv)
    raise ValueError(tool)
