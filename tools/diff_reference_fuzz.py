#!/usr/bin/env python
"""Differential fuzz against the UNMODIFIED reference over the drivers' OPTIONS (build container only: needs /root/reference).
tools/diff_reference_ranks.py varies the windows and the rank count on one kind of command line; this one varies the command
line: genotype format (phased with / or |, pairs, diplo, haplo), mixed ploidy (--haploid / --ploidy / --ploidyFile), half-missing
genotypes, .gz input, populations by -p / --popsFile / none, every --analysis subset, --hapDist, --samples, coordinate / sites /
predefined / cat windows, -D, --include / --exclude, --minData, --roundTo, distMat's formats and -Mi, fourPop's --polarize /
--fixed, freq.py's --target derived / --asCounts / --indFreqs / --keepNanLines / --threshold.  Every case: the reference's
single-process output against the drop-in driver (CPU stand-in engine with the oracle's numbers) on one rank and on 2 or 3 ranks
with a random block size.  `ok` = byte-identical (het_* columns aligned: hash order in the reference); `tie(n)` = n cells one
unit of the rounding digit apart; anything else is a DIFF.
    python tools/diff_reference_fuzz.py [n_cases] [seed] [jobs]"""
import concurrent.futures
import os
import signal
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from genomics_general_amd import synth                                          # noqa: E402
import test_dist                                                               # noqa: E402  (CLI_WORKER)
import test_gpu_golden as G                                                    # noqa: E402  (compare_text)
from golden_util import align_columns                                          # noqa: E402

REF = "/root/reference"
WRAP = ("import sys, runpy, numpy as np; np.NaN = np.nan; sys.path.insert(0, %r); "
        "sys.argv = sys.argv[1:]; runpy.run_path(sys.argv[0], run_name='__main__')" % REF)
ANALYSES = ["popFreq", "popDist", "popPairDist", "indPairDist", "indHet", "hapStats"]


def pick(rng, seq):
    return seq[int(rng.integers(0, len(seq)))]


LONG = {"on": False}      # long_windows(): scaffolds of 6000 - 14 000 sites and windows of more than 4096 sites (fixed-tree sums + refinement)


def make_input(tmp, case, rng, tool):
    """-> dict(geno, fmt, names (columns), n_dip, scafs, lens, ploidy_argv)"""
    n_scaf = int(pick(rng, [1, 1, 2, 3, 4]))
    n_dip = int(pick(rng, [4, 5, 6, 8, 12]))
    lens = [int(rng.integers(300, 3000)) for _ in range(n_scaf)]
    density = float(pick(rng, [1.0, 0.6, 0.2, 0.05]))
    if LONG["on"]:
        n_scaf = int(pick(rng, [1, 2]))
        lens = [int(rng.integers(6000, 14000)) for _ in range(n_scaf)]
        density = float(pick(rng, [1.0, 1.0, 0.7]))
    sid, pos = [], []
    for k, ln in enumerate(lens):
        p = np.arange(1, ln + 1)
        p = p[rng.random(ln) < density]
        if rng.random() < 0.3 and len(p) > 50:
            a = int(rng.integers(0, len(p) // 2))
            p = np.concatenate([p[:a], p[a + len(p) // 4:]])
        if len(p) == 0:
            p = np.array([int(rng.integers(1, ln + 1))])
        sid.append(np.full(len(p), k))
        pos.append(p)
    sid, pos = np.concatenate(sid), np.concatenate(pos)
    codes = synth.gen_codes(int(rng.integers(1, 1 << 30)), sid, pos, n_dip, int(pick(rng, [1, 2, 4])),
                            var_thr=int(pick(rng, [3000, 6554, 30000, 60000])), miss_thr=int(pick(rng, [0, 1000, 6000, 20000, 45000])))
    if tool == "freq.py":
        fmt = pick(rng, ["phased", "phased", "diplo", "pairs"])
    else:
        fmt = pick(rng, ["phased", "phased", "phased", "pairs", "diplo", "haplo"])
    if fmt != "diplo" and rng.random() < 0.4:                                   # half-missing genotypes: A/N
        codes = np.where(rng.random(codes.shape) < float(pick(rng, [0.02, 0.2])), np.int8(0), codes)
    haploid = ()
    if fmt in ("phased", "pairs") and rng.random() < 0.3:
        haploid = tuple(sorted(int(x) for x in rng.choice(n_dip, size=int(rng.integers(1, max(2, n_dip // 2))), replace=False)))
    if fmt == "haplo":
        names = ["h%d" % k for k in range(2 * n_dip)]
    else:
        names = ["s%d" % d for d in range(n_dip)]
    scafs = [pick(rng, ["chr%d", "scaffold_%d", "%d"]) % (k + 1) for k in range(n_scaf)]
    geno = os.path.join(tmp, "c%d.geno%s" % (case, ".gz" if rng.random() < 0.25 else ""))
    synth.write_geno(geno, scafs, sid, pos, codes, names, sep=pick(rng, ["/", "/", "|"]), fmt=fmt, haploid=haploid)
    # --inferPloidy on a file whose ploidy changes along it (genomics.py:1108-1111: per window and sample): the cells of some samples
    # lose their second allele over one to three stretches of rows (window boundaries fall anywhere in them)
    shifting = fmt in ("phased", "pairs") and tool != "freq.py" and rng.random() < 0.25
    if shifting:
        import gzip
        op = gzip.open if geno.endswith(".gz") else open
        with op(geno, "rt") as f:
            lines = f.read().splitlines()
        n_rows = len(lines) - 1
        for _ in range(int(rng.integers(1, 4))):
            who = rng.choice(n_dip, size=int(rng.integers(1, max(2, n_dip // 2))), replace=False)
            a = int(rng.integers(0, max(n_rows, 1)))
            b = min(n_rows, a + int(rng.integers(1, max(2, n_rows))))
            for r in range(a, b):
                f_ = lines[1 + r].split("\t")
                for d in who:
                    f_[2 + int(d)] = f_[2 + int(d)][0]
                lines[1 + r] = "\t".join(f_)
        with op(geno, "wt") as f:
            f.write("\n".join(lines) + "\n")
    header_argv = []
    if rng.random() < 0.3:                                       # irregular text: blanks as separators, comment lines, no header line
        import gzip
        op = gzip.open if geno.endswith(".gz") else open
        with op(geno, "rt") as f:
            lines = f.read().splitlines()
        # (freq.py reads a comment line as a site: scaffold `#`, garbage counts; the drop-in skips it like the window drivers)
        how = pick(rng, ["blanks", "comments", "noheader", "crlf", "trailing"] if tool != "freq.py" else ["blanks", "crlf", "trailing"])
        if how == "blanks":
            lines = [ln.replace("\t", " ") for ln in lines]
        elif how == "crlf":
            lines = [ln + "\r" for ln in lines]
        elif how == "trailing":
            lines = [ln + pick(rng, [" ", "\t", "  "]) for ln in lines]
        elif how == "comments":
            for _ in range(int(rng.integers(1, 4))):
                lines.insert(int(rng.integers(1, len(lines) + 1)), "# a comment line")
        elif tool != "freq.py":
            header_argv = (["--headers"] + lines[0].split()) if tool == "distMat.py" else ["--header", lines[0]]
            lines = lines[1:]
        with op(geno, "wt") as f:
            f.write("\n".join(lines) + "\n")
    if geno.endswith(".gz") and rng.random() < 0.6:
        # the way bgzip writes it (BGZF: members of a few hundred to a few thousand bytes of text here, so that lines straddle them):
        # gzip.open reads it like any gzip file, the drivers take the members apart and inflate them member-wise (on the device)
        import gzip
        from genomics_general_amd import genoio
        with gzip.open(geno, "rb") as f:
            text = f.read()
        with open(geno, "wb") as f:
            f.write(genoio.bgzf_compress(text, level=int(rng.integers(1, 10)), block=int(rng.integers(300, 20000))).tobytes())
    ploidy_argv = []
    if shifting:
        ploidy_argv = ["--inferPloidy"]
    elif haploid:
        # a --ploidy LIST is dealt to the samples in the hash order of a set once populations are named (popgenWindows.py:277-296)
        how = pick(rng, ["haploid", "file"])
        if how == "haploid":
            hn = [names[k] for k in haploid]
            ploidy_argv = ["--haploid"] + (hn if tool in ("distMat.py", "freq.py") else [",".join(hn)])
        elif how == "ploidy":
            ploidy_argv = ["--ploidy"] + [("1" if k in haploid else "2") for k in range(n_dip)]
        else:
            pf = os.path.join(tmp, "c%d.ploidy" % case)
            with open(pf, "w") as f:
                for k in range(n_dip):
                    f.write("%s\t%d\n" % (names[k], 1 if k in haploid else 2))
            ploidy_argv = ["--ploidyFile", pf]
    elif tool != "freq.py" and fmt != "haplo" and rng.random() < 0.15:
        ploidy_argv = pick(rng, [["--inferPloidy"], ["--ploidy", "2"]])
    return dict(geno=geno, fmt=fmt, names=names, scafs=scafs, lens=lens, haploid=haploid, ploidy_argv=ploidy_argv + header_argv,
                n_sites=len(pos))


def window_argv(tmp, case, rng, tool, inp):
    overlap_flag = "--overlap" if tool in ("ABBABABAwindows.py", "fourPopWindows.py") else "-O"
    kinds = ["coordinate"] * 5 + ["sites"] * 3 + ["predefined"] * 2 + (["cat"] if tool == "distMat.py" else [])
    kind = pick(rng, kinds)
    argv = []
    if kind == "coordinate":
        w = int(rng.integers(30, 1200)) if not LONG["on"] else int(rng.integers(4300, 9000))
        argv += ["-w", str(w)]
        if rng.random() < 0.5:
            argv += ["-s", str(int(rng.integers(10, 2 * w)))]
        if rng.random() < 0.2:
            argv = ["--windType", "coordinate"] + argv
    elif kind == "sites":
        w = int(rng.integers(10, 300)) if not LONG["on"] else int(rng.integers(4200, 5500))
        argv += ["--windType", "sites", "-w", str(w)]
        if rng.random() < 0.5:
            argv += [overlap_flag, str(int(rng.integers(1, w)))]
        elif rng.random() < 0.6:                       # with an overlap a window cut short by -D may never advance (both loop / stop)
            argv += ["-D", str(int(rng.integers(w, 8 * w)))]
    elif kind == "predefined":
        cf = os.path.join(tmp, "c%d.coords" % case)
        with_id = rng.random() < 0.5
        rows = []
        for k in range(int(rng.integers(1, 12))):
            s = int(rng.integers(0, len(inp["scafs"])))
            a = int(rng.integers(1, inp["lens"][s] + 200))
            b = a + (int(rng.integers(0, 900)) if not LONG["on"] else int(rng.integers(4200, 9000)))
            name = inp["scafs"][s] if rng.random() > 0.1 else "absent"
            rows.append((name, a, b, "w%d" % k))
        if rng.random() < 0.5:
            order = {n: i for i, n in enumerate(inp["scafs"] + ["absent"])}
            rows.sort(key=lambda r: (order[r[0]], r[1]))
        with open(cf, "w") as f:
            for r in rows:
                f.write("\t".join(str(x) for x in (r if with_id else r[:3])) + "\n")
        argv += ["--windType", "predefined", "--windCoords", cf]
    else:
        argv += ["--windType", "cat"]
    if kind != "cat":
        argv += ["-m", str(int(pick(rng, [0, 1, 1, 3, 10, 25, 60])))] if not (kind == "predefined" and rng.random() < 0.3) else []
        if "-m" in argv and argv[argv.index("-m") + 1] == "0" and kind == "predefined":
            argv[argv.index("-m") + 1] = "1"                                   # -m 0 means "the window size": none is given here
    if rng.random() < 0.5:
        argv += ["--writeFailedWindows"]
    if rng.random() < 0.5 and kind != "cat":
        argv += ["--addWindowID"]
    if len(inp["scafs"]) > 1 and rng.random() < 0.25 and kind in ("coordinate", "sites"):
        lf = os.path.join(tmp, "c%d.scafs" % case)
        with open(lf, "w") as f:
            for s in inp["scafs"]:
                if rng.random() < 0.5:
                    f.write(s + "\n")
            f.write("other\n")
        # --include with sites windows: the reference never leaves its skip loop at the end of the file (genomics.py:2084-2087)
        argv += [pick(rng, ["--include", "--exclude"]) if kind == "coordinate" else "--exclude", lf]
    return argv


def pops(rng, names, n_pops, tmp, case, flags=None, min_size=1):
    """population arguments: -p NAME s1,s2 ... or --popsFile + bare names"""
    per = max(1, len(names) // n_pops)
    groups = [names[k * per:(k + 1) * per] for k in range(n_pops)]
    if rng.random() < 0.3:                                                       # unequal sizes, a sample left out
        groups = [g[:max(min_size, len(g) - int(rng.integers(0, 2)))] for g in groups]
    flags = flags or ["-p"] * n_pops
    pn = [pick(rng, ["pop%d", "P%d", "x%d"]) % k for k in range(n_pops)]
    if rng.random() < 0.25:
        pf = os.path.join(tmp, "c%d.pops" % case)
        with open(pf, "w") as f:
            for n, g in zip(pn, groups):
                for s in g:
                    f.write("%s\t%s\n" % (s, n))
            if len(names) > n_pops * per:
                f.write("%s\tunused\n" % names[-1])
        argv = ["--popsFile", pf]
        for fl, n in zip(flags, pn):
            argv += [fl, n]
        return argv
    argv = []
    for fl, n, g in zip(flags, pn, groups):
        argv += [fl, n, ",".join(g)]
    return argv


def make_case(tmp, case, rng, tools):
    tool = pick(rng, tools)
    inp = make_input(tmp, case, rng, tool)
    names = inp["names"]
    digits = 4
    if tool == "freq.py":
        argv = ["-g", inp["geno"], "-f", {"pairs": "alleles"}.get(inp["fmt"], inp["fmt"])]
        r = rng.random()
        if r < 0.5:
            argv += pops(rng, names, int(pick(rng, [2, 2, 3])), tmp, case)
        elif r < 0.7:
            argv += ["--indFreqs"]
        argv += ["--target", "derived"]                       # minor: the reference breaks ties with np.random.choice
        if rng.random() < 0.4:
            argv += ["--asCounts"]
        if rng.random() < 0.3:
            argv += ["--keepNanLines"]
        if rng.random() < 0.4:
            argv += ["--minData", str(pick(rng, [0.0, 0.3, 0.75, 1.0]))]
        if rng.random() < 0.25 and "--asCounts" not in argv:
            argv += ["--threshold", str(pick(rng, [0.2, 0.5]))]
        if rng.random() < 0.3:
            argv += ["-t", "2"]
        return tool, argv + inp["ploidy_argv"], digits, inp
    argv = ["-g", inp["geno"], "-f", inp["fmt"]] + window_argv(tmp, case, rng, tool, inp) + inp["ploidy_argv"]
    if tool == "popgenWindows.py":
        an = [a for a in ANALYSES if rng.random() < 0.4]
        if inp["haploid"] or inp["fmt"] == "haplo":
            an = [a for a in an if a != "indHet"]                    # sampleHet indexes a second haplotype: the worker dies
        if an:
            argv += ["--analysis"] + an
            if "hapStats" in an and rng.random() < 0.6:
                argv += ["--hapDist", str(pick(rng, [0.01, 0.05, 0.3]))]
        r = rng.random()
        if r < 0.8:
            n_pops = int(pick(rng, [1, 2, 2, 3, 4]))
            if "popFreq" in an and inp["fmt"] == "haplo":
                n_pops = min(n_pops, 2)                              # Tajima's D of a single haplotype divides by zero there
            argv += pops(rng, names, n_pops, tmp, case, min_size=2 if "popFreq" in an and (inp["haploid"] or inp["fmt"] == "haplo") else 1)
        if rng.random() < 0.25 and ("indPairDist" in an or "indHet" in an):
            k = int(rng.integers(2, len(names) + 1))
            argv += ["--samples", ",".join(str(x) for x in rng.choice(names, size=k, replace=False))]
        if rng.random() < 0.6:
            argv += ["--minData", str(pick(rng, [0.0, 0.01, 0.3, 0.8, 1.0]))]
        if rng.random() < 0.7:
            digits = int(pick(rng, [2, 3, 6, 8, 10]))
            argv += ["--roundTo", str(digits)]
        if rng.random() < 0.2:
            argv += ["-T", "2"]
    elif tool in ("ABBABABAwindows.py", "fourPopWindows.py"):
        argv += pops(rng, names, 4, tmp, case, flags=["-P1", "-P2", "-P3", "-O"])
        if rng.random() < 0.7:
            argv += ["--minData", str(pick(rng, [0.0, 0.01, 0.5, 1.0]))]
        if tool == "fourPopWindows.py":
            r = rng.random()
            if r < 0.3:
                argv += ["--polarize"]
            elif r < 0.6:
                argv += ["--fixed"]
            elif r < 0.7:
                argv += ["--polarize", "--fixed"]
    else:
        argv += ["--outFormat", pick(rng, ["raw", "phylip", "nexus"])]
        inp["named"] = False
        if rng.random() < 0.4:
            argv += ["--windowDataOutFile", "{out}.windows"]
        if rng.random() < 0.3:
            argv += ["-Mi", str(int(pick(rng, [1, 20, 200])))]
        if rng.random() < 0.4:
            argv += ["--includeSameWithSame"]
        if rng.random() < 0.3:
            k = int(rng.integers(2, len(names) + 1))
            argv += ["--samples"] + [str(x) for x in rng.choice(names, size=k, replace=False)]
        if rng.random() < 0.5:
            digits = int(pick(rng, [3, 6, 8]))
            argv += ["--roundTo", str(digits)]
    # the input on stdin (plain text; the reference needs the sample names on the command line then)
    named = ("-p" in argv or "-P1" in argv or "--samples" in argv or "--header" in argv or "--headers" in argv)
    if named and not inp["geno"].endswith(".gz") and rng.random() < 0.15:
        argv[argv.index("-g") + 1] = "<" + inp["geno"]
    return tool, argv, digits, inp


def split_stdin(argv):
    """`-g <path` (a marker of make_case) -> (argv without -g, path to pipe into stdin)"""
    if "-g" in argv and argv[argv.index("-g") + 1].startswith("<"):
        k = argv.index("-g")
        return argv[:k] + argv[k + 2:], argv[k + 1][1:]
    return argv, None


def run_case(case, tool, argv, digits, tmp, sizes, blocks):
    ref_out = os.path.join(tmp, "ref%d.out" % case)
    env = dict(os.environ, PYTHONHASHSEED="0")
    argv, piped = split_stdin(argv)
    # the reference in its own process group: a hang (a dead worker, a loop of the parent) is ended with all its workers
    pr = subprocess.Popen([sys.executable, "-c", WRAP, os.path.join(REF, tool)] + [a.format(out=ref_out) for a in argv] + ["-o", ref_out],
                          cwd=tmp, stdin=open(piped, "rb") if piped else subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          env=env, start_new_session=True)
    try:
        ref_err = pr.communicate(timeout=100)[1].decode()
        hung = False
    except subprocess.TimeoutExpired:
        os.killpg(pr.pid, signal.SIGKILL)
        ref_err = pr.communicate()[1].decode()
        hung = True
    ref_failed = hung or pr.returncode != 0
    tb = [ln for ln in ref_err.strip().splitlines() if "Error" in ln]
    ref_msg = ("hangs" + (": " + tb[-1] if tb else "")) if hung else (ref_err.strip().splitlines() or ["?"])[-1]
    ref_msg = ref_msg[:170]
    want = open(ref_out).read() if os.path.exists(ref_out) else ""
    want_w = open(ref_out + ".windows").read() if os.path.exists(ref_out + ".windows") else None
    verdicts, bad, notes = [], 0, []
    for size, block in zip(sizes, blocks):
        out = os.path.join(tmp, "got%d_%d.out" % (case, size))
        procs = []
        for rank in range(size):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT="29500",
                       PG_COMM="file", PG_RDZV_FILE=os.path.join(tmp, "rdzv_%d_%d" % (case, size)), PG_STREAM_BYTES=str(block))
            procs.append(subprocess.Popen([sys.executable, "-c", test_dist.CLI_WORKER, tool] + [a.format(out=out) for a in argv] + ["-o", out],
                                          env=env, stdin=open(piped, "rb") if piped else subprocess.DEVNULL, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE))
        errs = []
        for p in procs:
            try:
                errs.append(p.communicate(timeout=300)[1].decode())
            except subprocess.TimeoutExpired:
                p.kill()
                errs.append("TIMEOUT " + p.communicate()[1].decode())
        ours_failed = any(p.returncode != 0 for p in procs)
        if ref_failed:
            # the reference stopped (an assert, a crash of a worker = a hang): the driver must stop with an error too, or its output is
            # unchecked; reported, not counted
            verdicts.append("%d:%s" % (size, "both-stop" if ours_failed else "ref-only-stop"))
            if size == sizes[0]:
                notes.append("reference: " + ref_msg)
                if ours_failed:
                    notes.append("driver:    " + ([ln for e in errs for ln in e.strip().splitlines() if ln.strip()] or ["?"])[-1][:150])
            continue
        if ours_failed:
            verdicts.append("%d:FAILED" % size)
            bad += 1
            notes.append(([ln for e in errs for ln in e.strip().splitlines() if ln.strip()] or ["?"])[-1][:200])
            continue
        got = align_columns(open(out).read(), want)
        v = "ok"
        if got != want:
            try:
                n = G.compare_text(got, want, digits)
                v = "tie(%d)" % n
                if n > max(2, len(want.split()) // 50):
                    v, bad = "DIFF(%d cells)" % n, bad + 1
            except AssertionError as e:
                v, bad = "DIFF", bad + 1
                notes.append(str(e)[:200] + " | kept: %s %s" % (ref_out, out))
        if v in ("ok",) and want_w is not None and open(out + ".windows").read() != want_w:
            v, bad = "DIFF(windows file)", bad + 1
        verdicts.append("%d:%s" % (size, v))
    line = "case %3d  %-18s rows %4d  %-34s %s%s" % (case, tool, want.count("\n"), " ".join(verdicts),
                                                     " ".join(os.path.basename(a) if a.startswith(tmp) else a for a in argv[2 if not piped else 0:]),
                                                     "  < " + os.path.basename(piped) if piped else "")
    return bad, line, notes


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260926
    jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp(prefix="pg_fuzz_")
    tools = os.environ.get("PG_DIFF_TOOLS", "popgenWindows.py,popgenWindows.py,popgenWindows.py,distMat.py,distMat.py,"
                                            "ABBABABAwindows.py,fourPopWindows.py,freq.py").split(",")
    todo = []
    for case in range(n_cases):
        tool, argv, digits, inp = make_case(tmp, case, rng, tools)
        sizes = (1, int(pick(rng, [2, 3])))
        blocks = [int(pick(rng, [2000, 20000, 1 << 30])) for _ in sizes]
        todo.append((case, tool, argv, digits, tmp, sizes, blocks))
    bad = 0
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        for b, line, notes in ex.map(lambda a: run_case(*a), todo):
            bad += b
            print(line, flush=True)
            for n in notes:
                print("          ", n, flush=True)
    print("differences: %d of %d cases (seed %d)   files kept in %s" % (bad, n_cases, seed, tmp))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
