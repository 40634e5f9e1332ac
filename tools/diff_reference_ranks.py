#!/usr/bin/env python
"""Differential run against the UNMODIFIED reference (build container only: needs /root/reference): random `.geno` files x random
window parameters; the reference's single-process output against the drop-in drivers on 1, 2, 3 and 8 ranks (window-range shards of
the input, genomics_general_amd/shardplan.py; CPU stand-in engine with the oracle's numbers, ranks as processes with the file
communicator).  Everything must be byte-identical.     python tools/diff_reference_ranks.py [n_cases] [seed]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genomics_general_amd import synth                                          # noqa: E402
import test_dist                                                               # noqa: E402  (CLI_WORKER: the drivers on the stand-in engine)

REF = "/root/reference"
WRAP = ("import sys, runpy, numpy as np; np.NaN = np.nan; sys.path.insert(0, %r); "
        "sys.argv = sys.argv[1:]; runpy.run_path(sys.argv[0], run_name='__main__')" % REF)


def make_geno(path, rng):
    n_scaf = int(rng.choice([1, 1, 2, 4]))
    n_dip = int(rng.choice([4, 6, 8]))
    lens = [int(rng.integers(600, 4000)) for _ in range(n_scaf)]
    density = float(rng.choice([1.0, 0.5, 0.15]))
    sid, pos = [], []
    for k, ln in enumerate(lens):
        p = np.arange(1, ln + 1)
        p = p[rng.random(ln) < density]
        if rng.random() < 0.3 and len(p) > 50:                                  # a hole: empty windows
            a = int(rng.integers(0, len(p) // 2))
            p = np.concatenate([p[:a], p[a + len(p) // 4:]])
        sid.append(np.full(len(p), k))
        pos.append(p)
    sid, pos = np.concatenate(sid), np.concatenate(pos)
    codes = synth.gen_codes(int(rng.integers(1, 1 << 30)), sid, pos, n_dip, 2, var_thr=int(rng.choice([6554, 30000])),
                            miss_thr=int(rng.choice([1000, 6000, 20000])))
    names = ["s%d" % d for d in range(n_dip)]
    synth.write_geno(path, ["chr%d" % (k + 1) for k in range(n_scaf)], sid, pos, codes, names, sep="/", fmt="phased")
    return names, n_dip


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260926)
    tmp = tempfile.mkdtemp(prefix="pg_diff_")
    results = {"bad": 0}
    for case in range(n_cases):
        geno = os.path.join(tmp, "c%d.geno" % case)
        names, n_dip = make_geno(geno, rng)
        half = n_dip // 2
        tools = os.environ.get("PG_DIFF_TOOLS", "popgenWindows.py,popgenWindows.py,popgenWindows.py,distMat.py").split(",")
        tool = str(rng.choice(tools))
        argv = ["-g", geno, "-f", "phased"]
        if tool == "freq.py":
            argv += ["-p", "A", ",".join(names[:half]), "-p", "B", ",".join(names[half:])]
            if rng.random() < 0.5:
                argv += ["--target", "derived"]
                if rng.random() < 0.5:
                    argv += ["--asCounts"]
            ref_out = os.path.join(tmp, "ref%d.out" % case)
            run_case(case, tool, argv, ref_out, tmp, rng, results)
            continue
        if rng.random() < 0.7:
            w = int(rng.integers(50, 900))
            argv += ["-w", str(w)]
            if rng.random() < 0.6:
                argv += ["-s", str(int(rng.integers(20, 2 * w)))]
        else:
            w = int(rng.integers(20, 300))
            argv += ["--windType", "sites", "-w", str(w)]
            if rng.random() < 0.6:
                argv += ["-O", str(int(rng.integers(1, w)))]
        argv += ["-m", str(int(rng.integers(1, 30)))]
        if rng.random() < 0.6:
            argv += ["--writeFailedWindows"]
        if rng.random() < 0.6:
            argv += ["--addWindowID"]
        if tool == "popgenWindows.py":
            argv += ["-p", "A", ",".join(names[:half]), "-p", "B", ",".join(names[half:]), "--roundTo", "6"]
        elif tool in ("ABBABABAwindows.py", "fourPopWindows.py"):
            argv = [a if a != "-O" else "--overlap" for a in argv]
            q = max(n_dip // 4, 1)
            for flag, k in (("-P1", 0), ("-P2", 1), ("-P3", 2), ("-O", 3)):
                argv += [flag, "p%d" % k, ",".join(names[k * q:(k + 1) * q])]
            argv += ["--minData", str(float(rng.choice([0.01, 0.5])))]
        else:
            argv += ["--outFormat", "raw", "--windowDataOutFile", "{out}.windows"]
        ref_out = os.path.join(tmp, "ref%d.out" % case)
        run_case(case, tool, argv, ref_out, tmp, rng, results)
    print("differences: %d" % results["bad"])
    return 1 if results["bad"] else 0


def run_case(case, tool, argv, ref_out, tmp, rng, results):
    if True:
        r = subprocess.run([sys.executable, "-c", WRAP, os.path.join(REF, tool)] + [a.format(out=ref_out) for a in argv] + ["-o", ref_out],
                           cwd=tmp, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            print("case %d: the reference failed (%s): skipped" % (case, r.stderr.decode()[-200:].strip().splitlines()[-1:]))
            return
        want = open(ref_out).read()
        want_w = open(ref_out + ".windows").read() if os.path.exists(ref_out + ".windows") else None
        verdicts = []
        for size in (1, 2, 3, 8):
            out = os.path.join(tmp, "got%d_%d.out" % (case, size))
            procs = []
            for rank in range(size):
                env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                           MASTER_PORT="29500", PG_COMM="file", PG_RDZV_FILE=os.path.join(tmp, "rdzv_%d_%d" % (case, size)),
                           PG_STREAM_BYTES=str(int(rng.choice([3000, 20000, 1 << 30]))), PG_TIMING="1")
                procs.append(subprocess.Popen([sys.executable, "-c", test_dist.CLI_WORKER, tool] + [a.format(out=out) for a in argv] + ["-o", out],
                                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
            errs = [p.communicate(timeout=600)[1].decode() for p in procs]
            ok = all(p.returncode == 0 for p in procs) and open(out).read() == want
            if ok and want_w is not None:
                ok = open(out + ".windows").read() == want_w
            ranges = sum('"window_ranges": true' in e for e in errs)
            verdicts.append("%d:%s%s" % (size, "ok" if ok else "DIFF", "(ranges)" if ranges == size and size > 1 else ""))
            results["bad"] += 0 if ok else 1
            if not ok:
                print("   rank errors:", [e[-300:] for e in errs if "Traceback" in e][:1], "| kept:", ref_out, out)
        print("case %2d  %-18s %-60s rows %3d  %s" % (case, tool, " ".join(argv[4:12]), want.count("\n"), " ".join(verdicts)), flush=True)


if __name__ == "__main__":
    sys.exit(main())
