"""-m gpu: a mid-size random input (three scaffolds x 30 000 sites x 32 diploids, 10 % variable sites, 3 % missing calls; 13 MB of
`.geno` text) through the drop-in command lines, against the ORACLE's command lines on the same file: the goldens pin the
reference's behaviour on small fixtures, this pins the whole chain -- device tokenizer (several blocks, carried rows), host
tokenizer pipeline, packed input, windows across scaffolds, kernels, finalisers, formatting -- at a size where blocks, groups
and tiles are no longer single."""
import os

import numpy as np
import pytest

import oracle_cli
from golden_util import align_columns
from genomics_general_amd import genoio, synth

import test_gpu_golden as G

pytestmark = pytest.mark.gpu
N_DIP, N_POPS, PER_SCAF, N_SCAF = 32, 4, 30000, 3
NAMES = ["s%d" % d for d in range(N_DIP)]


@pytest.fixture(scope="module")
def geno(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("e2e") / "random.geno")
    with open(path, "wb") as f:
        for k in range(N_SCAF):
            sid = np.full(PER_SCAF, k, dtype=np.int64)
            codes = synth.gen_codes(4242, sid, np.arange(1, PER_SCAF + 1), N_DIP, N_POPS, var_thr=6500, miss_thr=2000)
            part = path + ".part"
            synth.write_geno_fast(part, codes, NAMES, "scaf%d" % (k + 1), 1)
            with open(part, "rb") as g:
                if k:
                    g.readline()
                f.write(g.read())
            os.remove(part)
    return path


def popgen_argv(path):
    per = N_DIP // N_POPS
    argv = ["-g", path, "-f", "phased", "-w", "4000", "-s", "3000", "-m", "40", "--roundTo", "6", "--analysis", "popDist", "popPairDist",
            "indHet", "popFreq"]
    for k in range(N_POPS):
        argv += ["-p", "p%d" % k, ",".join(NAMES[k * per:(k + 1) * per])]
    return argv


def abba_argv(path):
    return ["-g", path, "-f", "phased", "--windType", "sites", "-w", "2500", "--overlap", "500", "-m", "100", "--minData", "0.5",
            "-P1", "a", ",".join(NAMES[0:8]), "-P2", "b", ",".join(NAMES[8:16]), "-P3", "c", ",".join(NAMES[16:24]),
            "-O", "o", ",".join(NAMES[24:32])]


@pytest.fixture(scope="module")
def want(geno):
    return {"popgenWindows.py": oracle_cli.run("popgenWindows.py", popgen_argv(geno)),
            "ABBABABAwindows.py": oracle_cli.run("ABBABABAwindows.py", abba_argv(geno))}


@pytest.mark.parametrize("tool", ["popgenWindows.py", "ABBABABAwindows.py"])
@pytest.mark.parametrize("mode", ["device_tokenizer", "device_tokenizer_one_block", "host_tokenizer", "pgeno", "pgeno_raw"])
def test_drivers_match_the_oracle_on_a_random_mid_size_input(tool, mode, geno, want, tmp_path, monkeypatch):
    path = geno
    if mode.startswith("pgeno"):
        path = str(tmp_path / "random.pgeno")
        genoio.pack_geno(geno, path, "phased", block_bytes=1 << 20, codec="none" if mode == "pgeno_raw" else "zlib")
    if mode != "device_tokenizer_one_block":
        monkeypatch.setenv("PG_STREAM_BYTES", str(3 << 20))                     # several blocks, rows carried across them
    if mode == "host_tokenizer":
        monkeypatch.setenv("PG_GPU_TOKENIZER", "0")
    argv = (popgen_argv if tool == "popgenWindows.py" else abba_argv)(path)
    out = str(tmp_path / "out.csv")
    G.MAINS[tool](argv + ["-o", out])
    with open(out) as f:
        got = f.read()
    w = want[tool]
    assert len(w.splitlines()) > 20
    assert G.compare_text(align_columns(got, w), w, 6 if tool == "popgenWindows.py" else 4) == 0       # (asserts on any difference)


@pytest.mark.parametrize("tool", ["popgenWindows.py", "ABBABABAwindows.py"])
def test_long_windows_print_the_reference_digits(tool, geno, tmp_path, monkeypatch, capfd):
    """windows of 7500 / 10 000 sites: their float64 sums are formed with fixed reduction trees, and a window whose value is within
    reach of a rounding tie of the printed digit is computed again in NumPy's order (cli._refine_long_windows).  With --roundTo 12 every
    value is within reach (the trees agree to ~1e-13 only), so every window takes that second pass and the text must be the oracle's,
    cell for cell; ABBABABAwindows.py (4 digits) recomputes next to nothing and must be the oracle's as well"""
    per = N_DIP // N_POPS
    if tool == "popgenWindows.py":
        argv = ["-g", geno, "-f", "phased", "-w", "7500", "-m", "40", "--roundTo", "12", "--analysis", "popDist", "popPairDist"]
        for k in range(N_POPS):
            argv += ["-p", "p%d" % k, ",".join(NAMES[k * per:(k + 1) * per])]
    else:
        argv = ["-g", geno, "-f", "phased", "-w", "10000", "-m", "100", "--minData", "0.5",
                "-P1", "a", ",".join(NAMES[0:8]), "-P2", "b", ",".join(NAMES[8:16]), "-P3", "c", ",".join(NAMES[16:24]),
                "-O", "o", ",".join(NAMES[24:32])]
    want = oracle_cli.run(tool, argv)
    monkeypatch.setenv("PG_TIMING", "1")
    out = str(tmp_path / "out.csv")
    G.MAINS[tool](argv + ["-o", out])
    with open(out) as f:
        got = f.read()
    assert len(want.splitlines()) >= 10
    assert got == want
    timing = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("PG_TIMING ")]
    if tool == "popgenWindows.py":
        import json
        assert json.loads(timing[-1][len("PG_TIMING "):]).get("windows_recomputed_in_numpy_order", 0) >= 10


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two ranks (both on device 0 here, so the exchange goes through files: RCCL
    refuses duplicate devices), rank 0's JSON line on stdout, the result all-gather inside the reported time"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PG_COMM"] = "file"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-tiers"], env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["comm_ranks"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["config"]["name"] == "tiny" and "result_allgather_ms_per_step" in d
    assert d["config"]["windows_per_gpu"] * 2 * 3 / (d["ms_per_step"] * 3 / 1e3) == pytest.approx(d["value"], rel=1e-3)
    # 8 ranks: the headline keeps its shape, the rank's share of configs[4] is measured behind it (here a small total, so that eight
    # ranks fit the one GPU of the box); a share that cannot be set up is an `error` entry, never a hang
    env8 = dict(env, PG_BENCH_C5_SITES="2400000")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-tiers"], env=env8, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["config"]["name"] == "tiny" and d["scaling"] == "weak"
    assert d["c5_share"].get("windows_per_gpu") == 6 and d["c5_share"]["windows_per_sec"] > 0, d["c5_share"]
    # --strong: ONE data set, cut into window ranges by the drivers' plan (shardplan); both ranks must get work, together all of it
    for n in (2, 3):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--workload", "tiny", "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline", "--no-tiers", "--strong"], env=env, capture_output=True, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
        assert d["scaling"] == "strong" and d["n_gpus"] == n
        assert all(w > 0 for w in d["strong"]["windows_per_rank"]) and sum(d["strong"]["windows_per_rank"]) == 8      # 400 000 sites / 50 kb
        assert sum(d["strong"]["rank_bytes_share"]) == pytest.approx(1.0, abs=1e-3) and max(d["strong"]["rank_bytes_share"]) <= 1.0 / n + 0.13
        assert 8 * 2 / (d["ms_per_step"] * 2 / 1e3) == pytest.approx(d["value"], rel=1e-3)


@pytest.mark.parametrize("name,tool,size", [("holes_distmat_cat_nexus", "distMat.py", 2), ("holes_distmat_cat_nexus", "distMat.py", 3),
                                            ("sparse_predefined", "popgenWindows.py", 2), ("sparse_predefined", "popgenWindows.py", 3),
                                            # window ranges inside scaffold runs (shardplan): one scaffold and four, 2 / 3 / 8 ranks
                                            ("one_popgen_overlap_failed_id", "popgenWindows.py", 8), ("one_popgen_sites", "popgenWindows.py", 3),
                                            ("one_distmat_windows_id", "distMat.py", 2), ("four_popgen_id", "popgenWindows.py", 8),
                                            ("four_abba_overlap", "ABBABABAwindows.py", 3), ("four_fourpop", "fourPopWindows.py", 2)])
def test_cat_and_predefined_windows_on_several_ranks(name, tool, size, tmp_path):
    """WORLD_SIZE ranks, all on device 0 (so the exchange goes through files).  Coordinate and sites windows: every rank reads,
    tokenises (on the device) and computes its window range of the one- or four-scaffold file, one gather of the rows.  `distMat.py --windType cat`: every rank counts its
    share of the lines on the GPU (pg_pairwise), the counts are summed across the ranks, the matrix finished from the sums
    (pg_indpairdist_mean_from_counts) is the reference's.  `--windType predefined`: the file is cut at the scaffold runs the plan
    (windows.plan_predefined_shards) allows, every rank streams its own windows, one gather of the rows"""
    import gzip
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    sys.path.insert(0, gold)
    from cases import CASES
    case = [c for c in CASES if c["name"] == name][0]
    geno = str(tmp_path / (case["fixture"] + ".geno"))
    with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
        g.write(f.read())
    out = str(tmp_path / "ranks.out")
    argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
    procs = []
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(36000 + (os.getpid() + size) % 2000), PG_COMM="file", PG_COMM_TIMEOUT="90", PG_TIMING="1",
                   PG_RDZV_FILE=str(tmp_path / "rdzv"))         # (ranks started by hand: a rendezvous name of their own)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, tool)] + argv, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    errs = [p.communicate(timeout=600)[1].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join("--- rank %d (rc %s)\n%s" % (r, p.returncode, e[-1500:])
                                                            for r, (p, e) in enumerate(zip(procs, errs)))
    assert sum('"sharded_input": true' in ln for e in errs for ln in e.splitlines() if ln.startswith("PG_TIMING ")) == size
    with open(out) as f, open(os.path.join(gold, case["name"] + ".out")) as g:
        got, want = f.read(), g.read()
    G.compare_text(align_columns(got, want), want, G.round_digits(case))
    if os.path.exists(os.path.join(gold, name + ".out.windows")) and "windowDataOutFile" in " ".join(case["argv"]):
        with open(out + ".windows") as f, open(os.path.join(gold, name + ".out.windows")) as g:
            assert f.read() == g.read()


def test_infer_ploidy_follows_a_cell_of_another_width_on_the_device(tmp_path, monkeypatch):
    """--inferPloidy on a file in which ONE cell far behind the first block has another width (a diploid sample's cell written with
    one allele): the reference infers the ploidy per window (genomics.py:1108-1111), so that window sees the sample as haploid.  The
    engine reads the cell widths of the whole input first (pg_text_cell_widths) and computes every window under its own ploidies
    (cli.MultiLayoutBatch: pg_set_samples per group of windows).  Expected: the oracle's command line, which the ploidyshift goldens
    pin to the reference.  (Until round 5 the drivers refused such a file.)"""
    import gzip
    import sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    from cases import CASES
    from golden_util import align_columns
    import oracle_cli
    case = [c for c in CASES if c["name"] == "mixed_inferploidy"][0]
    lines = gzip.open(os.path.join(gold, "mixed.geno.gz"), "rb").read().split(b"\n")
    cells = lines[1500].split(b"\t")
    cells[2] = cells[2][:1]
    odd = str(tmp_path / "odd.geno")
    with open(odd, "wb") as f:
        f.write(b"\n".join(lines[:1500] + [b"\t".join(cells)] + lines[1501:]))
    monkeypatch.setenv("PG_STREAM_BYTES", "20000")
    out = str(tmp_path / "o.csv")
    argv = [a.format(geno=odd, dir=gold, out=out) for a in case["argv"]]
    G.MAINS[case["tool"]](argv + ["-o", out])
    want = oracle_cli.run(case["tool"], argv)
    with open(out) as f:
        got = f.read()
    G.compare_text(align_columns(got, want), want, G.round_digits(case))
    with open(os.path.join(gold, case["name"] + ".out")) as f:
        assert f.read() != want


def test_a_failing_rank_ends_the_whole_launch_on_the_device(tmp_path):
    """8 ranks on device 0 (PG_COMM=file), the HIP engine: a position that is not a number in rank 5's share of the text makes that
    rank's tokenizer raise; every process is gone within seconds with a non-zero exit code, the error spelled out once (VERDICT
    round 4: one rank used to sit in the exchange until PG_COMM_TIMEOUT)"""
    import gzip
    import subprocess
    import sys
    import time
    from test_dist import _corrupt_one_share
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    sys.path.insert(0, gold)
    from cases import CASES
    case = [c for c in CASES if c["name"] == "one_popgen_overlap_failed_id"][0]
    geno = str(tmp_path / "one.geno")
    with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
        g.write(f.read())
    size, bad_rank = 8, 5
    bad = _corrupt_one_share(geno, size, bad_rank, tmp_path)
    out = str(tmp_path / "never.out")
    argv = [a.format(geno=bad, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
    procs, t0 = [], time.time()
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(39000 + os.getpid() % 1500), PG_COMM="file", PG_COMM_TIMEOUT="120", PG_RDZV_FILE=str(tmp_path / "rdzv"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "popgenWindows.py")] + argv, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    errs = []
    for p in procs:
        try:
            _, e = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            p.kill()
            _, e = p.communicate()
            e += b"\n[killed by the test after 90 s]"
        errs.append(e.decode())
    took = time.time() - t0
    assert all(p.returncode not in (0, None) for p in procs), [p.returncode for p in procs]
    assert not any("killed by the test" in e for e in errs), "a rank hung: " + " | ".join(e[-200:] for e in errs)
    assert took < 45, "the launch took %.0f s to end" % took                      # (eight device contexts on one GPU start one after the other)
    assert "Traceback" in errs[bad_rank] and sum("Traceback" in e for e in errs) == 1, [e[-300:] for e in errs]
    assert sum("rank %d failed" % bad_rank in e for e in errs) >= size - 2
