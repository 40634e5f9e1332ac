"""Multi-GPU sharding / gather logic on CPU: world_size 2 with the gloo stand-in communicator, plus shard arithmetic."""
import os
import subprocess
import sys

import numpy as np
import pytest

from genomics_general_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from genomics_general_amd import dist
world = dist.world_from_env()
comm = dist.GlooComm(world)
n_total, k = 11, 3
lo, hi = dist.shard_range(n_total, world.size, world.rank)
full = (np.arange(n_total * k, dtype=np.float64).reshape(n_total, k) * 1.5) - 7
full[4, 1] = np.nan
got = dist.gather_table(comm, full[lo:hi], n_total)
assert got.shape == full.shape
assert np.array_equal(np.isnan(got), np.isnan(full)) and np.allclose(np.nan_to_num(got), np.nan_to_num(full))
t = comm.allgather(np.array([float(world.rank + 1)]))
assert t.ravel().tolist() == [1.0, 2.0]
# empty shard on one rank
lo1, hi1 = dist.shard_range(1, world.size, world.rank)
one = dist.gather_table(comm, np.full((hi1 - lo1, 2), 5.0), 1)
assert one.tolist() == [[5.0, 5.0]]
comm.barrier(); comm.close()
print("rank", world.rank, "ok")
''' % ROOT


def test_shard_arithmetic():
    for n in (0, 1, 7, 200, 60001):
        for size in (1, 2, 3, 8):
            rng = [dist.shard_range(n, size, r) for r in range(size)]
            assert rng[0][0] == 0 and rng[-1][1] == n
            assert all(rng[r][1] == rng[r + 1][0] for r in range(size - 1))
            counts = dist.shard_counts(n, size)
            assert sum(counts) == n and max(counts) - min(counts) <= 1


def test_gather_table_world_size_2_gloo(tmp_path):
    port = 29000 + os.getpid() % 2000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_unique_id_file_rendezvous(tmp_path, monkeypatch):
    monkeypatch.setenv("PG_RDZV_FILE", str(tmp_path / "rdzv"))
    uid0, path = dist.exchange_unique_id(dist.World(0, 2, 0), lambda: bytes(range(128)))
    uid1, _ = dist.exchange_unique_id(dist.World(1, 2, 1), lambda: b"", timeout_s=5)
    assert uid0 == uid1 == bytes(range(128)) and os.path.exists(path)


def test_rendezvous_path_and_device_folding(monkeypatch):
    """the RCCL unique-id hand-over file: keyed by the launcher under torch.distributed.run, by MASTER_ADDR/PORT otherwise,
    PG_RDZV_FILE wins; LOCAL_RANK folds into the visible devices"""
    import os
    from genomics_general_amd import _lib, dist
    for k in ("PG_RDZV_FILE", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29999")
    plain = dist._rdzv_path()
    assert plain == "/tmp/pg_rdzv_127.0.0.1_29999"
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    assert dist._rdzv_path() == plain + "_none_%d" % os.getppid()
    monkeypatch.setenv("PG_RDZV_FILE", "/tmp/x_y")
    assert dist._rdzv_path() == "/tmp/x_y"
    monkeypatch.setattr(_lib, "device_count", lambda: 1)
    assert dist.device_for(dist.World(3, 8, 3)) == 0
    monkeypatch.setattr(_lib, "device_count", lambda: 8)
    assert dist.device_for(dist.World(3, 8, 3)) == 3
    monkeypatch.setattr(_lib, "device_count", lambda: 0)
    assert dist.device_for(dist.World(3, 8, 3)) == 3


def test_unique_id_handover_between_processes(tmp_path, monkeypatch):
    """rank 0 publishes, another process picks the 128 bytes up; a stale file of a dead launch is replaced, not read"""
    import multiprocessing as mp
    import os
    from genomics_general_amd import dist
    path = str(tmp_path / "rdzv")
    monkeypatch.setenv("PG_RDZV_FILE", path)
    with open(path, "wb") as f:
        f.write(b"\\x01" * 128)                                            # leftover
    uid = bytes(range(128))
    got0, p0 = dist.exchange_unique_id(dist.World(0, 2, 0), lambda: uid)
    assert got0 == uid and p0 == path
    ctx = mp.get_context("fork")
    q = ctx.Queue()

    def other():
        q.put(dist.exchange_unique_id(dist.World(1, 2, 1), None, timeout_s=20.0)[0])

    pr = ctx.Process(target=other)
    pr.start()
    assert q.get(timeout=30) == uid
    pr.join(10)
    os.remove(path)


CLI_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "tests", "golden"))
from genomics_general_amd import cli, dist
from cpu_engine import CpuEngine
cli.Engine = CpuEngine                                        # numbers from the oracle: this test is about sharding and gathering
dist.RcclComm = lambda engine, world: dist.GlooComm(world)
tool, argv = sys.argv[1], sys.argv[2:]
rc = {"popgenWindows.py": cli.popgen_main, "ABBABABAwindows.py": cli.abbababa_main, "distMat.py": cli.distmat_main,
      "freq.py": cli.freq_main, "fourPopWindows.py": cli.fourpop_main}[tool](argv)
sys.exit(rc or 0)
''' % (ROOT, ROOT, ROOT)


def _run_two_ranks(tool, argv, port):
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PG_STREAM_BYTES="6000")
        procs.append(subprocess.Popen([sys.executable, "-c", CLI_WORKER, tool] + argv, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        assert p.returncode == 0, o.decode()[-1500:]


def test_drivers_with_two_ranks_write_the_single_rank_output(tmp_path):
    """popgenWindows / ABBABABAwindows / distMat with WORLD_SIZE=2 (gloo stand-in communicator, CPU stand-in engine), input
    streamed in 6 kB blocks: every block's windows are split over the ranks, gathered, and rank 0 writes the reference's file"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES
    from golden_util import align_columns
    import test_gpu_golden as G
    gold = os.path.join(ROOT, "tests", "golden")
    # (ploidyshift_*: --inferPloidy on a file whose ploidy changes -- every rank reads the whole input, the windows are split)
    for k, name in enumerate(("sparse_overlap_failed_id", "abba_windows_sites", "multi_distmat", "sparse_predefined",
                              "ploidyshift_popgen_sliding_ind", "ploidyshift_fourpop")):
        case = [c for c in CASES if c["name"] == name][0]
        out = str(tmp_path / (name + ".out"))
        geno = os.path.join(gold, case["fixture"] + ".geno.gz")
        argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
        _run_two_ranks(case["tool"], argv, 31000 + (os.getpid() + 7 * k) % 2000)
        with open(out) as f, open(os.path.join(gold, name + ".out")) as g:
            got, want = f.read(), g.read()
        G.compare_text(align_columns(got, want), want, G.round_digits(case))


def test_two_ranks_shard_the_input_at_scaffold_runs_and_gather_rows_once(tmp_path):
    """Plain-text input on disk: each rank reads, tokenises and computes only its own run-aligned slice (cf. the slice-parallel
    freq.py:23-28 of the reference), formats its own rows, and ONE gather at the end lets rank 0 write the reference's file --
    window IDs, failed windows and the stderr totals included.  On the two-scaffold fixture each rank must have consumed at most
    60 % of the input bytes."""
    import gzip
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES
    from golden_util import align_columns
    import test_gpu_golden as G
    gold = os.path.join(ROOT, "tests", "golden")
    from genomics_general_amd import genoio
    from test_host import _bgzf_write
    for k, (name, packed) in enumerate((("c1_popgen", False), ("sparse_overlap_failed_id", False), ("abba_windows_sites", False),
                                        ("sparse_stepgap", False), ("c1_popgen", True), ("sparse_overlap_failed_id", True),
                                        ("c1_popgen", "bgzf"), ("sparse_overlap_failed_id", "bgzf"),
                                        # distMat.py: matrices and the window side file (IDs shifted per rank) meet in one gather each
                                        ("multi_distmat", False), ("multi_distmat_windows_id", False),
                                        ("multi_distmat_windows_id", "bgzf"))):
        case = [c for c in CASES if c["name"] == name][0]
        geno = str(tmp_path / (case["fixture"] + ".geno"))
        with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
            g.write(f.read())
        if packed == "bgzf":                           # bgzip-compressed text: cuts are (member, offset in member) pairs
            with open(geno, "rb") as f:
                _bgzf_write(geno + ".gz", f.read(), blk=20000)
            geno = geno + ".gz"
        elif packed:                                   # the same from a packed file: row ranges from the block headers
            fmt = case["argv"][case["argv"].index("-f") + 1]
            genoio.pack_geno(geno, geno[:-5] + ".pgeno", fmt, block_bytes=30000, codec="none" if k % 2 else "zlib")
            geno = geno[:-5] + ".pgeno"
        out = str(tmp_path / (name + ".out"))
        argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
        procs = []
        port = 33000 + (os.getpid() + 11 * k) % 2000
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       PG_STREAM_BYTES="20000", PG_TIMING="1")
            procs.append(subprocess.Popen([sys.executable, "-c", CLI_WORKER, case["tool"]] + argv, env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE))
        timing = []
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, e.decode()[-1500:]
            timing += [json.loads(ln[len("PG_TIMING "):]) for ln in e.decode().splitlines() if ln.startswith("PG_TIMING ")]
        with open(out) as f, open(os.path.join(gold, name + ".out")) as g:
            got, want = f.read(), g.read()
        G.compare_text(align_columns(got, want), want, G.round_digits(case))
        if os.path.exists(os.path.join(gold, name + ".out.windows")):
            with open(out + ".windows") as f, open(os.path.join(gold, name + ".out.windows")) as g:
                assert f.read() == g.read()
        assert len(timing) == 2 and all(t["sharded_input"] for t in timing), timing
        if name == "c1_popgen" and packed != "bgzf":
            for t in timing:
                assert t["text_bytes"] <= (0.65 if packed else 0.6) * t["input_bytes"], timing
        if packed == "bgzf":                           # (text_bytes counts inflated bytes there) both ranks worked on sites
            assert all(t["sites"] > 0 for t in timing), timing
        n_lines = sum(1 for ln in gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"))) - 1
        overlapping = "-s" in case["argv"] or "-O" in case["argv"] or "--overlap" in case["argv"]
        # (window ranges: with -s < -w, or -O, the rows of the windows that straddle a cut are read by both neighbours; with
        # -s > -w the rows between the last window of one rank and the first of the next are read by nobody)
        got_rows = sum(t["sites"] for t in timing)
        assert (0.9 * n_lines <= got_rows <= 1.25 * n_lines) if overlapping else got_rows == n_lines


def test_freq_with_two_ranks_cuts_the_input_at_line_boundaries(tmp_path):
    """freq.py with WORLD_SIZE=2: plain text is cut at the line boundary nearest to the middle (sites are independent), each rank
    writes its rows, one gather, rank 0 writes the reference's file; gzipped input cannot be cut: rank 0 does the whole job"""
    import gzip
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES
    gold = os.path.join(ROOT, "tests", "golden")
    from genomics_general_amd import genoio
    from test_host import _bgzf_write
    for k, (name, plain) in enumerate((("abba_freq_derived", True), ("abba_freq_indfreqs", True), ("abba_freq_counts", True),
                                       ("abba_freq_derived", False), ("abba_freq_counts", "bgzf"), ("abba_freq_derived", "pgeno"))):
        case = [c for c in CASES if c["name"] == name]
        if not case:
            continue
        case = case[0]
        geno = os.path.join(gold, case["fixture"] + ".geno.gz")
        if plain:
            geno = str(tmp_path / (case["fixture"] + ".geno"))
            with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
                g.write(f.read())
        if plain == "bgzf":                                # bgzip-compressed text and packed rows are cut at line / row boundaries too
            with open(geno, "rb") as f:
                _bgzf_write(geno + ".gz", f.read(), blk=15000)
            geno = geno + ".gz"
        elif plain == "pgeno":
            genoio.pack_geno(geno, geno[:-5] + ".pgeno", "phased", block_bytes=30000, codec="none")
            geno = geno[:-5] + ".pgeno"
        out = str(tmp_path / (name + ".out"))
        argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
        _run_two_ranks(case["tool"], argv, 35000 + (os.getpid() + 13 * k) % 2000)
        with open(out) as f, open(os.path.join(gold, name + ".out")) as g:
            assert f.read() == g.read(), name


def test_line_shards_of_bgzf_and_packed_input_cover_the_input_once(tmp_path):
    """shard_lines on bgzip-compressed text (cuts = member + offset inside, no exchange) and on `.pgeno` rows (freq.py and
    `distMat.py --windType cat` on several ranks): the ranks' shares are disjoint and cover every row, for any number of ranks"""
    from genomics_general_amd import genoio
    from test_host import _bgzf_write
    lines = [("chr%d\t%d\t" % (1 + i // 40, i + 1) + "\t".join("A/C" if (i + j) % 3 else "N/N" for j in range(3))).encode() for i in range(131)]
    text = b"#CHROM\tPOS\ts1\ts2\ts3\n" + b"\n".join(lines) + b"\n"
    plain = str(tmp_path / "x.geno")
    with open(plain, "wb") as f:
        f.write(text)
    for blk in (300, 1000, 100000):
        bz = str(tmp_path / ("x%d.geno.gz" % blk))
        _bgzf_write(bz, text, blk=blk)
        for size in (1, 2, 3, 8, 50):
            got = []
            for rank in range(size):
                r = genoio.open_input(bz)
                r.read_header()
                assert r.shard_lines(dist.World(rank, size, rank))
                body = bytes(r.read_block(None))
                assert body == b"" or body.endswith(b"\n")
                got.append(body)
                r.close()
            assert b"".join(got) == b"\n".join(lines) + b"\n", (blk, size)
    for codec in ("none", "zlib"):
        pk = str(tmp_path / ("x_%s.pgeno" % codec))
        genoio.pack_geno(plain, pk, "phased", block_bytes=900, codec=codec)
        for size in (1, 2, 3, 8, 200):
            rows = []
            for rank in range(size):
                r = genoio.open_input(pk)
                assert r.shard_lines(dist.World(rank, size, rank))
                for b in r.read_block(None):
                    rows += [int(p) for p in b.positions()]
                r.close()
            assert rows == list(range(1, 132)), (codec, size)


def test_line_shards_cover_the_input_once():
    """BlockReader.shard_lines: the ranks' byte ranges are disjoint, start at line starts and cover every data line, for any
    number of ranks (more ranks than lines included)"""
    import tempfile
    from genomics_general_amd import genoio
    lines = [("chr1\t%d\t" % (i + 1) + "\t".join("A/C" if (i + j) % 3 else "N/N" for j in range(1 + i % 5))).encode() for i in range(57)]
    text = b"#CHROM\tPOS\ts1\n" + b"\n".join(lines) + b"\n"
    with tempfile.NamedTemporaryFile(suffix=".geno", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        for size in (1, 2, 3, 8, 64, 100):
            got = []
            for rank in range(size):
                r = genoio.open_input(path)
                r.read_header()
                assert r.shard_lines(dist.World(rank, size, rank))
                body = bytes(r.read_block(None))
                assert body == b"" or body.endswith(b"\n")
                got.append(body)
                r.close()
            assert b"".join(got) == b"\n".join(lines) + b"\n", size
    finally:
        os.remove(path)


FILE_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from genomics_general_amd import dist
world = dist.world_from_env()
comm = dist.make_comm(None, world)
assert isinstance(comm, dist.FileComm)
n_total, k = 13, 2
lo, hi = dist.shard_range(n_total, world.size, world.rank)
full = np.arange(n_total * k, dtype=np.float64).reshape(n_total, k) - 3.25
full[2, 0] = np.nan
for it in range(20):                                   # many exchanges: files of finished exchanges are removed on the way
    got = dist.gather_table(comm, full[lo:hi] + it, n_total)
    assert np.array_equal(np.isnan(got), np.isnan(full)) and np.allclose(np.nan_to_num(got), np.nan_to_num(full + it))
parts = dist.gather_bytes(comm, b"rank%%d" %% world.rank * (world.rank + 1))
assert parts == [b"rank0", b"rank1rank1", b"rank2rank2rank2"][:world.size]
comm.barrier(); comm.close()
print("rank", world.rank, "ok", len(os.listdir(os.path.dirname(comm.dir))) >= 0)
''' % ROOT


def test_file_comm_world_size_3(tmp_path):
    """PG_COMM=file (ranks that share one device: bench.py --gpus N and the drivers on a single-GPU box): all-gathers through
    renamed files, same interface and results as the RCCL / gloo communicators"""
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3", PG_COMM="file",
                   PG_RDZV_FILE=str(tmp_path / "rdzv"))
        procs.append(subprocess.Popen([sys.executable, "-c", FILE_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        assert p.returncode == 0 and ("rank %d ok" % rank) in o.decode(), o.decode()[-2000:]
    assert os.listdir(str(tmp_path)) == [], "the exchange directory and the pointer to it are removed by rank 0's close()"


def test_a_refused_rccl_communicator_falls_back_to_files(monkeypatch, tmp_path, capsys):
    """RCCL refusing to initialise (no peer access, two ranks on one device, ...) must not lose a run whose only exchange is the
    gather of finished rows: make_comm says so on stderr and hands out the file communicator; a rank that never shows up
    (TimeoutError of the unique-id hand-over) stays an error"""
    monkeypatch.delenv("PG_COMM", raising=False)
    monkeypatch.setenv("PG_RDZV_FILE", str(tmp_path / "rdzv"))

    def refuse(engine, world):
        raise RuntimeError("ncclCommInitRank: invalid usage")
    monkeypatch.setattr(dist, "RcclComm", refuse)
    import threading
    comms = [None, None]

    def make(r):
        comms[r] = dist.make_comm(None, dist.World(r, 2, r))
    th = [threading.Thread(target=make, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert all(isinstance(c, dist.FileComm) and c.size == 2 for c in comms) and comms[0].dir == comms[1].dir
    assert "travel through files" in capsys.readouterr().err

    def absent(engine, world):
        raise TimeoutError("rank 1: no RCCL unique id")
    monkeypatch.setattr(dist, "RcclComm", absent)
    with pytest.raises(TimeoutError):
        dist.make_comm(None, dist.World(1, 2, 1))


def test_cat_window_counts_are_summed_over_the_ranks(tmp_path):
    """`distMat.py --windType cat` (one window = every site of the input): every rank counts a share of the LINES, the pair counts
    (additive over sites) are summed across the ranks, the matrix is the reference's.  Two and three ranks; with --minPerInd
    the per-haplotype called counts are summed the same way"""
    import gzip
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES
    from golden_util import align_columns
    import test_gpu_golden as G
    gold = os.path.join(ROOT, "tests", "golden")
    case = [c for c in CASES if c["name"] == "holes_distmat_cat_nexus"][0]
    geno = str(tmp_path / (case["fixture"] + ".geno"))
    with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
        g.write(f.read())
    n_lines = sum(1 for _ in open(geno)) - 1
    for k, (size, extra) in enumerate(((2, []), (3, []), (2, ["--minPerInd", "1"]))):
        out = str(tmp_path / ("cat%d.out" % k))
        argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + extra + ["-o", out]
        port = 35000 + (os.getpid() + 13 * k) % 2000
        procs = []
        for rank in range(size):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), PG_TIMING="1")
            procs.append(subprocess.Popen([sys.executable, "-c", CLI_WORKER, case["tool"]] + argv, env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE))
        timing = []
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, e.decode()[-1500:]
            timing += [json.loads(ln[len("PG_TIMING "):]) for ln in e.decode().splitlines() if ln.startswith("PG_TIMING ")]
        with open(out) as f, open(os.path.join(gold, case["name"] + ".out")) as g:
            got, want = f.read(), g.read()
        G.compare_text(align_columns(got, want), want, G.round_digits(case))
        assert len(timing) == size and all(t["sharded_input"] for t in timing), timing
        assert sum(t["sites"] for t in timing) == n_lines and max(t["sites"] for t in timing) < 0.7 * n_lines, timing
    # more ranks than lines: the ranks without a line contribute zero counts and still take part in every exchange
    small = str(tmp_path / "two_lines.geno")
    with open(geno) as f, open(small, "w") as g:
        g.write("".join(f.readlines()[:3]))
    outs = []
    for k, size in enumerate((1, 4)):
        out = str(tmp_path / ("small%d.out" % size))
        procs = []
        for rank in range(size):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(39000 + (os.getpid() + 23 * k) % 2000))
            procs.append(subprocess.Popen([sys.executable, "-c", CLI_WORKER, "distMat.py", "-g", small, "-f", "phased", "--windType", "cat",
                                           "--outFormat", "raw", "-o", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, e.decode()[-1500:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] and len(outs[0]) > 20


def test_predefined_windows_are_sharded_when_the_file_agrees_with_the_window_list(tmp_path):
    """`--windType predefined` on plain text with 2 and 3 ranks: every rank scans its share of the bytes for scaffold runs
    (pg_text_runs), the plan cuts the file at run boundaries, every rank streams its own windows (window-list IDs, failed windows
    and windows beyond a scaffold's last row included) and the reference's output comes out of the one gather.  A window list
    that disagrees with the file (scaffolds in another order) falls back to replicated ingestion with the same output"""
    import gzip
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES
    from golden_util import align_columns
    import test_gpu_golden as G
    gold = os.path.join(ROOT, "tests", "golden")
    case = [c for c in CASES if c["name"] == "sparse_predefined"][0]
    geno = str(tmp_path / (case["fixture"] + ".geno"))
    with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
        g.write(f.read())
    n_lines = sum(1 for _ in open(geno)) - 1
    for k, size in enumerate((2, 3)):
        out = str(tmp_path / ("pre%d.out" % k))
        argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
        port = 37000 + (os.getpid() + 17 * k) % 2000
        procs = []
        for rank in range(size):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), PG_TIMING="1", PG_STREAM_BYTES="20000")
            procs.append(subprocess.Popen([sys.executable, "-c", CLI_WORKER, case["tool"]] + argv, env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE))
        timing = []
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, e.decode()[-1500:]
            timing += [json.loads(ln[len("PG_TIMING "):]) for ln in e.decode().splitlines() if ln.startswith("PG_TIMING ")]
        with open(out) as f, open(os.path.join(gold, case["name"] + ".out")) as g:
            got, want = f.read(), g.read()
        G.compare_text(align_columns(got, want), want, G.round_digits(case))
        assert len(timing) == size and all(t["sharded_input"] for t in timing), timing
        assert sum(t["sites"] for t in timing) == n_lines and max(t["sites"] for t in timing) < 0.8 * n_lines
    # a window list whose scaffolds come in another order than the file's runs: the forward-only walk leaves the chr1 windows empty
    # (the reader is past chr1 when they are asked for); the plan refuses, every rank reads everything, the output is the
    # single-rank one
    coords = str(tmp_path / "swapped.txt")
    with open(coords, "w") as f:
        f.write("chr3 1 1000 a\nchr3 2000 2600 b\nchr1 100 900 c\nchr1 500 1500 d\n")
    outs = []
    for k, size in enumerate((1, 2)):
        out = str(tmp_path / ("swap%d.out" % size))
        argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
        argv[argv.index("--windCoords") + 1] = coords
        procs = []
        for rank in range(size):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(38000 + (os.getpid() + 19 * k) % 2000), PG_TIMING="1")
            procs.append(subprocess.Popen([sys.executable, "-c", CLI_WORKER, case["tool"]] + argv, env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE))
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, e.decode()[-1500:]
            assert '"sharded_input": true' not in e.decode()
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") == 5


def test_file_comm_ignores_the_leftovers_of_a_dead_launch(tmp_path, monkeypatch):
    """a launch that died under the same MASTER_ADDR / MASTER_PORT leaves a pointer, a directory, a `go` file and
    finished exchanges behind.  A new launch must not read any of it: rank 0 opens a fresh directory, and the other ranks adopt a
    directory only when its `go` file quotes the token they left there -- whichever of them starts first."""
    import json
    import threading
    import time
    monkeypatch.setenv("PG_RDZV_FILE", str(tmp_path / "rdzv"))
    stale = tmp_path / "rdzv.d.dead"
    stale.mkdir()
    (tmp_path / "rdzv.dir").write_text(str(stale))
    (stale / "go").write_text(json.dumps({"1": "an old token"}))
    np.save(str(stale / "g0_r0.npy"), np.array([666.0]))
    np.save(str(stale / "g0_r1.npy"), np.array([667.0]))
    got = [None, None]

    def run(r, delay):
        time.sleep(delay)
        c = dist.FileComm(dist.World(r, 2, r), timeout_s=30)
        got[r] = (c.dir, c.allgather(np.array([float(r + 1)])).ravel().tolist())
        c.close()
    th = [threading.Thread(target=run, args=(0, 0.3)), threading.Thread(target=run, args=(1, 0.0))]     # rank 1 meets the stale pointer first
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert got[0][1] == got[1][1] == [1.0, 2.0] and got[0][0] == got[1][0] != str(stale)
    assert sorted(os.listdir(str(tmp_path))) == ["rdzv.d.dead"]


def test_rank_0_failing_to_create_the_rccl_id_tells_the_waiting_ranks(tmp_path, monkeypatch):
    """when rank 0 cannot create the unique id, the other ranks must not sit out the 180 s hand-over timeout: the
    failure travels through the hand-over file and every rank ends up in the same fallback"""
    monkeypatch.setenv("PG_RDZV_FILE", str(tmp_path / "rdzv"))

    def broken():
        raise RuntimeError("librccl.so not found")
    with pytest.raises(RuntimeError):
        dist.exchange_unique_id(dist.World(0, 2, 0), broken)
    with pytest.raises(RuntimeError, match="rank 0 reported"):
        dist.exchange_unique_id(dist.World(1, 2, 1), None, timeout_s=5)


def _launch(tool, argv, size, port, env_extra=None):
    """the driver on `size` ranks (CPU stand-in engine); PG_COMM=file from 4 ranks on: no torch import in eight processes"""
    procs = []
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PG_TIMING="1", PG_STREAM_BYTES="20000")
        env.update(env_extra or {})
        if size >= 4:
            env["PG_COMM"] = "file"
        procs.append(subprocess.Popen([sys.executable, "-c", CLI_WORKER, tool] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    import json
    timing = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
        assert p.returncode == 0, e.decode()[-1500:]
        timing += [json.loads(ln[len("PG_TIMING "):]) for ln in e.decode().splitlines() if ln.startswith("PG_TIMING ")]
    return sorted(timing, key=lambda t: t["rank"])


@pytest.mark.parametrize("name,sizes", [("one_popgen_overlap_failed_id", (2, 3, 8)), ("one_popgen_stepgap", (8,)), ("one_popgen_sites", (3, 8)),
                                        ("one_distmat_windows_id", (8,)), ("four_popgen_id", (2, 3, 8)), ("four_abba_overlap", (8,)),
                                        ("four_fourpop", (3,)), ("bigpos_popgen_sites", (3,))])
def test_window_ranges_shard_one_and_four_scaffolds_over_2_3_and_8_ranks(name, sizes, tmp_path):
    """SURVEY 8e "else split a scaffold's window range": a ONE-scaffold file and a FOUR-scaffold file (the layout of the north-star
    data set) on 2, 3 and 8 ranks.  Every rank reads, tokenises and computes only its window range (genomics_general_amd.shardplan:
    cuts inside scaffold runs), the output is the unmodified reference's (window IDs, failed and empty windows, -s < -w overlap,
    -s > -w gaps, sites windows, the distMat side file), every rank got data and none read more than 1.3 / N of the bytes plus the
    lines of one window span.  Cuts between scaffold runs alone (round 3) gave shares [1, 0, ...] and [.25, 0, .25, 0, ...] here."""
    import gzip
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES, FIXTURES
    from golden_util import align_columns
    import test_gpu_golden as G
    gold = os.path.join(ROOT, "tests", "golden")
    case = [c for c in CASES if c["name"] == name][0]
    geno = str(tmp_path / (case["fixture"] + ".geno"))
    with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
        g.write(f.read())
    n_lines = sum(1 for _ in open(geno)) - 1
    size_b = os.path.getsize(geno)
    av = case["argv"]
    w = int(av[av.index("-w") + 1])
    step = int(av[av.index("-s") + 1]) if "-s" in av else w
    if "sites" in av:                                            # a window span in lines
        span_lines = w + 1
    else:
        span_lines = (w + step) * n_lines / sum(FIXTURES[case["fixture"]]["scaf_len"]) * 1.5 + 2
    for k, size in enumerate(sizes):
        out = str(tmp_path / ("%s_%d.out" % (name, size)))
        argv = [a.format(geno=geno, dir=gold, out=out) for a in av] + ["-o", out]
        timing = _launch(case["tool"], argv, size, 41000 + (os.getpid() * 7 + 31 * k + len(name)) % 4000,
                         {"PG_RDZV_FILE": str(tmp_path / ("rdzv%d" % size))})
        with open(out) as f, open(os.path.join(gold, name + ".out")) as g:
            got, want = f.read(), g.read()
        G.compare_text(align_columns(got, want), want, G.round_digits(case))
        if os.path.exists(os.path.join(gold, name + ".out.windows")):
            with open(out + ".windows") as f, open(os.path.join(gold, name + ".out.windows")) as g:
                assert f.read() == g.read()
        assert len(timing) == size and all(t["sharded_input"] and t["window_ranges"] for t in timing), timing
        n_windows = want.count("\n") - 1 if case["tool"] != "distMat.py" else open(os.path.join(gold, name + ".out.windows")).read().count("\n")
        # (ranks whose equal split of the bytes falls into one and the same window have nothing of their own: 8 windows on 8 ranks)
        assert sum(t["sites"] > 0 for t in timing) >= min(size, n_windows // 2), [t["sites"] for t in timing]
        for t in timing:
            assert t["text_bytes"] <= 1.3 / size * size_b + span_lines * size_b / n_lines + 64, (size, [x["text_bytes"] for x in timing], size_b)


@pytest.mark.parametrize("name,size,bgzf", [("ploidyshift_popgen", 3, False), ("ploidyshift_popgen_sites_pairs", 2, True),
                                             ("ploidyshift_abba", 3, True), ("ploidyshift_distmat", 2, False)])
def test_infer_ploidy_with_changing_ploidy_on_2_and_3_ranks(name, size, bgzf, tmp_path):
    """--inferPloidy on a file whose cell widths change (missing #1): every rank scans the widths and reads the
    whole input (no window-range shards: the host tokenizer under the widest ploidies), the windows of every block are split over the
    ranks and computed under their own ploidies, the rows are gathered per block; plain text and BGZF, 9 kB blocks"""
    import gzip
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES
    from golden_util import align_columns
    import test_gpu_golden as G
    import test_cli_cpu
    gold = os.path.join(ROOT, "tests", "golden")
    case = [c for c in CASES if c["name"] == name][0]
    with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f:
        text = f.read()
    geno = str(tmp_path / (case["fixture"] + (".geno.gz" if bgzf else ".geno")))
    if bgzf:
        test_cli_cpu.write_bgzf(geno, text, 4000, empty_member_at=3)
    else:
        with open(geno, "wb") as g:
            g.write(text)
    out = str(tmp_path / (name + ".out"))
    argv = [a.format(geno=geno, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
    timing = _launch(case["tool"], argv, size, 47000 + (os.getpid() * 5 + len(name)) % 2000,
                     {"PG_RDZV_FILE": str(tmp_path / "rdzv"), "PG_COMM": "file", "PG_STREAM_BYTES": "9000"})
    with open(out) as f, open(os.path.join(gold, name + ".out")) as g:
        got, want = f.read(), g.read()
    G.compare_text(align_columns(got, want), want, G.round_digits(case))
    assert timing and not any(t["sharded_input"] for t in timing), timing


def _corrupt_one_share(geno, size, bad_rank, tmp_path):
    """a copy of the text file `geno` with a position that is not a number on a line in the middle of rank bad_rank's equal share of the
    bytes (the shard plan cuts near the equal split, so that line is tokenised by that rank alone)"""
    with open(geno, "rb") as f:
        text = f.read()
    at = text.index(b"\n", int(len(text) * (bad_rank + 0.5) / size)) + 1
    line_end = text.index(b"\n", at)
    fields = text[at:line_end].split(b"\t")
    fields[1] = b"12x4"
    bad = str(tmp_path / "bad.geno")
    with open(bad, "wb") as f:
        f.write(text[:at] + b"\t".join(fields) + text[line_end:])
    return bad


@pytest.mark.parametrize("comm", ["file", "gloo"])
def test_a_failing_rank_ends_the_whole_launch_within_seconds(comm, tmp_path):
    """with 8 ranks and an input that makes ONE rank's tokenizer raise, every process must be gone within seconds,
    with a non-zero exit code, the error named once by the rank that met it and one line by each of the others -- not one rank sitting
    in the exchange until PG_COMM_TIMEOUT (300 s).  The failing rank leaves a marker next to the rendezvous file
    (cli.guarded_main, dist.mark_failed); every wait loop of the file communicator, and the helper-thread watch around a collective
    of the RCCL communicator (here its gloo stand-in is wrapped the same way), looks for it."""
    import gzip
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import CASES
    gold = os.path.join(ROOT, "tests", "golden")
    case = [c for c in CASES if c["name"] == "one_popgen_overlap_failed_id"][0]
    geno = str(tmp_path / "one.geno")
    with gzip.open(os.path.join(gold, case["fixture"] + ".geno.gz"), "rb") as f, open(geno, "wb") as g:
        g.write(f.read())
    size, bad_rank = (8, 5) if comm == "file" else (3, 1)
    bad = _corrupt_one_share(geno, size, bad_rank, tmp_path)
    out = str(tmp_path / "never.out")
    argv = [a.format(geno=bad, dir=gold, out=out) for a in case["argv"]] + ["-o", out]
    worker = CLI_WORKER.replace("dist.RcclComm = lambda engine, world: dist.GlooComm(world)", GUARDED_GLOO) if comm == "gloo" else CLI_WORKER
    procs, t0 = [], time.time()
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(43000 + (os.getpid() * 3 + size) % 2000), PG_STREAM_BYTES="20000", PG_COMM_TIMEOUT="120",
                   PG_RDZV_FILE=str(tmp_path / "rdzv"))
        if comm == "file":
            env["PG_COMM"] = "file"
        procs.append(subprocess.Popen([sys.executable, "-c", worker, case["tool"]] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    errs = []
    for p in procs:
        try:
            _, e = p.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
            _, e = p.communicate()
            e += b"\n[killed by the test after 60 s]"
        errs.append(e.decode())
    took = time.time() - t0
    assert all(p.returncode not in (0, None) for p in procs), [p.returncode for p in procs]
    assert not any("killed by the test" in e for e in errs), "a rank hung: " + " | ".join(e[-200:] for e in errs)
    assert took < 30, "the launch took %.0f s to end" % took
    assert "12x4" in errs[bad_rank] or "position" in errs[bad_rank], errs[bad_rank][-600:]
    named = [r for r, e in enumerate(errs) if "rank %d failed" % bad_rank in e]
    assert len(named) >= size - 2 and bad_rank not in named, (named, [e[-300:] for e in errs])
    assert sum("Traceback" in e for e in errs) == 1, "the error must be spelled out once, by the rank that met it"


def test_the_marker_of_an_earlier_failed_launch_does_not_end_a_healthy_one(tmp_path):
    """under a stable rendezvous name a marker of a launch that failed ten seconds ago -- rank 1's, with that
    launch's token or without any -- was believed by a rank of the next, healthy launch that started before this launch's rank 0
    had cleared it.  Markers quote their launch's token now; one without a token must be as young as the process that reads it."""
    import time
    rdzv = str(tmp_path / "rdzv")
    for name, body in (("1", "pg-failure token=dir-rdzv.d.old start=1.0\nValueError: the old error"), ("2", "ValueError: older format")):
        with open(rdzv + ".failed_r" + name, "w") as f:
            f.write(body)
        os.utime(rdzv + ".failed_r" + name, (time.time() - 10, time.time() - 10))
    procs = []
    for rank in (1, 2, 0):                                   # rank 0 comes last, late
        if rank == 0:
            time.sleep(1.0)
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3", PG_COMM="file", PG_RDZV_FILE=rdzv, PG_COMM_TIMEOUT="60")
        procs.append((rank, subprocess.Popen([sys.executable, "-c", FILE_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for rank, p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        assert p.returncode == 0 and ("rank %d ok" % rank) in o.decode(), (rank, o.decode()[-2000:])
    assert os.listdir(str(tmp_path)) == []


def test_failure_markers_are_tied_to_the_launch(tmp_path, monkeypatch):
    from genomics_general_amd import dist
    monkeypatch.setenv("PG_RDZV_FILE", str(tmp_path / "rdzv"))
    w0, w1, w2 = (dist.World(r, 3, r) for r in range(3))
    dist.set_launch_token("dir-A")
    dist.mark_failed(w1, ValueError("boom"))
    assert dist.peer_failure(w2) == "rank 1 failed: ValueError: boom" and dist.peer_failure(w1) is None
    dist.set_launch_token("dir-B")                          # another launch: not believed
    assert dist.peer_failure(w2) is None
    dist.set_launch_token(None)                             # before the rendezvous: rank 0 (it has cleared older markers) believes it
    assert dist.peer_failure(w2) is None and dist.peer_failure(w0) == "rank 1 failed: ValueError: boom"
    # a rank that leaves because of a peer leaves a marker too, but does not overwrite its own first one
    dist.mark_failed(w2, dist.PeerFailed("rank 1 failed: ValueError: boom"))
    assert "rank 1 failed" in dist.peer_failure(dist.World(1, 3, 1))
    dist.mark_failed(w1, dist.PeerFailed("rank 2 failed: x"))
    assert dist._read_marker(str(tmp_path / "rdzv") + ".failed_r1")[1] == "ValueError: boom"
    # start-up: a rank removes its own marker when that is older than the process
    old = str(tmp_path / "rdzv") + ".failed_r2"
    os.utime(old, (dist._T_START - 5, dist._T_START - 5))
    dist.forget_own_marker(w2)
    assert not os.path.exists(old)


GUARDED_GLOO = """
class _Guarded(dist.RcclComm):
    def __init__(self, engine, world):
        import os
        self.size, self.rank, self.world = world.size, world.rank, world
        self.timeout_s = float(os.environ.get("PG_COMM_TIMEOUT", "300"))
        if world.rank == 0:
            dist.clear_failures(world)
        self.g = self._guarded(dist.GlooComm, world)
    def allgather(self, arr):
        return self._guarded(self.g.allgather, arr)
    def barrier(self):
        self._guarded(self.g.barrier)
    def close(self):
        pass
dist.RcclComm = _Guarded
"""


def test_bench_spawn_ranks_stops_everybody_when_one_rank_fails(tmp_path):
    """bench.py --gpus N without a launcher: a rank that dies must not leave the parent waiting for rank 0 (which waits for the dead one)"""
    import time
    script = str(tmp_path / "fake_bench.py")
    with open(script, "w") as f:
        f.write("import os, sys, time\nsys.path.insert(0, %r)\nimport bench\n"
                "if 'RANK' not in os.environ:\n    sys.exit(bench.spawn_ranks(4, script=os.path.abspath(__file__)))\n"
                "if os.environ['RANK'] == '2':\n    sys.exit(7)\ntime.sleep(120)\n" % ROOT)
    t0 = time.time()
    p = subprocess.run([sys.executable, script], capture_output=True, timeout=90)
    assert p.returncode == 7 and time.time() - t0 < 30, (p.returncode, time.time() - t0, p.stderr.decode()[-500:])
    assert b"rank 2 ended with exit code 7" in p.stderr
