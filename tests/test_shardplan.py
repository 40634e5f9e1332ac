"""Window-range sharding of the input (genomics_general_amd.shardplan): whatever the number of scaffold runs, the windows the
ranks generate from their own shares of the lines -- cut INSIDE runs --, concatenated in rank order with the IDs shifted, are the
windows of the whole input, and the shares are balanced.  The ranks are threads here (one reader each, a barrier communicator);
the subprocess tests of tests/test_dist.py run the drivers on the same plan."""
import os
import threading

import numpy as np
import pytest

from genomics_general_amd import dist, genoio, shardplan
from genomics_general_amd import windows as W
from test_windows import _stream_rows


class _Shared:
    def __init__(self, n):
        self.slots, self.barrier = [None] * n, threading.Barrier(n)


class ThreadComm:
    def __init__(self, shared, rank, size):
        self.sh, self.rank, self.size = shared, rank, size

    def allgather(self, arr):
        self.sh.slots[self.rank] = np.ascontiguousarray(arr, dtype=np.float64).ravel().copy()
        self.sh.barrier.wait()
        out = np.stack(list(self.sh.slots))
        self.sh.barrier.wait()
        return out


def on_ranks(n, fn):
    """fn(world, comm) on n threads -> list of results in rank order (an exception of any rank is raised here)"""
    sh, out, err = _Shared(n), [None] * n, []

    def work(r):
        try:
            out[r] = fn(dist.World(r, n, r), ThreadComm(sh, r, n))
        except BaseException as exc:                            # noqa: B902
            err.append(exc)
            sh.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise [e for e in err if not isinstance(e, threading.BrokenBarrierError)][0] if any(
            not isinstance(e, threading.BrokenBarrierError) for e in err) else err[0]
    return out


def random_input(rng, max_runs=5, max_rows=120, pool=("c0", "c1", "c2", "c3"), dense=False, span=600):
    names, starts, pos, prev = [], [], [], None
    for _ in range(int(rng.integers(1, max_runs + 1))):
        nm = str(rng.choice([x for x in pool if x != prev]))
        prev = nm
        starts.append(len(pos))
        names.append(nm)
        n = int(rng.integers(1, max_rows))
        if dense:
            p0 = int(rng.integers(1, 40))
            pos += list(range(p0, p0 + n))
        else:
            pos += list(np.sort(rng.integers(1, span, size=n)))      # duplicates of a position included
    return np.array(starts), names, np.array(pos, dtype=np.int32)


def write_geno(path, starts, names, pos, rng=None, comments=False, cell=None):
    """the cell of a row is its global row number (nothing tokenises these files) unless a genotype `cell` is given"""
    run_of = np.searchsorted(starts, np.arange(len(pos)), side="right") - 1
    with open(path, "wb") as f:
        f.write(b"#CHROM\tPOS\ts0\n")
        for i, p in enumerate(pos):
            if comments and rng is not None and not rng.integers(0, 25):
                f.write(b"#a comment line\n")
            f.write(("%s\t%d\t%s\n" % (names[run_of[i]], p, cell if cell else i)).encode())


def parse_rows(body):
    """(run starts, run names, positions, global rows) of the data lines of a share"""
    rs, rn, pos, rows = [], [], [], []
    for ln in bytes(body).split(b"\n"):
        if not ln.strip() or ln.startswith(b"#"):
            continue
        f = ln.split()
        nm = f[0].decode()
        if not rn or rn[-1] != nm:
            rs.append(len(pos))
            rn.append(nm)
        pos.append(int(f[1]))
        rows.append(int(f[2]))
    return np.array(rs, dtype=np.int64), rn, np.array(pos, dtype=np.int32), rows


def stream_share(S, rs, rn, pos, row0, rng):
    """feed a share to its window stream in random pieces (as the drivers do block by block) -> rows with global row ranges"""
    n = len(pos)
    cuts = sorted(set(rng.integers(0, n + 1, size=int(rng.integers(0, 4))).tolist() + [n]))
    run_of = np.searchsorted(rs, np.arange(n), side="right") - 1 if n else np.zeros(0, dtype=int)
    got, hist, keep = [], [], 0
    for c, final in [(c, False) for c in cuts] + [(n, True)]:    # the drivers signal the end with one more, empty, block
        a, b = keep, max(c, keep)
        ro = run_of[a:b]
        chg = np.flatnonzero(np.concatenate([[True], ro[1:] != ro[:-1]])) if b > a else np.array([], dtype=int)
        T, kf = S.feed(chg, [rn[ro[i]] for i in chg], pos[a:b], final=final)
        got += _stream_rows(T, hist, a + row0)
        keep = a + kf
    return got


def _wparams(kind, w, x):
    if kind == "coordinate":
        return dict(windType="coordinate", windSize=w, stepSize=x, overlap=0, maxDist=np.inf)
    return dict(windType="sites", windSize=w, stepSize=None, overlap=x, maxDist=np.inf)


def check_plan(path, starts, names, pos, kind, w, x, n_ranks, rng, inc=None, exc=None, opener=None, min_sites=None):
    """the windows of the ranks' shares == the windows of the whole input; returns the plans"""
    if kind == "coordinate":
        whole = W.coord_windows(starts, names, pos, w, x, inc, exc)
    else:
        whole = W.sites_windows(starts, names, pos, w, x, np.inf, min_sites, inc, exc)
    want = _stream_rows(whole, [])
    wanted = lambda nm: W._wanted(nm, inc, exc)                   # noqa: E731
    wp = _wparams(kind, w, x)

    def rank(world, comm):
        rd = (opener or genoio.open_input)(path)
        rd.read_header()
        plan = shardplan.shard_reader(rd, world, comm, wp, wanted)
        assert plan is not None
        if getattr(rd, "packed", False):
            a, b = rd._rows
            rs = np.array([s - a for s in starts if a < s < b], dtype=np.int64)
            first = int(np.searchsorted(starts, a, side="right")) - 1
            rn = [names[first]] + [names[k] for k, s in enumerate(starts) if a < s < b] if b > a else []
            share = (np.concatenate([[0], rs]).astype(np.int64) if b > a else rs, rn, pos[a:b], list(range(a, b)))
        else:
            share = parse_rows(rd.read_block(None))
        rd.close()
        return plan, share

    res = on_ranks(n_ranks, rank)
    got, shift, covered = [], 0, 0
    for plan, (rs, rn, ps, rows) in res:
        assert rows == list(range(rows[0], rows[0] + len(rows))) if rows else True
        if kind == "coordinate":
            S = W.CoordWindowStream(w, x, inc, exc, start=plan.start, stop=plan.stop)
        else:
            S = W.SitesWindowStream(w, x, np.inf, min_sites, inc, exc)
        part = stream_share(S, rs, rn, ps, rows[0] if rows else 0, rng)
        got += [(s, st, en, r_, i + shift, m) for (s, st, en, r_, i, m) in part]
        shift += S.done
    assert got == want, (kind, w, x, n_ranks, inc, exc, [(p.start, p.stop) for p, _ in res])
    return res


@pytest.mark.parametrize("seed", range(6))
def test_coordinate_windows_of_window_range_shards_are_the_windows_of_the_whole_input(seed, tmp_path):
    rng = np.random.default_rng(7000 + seed)
    path = str(tmp_path / "x.geno")
    for trial in range(60):
        starts, names, pos = random_input(rng, max_runs=1 if trial % 3 == 0 else 5, dense=trial % 4 == 1)
        write_geno(path, starts, names, pos, rng, comments=trial % 5 == 2)
        w = int(rng.integers(3, 150))
        step = int(rng.integers(1, 2 * w))
        inc = exc = None
        z = int(rng.integers(0, 4))
        if z == 1:
            inc = [str(v) for v in rng.choice(["c0", "c1", "c2", "c3"], size=2, replace=False)]
        if z == 2:
            exc = [str(v) for v in rng.choice(["c0", "c1", "c2", "c3"], size=1)]
        for n_ranks in (2, 3, 8):
            check_plan(path, starts, names, pos, "coordinate", w, step, n_ranks, rng, inc, exc)


@pytest.mark.parametrize("seed", range(4))
def test_sites_windows_of_window_range_shards_are_the_windows_of_the_whole_input(seed, tmp_path):
    rng = np.random.default_rng(7100 + seed)
    path = str(tmp_path / "x.geno")
    for trial in range(60):
        starts, names, pos = random_input(rng, max_runs=1 if trial % 3 == 0 else 5)
        write_geno(path, starts, names, pos, rng, comments=trial % 5 == 2)
        w = int(rng.integers(2, 40))
        ov = int(rng.integers(0, w))
        ms = int(rng.integers(1, w + 1))                     # (-m may lie below -O: ADVICE round 4)
        exc = [str(v) for v in rng.choice(["c0", "c1", "c2", "c3"], size=1)] if trial % 4 == 3 else None
        for n_ranks in (2, 3, 8):
            check_plan(path, starts, names, pos, "sites", w, ov, n_ranks, rng, None, exc, min_sites=ms)


@pytest.mark.parametrize("seed", range(3))
def test_shards_with_sixteen_ranks_sparse_positions_and_scaffold_names_that_prefix_each_other(seed, tmp_path):
    """minSites anywhere in 1..w (also below the overlap), 2 / 5 / 16 ranks, positions spread over 10^6 (most
    coordinate windows empty), scaffold names c, c1, c10, c100 (a run boundary must not be found by a prefix match)"""
    rng = np.random.default_rng(7300 + seed)
    path = str(tmp_path / "x.geno")
    pool = ("c", "c1", "c10", "c100")
    for trial in range(40):
        starts, names, pos = random_input(rng, max_runs=1 if trial % 3 == 0 else 6, pool=pool, span=1000000 if trial % 2 else 600)
        write_geno(path, starts, names, pos, rng, comments=trial % 5 == 2)
        w = int(rng.integers(2, 40))
        ov = int(rng.integers(0, w))
        ms = int(rng.integers(1, w + 1))
        exc = [str(rng.choice(pool))] if trial % 4 == 3 else None
        cw = int(rng.integers(3, 150)) * (2000 if trial % 2 else 1)
        cstep = int(rng.integers(1, 2 * cw))
        for n_ranks in (2, 5, 16):
            check_plan(path, starts, names, pos, "sites", w, ov, n_ranks, rng, None, exc, min_sites=ms)
            check_plan(path, starts, names, pos, "coordinate", cw, cstep, n_ranks, rng, None, exc)


@pytest.mark.parametrize("seed", range(3))
def test_window_range_shards_of_bgzf_and_packed_input(seed, tmp_path):
    """the same plan on bgzip-compressed text (cuts are (member, offset in member) pairs found on the inflated stream) and on
    `.pgeno` files (cuts are rows, positions read from the blocks' position arrays; coordinate and sites windows)"""
    from test_host import _bgzf_write
    rng = np.random.default_rng(7200 + seed)
    path = str(tmp_path / "x.geno")
    for trial in range(25):
        starts, names, pos = random_input(rng, max_runs=1 if trial % 3 == 0 else 4, dense=trial % 4 == 1)
        w = int(rng.integers(3, 150))
        step = int(rng.integers(1, 2 * w))
        exc = [str(v) for v in rng.choice(["c0", "c1", "c2", "c3"], size=1)] if trial % 4 == 2 else None
        write_geno(path, starts, names, pos)
        with open(path, "rb") as f:
            _bgzf_write(path + ".gz", f.read(), blk=int(rng.integers(200, 3000)))
        for n_ranks in (2, 3, 8):
            check_plan(path + ".gz", starts, names, pos, "coordinate", w, step, n_ranks, rng, None, exc)
        write_geno(path, starts, names, pos, cell="A/C")
        genoio.pack_geno(path, path[:-5] + ".pgeno", "phased", block_bytes=int(rng.integers(300, 4000)),
                         codec="zlib" if trial % 2 else "none")
        ws = int(rng.integers(2, 40))
        ov = int(rng.integers(0, ws))
        ms = int(rng.integers(1, ws + 1))
        for n_ranks in (2, 3, 8):
            check_plan(path[:-5] + ".pgeno", starts, names, pos, "coordinate", w, step, n_ranks, rng, None, exc)
            check_plan(path[:-5] + ".pgeno", starts, names, pos, "sites", ws, ov, n_ranks, rng, None, exc, min_sites=ms)


def test_shares_of_a_one_scaffold_and_a_four_scaffold_file_are_balanced(tmp_path):
    """what cuts between scaffold runs cannot give: 8 ranks on ONE scaffold, and on four (the north-star data set's layout) -- every
    rank reads about 1/8 of the bytes (its share plus at most the lines of one window span), the windows are the whole input's"""
    rng = np.random.default_rng(5)
    for n_scaf in (1, 4):
        per = 24000 // n_scaf
        starts = np.arange(n_scaf) * per
        names = ["chr%d" % (k + 1) for k in range(n_scaf)]
        pos = np.tile(np.arange(1, per + 1), n_scaf).astype(np.int32)
        path = str(tmp_path / ("s%d.geno" % n_scaf))
        write_geno(path, starts, names, pos)
        size = os.path.getsize(path)
        for kind, w, x in (("coordinate", 500, 500), ("coordinate", 600, 200), ("sites", 400, 100)):
            for n_ranks in (2, 3, 8):
                res = check_plan(path, starts, names, pos, kind, w, x, n_ranks, rng, min_sites=w if kind == "sites" else None)
                span = (w + (x if kind == "coordinate" else 0)) * 16      # bytes of the lines of one window + one step
                for plan, share in res:
                    assert len(share[3]) > 0
                    assert plan.share <= 1.0 / n_ranks + span / size + 0.01, (kind, n_ranks, plan.share)
                assert plan.scanned <= 3 * span + size // n_ranks * (kind == "sites")
