"""-m gpu: parity at BASELINE.json's full sizes.  Every configuration (C2 popgenWindows, C3 ABBA-BABA, C4 distMat, the
north-star single-GPU shape of C5) is run through the HIP path at its full size and sampled windows are downloaded and
compared with the CPU oracle: the integer matrices D / C, l / S / pair sums and sitesUsed bit-exact, pi / dxy / Fst, the
ABBA-BABA statistics, theta / Tajima's D and the individual-pair means within 1e-9 relative (north-star tolerance: 1e-6).
Size-independent properties (run-to-run bit identity, additivity of D and C over a split window, agreement of the independent
pairwise code paths) ride along as secondary checks."""
import numpy as np
import pytest

from genomics_general_amd import synth
from genomics_general_amd.engine import Engine
from oracle import popgen_oracle as orc

import gpu_util as G

pytestmark = pytest.mark.gpu

N_SITES, N_DIP, N_POPS, WIND = 10_000_000, 100, 4, 50_000


def resident(n_sites, n_dip, n_pops, n_scaf):
    names, lay = G.make_layout(n_dip, n_pops)
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(n_sites)
    e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, n_sites // n_scaf, n_dip, n_pops, G.slot_gen_hap(names, lay), synth.VAR_THR,
                 synth.MISS_THR)
    return e, lay


def oracle_window(e, lay, a, b):
    """the reference-order alignment of resident sites [a,b) (downloaded from the device) and its integer matrices"""
    codes = e.download(int(a), int(b - a))
    aln, _ = orc.aln_from_codes(codes, lay.hap_names, lay.hap_sample_name, lay.hap_group)
    return aln


def check_popdist(e, lay, lo, hi, sel, st, min_sites, same=False):
    """D / C bit-exact and pi / dxy / Fst within 1e-9 of the oracle (same: to the last bit) on the windows `sel` of a full-size batch
    whose statistics `st` were computed over ALL windows of the batch"""
    D, C = e.batch(lo[sel], hi[sel]).pairCounts(reference_order=True)
    for k, w in enumerate(sel):
        aln = oracle_window(e, lay, lo[w], hi[w])
        Do, Co = orc.pair_counts_gemm(aln)                       # genomics.py:903-916, 1042-1047
        assert np.array_equal(C[k], Co), "C differs in window %d" % w
        assert np.array_equal(D[k], Do), "D differs in window %d" % w
        so, _ = orc.group_dist_stats(aln, Do, Co, True, min_sites, 0.01)     # genomics.py:956-995
        assert Do.max() > 0 and Co.min() >= 0 and Co.max() > (hi[w] - lo[w]) // 2, "window %d holds no called data" % w
        for key, v in so.items():
            assert np.isfinite(v), (key, w)
            assert G.same(st[key][w], v) if same else G.close(st[key][w], v), (key, w, st[key][w], v)


# ---------------------------------------------------------------------------------------------------------
# C2 (popgenWindows pi / dxy / Fst) and C3 (ABBA-BABA): 10^7 sites x 200 haplotypes, 200 windows of 50 kb
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2():
    e, lay = resident(N_SITES, N_DIP, N_POPS, 4)
    lo = np.arange(0, N_SITES, WIND, dtype=np.int64)
    yield e, lay, lo, lo + WIND
    e.close()


def test_c2_popdist_windows_match_the_oracle(c2):
    e, lay, lo, hi = c2
    st = e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    assert len(st) == 4 + 2 * 12
    check_popdist(e, lay, lo, hi, [0, 113, 199], st, 100)
    # run-to-run bit identity and bounds over all 200 windows
    again = e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    for k, v in st.items():
        assert v.shape == (200,) and np.array_equal(v, again[k]) and np.all(np.isfinite(v)), k
        assert np.all(v <= 1) and (k.startswith("Fst_") or np.all(v >= 0)), k


def test_c2_popdist_in_numpy_order_is_the_oracle_to_the_last_bit(c2, monkeypatch):
    """windows of 50 kb take the fixed-tree finisher by default (pg_popdist_stats: NumPy's order up to 4096 sites a window); forced,
    k_popdist_np reproduces np.nanmean over the 50 x 50, and 100 x 100 blocks (pieces of 8192 values, genomics.py:976-992) exactly"""
    e, lay, lo, hi = c2
    monkeypatch.setenv("PG_POPDIST_TREE", "1")
    sel = np.array([0, 113, 199])
    st = e.batch(lo[sel], hi[sel]).groupDistStats(True, 100, 0.01)
    full = {k: np.full(200, np.nan) for k in st}
    for k in st:
        full[k][sel] = st[k]
    check_popdist(e, lay, lo, hi, list(sel), full, 100, same=True)


@pytest.mark.parametrize("tree", ["default", "numpy"])
def test_c3_abbababa_windows_match_the_oracle(c2, tree, monkeypatch):
    """BASELINE.json configs[2]: P1/P2/P3/O of 25 diploids each on 10^7 sites, 50 kb windows (genomics.py:1647-1695); windows this
    long take the fixed-tree sums by default (1e-9); forced into NumPy's order (several pieces of 8192 used sites) they are the
    oracle's to the last bit"""
    e, lay, lo, hi = c2
    if tree == "numpy":
        monkeypatch.setenv("PG_QUARTET_TREE", "1")
    got = e.batch(lo, hi).ABBABABA("p0", "p1", "p2", "p3", 0.5)
    for w in (1, 77, 198):
        want = orc.abbababa(oracle_window(e, lay, lo[w], hi[w]), "p0", "p1", "p2", "p3", 0.5)
        assert int(got["sitesUsed"][w]) == want["sitesUsed"] and want["sitesUsed"] > 100
        for key in ("D", "fd", "fdM", "ABBA", "BABA"):
            assert (G.same if tree == "numpy" else G.close)(got[key][w], want[key]), (key, w, got[key][w], want[key])
    again = e.batch(lo, hi).ABBABABA("p0", "p1", "p2", "p3", 0.5)
    for k in got:
        assert np.array_equal(got[k], again[k], equal_nan=True), k


def test_c2_popfreq_and_indpair_windows_match_the_oracle(c2):
    e, lay, lo, hi = c2
    sel = [5, 150]
    f = e.batch(lo[sel], hi[sel]).groupFreqStats()
    wb = e.batch(lo[sel], hi[sel])
    ip = wb.indPairTable(includeSameWithSame=True)
    for k, w in enumerate(sel):
        aln = oracle_window(e, lay, lo[w], hi[w])
        want = orc.group_freq_stats(aln)                          # genomics.py:1002-1028
        for name in lay.sampleData.popNames:
            assert int(f["l_" + name][k]) == want["l_" + name]
            if want["l_" + name] >= 1:
                assert int(f["S_int_" + name][k]) == want["S_" + name]
                for key in ("thetaPi_", "thetaW_", "TajD_"):          # thetaPi: the reference's sequential sum, bit for bit
                    assert float(f[key + name][k]) == want[key + name], (key, name, w, f[key + name][k], want[key + name])
        Do, Co = orc.pair_counts_gemm(aln)
        dmo, _ = orc.ind_pair_dists(aln, orc.dist_from_counts(Do, Co), include_same=True)    # genomics.py:934-954
        for s in range(0, lay.n_samp, 7):
            for t in range(s, lay.n_samp, 5):
                assert G.close(ip[k, lay.sample_pair_index(s, t)], dmo[lay.ind_order[s]][lay.ind_order[t]]), (w, s, t)


def test_integer_matrices_are_additive_over_a_split_window_and_bounded(c2):
    e, lay, lo, hi = c2
    a, c = int(lo[37]), int(hi[37])
    b = a + 20011                                           # not a multiple of 32: exercises word tails
    D, C = e.batch([a, a, b], [c, b, c]).pairCounts(reference_order=False)
    assert np.array_equal(D[0], D[1] + D[2]) and np.array_equal(C[0], C[1] + C[2])
    assert np.array_equal(D[0], D[0].T) and np.array_equal(C[0], C[0].T)
    assert D[0].max() <= C[0].max() <= c - a and np.all(D[0] <= C[0]) and np.all(np.diag(C[0]) == 0)
    assert C[0].sum() > 0 and D[0].sum() > 0


@pytest.mark.parametrize("env", ["PG_NO_DIP", "PG_OVERLAP", "PG_PACK2", "PG_GROUP_WORDS=128", "PG_PAIR_TILE=c", "PG_PAIR_TILE=none"])
def test_independent_pipelines_agree_at_full_window_size(c2, env, monkeypatch):
    e, lay, lo, hi = c2
    sel = [0, 61, 199]
    want = e.batch(lo[sel], hi[sel]).pairCounts(reference_order=False)
    st_want = e.batch(lo[:40], hi[:40]).groupDistStats(True, 100, 0.01)
    name, _, val = env.partition("=")                                       # (PG_GROUP_WORDS: 4096-site compaction groups instead of 2048;
    monkeypatch.setenv(name, val or "1")                                    # the popcount kernels need their own context: test_gpu_kernels)
    got = e.batch(lo[sel], hi[sel]).pairCounts(reference_order=False)
    st_got = e.batch(lo[:40], hi[:40]).groupDistStats(True, 100, 0.01)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    for k in st_want:
        assert np.array_equal(st_got[k], st_want[k]), k


# ---------------------------------------------------------------------------------------------------------
# C4 (distMat): 10^6 sites x 2000 haplotypes, 10 windows of 100 kb
# ---------------------------------------------------------------------------------------------------------
def test_c4_distmat_window_matches_the_oracle():
    e, lay = resident(1_000_000, 1000, 1, 1)
    lo = np.arange(0, 1_000_000, 100_000, dtype=np.int64)
    hi = lo + 100_000
    tab = e.batch(lo, hi).indPairTable(includeSameWithSame=False)            # what distMat.py prints (distMat.py:42)
    w = 6
    D, C = e.batch(lo[w:w + 1], hi[w:w + 1]).pairCounts(reference_order=True)
    aln = oracle_window(e, lay, lo[w], hi[w])
    e.close()
    Do, Co = orc.pair_counts_gemm(aln, dtype=np.float32)          # 1 999 000 pairs x 100 000 sites; exact (counts < 2^24)
    assert np.array_equal(C[0], Co) and np.array_equal(D[0], Do)
    # individual-pair means: the oracle's block-by-block nanmean (genomics.py:934-954) on a sample of 60 of the 1000 individuals
    dm = orc.dist_from_counts(Do, Co)
    rng = np.random.default_rng(4)
    inds = np.sort(rng.choice(lay.n_samp, 60, replace=False))
    ref_pos = {nm: k for k, nm in enumerate(aln.names)}
    rows = [ref_pos[lay.hap_names[s]] for i in inds for s in lay.ind_slots[lay.ind_order[i]]]
    sub = orc.Aln(aln.num[rows][:, :1], [aln.names[r] for r in rows], [aln.sample_names[r] for r in rows], ["all"] * len(rows))
    want, _ = orc.ind_pair_dists(sub, dm[np.ix_(rows, rows)], include_same=False)
    for i in inds:
        for j in inds:
            a, b = lay.ind_order[i], lay.ind_order[j]
            assert G.close(tab[w, lay.sample_pair_index(i, j)], want[a][b]), (a, b)


# ---------------------------------------------------------------------------------------------------------
# north-star single-GPU shape (first 10^8 sites of C5): 10^8 sites x 400 haplotypes, 2000 windows of 50 kb
# ---------------------------------------------------------------------------------------------------------
def test_northstar_shape_windows_match_the_oracle():
    n_sites = 100_000_000
    e, lay = resident(n_sites, 200, 4, 4)
    # 40 GB of rows: Engine.reserve went through pg_reserve_sites_tuned (several placements probed while empty, one kept)
    assert e.placement is not None and len(e.placement[0]) >= 2 and 0 <= e.placement[1] < len(e.placement[0]), e.placement
    assert min(e.placement[0]) == e.placement[0][e.placement[1]] > 0
    # ... and then several sets of the planes the pack kernel writes (pg_tune_planes)
    pp = e.plane_placement
    assert pp is not None and len(pp[0]) >= 2 and min(pp[0]) == pp[0][pp[1]] > 0, pp
    lo = np.arange(0, n_sites, WIND, dtype=np.int64)
    hi = lo + WIND
    st = e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    check_popdist(e, lay, lo, hi, [3, 1999], st, 100)
    again = e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    for k, v in st.items():
        assert v.shape == (2000,) and np.array_equal(v, again[k]) and np.all(np.isfinite(v)), k
    # the planes chosen once more, now on the filled rows: other memory behind them, the same table
    pp = e.tune_planes(n_sites, 3)
    assert len(pp[0]) == 3 and min(pp[0]) == pp[0][pp[1]] > 0, pp
    again = e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    for k, v in st.items():
        assert np.array_equal(v, again[k]), k
    e.close()


@pytest.mark.parametrize("mode", ["default", "PG_NO_DIP"])
def test_one_window_beyond_the_exact_range_of_an_f32_accumulator(mode, monkeypatch):
    """2.2e7 sites in ONE window: more jointly called sites than an f32 accumulator of the fp4 matrix-core path can count exactly
    (it holds count / 4: exact below 2^24 = 1.68e7).  The launcher cuts the word range into parts below 2^23 sites and the parts
    meet in integer atomics; D and C must still equal the site-by-site counts of the downloaded rows."""
    if mode != "default":
        monkeypatch.setenv(mode, "1")
    n = 22_000_000
    names, lay = G.make_layout(4, 2)
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(n)
    e.synth_fill(0, n, 0, synth.SEED_DEFAULT, n, 4, 2, G.slot_gen_hap(names, lay), synth.VAR_THR, 300)     # 1 % missing
    D, C = e.batch([0], [n]).pairCounts(reference_order=False)
    rows = e.download(0, n)
    called = rows != 0
    N = lay.n_hap
    assert called[:, 0].sum() > (1 << 24)
    for i in range(N):
        for j in range(i + 1, N):
            both = called[:, i] & called[:, j]
            assert C[0, i, j] == int(both.sum()) == C[0, j, i], (i, j)
            assert D[0, i, j] == int((both & (rows[:, i] != rows[:, j])).sum()) == D[0, j, i], (i, j)
    e.close()


def test_reservation_without_placement_trials_and_regrowth(monkeypatch):
    """PG_PLACE_TRIALS=1 is the plain reservation; a tuned reservation that grows drops the rows like the plain one and the new
    rows are usable (5 GiB -> 6 GiB of rows at 16 haplotypes)"""
    names, lay = G.make_layout(8, 2)
    monkeypatch.setenv("PG_PLACE_TRIALS", "1")
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(300_000_000)
    assert e.placement is None
    assert e.plane_placement is None
    e.close()
    monkeypatch.delenv("PG_PLACE_TRIALS")
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(340_000_000)
    first = e.placement
    assert first is not None and len(first[0]) >= 2
    assert e.plane_placement is not None
    e.reserve(400_000_000)
    assert e.placement is not None and e.placement is not first
    n = 400_000_000
    e.synth_fill(n - 200_000, 200_000, 0, synth.SEED_DEFAULT, 200_000, 8, 2, G.slot_gen_hap(names, lay), synth.VAR_THR, synth.MISS_THR)
    D, C = e.batch([n - 200_000], [n]).pairCounts(reference_order=True)
    Do, Co = orc.pair_counts_gemm(oracle_window(e, lay, n - 200_000, n))
    assert np.array_equal(D[0], Do) and np.array_equal(C[0], Co)
    e.close()
