"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.
Integer outputs must be bit-exact; float64 outputs within 1e-9 relative (north_star tolerance is 1e-6)."""
import numpy as np
import pytest

from genomics_general_amd import synth
from oracle import popgen_oracle as orc

import gpu_util as G

pytestmark = pytest.mark.gpu


def oracle_aln(lay, codes, lo, hi):
    aln, _ = orc.aln_from_codes(codes[lo:hi], lay.hap_names, lay.hap_sample_name,
                                [g if g is not None else "~none" for g in lay.hap_group])
    return aln


@pytest.mark.parametrize("n_dip,n_pops,L,wins", [
    (8, 2, 3000, [(0, 1000), (1000, 2000), (2000, 3000)]),
    (25, 4, 4096, [(0, 4096), (5, 37), (100, 100), (7, 8), (4000, 4096)]),      # whole, tiny, empty, single-site, tail
    (37, 3, 2500, [(0, 1250), (600, 1900), (1250, 2500)]),                       # odd haplotype count, overlapping windows
    (150, 4, 1500, [(0, 700), (700, 1500)]),                                     # several 64-column chunks
    (530, 3, 2200, [(0, 2200), (100, 421), (2150, 2200)]),                        # > 1024 haplotype slots: presence pre-pass
])
@pytest.mark.parametrize("pack", ["default", "PG_PACK2", "PG_GROUP_WORDS=128", "PG_PAIR_VALU", "PG_PAIR_TILE=c", "PG_PAIR_TILE=none", "PG_PACK_BURST=0",
                                  "PG_GROUP_WORDS=20"])
def test_pairwise_counts_bit_exact(n_dip, n_pops, L, wins, pack, monkeypatch):
    G.set_mode(monkeypatch, pack)
    e, lay, codes, _ = G.make_engine(n_dip, n_pops, L, seed=11 + n_dip)
    lo = np.array([w[0] for w in wins]); hi = np.array([w[1] for w in wins])
    D, C = e.batch(lo, hi).pairCounts(reference_order=True)
    for k, (a, b) in enumerate(wins):
        Do, Co = orc.pair_counts_gemm(oracle_aln(lay, codes, a, b))
        assert np.array_equal(C[k], Co), "C differs in window %d" % k
        assert np.array_equal(D[k], Do), "D differs in window %d" % k
    e.close()


@pytest.mark.parametrize("n_dip,L,p_miss,mix", [
    (12, 3000, 0.1, "uniform"),        # nearly every site carries all four alleles: three virtual sites per site
    (40, 5000, 0.3, "mixed"),          # mono-, bi-, tri- and tetra-allelic sites side by side, heavy missingness
    (9, 700, 0.0, "uniform"),          # haploid-called mismatch impossible, no missing data
    (530, 1300, 0.05, "mixed"),        # > 1024 haplotype slots: presence pre-pass + word prefix scan
    (150, 1500, 0.1, "uniform"),       # two waves per block: list / flush code behind block barriers
    (300, 800, 0.2, "mixed"),          # four waves per block
])
@pytest.mark.parametrize("pack", ["default", "PG_PACK2", "PG_GROUP_WORDS=128", "PG_PAIR_VALU", "PG_PAIR_TILE=c", "PG_PAIR_TILE=none", "PG_PACK_BURST=0",
                                  "PG_GROUP_WORDS=20"])
def test_pairwise_counts_with_three_and_four_alleles_per_site(n_dip, L, p_miss, mix, pack, monkeypatch):
    """sites with k alleles become k-1 virtual biallelic sites in k_pack2 / k_pack3 (both kernels at every block size: one, two
    and four waves); D must still be the plain Hamming count.  The uniform cases overflow the default XV reservation: the
    call is repeated with the worst-case reservation"""
    G.set_mode(monkeypatch, pack)
    rng = np.random.default_rng(1000 + n_dip)
    names, lay = G.make_layout(n_dip, 2)
    H = lay.n_hap
    if mix == "uniform":
        idx = rng.integers(0, 4, size=(L, H))
    else:
        k = rng.integers(1, 5, size=L)                                    # alleles per site
        perm = np.argsort(rng.random((L, 4)), axis=1)                     # which alleles
        pick = (rng.random((L, H)) * k[:, None]).astype(np.int64)         # skewed towards few alleles at many sites
        pick = np.where(rng.random((L, H)) < 0.8, 0, pick)
        idx = np.take_along_axis(perm, pick, axis=1)
    codes = (1 << idx).astype(np.int8)
    codes[rng.random((L, H)) < p_miss] = 0
    codes[::2, 1] = codes[::2, 0]                                         # keep the diploid shortcut honest on some rows
    e = G.Engine(0)
    e.set_layout(lay)
    e.load_sites(codes)
    wins = [(0, L), (L // 3, L // 3 + 70), (L - 40, L), (5, 5)]
    lo = np.array([w[0] for w in wins]); hi = np.array([w[1] for w in wins])
    D, C = e.batch(lo, hi).pairCounts(reference_order=True)
    for kk, (a, b) in enumerate(wins):
        Do, Co = orc.pair_counts_gemm(oracle_aln(lay, codes, a, b))
        assert np.array_equal(C[kk], Co), "C differs in window %d" % kk
        assert np.array_equal(D[kk], Do), "D differs in window %d" % kk
    e.close()


def test_batch_of_empty_windows_after_a_real_batch_gives_zero_matrices():
    """the per-window word counters of the previous call must not leak into a batch whose windows are all empty"""
    e, lay, codes, _ = G.make_engine(10, 2, 2100, seed=3)
    D, C = e.batch([0, 300], [2100, 900]).pairCounts(reference_order=False)
    assert D.any() and C.any()
    D, C = e.batch([5, 700, 2100], [5, 700, 2100]).pairCounts(reference_order=False)
    assert not D.any() and not C.any()
    e.close()


def test_other_planes_behind_the_pack_kernel_give_the_same_counts():
    """pg_tune_planes (Engine.tune_planes): several sets of the planes the pack kernel writes are tried on the resident rows and one
    is kept -- whichever it is, the counts are the ones of the oracle; the call is refused outside the reservation"""
    e, lay, codes, _ = G.make_engine(12, 3, 130_000, seed=11, miss_thr=9000)
    lo, hi = [0, 40_000, 90_001], [40_000, 90_001, 130_000]
    before = e.batch(lo, hi).pairCounts(reference_order=True)
    for trials in (2, 8, 50):
        ms, kept = e.tune_planes(130_000, trials)
        assert len(ms) == min(trials, 8) and 0 <= kept < len(ms) and min(ms) == ms[kept] > 0, (ms, kept)
        after = e.batch(lo, hi).pairCounts(reference_order=True)
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    assert e.tune_planes(130_000, 1) is None
    with pytest.raises(Exception, match="inside the reservation"):
        e.tune_planes(10**9, 2)
    Do, Co = orc.pair_counts_gemm(oracle_aln(lay, codes, 40_000, 90_001))
    assert np.array_equal(before[0][1], Do) and np.array_equal(before[1][1], Co)
    e.close()


def test_pairwise_matches_reference_pair_loop_small():
    """the faithful pair-by-pair loop of the reference (not the GEMM shortcut) on a small case"""
    e, lay, codes, _ = G.make_engine(6, 2, 700, seed=5, miss_thr=20000)
    D, C = e.batch([0], [700]).pairCounts(reference_order=True)
    Do, Co = orc.pair_counts_loop(oracle_aln(lay, codes, 0, 700))
    assert np.array_equal(D[0], Do) and np.array_equal(C[0], Co)
    e.close()


@pytest.mark.parametrize("min_sites,min_data,miss", [(1, 0.01, 5000), (40, 0.5, 30000), (300, 0.01, 20000)])
def test_group_dist_stats(min_sites, min_data, miss):
    e, lay, codes, _ = G.make_engine(20, 4, 2400, seed=21, miss_thr=miss)
    e.set_sum_order(1)          # NumPy's order for every window: these tests hold the NumPy-order kernels against the oracle with ==
    wins = [(0, 800), (800, 1100), (1100, 2400)]
    wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
    st = wb.groupDistStats(doPairs=True, minSites=min_sites, minData=min_data)
    for k, (a, b) in enumerate(wins):
        aln = oracle_aln(lay, codes, a, b)
        Do, Co = orc.pair_counts_gemm(aln)
        so, _ = orc.group_dist_stats(aln, Do, Co, True, min_sites, min_data)
        for key, v in so.items():
            assert G.same(st[key][k], v), (key, k, st[key][k], v)        # the sums in NumPy's order (k_popdist_np): to the last bit
    e.close()


@pytest.mark.parametrize("sizes,names,haploid,L", [
    ((3, 7, 12), ("zeta", "alpha", "m"), (), 900),                     # np.unique's order is not the command line's; 6..24 haplotypes
    ((1, 2, 5, 9), ("b", "a", "d", "c"), (0, 4, 9), 400),              # a population of one diploid; haploid samples: odd blocks
    ((40, 33), ("x", "X"), (), 300),                                   # blocks of 6400 / 5280 / 21316 values: several levels of halves
    ((90, 85), ("b", "a"), (3, 100), 120),                             # 178 + 169 haplotypes: 120 409 values = 15 pieces of 8192, 960 runs
    ((2, 2), ("p1", "p0"), (1,), 37),                                  # blocks of fewer than 8 values / tiny windows: ties of the quotients
])
def test_group_dist_stats_to_the_last_bit_in_numpy_order(sizes, names, haploid, L):
    """pi / dxy / Fst against the oracle (which calls np.nanmean on the blocks as the reference does) with ==: unequal populations,
    population names whose sorted order differs from their order on the command line, haploid samples, windows of a few sites"""
    from genomics_general_amd.engine import Engine
    from genomics_general_amd.samples import HapLayout, SampleData
    n_dip = sum(sizes)
    inds = ["s%d" % d for d in range(n_dip)]
    pop_inds, at = [], 0
    for n in sizes:
        pop_inds.append(inds[at:at + n])
        at += n
    ploidy = {nm: (1 if k in haploid else 2) for k, nm in enumerate(inds)}
    lay = HapLayout(SampleData(indNames=list(inds), popNames=list(names), popInds=pop_inds, ploidyDict=ploidy), inds, "phased")
    sid, pos = synth.dense_sites(L, 1)
    sg = np.array([2 * inds.index(nm) + k for nm in lay.ind_order for k in range(len(lay.ind_slots[nm]))], dtype=np.int32)
    codes = synth.gen_codes(99, sid, pos, n_dip, len(sizes), hap_index=sg, var_thr=30000, miss_thr=9000)
    e = Engine(0)
    e.set_layout(lay)
    e.load_sites(codes)
    e.set_sum_order(1)          # NumPy's order for every window: these tests hold the NumPy-order kernels against the oracle with ==
    wins = [(0, L), (0, L // 3), (L // 3, L // 3 + 11), (5, 6), (max(0, L - 40), L)]
    for min_sites, min_data in ((1, 0.01), (8, 0.5)):
        st = e.batch([w[0] for w in wins], [w[1] for w in wins]).groupDistStats(doPairs=True, minSites=min_sites, minData=min_data)
        for k, (a, b) in enumerate(wins):
            aln = oracle_aln(lay, codes, a, b)
            Do, Co = orc.pair_counts_gemm(aln)
            so, _ = orc.group_dist_stats(aln, Do, Co, True, min_sites, min_data)
            for key, v in so.items():
                assert G.same(st[key][k], v), (key, k, (a, b), min_sites, st[key][k], v)
    e.close()


def test_sum_order_can_be_set_through_the_api():
    """pg_set_sum_order: 1 = NumPy's order for every window (a window of 6000 sites == the oracle), 2 = for none (a window of 200
    sites within 1e-9), 0 = by window length again (cli.NP_MAX_SITES = PG_NP_MAX_SITES = 256 sites)"""
    e, lay, codes, _ = G.make_engine(10, 2, 6500, seed=73, var_thr=40000, miss_thr=9000)

    def stats(a, b):
        aln = oracle_aln(lay, codes, a, b)
        Do, Co = orc.pair_counts_gemm(aln)
        return orc.group_dist_stats(aln, Do, Co, True, 5, 0.01)[0]
    long_w, short_w = stats(0, 6000), stats(6000, 6200)
    for mode, same_long, same_short in ((1, True, True), (2, False, False), (0, False, True)):
        e.set_sum_order(mode)
        st = e.batch([0, 6000], [6000, 6200]).groupDistStats(True, 5, 0.01)
        for key in long_w:
            assert (G.same if same_long else G.close)(st[key][0], long_w[key]), (mode, key, st[key][0], long_w[key])
            assert (G.same if same_short else G.close)(st[key][1], short_w[key]), (mode, key, st[key][1], short_w[key])
    e.close()


def test_populations_too_large_for_the_numpy_order_tree_keep_the_fixed_tree():
    """two populations of 500 haplotypes: the (x+y, x+y) block has 10^6 values, more runs than a thread block's LDS holds
    (pg_abi.cpp np_prepare): every window takes the upper-triangle finisher -- within 1e-9 of the oracle"""
    e, lay, codes, _ = G.make_engine(500, 2, 300, seed=72, var_thr=40000, miss_thr=9000)
    st = e.batch([0, 100], [300, 140]).groupDistStats(True, 5, 0.01)
    for k, (a, b) in enumerate([(0, 300), (100, 140)]):
        aln = oracle_aln(lay, codes, a, b)
        Do, Co = orc.pair_counts_gemm(aln)
        so, _ = orc.group_dist_stats(aln, Do, Co, True, 5, 0.01)
        for key, v in so.items():
            assert G.close(st[key][k], v), (key, k, st[key][k], v)
    e.close()


def test_summation_order_is_chosen_window_by_window():
    """windows of up to cli.NP_MAX_SITES (256) sites: NumPy's order (== the oracle); longer ones: the fixed trees (1e-9; the drivers
    compute a window again in NumPy's order where a printed digit could differ); a window's numbers do not depend on the batch it is
    in (pg_popdist_stats, quartet_stats: the kernels skip each other's windows)"""
    from genomics_general_amd.cli import NP_MAX_SITES as T
    e, lay, codes, _ = G.make_engine(12, 4, 9000, seed=71, var_thr=40000, miss_thr=9000)
    wins = [(0, 9000), (0, T - 56), (3000, 3000 + T + 1), (100, 100 + T), (4196, 4196), (8000, 8000 + T // 2), (10, 10 + T + 1), (500, 4596)]
    lo, hi = [w[0] for w in wins], [w[1] for w in wins]
    st = e.batch(lo, hi).groupDistStats(True, 5, 0.01)
    ab = e.batch(lo, hi).ABBABABA("p0", "p1", "p2", "p3", 0.3)
    for k, (a, b) in enumerate(wins):
        one = e.batch([a], [b]).groupDistStats(True, 5, 0.01)
        one_ab = e.batch([a], [b]).ABBABABA("p0", "p1", "p2", "p3", 0.3)
        for key in st:
            assert G.same(st[key][k], one[key][0]), (key, k)
        for key in ab:
            assert G.same(ab[key][k], one_ab[key][0]), (key, k)
        if b == a:
            continue
        short = b - a <= T
        aln = oracle_aln(lay, codes, a, b)
        Do, Co = orc.pair_counts_gemm(aln)
        so, _ = orc.group_dist_stats(aln, Do, Co, True, 5, 0.01)
        for key, v in so.items():
            assert (G.same if short else G.close)(st[key][k], v), (key, k, st[key][k], v)
        want = orc.abbababa(aln, "p0", "p1", "p2", "p3", 0.3)
        for key in ("D", "fd", "fdM", "ABBA", "BABA"):
            assert (G.same if short else G.close)(ab[key][k], want[key]), (key, k, ab[key][k], want[key])
    e.close()


def test_ind_pair_dists_with_and_without_popdist_mask():
    e, lay, codes, names = G.make_engine(9, 3, 1500, seed=33, miss_thr=25000)
    wins = [(0, 700), (700, 1500)]
    for after_popdist in (False, True):
        for same in (False, True):
            wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
            if after_popdist:
                wb.groupDistStats(True, 120, 0.01)
            got = wb.indPairDists(includeSameWithSame=same)
            for k, (a, b) in enumerate(wins):
                aln = oracle_aln(lay, codes, a, b)
                Do, Co = orc.pair_counts_gemm(aln)
                if after_popdist:
                    _, dm = orc.group_dist_stats(aln, Do, Co, True, 120, 0.01)
                else:
                    dm = orc.dist_from_counts(Do, Co)
                want, _ = orc.ind_pair_dists(aln, dm, include_same=same)
                for i in names:
                    for j in names:
                        # rows = the haplotypes of the individual whose name sorts first (what popgenWindows.py reads): to the last
                        # bit; the transposed block of the other orientation is added up in another order
                        assert (G.same if i <= j else G.close)(got[i][j][k], want[i][j]), (after_popdist, same, i, j, k)
    e.close()


@pytest.mark.parametrize("n_dip,min_data,miss", [(16, 0.01, 5000), (16, 0.5, 20000), (16, 0.9, 9000), (16, 0.0, 50000),
                                                 (16, 1.0, 50000),      # no window has a good site: sitesUsed is nan
                                                 (150, 0.3, 9000),      # 304-byte rows: two screening passes
                                                 (300, 0.3, 9000),      # 608-byte rows: three of four
                                                 (600, 0.3, 9000)])     # 1200-byte rows: quad-layout screening
def test_abbababa_sums(n_dip, min_data, miss):
    e, lay, codes, _ = G.make_engine(n_dip, 4, 6000, seed=44, var_thr=45000, miss_thr=miss)
    e.set_sum_order(1)          # NumPy's order for every window: these tests hold the NumPy-order kernels against the oracle with ==
    wins = [(0, 3000), (3000, 3010), (3010, 6000), (10, 10), (2, 4097), (100, 230)]
    wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
    got = wb.ABBABABA("p0", "p1", "p2", "p3", min_data)
    for k, (a, b) in enumerate(wins):
        if b == a:
            assert np.isnan(got["sitesUsed"][k])       # no good site: nan in the reference (genomics.py:1693-1695), not 0
            continue
        want = orc.abbababa(oracle_aln(lay, codes, a, b), "p0", "p1", "p2", "p3", min_data)
        assert G.close(got["sitesUsed"][k], want["sitesUsed"])
        if want["sitesUsed"] > 0:
            for key in ("D", "fd", "fdM", "ABBA", "BABA"):             # the sums in NumPy's order (k_quartet_np): to the last bit
                assert G.same(got[key][k], want[key]), (key, k, got[key][k], want[key])
    e.close()


def test_quartet_sums_in_numpy_order_over_several_pieces_of_8192_used_sites(monkeypatch):
    """a window of 40 000 mostly variable sites forced into NumPy's order: more than 8192 used sites, so the sum is a chain of
    pieces (np.add.reduce's buffer) on top of the pairwise trees -- ABBABABA and fourPop against the oracle with =="""
    monkeypatch.setenv("PG_QUARTET_TREE", "1")
    e, lay, codes, _ = G.make_engine(12, 4, 40000, seed=91, var_thr=62000, miss_thr=2000)
    wins = [(0, 40000), (5, 20005)]
    wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
    got = wb.ABBABABA("p0", "p1", "p2", "p3", 0.3)
    got4 = wb.fourPop("p0", "p1", "p2", "p3", 0.3)
    for k, (a, b) in enumerate(wins):
        aln = oracle_aln(lay, codes, a, b)
        want = orc.abbababa(aln, "p0", "p1", "p2", "p3", 0.3)
        assert want["sitesUsed"] > 8192 and got["sitesUsed"][k] == want["sitesUsed"], want["sitesUsed"]
        for key in ("D", "fd", "fdM", "ABBA", "BABA"):
            assert G.same(got[key][k], want[key]), (key, k, got[key][k], want[key])
        want4 = orc.four_pop(aln, "p0", "p1", "p2", "p3", 0.3, False, False)
        assert got4["sitesUsed"][k] == want4["sitesUsed"]
        for key in orc.FOURPOP_STATS:
            assert G.same(got4[key][k], want4[key]), (key, k, got4[key][k], want4[key])
    e.close()


@pytest.mark.parametrize("mode", ["minor", "polarize", "fixed"])
@pytest.mark.parametrize("n_dip,min_data,miss", [(16, 0.01, 5000), (16, 0.5, 20000), (8, 0.0, 50000), (40, 0.9, 3000)])
def test_fourpop_sums(mode, n_dip, min_data, miss):
    """genomics.fourPop: all 12 statistics + sitesUsed in the three allele-choice modes, incl. the argsort tie rule (8 or 16
    diploids give many 50:50 sites) and 0/0 frequencies under --minData 0"""
    e, lay, codes, _ = G.make_engine(n_dip, 4, 6000, seed=45 + n_dip, var_thr=45000, miss_thr=miss)
    e.set_sum_order(1)          # NumPy's order for every window: these tests hold the NumPy-order kernels against the oracle with ==
    wins = [(0, 3000), (3000, 3010), (3010, 6000), (10, 10), (1, 2100)]
    wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
    got = wb.fourPop("p0", "p1", "p2", "p3", min_data, polarize=mode == "polarize", fixed=mode == "fixed")
    for k, (a, b) in enumerate(wins):
        if b == a:
            assert got["sitesUsed"][k] == 0
            continue
        want = orc.four_pop(oracle_aln(lay, codes, a, b), "p0", "p1", "p2", "p3", min_data, mode == "polarize", mode == "fixed")
        assert got["sitesUsed"][k] == want["sitesUsed"], (k, got["sitesUsed"][k], want["sitesUsed"])
        if want["sitesUsed"] > 0:
            for key in orc.FOURPOP_STATS:
                assert G.same(got[key][k], want[key]), (key, k, got[key][k], want[key])
    e.close()


@pytest.mark.parametrize("n_dip,n_pops,miss", [(10, 2, 400), (24, 6, 150),      # six populations: lanes own pops q and q+4
                                                (150, 3, 20), (300, 3, 10),      # 304- and 608-byte rows
                                                (600, 2, 5)])                    # 1200-byte rows: thread-per-site kernel
def test_group_freq_stats_exact_integers(n_dip, n_pops, miss):
    e, lay, codes, _ = G.make_engine(n_dip, n_pops, 5000, seed=55, miss_thr=miss)
    wins = [(0, 2500), (2500, 5000), (17, 18), (31, 4097), (1000, 1000), (4990, 5000)]
    got = e.batch([w[0] for w in wins], [w[1] for w in wins]).groupFreqStats()
    for k, (a, b) in enumerate(wins):
        want = orc.group_freq_stats(oracle_aln(lay, codes, a, b))
        for key, v in want.items():
            if key.startswith("l_") or key.startswith("S_"):
                g = got[key][k]
                assert (g == v) or (g != g and v != v), (key, k, g, v)
            else:
                # thetaPi is the reference's site-by-site float64 sum (k_popfreq_ordered): bit for bit, and with it thetaW / TajD
                g = float(got[key][k])
                assert g == v or (g != g and v != v), (key, k, got[key][k], v)
    e.close()


def test_site_counts_and_hap_called_exact():
    e, lay, codes, _ = G.make_engine(13, 3, 3000, seed=66, miss_thr=15000, extra_nopop=1)
    cnt = e.batch([0], [1]).siteCounts(100, 2900)
    lut = {1: 0, 2: 1, 4: 2, 8: 3}
    ps = [0] + list(np.cumsum(lay.pop_sizes))
    want = np.zeros_like(cnt)
    sub = codes[100:2900]
    for q in range(lay.n_pops):
        for code, b in lut.items():
            want[:, q, b] = (sub[:, ps[q]:ps[q + 1]] == code).sum(axis=1)
    assert np.array_equal(cnt, want)
    wins = [(0, 3000), (5, 1500), (2999, 3000), (40, 40)]
    called = e.batch([w[0] for w in wins], [w[1] for w in wins]).hapCalled()
    for k, (a, b) in enumerate(wins):
        assert np.array_equal(called[k], (codes[a:b] != 0).sum(axis=0))
    e.close()


def test_device_generator_matches_numpy_spec():
    names, lay = G.make_layout(30, 4)
    from genomics_general_amd.engine import Engine
    e = Engine(0)
    e.set_layout(lay)
    L, scaf_len = 20000, 7000
    e.reserve(L)
    sg = G.slot_gen_hap(names, lay)
    e.synth_fill(0, L, 123456, 20260925, scaf_len, 30, 4, sg, synth.VAR_THR, synth.MISS_THR)
    got = e.download(0, L)
    gi = 123456 + np.arange(L)
    want = synth.gen_codes(20260925, gi // scaf_len, gi % scaf_len + 1, 30, 4, hap_index=sg)
    assert np.array_equal(got, want)
    e.close()


def test_errors_are_loud_not_fatal():
    from genomics_general_amd._lib import PopgenError
    e, lay, codes, _ = G.make_engine(4, 2, 100, seed=1)
    with pytest.raises(PopgenError):
        e.batch([0], [101]).pairCounts()
    with pytest.raises(PopgenError):
        e.batch([50], [40]).hapCalled()
    e.close()


@pytest.mark.parametrize("mode", ["default", "PG_NO_DIP", "PG_PAIR_VALU", "PG_PAIR_TILE=c", "PG_PAIR_TILE=none"])
def test_every_pairwise_code_path_gives_the_same_integers(mode, monkeypatch):
    """diploid fast path (called counts per individual), per-haplotype called counts, popcount kernels, and the matrix-core kernels
    in both forms (LDS-staged blocks, one-wave blocks)"""
    G.set_mode(monkeypatch, mode)
    e, lay, codes, _ = G.make_engine(70, 4, 5000, seed=77, var_thr=9000, miss_thr=4000)
    wins = [(0, 2100), (2100, 4167), (4167, 5000), (13, 14)]
    D, C = e.batch([w[0] for w in wins], [w[1] for w in wins]).pairCounts(reference_order=True)
    for k, (a, b) in enumerate(wins):
        Do, Co = orc.pair_counts_gemm(oracle_aln(lay, codes, a, b))
        assert np.array_equal(C[k], Co) and np.array_equal(D[k], Do), (mode, k)
    e.close()


@pytest.mark.parametrize("n_dip", [31, 33, 64, 65, 97, 128, 129, 160, 161, 190, 193, 224])
def test_called_counts_at_every_tile_count_of_the_one_wave_per_simd_kernel(n_dip):
    """k_pairC_big: one to seven tile rows (one wave up to 14 tiles, two beyond), windows shorter and longer than its LDS ring,
    odd group counts, a one-site window"""
    e, lay, codes, _ = G.make_engine(n_dip, 3, 3300, seed=500 + n_dip, miss_thr=6000)
    wins = [(0, 3300), (0, 129), (5, 1800), (3290, 3300), (1000, 1001), (640, 2432)]
    D, C = e.batch([w[0] for w in wins], [w[1] for w in wins]).pairCounts(reference_order=True)
    for k, (a, b) in enumerate(wins):
        Do, Co = orc.pair_counts_gemm(oracle_aln(lay, codes, a, b))
        assert np.array_equal(C[k], Co) and np.array_equal(D[k], Do), (n_dip, k)
    e.close()


@pytest.mark.parametrize("n_dip", [40, 100, 200])
def test_called_counts_of_one_long_window_are_summed_over_its_parts(n_dip):
    """few long windows are cut into parts that add into the matrix atomically (both wave counts of k_pairC_big)"""
    e, lay, codes, _ = G.make_engine(n_dip, 2, 41000, seed=900 + n_dip, miss_thr=5000)
    D, C = e.batch([0, 20000], [41000, 20500]).pairCounts(reference_order=True)
    for k, (a, b) in enumerate([(0, 41000), (20000, 20500)]):
        Do, Co = orc.pair_counts_gemm(oracle_aln(lay, codes, a, b))
        assert np.array_equal(C[k], Co) and np.array_equal(D[k], Do), (n_dip, k)
    e.close()


def test_half_missing_genotypes_fall_back_to_haplotype_level_called_counts():
    """phased data such as `A|N`: the two haplotypes of an individual differ in calledness, so the per-individual
    called-count shortcut must be abandoned (and the result still be exact)"""
    from genomics_general_amd.engine import Engine
    names, lay = G.make_layout(21, 3)
    sid, pos = synth.dense_sites(3000, 1)
    codes = synth.gen_codes(5, sid, pos, 21, 3, hap_index=G.slot_gen_hap(names, lay), var_thr=30000, miss_thr=3000).copy()
    rng = np.random.default_rng(0)
    knock = rng.random(codes.shape) < 0.02
    codes[knock] = 0
    e = Engine(0)
    e.set_layout(lay)
    e.load_sites(codes)
    e.set_sum_order(1)          # NumPy's order for every window: these tests hold the NumPy-order kernels against the oracle with ==
    wins = [(0, 1500), (1500, 3000)]
    wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
    D, C = wb.pairCounts(reference_order=True)
    st = wb.groupDistStats(True, 5, 0.01)
    for k, (a, b) in enumerate(wins):
        aln = oracle_aln(lay, codes, a, b)
        Do, Co = orc.pair_counts_gemm(aln)
        assert np.array_equal(C[k], Co) and np.array_equal(D[k], Do)
        so, _ = orc.group_dist_stats(aln, Do, Co, True, 5, 0.01)
        for key, v in so.items():
            assert G.same(st[key][k], v), (key, k)
    e.close()


def test_dense_polymorphism_and_multiallelic_sites():
    """every site variable, many third alleles: the compaction emits whole words"""
    e, lay, codes, _ = G.make_engine(12, 2, 2500, seed=88, var_thr=65536, miss_thr=9000)
    D, C = e.batch([0, 700], [700, 2500]).pairCounts(reference_order=True)
    for k, (a, b) in enumerate([(0, 700), (700, 2500)]):
        Do, Co = orc.pair_counts_gemm(oracle_aln(lay, codes, a, b))
        assert np.array_equal(C[k], Co) and np.array_equal(D[k], Do)
    e.close()


def test_rccl_communicator_single_rank_roundtrip(tmp_path, monkeypatch):
    """the RCCL path of the C-ABI (lazy dlopen, unique id, comm init, all-gather, barrier) with one rank on one GPU"""
    from genomics_general_amd import dist
    from genomics_general_amd.engine import Engine
    monkeypatch.setenv("PG_RDZV_FILE", str(tmp_path / "rdzv"))
    names, lay = G.make_layout(4, 2)
    e = Engine(0)
    e.set_layout(lay)
    comm = dist.RcclComm(e, dist.World(0, 1, 0))
    x = np.arange(12, dtype=np.float64) * 0.5 - 1
    x[3] = np.nan
    got = comm.allgather(x)
    assert got.shape == (1, 12) and np.array_equal(np.isnan(got[0]), np.isnan(x)) and np.allclose(np.nan_to_num(got[0]), np.nan_to_num(x))
    comm.barrier()
    tab = dist.gather_table(comm, x.reshape(4, 3), 4)
    assert tab.shape == (4, 3)
    e.close()


@pytest.mark.parametrize("after_popdist,after_indpair,max_dist,var_thr", [
    (False, False, 0, 2500), (True, False, 0, 2500), (False, True, 0.002, 6000), (True, True, 0.01, 20000)])
def test_sample_het_and_h12_finished_on_the_device_match_the_oracle(after_popdist, after_indpair, max_dist, var_thr):
    """pg_sample_het / pg_hapstats (genomics.py:918-929, 1079-1098, 1239-1261) on the matrix the reference's worker has cached
    at that point: plain, masked by a preceding groupDistStats(minSites), or with the nan diagonal indPairDists leaves"""
    e, lay, codes, names = G.make_engine(17, 3, 2600, seed=71, var_thr=var_thr, miss_thr=2500, extra_nopop=2)
    wins = [(0, 300), (300, 340), (340, 2600), (7, 8), (1000, 1064)]
    ms = 25
    wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
    if after_popdist:
        wb.groupDistStats(True, ms, 0.01)
    if after_indpair:
        wb.indPairDists(includeSameWithSame=False)
    het = wb.sampleHet()
    h = wb.H12stats(max_dist)
    sizes_seen = set()
    for k, (a, b) in enumerate(wins):
        aln = oracle_aln(lay, codes, a, b)
        Do, Co = orc.pair_counts_gemm(aln)
        if after_popdist:
            _, dm = orc.group_dist_stats(aln, Do, Co, True, ms, 0.01)
        else:
            dm = orc.dist_from_counts(Do, Co)
        if after_indpair:
            np.fill_diagonal(dm, np.nan)
        want_het = orc.sample_het(aln, dm, Co)
        for nm in names:
            assert G.close(het[nm][k], want_het["het_" + nm]), (nm, k, het[nm][k], want_het["het_" + nm])
        want_h = orc.h12_stats(aln, dm, max_dist)
        for p in lay.sampleData.popNames:
            for st in ("H1_", "H12_", "H2_"):
                assert G.close(h[st + p][k], want_h[st + p]), (st + p, k, h[st + p][k], want_h[st + p])
            sizes_seen.add(round(float(want_h["H1_" + p]), 6))
    assert len(sizes_seen) >= 2                     # the cases are not all "every haplotype its own cluster"
    e.close()


@pytest.mark.parametrize("fixture,fmt", [("c1", "phased"), ("sparse", "phased"), ("abba_pairs", "pairs"), ("abba_diplo", "diplo"),
                                         ("haplo", "haplo")])
def test_device_tokenizer_equals_the_host_tokenizer(fixture, fmt):
    """pg_tokenize_text (K0 on the device) against pg_encode_text on the reference-golden fixtures in the four genotype formats:
    rows, positions, scaffold runs; with a layout that reorders and drops samples; written behind rows that are already there"""
    import gzip
    import os
    from genomics_general_amd import genoio
    from genomics_general_amd.engine import Engine
    from genomics_general_amd.samples import HapLayout, SampleData
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raw = gzip.open(os.path.join(gold, fixture + ".geno.gz"), "rb").read()
    names = raw[:raw.index(b"\n")].decode().split()[2:]
    body = raw[raw.index(b"\n") + 1:]
    pl = {nm: (1 if fmt == "haplo" else 2) for nm in names}
    for order in (list(names), names[3:] + names[1:2]):
        lay = HapLayout(SampleData(indNames=order, ploidyDict={nm: pl[nm] for nm in order}), names, fmt)
        want = genoio.encode(body, lay)
        e = Engine(0)
        e.set_layout(lay)
        e.reserve(want.n_sites + 100)
        got = e.tokenize_text(body, row_offset=37)
        assert got is not None, "regular fixture refused by the device tokenizer"
        n, pos, starts, run_names = got
        assert n == want.n_sites and np.array_equal(pos, want.pos)
        assert np.array_equal(e.download(37, n), want.gt)
        assert np.array_equal(starts, want.run_starts) and run_names == want.run_names
        tiny = e.tokenize_text(body, row_offset=37, max_runs=1)                  # more runs than room: asked again with room
        assert tiny is not None and np.array_equal(tiny[2], want.run_starts)
        e.close()


def test_device_tokenizer_in_three_steps_two_blocks_in_flight(tmp_path):
    """pg_tokenize_submit / _parse / _collect in the order the drivers' ingestion thread uses -- parse(k), submit(k+1), collect(k): the
    kernels of one block run while the text of the next crosses PCIe -- from a file (pread by the staging threads) and from memory,
    with an upper bound of the rows instead of a count; equal to the host tokenizer block by block.  An irregular block is refused
    at submit or at collect and leaves the slots usable"""
    import gzip
    import os
    from genomics_general_amd import genoio
    from genomics_general_amd.engine import Engine
    from genomics_general_amd.samples import HapLayout, SampleData
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raw = gzip.open(os.path.join(gold, "abba.geno.gz"), "rb").read()
    names = raw[:raw.index(b"\n")].decode().split()[2:]
    head = raw.index(b"\n") + 1
    path = str(tmp_path / "abba.geno")
    with open(path, "wb") as f:
        f.write(raw)
    lines = raw[head:].split(b"\n")[:-1]
    cuts = [0, 700, 1500, 1501, 4000, len(lines)]
    blocks = [b"\n".join(lines[a:b]) + b"\n" for a, b in zip(cuts[:-1], cuts[1:])]
    offs = [head + sum(len(x) for x in blocks[:k]) for k in range(len(blocks))]
    lay = HapLayout(SampleData(indNames=names[2:] + names[:1]), names, "phased")
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(len(lines) + 64)
    fd = os.open(path, os.O_RDONLY)
    for from_file in (True, False):
        src = lambda k: dict(file=(fd, offs[k])) if from_file else {}             # noqa: E731
        row, got = 5, []
        assert e.tokenize_submit(blocks[0], 0, **src(0))
        for k in range(len(blocks)):
            bound = len(blocks[k]) // (4 * len(names) + 4) + 1
            n = e.tokenize_parse(k % 2, row, bound)
            assert n == cuts[k + 1] - cuts[k]
            if k + 1 < len(blocks):
                assert e.tokenize_submit(blocks[k + 1], (k + 1) % 2, **src(k + 1))
            got.append(e.tokenize_collect(k % 2, blocks[k], n))
            row += n
        row = 5
        for k, g in enumerate(got):
            want = genoio.encode(blocks[k], lay)
            assert g is not None and g[0] == want.n_sites and np.array_equal(g[1], want.pos)
            assert np.array_equal(g[2], want.run_starts) and g[3] == want.run_names
            assert np.array_equal(e.download(row, g[0]), want.gt), (from_file, k)
            row += g[0]
    os.close(fd)
    dirty = b"\n".join(lines[:50] + [b"# a comment"] + lines[50:60]) + b"\n"
    assert e.tokenize_submit(dirty, 1)                                           # the first line is regular: the kernels find the comment
    n = e.tokenize_parse(1, 0, 100)
    assert n is not None and e.tokenize_collect(1, dirty, n) is None
    assert not e.tokenize_submit(blocks[0][:-1], 0)                              # no final line feed: refused at once
    assert e.tokenize_submit(blocks[0], 0) and e.tokenize_parse(0, 0, 3) is None   # rows that do not fit the bound
    assert e.tokenize_submit(blocks[0], 0)
    n = e.tokenize_parse(0, 0, 5000)
    g = e.tokenize_collect(0, blocks[0], n)
    assert g is not None and g[0] == cuts[1]
    e.close()


def test_device_tokenizer_refuses_irregular_blocks():
    """a comment line, doubled separators, a cell of another width than its column's, a missing final line feed: the fast path says
    no (the drivers then use the host tokenizer); a position that is not a number likewise.  Mixed ploidy (narrower cells for the
    haploid samples) IS regular: the widths are per column -- unless the declared ploidies do not fit the cells"""
    import gzip
    import os
    from genomics_general_amd.engine import Engine
    from genomics_general_amd.samples import HapLayout, SampleData
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raw = gzip.open(os.path.join(gold, "c1.geno.gz"), "rb").read()
    names = raw[:raw.index(b"\n")].decode().split()[2:]
    body = raw[raw.index(b"\n") + 1:]
    lines = body.split(b"\n")
    lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(len(lines) + 10)
    assert e.tokenize_text(body) is not None
    bad = [b"\n".join(lines[:50] + [b"# a comment"] + lines[50:]),
           b"\n".join(lines[:50] + [lines[50].replace(b"\t", b"\t\t", 3)] + lines[51:]),
           b"\n".join(lines[:50] + [lines[50][:-3] + b"A"] + lines[51:]),
           body[:-1],
           b"\n".join(lines[:7] + [lines[7].replace(lines[7].split()[1], b"12x")] + lines[8:])]
    for k, text in enumerate(bad):
        assert e.tokenize_text(text) is None, k
    mixed = gzip.open(os.path.join(gold, "mixed.geno.gz"), "rb").read()
    mnames = mixed[:mixed.index(b"\n")].decode().split()[2:]
    ml = HapLayout(SampleData(indNames=list(mnames), ploidyDict={nm: (1 if nm in ("s1", "s6", "s9") else 2) for nm in mnames}), mnames, "phased")
    from genomics_general_amd import genoio
    mbody = mixed[mixed.index(b"\n") + 1:]
    for order in (list(mnames), mnames[4:] + mnames[1:2]):                     # all samples; a subset that starts with diploids
        ml = HapLayout(SampleData(indNames=order, ploidyDict={nm: (1 if nm in ("s1", "s6", "s9") else 2) for nm in order}), mnames, "phased")
        want = genoio.encode(mbody, ml)
        e.set_layout(ml)
        e.reserve(5000)
        got = e.tokenize_text(mbody, row_offset=11)
        assert got is not None and got[0] == want.n_sites and np.array_equal(got[1], want.pos)
        assert np.array_equal(e.download(11, got[0]), want.gt) and got[3] == want.run_names
    # every sample declared diploid: the haploid samples' cells are too narrow for that
    e.set_layout(HapLayout(SampleData(indNames=list(mnames)), mnames, "phased"))
    e.reserve(5000)
    assert e.tokenize_text(mbody) is None
    # one line whose haploid cell became diploid (another width than the block's first line shows)
    ml = HapLayout(SampleData(indNames=list(mnames), ploidyDict={nm: (1 if nm in ("s1", "s6", "s9") else 2) for nm in mnames}), mnames, "phased")
    e.set_layout(ml)
    e.reserve(5000)
    mlines = mbody.split(b"\n")
    cells = mlines[40].split(b"\t")
    cells[3] = b"A/C"                                                           # s1 is haploid
    assert e.tokenize_text(b"\n".join(mlines[:40] + [b"\t".join(cells)] + mlines[41:])) is None
    e.close()


def test_device_tokenizer_line_prefixes_beyond_the_fast_path():
    """scaffold names longer than the 64 bytes the ballots see, blanks in front of the scaffold, several blanks between scaffold
    and position, signed positions: the byte-by-byte walk gives what the host tokenizer gives, also next to fast-path lines"""
    import gzip
    import os
    from genomics_general_amd import genoio
    from genomics_general_amd.engine import Engine
    from genomics_general_amd.samples import HapLayout, SampleData
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raw = gzip.open(os.path.join(gold, "c1.geno.gz"), "rb").read()
    names = raw[:raw.index(b"\n")].decode().split()[2:]
    lines = raw[raw.index(b"\n") + 1:].split(b"\n")[:-1]
    long_a, long_b = b"scaffold_" + b"x" * 70 + b"_a", b"scaffold_" + b"x" * 70 + b"_b"
    out = []
    for k, ln in enumerate(lines):
        scaf, pos, cells = ln.split(None, 2)
        if 100 <= k < 160:
            scaf = long_a if k < 130 else long_b
        if 200 <= k < 230:
            ln = b" " + scaf + b"\t" + pos + b"\t" + cells
        elif 300 <= k < 330:
            ln = scaf + b" \t " + pos + b"  " + cells
        elif 400 <= k < 410:
            ln = scaf + b"\t+" + pos + b"\t" + cells
        elif k == 420:
            ln = b"q" * 50 + b"\t" + pos + b"\t" + cells           # the position straddles the 64-byte window
        else:
            ln = scaf + b"\t" + pos + b"\t" + cells
        out.append(ln)
    body = b"\n".join(out) + b"\n"
    lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
    want = genoio.encode(body, lay)
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(want.n_sites)
    got = e.tokenize_text(body)
    assert got is not None
    n, pos, starts, run_names = got
    assert n == want.n_sites and np.array_equal(pos, want.pos) and np.array_equal(e.download(0, n), want.gt)
    assert np.array_equal(starts, want.run_starts) and run_names == want.run_names
    e.close()


def test_uploads_queued_right_behind_a_growing_reservation_are_not_wiped():
    """Regression (commit 2f55857): pg_reserve_sites clears the new rows on the context's stream; an asynchronous upload or the
    device tokenizer, queued on the copy stream right after a GROWING reservation, could be overtaken by that clearing -- rare
    wrong (zeroed) rows in the streaming drivers.  200 growing reservations, each followed at once by an upload of known rows
    through one of the three routes; every row is read back and compared."""
    from genomics_general_amd.engine import Engine
    names, lay = G.make_layout(16, 2)
    H = lay.n_hap
    rng = np.random.default_rng(5)
    e = Engine(0)
    e.set_layout(lay)
    pitch = e.row_pitch
    n_rows = 4096
    rows = (1 << rng.integers(0, 4, size=(n_rows, H))).astype(np.int8)
    rows[rng.random((n_rows, H)) < 0.05] = 0
    staged = e.pinned.empty((n_rows, pitch), np.int8)
    staged[:] = 0
    staged[:, :H] = rows
    # the same rows as `.geno` text for the device tokenizer (phased diploid cells)
    letters = np.array(list("NACNGNNNT"))
    lines = []
    for k in range(n_rows):
        c = letters[rows[k]]
        lines.append("sc1\t%d\t%s\n" % (k + 1, "\t".join("%s|%s" % (c[2 * d], c[2 * d + 1]) for d in range(H // 2))))
    text = "".join(lines).encode()
    bad = 0
    for it in range(200):
        # a reservation larger than every one before: the rows are re-allocated and cleared (tens of megabytes: the clearing is
        # still running when the next call is queued)
        e.reserve(600_000 + 3_000 * it)
        off = int(rng.integers(0, 1000))
        route = it % 3
        if route == 0:
            e.upload_async(staged, off)
            e.upload_wait()
        elif route == 1:
            got = e.tokenize_text(text, row_offset=off, n_rows=n_rows)
            assert got is not None and got[0] == n_rows
        else:
            e.upload(rows, off)
        back = e.download(off, n_rows)
        bad += int(not np.array_equal(back, rows))
    e.close()
    assert bad == 0, "%d of 200 uploads lost rows to the reservation's clearing" % bad


@pytest.mark.parametrize("include_same", [False, True])
def test_individual_pair_means_from_supplied_counts_equal_the_fused_path(include_same):
    """pg_indpairdist_mean_from_counts on the counts pg_pairwise hands out == pg_indpairdist_mean on the same windows, and
    the counts of the two halves of a window add up to the window's (what the ranks of a `cat` run rely on)"""
    e, lay, codes, _ = G.make_engine(23, 3, 4000, seed=4242, miss_thr=9000)
    wins = [(0, 4000), (0, 1777), (1777, 4000), (10, 11)]
    wb = e.batch([w[0] for w in wins], [w[1] for w in wins])
    D, C = wb.pairCounts(reference_order=False)
    want = wb.indPairTable(includeSameWithSame=include_same)
    got = e.indPairTableFromCounts(D, C, includeSameWithSame=include_same)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(D[1] + D[2], D[0]) and np.array_equal(C[1] + C[2], C[0])
    summed = e.indPairTableFromCounts((D[1] + D[2])[None], (C[1] + C[2])[None], includeSameWithSame=include_same)
    assert np.array_equal(summed[0], want[0], equal_nan=True)
    e.close()


def test_populations_that_split_an_individual_take_the_general_finaliser():
    """pg_set_samples takes per-haplotype populations; a C-ABI caller may put the two haplotypes of a diploid
    individual into different populations (odd population boundaries).  The individual-wise fast forms of k_popdist_fin (8-byte
    loads at even slots) must not run then: sums of D / C per population pair == the plain sums over the pair matrices"""
    import ctypes as C
    from genomics_general_amd import _lib
    from genomics_general_amd._lib import check
    L = _lib.lib()
    rng = np.random.default_rng(77)
    n_hap, n_sites = 16, 900
    hap_pop = np.array([0] * 5 + [1] * 4 + [2] * 7, dtype=np.int32)           # boundaries at slots 5 and 9: inside individuals 2 and 4
    hap_sample = np.repeat(np.arange(8), 2).astype(np.int32)
    codes = (1 << rng.integers(0, 2, size=(n_sites, n_hap))).astype(np.int8)
    miss = rng.random((n_sites, n_hap // 2)) < 0.1
    codes[np.repeat(miss, 2, axis=1)] = 0                                    # a genotype is missing as a whole
    h = C.c_void_p()
    check(L.pg_ctx_create(C.byref(h), 0))
    check(L.pg_set_samples(h, n_hap, hap_pop, hap_sample, 3))
    check(L.pg_reserve_sites(h, n_sites))
    check(L.pg_upload_sites(h, 0, np.ascontiguousarray(codes), n_sites))
    lo, hi = np.array([0, 300], dtype=np.int64), np.array([300, 900], dtype=np.int64)
    D = np.zeros((2, n_hap, n_hap), dtype=np.int32)
    Cc = np.zeros((2, n_hap, n_hap), dtype=np.int32)
    check(L.pg_pairwise(h, lo, hi, 2, D, Cc))
    sums, cnts = np.zeros((2, 6)), np.zeros((2, 6), dtype=np.int64)
    check(L.pg_popdist(h, lo, hi, 2, 1, sums, cnts))
    L.pg_ctx_destroy(h)
    start = [0, 5, 9, 16]
    for w in range(2):
        k = 0
        for x in range(3):
            for y in range(x, 3):
                tot, n = 0.0, 0
                for i in range(start[x], start[x + 1]):
                    for j in range(start[y], start[y + 1]):
                        if (x == y and j <= i) or Cc[w, i, j] < 1:
                            continue
                        tot += D[w, i, j] / Cc[w, i, j]
                        n += 1
                assert cnts[w, k] == n and abs(sums[w, k] - tot) < 1e-9, (w, x, y, sums[w, k], tot, cnts[w, k], n)
                k += 1


@pytest.mark.parametrize("n_pops,seed,var_thr,miss_thr", [(4, 5, 30000, 5000), (2, 6, 50000, 20000), (3, 7, 12000, 40000), (5, 8, 60000, 0)])
def test_site_target_equals_the_numpy_form_of_freq_py(n_pops, seed, var_thr, miss_thr):
    """pg_site_target (k_site_counts + k_site_target: freq.py's target allele, counts / frequencies rounded to 4 places, --threshold,
    keep flags) against the NumPy statements of freq.py:60-105 / genomics.py:636-669 (tests/cpu_engine.py: siteTarget): bit for bit"""
    import cpu_engine
    e, lay, codes, _ = G.make_engine(4 * n_pops + 1, n_pops, 20000, seed=seed, var_thr=var_thr, miss_thr=miss_thr, extra_nopop=1)
    ce = cpu_engine.CpuEngine()
    ce.set_layout(lay)
    ce.reserve(len(codes))
    ce.gt[:len(codes)] = codes
    for target in ("derived", "minor"):
        for as_counts, thr, min_data in ((False, None, 0.0), (True, None, 0.0), (False, 0.3, 2.0), (False, None, 5.5), (True, None, 3.0),
                                         (False, 1.0, 0.0)):
            for a, b in ((0, 20000), (17, 4113), (19999, 20000)):
                vals, keep = e.batch([0], [0]).siteTarget(a, b, target, min_data, as_counts, thr)
                want, wkeep = ce.batch([0], [0]).siteTarget(a, b, target, min_data, as_counts, thr)
                assert vals.dtype == want.dtype and np.array_equal(keep, wkeep), (target, as_counts, thr, min_data)
                if as_counts:
                    assert np.array_equal(vals, want)
                else:                                       # the same float64 values (0.0 against -0.0 cannot occur: frequencies), NaN where NaN
                    assert np.array_equal(np.isnan(vals), np.isnan(want))
                    assert np.array_equal(np.nan_to_num(vals, nan=-1.0), np.nan_to_num(want, nan=-1.0))
                assert 0 < keep.sum() <= len(keep) or (b - a) == 1
    from genomics_general_amd._lib import PopgenError
    with pytest.raises(PopgenError):
        e.batch([0], [0]).siteTarget(0, 30000, "minor")
    e.close()


def test_deferred_result_tables_equal_the_immediate_ones():
    """pg_set_deferred_results: the table of call k is copied into its page-locked array while call k + 1 computes (two device-side
    buffers alternate); after results_wait() every table equals the one the immediate route returns -- also when a pageable array is
    passed (then the call behaves as before) and when the table grows"""
    e, lay, codes, _ = G.make_engine(400, 1, 6000, seed=9, var_thr=30000, miss_thr=8000)
    wins = [([0, 3000], [3000, 6000]), ([0, 100, 5000], [6000, 2100, 6000]), ([17], [5999]), ([0, 2000, 4000], [2000, 4000, 6000])]
    want = [e.batch(lo, hi).indPairTable().copy() for lo, hi in wins]
    assert want[0].nbytes >= (1 << 20)                     # (large enough to land in page-locked memory)
    e.set_deferred_results(True)
    for rep in range(3):
        got = [e.batch(lo, hi).indPairTable() for lo, hi in wins * 2]
        e.results_wait()
        for k, g in enumerate(got):
            w = want[k % len(wins)]
            assert g.shape == w.shape and np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(np.nan_to_num(g), np.nan_to_num(w)), (rep, k)
    small = e.batch([0], [50]).indPairTable()              # below the page-locked threshold? (n_samp = 400: 80 200 pairs = 0.6 MB: pageable)
    e.sync()
    e.set_deferred_results(False)
    assert np.array_equal(np.nan_to_num(small), np.nan_to_num(e.batch([0], [50]).indPairTable()))
    e.close()
