"""-m gpu: size-independent properties at BASELINE.json's C2 size (10^7 sites x 200 haplotypes, 50 kb windows), where the
CPU oracle is far too slow to run: run-to-run bit identity, additivity of the integer matrices over a split window, agreement of
the independent pairwise pipelines, bounds and symmetry."""
import numpy as np
import pytest

from genomics_general_amd import synth
from genomics_general_amd.engine import Engine

import gpu_util as G

pytestmark = pytest.mark.gpu

N_SITES, N_DIP, N_POPS, WIND = 10_000_000, 100, 4, 50_000


@pytest.fixture(scope="module")
def c2():
    names, lay = G.make_layout(N_DIP, N_POPS)
    e = Engine(0)
    e.set_layout(lay)
    e.reserve(N_SITES)
    e.synth_fill(0, N_SITES, 0, synth.SEED_DEFAULT, N_SITES // 4, N_DIP, N_POPS, G.slot_gen_hap(names, lay), synth.VAR_THR, synth.MISS_THR)
    lo = np.arange(0, N_SITES, WIND, dtype=np.int64)
    yield e, lay, lo, lo + WIND
    e.close()


def test_full_size_statistics_are_bit_identical_run_to_run(c2):
    e, lay, lo, hi = c2
    a = e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    b = e.batch(lo, hi).groupDistStats(True, 100, 0.01)
    assert len(a) == 4 + 2 * 12
    for k in a:
        assert a[k].shape == (200,) and np.array_equal(a[k], b[k]), k
        assert np.all(np.isfinite(a[k]))
    # pi within [0,1], dxy >= 0, Fst <= 1
    for k, v in a.items():
        if k.startswith("pi_") or k.startswith("dxy_"):
            assert np.all((v >= 0) & (v <= 1))
        else:
            assert np.all(v <= 1)


def test_integer_matrices_are_additive_over_a_split_window_and_bounded(c2):
    e, lay, lo, hi = c2
    a, c = int(lo[37]), int(hi[37])
    b = a + 20011                                           # not a multiple of 32: exercises word tails
    D, C = e.batch([a, a, b], [c, b, c]).pairCounts(reference_order=False)
    assert np.array_equal(D[0], D[1] + D[2]) and np.array_equal(C[0], C[1] + C[2])
    assert np.array_equal(D[0], D[0].T) and np.array_equal(C[0], C[0].T)
    assert D[0].max() <= C[0].max() <= c - a and np.all(D[0] <= C[0]) and np.all(np.diag(C[0]) == 0)
    assert C[0].sum() > 0 and D[0].sum() > 0


@pytest.mark.parametrize("env", ["PG_PAIR_V1", "PG_NO_DIP", "PG_OVERLAP"])
def test_independent_pipelines_agree_at_full_window_size(c2, env, monkeypatch):
    e, lay, lo, hi = c2
    sel = [0, 61, 199]
    want = e.batch(lo[sel], hi[sel]).pairCounts(reference_order=False)
    st_want = e.batch(lo[:40], hi[:40]).groupDistStats(True, 100, 0.01)
    monkeypatch.setenv(env, "1")
    got = e.batch(lo[sel], hi[sel]).pairCounts(reference_order=False)
    st_got = e.batch(lo[:40], hi[:40]).groupDistStats(True, 100, 0.01)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    for k in st_want:
        assert np.array_equal(st_got[k], st_want[k]), k


def test_abbababa_and_popfreq_full_size_determinism(c2):
    e, lay, lo, hi = c2
    a = e.batch(lo, hi).ABBABABA("p0", "p1", "p2", "p3", 0.01)
    b = e.batch(lo, hi).ABBABABA("p0", "p1", "p2", "p3", 0.01)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.all(a["sitesUsed"] > 100) and np.all(np.abs(a["D"]) <= 1)
    f = e.batch(lo[:20], hi[:20]).groupFreqStats()
    g = e.batch(lo[:20], hi[:20]).groupFreqStats()
    for k in f:
        assert np.array_equal(f[k], g[k], equal_nan=True), k
