"""Helpers shared by the -m gpu tests."""
import numpy as np

from genomics_general_amd import synth
from genomics_general_amd.engine import Engine
from genomics_general_amd.samples import HapLayout, SampleData


def make_layout(n_dip, n_pops, extra_nopop=0, fmt="phased"):
    names = ["s%d" % d for d in range(n_dip)]
    n_in = n_dip - extra_nopop
    per = n_in // n_pops
    pop_inds = [names[k * per:(k + 1) * per] for k in range(n_pops)]
    pop_inds[-1] += names[n_pops * per:n_in]
    sd = SampleData(indNames=list(names), popNames=["p%d" % k for k in range(n_pops)], popInds=pop_inds)
    return names, HapLayout(sd, names, fmt)


def slot_gen_hap(names, lay):
    return np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(len(lay.ind_slots[nm]))], dtype=np.int32)


def make_engine(n_dip, n_pops, L, seed, var_thr=30000, miss_thr=5000, extra_nopop=0, n_scaf=1):
    names, lay = make_layout(n_dip, n_pops, extra_nopop)
    sid, pos = synth.dense_sites(L, n_scaf)
    sg = slot_gen_hap(names, lay)
    codes = synth.gen_codes(seed, sid, pos, n_dip, n_pops, hap_index=sg, var_thr=var_thr, miss_thr=miss_thr)
    e = Engine(0)
    e.set_layout(lay)
    e.load_sites(codes)
    return e, lay, codes, names


def same(a, b):
    """bit for bit: equal values of equal sign (-0.0 is not 0.0), or both nan"""
    a, b = np.float64(a), np.float64(b)
    return bool((a == b and np.signbit(a) == np.signbit(b)) or (np.isnan(a) and np.isnan(b)))


def close(a, b, tol=1e-9):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    return bool(np.all(ok | both_nan | same_inf))


def set_mode(monkeypatch, mode):
    """`NAME` or `NAME=value`: an environment switch of the library (A/B code paths); "default" sets nothing"""
    if mode != "default":
        name, _, val = mode.partition("=")
        monkeypatch.setenv(name, val or "1")
