#!/usr/bin/env python
"""Placement study of the pack kernel (C2 shape unless `northstar` is given): its time moves by +-7 % between boxes, processes and
allocations.  (1) several engines in ONE process, each with its own buffers: addresses and kernel time; (2) in one engine, the
called plane / the XV planes / the resident rows shifted by a few offsets (pg_debug_place): does the relative placement of the
buffers explain the spread?

    python tools/pack_variance.py [northstar] [n_engines]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from genomics_general_amd import _lib, synth, windows                       # noqa: E402
from genomics_general_amd._lib import check                                  # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

big = "northstar" in sys.argv
n_eng = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 3
n_dip, n_pops, n_sites, n_scaf, wind = (200, 4, 100_000_000, 4, 50_000) if big else (100, 4, 10_000_000, 4, 50_000)
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // n_pops
sd = SampleData(popNames=["pop%d" % k for k in range(n_pops)], popInds=[names[k * per:(k + 1) * per] for k in range(n_pops)])
lay = HapLayout(sd, names, "phased")
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
scaf_len = n_sites // n_scaf
run_starts = np.arange(n_scaf, dtype=np.int64) * scaf_len
positions = np.tile(np.arange(1, scaf_len + 1, dtype=np.int32), n_scaf)
T = windows.coord_windows(run_starts, ["chr%d" % (k + 1) for k in range(n_scaf)], positions, wind, wind)
del positions
L = _lib.lib()


def fill(e):
    e.reserve(n_sites)
    e.synth_fill(0, n_sites, 0, synth.SEED_DEFAULT, scaf_len, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR)


def addr(e, which):
    a, b = C.c_uint64(0), C.c_uint64(0)
    check(L.pg_debug_address(e._h, which, C.byref(a), C.byref(b)))
    return a.value, b.value


def pack_ms(e, passes=6):
    for _ in range(2):
        e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    e.kernel_time_reset()
    for _ in range(passes):
        e.batch(T.lo, T.hi).groupDistTable(True, 100, 0.01)
    e.sync()
    ms, n = e.kernel_time(_lib.K_PACK)
    return ms / n


def read_ms(e, passes=4):
    """a read-only stream over the same rows: k_popfreq_q (screening pass: every row once, no plane stores)"""
    e.batch(T.lo, T.hi).groupFreqStats()
    e.sync()
    e.kernel_time_reset()
    for _ in range(passes):
        e.batch(T.lo, T.hi).groupFreqStats()
    e.sync()
    ms, n = e.kernel_time(_lib.K_SITESTATS)
    return ms / n


def show(tag, e):
    t = pack_ms(e)
    tag = "%s [read-only stream %.4f ms]" % (tag, read_ms(e))
    a = [addr(e, w)[0] for w in range(3)]
    print("%-62s pack %.4f ms   gt %#x  Vp %#x  XV %#x   (Vp-gt) mod 1MiB %#x  (XV-gt) mod 1MiB %#x" % (
        tag, t, a[0], a[1], a[2], (a[1] - a[0]) & 0xFFFFF, (a[2] - a[0]) & 0xFFFFF), flush=True)
    return t


engines = []
for k in range(n_eng):
    e = Engine(0)
    e.set_layout(lay)
    fill(e)
    engines.append(e)                                                       # kept alive: later engines get other addresses
    show("engine %d" % k, e)
    show("engine %d again" % k, e)
e = engines[0]
for which, name in ((1, "Vp"), (2, "XV")):
    for lead in (65536, 0):
        check(L.pg_debug_place(e._h, which, lead))
        show("engine 0, %s shifted by %d" % (name, lead), e)
for lead in (4096, 128 << 10, (2 << 20) + 4096, 0, (4 << 20), 0, 8192, 0):
    check(L.pg_debug_place(e._h, 0, lead))
    fill(e)
    show("engine 0, rows shifted by %d" % lead, e)
