#!/usr/bin/env python
"""Where does the start-up of a run go (the device context is 41 % of a mid-size `.geno.gz` run)?
A fresh process: loading the library (dlopen + registration of the code objects), the runtime's start-up (hipGetDeviceCount),
hipSetDevice + streams, the first allocation, and the FIRST launch of a kernel of every translation unit (HIP loads a
translation unit's code object when one of its kernels is first used): pg_kernels.hip (k_synth), the pack + pair kernels
(pg_pair2 / pg_pair_big / pg_pair_mfma), the tokenizer and the inflate kernels.  Run it a few times; pass environment switches
to compare (HIP_ENABLE_DEFERRED_LOADING=0, ...).

    python tools/ctx_time.py"""
import ctypes as C
import json
import os
import sys
import time

t0 = time.perf_counter()
import numpy as np                                                           # noqa: E402
t_numpy = time.perf_counter() - t0
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
t0 = time.perf_counter()
from genomics_general_amd import _lib                                        # noqa: E402
L = _lib.lib()
t_dlopen = time.perf_counter() - t0
from genomics_general_amd import genoio, synth                               # noqa: E402
from genomics_general_amd.engine import Engine                               # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData               # noqa: E402

rec = {"import_numpy_s": round(t_numpy, 4), "load_library_s": round(t_dlopen, 4)}


def lap(name, fn):
    t = time.perf_counter()
    r = fn()
    rec[name] = round(time.perf_counter() - t, 4)
    return r


e = lap("engine_create_s", lambda: Engine(0))
tm = (C.c_double * 3)()
L.pg_ctx_create_times(tm)
rec["engine_create_split_s"] = {"hipGetDeviceCount": round(tm[0], 4), "hipSetDevice_and_first_stream": round(tm[1], 4), "two_more_streams_and_event": round(tm[2], 4)}
n_dip, n_pops, L_sites = 100, 4, 200_000
names = ["s%d" % d for d in range(n_dip)]
per = n_dip // n_pops
sd = SampleData(popNames=["pop%d" % k for k in range(n_pops)], popInds=[names[k * per:(k + 1) * per] for k in range(n_pops)])
lay = HapLayout(sd, names, "phased")
lap("set_layout_s", lambda: e.set_layout(lay))
lap("first_allocation_s", lambda: e.reserve(L_sites))
slot_gen = np.array([2 * names.index(nm) + k for nm in lay.ind_order for k in range(2)], dtype=np.int32)
lap("first_launch_pg_kernels_s", lambda: e.synth_fill(0, L_sites, 0, synth.SEED_DEFAULT, L_sites, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR))
lap("second_launch_pg_kernels_s", lambda: e.synth_fill(0, L_sites, 0, synth.SEED_DEFAULT, L_sites, n_dip, n_pops, slot_gen, synth.VAR_THR, synth.MISS_THR))
lo = np.arange(0, L_sites, 50000, dtype=np.int64)
hi = lo + 50000
lap("first_pass_pack_pair_finish_s", lambda: e.batch(lo, hi).groupDistTable(True, 100, 0.01))
lap("second_pass_s", lambda: e.batch(lo, hi).groupDistTable(True, 100, 0.01))
text = b"".join(b"chr1\t%d\t" % (i + 1) + b"\t".join([b"A/A", b"A/T", b"T/T", b"N/N"][(i * 7 + c * c) % 4] for c in range(n_dip)) + b"\n" for i in range(20000))
lap("first_tokenize_s", lambda: e.tokenize_text(text))
lap("second_tokenize_s", lambda: e.tokenize_text(text))
bz = genoio.bgzf_compress(text)
tab, used, n_text = genoio.bgzf_walk(bz)
out = np.empty(n_text, dtype=np.uint8)
vp = lambda a: C.c_void_p(a.ctypes.data)                                     # noqa: E731
lap("first_inflate_s", lambda: _lib.check(L.pg_inflate_device(e._h, vp(bz), used, vp(tab[0]), vp(tab[1]), vp(tab[2]), vp(tab[3]), len(tab[0]), vp(out), None)))
lap("second_inflate_s", lambda: _lib.check(L.pg_inflate_device(e._h, vp(bz), used, vp(tab[0]), vp(tab[1]), vp(tab[2]), vp(tab[3]), len(tab[0]), vp(out), None)))
rec["env"] = {k: v for k, v in os.environ.items() if k.startswith(("HIP_", "HSA_", "GPU_", "AMD_", "ROCR_")) and k != "HSA_ENABLE_IPC_MODE_LEGACY"}
print(json.dumps(rec))
