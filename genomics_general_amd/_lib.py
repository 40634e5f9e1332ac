"""ctypes binding of libpopgen_hip.so (include/popgen_hip.h).

There is no CPU fallback: if the shared library is missing, or (for anything but the host tokenizer)
no MI355X is visible, the call fails loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PG_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpopgen_hip.so")   # PG_LIBRARY: e.g. the sanitizer build (make asan)


class PopgenError(RuntimeError):
    """A libpopgen_hip.so call returned a negative pg_status."""

    def __init__(self, code, msg):
        super().__init__("libpopgen_hip: [%d] %s" % (code, msg))
        self.code = code


PG_ERR_ARG, PG_ERR_HIP, PG_ERR_NODEV, PG_ERR_PARSE, PG_ERR_RCCL, PG_ERR_STATE = -1, -2, -3, -4, -5, -6
FMT = {"phased": 0, "pairs": 1, "haplo": 2, "diplo": 3}
FMT_NARROW_OK = 0x100        # OR'ed into pg_encode_text's fmt: cells may hold fewer alleles than their column's slots
K_PACK, K_PAIRWISE, K_POPDIST_FIN, K_SITESTATS, K_SYNTH, K_PAIRD, K_INDPAIR_FIN, K_RESULT_D2H, K_ORDERED = 0, 1, 2, 3, 4, 5, 6, 7, 8
KERNEL_NAMES = {K_PACK: "k_pack", K_PAIRWISE: "k_pairC", K_POPDIST_FIN: "k_popdist_fin",
                K_SITESTATS: "k_sitestats", K_SYNTH: "k_synth", K_PAIRD: "k_pairD", K_INDPAIR_FIN: "k_indpair_fin",
                K_RESULT_D2H: "result_d2h", K_ORDERED: "k_popfreq_ordered"}

_lib = None

_P = C.c_void_p
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

# name -> (restype, argtypes); every name here must be declared in include/popgen_hip.h
SIGNATURES = {
    "pg_abi_version": (C.c_int, []),
    "pg_last_error": (C.c_char_p, []),
    "pg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pg_ctx_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "pg_ctx_destroy": (C.c_int, [_P]),
    "pg_sync": (C.c_int, [_P]),
    "pg_set_samples": (C.c_int, [_P, C.c_int, _i32p, _i32p, C.c_int]),
    "pg_set_reference_order": (C.c_int, [_P, _i32p, _i32p]),
    "pg_set_sample_rank": (C.c_int, [_P, _i32p]),
    "pg_set_sum_order": (C.c_int, [_P, C.c_int]),
    "pg_np_tree": (C.c_int, [C.c_int, _i32p, C.c_int64, C.POINTER(C.c_int64)]),
    "pg_reserve_sites": (C.c_int, [_P, C.c_int64]),
    "pg_reserve_sites_tuned": (C.c_int, [_P, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pg_tune_planes": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pg_upload_sites": (C.c_int, [_P, C.c_int64, _i8p, C.c_int64]),
    "pg_download_sites": (C.c_int, [_P, C.c_int64, _i8p, C.c_int64]),
    "pg_row_pitch": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "pg_upload_sites_async": (C.c_int, [_P, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]),
    "pg_upload_packed_async": (C.c_int, [_P, C.c_int64, C.c_void_p, C.c_int64, C.c_int, _i32p]),
    "pg_upload_wait": (C.c_int, [_P]),
    "pg_tokenize_text": (C.c_int, [_P, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int64, _i64p, C.c_int64,
                                   _i64p, _i64p, _i32p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "pg_tokenize_file": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int64, _i64p, C.c_int64,
                                   _i64p, _i64p, _i32p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "pg_tokenize_submit": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p,
                                     C.POINTER(C.c_int)]),
    "pg_tokenize_parse": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "pg_tokenize_collect": (C.c_int, [_P, C.c_int, _i64p, C.c_int64, _i64p, _i64p, _i32p, C.c_int64, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "pg_stage_file": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "pg_unpack_staged": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.c_int, _i32p, C.c_int64]),
    "pg_stage_sync": (C.c_int, [_P]),
    "pg_tokenize_stats": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pg_move_rows": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64]),
    "pg_synth_fill": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_int32, C.c_int32,
                                _i32p, C.c_int32, C.c_int32]),
    "pg_encode_text": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int, _i8p, _i64p,
                                 _i64p, _i32p, C.c_int64, C.POINTER(C.c_int64), C.c_int]),
    "pg_gzip_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "pg_gzip_read_lines": (C.c_int, [_P, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pg_gzip_close": (C.c_int, [_P]),
    "pg_gzip_stats": (C.c_int, [_P, _i64p]),
    "pg_text_cell_widths": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, _i32p, _i32p, _i64p, _i32p, C.c_int64, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64)]),
    "pg_scaffold_runs": (C.c_int, [C.c_void_p, _i64p, _i32p, C.c_int64, _i64p, C.c_int64, C.POINTER(C.c_int64)]),
    "pg_count_lines": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)]),
    "pg_usable_cpus": (C.c_int, []),
    "pg_text_runs": (C.c_int, [C.c_void_p, C.c_size_t, _i64p, C.c_int64, C.POINTER(C.c_int64)]),
    "pg_text_seek_pos": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_int64, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pg_text_skip_rows": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    # (struct array + many pointers: genomics_general_amd/vcf.py passes explicit ctypes objects)
    "pg_encode_vcf": (C.c_int, None),
    "pg_vcf_render_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char, C.c_char, C.c_int,
                                     C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int]),
    "pg_vcf_dev_config": (C.c_int, None),
    "pg_vcf_dev_set_prev": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_char_p, C.c_int]),
    "pg_vcf_dev_prev": (C.c_int, [_P, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_int)]),
    "pg_vcf_dev_submit": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64]),
    "pg_vcf_dev_submit_bgzf": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "pg_vcf_dev_parse": (C.c_int, [_P, C.c_int]),
    "pg_vcf_dev_collect": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pg_vcf_dev_set_output": (C.c_int, [_P, C.c_int]),
    "pg_vcf_dev_rows_bgzf": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int64]),
    "pg_bgzf_compress_device": (C.c_int, [_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pg_vcf_dev_rows": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int64]),
    "pg_vcf_dev_text": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int64]),
    "pg_vcf_dev_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pg_format_freq_rows": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_void_p, _i64p, _i32p, C.c_char_p, _i64p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.POINTER(C.c_int64), C.c_int]),
    "pg_inflate_chunks": (C.c_int, [C.c_void_p, _i64p, _i64p, _i64p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    "pg_bgzf_walk": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pg_inflate_members": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _i64p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "pg_bgzf_compress": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int]),
    "pg_inflate_device": (C.c_int, [_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.POINTER(C.c_double)]),
    "pg_tokenize_submit_bgzf": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _i32p,
                                          C.POINTER(C.c_int)]),
    "pg_tokenize_run_names": (C.c_int, [_P, C.c_int, _i64p, _i32p, C.c_int64, C.c_void_p, C.c_int64]),
    "pg_decode_packed": (C.c_int, [_u8p, C.c_int64, C.c_int, C.c_int, _i32p, _i32p, C.c_int, _i8p, C.c_int]),
    "pg_pairwise": (C.c_int, [_P, _i64p, _i64p, C.c_int, _i32p, _i32p]),
    "pg_popdist": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, _f64p, _i64p]),
    "pg_popdist_stats": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, C.c_double, C.c_int, _f64p]),
    "pg_indpairdist": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, _f64p, _i64p]),
    "pg_indpairdist_mean": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, C.c_int, _f64p]),
    "pg_indpairdist_mean_from_counts": (C.c_int, [_P, _i32p, _i32p, C.c_int, C.c_int, C.c_int, _f64p]),
    "pg_sample_het": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, _f64p]),
    "pg_hapstats": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, C.c_int, C.c_double, _i32p, _f64p]),
    "pg_abbababa": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _f64p, _i64p]),
    "pg_fourpop": (C.c_int, [_P, _i64p, _i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _f64p,
                             _i64p]),
    "pg_popfreq": (C.c_int, [_P, _i64p, _i64p, C.c_int, _i64p, _i64p, _i64p, _f64p]),
    "pg_hap_called": (C.c_int, [_P, _i64p, _i64p, C.c_int, _i64p]),
    "pg_site_counts": (C.c_int, [_P, C.c_int64, C.c_int64, _i32p]),
    "pg_format_float_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_char, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.POINTER(C.c_int64), C.c_int]),
    "pg_set_deferred_results": (C.c_int, [_P, C.c_int]),
    "pg_results_wait": (C.c_int, [_P]),
    "pg_site_target": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    "pg_kernel_time": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pg_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "pg_host_free": (C.c_int, [C.c_void_p]),
    "pg_kernel_time_select": (C.c_int, [_P, C.c_uint32]),
    "pg_kernel_time_reset": (C.c_int, [_P]),
    "pg_debug_address": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pg_debug_place": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "pg_debug_cu_split": (C.c_int, [_P, C.c_int]),
    "pg_ctx_create_times": (C.c_int, [C.POINTER(C.c_double)]),
    "pg_set_scratch_limit": (C.c_int, [_P, C.c_int64]),
    "pg_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pg_comm_init": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p]),
    "pg_comm_allgather_f64": (C.c_int, [_P, _f64p, _f64p, C.c_int64]),
    "pg_comm_barrier": (C.c_int, [_P]),
    "pg_comm_destroy": (C.c_int, [_P]),
}


def lib():
    """Load (once) and return the shared library; raise ImportError with a build hint if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C genomics_general_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
                              "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            if args is not None:
                fn.argtypes = args
        if L.pg_abi_version() != 1:
            raise ImportError("libpopgen_hip.so ABI version %d != 1" % L.pg_abi_version())
        _lib = L
    return _lib


def text_ptr(buf):
    """(address, length, keep-alive) of a bytes-like object -- bytes, memoryview, mmap -- for the `const char *` arguments of the
    tokenizer (a memory-mapped input file goes in without a copy)"""
    arr = np.frombuffer(buf, dtype=np.uint8)
    return C.c_void_p(arr.ctypes.data if arr.size else 0), int(arr.size), arr


def check(rc):
    if rc < 0:
        raise PopgenError(rc, lib().pg_last_error().decode("utf-8", "replace"))
    return rc


def device_count():
    """Number of visible GPUs (0 when none; never raises for 'no device')."""
    n = C.c_int(0)
    rc = lib().pg_device_count(C.byref(n))
    if rc == PG_ERR_NODEV:
        return 0
    check(rc)
    return n.value


def usable_cpus():
    """CPUs this process may really use (affinity mask and cgroup quota applied): what thread pools are sized from"""
    return int(lib().pg_usable_cpus())
