"""Host side of the MI355X engine: a ctypes-driven context plus `WindowBatch`, which plays the role the
list of per-window `genomics.Alignment` objects plays in the reference's worker loop
(popgenWindows.py:44-52, ABBABABAwindows.py:37-38, distMat.py:39-42).

`WindowBatch` keeps the reference's method names, argument meaning and result keys:
    groupDistStats(doPairs, minSites, minData)  -> genomics.py:956-995
    groupFreqStats()                            -> genomics.py:1002-1028
    indPairDists(includeSameWithSame, minSites) -> genomics.py:934-954
    ABBABABA(P1, P2, P3, P4, minData)           -> genomics.py:1647-1695
    pairCounts()                                -> the integers behind distMatrix()/pairNonNan()
All per-site / per-pair work AND the float64 finalisation of pi / dxy / Fst happen in HIP kernels; what is left here is
naming the columns (and, for popFreq / indPairDist / ABBA-BABA, a few O(1)-per-window NumPy expressions).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def encode_text(buf, layout, n_threads=0, head_rows=0, pitch=None, alloc=None, full_out=None, narrow_ok=False):
    """K0 host tokenizer.  buf: bytes of complete `.geno` data lines (no header).
    Returns (gt int8 [L][pitch or n_hap] in slot order, pos int64 [L], scaf_off int64 [L], scaf_len int32 [L]).
    head_rows > 0: gt and pos are views into arrays with that many spare rows in front (gt.base / pos.base), so that a
    caller can put carried-over rows before the new ones without copying the new ones.
    pitch: bytes per output row (the engine's row pitch: columns past n_hap stay zero, and the rows can be uploaded with one
    linear copy); alloc(shape, dtype): allocator of the two big arrays (page-locked memory for asynchronous uploads);
    full_out: a list that receives (gt_full, pos_full), the arrays including the spare head rows."""
    L = _lib.lib()
    n = C.c_int64(0)
    ptr, nbytes, _keep = _lib.text_ptr(buf)
    check(L.pg_count_lines(ptr, nbytes, C.byref(n)))
    cap = max(int(n.value), 1)
    width = int(pitch) if pitch else layout.n_hap
    assert width >= layout.n_hap
    alloc = alloc or np.zeros                   # (the tokenizer clears every row it writes, pad columns included)
    gt_full, pos_full = alloc((head_rows + cap, width), np.int8), alloc((head_rows + cap,), np.int64)
    if full_out is not None:
        full_out.append((gt_full, pos_full))
    gt, pos = gt_full[head_rows:], pos_full[head_rows:]
    soff = np.zeros(cap, dtype=np.int64)
    slen = np.zeros(cap, dtype=np.int32)
    got = C.c_int64(0)
    # narrow_ok: a cell may hold fewer alleles than its column's ploidy (rows tokenised under the widest layout of a file whose
    # ploidy changes along it: --inferPloidy)
    check(L.pg_encode_text(ptr, nbytes, _lib.FMT[layout.genoFormat] | (_lib.FMT_NARROW_OK if narrow_ok else 0), len(layout.col_ploidy), layout.max_ploidy,
                           np.ascontiguousarray(layout.col_slot), layout.col_ploidy, width, gt, pos, soff, slen,
                           cap, C.byref(got), n_threads))
    k = int(got.value)
    return gt[:k], pos[:k], soff[:k], slen[:k]


class PinnedPool:
    """NumPy arrays in page-locked host memory (pg_host_alloc) for large results: device-to-host copies into them run at
    PCIe speed.  A buffer returns to the pool when its array is garbage collected; the pool keeps at most `keep` bytes."""

    def __init__(self, keep=8 << 30):
        import threading
        self._free = {}                  # nbytes -> [address]
        self._held = 0
        self._keep = keep
        self._L = _lib.lib()
        self._lock = threading.Lock()    # arrays are allocated by the tokenizer thread and released wherever they die

    def _release(self, addr, nbytes):
        with self._lock:
            if self._held + nbytes <= self._keep:
                self._free.setdefault(nbytes, []).append(addr)
                self._held += nbytes
                return
        self._L.pg_host_free(C.c_void_p(addr))

    def zeros(self, shape, dtype):
        arr = self.empty(shape, dtype)
        arr[...] = 0
        return arr

    def empty(self, shape, dtype):
        import weakref
        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = max(count * dtype.itemsize, 1)
        if nbytes >= (1 << 20):
            # size classes (1/8 steps between powers of two) so that the slightly different blocks of a streamed input reuse
            # each other's buffers: page-locking a fresh gigabyte costs more than tokenising into it
            step = 1 << max(nbytes.bit_length() - 4, 0)
            nbytes = (nbytes + step - 1) // step * step
        addr = None
        with self._lock:
            lst = self._free.get(nbytes)
            if lst:
                addr = lst.pop()
                self._held -= nbytes
        if addr is None:
            p = C.c_void_p()
            check(self._L.pg_host_alloc(nbytes, C.byref(p)))
            addr = p.value
        buf = (C.c_char * nbytes).from_address(addr)
        arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
        weakref.finalize(buf, self._release, addr, nbytes)        # buf lives as long as any view of it
        return arr


class _TimedLib:
    """PG_TIMING=1: the library behind a proxy that adds up calls and wall seconds per entry point and thread (cli.Run.report_timing
    prints them as "lib_calls": where a driver's time goes between Python and the C-ABI)"""

    def __init__(self, lib):
        self._lib = lib
        self.calls = {}                  # (thread name, function) -> [calls, seconds]

    def __getattr__(self, name):
        import threading
        import time
        fn = getattr(self._lib, name)

        def timed(*args):
            t0 = time.perf_counter()
            try:
                return fn(*args)
            finally:
                slot = self.calls.setdefault((threading.current_thread().name, name), [0, 0.0])
                slot[0] += 1
                slot[1] += time.perf_counter() - t0
        setattr(self, name, timed)
        return timed


class Engine:
    """One device context (pg_ctx).  Not thread-safe; one per GPU."""

    def __init__(self, device=0):
        import os
        self._L = _TimedLib(_lib.lib()) if os.environ.get("PG_TIMING") else _lib.lib()
        h = C.c_void_p()
        check(self._L.pg_ctx_create(C.byref(h), device))
        self._h = h
        self.device = device
        self.layout = None
        self.n_sites = 0
        self.placement = None
        self.plane_placement = None
        self.pinned = PinnedPool()

    def close(self):
        if getattr(self, "_h", None):
            self._L.pg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration / data ---------------------------------------------------------------
    def set_layout(self, layout):
        check(self._L.pg_set_samples(self._h, layout.n_hap, layout.hap_pop, layout.hap_sample, layout.n_pops))
        if layout.n_pops > 0:
            # the order the reference's sums run in: Alignment rows are the haplotypes sorted by name (genomics.py:1122), the
            # populations np.unique's sorted labels (genomics.py:965)
            names = np.array(layout.sampleData.popNames)
            rank = np.empty(layout.n_pops, dtype=np.int32)
            rank[np.argsort(names)] = np.arange(layout.n_pops, dtype=np.int32)
            check(self._L.pg_set_reference_order(self._h, np.ascontiguousarray(pop_row_order(layout)), rank))
        self.layout = layout
        self.n_sites = 0
        self._pair_first = "slot"

    def set_sum_order(self, mode):
        """0: the float64 sums of pi / dxy / Fst and of the ABBA-BABA statistics in NumPy's order for windows of up to 256 sites (PG_NP_MAX_SITES);
        1: for every window; 2: for none (pg_set_sum_order)"""
        check(self._L.pg_set_sum_order(self._h, int(mode)))

    def set_pair_first(self, first):
        """which individual of a pair supplies the rows of the haplotype block pg_indpairdist_mean averages: "slot" (the earlier one
        in layout.ind_order) or "name" (the one whose name sorts first)"""
        if self.__dict__.get("_pair_first", "slot") != first:
            n = self.layout.n_samp
            rank = np.arange(n, dtype=np.int32)
            if first == "name":
                rank[np.argsort(np.array(self.layout.ind_order))] = np.arange(n, dtype=np.int32)
            check(self._L.pg_set_sample_rank(self._h, np.ascontiguousarray(rank)))
            self._pair_first = first

    def reserve(self, n_sites):
        """room for n_sites resident rows (growing drops the rows).  Reservations of 4 GiB and more try up to PG_PLACE_TRIALS
        (default 4, 1 = off) physical placements and keep the one the pack kernel streams fastest from (pg_reserve_sites_tuned);
        self.placement = (probe ms per candidate, index kept) of the last such reservation."""
        import os
        trials = int(os.environ.get("PG_PLACE_TRIALS", "4"))
        if int(n_sites) > self.n_sites:
            self.placement = self.plane_placement = None      # (a growing reservation: new rows, nothing chosen yet)
        if trials > 1:
            ms = (C.c_double * 8)()
            n, k = C.c_int(0), C.c_int(-1)
            check(self._L.pg_reserve_sites_tuned(self._h, int(n_sites), trials, ms, C.byref(n), C.byref(k)))
            if n.value:
                self.placement = ([round(ms[i], 4) for i in range(n.value)], int(k.value))
                # ... and, with the rows where they now are, the planes the pack kernel writes (PG_PLANE_TRIALS, default 4, 1 = off)
                self.tune_planes(n_sites)
        else:
            check(self._L.pg_reserve_sites(self._h, int(n_sites)))
        self.n_sites = max(self.n_sites, int(n_sites))

    def tune_planes(self, n_sites, trials=None, window_sites=50000):
        """pg_tune_planes: up to `trials` (PG_PLANE_TRIALS, default 4; at most 8) sets of the planes the pack kernel writes, tried
        on windows of window_sites of the resident rows [0, n_sites), the fastest kept; self.plane_placement = (probe ms per
        candidate, index kept).  reserve() calls it on the empty rows; a choice made on the loaded rows is worth more
        (tools/plane_placement.py).  Results do not depend on it."""
        import os
        trials = int(os.environ.get("PG_PLANE_TRIALS", "4")) if trials is None else int(trials)
        if trials < 2:
            return None
        ms = (C.c_double * 8)()
        n, k = C.c_int(0), C.c_int(-1)
        check(self._L.pg_tune_planes(self._h, int(n_sites), int(window_sites), trials, ms, C.byref(n), C.byref(k)))
        self.plane_placement = ([round(ms[i], 4) for i in range(n.value)], int(k.value))
        return self.plane_placement

    def upload(self, gt, offset=0):
        gt = np.ascontiguousarray(gt, dtype=np.int8)
        assert gt.ndim == 2 and gt.shape[1] == self.layout.n_hap
        check(self._L.pg_upload_sites(self._h, int(offset), gt, gt.shape[0]))

    def load_sites(self, gt):
        self.reserve(len(gt))
        self.upload(gt, 0)

    # ---- streamed ingestion (uploads on the context's copy stream) ---------------------------------------------
    @property
    def row_pitch(self):
        p = C.c_int(0)
        check(self._L.pg_row_pitch(self._h, C.byref(p)))
        return p.value

    def upload_async(self, gt, offset=0):
        """Queue the upload of int8 rows [n][pitch >= n_hap] (C-contiguous; page-locked for a real DMA) to resident rows
        offset.. and return; `gt` must stay alive and unchanged until upload_wait()."""
        assert gt.ndim == 2 and gt.dtype == np.int8 and gt.shape[1] >= self.layout.n_hap
        assert gt.strides[1] == 1 and gt.strides[0] >= gt.shape[1]
        check(self._L.pg_upload_sites_async(self._h, int(offset), C.c_void_p(gt.ctypes.data), gt.shape[0], gt.strides[0]))
        self._in_flight = gt

    def upload_packed_async(self, cells, offset, slot_src):
        """Queue the upload of packed cells uint8 [n][n_cols] (`.pgeno` payload); they are expanded into slot order on the
        device.  slot_src: layout.slot_src (2 * column + allele index per slot)."""
        assert cells.ndim == 2 and cells.dtype == np.uint8 and cells.flags.c_contiguous
        check(self._L.pg_upload_packed_async(self._h, int(offset), C.c_void_p(cells.ctypes.data), cells.shape[0], cells.shape[1],
                                             np.ascontiguousarray(slot_src, dtype=np.int32)))
        self._in_flight = cells

    def upload_wait(self):
        check(self._L.pg_upload_wait(self._h))
        self._in_flight = None

    def tokenize_text(self, buf, row_offset=0, n_rows=None, max_runs=1 << 16, at_most=False, file=None):
        """K0 on the device: complete `.geno` data lines (bytes-like: bytes, memoryview, mmap) -> resident rows row_offset ..;
        returns (n, pos int64 [n], run_starts int64 [r], run_names) or None when the block is not of the regular layout the device
        tokenizer handles (the caller then takes the host tokenizer).  The rows must be reserved: n_rows = number of data lines
        (pg_count_lines) if known; with at_most an upper bound of it (the device counts the lines itself: no pass of the host
        over the text).  file = (file descriptor, offset of buf in the file) of a plain-text block: the staging threads then read
        the text with pread() from the page cache instead of copying it out of the mapping (pg_tokenize_file)."""
        lay = self.layout
        ptr, nbytes, _keep = _lib.text_ptr(buf)
        if n_rows is None:
            n = C.c_int64(0)
            check(self._L.pg_count_lines(ptr, nbytes, C.byref(n)))
            n_rows = int(n.value)
        cap = max(int(n_rows), 1)
        pos = np.empty(cap, dtype=np.int64)
        while True:
            rrow, roff, rlen = np.empty(max_runs, dtype=np.int64), np.empty(max_runs, dtype=np.int64), np.empty(max_runs, dtype=np.int32)
            got, nr, ok = C.c_int64(0), C.c_int64(0), C.c_int(0)
            tail = (_lib.FMT[lay.genoFormat], len(lay.col_ploidy), lay.max_ploidy, np.ascontiguousarray(lay.col_slot), lay.col_ploidy,
                    int(row_offset), pos, cap, rrow, roff, rlen, max_runs, C.byref(got), C.byref(nr), C.byref(ok))
            if file is not None:
                check(self._L.pg_tokenize_file(self._h, int(file[0]), int(file[1]), nbytes, *tail))
            else:
                check(self._L.pg_tokenize_text(self._h, ptr, nbytes, *tail))
            if not ok.value and nr.value > max_runs and max_runs < cap:          # a block of very many short scaffolds: once more
                max_runs = cap
                continue
            break
        if not ok.value or (got.value != n_rows and not at_most):
            return None
        k, r = int(got.value), int(nr.value)
        names = [bytes(buf[int(roff[i]):int(roff[i]) + int(rlen[i])]).decode("utf-8", "replace") for i in range(r)]
        return k, pos[:k], rrow[:r].copy(), names

    # the same in three steps, two blocks in flight: parse(k) -> submit(k+1) -> collect(k) lets the kernels of block k run while the
    # text of block k+1 crosses PCIe (cli.Run._chunks_device)
    def tokenize_submit(self, buf, slot, file=None):
        """text of a block -> device buffer of `slot` (0 / 1), its lines counted behind the copies; False: not the regular layout"""
        lay = self.layout
        ptr, nbytes, _keep = _lib.text_ptr(buf)
        ok = C.c_int(0)
        check(self._L.pg_tokenize_submit(self._h, int(slot), None if file is not None else ptr, int(file[0]) if file is not None else -1,
                                         int(file[1]) if file is not None else 0, nbytes, _lib.FMT[lay.genoFormat], len(lay.col_ploidy),
                                         lay.max_ploidy, np.ascontiguousarray(lay.col_slot), lay.col_ploidy, C.byref(ok)))
        return bool(ok.value)

    def tokenize_submit_bgzf(self, span, slot):
        """a block of bgzip-compressed text (genoio.BgzfSpan) -> device buffer of `slot`: the members cross PCIe deflated and are
        inflated on the device (pg_tokenize_submit_bgzf); False: not the regular layout"""
        lay = self.layout
        in_off, in_len, out_len, crc = span.tab
        ok = C.c_int(0)
        vp = lambda a: C.c_void_p(a.ctypes.data)                            # noqa: E731
        src = (None, int(span.file[0]), int(span.file[1])) if span.file is not None else (vp(span.comp), -1, 0)
        check(self._L.pg_tokenize_submit_bgzf(self._h, int(slot), src[0], src[1], src[2], len(span.comp), vp(in_off), vp(in_len), vp(out_len), vp(crc),
                                              len(in_off), span.head, len(span.head), len(span), span.first_line, len(span.first_line),
                                              _lib.FMT[lay.genoFormat], len(lay.col_ploidy), lay.max_ploidy,
                                              np.ascontiguousarray(lay.col_slot), lay.col_ploidy, C.byref(ok)))
        return bool(ok.value)

    def set_deferred_results(self, on=True):
        """large result tables (WindowBatch.indPairTable into page-locked memory) are copied back on a stream of their own while the
        next call's kernels run; a table is complete after results_wait() / sync() (pg_set_deferred_results)"""
        check(self._L.pg_set_deferred_results(self._h, 1 if on else 0))

    def results_wait(self):
        check(self._L.pg_results_wait(self._h))

    def inflate_members(self, comp, tab, dst):
        """BGZF members (genoio.bgzf_walk's table over the bytes comp) inflated on the device into the host array dst
        (pg_inflate_device: k_inflate -- the CRC-32 taken on the way out --, then one copy back -- page-locked dst: at the link's rate); returns the kernels' ms.
        For text whose parser lives on the host (the VCF drop-in)."""
        in_off, in_len, out_len, crc = tab
        arr = np.frombuffer(comp, dtype=np.uint8)
        total = int(out_len.sum(dtype=np.int64))
        if dst.dtype != np.uint8 or not dst.flags.c_contiguous or dst.size < total:
            raise ValueError("inflate_members: dst must be a contiguous uint8 array of at least %d bytes" % total)
        vp = lambda a: C.c_void_p(a.ctypes.data if a.size else 0)           # noqa: E731
        ms = C.c_double(0)
        check(self._L.pg_inflate_device(self._h, vp(arr), arr.size, vp(in_off), vp(in_len), vp(out_len), vp(crc), len(in_off), vp(dst), C.byref(ms)))
        return ms.value

    # ---- VCF lines parsed on the device (pg_vcf_dev_*; genomics_general_amd/vcf.py) ----
    def vcf_config(self, plan):
        """the option set of a parseVCF run (vcf.Plan) -> the device; None when the device path takes it, else the reason it does not"""
        taken, why = C.c_int(0), C.c_char_p()
        fn = self._L.pg_vcf_dev_config
        check(fn(self._h, *plan.site_args(), C.c_char(plan.sep.encode()), 1 if plan.args.addRefTrack else 0, C.byref(taken), C.byref(why)))
        return None if taken.value else (why.value or b"").decode()

    def vcf_submit(self, slot, block, file=None):
        """a block of whole lines (bytes-like; file = (descriptor, offset) when it is a view of a memory-mapped file: the staging
        threads then read it themselves) or a genoio.BgzfSpan (members still deflated) -> text slot `slot`"""
        if hasattr(block, "tab"):
            in_off, in_len, out_len, crc = block.tab
            vp = lambda a: C.c_void_p(a.ctypes.data if a.size else 0)           # noqa: E731
            comp = np.frombuffer(block.comp, dtype=np.uint8)
            check(self._L.pg_vcf_dev_submit_bgzf(self._h, int(slot), vp(comp), comp.size, vp(in_off), vp(in_len), vp(out_len), vp(crc), len(in_off),
                                                 block.head, len(block.head), len(block), len(block.first_line) + 1, 1))
            return comp
        if file is not None:
            check(self._L.pg_vcf_dev_submit(self._h, int(slot), None, int(file[0]), int(file[1]), len(block)))
            return None
        ptr, n, keep = _lib.text_ptr(block)
        check(self._L.pg_vcf_dev_submit(self._h, int(slot), ptr, -1, 0, n))
        return keep

    def vcf_set_prev(self, chrom, pos):
        """--excludeDuplicates: the CHROM / POS tokens (bytes; None: none) of the last data line of blocks the host parsed"""
        check(self._L.pg_vcf_dev_set_prev(self._h, chrom, len(chrom or b""), pos, len(pos or b"")))

    def vcf_prev(self, slot):
        """(CHROM, POS) tokens of the data line before the block collected from `slot`, (None, None) when there is none, or None: the
        block before ended in a line the device's key could not hold and went to the host parser for it -- the caller has its key"""
        a, b = C.create_string_buffer(128), C.create_string_buffer(128)
        na, nb = C.c_int(-1), C.c_int(-1)
        check(self._L.pg_vcf_dev_prev(self._h, int(slot), a, C.byref(na), b, C.byref(nb)))
        if na.value == -2:
            return None
        return (None, None) if na.value < 0 else (a.raw[:na.value], b.raw[:nb.value])

    def vcf_parse(self, slot):
        check(self._L.pg_vcf_dev_parse(self._h, int(slot)))

    def vcf_collect(self, slot):
        """(bytes of the block's rows, their number, -1) or (0, 0, the first line the device does not take: the block is the host parser's)"""
        n, rows, line, gz = C.c_int64(0), C.c_int64(0), C.c_int64(-1), C.c_int64(0)
        check(self._L.pg_vcf_dev_collect(self._h, int(slot), C.byref(n), C.byref(rows), C.byref(line), C.byref(gz)))
        self.vcf_bgzf_bytes = gz.value              # (after vcf_set_output(True): the rows as BGZF members, for vcf_rows_bgzf)
        return n.value, rows.value, line.value

    def vcf_set_output(self, bgzf_members):
        """the rows of the blocks parsed from now on are also deflated on the device (k_deflate) into BGZF members"""
        check(self._L.pg_vcf_dev_set_output(self._h, 1 if bgzf_members else 0))

    def vcf_rows_bgzf(self, slot, nbytes):
        out = self.pinned.empty((max(int(nbytes), 1),), np.uint8)[:int(nbytes)]
        check(self._L.pg_vcf_dev_rows_bgzf(self._h, int(slot), C.c_void_p(out.ctypes.data), int(nbytes)))
        return out

    def bgzf_compress(self, text):
        """text (bytes-like) -> (BGZF members without the end-of-file member as a uint8 array, the kernels' ms): k_deflate
        (pg_bgzf_compress_device; tests, tools/bgzip.py --device)"""
        arr = np.frombuffer(text, dtype=np.uint8)
        cap = len(arr) + (len(arr) // 65280 + 2) * 64 + 65536
        out = self.pinned.empty((cap,), np.uint8)
        n, ms = C.c_int64(0), C.c_double(0)
        check(self._L.pg_bgzf_compress_device(self._h, C.c_void_p(arr.ctypes.data if len(arr) else 0), len(arr), C.c_void_p(out.ctypes.data), cap,
                                              C.byref(n), C.byref(ms)))
        return out[:n.value], ms.value

    def vcf_rows(self, slot, nbytes):
        out = self.pinned.empty((max(int(nbytes), 1),), np.uint8)[:int(nbytes)]
        check(self._L.pg_vcf_dev_rows(self._h, int(slot), C.c_void_p(out.ctypes.data), int(nbytes)))
        return out

    def vcf_text(self, slot, nbytes):
        out = self.pinned.empty((max(int(nbytes), 1),), np.uint8)[:int(nbytes)]
        check(self._L.pg_vcf_dev_text(self._h, int(slot), C.c_void_p(out.ctypes.data), int(nbytes)))
        return out

    def vcf_stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        check(self._L.pg_vcf_dev_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def tokenize_parse(self, slot, row_offset, row_capacity, max_runs=1 << 16):
        """queue the parse of the block in `slot` into resident rows row_offset .. (at most row_capacity of them); returns the number of
        its lines, or None when they do not fit"""
        n, ok = C.c_int64(0), C.c_int(0)
        check(self._L.pg_tokenize_parse(self._h, int(slot), int(row_offset), int(row_capacity), int(max_runs), C.byref(n), C.byref(ok)))
        return int(n.value) if ok.value else None

    def tokenize_collect(self, slot, buf, n_rows, max_runs=1 << 16):
        """wait for the parse of `slot`; (n, pos, run_starts, run_names) or None (irregular text met on the device, too many runs)"""
        pos = np.empty(max(int(n_rows), 1), dtype=np.int64)
        rrow, roff, rlen = np.empty(max_runs, dtype=np.int64), np.empty(max_runs, dtype=np.int64), np.empty(max_runs, dtype=np.int32)
        got, nr, ok = C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(self._L.pg_tokenize_collect(self._h, int(slot), pos, len(pos), rrow, roff, rlen, max_runs, C.byref(got), C.byref(nr), C.byref(ok)))
        if not ok.value:
            return None
        k, r = int(got.value), int(nr.value)
        if hasattr(buf, "tab"):                      # a BgzfSpan: the host never had this text, the names come back from the device
            out = np.empty(max(int(rlen[:r].sum()), 1), dtype=np.uint8)
            check(self._L.pg_tokenize_run_names(self._h, int(slot), roff, rlen, r, C.c_void_p(out.ctypes.data), len(out)))
            ends = np.cumsum(rlen[:r])
            names = [bytes(out[int(e) - int(n):int(e)]).decode("utf-8", "replace") for e, n in zip(ends, rlen[:r])]
        else:
            names = [bytes(buf[int(roff[i]):int(roff[i]) + int(rlen[i])]).decode("utf-8", "replace") for i in range(r)]
        return k, pos[:k], rrow[:r].copy(), names

    # packed cells (`.pgeno`, codec none) straight from the file: the staging threads of the tokenizer read them, k_unpack expands them
    def stage_file(self, slot, fd, file_offset, nbytes, dst_offset, capacity):
        check(self._L.pg_stage_file(self._h, int(slot), int(fd), int(file_offset), int(nbytes), int(dst_offset), int(capacity)))

    def unpack_staged(self, slot, src_offset, n_rows, n_cols, slot_src, row_offset):
        check(self._L.pg_unpack_staged(self._h, int(slot), int(src_offset), int(n_rows), int(n_cols),
                                       np.ascontiguousarray(slot_src, dtype=np.int32), int(row_offset)))

    def stage_sync(self):
        check(self._L.pg_stage_sync(self._h))

    def tokenize_stats(self):
        """{"h2d_s", "kernels_s", "bytes"} of the device tokenizer so far (pg_tokenize_stats)"""
        a, b, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        check(self._L.pg_tokenize_stats(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return {"h2d_s": a.value, "kernels_s": b.value, "bytes": int(n.value)}

    def move_rows(self, src_row, dst_row, n):
        check(self._L.pg_move_rows(self._h, int(src_row), int(dst_row), int(n)))

    def download(self, offset, n):
        out = np.zeros((n, self.layout.n_hap), dtype=np.int8)
        check(self._L.pg_download_sites(self._h, int(offset), out, n))
        return out

    def synth_fill(self, offset, n_sites, first_site_index, seed, scaf_len, n_dip, n_pops_gen, slot_gen_hap,
                   var_thr, miss_thr):
        check(self._L.pg_synth_fill(self._h, int(offset), int(n_sites), int(first_site_index), int(seed), int(scaf_len),
                                    int(n_dip), int(n_pops_gen), np.ascontiguousarray(slot_gen_hap, dtype=np.int32),
                                    int(var_thr), int(miss_thr)))

    def sync(self):
        check(self._L.pg_sync(self._h))

    def set_scratch_limit(self, nbytes):
        check(self._L.pg_set_scratch_limit(self._h, int(nbytes)))

    def kernel_time(self, kernel_id):
        ms, n = C.c_double(0), C.c_int64(0)
        check(self._L.pg_kernel_time(self._h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_select(self, kernel_ids=None):
        """Bracket only these kernel families with HIP events (None = all)."""
        mask = 0xFFFFFFFF if kernel_ids is None else sum(1 << int(k) for k in kernel_ids)
        check(self._L.pg_kernel_time_select(self._h, mask))

    def kernel_time_reset(self):
        check(self._L.pg_kernel_time_reset(self._h))

    def batch(self, win_lo, win_hi):
        return WindowBatch(self, win_lo, win_hi)

    # ---- multi-GPU ---------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(_lib.lib().pg_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, n_ranks, rank, uid):
        check(self._L.pg_comm_init(self._h, n_ranks, rank, uid))

    def comm_allgather(self, send):
        send = np.ascontiguousarray(send, dtype=np.float64).ravel()
        recv = np.zeros(send.size * self._comm_ranks(), dtype=np.float64)
        check(self._L.pg_comm_allgather_f64(self._h, send, recv, send.size))
        return recv.reshape(self._comm_ranks(), -1)

    def _comm_ranks(self):
        return self._n_ranks

    def comm_setup(self, n_ranks, rank, uid):
        self.comm_init(n_ranks, rank, uid)
        self._n_ranks = n_ranks

    def comm_barrier(self):
        check(self._L.pg_comm_barrier(self._h))

    def indPairTableFromCounts(self, D, C, includeSameWithSame=False, minSites=None):
        """WindowBatch.indPairTable() from pair counts supplied by the caller (D, C as pairCounts(reference_order=False) returns
        them): the counts of disjoint parts of a window add, so the ranks of a multi-GPU `distMat.py --windType cat` run sum
        theirs and finish the means from the sums (pg_indpairdist_mean_from_counts)."""
        lay = self.layout
        n = lay.n_samp
        D = np.ascontiguousarray(D, dtype=np.int32)
        C = np.ascontiguousarray(C, dtype=np.int32)
        assert D.shape == C.shape == (D.shape[0], lay.n_hap, lay.n_hap)
        tab = np.zeros((D.shape[0], n * (n + 1) // 2), np.float64)
        self.set_pair_first("slot")
        check(self._L.pg_indpairdist_mean_from_counts(self._h, D, C, D.shape[0], int(minSites) if minSites else 0,
                                                      1 if includeSameWithSame else 0, tab))
        return tab


class WindowBatch:
    """Statistics of a set of windows [lo,hi) of the engine's resident sites."""

    def __init__(self, engine, win_lo, win_hi):
        self.e = engine
        self.lay = engine.layout
        self.lo = np.ascontiguousarray(win_lo, dtype=np.int64)
        self.hi = np.ascontiguousarray(win_hi, dtype=np.int64)
        self.n = len(self.lo)
        self._popdist_min_sites = None     # set once groupDistStats has "masked the cached matrix"
        self._diag_nan = False             # indPairDists() without includeSameWithSame leaves a nan diagonal in the cache

    # -- integers ---------------------------------------------------------------------------------
    def pairCounts(self, reference_order=True):
        """D[w][i][j], C[w][i][j] (int32).  reference_order: rows/cols permuted to the reference's
        sorted-haplotype-name order (genomics.py:1122); otherwise device slot order."""
        N = self.lay.n_hap
        D = np.zeros((self.n, N, N), dtype=np.int32)
        Cc = np.zeros((self.n, N, N), dtype=np.int32)
        check(self.e._L.pg_pairwise(self.e._h, self.lo, self.hi, self.n, D, Cc))
        if reference_order:
            o = self.lay.ref_order
            D = D[:, o][:, :, o]
            Cc = Cc[:, o][:, :, o]
        return D, Cc

    def hapCalled(self):
        """seqNonNan per window: int64 [n_win][n_hap] in device slot order (genomics.py:1038-1040)."""
        out = np.zeros((self.n, self.lay.n_hap), dtype=np.int64)
        check(self.e._L.pg_hap_called(self.e._h, self.lo, self.hi, self.n, out))
        return out

    def siteCounts(self, site_lo, site_hi):
        out = np.zeros((site_hi - site_lo, self.lay.n_pops, 4), dtype=np.int32)
        check(self.e._L.pg_site_counts(self.e._h, int(site_lo), int(site_hi), out))
        return out

    def siteTarget(self, site_lo, site_hi, target, minData=0.0, asCounts=False, threshold=None):
        """freq.py's columns for the sites [site_lo, site_hi) (pg_site_target: k_site_counts + k_site_target): (values [n][n_pops] --
        int64 counts of the target allele when asCounts, else float64 frequencies rounded to 4 places --, keep [n] uint8: 0 for a row
        that is all NaN / all zero).  target: "derived" (the last population is the outgroup) or "minor"."""
        n, P = int(site_hi - site_lo), self.lay.n_pops
        vals = np.empty((n, P), dtype=np.int64 if asCounts else np.float64)
        keep = np.empty(n, dtype=np.uint8)
        check(self.e._L.pg_site_target(self.e._h, int(site_lo), int(site_hi), {"derived": 1, "minor": 2}[target], float(minData),
                                       1 if asCounts else 0, 1 if (threshold and not asCounts) else 0, float(threshold or 0.0),
                                       C.c_void_p(vals.ctypes.data), C.c_void_p(keep.ctypes.data)))
        return vals, keep

    # -- popDist / popPairDist ----------------------------------------------------------------------
    def groupDistTable(self, doPairs=True, minSites=None, minData=0.01):
        """pi / dxy / Fst of every window as one float64 table [window][column], finished on the device (k_popstats: the float64
        operations of genomics.py:976-993 in the reference's order).  Returns (table, column names); columns: pi per population,
        then dxy and Fst per unordered population pair."""
        lay = self.lay
        P = lay.n_pops
        ms = int(minSites) if minSites else 0
        pairs = bool(P > 1 and doPairs)
        ncols = P + (P * (P - 1) if pairs else 0)
        tab = np.zeros((self.n, ncols), dtype=np.float64)
        check(self.e._L.pg_popdist_stats(self.e._h, self.lo, self.hi, self.n, ms, float(minData), 1 if pairs else 0, tab))
        self._popdist_min_sites = ms
        cols = lay.__dict__.setdefault("_dist_cols", {}).get(pairs)
        if cols is None:
            names = lay.sampleData.popNames
            cols = ["pi_" + names[x] for x in range(P)]
            if pairs:
                pr = [(names[x], names[y]) for x in range(P - 1) for y in range(x + 1, P)]
                cols += ["dxy_%s_%s" % ab for ab in pr] + ["Fst_%s_%s" % ab for ab in pr]
            lay._dist_cols[pairs] = cols
        return tab, cols

    def groupDistStats(self, doPairs=True, minSites=None, minData=0.01):
        """groupDistTable as {stat name: array over windows}, both key orders for the pair statistics."""
        tab, cols = self.groupDistTable(doPairs, minSites, minData)
        out = {}
        for k, name in enumerate(cols):
            out[name] = tab[:, k]
        P = self.lay.n_pops
        if len(cols) > P:
            names = self.lay.sampleData.popNames
            npo = P * (P - 1) // 2
            k = 0
            for x in range(P - 1):
                for y in range(x + 1, P):
                    out["dxy_%s_%s" % (names[y], names[x])] = tab[:, P + k]
                    out["Fst_%s_%s" % (names[y], names[x])] = tab[:, P + npo + k]
                    k += 1
        return out

    def groupDistSums(self, minSites=None):
        """The raw K3 output: float64 sums of D/C and valid-pair counts per unordered population pair (pg_popdist)."""
        P = self.lay.n_pops
        npairs = P * (P + 1) // 2
        sums = np.zeros((self.n, npairs), dtype=np.float64)
        cnts = np.zeros((self.n, npairs), dtype=np.int64)
        check(self.e._L.pg_popdist(self.e._h, self.lo, self.hi, self.n, int(minSites) if minSites else 0, sums, cnts))
        return sums, cnts

    # -- popFreq --------------------------------------------------------------------------------------
    def groupFreqStats(self):
        lay = self.lay
        P = lay.n_pops
        l = np.zeros(self.n, dtype=np.int64)
        S = np.zeros((self.n, P), dtype=np.int64)
        prs = np.zeros((self.n, P), dtype=np.int64)
        seq = np.zeros((self.n, P), dtype=np.float64)
        check(self.e._L.pg_popfreq(self.e._h, self.lo, self.hi, self.n, l, S, prs, seq))
        out = {}
        has = l >= 1
        for x, name in enumerate(lay.sampleData.popNames):
            N = lay.pop_sizes[x]
            with np.errstate(divide="ignore", invalid="ignore"):
                # thetaPi: the reference's site-by-site sum, formed in that order on the device (k_popfreq_ordered); a population
                # of one haplotype has 0/0 per site there (and a dead worker after it): nan
                theta_pi = np.where(has, seq[:, x] if N > 1 else np.nan, np.nan)
                a = np.sum(1. / np.arange(1, N))
                theta_w = np.where(has, S[:, x] / a, np.nan)
                taj = np.where(has, _tajima_d(N, S[:, x].astype(np.float64), theta_pi), np.nan)
            out["l_" + name] = l
            out["S_" + name] = np.where(has, S[:, x].astype(np.float64), np.nan)
            out["S_int_" + name] = S[:, x]
            out["thetaPi_" + name] = theta_pi
            out["thetaW_" + name] = theta_w
            out["TajD_" + name] = taj
        return out

    # -- indHet / hapStats: finished on the device, nothing N x N leaves the GPU ---------------------
    def _cache_state(self):
        """What the reference worker's cached `_distMat_` carries at this point: the minSites mask of a preceding
        groupDistStats (genomics.py:959-961) and a nan diagonal (genomics.py:963, or indPairDists without
        includeSameWithSame, genomics.py:940)."""
        ms = self._popdist_min_sites or 0
        return int(ms), bool(self._popdist_min_sites is not None or self._diag_nan)

    def sampleHet(self):
        """{individual: array over windows} like Alignment.sampleHet() (genomics.py:918-929); pg_sample_het keeps the
        reference's operator precedence (a value only where bit 1 of the jointly called site count is set)."""
        lay = self.lay
        ms, _ = self._cache_state()
        tab = np.zeros((self.n, lay.n_samp), dtype=np.float64)
        check(self.e._L.pg_sample_het(self.e._h, self.lo, self.hi, self.n, ms, tab))
        return {name: tab[:, k] for k, name in enumerate(lay.ind_order)}

    def H12stats(self, maxDist=0):
        """H1 / H12 / H2 per population like Alignment.H12stats (genomics.py:1079-1098): pg_hapstats clusters every window's
        match matrix on the device, ties broken in the reference's row order (haplotype names sorted)."""
        lay = self.lay
        ms, diag_nan = self._cache_state()
        order = pop_row_order(lay)
        tab = np.zeros((self.n, lay.n_pops, 3), dtype=np.float64)
        check(self.e._L.pg_hapstats(self.e._h, self.lo, self.hi, self.n, ms, 1 if diag_nan else 0, float(maxDist),
                                    np.ascontiguousarray(order), tab))
        out = {}
        for x, name in enumerate(lay.sampleData.popNames):
            out["H1_" + name], out["H12_" + name], out["H2_" + name] = tab[:, x, 0], tab[:, x, 1], tab[:, x, 2]
        return out

    def indPairSums(self, minSites=None):
        """Raw K6 output (pg_indpairdist): sums of D/C and valid-pair counts per unordered individual pair."""
        n = self.lay.n_samp
        npairs = n * (n + 1) // 2
        big = self.n * npairs * 8 >= (1 << 20)            # large tables land in page-locked memory (direct DMA)
        alloc = self.e.pinned.empty if big else np.zeros   # fully overwritten by the copy back
        sums = alloc((self.n, npairs), np.float64)
        cnts = alloc((self.n, npairs), np.int64)
        check(self.e._L.pg_indpairdist(self.e._h, self.lo, self.hi, self.n, int(minSites) if minSites else 0, sums, cnts))
        return sums, cnts

    # -- indPairDist -------------------------------------------------------------------------------------
    def indPairTable(self, includeSameWithSame=False, minSites=None, first="slot"):
        """Alignment.indPairDists as one array [window][pair]: the nanmean of every individual pair's haplotype block, finished
        on the device (pg_indpairdist_mean).  Pair (s<=t) in slot order of the individuals sits at lay.sample_pair_index(s,t).
        As in the reference (which mutates its cached distance matrix), a preceding groupDistStats leaves its minSites mask
        and nan diagonal in force."""
        lay = self.lay
        n = lay.n_samp
        npairs = n * (n + 1) // 2
        # which individual of a pair supplies the rows of the haplotype block (the order np.nanmean adds a 2 x 2 block up in):
        # first="slot": the earlier one in lay.ind_order (distMat.py:44-45: the order of its samples); "name": the one whose name
        # sorts first (popgenWindows.py:55-57)
        self.e.set_pair_first(first)
        # fully overwritten by the copy back; large tables land in page-locked memory (direct DMA)
        tab = (self.e.pinned.empty if self.n * npairs * 8 >= (1 << 20) else np.zeros)((self.n, npairs), np.float64)
        ms = int(minSites) if minSites else 0
        diag_nan = not includeSameWithSame
        if self._popdist_min_sites is not None:
            ms = max(ms, self._popdist_min_sites)
            diag_nan = True
        if diag_nan:
            self._diag_nan = True
        if minSites:
            self._popdist_min_sites = max(int(minSites), self._popdist_min_sites or 0)   # the mask stays in the cache
        check(self.e._L.pg_indpairdist_mean(self.e._h, self.lo, self.hi, self.n, ms, 0 if diag_nan else 1, tab))
        return tab

    def indPairDists(self, includeSameWithSame=False, minSites=None):
        """{name: {name: array over windows}} like Alignment.indPairDists(asDict=True); views into indPairTable()."""
        lay = self.lay
        n = lay.n_samp
        tab = self.indPairTable(includeSameWithSame, minSites, first="name")       # out[a][b] for a before b in sorted names
        out = {a: {} for a in lay.ind_order}
        for s in range(n):
            for t in range(s, n):
                v = tab[:, lay.sample_pair_index(s, t)]
                out[lay.ind_order[s]][lay.ind_order[t]] = v
                out[lay.ind_order[t]][lay.ind_order[s]] = v
        return out

    # -- ABBA-BABA ----------------------------------------------------------------------------------------
    def ABBABABA(self, P1, P2, P3, P4, minData):
        names = self.lay.sampleData.popNames
        ids = [names.index(p) for p in (P1, P2, P3, P4)]
        sums = np.zeros((self.n, 6), dtype=np.float64)
        used = np.zeros(self.n, dtype=np.int64)
        check(self.e._L.pg_abbababa(self.e._h, self.lo, self.hi, self.n, ids[0], ids[1], ids[2], ids[3], float(minData),
                                    sums, used))
        # a window without one good site (biallelic, enough data): genomics.py:1693-1695 zips six names with seven values, so
        # the reference's sitesUsed there is nan, not 0 (pg_abbababa tells these windows as -1); fourPop's zip is even: 0
        none_good = used < 0
        used_f = np.where(none_good, np.nan, used.astype(np.float64))
        with np.errstate(divide="ignore", invalid="ignore"):
            out = {"D": sums[:, 0] * 1. / sums[:, 1], "fd": sums[:, 0] * 1. / sums[:, 2], "fdM": sums[:, 0] * 1. / sums[:, 3],
                   "ABBA": np.where(none_good, np.nan, sums[:, 4]), "BABA": np.where(none_good, np.nan, sums[:, 5]),
                   "sitesUsed": used_f}
        return out

    # -- four-population statistics ----------------------------------------------------------------------------
    def fourPop(self, P1, P2, P3, P4, minData, polarize=False, fixed=False):
        """genomics.fourPop (genomics.py:1585-1643): the window sums come from pg_fourpop, the ratios are formed here exactly
        as `x.sum()*1./y.sum()` in genomics.py:1420-1554."""
        names = self.lay.sampleData.popNames
        ids = [names.index(p) for p in (P1, P2, P3, P4)]
        sel = 1 if polarize else 2 if fixed else 0             # genomics.py:1610-1615: polarize wins over fixed
        sums = np.zeros((self.n, 14), dtype=np.float64)
        used = np.zeros(self.n, dtype=np.int64)
        check(self.e._L.pg_fourpop(self.e._h, self.lo, self.hi, self.n, ids[0], ids[1], ids[2], ids[3], float(minData), sel,
                                   sums, used))
        f4, f4c = sums[:, 0], sums[:, 6]
        with np.errstate(divide="ignore", invalid="ignore"):
            out = {"D": f4 * 1. / sums[:, 1], "fd": f4 * 1. / sums[:, 2], "fdm": f4 * 1. / sums[:, 3],
                   "fd'": f4c * 1. / sums[:, 7], "fdm'": f4c * 1. / sums[:, 8], "fdh": f4c * 1. / sums[:, 9],
                   "fdh2": f4c * 1. / sums[:, 10], "fh": f4c * 1. / sums[:, 11],
                   "ABBA": sums[:, 4], "BABA": sums[:, 5], "ABAA": sums[:, 12], "BAAA": sums[:, 13], "sitesUsed": used}
        return out                       # windows with sitesUsed == 0 carry 0.0 / nan; the drivers never print them


def pop_row_order(lay):
    """for each population the slots of its haplotypes in the reference's row order (haplotype names sorted, genomics.py:1122),
    concatenated in population order: int32 [slots in populations]"""
    order = lay.__dict__.get("_pop_row_order")
    if order is None:
        rank = np.empty(lay.n_hap, dtype=np.int64)
        rank[lay.ref_order] = np.arange(lay.n_hap)
        in_pop = np.where(lay.hap_pop >= 0)[0]
        order = in_pop[np.lexsort((rank[in_pop], lay.hap_pop[in_pop]))].astype(np.int32)
        lay._pop_row_order = order
    return order


def _tajima_d(n, S, theta_pi):
    """Tajima's D of genomics.TajimaD (genomics.py:619-632) for arrays over windows (n = haplotypes of the population):
    (thetaPi - S/h1) / sqrt(v1*S + v2*S*(S-1)) with the usual constants from the harmonic sums h1 = sum 1/i, h2 = sum 1/i^2."""
    if n < 2:
        return np.full_like(theta_pi, np.nan)
    h1 = h2 = 0.0
    for i in range(1, n):                 # left to right, as the reference's Python sums (a pairwise np.sum differs in the last ulp)
        h1 += 1.0 / i
        h2 += 1.0 / (i ** 2)
    v1 = ((n + 1.0) / (3 * (n - 1)) - 1.0 / h1) / h1
    v2 = (2.0 * (n * n + n + 3) / (9 * n * (n - 1)) - (n + 2) / (h1 * n) + h2 / h1 ** 2) / (h1 ** 2 + h2)
    return (theta_pi - 1.0 * S / h1) / np.sqrt(v1 * S + v2 * S * (S - 1))
