"""Drop-in command lines: popgenWindows.py, ABBABABAwindows.py and distMat.py of the reference, same flags, same
`.geno` input, same CSV / matrix output, with the per-window numeric work done by
libpopgen_hip.so on an MI355X.  There is no CPU engine: without the library or a GPU these exit with an error.

Reference flag tables: popgenWindows.py:172-210, ABBABABAwindows.py:111-145, distMat.py:118-156.
Additive flags: --device N (GPU index; default LOCAL_RANK or 0).  Under a one-process-per-GPU launcher
(RANK/WORLD_SIZE set) windows are sharded across ranks and rank 0 writes the output.
"""
import argparse
import ctypes as C
import gzip
import itertools
import os
import sys

import numpy as np

from . import _lib, dist, genoio, windows
from .engine import Engine
from .samples import HapLayout, SampleData

WINDOW_FLAGS = [
    (("--windType",), dict(choices=None, default="coordinate", help="Type of windows to make")),
    (("-w", "--windSize"), dict(type=int, metavar="sites", help="Window size in bases (or sites)")),
    (("-s", "--stepSize"), dict(type=int, metavar="sites", help="Step size for sliding window")),
    (("-m", "--minSites"), dict(type=int, metavar="sites", default=1, help="Minimum good sites per window")),
    (("-D", "--maxDist"), dict(type=int, help="Maximum span distance for sites window")),
    (("--windCoords",), dict(help="Window coordinates file (scaffold start end)")),
]
IO_FLAGS = [
    (("-g", "--genoFile"), dict(help="Input genotypes file (.gz by suffix; stdin if absent)")),
    (("-o", "--outFile"), dict(help="Results file (.gz by suffix; stdout if absent)")),
    (("--exclude",), dict(help="File of scaffolds to exclude")),
    (("--include",), dict(help="File of scaffolds to analyse")),
    (("-f", "--genoFormat"), dict(choices=("phased", "pairs", "haplo", "diplo"), required=True,
                                  help="Format of genotypes in genotypes file")),
    (("--verbose",), dict(action="store_true", help="Verbose output")),
    (("--addWindowID",), dict(action="store_true", help="Add window name or number as first column")),
    (("--writeFailedWindows",), dict(action="store_true", help="Write output even for windows with too few sites.")),
    (("--device",), dict(type=int, default=None, help="GPU index (MI355X engine)")),
]
PLOIDY_FLAGS = [
    (("--ploidy",), dict(type=int, nargs="+", help="Ploidy for each sample")),
    (("--ploidyFile",), dict(help="File with samples names and ploidy as columns")),
    (("--inferPloidy",), dict(action="store_true", help="Ploidy will be inferred in each window (NOT RECOMMENDED): one extra pass over the "
                                                         "input looks at the cell widths; where they never change, they decide once")),
]


ENGINE_EPILOG = ("MI355X engine: one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as a launcher sets them), "
                 "the input is split over the ranks at scaffold runs and the finished rows are gathered once.  Environment: "
                 "PG_STREAM_BYTES text bytes per input block (default 1 GiB); PG_SCRATCH_GIB scratch budget of the pairwise pipeline "
                 "(default 48); PG_PLACE_TRIALS=1 no placement trials when 4 GiB or more of rows are reserved (default: up to 4 "
                 "allocations of the rows are held together and probed, the fastest kept -- on a device shared with other jobs set it "
                 "to 1); PG_GPU_TOKENIZER=0 tokenise on host threads; PG_HOST_THREADS host threads per process; PG_TIMING=1 per-phase "
                 "times on stderr.  See README.md for the full list.")


def _add(parser, table, **override):
    for names, kw in table:
        kw = dict(kw)
        if names[0] in override:
            kw.update(override[names[0]])
        parser.add_argument(*names, **kw)


def _lines(path):
    with open(path, "rt") as f:
        return [ln.rstrip() for ln in f.readlines()]


def _no_line_is_read(args, header_given):
    """True when no data line of the input lies on a scaffold the contig lists let through (genomics.py:2016) -- then the reference
    never looks a sample name up.  Text files only (a pipe cannot be read twice); stops at the first line that passes."""
    import gzip
    path = getattr(args, "genoFile", None)
    if not path or str(path).endswith(".pgeno"):
        return False
    inc = set(_lines(args.include)) if getattr(args, "include", None) else None
    exc = set(_lines(args.exclude)) if getattr(args, "exclude", None) else None
    with (gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "rt")) as f:
        if not header_given:
            f.readline()
        for line in f:
            tok = line.split(None, 1)
            if tok and ((not inc and not exc) or (inc and tok[0] in inc) or (exc and tok[0] not in exc)):
                return False
    return True


class _BgzfTextOut:
    """`-o out.csv.gz` (popgenWindows.py:316, freq.py: "If you add `.gz` it will be gzipped"): the reference's gzip.open(path, "wt") is
    Python's gzip module at level 9 on the calling thread -- about 10 MB/s, seconds for the per-site table of freq.py behind a run of
    a third of a second.  Here: BGZF (a valid gzip file for every reader) deflated 16 MiB at a time by the library's host threads
    (genoio.BgzfWriter: pg_bgzf_compress).  A run that fails leaves the file without its end-of-file member."""

    def __init__(self, path):
        self.buffer = genoio.BgzfWriter(path)

    def write(self, text):
        self.buffer.write(text.encode() if isinstance(text, str) else text)
        return len(text)

    def flush(self):
        pass

    def close(self):
        self.buffer.close()


def _open_out(path):
    if not path:
        return sys.stdout
    if path.endswith(".gz"):
        return gzip.open(path, "wt") if os.environ.get("PG_OUT_GZIP_MODULE") else _BgzfTextOut(path)
    return open(path, "wt")


def _window_setup(args, overlap):
    """The window-parameter asserts shared by the three drivers (popgenWindows.py:216-244)."""
    wt = args.windType
    p = dict(windType=wt, windSize=args.windSize, stepSize=None, overlap=0, maxDist=np.inf, windCoords=None)
    if wt == "coordinate":
        assert args.windSize, "Window size must be provided."
        p["stepSize"] = args.stepSize if args.stepSize else args.windSize
        assert not overlap, "Overlap does not apply to coordinate windows. Use --stepSize instead."
        assert not args.maxDist, "Maximum distance only applies to sites windows."
    elif wt == "sites":
        assert args.windSize, "Window size (number of sites) must be provided."
        p["overlap"] = overlap if overlap else 0
        p["maxDist"] = args.maxDist if args.maxDist else np.inf
        assert not args.stepSize, "Step size only applies to coordinate windows. Use --overlap instead."
    elif wt == "predefined":
        assert args.windCoords, "Please provide a file of window coordinates."
        p["windCoords"] = args.windCoords
        assert not overlap, "Overlap does not apply for predefined windows."
        assert not args.maxDist, "Maximum does not apply for predefined windows."
        assert not args.stepSize, "Step size does not apply for predefined windows."
        assert not args.include, "You cannot only include specific scaffolds if using predefined windows."
        assert not args.exclude, "You cannot exclude specific scaffolds if using predefined windows."
    return p


def _ploidy_dict(args, inds, haploid_list):
    """popgenWindows.py:293-305."""
    if args.ploidy is not None:
        pl = args.ploidy if len(args.ploidy) != 1 else args.ploidy * len(inds)
        assert len(pl) == len(inds), "Incorrect number of ploidy values supplied."
        return dict(zip(inds, pl))
    if args.ploidyFile is not None:
        with open(args.ploidyFile, "rt") as pf:
            return dict([[s[0], int(s[1])] for s in [ln.split() for ln in pf]])
    if args.inferPloidy:
        # The reference leaves the ploidy open and genoToAlignment takes the number of sequences splitSeq returns for the window
        # (genomics.py:1108-1111, 390-396): per window and sample, what the window's shortest cell holds.  In the phased / pairs
        # formats that is a matter of the cell widths of the whole input: one pass over the text (pg_text_cell_widths) finds the
        # rows at which they change.  No change (the usual case): the widths decide once, every fast path stays.  Otherwise the
        # run tokenises under the widest ploidies and computes every window under its own (Run, MultiLayoutBatch).
        header = getattr(args, "header", None) or (" ".join(args.headers) if getattr(args, "headers", None) else None)
        if args.genoFormat in ("phased", "pairs") and not (args.genoFile and str(args.genoFile).endswith(".pgeno")):
            if args.genoFile is None:
                args.genoFile = _spool_stdin()                   # (the pass over the widths, then the run itself, read it)
            seg = genoio.scan_ploidy_segments(args.genoFile, args.genoFormat, list(inds), header)
            if len(seg.starts) > 1:
                args._ploidy_segments = seg
            return seg.max_ploidy()
        inferred = genoio.first_row_ploidy(args.genoFile, args.genoFormat, header)
        for s in inds:
            assert s in inferred, "sample %s is not in the genotype file header" % s
        return {s: inferred[s] for s in inds}
    d = dict(zip(inds, [1 if args.genoFormat == "haplo" else 2] * len(inds)))
    for s in haploid_list or []:
        d[s] = 1
    return d


def _spool_stdin():
    """--inferPloidy on a piped input: the text is needed twice (the cell widths of the whole input, then the run), so it goes into
    a temporary file first, removed when the process ends"""
    import atexit
    import os
    import tempfile
    fd, path = tempfile.mkstemp(prefix="pg_stdin_", suffix=".geno")
    atexit.register(lambda: os.path.exists(path) and os.remove(path))
    with os.fdopen(fd, "wb") as f:
        while True:
            piece = genoio.STDIN.read(64 << 20)
            if not piece:
                break
            f.write(piece)
    return path


def _make_windows(p, data, minSites, coords_keep=4):
    inc = _lines(p["include"]) if p.get("include") else None
    exc = _lines(p["exclude"]) if p.get("exclude") else None
    if p["windType"] == "coordinate":
        return windows.coord_windows(data.run_starts, data.run_names, data.pos, p["windSize"], p["stepSize"], inc, exc)
    if p["windType"] == "sites":
        return windows.sites_windows(data.run_starts, data.run_names, data.pos, p["windSize"], p["overlap"],
                                     p["maxDist"], minSites, inc, exc)
    return windows.predefined_windows(data.run_starts, data.run_names, data.pos, _read_coords(p["windCoords"], coords_keep))


def _read_coords(path, coords_keep=4):
    """Windows file of --windCoords: scaffold, start, end[, ID] per line."""
    coords = []
    with open(path, "rt") as wc:
        for line in wc:
            f = line.split()[:coords_keep]
            if len(f) >= 3:
                coords.append(tuple([f[0], int(f[1]), int(f[2])] + f[3:4]))
    return coords


class Run:
    """Shared plumbing: input -> layout -> windows -> engine (+ multi-GPU shard of the window list).

    The input is consumed in blocks of PG_STREAM_BYTES bytes of text (default 1 GiB) when the window type allows it
    (coordinate, sites and predefined windows: windows.CoordWindowStream / SitesWindowStream / PredefinedWindowStream), so host
    memory stays bounded for inputs of any size; cat windows read the whole input as one block.  Drivers iterate `for _ in run.chunks():`; inside the loop
    T, w0, w1, lo, hi, batch() and gather() refer to the windows that became certain with the current block.  Drivers that
    do not stream construct Run(..., stream=False): the single chunk is loaded by the constructor."""

    def __init__(self, args, sampleData, wparams, minSites, header_line=None, coords_keep=4, windows_fn=None, stream=False,
                 shardable=False):
        import os
        import time
        self.world = dist.world_from_env()
        if self.world.size > 1 and not os.environ.get("PG_HOST_THREADS"):
            # N ranks on one node: the native helpers (line count, tokenizers, staging copies of the device tokenizer) share the cores
            os.environ["PG_HOST_THREADS"] = str(max(1, _lib.usable_cpus() // self.world.size))
        self._t_start = time.perf_counter()
        self._timeline = [] if os.environ.get("PG_TIMELINE") else None
        self.timing = {"read_s": 0.0, "text_bytes": 0, "tokenize_s": 0.0, "windows_s": 0.0, "sites": 0, "windows": 0,
                       "engine_and_upload_s": 0.0, "upload_s": 0.0, "prep_wait_s": 0.0, "chunks": 0}   # printed as JSON on stderr when PG_TIMING=1
        # the device context (HIP runtime start-up, streams: 0.1 - 0.2 s) is created by a helper thread while this one opens the input,
        # reads the header and sets up samples and windows
        import threading
        made = {}

        def make_engine():
            try:
                t_e = time.perf_counter()
                made["engine"] = Engine(args.device if args.device is not None else dist.device_for(self.world))
                made["seconds"] = time.perf_counter() - t_e
            except BaseException as exc:
                made["error"] = exc
        engine_thread = threading.Thread(target=make_engine)
        engine_thread.start()
        t0 = time.perf_counter()
        self._reader = genoio.open_input(args.genoFile)
        if header_line:
            names = header_line.split()[2:]
        else:
            names = self._reader.read_header().decode("utf-8", "replace").split()[2:]
        self.timing["read_s"] += time.perf_counter() - t0
        f = getattr(self._reader, "f", None)
        if (args.genoFile and str(args.genoFile).endswith(".gz") and not isinstance(f, genoio.BgzfFile) and self.world.rank == 0
                and os.path.getsize(args.genoFile) > int(os.environ.get("PG_GZIP_HINT_BYTES", 64 << 20))):
            # one gzip stream has no independent pieces: it is inflated serially (~ 0.4 GB/s of text), whatever the GPU does
            sys.stderr.write("note: %s is a single gzip stream and is inflated serially; written by `bgzip` (or tools/bgzip.py) the same "
                             "text is inflated on the GPU, about a hundred times faster\n" % args.genoFile)
        try:
            self.layout = HapLayout(sampleData, names, args.genoFormat)
        except KeyError:
            # Names that are not in the header.  The reference looks a name up in the lines it adds to a window and nowhere else
            # (genomics.py:1993), so a run whose contig lists leave no line of the file ends there with the header row and nothing
            # else; any other run stops at its first line.  Same here: no line will be read, under columns made up for the purpose.
            if not _no_line_is_read(args, header_line is not None):
                raise
            wanted = list(dict.fromkeys(sampleData.indNames))
            names = wanted + ["\0column %d" % k for k in range(len(names) - len(wanted))]
            self.layout = HapLayout(sampleData, names, args.genoFormat)
        self._infer_ploidy = bool(getattr(args, "inferPloidy", False))
        # --inferPloidy on a file whose cell widths change (genoio.PloidySegments): self.layout holds the widest ploidies, the rows
        # are tokenised under it by the host tokenizer (narrower cells leave their other slots missing) and stay on the host;
        # batch() computes every window under the layout of its own ploidies (MultiLayoutBatch).  Every rank reads the whole
        # input, the windows of a block are split over the ranks.
        self._segments = getattr(args, "_ploidy_segments", None)
        self._file_names = names
        if self._segments is not None:
            shardable = False
        self._wparams = dict(wparams, include=args.include, exclude=args.exclude)
        self._minSites, self._coords_keep, self._windows_fn = minSites, coords_keep, windows_fn
        self._streamer = None
        self._block_bytes = None
        # every rank of a multi-GPU run tokenises the input itself: share the host cores instead of oversubscribing them
        self._tok_threads = max(1, _lib.usable_cpus() // self.world.size) if self.world.size > 1 else 0
        if stream and windows_fn is None and wparams["windType"] in ("coordinate", "sites", "predefined"):
            inc = _lines(args.include) if args.include else None
            exc = _lines(args.exclude) if args.exclude else None
            if wparams["windType"] == "coordinate":
                self._streamer = windows.CoordWindowStream(wparams["windSize"], wparams["stepSize"], inc, exc)
            elif wparams["windType"] == "predefined":
                self._streamer = windows.PredefinedWindowStream(_read_coords(wparams["windCoords"], coords_keep))
            else:
                self._streamer = windows.SitesWindowStream(wparams["windSize"], wparams["overlap"], wparams["maxDist"],
                                                           minSites, inc, exc)
            self._block_bytes = int(os.environ.get("PG_STREAM_BYTES", 1 << 30))
        # bgzip-compressed text on one rank: the first block (its members read, walked, its first and last line found) is fetched
        # while the device context is still being created
        self._first_block = None
        if (self.world.size == 1 and self._block_bytes is not None and isinstance(getattr(self._reader, "f", None), genoio.BgzfFile)
                and self._segments is None and hasattr(Engine, "tokenize_submit_bgzf") and device_tokenizer_takes(self.layout)
                and os.environ.get("PG_GPU_TOKENIZER", "1") != "0" and os.environ.get("PG_BGZF_DEVICE", "1") != "0"):
            self._reader.spans = True
            box = {}

            def first_block():
                try:
                    t_f = time.perf_counter()
                    box["block"] = self._reader.read_block(self._block_bytes)
                    box["seconds"] = time.perf_counter() - t_f
                except BaseException as exc:
                    box["error"] = exc
            th = threading.Thread(target=first_block, name="first-block")
            th.start()
            self._first_block = (th, box)
        t0 = time.perf_counter()
        engine_thread.join()
        if "error" in made:
            raise made["error"]
        self.engine = made["engine"]
        self.engine.set_layout(self.layout)
        self.comm = dist.make_comm(self.engine, self.world)
        self.timing["context_s"] = time.perf_counter() - t0          # what this thread still waited for the device context + the communicator (part of engine_and_upload_s)
        self.timing["context_create_s"] = made["seconds"]            # what the context took on its thread
        self.timing["engine_and_upload_s"] += time.perf_counter() - t0
        self.n_tested = 0
        # Multi-GPU ingestion.  Sharded (a driver that writes its rows through open_sink(), coordinate or sites windows, plain
        # text or `.pgeno` on disk, enough scaffold runs): every rank reads, tokenises and computes only its own run-aligned
        # slice of the input and formats its own rows; ONE gather of the finished rows at the end, rank 0 writes.  Otherwise
        # replicated: every rank tokenises the whole input, the windows of every block are split over the ranks, one all-gather
        # of the statistics per block.
        self.sharded = False
        self._wshare = (self.world.size, self.world.rank)
        self.shard_plan = None
        if (shardable and self.world.size > 1 and self._streamer is not None and wparams["windType"] in ("coordinate", "sites")
                and os.environ.get("PG_SHARD_INPUT", "1") != "0"):
            from . import shardplan
            inc = set(_lines(args.include)) if args.include else None
            exc = set(_lines(args.exclude)) if args.exclude else None
            wanted = lambda nm: windows._wanted(nm, inc, exc)            # noqa: E731
            # window ranges (cuts inside scaffold runs: every rank about 1/N of the bytes whatever the number of scaffolds);
            # PG_SHARD_INPUT=runs keeps to cuts between scaffold runs, which is also what inputs the plan cannot take fall back to
            if os.environ.get("PG_SHARD_INPUT", "1") != "runs":
                self.shard_plan = shardplan.shard_reader(self._reader, self.world, self.comm, wparams, wanted)
            if self.shard_plan is not None:
                self.sharded = True
                if wparams["windType"] == "coordinate":
                    self._streamer = windows.CoordWindowStream(wparams["windSize"], wparams["stepSize"], inc, exc,
                                                               start=self.shard_plan.start, stop=self.shard_plan.stop)
            else:
                self.sharded = self._reader.shard(self.world, self.comm, wanted)
            if self.sharded:
                self._wshare = (1, 0)
                self._tok_threads = max(1, _lib.usable_cpus() // self.world.size)
        # predefined windows: shardable when the file's scaffold runs agree with the window list (windows.plan_predefined_shards);
        # every rank scans its equal share of the bytes for run starts (pg_text_runs), one gather makes the run list of the file
        self.shift_ids = True
        if (shardable and self.world.size > 1 and self._streamer is not None and wparams["windType"] == "predefined"
                and os.environ.get("PG_SHARD_INPUT", "1") != "0" and hasattr(self._reader, "text_runs")):
            import json
            mine = self._reader.text_runs(self.world)
            parts = dist.gather_bytes(self.comm, json.dumps(mine).encode())
            lists = [json.loads(p.decode()) for p in parts]
            if all(x is not None for x in lists):
                runs = []
                for lst in lists:
                    for off, name in lst:
                        if not runs or runs[-1][1] != name:
                            runs.append((int(off), name))
                coords = self._streamer.coords
                plan = windows.plan_predefined_shards(runs, self._reader.tell(), self._reader.input_size(), coords, self.world.size)
                if plan is not None:
                    a, b, idx, tail = plan[self.world.rank]
                    self._reader.restrict(a, b)
                    self._streamer = windows.PredefinedWindowStream([coords[k] for k in idx], scaf_order=[w[0] for w in coords], tail=tail)
                    self.sharded, self._wshare, self.shift_ids = True, (1, 0), False
                    self._tok_threads = max(1, _lib.usable_cpus() // self.world.size)
        # `cat` (one window = every site of the input): sites are independent and pair counts add, so every rank takes a share of
        # the LINES, counts its share, and the counts are summed across the ranks before the means are formed (distmat_main)
        self.cat_sharded = False
        if (shardable and self.world.size > 1 and wparams["windType"] == "cat" and hasattr(self._reader, "shard_lines")
                and os.environ.get("PG_SHARD_INPUT", "1") != "0"):
            self.cat_sharded = bool(self._reader.shard_lines(self.world))
            if self.cat_sharded:
                self._wshare = (1, 0)
                self._tok_threads = max(1, _lib.usable_cpus() // self.world.size)
        if not stream:
            for _ in self.chunks():
                break

    def chunks(self):
        """Generator over the pieces of the input; see the class docstring.

        Three stages run beside each other: a reader thread fetches (and gunzips) block k+2, a tokenizer thread turns block k+1
        into rows and finds its windows (K0 + windows.*Stream), and this thread uploads and computes block k.  With an engine
        that offers them (Engine.upload_async / upload_wait; tests/cpu_engine.py mimics them), the rows are tokenised straight
        into page-locked memory at the engine's row pitch and go down asynchronously into one half of the resident rows while
        the windows of the previous block are still being computed in the other half; packed `.pgeno` cells are uploaded as
        they are and expanded on the device (PG_HOST_UNPACK=1 keeps the host decoder)."""
        import os
        import queue
        import threading
        import time
        eng = self.engine
        if self._device_tokenizer():
            try:
                yield from self._chunks_device()
            except _lib.PopgenError as exc:
                self._explain_ploidy_error(exc)
                raise
            return
        seg = self._segments
        piped = hasattr(eng, "upload_async") and seg is None
        pitch = eng.row_pitch if piped else None
        alloc = eng.pinned.empty if piped else None
        keep_packed = bool(piped and getattr(self._reader, "packed", False) and not os.environ.get("PG_HOST_UNPACK"))
        # the next block is read (and gunzipped) by a helper thread while this one is tokenised and computed
        blocks = queue.Queue(maxsize=1)
        prepared = queue.Queue(maxsize=1)
        stop = threading.Event()

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                first = self._take_first_block()
                if first is not None and (not put(blocks, first) or len(first) == 0):
                    return
                while True:
                    b = self._reader.read_block(self._block_bytes)
                    if not put(blocks, b) or self._block_bytes is None or len(b) == 0:
                        return
            except BaseException as exc:                  # surfaced in the consumer
                put(blocks, exc)

        def prepare():
            """tokenise block after block behind the rows carried over from the previous one; find the windows that are certain"""
            carry = None
            rows_done = 0                     # data rows of the input in front of the current block
            try:
                while True:
                    t0 = time.perf_counter()
                    body = blocks.get()
                    if isinstance(body, BaseException):
                        raise body
                    final = self._streamer is None or len(body) == 0
                    tm = {"read_s": time.perf_counter() - t0}             # time this stage waited for the reader
                    t0 = time.perf_counter()
                    kw = dict(narrow_ok=True) if seg is not None else {}
                    block = self._reader.to_geno(body, self.layout, n_threads=self._tok_threads,
                                                 head_rows=carry.n_sites if carry is not None else 0, pitch=pitch, alloc=alloc,
                                                 keep_packed=keep_packed, **kw)
                    del body
                    row0 = rows_done - (carry.n_sites if carry is not None else 0)    # index, among the input's data rows, of row 0 of `data`
                    rows_done += int(block.n_sites)
                    data = genoio.concat(carry, block)
                    data.row0 = row0
                    tm["tokenize_s"] = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    if self._streamer is not None:
                        T, keep_from = self._streamer.feed(data.run_starts, data.run_names, data.pos, final)
                    else:
                        T = (self._windows_fn(data) if self._windows_fn
                             else _make_windows(self._wparams, data, self._minSites, self._coords_keep))
                        T.dup = np.zeros(T.n, dtype=bool)
                        keep_from = data.n_sites
                    tm["windows_s"] = time.perf_counter() - t0
                    carry = None if final else genoio.tail(data, keep_from)
                    if not put(prepared, (data, T, final, int(block.n_sites), tm)) or final:
                        return
            except BaseException as exc:
                put(prepared, exc)

        threading.Thread(target=produce, daemon=True, name="reader").start()
        threading.Thread(target=prepare, daemon=True, name="prepare").start()

        def take(block=True):
            try:
                item = prepared.get(block=block)
            except queue.Empty:
                return None
            if isinstance(item, _lib.PopgenError):
                self._explain_ploidy_error(item)
            if isinstance(item, BaseException):
                raise item
            return item

        def stage(item, k, other_half_busy):
            """this rank's windows of the chunk, the rows they cover, and (piped) the start of their upload into half k % 2 of the
            resident rows; returns None when the rows do not fit and the other half cannot be given up yet"""
            data, T, final, n_new, tm = item
            w0, w1 = dist.shard_range(T.n, self._wshare[0], self._wshare[1])
            lo, hi = T.lo[w0:w1], T.hi[w0:w1]
            nz = hi > lo
            s0, s1 = (int(lo[nz].min()), int(hi[nz].max())) if np.any(nz) else (0, 0)
            base = 0
            t1 = time.perf_counter()
            if piped:
                rows = s1 - s0
                if rows > self._half:
                    if other_half_busy:
                        return None
                    self._half = max(rows + rows // 4, 1 << 16)
                    eng.reserve(2 * self._half)
                base = (k % 2) * self._half
                if rows > 0 and data.packed:
                    eng.upload_packed_async(data.gt[s0:s1], base, self.layout.slot_src)
                elif rows > 0:
                    eng.upload_async(data.gt[s0:s1], base)
            lo, hi = lo - s0 + base, hi - s0 + base
            lo[~nz] = 0
            hi[~nz] = 0
            return dict(data=data, T=T, final=final, n_new=n_new, tm=tm, w0=w0, w1=w1, lo=lo, hi=hi, s0=s0, s1=s1, base=base,
                        t_stage=time.perf_counter() - t1)

        self._half = 0
        k = 0
        staged, waiting = None, None          # chunk k+1: upload already queued / taken from the queue but not yet uploadable
        try:
            while True:
                t0 = time.perf_counter()
                if staged is not None:
                    cur, staged = staged, None
                else:
                    item = waiting if waiting is not None else take()
                    waiting = None
                    cur = stage(item, k, other_half_busy=False)
                self.timing["prep_wait_s"] += time.perf_counter() - t0      # time this thread waited for the tokenizer stage
                t0 = time.perf_counter()
                if piped:
                    eng.upload_wait()
                elif seg is None:
                    eng.load_sites(cur["data"].gt[cur["s0"]:cur["s1"]])
                self.timing["upload_s"] += time.perf_counter() - t0 + cur["t_stage"]
                self.timing["engine_and_upload_s"] += time.perf_counter() - t0 + cur["t_stage"]
                for key in ("read_s", "tokenize_s", "windows_s"):
                    self.timing[key] += cur["tm"][key]
                self.timing["text_bytes"] = self._reader.bytes_read
                self.data, self.T, self.w0, self.w1 = cur["data"], cur["T"], cur["w0"], cur["w1"]
                self.lo, self.hi, self.site0 = cur["lo"], cur["hi"], cur["s0"] - cur["base"]
                self.timing["sites"] += cur["n_new"]
                self.timing["windows"] += int(self.T.n)
                self.timing["chunks"] += 1
                self.n_tested += int(self.T.n)
                # when the tokenizer is ahead, the next chunk's rows go down while this chunk's windows are computed
                if piped and not cur["final"]:
                    waiting = take(block=False)
                    if waiting is not None:
                        staged = stage(waiting, k + 1, other_half_busy=True)
                        if staged is not None:
                            waiting = None
                if self.T.n:
                    yield self
                if cur["final"]:
                    break
                k += 1
        finally:
            stop.set()
            if piped:
                eng.upload_wait()
        self._reader.close()

    def _ev(self, label, t0):
        """PG_TIMELINE=1: (thread, label, start, end) of a step, seconds since the run began; report_timing prints the list"""
        if self._timeline is not None:
            import threading
            import time
            self._timeline.append((threading.current_thread().name, label, round(t0 - self._t_start, 5),
                                   round(time.perf_counter() - self._t_start, 5)))

    def _take_first_block(self):
        """the block the constructor's helper thread has fetched (bgzip-compressed input on one rank), or None"""
        if self._first_block is None:
            return None
        th, box = self._first_block
        self._first_block = None
        th.join()
        if "error" in box:
            raise box["error"]
        self.timing["first_block_prefetch_s"] = box["seconds"]
        return box["block"]

    def _explain_ploidy_error(self, exc):
        """--inferPloidy in the haplo / diplo formats (one character per cell whatever the ploidy, genomics.py:390-396) and on `.pgeno`
        input: nothing to infer from, the format decides; a cell of another width is an error here as it is a KeyError there.  (In
        the phased / pairs formats the cell widths of the whole input are read first -- genoio.scan_ploidy_segments -- and windows
        are computed under their own ploidies, so no such error arises.)"""
        if self._infer_ploidy and exc.code == _lib.PG_ERR_PARSE and "ploidy" in str(exc):
            raise SystemExit("--inferPloidy: %s.\nIn the %s format a cell is one character per genotype; a cell of another width "
                             "cannot be read.  Give explicit ploidies (--ploidy / --ploidyFile / --haploid) if the file mixes formats."
                             % (str(exc), self.layout.genoFormat))

    def _device_tokenizer(self):
        """K0 on the device (Engine.tokenize_text; PG_GPU_TOKENIZER=0 keeps the host tokenizer): plain or gzipped text in one of the
        regular layouts (cells of their columns' widths: mixed ploidy included); what only the text can tell (comment lines, runs
        of blanks, carriage returns, a cell of another width) is found by the kernels block by block, and such a block goes
        through the host tokenizer."""
        import os
        if self._segments is not None:
            return False
        if getattr(self._reader, "packed", False):
            # `.pgeno` with raw cells (codec none): the cells go from the file to the device through the same staging threads and
            # are expanded there (pg_stage_file / pg_unpack_staged); deflated cells are inflated by host threads (chunks())
            return (getattr(self._reader, "codec", None) == "none" and hasattr(self.engine, "stage_file")
                    and not os.environ.get("PG_HOST_UNPACK"))
        if os.environ.get("PG_GPU_TOKENIZER", "1") == "0" or not hasattr(self.engine, "tokenize_text"):
            return False
        return device_tokenizer_takes(self.layout)

    def _chunks_device(self):
        """chunks() with the tokenizer on the device, three stages beside each other:
          reader thread     hands over block k+2 (a view of the memory-mapped text, or gunzipped bytes);
          ingestion thread  has block k+1 copied down and tokenised (pg_tokenize_file / pg_tokenize_text on the copy streams) into ONE
                            HALF of the resident rows, behind the rows carried over from block k (pg_move_rows), and finds the
                            windows that are certain by then (windows.*Stream.feed);
          this thread       computes the windows of block k, whose rows sit in the OTHER half, and writes their rows.
        The host keeps positions and scaffold runs only; no row ever exists in host memory.  The number of rows a block needs is
        bounded from its first line (every line of the regular layout is at least that long), the device counts the lines itself.
        The halves grow at quiet points (when this thread has finished everything handed over so far)."""
        import os
        import queue
        import threading
        import time
        eng = self.engine
        L = _lib.lib()
        blocks = queue.Queue(maxsize=1)
        ready = queue.Queue(maxsize=2)
        halves = threading.Semaphore(2)           # a half is taken when a block is tokenised into it, given back when its windows are done
        stop = threading.Event()

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                first = self._take_first_block()              # (fetched beside the creation of the device context)
                if first is not None and (not put(blocks, first) or len(first) == 0):
                    return
                while True:
                    t_e = time.perf_counter()
                    b = self._reader.read_block(self._block_bytes)
                    self._ev("read_block", t_e)
                    if not put(blocks, b) or self._block_bytes is None or len(b) == 0:
                        return
            except BaseException as exc:
                put(blocks, exc)

        def row_bound(body):
            """at most this many data lines: a line of the regular layout is its cells + a scaffold, a position and two blanks (>= 4
            bytes); None when the first line does not look like one"""
            head = bytes(body[:1 << 16])
            nl = head.find(b"\n")
            f = head[:nl if nl >= 0 else len(head)].split(None, 2)
            if nl < 0 or len(f) < 3 or head.startswith(b"#"):
                return None
            return len(body) // (len(f[2]) + 1 + 4) + 1

        def count(body):
            ptr, nbytes, _keep = _lib.text_ptr(body)
            cnt = C.c_int64(0)
            if nbytes:
                _lib.check(L.pg_count_lines(ptr, nbytes, C.byref(cnt)))
            return int(cnt.value)

        st = {"half": 0}

        def grow(rows, carry_row0, c_n):
            """both halves to `rows` rows; the carried rows travel through the host once.  Only at a quiet point: this drops the rows
            the compute thread may be working on"""
            ready.join()
            saved = eng.download(carry_row0, c_n) if c_n else None
            st["half"] = rows + rows // 4 + 1024
            eng.reserve(2 * st["half"])
            return saved

        if hasattr(eng, "tokenize_submit_bgzf") and hasattr(self._reader, "spans") and os.environ.get("PG_BGZF_DEVICE", "1") != "0":
            self._reader.spans = True                 # bgzip-compressed text arrives as blocks of deflated members
            if isinstance(self._reader.f, genoio.BgzfFile) and hasattr(eng, "pinned"):
                self._reader.f.alloc = eng.pinned.empty   # ... read straight into page-locked buffers: one DMA each
        # a block of a memory-mapped file is a view, a block of raw `.pgeno` cells a list of file offsets: nothing to read ahead
        mapped = getattr(self._reader, "mm", None) is not None or bool(getattr(self._reader, "packed", False))
        done_reading = []

        def fetch():
            t0 = time.perf_counter()
            if mapped:
                body = b"" if done_reading else self._reader.read_block(self._block_bytes)
                if self._block_bytes is None or len(body) == 0:
                    done_reading.append(True)
            else:
                body = blocks.get()
                if isinstance(body, BaseException):
                    raise body
            return body, time.perf_counter() - t0

        packed = bool(getattr(self._reader, "packed", False))
        staged = {}                                   # packed route: slot -> [(offset in the staging buffer, rows)], positions

        def submit(body, slot):
            """the block's text (or packed cells) on its way to the device (slot 0 / 1); False: the fast path does not take it"""
            if packed:
                if not len(body):
                    return False
                nc = self._reader.n_cols
                total = sum(b.n for b in body) * nc
                parts, at = [], 0
                for b in body:
                    eng.stage_file(slot, b.fd, b.cells_off, b.n * nc, at, total)
                    parts.append((at, b.n))
                    at += b.n * nc
                staged[slot] = (parts, np.concatenate([b.positions() for b in body]).astype(np.int64, copy=False))
                return True
            if not len(body) or not hasattr(eng, "tokenize_submit"):
                return False
            if isinstance(body, genoio.BgzfSpan):     # members of a bgzip file, still deflated: inflated on the device
                ok = eng.tokenize_submit_bgzf(body, slot)
                self.timing["bgzf_blocks_inflated_on_device"] = self.timing.get("bgzf_blocks_inflated_on_device", 0) + int(ok)
                self.timing["bgzf_compressed_bytes"] = self.timing.get("bgzf_compressed_bytes", 0) + (len(body.comp) if ok else 0)
                return ok
            return eng.tokenize_submit(body, slot, file=self._reader.file_range(body) if hasattr(self._reader, "file_range") else None)

        def parse(slot, row_offset, cap):
            if not packed:
                return eng.tokenize_parse(slot, row_offset, cap)
            row = row_offset
            for at, n in staged[slot][0]:             # k_unpack per block of the file, queued on the copy stream
                eng.unpack_staged(slot, at, n, self._reader.n_cols, self.layout.slot_src, row)
                row += n
            return row - row_offset

        def collect(slot, body, n_lines):
            if not packed:
                return eng.tokenize_collect(slot, body, n_lines)
            eng.stage_sync()
            starts, names, row = [], [], 0
            for b in body:
                for s_, n_ in zip(b.starts, b.names):
                    if names and names[-1] == n_ and int(s_) == 0:
                        continue                      # the run continues across the seam of two blocks of the file
                    starts.append(row + int(s_))
                    names.append(n_)
                row += b.n
            return n_lines, staged.pop(slot)[1], np.asarray(starts, dtype=np.int64), names

        def ingest():
            carry, carry_row0, k = None, 0, 0
            try:
                t_e = time.perf_counter()
                body, read_s = fetch()
                self._ev("fetch", t_e)
                t_e = time.perf_counter()
                sub = submit(body, 0)
                self._ev("submit", t_e)
                while True:
                    final = self._streamer is None or len(body) == 0
                    tm = {"read_s": read_s, "host_tokenized": 0}
                    t0 = time.perf_counter()
                    while not halves.acquire(timeout=0.2):    # block k goes where block k-2 was: that one must be done
                        if stop.is_set():
                            return
                    tm["half_wait_s"] = time.perf_counter() - t0
                    self._ev("half_wait", t0)
                    t0 = time.perf_counter()
                    c_n = carry.n_sites if carry is not None else 0
                    if packed:
                        bound = sum(b.n for b in body)
                    else:
                        bound = row_bound(body) if len(body) else 0
                    if bound is None:
                        if isinstance(body, genoio.BgzfSpan):
                            body = bytes(body)
                        bound, sub = count(body), False       # (not a regular first line: the host tokenizer will take the block)
                    saved = None
                    if c_n + bound > st["half"]:
                        saved = grow(c_n + bound, carry_row0, c_n)
                    base = (k % 2) * st["half"]
                    if saved is not None:
                        eng.upload(saved, base)
                    elif c_n:
                        eng.move_rows(carry_row0, base, c_n)
                    # parse(k) is queued, then the text of block k+1 crosses PCIe while those kernels run, then the results of k
                    t_e = time.perf_counter()
                    n_lines = parse(k % 2, base + c_n, bound) if sub else None
                    self._ev("parse", t_e)
                    nxt, nxt_sub, read_s = None, False, 0.0
                    if not final:
                        t_e = time.perf_counter()
                        nxt, read_s = fetch()
                        self._ev("fetch", t_e)
                        t_e = time.perf_counter()
                        nxt_sub = submit(nxt, (k + 1) % 2)
                        self._ev("submit", t_e)
                    t_e = time.perf_counter()
                    got = collect(k % 2, body, n_lines) if n_lines is not None else None
                    self._ev("collect", t_e)
                    if got is not None:
                        n, pos, starts, names = got
                        block = genoio.GenoData(None, pos, starts, names)
                    else:                                     # a block the fast path refuses (or an empty one): host tokenizer
                        block = self._reader.to_geno(body, self.layout, n_threads=self._tok_threads)
                        if block.n_sites:
                            tm["host_tokenized"] = 1
                            if c_n + block.n_sites > st["half"]:          # (the bound holds for regular lines only)
                                saved = grow(c_n + block.n_sites, base, c_n)
                                base = (k % 2) * st["half"]
                                if saved is not None:
                                    eng.upload(saved, base)
                            eng.upload(np.ascontiguousarray(block.gt[:, :self.layout.n_hap]), base + c_n)
                        block = genoio.GenoData(None, block.pos, block.run_starts, block.run_names)
                    del body
                    data = genoio.concat_meta(carry, block)
                    if data is None:
                        data = genoio.GenoData(None, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), [])
                    tm["tokenize_s"] = time.perf_counter() - t0 - read_s
                    t0 = time.perf_counter()
                    if self._streamer is not None:
                        T, keep_from = self._streamer.feed(data.run_starts, data.run_names, data.pos, final)
                    else:
                        T = (self._windows_fn(data) if self._windows_fn
                             else _make_windows(self._wparams, data, self._minSites, self._coords_keep))
                        T.dup = np.zeros(T.n, dtype=bool)
                        keep_from = data.n_sites
                    tm["windows_s"] = time.perf_counter() - t0
                    self._ev("windows", t0)
                    carry = None if final else genoio.tail_meta(data, keep_from)
                    carry_row0 = base + keep_from
                    if not put(ready, (data, T, base, final, int(block.n_sites), tm)) or final:
                        return
                    body, sub = nxt, nxt_sub
                    k += 1
            except BaseException as exc:
                put(ready, exc)

        if not mapped:
            threading.Thread(target=produce, daemon=True, name="reader").start()
        threading.Thread(target=ingest, daemon=True, name="ingest").start()
        self.timing["device_tokenizer"] = 1
        self.timing["packed_cells_from_file"] = int(packed)
        self.timing["host_tokenized_blocks"] = 0
        if packed:
            lay = self.layout
            if len(lay.col_ploidy) != self._reader.n_cols:
                raise ValueError("layout was built for %d columns, the file has %d" % (len(lay.col_ploidy), self._reader.n_cols))
            wanted_cols = lay.col_ploidy > 0
            if np.any(lay.col_ploidy[wanted_cols] != self._reader.ploidy[wanted_cols]):
                bad = int(np.flatnonzero(wanted_cols & (lay.col_ploidy != self._reader.ploidy))[0])
                raise ValueError("sample %s was packed with ploidy %d but ploidy %d is requested" % (
                    self._reader.names[bad], int(self._reader.ploidy[bad]), int(lay.col_ploidy[bad])))
        import sys
        switch = sys.getswitchinterval()
        # the ingestion thread needs the interpreter for microseconds between two native calls, dozens of times per block: every time
        # it would wait a whole switch interval (5 ms by default) for this thread to let go while it formats rows
        sys.setswitchinterval(float(os.environ.get("PG_SWITCH_INTERVAL", "0.00005")))
        try:
            while True:
                t0 = time.perf_counter()
                item = ready.get()
                if isinstance(item, BaseException):
                    raise item
                self.timing["prep_wait_s"] += time.perf_counter() - t0      # time this thread waited for the ingestion thread
                self._ev("wait_ready", t0)
                data, T, base, final, n_new, tm = item
                for key in ("read_s", "tokenize_s", "windows_s"):
                    self.timing[key] += tm[key]
                self.timing["host_tokenized_blocks"] += tm["host_tokenized"]
                w0, w1 = dist.shard_range(T.n, self._wshare[0], self._wshare[1])
                lo, hi = T.lo[w0:w1] + base, T.hi[w0:w1] + base
                nz = T.hi[w0:w1] > T.lo[w0:w1]
                lo[~nz] = 0
                hi[~nz] = 0
                self.timing["text_bytes"] = self._reader.bytes_read
                self.data, self.T, self.w0, self.w1 = data, T, w0, w1
                self.lo, self.hi, self.site0 = lo, hi, 0
                self.timing["sites"] += n_new
                self.timing["windows"] += int(T.n)
                self.timing["chunks"] += 1
                self.n_tested += int(T.n)
                t_y = time.perf_counter()
                try:
                    if T.n:
                        yield self
                finally:
                    self._ev("chunk", t_y)
                    ready.task_done()                         # the rows of this block are no longer needed: its half may be rewritten
                    halves.release()
                if T.n:                                       # (what the caller did with the chunk: kernels, statistics, rows)
                    key = "compute_first_chunk_s" if "compute_first_chunk_s" not in self.timing else "compute_other_chunks_s"
                    self.timing[key] = self.timing.get(key, 0.0) + time.perf_counter() - t_y
                if final:
                    break
        finally:
            stop.set()
            sys.setswitchinterval(switch)
        self._reader.close()

    def report_timing(self):
        """Tier-T2 evidence (text end to end): per-phase wall seconds as one JSON line on stderr when PG_TIMING=1."""
        import json
        import os
        import time
        if os.environ.get("PG_TIMING") and (self.world.rank == 0 or self.sharded or self.cat_sharded):
            t = dict(self.timing)
            t["rank"], t["sharded_input"], t["input_bytes"] = self.world.rank, self.sharded or self.cat_sharded, self._reader.input_size()
            t["window_ranges"] = self.shard_plan is not None      # cuts inside scaffold runs (shardplan) / between runs only
            t["plan_scanned_bytes"] = int(self.shard_plan.scanned) if self.shard_plan is not None else 0
            t["total_s"] = time.perf_counter() - self._t_start
            # read / tokenize / windows run in their own threads: what this thread spent is the wait for them, the uploads it
            # waited for, and the statistics + output
            t["compute_and_write_s"] = t["total_s"] - t["prep_wait_s"] - t["engine_and_upload_s"]
            if t.get("device_tokenizer") and hasattr(self.engine, "tokenize_stats"):
                ts = self.engine.tokenize_stats()            # inside tokenize_s: the copies of the text (PCIe) and the kernels behind them
                t["tokenizer_h2d_s"], t["tokenizer_kernels_s"], t["tokenizer_bytes"] = ts["h2d_s"], ts["kernels_s"], ts["bytes"]
            gz = getattr(getattr(self._reader, "f", None), "stats", None)
            if gz is not None and not isinstance(getattr(self._reader, "f", None), genoio.BgzfFile):
                t["gzip_reader"] = gz()                      # ONE gzip stream: which decoder read it
            calls = getattr(getattr(self.engine, "_L", None), "calls", None)
            if calls:                                        # C-ABI calls per thread: [calls, seconds], the dozen largest
                top = sorted(calls.items(), key=lambda kv: -kv[1][1])[:14]
                t["lib_calls"] = {"%s:%s" % k: [v[0], round(v[1], 4)] for k, v in top}
            # (the stages overlap when tokenize_s + windows_s + compute_and_write_s > total_s - context_s)
            sys.stderr.write("PG_TIMING " + json.dumps(t) + "\n")
            if self._timeline is not None:
                sys.stderr.write("PG_TIMELINE " + json.dumps(self._timeline) + "\n")

    def finish(self):
        """the end of a driver: nobody leaves before everybody's rows are written; the communicator is closed (the file communicator
        removes its exchange files there)"""
        self.comm.barrier()
        self.comm.close()

    def batch(self, mask):
        """WindowBatch over this rank's windows selected by boolean `mask`."""
        if self._segments is not None:
            return MultiLayoutBatch(self, mask)
        return self.engine.batch(self.lo[mask], self.hi[mask])

    def layout_for(self, ploidy):
        """(HapLayout, columns of self.layout's rows it takes) for the ploidies `ploidy[i]` of self._segments.inds[i]: an individual's
        slots under fewer alleles are the first ones of its slots under self.layout (splitSeq zips the cells, genomics.py:390-396:
        what a shorter cell in the window leaves of the others is their first characters)"""
        key = bytes(np.asarray(ploidy, dtype=np.int32).data)
        cache = self.__dict__.setdefault("_layouts", {})
        if key not in cache:
            sd0 = self.layout.sampleData
            pl = dict(sd0.ploidy)
            pl.update({nm: int(v) for nm, v in zip(self._segments.inds, ploidy)})
            sd = SampleData(indNames=list(sd0.indNames), popNames=list(sd0.popNames), popInds=[list(sd0.popInds[p]) for p in sd0.popNames],
                            popNumbers=list(sd0.popNumbers), ploidyDict=pl)
            lay = HapLayout(sd, self._file_names, self.layout.genoFormat)
            cols = np.array([self.layout.ind_slots[nm][k] for nm in lay.ind_order for k in range(len(lay.ind_slots[nm]))], dtype=np.int64)
            cache[key] = (lay, cols)
        return cache[key]

    def gather(self, table):
        """the statistics of ALL windows of the current chunk on every rank (replicated ingestion: one all-gather per chunk);
        with sharded ingestion the chunk's windows are this rank's alone"""
        if self.sharded:
            return np.asarray(table, dtype=np.float64)
        return dist.gather_table(self.comm, table, self.T.n)

    def open_sink(self, path, header_text, id_column=False, id_sep=","):
        return _RowSink(self, path, header_text, id_column, id_sep)


def device_tokenizer_takes(layout):
    """the layouts pg_tokenize_text handles: any ploidies in the phased / pairs formats, diploid cells only in `diplo`, haploid
    only in `haplo`"""
    pl = set(int(x) for x in layout.col_ploidy if x > 0)
    fmt = layout.genoFormat
    return len(pl) >= 1 and not (fmt == "diplo" and pl != {2}) and not (fmt == "haplo" and pl != {1})


class _RowSink:
    """Where a table driver's finished rows go.  Replicated ingestion: rank 0 formats and writes everything (`local` is true
    there only).  Sharded ingestion: every rank formats the rows of its own slice; rank 0 writes its own as it goes, the others
    keep theirs, and close() brings them to rank 0 in ONE gather (rank order = input order) -- the sorter / writer threads of
    popgenWindows.py:108-157.  Window IDs count the windows of the whole input (genomics.py:2011): the ranks' local IDs are
    shifted by the number of windows of the ranks before them."""

    def __init__(self, run, path, header_text, id_column, id_sep=","):
        self.run, self.id_column, self.id_sep = run, id_column, id_sep
        self.local = run.sharded or run.world.rank == 0
        self.rows, self.written = [], 0
        self.out = None
        if run.world.rank == 0:
            self.out = _open_out(path)
            self.out.write(header_text)

    def write(self, text):
        if self.out is not None:
            self.out.write(text)
        else:
            self.rows.append(text)
        self.written += 1

    def close(self):
        """-> (windows tested, rows written) of the whole job"""
        run = self.run
        tested, written = run.n_tested, self.written
        if run.sharded:
            counts = run.comm.allgather(np.array([float(run.n_tested), float(self.written)])).reshape(run.world.size, 2)
            if self.id_column and run.world.rank > 0 and run.shift_ids:           # (predefined windows carry the IDs of their list)
                shift = int(counts[:run.world.rank, 0].sum())
                sep = self.id_sep
                self.rows = [str(int(r[:r.index(sep)]) + shift) + r[r.index(sep):] for r in self.rows]
            parts = dist.gather_bytes(run.comm, "".join(self.rows).encode() if run.world.rank > 0 else b"")
            if self.out is not None:
                for part in parts[1:]:
                    self.out.write(part.decode())
            tested, written = int(counts[:, 0].sum()), int(counts[:, 1].sum())
        if self.out is not None and self.out is not sys.stdout:
            self.out.close()
        return tested, written


def _fmt_cell(v):
    return str(v)


WIDE_ROW_COLS = 1                # popgenWindows.py: rows of at least this many float columns (and no other kind) are formatted natively (round 6: every such row -- 50 000 rows of 16 columns: 0.09 instead of 0.20 s; it was 256)
NP_MAX_SITES = 256               # csrc/pg_internal.h PG_NP_MAX_SITES: windows of up to this many sites get their sums in NumPy's order anyway


class MultiLayoutBatch:
    """--inferPloidy on an input whose cell widths change: the selected windows of the current chunk, grouped by the ploidies the
    reference would infer for them (per window and sample: the fewest alleles a cell of the sample holds in the window,
    genomics.py:1108-1111, 390-396).  Any WindowBatch method called on it runs group by group -- the engine gets the group's layout
    (pg_set_samples: other haplotype counts, names and sort order), the group's rows with the columns that layout keeps, and the
    group's windows -- and the results are put back in window order.  Results: arrays over the windows, dictionaries / tuples of
    such (the keys are population and individual names, the same under every layout)."""

    def __init__(self, run, mask):
        self.run = run
        self.lo, self.hi = run.lo[mask], run.hi[mask]            # rows of run.data, counted from run.site0
        g = int(run.data.row0) + int(run.site0)
        pl = run._segments.window_ploidy(g + self.lo, g + self.hi)
        groups = {}
        for w in range(len(self.lo)):
            groups.setdefault(pl[w].tobytes(), []).append(w)
        self.groups = [(np.frombuffer(k, dtype=np.int32), np.array(v, dtype=np.int64)) for k, v in groups.items()]
        self.n = len(self.lo)
        self._batches = {}
        run._ml_count = run.__dict__.get("_ml_count", 0) + 1
        self._token = run._ml_count                   # (not id(self): a later object may get the address of a dead one)

    def _each(self):
        """(windows of the group, its WindowBatch) group by group.  A group's batch object lives as long as this object -- the
        reference's Alignment carries state from one statistic to the next (groupDistStats leaves its minSites mask and a nan
        diagonal in the cached distance matrix, genomics.py:959-963, which indPairDists / sampleHet / H12stats then see) and so
        does WindowBatch --; what is loaded again when the engine holds another group is the layout and the rows."""
        run, eng = self.run, self.run.engine
        for g, (ploidy, sel) in enumerate(self.groups):
            lo, hi = self.lo[sel], self.hi[sel]
            r0, r1 = int(lo.min()), int(hi.max())
            if run.__dict__.get("_ml_loaded") != (self._token, g):
                lay, cols = run.layout_for(ploidy)
                rows = np.ascontiguousarray(run.data.gt[run.site0 + r0:run.site0 + r1][:, cols])
                first = eng.__dict__.get("_pair_first", "slot")
                eng.set_layout(lay)
                if first != "slot" and hasattr(eng, "set_pair_first"):
                    eng.set_pair_first(first)
                eng.load_sites(rows)
                run._ml_loaded = (self._token, g)
            if g not in self._batches:
                self._batches[g] = eng.batch(lo - r0, hi - r0)
            yield sel, self._batches[g]

    def _merge(self, parts, pad=None):
        first = parts[0][1]
        if isinstance(first, dict):
            return {k: self._merge([(sel, r[k]) for sel, r in parts], pad) for k in first}
        if isinstance(first, tuple):
            return tuple(self._merge([(sel, r[i]) for sel, r in parts], pad) for i in range(len(first)))
        if isinstance(first, list) and all(isinstance(x, str) for x in first):
            return first                                          # column names
        arrs = [(sel, np.asarray(r)) for sel, r in parts]
        for sel, a in arrs:
            if a.ndim < 1 or a.shape[0] != len(sel):
                raise NotImplementedError("a result that is not an array over the windows, under --inferPloidy with changing ploidy")
        trail = tuple(max(a.shape[d] for _, a in arrs) for d in range(1, arrs[0][1].ndim))
        if pad is None and any(a.shape[1:] != trail for _, a in arrs):
            raise NotImplementedError("a per-haplotype result under --inferPloidy with changing ploidy")
        out = np.full((self.n,) + trail, 0 if pad is None else pad, dtype=np.result_type(*[a.dtype for _, a in arrs]))
        for sel, a in arrs:
            out[(sel,) + tuple(slice(0, d) for d in a.shape[1:])] = a
        return out

    def hapCalled(self):
        """called sites per haplotype; the windows' haplotype counts differ: padded with the largest integer (distMat.py:40 takes the minimum)"""
        return self._merge([(sel, b.hapCalled()) for sel, b in self._each()], pad=np.iinfo(np.int64).max)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)

        def call(*args, **kwargs):
            return self._merge([(sel, getattr(b, name)(*args, **kwargs)) for sel, b in self._each()])
        return call


def _near_rounding_tie(v, digits, ratio=False, difference=False):
    """which values could print differently with other last bits of a float64 sum: within reach of a rounding tie of the printed
    digit, of zero (its sign), or -- ratios of sums -- so large / infinite that a denominator is rounding noise.  The fixed-tree sums
    of a long window differ from NumPy's by at most ~ n eps (1e-11 relative for 10^5 terms)."""
    v = np.asarray(v, dtype=np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        s = np.abs(v) * 10.0 ** digits
        d = np.abs(s - np.floor(s) - 0.5)
        near = np.isfinite(v) & ((d < np.maximum(1e-6, s * 1e-10)) | (np.abs(v) < 1e-12))
        if difference or ratio:
            # 1 - pi_s / pi_t, (ABBA - BABA) / (ABBA + BABA), ...: what is printed is what is LEFT of sums that cancel, so the error of
            # the value is absolute (n eps of the sums' magnitude: 1e-11 bounds it relative to a denominator of the value's own
            # scale), not relative to the small value
            near |= np.isfinite(v) & (d < 10.0 ** digits * 1e-11)
        if ratio:
            near |= np.isinf(v) | (np.isfinite(v) & (np.abs(v) > 100.0))
    return near


def _refine_long_windows(run, good, sites_local, sd, digits, again, ratio_keys=(), also=None):
    """The statistics `sd` of this rank's windows `good` were formed with fixed reduction trees where a window has more than NP_MAX_SITES (256)
    sites.  Where one of them is within reach of a rounding tie (so that the reference's summation order could print another
    digit), that window is computed again in NumPy's order (`again(batch)`): the text is the reference's for every window length
    without paying for that order everywhere."""
    idx = np.flatnonzero(good)
    long_w = sites_local[idx] > NP_MAX_SITES
    if not np.any(long_w):
        return
    near = np.zeros(len(idx), dtype=bool) if also is None else np.asarray(also, dtype=bool).copy()
    for key, v in sd.items():
        if np.asarray(v).dtype.kind == "f":
            near |= _near_rounding_tie(v, digits, ratio=key in ratio_keys, difference=key.startswith("Fst_"))
    flagged = long_w & near
    if not np.any(flagged):
        return
    mask = np.zeros(len(good), dtype=bool)
    mask[idx[flagged]] = True
    run.engine.set_sum_order(1)
    try:
        sd2 = again(run.batch(mask))
    finally:
        run.engine.set_sum_order(0)
    for key in sd:
        a = np.array(sd[key], copy=True)
        a[flagged] = sd2[key]
        sd[key] = a
    run.timing["windows_recomputed_in_numpy_order"] = run.timing.get("windows_recomputed_in_numpy_order", 0) + int(flagged.sum())


def guarded_main(fn):
    """A driver's entry point under a multi-rank launch: whatever ends this rank -- a parse error in its share of the input, an
    assertion, a failed library call -- is left as a marker next to the launch's rendezvous file (dist.mark_failed) before it
    propagates, and every wait of the other ranks (the exchange of the shard plan, the gather of the rows, the final barrier) looks
    for such markers: they stop within a fraction of a second with one line naming the failed rank, instead of hanging the way the
    reference does when a worker dies (popgenWindows.py:456-460) or sitting out PG_COMM_TIMEOUT."""
    import functools

    @functools.wraps(fn)
    def wrapper(argv=None):
        try:
            return fn(argv)
        except BaseException as exc:
            world = dist.world_from_env()
            if world.size > 1 and not (isinstance(exc, SystemExit) and exc.code in (0, None)):
                dist.mark_failed(world, exc)
                if isinstance(exc, dist.PeerFailed):
                    sys.stderr.write("rank %d stops: %s\n" % (world.rank, exc))
                    sys.stderr.flush()
                    os._exit(3)                     # (helper threads of the ingestion may still be inside the library)
            raise
    return wrapper


# ==========================================================================================================
# popgenWindows.py
# ==========================================================================================================
@guarded_main
def popgen_main(argv=None):
    ap = argparse.ArgumentParser(prog="popgenWindows.py", epilog=ENGINE_EPILOG)
    _add(ap, WINDOW_FLAGS, **{"--windType": dict(choices=("sites", "coordinate", "predefined"))})
    ap.add_argument("-O", "--overlap", type=int, metavar="sites", help="Overlap for sites sliding window")
    ap.add_argument("--minData", type=float, metavar="prop", default=0.01,
                    help="Minimum proportion of individuals (or pairs) with >=minSites data")
    ap.add_argument("-p", "--population", action="append", nargs="+", metavar=("popName", "[samples]"),
                    help="Pop name and optionally sample names (separated by commas)")
    ap.add_argument("--popsFile", help="Optional file of sample names and populations")
    ap.add_argument("--samples", metavar="sample names", help="Samples to include for individual analysis")
    _add(ap, PLOIDY_FLAGS)
    ap.add_argument("--haploid", metavar="sample names", help="Samples that are haploid (comma separated)")
    ap.add_argument("--analysis", nargs="+", default=("popDist", "popPairDist"),
                    choices=("popFreq", "popDist", "popPairDist", "indPairDist", "indHet", "hapStats"),
                    help="Type of statistics to get")
    ap.add_argument("--hapDist", type=float, default=0)
    ap.add_argument("--roundTo", type=int, default=4, help="Round stats to X decimal places")
    ap.add_argument("--header", help="Header text if no header in input")
    ap.add_argument("-T", "--threads", type=int, default=1, metavar="threads",
                    help="accepted for compatibility; the GPU engine does not use worker processes")
    _add(ap, IO_FLAGS)
    args = ap.parse_args(argv)

    wp = _window_setup(args, args.overlap)
    minSites = args.minSites
    if not minSites:
        minSites = args.windSize
    # samples and populations (popgenWindows.py:259-291)
    popNames, popInds, allInds = [], [], []
    if args.population is not None:
        for p in args.population:
            popNames.append(p[0])
            popInds.append(p[1].split(",") if len(p) > 1 else [])
        if args.popsFile:
            with open(args.popsFile, "rt") as pf:
                for ind, pop in (ln.split()[:2] for ln in pf if ln.strip()):
                    if pop in popNames:
                        popInds[popNames.index(pop)].append(ind)
        for p in popInds:
            assert len(p) >= 1, "All populations must be represented by at least one sample."
        for p in popInds:
            for i in p:
                if i not in allInds:
                    allInds.append(i)
    if args.samples is not None:
        for i in args.samples.split(","):
            if i not in allInds:
                allInds.append(i)
    if len(allInds) == 0:
        assert args.genoFile, "sample names are read from the file header: give -g, or -p/--samples when piping"
        allInds = genoio.read_header_names(args.genoFile)
    pop_analysis = any(a in args.analysis for a in ("popFreq", "popDist", "popPairDist", "hapStats"))
    if len(popNames) == 0 and pop_analysis:
        popNames.append("all")
        popInds.append(list(allInds))
    ploidyDict = _ploidy_dict(args, allInds, args.haploid.split(",") if args.haploid else None)
    sampleData = SampleData(indNames=list(allInds), popNames=popNames, popInds=popInds, ploidyDict=ploidyDict)
    # (the reference trips over such a sample -- a TypeError, or a hang -- in the first window it computes statistics for, not before:
    # a run all of whose windows fail --minSites writes its rows of nan; the assertion waits for the first good window too)
    nopop = [i for i in sampleData.indNames if sampleData.getPop(i) is None] if pop_analysis else []

    # statistics, in output order (popgenWindows.py:326-354)
    stats = []
    if "popFreq" in args.analysis:
        for pre in ("l_", "S_", "thetaPi_", "thetaW_", "TajD_"):
            stats += [pre + n for n in popNames]
    if "popDist" in args.analysis:
        stats += ["pi_" + n for n in popNames]
    if "popPairDist" in args.analysis:
        stats += ["dxy_" + x + "_" + y for x, y in itertools.combinations(popNames, 2)]
        stats += ["Fst_" + x + "_" + y for x, y in itertools.combinations(popNames, 2)]
    if "indPairDist" in args.analysis:
        stats += ["_".join(["d", i, j]) for i, j in itertools.combinations_with_replacement(sorted(allInds), 2)]
    if "indHet" in args.analysis:
        stats += ["het_" + n for n in allInds]
    if "hapStats" in args.analysis:
        for pre in ("H1_", "H12_", "H2_"):
            stats += [pre + n for n in popNames]
    int_stat = [s.startswith("l_") or s.startswith("S_") for s in stats]
    # H12stats answers a population that is ONE cluster with the integer `H2 = 0` (genomics.py:1092-1093): printed "0", not "0.0"
    # (H2 of two or more clusters is a sum of positive squares, never exactly zero)
    h2_stat = [s.startswith("H2_") and "hapStats" in args.analysis for s in stats]
    int_cols = [c for c, f in enumerate(int_stat) if f]
    h2_cols = [c for c, f in enumerate(h2_stat) if f]

    run = Run(args, sampleData, wp, minSites, header_line=args.header, coords_keep=3, stream=True, shardable=True)
    sink = run.open_sink(args.outFile, ("windowID," if args.addWindowID else "") + "scaffold,start,end,mid,sites," + ",".join(stats) + "\n",
                         id_column=args.addWindowID)
    last_row = None                      # (ok, text) of the previously emitted window: a dup row repeats it verbatim
    import time
    tm = run.timing
    tm.update(main_stats_s=0.0, main_refine_s=0.0, main_format_s=0.0)    # where the main thread's time goes inside a chunk (PG_TIMING)
    for _ in run.chunks():
        T = run.T
        sites_local = T.sites[run.w0:run.w1]
        good = sites_local >= minSites
        table = np.full((run.w1 - run.w0, len(stats)), np.nan)
        t_c = time.perf_counter()
        if np.any(good) and stats:
            assert not nopop, "samples without a population cannot be mixed with population statistics: " + ",".join(nopop[:5])
            wb = run.batch(good)
            sd = {}
            if "popFreq" in args.analysis:
                sd.update(wb.groupFreqStats())
            if "popDist" in args.analysis or "popPairDist" in args.analysis:
                def dist_stats(b):
                    return b.groupDistStats(doPairs="popPairDist" in args.analysis, minSites=minSites, minData=args.minData)
                gd = dist_stats(wb)
                t_r = time.perf_counter()
                _refine_long_windows(run, good, sites_local, gd, args.roundTo, dist_stats)
                tm["main_refine_s"] += time.perf_counter() - t_r
                sd.update(gd)
            if "indPairDist" in args.analysis:
                pdd = wb.indPairDists()
                for i, j in itertools.combinations_with_replacement(sorted(pdd.keys()), 2):
                    sd["_".join(["d", i, j])] = pdd[i][j]
            if "indHet" in args.analysis:
                for k, v in wb.sampleHet().items():
                    sd["het_" + k] = v
            if "hapStats" in args.analysis:
                sd.update(wb.H12stats(maxDist=args.hapDist))
            for c, s in enumerate(stats):
                table[good, c] = sd[s]
        tm["main_stats_s"] += time.perf_counter() - t_c
        full = run.gather(table)
        if not sink.local:
            continue
        t_c = time.perf_counter()
        # the rows as text: one rounding pass over the table (np.round is what round(np.float64) calls), then Python numbers, whose
        # str() is numpy's for float64 (shortest repr, "nan", "inf", "-0.0") -- no numpy scalar per cell (popgenWindows.py:66-75)
        # (rows of hundreds of float columns -- indPairDist of many individuals -- and nothing but floats: formatted natively, all at once)
        wide = full.shape[1] >= WIDE_ROW_COLS and not int_cols and not h2_cols and T.n > 0
        R = _float_rows(full, args.roundTo, sep=",").split("\n") if wide else np.round(full, args.roundTo).tolist()
        ids, start, end, mid = T.ID, T.start, T.end, T.mid
        sites, dup = np.asarray(T.sites).tolist(), np.asarray(T.dup).tolist()
        for k in range(T.n):
            if dup[k]:
                ok, text = last_row
            else:
                ok = sites[k] >= minSites
                head = ([ids[k]] if args.addWindowID else []) + [T.scaffold[k], start[k], end[k], mid[k], int(sites[k])]
                if wide:
                    text = ",".join(map(str, head)) + "," + R[k] + "\n"
                else:
                    vals = R[k]
                    for c in int_cols:
                        if vals[c] == vals[c]:
                            vals[c] = int(vals[c])
                    for c in h2_cols:
                        if full[k, c] == 0:                         # (before rounding: a small H2 that ROUNDS to zero prints "0.0")
                            vals[c] = 0
                    text = ",".join(map(str, head + vals)) + "\n"
                last_row = (ok, text)
            if not (ok or args.writeFailedWindows):
                continue
            sink.write(text)
        tm["main_format_s"] += time.perf_counter() - t_c
    tested, written = sink.close()
    if run.world.rank == 0:
        sys.stderr.write(str(tested) + " windows were tested.\n")
        sys.stderr.write(str(written) + " results were written.\n")
        sys.stderr.write("\nDone.\n")
    run.report_timing()
    run.finish()
    return 0


# ==========================================================================================================
# ABBABABAwindows.py
# ==========================================================================================================
FOURPOP_STATS = ["ABBA", "BABA", "ABAA", "BAAA", "D", "fd", "fd'", "fdm", "fdm'", "fdh", "fdh2", "fh"]   # fourPopWindows.py:241


@guarded_main
def abbababa_main(argv=None):
    return _quartet_main(argv, "ABBABABAwindows.py", ["ABBA", "BABA", "D", "fd", "fdM"], fourpop=False)


@guarded_main
def fourpop_main(argv=None):
    """fourPopWindows.py:105-150 flag table; statistics genomics.py:1585-1643."""
    return _quartet_main(argv, "fourPopWindows.py", FOURPOP_STATS, fourpop=True)


def _quartet_main(argv, prog, stats, fourpop):
    ap = argparse.ArgumentParser(prog=prog, epilog=ENGINE_EPILOG)
    _add(ap, WINDOW_FLAGS, **{"--windType": dict(choices=("sites", "coordinate", "predefined"))})
    ap.add_argument("--overlap", type=int, metavar="sites", help="Overlap for sites sliding window")
    ap.add_argument("--minData", type=float, metavar="proportion", default=0.01,
                    help="Min proportion of samples genotyped per site")
    for flag, dest, what in (("-P1", "pop1", "P1"), ("-P2", "pop2", "P2"), ("-P3", "pop3", "P3"), ("-O", "outgroup", "outgroup")):
        ap.add_argument(flag, "--" + dest, dest=dest, nargs="+", required=True, metavar=("popName", "[samples]"),
                        help="Pop name and optionally sample names for " + what)
    ap.add_argument("--popsFile", help="Optional file of sample names and populations")
    _add(ap, PLOIDY_FLAGS)
    ap.add_argument("--haploid", metavar="sample names", help="Samples that are haploid (comma separated)")
    ap.add_argument("--header", help="Header text if no header in input")
    ap.add_argument("-T", "--Threads", type=int, default=1, help="accepted for compatibility")
    if fourpop:
        ap.add_argument("--polarize", action="store_true", help="Ensure outgroup is fixed for ancestral allele")
        ap.add_argument("--fixed", action="store_true", help="Only count fixed SNPs")
    _add(ap, IO_FLAGS)
    args = ap.parse_args(argv)

    wp = _window_setup(args, args.overlap)
    minSites = args.minSites
    if not minSites:
        minSites = args.windSize
    minData = args.minData
    assert 0 <= minData <= 1, "minimum data per site must be between 0 and 1."

    popNames, popInds = [], []
    for p in (args.pop1, args.pop2, args.pop3, args.outgroup):
        popNames.append(p[0])
        popInds.append(p[1].split(",") if len(p) > 1 else [])
    if args.popsFile:
        with open(args.popsFile, "rt") as pf:
            for ind, pop in (ln.split()[:2] for ln in pf if ln.strip()):
                if pop in popNames:
                    popInds[popNames.index(pop)].append(ind)
    for p in popInds:
        assert len(p) >= 1, "All populations must be represented by at least one sample."
    allInds = []
    for p in popInds:
        for i in p:
            if i not in allInds:
                allInds.append(i)
    ploidyDict = _ploidy_dict(args, allInds, args.haploid.split(",") if args.haploid else None)
    sampleData = SampleData(popNames=popNames, popInds=popInds, ploidyDict=ploidyDict)

    run = Run(args, sampleData, wp, minSites, header_line=args.header, coords_keep=4, stream=True, shardable=True)
    sink = run.open_sink(args.outFile, ("windowID," if args.addWindowID else "") + "scaffold,start,end,mid,sites,sitesUsed," +
                         ",".join(stats) + "\n", id_column=args.addWindowID)
    last_row = None                      # (ok, text) of the previously emitted window: a dup row repeats it verbatim
    for _ in run.chunks():
        T = run.T
        sites_local = T.sites[run.w0:run.w1]
        good = sites_local >= minSites
        table = np.full((run.w1 - run.w0, 1 + len(stats)), np.nan)           # sitesUsed + the statistics
        if np.any(good):
            def quartet_stats(b):
                if fourpop:
                    return b.fourPop(popNames[0], popNames[1], popNames[2], popNames[3], minData, polarize=args.polarize, fixed=args.fixed)
                return b.ABBABABA(popNames[0], popNames[1], popNames[2], popNames[3], minData)
            sd = quartet_stats(run.batch(good))
            # (ratios of sums: a nan beside used sites is a 0 / 0 of sums that cancel -- in another order they may not)
            ratios = [s_ for s_ in stats if s_ not in ("ABBA", "BABA", "ABAA", "BAAA")]
            with np.errstate(invalid="ignore"):
                odd = np.zeros(int(good.sum()), dtype=bool)
                for s_ in ratios:
                    odd |= np.isnan(sd[s_]) & (np.nan_to_num(np.asarray(sd["sitesUsed"], dtype=np.float64)) > 0)
            _refine_long_windows(run, good, sites_local, sd, 4, quartet_stats, ratio_keys=ratios, also=odd)
            table[good, 0] = sd["sitesUsed"]
            for c, s in enumerate(stats):
                table[good, 1 + c] = sd[s]
        full = run.gather(table)
        if not sink.local:
            continue
        for k in range(T.n):
            if T.dup[k]:
                ok, text = last_row
            else:
                used = full[k, 0]
                ok = T.sites[k] >= minSites and used >= minSites           # ABBABABAwindows.py:35-46, fourPopWindows.py:36-48
                vals = [round(np.float64(v), 4) for v in full[k, 1:]] if ok else [np.nan] * len(stats)
                used_cell = int(used) if used == used else np.nan
                row = ([T.ID[k]] if args.addWindowID else []) + [T.scaffold[k], T.start[k], T.end[k], T.mid[k], int(T.sites[k]), used_cell] + vals
                text = ",".join(_fmt_cell(x) for x in row) + "\n"
                last_row = (ok, text)
            if not (ok or args.writeFailedWindows):
                continue
            sink.write(text)
    tested, written = sink.close()
    if run.world.rank == 0:
        sys.stderr.write("%d windows were tested\n%d results were written\n\nDone.\n" % (tested, written))
    run.report_timing()
    run.finish()
    return 0


# ==========================================================================================================
# distMat.py
# ==========================================================================================================
def _float_rows(M, roundTo, prefixes=None, sep=" "):
    """the rows of a float64 matrix as text: `M.round(roundTo).astype(str)` joined by blanks, a line feed behind every row, an
    optional prefix in front of each (pg_format_float_rows: repr(float) natively, on the host threads)"""
    import ctypes as C
    M = np.ascontiguousarray(M, dtype=np.float64)
    n, m = M.shape
    blob, off = None, None
    if prefixes is not None:
        enc = [p.encode() for p in prefixes]
        blob = b"".join(enc)
        off = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.int64)
    cap = n * (m * 26 + 1) + (len(blob) if blob else 0) + 16
    out = np.empty(cap, dtype=np.uint8)
    got = C.c_int64(0)
    _lib.check(_lib.lib().pg_format_float_rows(C.c_void_p(M.ctypes.data), n, m, int(roundTo), C.c_char(sep.encode()), blob,
                                               C.c_void_p(off.ctypes.data) if off is not None else None, C.c_void_p(out.ctypes.data), cap,
                                               C.byref(got), 0))
    return out[:got.value].tobytes().decode()


def _matrix_text(M, names, fmt, roundTo):
    """genomics.py:2288-2306 makeDistMatString / PhylipString / NexusString."""
    n = len(names)
    if fmt == "raw":
        return _float_rows(M, roundTo)
    if fmt == "phylip":
        return str(M.shape[0]) + "\n" + _float_rows(M, roundTo, [str(names[i]) + "  " for i in range(n)])
    s = "\nBEGIN Taxa;\nDIMENSIONS ntax={};\nTAXLABELS\n".format(n)
    s += "".join("[{}] '{}'\n".format(i + 1, names[i]) for i in range(n))
    s += ";\nEND; [Taxa]\n"
    s += "\nBEGIN Distances;\nDIMENSIONS ntax={};\nFORMAT labels=left diagonal triangle=both;\nMATRIX\n".format(n)
    s += _float_rows(M, roundTo, ["[{}] '{}'    ".format(i + 1, names[i]) for i in range(n)])
    return s + ";\nEND; [Distances]\n"


@guarded_main
def distmat_main(argv=None):
    ap = argparse.ArgumentParser(prog="distMat.py", epilog=ENGINE_EPILOG)
    _add(ap, WINDOW_FLAGS, **{"--windType": dict(choices=("sites", "coordinate", "predefined", "cat"))})      # -m: default 1 (distMat.py:123)
    ap.add_argument("-O", "--overlap", type=int, metavar="sites", help="Overlap for sites sliding window")
    ap.add_argument("-Mi", "--minPerInd", type=int, metavar="sites", help="Minimum sites per individual")
    ap.add_argument("--includeSameWithSame", action="store_true", help="Include comparisons of each haplotype to itself")
    ap.add_argument("--outFormat", choices=("raw", "phylip", "nexus"), default="phylip")
    ap.add_argument("--roundTo", type=int, default=4)
    ap.add_argument("--headers", nargs="+", help="Header fields if the input has no header line")
    ap.add_argument("--windowDataOutFile", help="Optional file for window coordinates")
    ap.add_argument("--samples", nargs="+", help="Samples to include")
    _add(ap, PLOIDY_FLAGS)
    ap.add_argument("--haploid", nargs="+", metavar="sample", help="Samples that are haploid")
    ap.add_argument("-T", "--threads", type=int, default=1, help="accepted for compatibility")
    _add(ap, IO_FLAGS)
    args = ap.parse_args(argv)

    minSites = args.minSites
    if args.windType == "cat":
        wp = dict(windType="cat", windSize=None, stepSize=None, overlap=0, maxDist=np.inf, windCoords=None)
        minSites = 1
    else:
        wp = _window_setup(args, args.overlap)
    if not minSites:
        minSites = args.windSize

    if args.samples:
        samples = list(args.samples)
    elif args.headers:
        samples = list(args.headers[2:])
    else:
        assert args.genoFile, "If piping from stdin, you need to specify either --samples or --headers"
        samples = genoio.read_header_names(args.genoFile)
    ploidyDict = _ploidy_dict(args, samples, args.haploid)
    sampleData = SampleData(indNames=list(samples), ploidyDict=ploidyDict)
    header_line = "\t".join(args.headers) if args.headers else None

    def cat_window(data):
        # parseGenoFile (genomics.py:1949-1967): every site of the file, positions ignored
        T = windows.WindowTable()
        T.add(None, float("nan"), float("nan"), 0, data.n_sites, None)      # first / last of a list of nan positions (distMat.py:36)
        T.finish(data.pos)
        T.mid = [float("nan")]
        return T

    # (coordinate and sites windows: the input is sharded over the ranks at scaffold-run boundaries like popgenWindows.py's, every
    # rank formats the matrices of its own windows, one gather per output file at the end: distMat.py:28-60, 284-289)
    run = Run(args, sampleData, wp, minSites, header_line=header_line, coords_keep=3,
              windows_fn=cat_window if args.windType == "cat" else None, stream=True, shardable=True)
    lay = run.layout
    n = len(samples)
    npairs = n * (n + 1) // 2
    # column (i<=j in --samples order) <- pair index in the engine's slot order of the individuals
    pos_of = {nm: k for k, nm in enumerate(lay.ind_order)}
    si = np.array([pos_of[nm] for nm in samples], dtype=np.int64)
    iu = np.triu_indices(n)
    a, b = np.minimum(si[iu[0]], si[iu[1]]), np.maximum(si[iu[0]], si[iu[1]])
    pair_col = a * n - a * (a - 1) // 2 + (b - a)
    out = run.open_sink(args.outFile, "")
    # the reference writes the side file's header without a newline and tab-separated rows after it (distMat.py:238-239, 58)
    wout = (run.open_sink(args.windowDataOutFile, ("windowID," if args.addWindowID else "") + "scaffold,start,end,mid,sites,",
                          id_column=args.addWindowID, id_sep="\t") if args.windowDataOutFile else None)
    last = None                          # (ok, matrix text, window-data text) of the previously emitted window (dup rows)
    for _ in run.chunks():
        T = run.T
        sites_local = T.sites[run.w0:run.w1]
        good = sites_local >= minSites
        table = np.full((run.w1 - run.w0, npairs + 1), np.nan)               # pair means + minPerInd verdict
        table[:, npairs] = 1.0
        if run.cat_sharded:
            # this rank's window is its share of the lines: sites, called counts and pair counts are summed over the ranks
            # (dist.sum_counts: one all-gather each), every rank finishes the one matrix from the sums
            H = lay.n_hap
            wb = run.batch(good) if np.any(good) else None
            called = wb.hapCalled()[0] if wb is not None else np.zeros(H, dtype=np.int64)
            head = dist.sum_counts(run.comm, np.concatenate([[int(sites_local[0])], called]))
            D, Cc = wb.pairCounts(reference_order=False) if wb is not None else (np.zeros((1, H, H), np.int32),) * 2
            D, Cc = dist.sum_counts(run.comm, D), dist.sum_counts(run.comm, Cc)
            assert head[0] < 2 ** 31, "more than 2^31 sites in one window"
            T.sites[0] = head[0]
            if head[0] >= minSites:
                if args.minPerInd:
                    table[0, npairs] = float(head[1:].min() >= args.minPerInd)
                tab = run.engine.indPairTableFromCounts(D, Cc, includeSameWithSame=args.includeSameWithSame)
                table[0, :npairs] = tab[0, pair_col]
            full = table
        elif np.any(good):
            wb = run.batch(good)
            if args.minPerInd:
                called = wb.hapCalled()
                table[good, npairs] = (called.min(axis=1) >= args.minPerInd).astype(np.float64)
            tab = wb.indPairTable(includeSameWithSame=args.includeSameWithSame)
            table[good, :npairs] = tab[:, pair_col]
        if not run.cat_sharded:
            full = run.gather(table)
        if not out.local:
            continue
        for k in range(T.n):
            if T.dup[k]:
                ok, mtext, wtext = last
            else:
                ok = T.sites[k] >= minSites and full[k, npairs] > 0
                M = np.full((n, n), np.nan)
                if ok:
                    M[iu] = full[k, :npairs]
                    M[(iu[1], iu[0])] = full[k, :npairs]
                mtext = _matrix_text(M, samples, args.outFormat, args.roundTo) if (ok or args.writeFailedWindows) else ""
                wd = ([T.ID[k]] if args.addWindowID else []) + [T.scaffold[k], T.start[k], T.end[k], T.mid[k], int(T.sites[k])]
                wtext = "\t".join(str(x) for x in wd) + "\n"
                last = (ok, mtext, wtext)
            if not (ok or args.writeFailedWindows):
                continue
            out.write(mtext)
            if wout is not None:
                wout.write(wtext)
    tested, written = out.close()
    if wout is not None:
        wout.close()
    if run.world.rank == 0:
        sys.stderr.write("{} windows were tested.\n{} results were written.\n\n### Done. ###\n".format(tested, written))
    run.report_timing()
    run.finish()
    return 0


# ==========================================================================================================
# freq.py  (SURVEY.md 8f "next" row 1: the raw output of the per-site population count kernel as a TSV)
# ==========================================================================================================
@guarded_main
def freq_main(argv=None):
    import time as _time
    t_begin = _time.perf_counter()                          # (PG_TIMING total_s: from here, the device context included)
    """Drop-in for the reference's freq.py (freq.py:30-113, 192-300): per-site per-population base counts, or the
    frequency / count of a target allele (`--target derived|minor`).  Counts come from k_site_counts (pg_site_counts).
    Divergence: for `--target minor` the reference breaks count ties with np.random.choice (genomics.py:664-669); here the
    tied allele with the lower base index is taken."""
    ap = argparse.ArgumentParser(prog="freq.py", epilog=ENGINE_EPILOG)
    ap.add_argument("-g", "--genoFile", help="Input geno file")
    ap.add_argument("-o", "--outFile", help="Output file")
    ap.add_argument("-f", "--genoFormat", choices=("phased", "diplo", "alleles"), default="phased")
    ap.add_argument("-p", "--population", action="append", nargs="+", metavar=("popName", "[samples]"))
    ap.add_argument("--popsFile")
    ap.add_argument("--indFreqs", action="store_true", help="treat every individual as its own population")
    ap.add_argument("--target", choices=("minor", "derived"), default=None)
    ap.add_argument("--asCounts", action="store_true")
    ap.add_argument("--ploidy", type=int, nargs="+")
    ap.add_argument("--ploidyFile")
    ap.add_argument("--haploid", nargs="+")
    ap.add_argument("--minData", type=float, default=0, metavar="proportion")
    ap.add_argument("--threshold", type=float, metavar="proportion")
    ap.add_argument("--keepNanLines", action="store_true")
    ap.add_argument("-t", "--threads", type=int, default=1, help="accepted for compatibility")
    ap.add_argument("--sliceSize", type=int, default=1000000, help="accepted for compatibility")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--test", action="store_true", help="accepted for compatibility")
    ap.add_argument("--device", type=int, default=None, help="GPU index (MI355X engine)")
    args = ap.parse_args(argv)

    reader = genoio.open_input(args.genoFile)                              # blocks of PG_STREAM_BYTES: bounded host memory
    headerInds = reader.read_header().decode("utf-8", "replace").split()[2:]
    if not args.indFreqs and not args.population:
        if args.target == "derived":
            popNames, popInds = ["ingroup", "outgroup"], [headerInds[:-1], [headerInds[-1]]]
        else:
            popNames, popInds = ["all"], [list(headerInds)]
    elif args.indFreqs:
        popNames, popInds = list(headerInds), [[i] for i in headerInds]
    else:
        popNames, popInds = [], []
        for p in args.population:
            popNames.append(p[0])
            popInds.append(p[1].split(",") if len(p) > 1 else [])
        if args.popsFile:
            with open(args.popsFile, "rt") as pf:
                for ind, pop in (ln.split()[:2] for ln in pf if ln.strip()):
                    if pop in popNames:
                        popInds[popNames.index(pop)].append(ind)
        for p in popInds:
            assert len(p) >= 1, "All populations must be represented by at least one sample."
    allInds = []
    for p in popInds:
        for i in p:
            if i not in allInds:
                allInds.append(i)
    if args.ploidy is not None:
        pl = args.ploidy if len(args.ploidy) != 1 else args.ploidy * len(allInds)
        assert len(pl) == len(allInds), "Incorrect number of ploidy values supplied."
        ploidyDict = dict(zip(allInds, pl))
    elif args.ploidyFile is not None:
        with open(args.ploidyFile, "rt") as pf:
            ploidyDict = dict([[s[0], int(s[1])] for s in [ln.split() for ln in pf]])
    else:
        ploidyDict = dict(zip(allInds, [2] * len(allInds)))
    for ind in args.haploid or []:
        ploidyDict[ind] = 1
    sampleData = SampleData(popNames=popNames, popInds=popInds, ploidyDict=ploidyDict)
    fmt = "pairs" if args.genoFormat == "alleles" else args.genoFormat
    layout = HapLayout(sampleData, headerInds, fmt)
    asCounts = args.asCounts if args.target else True                      # freq.py:222-224
    keepNan = args.keepNanLines if args.target else True
    minData = args.minData if args.target else 0

    # Multi-GPU: sites are independent, so every rank takes the lines between two line boundaries near the equal split of a
    # plain-text input (the slice-parallel reading of freq.py:23-28), formats its own rows, and ONE gather brings them to
    # rank 0 in rank order = input order.  Inputs that cannot be cut (gzip, stdin, .pgeno): rank 0 does the job.
    world = dist.world_from_env()
    if world.size > 1 and not os.environ.get("PG_HOST_THREADS"):
        os.environ["PG_HOST_THREADS"] = str(max(1, _lib.usable_cpus() // world.size))
    eng = Engine(args.device if args.device is not None else dist.device_for(world))
    eng.set_layout(layout)
    comm = dist.make_comm(eng, world)
    sharded = world.size > 1 and hasattr(reader, "shard_lines") and reader.shard_lines(world)
    if world.size > 1 and not sharded and world.rank > 0:
        dist.gather_bytes(comm, b"")
        comm.close()                                          # every rank takes part in the same exchanges, the closing barrier included
        reader.close()
        return 0
    out = _open_out(args.outFile) if world.rank == 0 else None
    if out is not None:
        out.write("scaffold\tposition\t" + "\t".join(popNames) + "\n")
    P = len(popNames)
    CH = 1 << 20
    block_bytes = int(os.environ.get("PG_STREAM_BYTES", 1 << 30))

    piped = hasattr(eng, "upload_async")                    # (tests/cpu_engine.py's stand-in mimics the interface)
    pitch = eng.row_pitch if piped else None
    alloc = eng.pinned.empty if piped else None
    import ctypes as C
    from ._lib import check, lib
    L = lib()
    # K0 on the device (default for text in a layout pg_tokenize_text takes; PG_GPU_TOKENIZER=0 keeps the host tokenizer)
    on_device = (os.environ.get("PG_GPU_TOKENIZER", "1") != "0" and hasattr(eng, "tokenize_text")
                 and not getattr(reader, "packed", False) and device_tokenizer_takes(layout))
    # bgzipped text: blocks arrive as spans of deflated members (in page-locked memory) and are inflated on the device
    spans = (on_device and hasattr(eng, "tokenize_submit_bgzf") and hasattr(reader, "spans") and isinstance(getattr(reader, "f", None), genoio.BgzfFile)
             and os.environ.get("PG_BGZF_DEVICE", "1") != "0" and not sharded)
    if spans:
        reader.spans = True
        reader.f.alloc = eng.pinned.empty
    stats = {"device_tokenizer": int(on_device), "bgzf_blocks_inflated_on_device": 0, "host_tokenized_blocks": 0, "blocks": 0, "sites": 0, "wait_for_block_s": 0.0, "tokenize_s": 0.0,
             "counts_s": 0.0, "format_s": 0.0, "write_s": 0.0}
    stats["context_s"] = round(_time.perf_counter() - t_begin, 4)      # (argument parsing, the input's header, the device context)
    t_start = t_begin

    def lap(key, t0):
        t1 = _time.perf_counter()
        stats[key] += t1 - t0
        return t1

    def site_blocks():
        """(GenoData of an input block, run index of each of its rows, a, b) for sub-blocks [a,b) of at most CH sites.  The next
        block is tokenised (into page-locked rows at the engine's pitch) by a helper thread while this one is counted and
        written."""
        import queue
        import threading
        ready = queue.Queue(maxsize=1)

        def prepare():
            try:
                while True:
                    body = reader.read_block(block_bytes)
                    if len(body) == 0:
                        ready.put(None)
                        return
                    if on_device:                           # the text itself goes to the device (this thread only reads ahead)
                        ready.put(body)
                    else:
                        ready.put(reader.to_geno(body, layout, pitch=pitch, alloc=alloc))
                    del body
            except BaseException as exc:
                ready.put(exc)

        threading.Thread(target=prepare, daemon=True, name="prepare").start()
        while True:
            t0 = _time.perf_counter()
            data = ready.get()
            t0 = lap("wait_for_block_s", t0)
            if data is None:
                return
            if isinstance(data, BaseException):
                raise data
            stats["blocks"] += 1
            if on_device:
                # K0 on the device: the rows are written where k_site_counts reads them, only positions and runs come back
                body = data
                got = None
                if isinstance(body, genoio.BgzfSpan):
                    # (a line of the regular layout: its cells + scaffold, position, two blanks; a bound of the rows, as in Run)
                    f3 = body.first_line.split(None, 2)
                    if len(f3) == 3 and not body.first_line.startswith(b"#"):
                        bound = len(body) // (len(f3[2]) + 1 + 4) + 1
                        eng.reserve(bound)
                        if eng.tokenize_submit_bgzf(body, 0):
                            n_lines = eng.tokenize_parse(0, 0, bound)
                            got = eng.tokenize_collect(0, body, n_lines) if n_lines is not None else None
                    if got is not None:
                        stats["bgzf_blocks_inflated_on_device"] += 1
                    else:
                        body = bytes(body)                  # not the regular layout: inflated on the host, the routes below
                if got is None:
                    ptr, nbytes, _keep = _lib.text_ptr(body)
                    cnt = C.c_int64(0)
                    _lib.check(L.pg_count_lines(ptr, nbytes, C.byref(cnt)))
                    eng.reserve(int(cnt.value))
                    got = eng.tokenize_text(body, row_offset=0, n_rows=int(cnt.value)) if cnt.value else None
                else:
                    _keep = None
                if got is not None:
                    data = genoio.GenoData(None, got[1], got[2], got[3])
                else:                                       # a block the device tokenizer refuses: host tokenizer, one upload
                    data = reader.to_geno(body, layout, pitch=pitch, alloc=alloc)
                    stats["host_tokenized_blocks"] += 1
                del body, _keep
                lap("tokenize_s", t0)
            stats["sites"] += data.n_sites
            run_of_row = np.repeat(np.arange(len(data.run_starts)), np.diff(np.append(data.run_starts, data.n_sites)))
            for a in range(0, data.n_sites, CH):
                yield data, run_of_row, a, min(data.n_sites, a + CH)

    def load(rows):
        if piped:
            eng.reserve(len(rows))
            eng.upload_async(rows, 0)                       # one linear DMA out of page-locked memory
            eng.upload_wait()
        else:
            eng.load_sites(rows)

    kept = []                                               # ranks > 0: their rows, until the gather
    if out is not None:
        out.flush()
        sink = out.buffer                                   # rows are written as bytes behind the header
    text_buf = np.empty(0, dtype=np.uint8)
    cur_data = None
    for data, run_of_row, a, b in site_blocks():
        if data is not cur_data:                            # scaffold names of the block as one byte string + offsets
            cur_data = data
            enc = [nm.encode() for nm in data.run_names]
            names_blob = b"".join(enc)
            name_off = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.int64)
            name_max, run0 = max([len(e) for e in enc] + [1]), 0
        t0 = _time.perf_counter()
        lo_, hi_ = (a, b) if data.gt is None else (0, b - a)  # tokenised on the device: the block's rows are resident
        if data.gt is not None:
            load(data.gt[a:b])
        keep = None                                                           # uint8 mask of the rows that are written
        if not args.target:
            mode, values = 0, np.ascontiguousarray(eng.batch([0], [0]).siteCounts(lo_, hi_))           # int32 [n][P][4]
        else:
            # target allele, counts / rounded frequencies, --threshold and the rows' keep flags on the device (pg_site_target)
            values, keep = eng.batch([0], [0]).siteTarget(lo_, hi_, args.target, minData, asCounts, args.threshold)
            mode = 1 if asCounts else 2
            if keepNan:
                keep = None
        t0 = lap("counts_s", t0)
        # the rows as text, formatted natively on all host threads (pg_format_freq_rows)
        cap = (b - a) * (name_max + 13 + P * (45 if mode == 0 else 22)) + 64
        if len(text_buf) < cap:
            text_buf = np.empty(cap, dtype=np.uint8)
        got = C.c_int64(0)
        check(L.pg_format_freq_rows(mode, b - a, P, C.c_void_p(values.ctypes.data), np.ascontiguousarray(data.pos[a:b], dtype=np.int64),
                                    np.ascontiguousarray(run_of_row[a:b], dtype=np.int32) - run0, names_blob, name_off,
                                    C.c_void_p(keep.ctypes.data) if keep is not None else None,
                                    C.c_void_p(text_buf.ctypes.data), cap, C.byref(got), 0))
        t0 = lap("format_s", t0)
        if out is not None:
            sink.write(memoryview(text_buf)[:got.value])
        else:
            kept.append(bytes(memoryview(text_buf)[:got.value]))
        lap("write_s", t0)
    reader.close()
    if world.size > 1:
        parts = dist.gather_bytes(comm, b"".join(kept))
        if out is not None:
            for part in parts[1:]:
                sink.write(part)
    if out is not None:
        sink.flush()
        if out is not sys.stdout:
            out.close()
    if os.environ.get("PG_TIMING"):
        import json
        stats["total_s"] = _time.perf_counter() - t_start
        sys.stderr.write("PG_TIMING " + json.dumps(dict({k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()}, rank=world.rank)) + "\n")
    if world.rank == 0:
        sys.stderr.write("\nDone\n")
    if world.size > 1:
        comm.close()
    return 0
