"""Window-range sharding of the input over the ranks of a multi-GPU launch ("partition = contiguous ranges of the
global window list, aligned to scaffold boundaries when scaffolds >= GPUs, else split a scaffold's window range").

The reference hands WINDOWS to its workers, whatever scaffold they lie on (popgenWindows.py:396-403, 445-447), and a coordinate
window is a pure function of (scaffold, position): window k of a scaffold run is [1 + k*step, w + k*step], and it exists iff
k == 0 or the run holds a site beyond the end of window k-1 (genomics.py:1988-2017: the generator yields window k-1 when it meets
such a site, slides, and yields what it has when the run ends).  So the data lines can be cut INSIDE a run:

  coordinate windows   rank r starts with window k0 >= 1 of run S: its first line is the run's first line with position >=
                       1 + k0*step; its predecessor ends in front of the first line with position > w + (k0-1)*step (every site
                       of window k0-1 is on its side; with -s < -w both read the overlap).  The cut is only made when both lines
                       exist in the run -- then every window < k0 exists (so the predecessor emits all of them, empty ones
                       included, without seeing the run's end) and window k0 exists (so the rank's first row is a real window:
                       nothing of the generator's state crosses the cut but the window counter).  Otherwise the cut moves to the
                       end of the run.
  sites windows        (no --maxDist) window j of a run is rows [j*(w-overlap), j*(w-overlap) + w): a pure function of the row
                       index, genomics.py:2032-2108.  One gather of per-share line counts gives every share boundary its row
                       index in its run; rank r starts at row j0*(w-overlap), its predecessor ends behind row
                       (j0-1)*(w-overlap) + w - 1, cut only when that row has a successor in the run (window j0 exists).
                       Both sides then see what the generator would have seen: the predecessor a run that ends with a full
                       window, the rank a run that begins at its first row.

A cut at the first line of a run needs both neighbours wanted (--include / --exclude): the generators re-emit their last window
after a skipped scaffold (genomics.py:2016-2023), and that state would cross the cut.

Every rank finds its own cut near r/N of the data (text: a walk over at most one window span of memory-mapped lines,
pg_text_seek_pos / pg_text_skip_rows; BGZF: the same over inflated members; `.pgeno`: block headers and position arrays), ONE
exchange shares the cuts, the rank restricts its reader to [its start, its successor's end_prev).
"""
import json

import numpy as np

from . import _lib, dist

I64_MAX = (1 << 63) - 1


class Cut(dict):
    """start: where the rank's first line is; end_prev: where the preceding rank's share ends (exclusive) -- byte offsets (text),
    [member file offset, offset in the member] (BGZF) or global rows (.pgeno); k0 > 0: the share starts with window k0 of a run of
    scaffold `name` (coordinate windows), k0 == 0: as if at the first line of a run."""

    def __init__(self, start, end_prev=None, k0=0, name=None):
        super().__init__(start=start, end_prev=start if end_prev is None else end_prev, k0=int(k0), name=name)


def _ceil_div(a, b):
    return -((-a) // b)


# ---- cursors over raw text ------------------------------------------------------------------------------------------------------
class _Text:
    """Forward walks over data lines.  Offsets are relative to `self.data`; token(off) is what travels between the ranks."""

    def line_at(self, off):
        """(offset, scaffold bytes, position) of the first data line that starts at or behind `off` (a line start); None at the end"""
        while True:
            self._need(off, 1)
            if off >= self._len():
                return None
            nl = self._find_nl(off)
            if nl < 0:
                if self._grow():
                    continue
                nl = self._len()
            head = bytes(self.data[off:min(nl, off + 4096)])
            f = head.split(None, 2)
            if len(f) >= 2 and not head.startswith(b"#"):
                return off, f[0], int(f[1])
            off = nl + 1

    def seek(self, off, scaf, pos_min):
        """(offset, state, position) of the first data line from `off` on that is not of scaffold `scaf` (state 0), or has a position
        >= pos_min (state 1); (end, -1, 0) when the text ends first"""
        import ctypes as C
        L = _lib.lib()
        o, st, ps, rows = C.c_int64(0), C.c_int32(0), C.c_int64(0), C.c_int64(0)
        while True:
            view = memoryview(self.data)[off:self._len()]
            ptr, nbytes, keep = _lib.text_ptr(view)
            whole = 1 if self._complete() else 0
            _lib.check(L.pg_text_seek_pos(ptr, nbytes, whole, scaf, len(scaf), pos_min, C.byref(o), C.byref(st), C.byref(ps), C.byref(rows)))
            del keep, ptr
            view.release()
            off += o.value
            self.walked += o.value
            if st.value >= 0 or whole:
                return off, st.value, ps.value
            self._grow()                                          # (nothing more: the text is complete now, one more call ends it)

    def skip_rows(self, off, n):
        """offset of the data line that follows n data lines from `off` on (the end of the text when there are fewer)"""
        import ctypes as C
        L = _lib.lib()
        o, rows = C.c_int64(0), C.c_int64(0)
        left = n
        while True:
            view = memoryview(self.data)[off:self._len()]
            ptr, nbytes, keep = _lib.text_ptr(view)
            _lib.check(L.pg_text_skip_rows(ptr, nbytes, left, C.byref(o), C.byref(rows)))
            del keep, ptr
            view.release()
            off += o.value
            self.walked += o.value
            left -= rows.value
            if off < self._len() or not self._grow():
                return off


class MapText(_Text):
    """a memory-mapped plain-text file: everything is there, offsets are file offsets"""

    def __init__(self, mm, size):
        self.data, self.size, self.walked = mm, size, 0

    def _len(self):
        return self.size

    def _need(self, off, n):
        pass

    def _grow(self):
        return False

    def _complete(self):
        return True

    def _find_nl(self, off):
        return self.data.find(b"\n", off, self.size)

    def start_at(self, guess, data_start):
        """offset of the first line that starts at or behind byte `guess`"""
        if guess <= data_start:
            return data_start
        nl = self.data.find(b"\n", guess - 1, self.size)
        return self.size if nl < 0 else nl + 1

    def token(self, off):
        return int(off)

    def end_token(self):
        return int(self.size)


class BgzfText(_Text):
    """the inflated text of a BGZF file from the first line that begins in or behind the member at file offset >= guess; grows on
    demand; tokens are (member file offset, offset inside the member) pairs"""

    def __init__(self, path, guess):
        from .genoio import BgzfFile
        self.bz = BgzfFile(path)
        self.bz.keep_track = True
        self.bz.seek_member(guess)
        self.skipped = len(self.bz.readline()) if guess > 0 else 0     # the line that straddles into the member belongs to the left
        self.data = bytearray()
        self.walked = 0
        self.chunk = 1 << 16
        self._grow()

    def _len(self):
        return len(self.data)

    def _need(self, off, n):
        while off + n > len(self.data) and self._grow():
            pass

    def _grow(self):
        more = self.bz.read(self.chunk)
        self.chunk = min(2 * self.chunk, 32 << 20)
        if not more:
            return False
        self.data += more
        return True

    def _complete(self):
        return self.bz.eof and not self.bz.buf

    def _find_nl(self, off):
        return self.data.find(b"\n", off)

    def token(self, off):
        if off >= len(self.data) and self._complete():
            return self.end_token()
        c, u = self.bz.virtual_of(self.skipped + off)
        return [int(c), int(u)]

    def end_token(self):
        return [int(self.bz.size), 0]

    def close(self):
        self.bz.close()


def coord_cut(text, off, w, step, wanted):
    """The cut for coordinate windows at or behind line start `off` of cursor `text` (see the module docstring)."""
    while True:
        ln = text.line_at(off)
        if ln is None:
            return Cut(text.end_token())
        off, scaf, p = ln
        name = scaf.decode("utf-8", "replace")
        if not wanted(name):
            off, st, _ = text.seek(off, scaf, I64_MAX)              # to the end of the skipped run; the next run is then entered
            if st != 0:                                             # by its window 1 at the earliest (its predecessor is not wanted)
                return Cut(text.end_token())
            continue
        k0 = max(1, _ceil_div(p, step))                          # the first window that starts behind p (rows in front of the probe may share its position)
        p0, e1 = 1 + k0 * step, w + (k0 - 1) * step + 1             # the rank's first position; the first position beyond window k0 - 1
        if e1 <= p0:                                                # (one walk: the farther line is looked for from the nearer one on)
            b, sb, _ = text.seek(off, scaf, e1)
            a, sa, _ = text.seek(b, scaf, p0) if sb == 1 else (b, sb, 0)
        else:
            a, sa, _ = text.seek(off, scaf, p0)
            b, sb, _ = text.seek(a, scaf, e1) if sa == 1 else (a, sa, 0)
        if sa == 1 and sb == 1:
            return Cut(text.token(a), text.token(b), k0, name)
        x, st = (a, sa) if sa != 1 else (b, sb)                     # the run ends first: cut where the next run begins
        if st != 0:
            return Cut(text.end_token())
        nxt = text.line_at(x)
        if nxt is None:
            return Cut(text.end_token())
        if wanted(nxt[1].decode("utf-8", "replace")):
            return Cut(text.token(nxt[0]))
        off = nxt[0]


def resolve(cuts, rank, end_token):
    """(start, end, start state, stop state) of rank `rank` from the cuts of all ranks (rank 0's is the start of the data):
    start / end tokens of its share, (name, k0) it starts with or None, (name, k0) its last run is cut at or None"""
    cuts = [dict(c) for c in cuts]
    for r in range(1, len(cuts)):                                    # (cuts come out in order; equal cuts = an empty share)
        if _lt(cuts[r]["start"], cuts[r - 1]["start"]):
            cuts[r] = dict(cuts[r - 1])
    mine = cuts[rank]
    nxt = cuts[rank + 1] if rank + 1 < len(cuts) else None
    start = mine["start"]
    end = nxt["end_prev"] if nxt is not None else end_token
    st0 = (mine["name"], mine["k0"]) if mine["k0"] > 0 else None
    st1 = (nxt["name"], nxt["k0"]) if nxt is not None and nxt["k0"] > 0 else None
    if nxt is not None and not _lt(start, nxt["start"]):             # nothing of its own: the overlap rows are the neighbours'
        end = start
    if _lt(end, start):
        end = start
    return start, end, st0, st1


def _lt(a, b):
    return tuple(a) < tuple(b) if isinstance(a, (list, tuple)) else a < b


def exchange(comm, cut):
    parts = dist.gather_bytes(comm, json.dumps(cut).encode())
    return [json.loads(p.decode()) for p in parts]


# ---- sites windows on plain text -------------------------------------------------------------------------------------------------
def share_bounds(mm, data_start, size, n):
    """line-aligned equal split of the data bytes into n shares: n + 1 offsets"""
    stride = max((size - data_start) // n, 1)
    out = [data_start]
    for r in range(1, n):
        guess = min(data_start + stride * r, size)
        nl = mm.find(b"\n", max(guess - 1, data_start), size)
        out.append(max(size if nl < 0 else nl + 1, out[-1]))
    out.append(size)
    return out


def share_runs(mm, a, b):
    """[[offset, scaffold, data lines]] of the pieces of scaffold runs among the lines of [a, b) (pg_text_runs + pg_count_lines)"""
    import ctypes as C
    if b <= a:
        return []
    L = _lib.lib()
    view = memoryview(mm)[a:b]
    ptr, nbytes, keep = _lib.text_ptr(view)
    cap = 1024
    while True:
        starts = np.zeros(cap, dtype=np.int64)
        n = C.c_int64(0)
        _lib.check(L.pg_text_runs(ptr, nbytes, starts, cap, C.byref(n)))
        if n.value <= cap:
            break
        cap = int(n.value)
    starts = [int(x) for x in starts[:n.value]]
    out = []
    for i, st in enumerate(starts):
        en = starts[i + 1] if i + 1 < len(starts) else nbytes
        cnt = C.c_int64(0)
        _lib.check(L.pg_count_lines(C.c_void_p(ptr.value + st), en - st, C.byref(cnt)))
        e = st
        while e < nbytes and view[e] not in (9, 32, 10, 13, 11, 12):
            e += 1
        out.append([a + st, bytes(view[st:e]).decode("utf-8", "replace"), int(cnt.value)])
    del keep, ptr
    view.release()
    return out


def merge_share_runs(lists):
    """runs of the whole file from the per-share lists: [(offset, scaffold, rows, [rows of the run in share 0, 1, ...])]"""
    runs = []
    for s, lst in enumerate(lists):
        for k, (off, name, cnt) in enumerate(lst):
            if k == 0 and runs and runs[-1][1] == name:
                runs[-1][2] += cnt
                runs[-1][3][s] = runs[-1][3].get(s, 0) + cnt
            else:
                runs.append([int(off), name, int(cnt), {s: int(cnt)}])
    return runs


def sites_cut(text, runs, bound, share, w, overlap, wanted):
    """The cut for sites windows near share boundary `bound` (the start of share number `share`): see the module docstring.  runs:
    merge_share_runs() of the whole file."""
    stride = w - overlap
    offs = [r[0] for r in runs]
    R = int(np.searchsorted(offs, bound, side="right")) - 1
    if R < 0:
        return Cut(text.token(bound))
    m = sum(c for s, c in runs[R][3].items() if s < share)          # rows of run R in front of the boundary
    base = bound
    while R < len(runs):
        off, name, n, _ = runs[R]
        if wanted(name):
            j0 = _ceil_div(m, stride)
            if j0 == 0 and (R == 0 or wanted(runs[R - 1][1])):
                return Cut(text.token(off))
            j0 = max(j0, 1)
            if n > (j0 - 1) * stride + w:
                a = text.skip_rows(base, j0 * stride - m)
                b = text.skip_rows(a, (j0 - 1) * stride + w - j0 * stride)
                return Cut(text.token(a), text.token(b))
        R += 1
        if R < len(runs):
            base, m = runs[R][0], 0
    return Cut(text.end_token())


# ---- `.pgeno`: rows instead of bytes ---------------------------------------------------------------------------------------------
def packed_runs(blocks):
    """(first global row, scaffold) of every run of a `.pgeno` file, from PackedReader._index()"""
    row, name = [], []
    for _, g, _, starts, names in blocks:
        for s_, n_ in zip(starts, names):
            if name and name[-1] == n_ and s_ == 0:
                continue
            row.append(g + s_)
            name.append(n_)
    return row, name


def packed_coord_cut(reader, blocks, total, guess, w, step, wanted):
    """coord_cut() on rows: positions come from the blocks' position arrays (read for the blocks the walk touches only)"""
    run_row, run_name = packed_runs(blocks)
    g = guess
    while g < total:
        R = int(np.searchsorted(run_row, g, side="right")) - 1
        end = run_row[R + 1] if R + 1 < len(run_row) else total
        name = run_name[R]
        if not wanted(name):
            g = end
            continue
        p = int(reader.positions(blocks, g, g + 1)[0])
        k0 = max(1, _ceil_div(p, step))                          # the first window that starts behind p (rows in front of the probe may share its position)
        a = _first_row_with(reader, blocks, g, end, 1 + k0 * step)
        b = _first_row_with(reader, blocks, g, end, w + (k0 - 1) * step + 1)
        if a is not None and b is not None:
            return Cut(a, b, k0, name)
        if end >= total:
            break
        if wanted(run_name[R + 1]):
            return Cut(end)
        g = end
    return Cut(total)


def _first_row_with(reader, blocks, a, b, pos_min, piece=1 << 16):
    """first row of [a, b) (rows of one run: sorted positions) with a position >= pos_min, None when there is none"""
    while a < b:
        e = min(a + piece, b)
        p = reader.positions(blocks, a, e)
        k = int(np.searchsorted(p, pos_min, side="left"))
        if k < len(p):
            return a + k
        a = e
        piece *= 4
    return None


def packed_sites_cut(blocks, total, guess, w, overlap, wanted):
    run_row, run_name = packed_runs(blocks)
    stride = w - overlap
    R = int(np.searchsorted(run_row, guess, side="right")) - 1
    m = guess - run_row[R]
    while R < len(run_row):
        off = run_row[R]
        n = (run_row[R + 1] if R + 1 < len(run_row) else total) - off
        if wanted(run_name[R]):
            j0 = _ceil_div(m, stride)
            if j0 == 0 and (R == 0 or wanted(run_name[R - 1])):
                return Cut(off)
            j0 = max(j0, 1)
            if n > (j0 - 1) * stride + w:
                return Cut(off + j0 * stride, off + (j0 - 1) * stride + w)
        R += 1
        m = 0
    return Cut(total)


# ---- the plan of one rank --------------------------------------------------------------------------------------------------------
class Plan:
    """what Run does with it: the reader is already restricted; start / stop go to windows.CoordWindowStream"""

    def __init__(self, start, stop, share, scanned=0):
        self.start, self.stop, self.share, self.scanned = start, stop, share, scanned


def shard_reader(reader, world, comm, wparams, wanted):
    """Restrict `reader` (genoio.BlockReader / PackedReader, header consumed) to this rank's window range and return the Plan, or
    None (reader untouched) when the input or the window type does not allow it: stdin, plain gzip, sites windows with --maxDist
    (a window's extent then depends on the positions of all rows before it) or on BGZF (no line counts without inflating it all)."""
    from .genoio import BgzfFile
    wt = wparams["windType"]
    if wt == "coordinate":
        w, step = int(wparams["windSize"]), int(wparams["stepSize"])
    elif wt == "sites" and np.isinf(wparams["maxDist"]):
        w, overlap = int(wparams["windSize"]), int(wparams["overlap"])
        if overlap >= w:
            return None
    else:
        return None
    N, r = world.size, world.rank
    if getattr(reader, "packed", False):
        blocks, total = reader._index()
        if total == 0:
            return None
        guess = total * r // N
        if r == 0:
            cut = Cut(0)
        elif wt == "coordinate":
            cut = packed_coord_cut(reader, blocks, total, guess, w, step, wanted)
        else:
            cut = packed_sites_cut(blocks, total, guess, w, overlap, wanted)
        start, end, st0, st1 = resolve(exchange(comm, cut), r, total)
        reader.restrict_rows(blocks, start, end)
        return Plan(st0, st1, (end - start) / total)
    if isinstance(reader.f, BgzfFile):
        if wt != "coordinate":
            return None
        size = reader.f.size
        text = None
        if r == 0:
            cut = Cut([0, 0])
        else:
            text = BgzfText(reader.path, size * r // N)
            cut = coord_cut(text, 0, w, step, wanted)
        start, end, st0, st1 = resolve(exchange(comm, cut), r, [size, 0])
        if text is not None:
            reader.bytes_read += text.walked
            text.close()
        reader.restrict_virtual(start, end, first=(r == 0))
        return Plan(st0, st1, None, text.walked if text is not None else 0)
    if not reader.seekable_text() or reader.mm is None:
        return None
    size, data_start = len(reader.mm), reader.tell()
    text = MapText(reader.mm, size)
    if wt == "coordinate":
        cut = Cut(data_start) if r == 0 else coord_cut(text, text.start_at(data_start + (size - data_start) * r // N, data_start), w, step, wanted)
    else:
        bounds = share_bounds(reader.mm, data_start, size, N)
        mine = share_runs(reader.mm, bounds[r], bounds[r + 1])
        text.walked += bounds[r + 1] - bounds[r]
        runs = merge_share_runs(exchange(comm, mine))
        cut = Cut(data_start) if r == 0 else sites_cut(text, runs, bounds[r], r, w, overlap, wanted)
    start, end, st0, st1 = resolve(exchange(comm, cut), r, size)
    reader.restrict(start, end)
    return Plan(st0, st1, (end - start) / max(size - data_start, 1), text.walked)
