"""Window index computation: which sites land in which window, and which (possibly empty) windows exist.

The reference does this one text line at a time inside Python generators (genomics.py:1971-2223).  Here the
positions of the whole input are already an int64 array (K0), so windows are computed as index ranges
[lo,hi) into it with searchsorted, reproducing the generators' observable behaviour:

  slidingCoordWindows   genomics.py:1971-2027  -> coord_windows
  slidingSitesWindows   genomics.py:2032-2108  -> sites_windows
  predefinedCoordWindows genomics.py:2112-2171 -> predefined_windows

including their quirks: the first window of every scaffold run is [1,windSize]; empty windows are emitted;
windows stop at the first one whose end reaches the run's last site; the window that precedes a skipped
(--exclude / not --include) scaffold is emitted a second time when more wanted data follows.  Positions must be
non-decreasing inside a scaffold run (the reference silently drops out-of-order sites; we raise).
Not reproduced: the reference's infinite loop at EOF under --include.
"""
import numpy as np


class WindowTable:
    """Columns of the emitted windows, in emission order."""

    def __init__(self):
        self.scaffold, self.start, self.end, self.lo, self.hi, self.ID = [], [], [], [], [], []

    def add(self, scaffold, start, end, lo, hi, ID):
        self.scaffold.append(scaffold)
        self.start.append(start)
        self.end.append(end)
        self.lo.append(lo)
        self.hi.append(hi)
        self.ID.append(ID)

    def finish(self, positions):
        self.lo = np.asarray(self.lo, dtype=np.int64)
        self.hi = np.asarray(self.hi, dtype=np.int64)
        self.n = len(self.lo)
        self.sites = self.hi - self.lo
        self._positions, self._mid = positions, None
        return self

    @property
    def mid(self):
        """GenoWindow.midPos (genomics.py:1795-1797): int(round(sum/len)), nan when empty.  Computed when first asked for -- by the
        thread that formats the rows, not by the one that finds the windows of the next block."""
        if self._mid is None:
            nz = self.hi > self.lo
            a = int(self.lo[nz].min()) if np.any(nz) else 0
            b = int(self.hi[nz].max()) if np.any(nz) else 0
            csum = np.concatenate([[0], np.cumsum(np.asarray(self._positions[a:b], dtype=np.int64))])
            tot = csum[np.where(nz, self.hi - a, 0)] - csum[np.where(nz, self.lo - a, 0)]
            with np.errstate(divide="ignore", invalid="ignore"):
                mean = tot / self.sites
            self._mid = [int(np.rint(m)) if s > 0 else float("nan") for m, s in zip(mean, self.sites)]
            self._positions = None
        return self._mid

    @mid.setter
    def mid(self, value):
        self._mid = value


def _wanted(scaf, include, exclude):                 # genomics.py:2016
    if not include and not exclude:
        return True
    if include:
        return scaf in include
    return scaf not in exclude


def _check_sorted(positions, a, b, scaf):
    p = positions[a:b]
    if len(p) > 1 and np.any(p[1:] < p[:-1]):
        raise ValueError("positions of scaffold %s are not sorted; sliding windows need sorted input" % scaf)


def _runs(run_starts, n_sites):
    ends = list(run_starts[1:]) + [n_sites]
    return list(zip([int(x) for x in run_starts], [int(x) for x in ends]))


def coord_windows(run_starts, run_names, positions, windSize, stepSize, include=None, exclude=None):
    """All windows slidingCoordWindows would yield.  run_starts[r] = first row of scaffold run r."""
    positions = np.asarray(positions)
    T = WindowTable()
    done = 0
    pending = None                       # last emitted window, re-emitted if a skipped run is followed by a wanted one
    runs = _runs(run_starts, len(positions))
    skipped_since = False
    for r, (a, b) in enumerate(runs):
        scaf = run_names[r]
        if not _wanted(scaf, include, exclude):
            skipped_since = True
            continue
        if skipped_since and pending is not None:
            done += 1
            T.add(*pending)              # same ID as its first emission (genomics.py:2001-2005 keeps window.ID)
        skipped_since = False
        _check_sorted(positions, a, b, scaf)
        p = positions[a:b].astype(np.int64)
        last = int(p[-1])
        k_last = 0 if last <= windSize else -((windSize - last) // stepSize)      # ceil((last-w)/step)
        k = np.arange(k_last + 1, dtype=np.int64)
        starts = 1 + k * stepSize
        ends = windSize + k * stepSize
        lo = a + np.searchsorted(p, starts, side="left")
        hi = a + np.searchsorted(p, ends, side="right")
        for i in range(len(k)):
            T.add(scaf, int(starts[i]), int(ends[i]), int(lo[i]), int(hi[i]), done + 1)
            done += 1
        pending = (scaf, int(starts[-1]), int(ends[-1]), int(lo[-1]), int(hi[-1]), done)
    return T.finish(positions)


def sites_windows(run_starts, run_names, positions, windSites, overlap, maxDist=np.inf, minSites=None,
                  include=None, exclude=None):
    """All windows slidingSitesWindows would yield; start/end are firstPos()/lastPos() (popgenWindows.py:39)."""
    positions = np.asarray(positions)
    if not minSites:
        minSites = windSites
    if overlap >= windSites:
        raise ValueError("overlap must be smaller than the window size (the reference would loop forever)")
    T = WindowTable()
    done = 0
    pending = None
    skipped_since = False
    finite = not np.isinf(maxDist)
    for r, (a, b) in enumerate(_runs(run_starts, len(positions))):
        scaf = run_names[r]
        if not _wanted(scaf, include, exclude):
            skipped_since = True
            continue
        if skipped_since and pending is not None:
            done += 1
            T.add(*pending)               # the generator re-yields its unchanged last window after a skipped scaffold
        skipped_since = False
        pending = None
        _check_sorted(positions, a, b, scaf)
        p = positions[a:b].astype(np.int64)
        n = b - a
        lo = hi = 0                       # current window = rows [lo,hi) of this run
        wid = done + 1
        while True:
            if hi < n:                    # grow (genomics.py:2052)
                cap = min(n, lo + windSites)
                if finite:
                    cap = min(cap, int(np.searchsorted(p, p[lo] + maxDist, side="right")))
                hi = max(hi, cap)
            emitted = (hi - lo) >= minSites
            if emitted:
                done += 1
                pending = (scaf, int(p[lo]), int(p[hi - 1]), a + lo, a + hi, wid)
                T.add(*pending)
            if hi >= n:                   # next line is another scaffold or EOF: the window object is left as it is
                if not emitted:
                    pending = None
                break
            if emitted:                   # GenoWindow.trim(leave=overlap) == rows[len-overlap:], genomics.py:1779-1788
                remove = (hi - lo) - overlap
                if remove <= 0:
                    raise ValueError("sites window of %d rows cannot advance with overlap %d under maxDist "
                                     "(the reference yields this window forever)" % (hi - lo, overlap))
                lo = min(lo + remove, hi)
                wid = done + 1
            elif hi > lo:
                lo += 1                   # trim(remove=1)
    return T.finish(positions)


def predefined_windows(run_starts, run_names, positions, windCoords):
    """All windows predefinedCoordWindows would yield.  windCoords: [(scaffold, start, end[, ID])].
    Faithful quirk: on consecutive windows of one scaffold the generator only trims the LEFT of the previous
    window's rows (GenoWindow.slide(newLimits)), so rows beyond the new end that were already read stay in."""
    positions = np.asarray(positions)
    all_scafs = [w[0] for w in windCoords]
    scafs = sorted(set(all_scafs), key=all_scafs.index)
    runs = _runs(run_starts, len(positions))
    n = len(positions)
    run_of_row = np.zeros(n, dtype=np.int64)
    for r, (a, b) in enumerate(runs):
        run_of_row[a:b] = r
    T = WindowTable()
    cur = 0                               # the "line in hand"
    prev = None                           # (scaffold, lo, hi) of the previous window's rows
    for w in windCoords:
        scaf, start, end = w[0], int(w[1]), int(w[2])
        ID = w[3] if len(w) > 3 else "NA"
        widx = scafs.index(scaf)
        kept = None
        if prev is not None and prev[0] == scaf and prev[2] > prev[1]:
            k_lo = prev[1] + int(np.searchsorted(positions[prev[1]:prev[2]], start, side="left"))
            if prev[2] > k_lo:
                kept = (k_lo, prev[2])
        while cur < n:                    # genomics.py:2142-2145
            name = run_names[run_of_row[cur]]
            if name not in scafs or scafs.index(name) < widx:
                cur = runs[run_of_row[cur]][1]
            else:
                break
        new = None
        if cur < n and run_names[run_of_row[cur]] == scaf:
            a, b = runs[run_of_row[cur]]
            _check_sorted(positions, a, b, scaf)
            p = positions[cur:b]
            x = cur + int(np.searchsorted(p, start, side="left"))
            y = cur + int(np.searchsorted(p, end, side="right"))
            if y > x:
                new = (x, y)
                cur = y
            else:
                cur = x
        if kept and new:
            rows = (kept[0], new[1])
        elif kept:
            rows = kept
        elif new:
            rows = new
        else:
            rows = (cur, cur) if cur <= n else (n, n)
        T.add(scaf, start, end, rows[0], rows[1], ID)
        prev = (scaf, rows[0], rows[1])
        if cur >= n:
            break
    return T.finish(positions)


class CoordWindowStream:
    """slidingCoordWindows over an input that arrives in consecutive pieces (bounded host memory).

    feed() gets the current buffer = rows carried over from the previous call + the new rows, as scaffold runs and
    positions, and returns the windows that are certain by now together with the first buffer row the caller has to keep
    for the next call.  A window [a,b] of the buffer's last run is certain once a site beyond b has been seen (that is when
    the reference's generator yields it, genomics.py:2001-2005); the last run's remaining windows follow when the run ends
    (next call starts with another scaffold, or final=True).  Concatenating the tables of all calls gives exactly
    coord_windows() of the whole input, IDs included.  The reference's re-emission of the last window after a skipped
    scaffold appears as a row with T.dup[i] = True: it repeats the previously emitted row verbatim (its sites may no longer
    be in the buffer, so lo = hi = 0 there)."""

    def __init__(self, windSize, stepSize, include=None, exclude=None, start=None, stop=None):
        """start / stop: this stream sees a window range of the input (shardplan).  start = (scaffold, k0): the first rows continue a
        run of that scaffold whose windows < k0 are another rank's (all of them exist); stop = (scaffold, k1): the run the input ends
        in -- of that scaffold, or not begun yet when no row of it is on this side -- goes on elsewhere, its windows < k1 all exist
        and are this stream's, the others are not."""
        self.w, self.step, self.include, self.exclude = int(windSize), int(stepSize), include, exclude
        self.done = 0
        self.have_pending = False        # some window has been emitted (it can be re-emitted after a skipped scaffold)
        self.skipped_since = False
        self.cont_name = None            # scaffold run that was still open at the end of the previous buffer
        self.cont_k0 = 0                 # its next window index
        self.cont_last = 0               # its last position seen
        self.stop = (stop[0], int(stop[1])) if stop else None
        if start:
            self.cont_name, self.cont_k0, self.have_pending = start[0], int(start[1]), True

    def feed(self, run_starts, run_names, positions, final):
        positions = np.asarray(positions)
        n = len(positions)
        T = WindowTable()
        T.dup = []
        runs = _runs(run_starts, n) if n else []
        keep_from = n
        cont_name, cont_k0, cont_last = None, 0, 0
        stop = self.stop if final else None          # the cut run is the one the input ends in
        if n == 0 and not final:                     # a piece without data rows says nothing about the open run
            T.finish(positions)
            T.dup = np.zeros(0, dtype=bool)
            return T, 0
        if self.cont_name is not None and (not runs or run_names[0] != self.cont_name):
            # the open run ended exactly at the buffer boundary and none of its rows had to be carried (stepSize > windSize):
            # its remaining windows are empty but still emitted
            kl = 0 if self.cont_last <= self.w else -((self.w - self.cont_last) // self.step)
            if stop is not None and not runs and stop[0] == self.cont_name:
                kl, stop = stop[1] - 1, None
            for k in range(self.cont_k0, kl + 1):
                T.add(self.cont_name, 1 + k * self.step, self.w + k * self.step, 0, 0, self.done + 1)
                T.dup.append(False)
                self.done += 1
                self.have_pending = True
            self.cont_name = None
        for r, (a, b) in enumerate(runs):
            scaf = run_names[r]
            is_cont = r == 0 and scaf == self.cont_name
            is_open = r == len(runs) - 1 and not final
            if not _wanted(scaf, self.include, self.exclude):
                self.skipped_since = True
                continue
            if self.skipped_since and self.have_pending and not is_cont:
                self.done += 1
                T.add("", 0, 0, 0, 0, 0)
                T.dup.append(True)
            self.skipped_since = False
            _check_sorted(positions, a, b, scaf)
            p = positions[a:b].astype(np.int64)
            last = int(p[-1])
            k0 = self.cont_k0 if is_cont else 0
            kl = 0 if last <= self.w else -((self.w - last) // self.step)          # ceil((last-w)/step)
            k_end = kl if is_open else kl + 1                                       # windows k0 .. k_end-1 are certain
            if stop is not None and r == len(runs) - 1 and scaf == stop[0]:
                k_end, stop = stop[1], None                                         # (the run goes on on another rank)
            if k_end > k0:
                k = np.arange(k0, k_end, dtype=np.int64)
                starts = 1 + k * self.step
                ends = self.w + k * self.step
                lo = a + np.searchsorted(p, starts, side="left")
                hi = a + np.searchsorted(p, ends, side="right")
                for i in range(len(k)):
                    T.add(scaf, int(starts[i]), int(ends[i]), int(lo[i]), int(hi[i]), self.done + 1)
                    T.dup.append(False)
                    self.done += 1
                self.have_pending = True
            if is_open:
                cont_name, cont_k0, cont_last = scaf, max(k0, kl), last
                keep_from = a + int(np.searchsorted(p, 1 + cont_k0 * self.step, side="left"))
        if stop is not None:
            # no row of the cut run is on this side (its rows in front of the cut lie in no window): its windows < k1 are empty
            if self.skipped_since and self.have_pending:
                self.done += 1
                T.add("", 0, 0, 0, 0, 0)
                T.dup.append(True)
            self.skipped_since = False
            for k in range(stop[1]):
                T.add(stop[0], 1 + k * self.step, self.w + k * self.step, 0, 0, self.done + 1)
                T.dup.append(False)
                self.done += 1
                self.have_pending = True
        self.cont_name, self.cont_k0, self.cont_last = cont_name, cont_k0, cont_last
        T.finish(positions)
        T.dup = np.asarray(T.dup, dtype=bool)
        return T, keep_from


class SitesWindowStream:
    """slidingSitesWindows over an input that arrives in consecutive pieces; same contract as CoordWindowStream.feed().
    A window of the buffer's last (still open) run is certain when it is full (windSites rows) or closed by maxDist, i.e. a
    row beyond it has been seen; otherwise it waits for more rows: its rows are carried, together with how far the window
    state had grown (the generator's `hi` never shrinks) and the ID it was created with."""

    def __init__(self, windSites, overlap, maxDist=np.inf, minSites=None, include=None, exclude=None):
        self.w, self.overlap, self.maxDist = int(windSites), int(overlap), maxDist
        self.minSites = int(minSites) if minSites else int(windSites)
        if self.overlap >= self.w:
            raise ValueError("overlap must be smaller than the window size (the reference would loop forever)")
        self.include, self.exclude = include, exclude
        self.done = 0
        self.have_pending = False
        self.skipped_since = False
        self.cont_name = None
        self.cont_hi = 0                  # rows of the open window state at the start of the carried rows
        self.cont_wid = 0
        self.cont_after_emit = False      # the open run stopped right after a full window that ended at the buffer's end

    def feed(self, run_starts, run_names, positions, final):
        positions = np.asarray(positions)
        n_all = len(positions)
        T = WindowTable()
        T.dup = []
        runs = _runs(run_starts, n_all) if n_all else []
        keep_from = n_all
        finite = not np.isinf(self.maxDist)
        cont_name, cont_hi, cont_wid, cont_after_emit = None, 0, 0, False
        if n_all == 0 and not final:                 # a piece without data rows says nothing about the open run
            T.finish(positions)
            T.dup = np.zeros(0, dtype=bool)
            return T, 0
        if self.cont_name is not None and (not runs or run_names[0] != self.cont_name):
            # the open run had no rows left to carry and nothing of it follows: it ended at the buffer boundary
            self.have_pending = self.cont_after_emit
            self.cont_name = None
        for r, (a, b) in enumerate(runs):
            scaf = run_names[r]
            is_cont = r == 0 and scaf == self.cont_name
            is_open = r == len(runs) - 1 and not final
            if not _wanted(scaf, self.include, self.exclude):
                self.skipped_since = True
                continue
            if self.skipped_since and self.have_pending and not is_cont:
                self.done += 1
                T.add("", 0, 0, 0, 0, 0)
                T.dup.append(True)
            self.skipped_since = False
            if not is_cont:
                self.have_pending = False
            _check_sorted(positions, a, b, scaf)
            p = positions[a:b].astype(np.int64)
            n = b - a
            lo = 0
            hi = self.cont_hi if is_cont else 0
            wid = self.cont_wid if is_cont else self.done + 1
            stopped = False
            if is_cont and self.cont_after_emit and n == hi:
                # no new row of this run has arrived since its last (full) window was emitted: for the generator that
                # emission was the run's last event (it breaks before trimming, genomics.py:2052-2056)
                if is_open:
                    cont_name, cont_hi, cont_wid, cont_after_emit = scaf, hi, wid, True
                    keep_from = a
                else:
                    self.have_pending = True
                continue
            while True:
                if hi < n:                    # grow (genomics.py:2052)
                    cap = min(n, lo + self.w)
                    if finite:
                        cap = min(cap, int(np.searchsorted(p, p[lo] + self.maxDist, side="right")))
                    hi = max(hi, cap)
                if is_open and hi >= n and (hi - lo) < self.w:
                    stopped = True            # not full and no row beyond it yet: wait for more rows
                    break
                emitted = (hi - lo) >= self.minSites
                if emitted:
                    self.done += 1
                    T.add(scaf, int(p[lo]), int(p[hi - 1]), a + lo, a + hi, wid)
                    T.dup.append(False)
                if hi >= n and not is_open:   # next line is another scaffold or EOF: the window object is left as it is
                    self.have_pending = emitted
                    break
                if emitted:                   # GenoWindow.trim(leave=overlap)
                    self.have_pending = True
                    remove = (hi - lo) - self.overlap
                    if remove <= 0:
                        raise ValueError("sites window of %d rows cannot advance with overlap %d under maxDist "
                                         "(the reference yields this window forever)" % (hi - lo, self.overlap))
                    lo = min(lo + remove, hi)
                    wid = self.done + 1
                elif hi > lo:
                    lo += 1                   # trim(remove=1)
                if is_open and hi >= n:       # a full window ended exactly at the buffer's end: its successor needs more rows
                    stopped = True
                    cont_after_emit = emitted
                    break
            if is_open:
                assert stopped
                cont_name, cont_hi, cont_wid = scaf, hi - lo, wid
                keep_from = a + lo
        self.cont_name, self.cont_hi, self.cont_wid, self.cont_after_emit = cont_name, cont_hi, cont_wid, cont_after_emit
        T.finish(positions)
        T.dup = np.asarray(T.dup, dtype=bool)
        return T, keep_from


class PredefinedWindowStream:
    """predefinedCoordWindows over an input that arrives in consecutive pieces (bounded host memory); same feed() contract as
    CoordWindowStream.  The generator walks the windows file and the genotype file in step ("line in hand"), so a window is
    certain as soon as the buffer holds a row behind its end on its scaffold, or a later scaffold run, or the input is over;
    the rows of the previous window stay in the buffer because the next window of the same scaffold keeps them (left trim only)."""

    def __init__(self, windCoords, scaf_order=None, tail=None):
        """scaf_order / tail: sharded ingestion (plan_predefined_shards).  The windows are this rank's, the order of the scaffolds
        is the whole window list's, and `tail` says what the reference's reader would still find behind this rank's rows:
        "wanted" (a run of a scaffold of the window list: the remaining windows come out empty), "rows" (only other rows: the
        next window skips them, comes out empty and the generator ends) or None (nothing: the end of the file)."""
        self.coords = list(windCoords)
        all_scafs = list(scaf_order) if scaf_order is not None else [w[0] for w in self.coords]
        self.scafs = sorted(set(all_scafs), key=all_scafs.index)
        self.rank = {s: k for k, s in enumerate(self.scafs)}
        self.tail = tail
        self.wi = 0                          # next window
        self.cur = 0                         # the "line in hand", buffer row
        self.prev = None                     # (scaffold, lo, hi) of the previous window's rows, buffer rows
        self.done = False                    # the generator has returned (genotype file exhausted while looking for a window)

    def feed(self, run_starts, run_names, positions, final):
        positions = np.asarray(positions)
        n = len(positions)
        runs = _runs(run_starts, n)
        starts = np.asarray([a for a, _ in runs], dtype=np.int64)
        T = WindowTable()
        cur, prev = self.cur, self.prev
        while self.wi < len(self.coords) and not self.done:
            w = self.coords[self.wi]
            scaf, start, end = w[0], int(w[1]), int(w[2])
            ID = w[3] if len(w) > 3 else "NA"
            widx = self.rank[scaf]
            c = cur
            while c < n:                      # genomics.py:2142-2145: skip the runs of scaffolds before this window's
                r = int(np.searchsorted(starts, c, side="right")) - 1
                name = run_names[r]
                if name not in self.rank or self.rank[name] < widx:
                    c = runs[r][1]
                else:
                    break
            if c >= n and not final:
                cur = c                       # skipped rows are never looked at again; the window waits for more input
                break
            if c == n and final and self.tail == "rows":
                c = n + 1                     # the rows behind this rank's are skipped: the reader arrives at the end of the file
            new = None
            if c < n:
                r = int(np.searchsorted(starts, c, side="right")) - 1
                if run_names[r] == scaf:
                    a, b = runs[r]
                    _check_sorted(positions, a, b, scaf)
                    p = positions[c:b]
                    x = c + int(np.searchsorted(p, start, side="left"))
                    y = c + int(np.searchsorted(p, end, side="right"))
                    if y == b and b == n and not final:
                        cur = c               # the run may go on in the next piece
                        break
                    if y > x:
                        new = (x, y)
                        c = y
                    else:
                        c = x
            kept = None
            if prev is not None and prev[0] == scaf and prev[2] > prev[1]:
                k_lo = prev[1] + int(np.searchsorted(positions[prev[1]:prev[2]], start, side="left"))
                if prev[2] > k_lo:
                    kept = (k_lo, prev[2])
            if kept and new:
                rows = (kept[0], new[1])
            elif kept:
                rows = kept
            elif new:
                rows = new
            else:
                rows = (c, c) if c <= n else (n, n)
            T.add(scaf, start, end, rows[0], rows[1], ID)
            prev = (scaf, rows[0], rows[1])
            cur = c
            self.wi += 1
            if final and (cur > n or (cur >= n and self.tail is None)):
                self.done = True
        T.finish(positions)
        T.dup = np.zeros(T.n, dtype=bool)
        keep_from = cur if prev is None else min(cur, prev[1])
        keep_from = min(keep_from, n)
        if self.wi >= len(self.coords) or self.done:          # no window is left to ask for a row: nothing is carried over
            self.cur, self.prev = 0, None
            return T, n
        self.cur = cur - keep_from
        self.prev = None if prev is None else (prev[0], prev[1] - keep_from, prev[2] - keep_from)
        return T, keep_from


def plan_predefined_shards(runs, data_start, size, coords, n_ranks, max_share=0.75):
    """Sharded ingestion for `--windType predefined`.  predefinedCoordWindows (genomics.py:2112-2171) walks the window list and the
    file in step and never goes back, so in general a window's content depends on everything in front of it.  It does NOT when
    (1) the windows are grouped by scaffold, (2) every scaffold of the window list is one run of the file, (3) those runs come in
    the order of the window list: then the reader stands at the start of a scaffold's run when its first window is asked for,
    whatever came before, and the file can be cut at run boundaries.
    runs: [(byte offset of the run's first line, scaffold)] of the whole file; coords: the window list.  Returns None (replicated
    ingestion) or, per rank, (first byte, end byte, indices of the rank's windows, tail) -- tail as PredefinedWindowStream takes it."""
    scaf_seq = [w[0] for w in coords]
    scafs = sorted(set(scaf_seq), key=scaf_seq.index)
    order = {s_: k for k, s_ in enumerate(scafs)}
    widx = [order[s_] for s_ in scaf_seq]
    if any(a > b for a, b in zip(widx, widx[1:])):
        return None
    wanted = [(off, name) for off, name in runs if name in order]
    names = [name for _, name in wanted]
    if len(set(names)) != len(names) or set(names) != set(scafs):
        return None
    if any(order[a] > order[b] for a, b in zip(names, names[1:])):
        return None
    starts = [off for off, _ in runs]
    cuts = [data_start]
    for r in range(1, n_ranks):
        target = data_start + (size - data_start) * r // n_ranks
        nxt = [o for o in starts if o >= target]
        cuts.append(max(nxt[0] if nxt else size, cuts[-1]))
    cuts.append(size)
    if max(cuts[r + 1] - cuts[r] for r in range(n_ranks)) > max_share * max(size - data_start, 1):
        return None
    plan = []
    for r in range(n_ranks):
        mine = set(name for off, name in wanted if cuts[r] <= off < cuts[r + 1])
        tail = ("wanted" if any(off >= cuts[r + 1] for off, _ in wanted) else
                "rows" if any(off >= cuts[r + 1] for off in starts) else None)
        plan.append((cuts[r], cuts[r + 1], [k for k, w in enumerate(coords) if w[0] in mine], tail))
    return plan
