"""`.geno` input for the drop-in drivers: header handling + bulk tokenisation through the C-ABI (K0).

Replaces the line-at-a-time GenoFileReader (genomics.py:1914-1945).  The file format is unchanged:
whitespace separated, first line `#CHROM POS name...` (or `--header` text), `.gz` by suffix, stdin if no file.
"""
import ctypes as C
import gzip
import os
import sys

import numpy as np

from . import _lib
from ._lib import check
from .engine import encode_text


class _Stdin:
    """sys.stdin.buffer with a push-back: --inferPloidy looks at the first data row of a piped input before the run reads it"""

    def __init__(self):
        self.head = b""

    def unread(self, data):
        self.head = data + self.head

    def read(self, n=-1):
        if not self.head:
            return sys.stdin.buffer.read(n)
        if n is None or n < 0:
            out, self.head = self.head + sys.stdin.buffer.read(), b""
            return out
        out, self.head = self.head[:n], self.head[n:]
        if len(out) < n:
            out += sys.stdin.buffer.read(n - len(out))
        return out

    def readline(self):
        if not self.head:
            return sys.stdin.buffer.readline()
        nl = self.head.find(b"\n")
        if nl < 0:
            out, self.head = self.head + sys.stdin.buffer.readline(), b""
            return out
        out, self.head = self.head[:nl + 1], self.head[nl + 1:]
        return out

    def tell(self):
        raise OSError("stdin is not seekable")


STDIN = _Stdin()


def read_all(path):
    """Whole input as bytes (gunzipped when the name ends in .gz; stdin when path is None)."""
    if path is None:
        return STDIN.read()
    if str(path).endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def bgzf_walk(buf, limit=None, max_text=0):
    """member table of the BGZF bytes buf[:limit] (pg_bgzf_walk): (in_off, in_len, out_len, crc) as uint32 arrays -- where each
    member's deflate stream lies in buf, its inflated size and checksum --, the bytes the walked members occupy, their text bytes.
    Stops in front of an incomplete member, or once the members hold max_text bytes of text (0: no limit)."""
    arr = np.frombuffer(buf, dtype=np.uint8)
    n = len(arr) if limit is None else max(min(int(limit), len(arr)), 0)
    cap = n // 28 + 1
    tab = np.empty((4, cap), dtype=np.uint32)
    k, used, text = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    ptr = [C.c_void_p(tab[i].ctypes.data) for i in range(4)]
    rc = _lib.lib().pg_bgzf_walk(C.c_void_p(arr.ctypes.data if n else 0), n, cap, int(max_text), ptr[0], ptr[1], ptr[2], ptr[3],
                                 C.byref(k), C.byref(used), C.byref(text))
    if rc == _lib.PG_ERR_PARSE:
        raise ValueError("input stops being BGZF in the middle (bad member header)")
    check(rc)
    m = int(k.value)
    return (tab[0, :m], tab[1, :m], tab[2, :m], tab[3, :m]), int(used.value), int(text.value)


def bgzf_inflate(buf, tab, n_threads=0, dst=None):
    """the text of the members `tab` (bgzf_walk) of buf as a uint8 array (dst when given: a uint8 array that holds it), inflated
    by the library's host threads (zlib; the checksums are verified)"""
    in_off, in_len, out_len, crc = tab
    arr = np.frombuffer(buf, dtype=np.uint8)
    out_off = np.zeros(len(out_len) + 1, dtype=np.int64)
    np.cumsum(out_len, out=out_off[1:])
    if dst is None:
        dst = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
    elif dst.dtype != np.uint8 or not dst.flags.c_contiguous or dst.size < int(out_off[-1]):
        raise ValueError("bgzf_inflate: dst must be a contiguous uint8 array of at least %d bytes" % int(out_off[-1]))
    rc = _lib.lib().pg_inflate_members(C.c_void_p(arr.ctypes.data if len(arr) else 0), C.c_void_p(in_off.ctypes.data),
                                       C.c_void_p(in_len.ctypes.data), out_off, C.c_void_p(out_len.ctypes.data),
                                       C.c_void_p(crc.ctypes.data), len(out_len), C.c_void_p(dst.ctypes.data), int(n_threads))
    if rc == _lib.PG_ERR_PARSE:
        raise ValueError(_lib.lib().pg_last_error().decode("utf-8", "replace"))
    check(rc)
    return dst[:int(out_off[-1])]


def bgzf_compress(text, level=6, block=65280, eof_marker=True, n_threads=0):
    """text (bytes-like) as BGZF bytes, the way bgzip writes them (pg_bgzf_compress: a pool of host threads)"""
    arr = np.frombuffer(text, dtype=np.uint8)
    cap = len(arr) + len(arr) // 1000 + (len(arr) // block + 2) * 64 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_int64(0)
    check(_lib.lib().pg_bgzf_compress(C.c_void_p(arr.ctypes.data if len(arr) else 0), len(arr), int(level), int(block), int(bool(eof_marker)),
                                      C.c_void_p(out.ctypes.data), cap, C.byref(n), int(n_threads)))
    return out[:int(n.value)]


class BgzfWriter:
    """Write-only file object that produces BGZF (what `... | bgzip > out.geno.gz` gives, VCF_processing/README.md:33): the bytes are
    collected and deflated 16 MiB at a time by the library's host threads (pg_bgzf_compress).  A valid gzip file for every other
    reader; the drivers inflate it on the GPU."""

    PIECE = 255 * 65280 + 65280 * 2                   # a multiple of the member size: only the file's last member is short

    def __init__(self, path, level=6, n_threads=0):
        self.f = open(path, "wb")
        self.level, self.n_threads = level, n_threads
        self.buf = bytearray()

    def write(self, data):
        self.buf += data
        while len(self.buf) >= self.PIECE:
            self.f.write(memoryview(bgzf_compress(memoryview(self.buf)[:self.PIECE], self.level, eof_marker=False, n_threads=self.n_threads)))
            del self.buf[:self.PIECE]
        return len(data)

    def write_members(self, members):
        """whole BGZF members made elsewhere (the device's k_deflate): what is still buffered as text goes out first, as members of
        its own -- a BGZF file is any sequence of members"""
        if len(self.buf):
            self.f.write(memoryview(bgzf_compress(bytes(self.buf), self.level, eof_marker=False, n_threads=self.n_threads)))
            self.buf = bytearray()
        self.f.write(memoryview(members))

    def flush(self):
        pass

    def close(self):
        if self.f is None:
            return
        self.f.write(memoryview(bgzf_compress(bytes(self.buf), self.level, eof_marker=True, n_threads=self.n_threads)))
        self.f.close()
        self.f = None

    def abort(self):
        """a failed run: the file is closed as it is -- what is still buffered is dropped and NO end-of-file member is written, so
        that the truncated output does not pass for a finished BGZF file (htslib: "no EOF marker")"""
        if self.f is not None:
            self.f.close()
            self.f = None
        self.buf = bytearray()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *exc):
        if exc_type is not None:
            self.abort()
        else:
            self.close()


class BgzfSpan:
    """A block of a BGZF input that is still deflated: `head` (text the reader already holds: what the previous block left behind
    its last line feed) followed by the text of whole members, cut behind the block's last line feed (len() = that many bytes of
    text).  Engine.tokenize_submit_bgzf sends the members to the device as they are and inflates them there; bytes(span) inflates
    them on the host (a block the device tokenizer does not take)."""

    def __init__(self, head, comp, tab, text_len, first_line, file=None):
        self.head, self.comp, self.tab, self.text_len, self.first_line = head, comp, tab, text_len, first_line
        self.file = file                  # (file descriptor, offset of comp in the file): the engine's staging threads then read the members themselves

    def __len__(self):
        return self.text_len

    def __bytes__(self):
        return (self.head + bgzf_inflate(self.comp, self.tab).tobytes())[:self.text_len]

    def members_text_len(self):
        return int(self.tab[2].sum(dtype=np.int64))

    def inflate_into(self, dst, engine=None, n_threads=0):
        """the block's text in dst (a uint8 array of at least len(head) + members_text_len() bytes; page-locked when it comes from
        the engine's pool): the head, then the members inflated behind it -- by the device when an engine is given (k_inflate +
        the CRC-32 checked on the way out, the text copied back), else by the library's host threads.  Returns dst[:len(self)]."""
        h, total = len(self.head), self.members_text_len()
        if dst.size < h + total:
            raise ValueError("BgzfSpan.inflate_into: %d bytes needed, dst holds %d" % (h + total, dst.size))
        if h:
            dst[:h] = np.frombuffer(self.head, dtype=np.uint8)
        if engine is not None:
            engine.inflate_members(self.comp, self.tab, dst[h:h + total])
        else:
            bgzf_inflate(self.comp, self.tab, n_threads, dst=dst[h:h + total])
        return dst[:self.text_len]

    def __getitem__(self, key):                 # (the first bytes of the block: what the ingestion loop looks at to bound its rows)
        if isinstance(key, slice) and key.start in (None, 0) and key.step is None:
            line = self.first_line + b"\n"
            return line[:key.stop] if key.stop is not None else line
        raise TypeError("a BgzfSpan offers its first line only")


class BgzfFile:
    """Read-only file object over a BGZF file (bgzip / htslib: a gzip file made of independent members of at most 64 KiB,
    each announcing its compressed size in a 'BC' extra field).  A plain gzip stream has to be inflated serially, which caps
    `.geno.gz` ingestion at a few hundred MB/s of text; BGZF members are independent: read_span() hands out whole blocks of them
    still deflated (they are inflated on the device), read(n) / readline() inflate them with the library's host threads
    (pg_inflate_members).  Offers read(n), readline() and close(), which is all BlockReader needs."""

    CHUNK = 32 << 20                    # compressed bytes fetched per refill at most (the first refills are small: header line, cuts)

    def __init__(self, path, n_threads=0):
        import os
        self.raw = open(path, "rb")
        self.n_threads = n_threads
        self.pending = b""               # compressed bytes not yet split into whole members
        self.buf = bytearray()           # inflated bytes not yet handed out
        self.eof = False
        self.cpos = 0                    # file offset of pending[0]
        self.produced = 0                # inflated bytes since the last seek_member()
        self.track = []                  # (inflated offset, member file offset) of the members behind the bytes still buffered
        self.keep_track = False          # ... of every member since seek_member (a scan that needs virtual positions)
        self.stop = None                 # (member file offset, bytes of that member to keep): where a rank's share ends
        self.size = os.path.getsize(path)
        self._next_read = 1 << 16
        self._ratio = 12.0               # text bytes per compressed byte, as seen so far (read_span sizes its reads with it)
        self.alloc = np.empty            # read_span: allocator of the buffers the members are read into (the engine's page-locked pool)

    @staticmethod
    def is_bgzf(path):
        try:
            with open(path, "rb") as f:
                h = f.read(18)
        except OSError:
            return False
        return (len(h) == 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[10:12] == b"\x06\x00" and h[12:14] == b"BC"
                and h[14:16] == b"\x02\x00")

    @staticmethod
    def _inflate(member):
        import zlib
        (in_off, in_len, out_len, crc), used, _ = bgzf_walk(member)
        if used != len(member) or len(in_off) != 1:
            raise ValueError("damaged BGZF member")
        out = zlib.decompress(bytes(member[int(in_off[0]):int(in_off[0]) + int(in_len[0])]), wbits=-15)
        if len(out) != int(out_len[0]) or (zlib.crc32(out) & 0xffffffff) != int(crc[0]):
            raise ValueError("damaged BGZF member (size or checksum)")
        return out

    def _stop_member(self, data, off):
        """the share ends inside the member at data[off:] (self.stop): its first stop[1] bytes; reads on until the member is whole"""
        while True:
            if off + 18 <= len(data):
                size = int.from_bytes(data[off + 16:off + 18], "little") + 1
                if off + size <= len(data):
                    return self._inflate(data[off:off + size])[:self.stop[1]]
            more = self.raw.read(1 << 16)
            if not more:
                raise ValueError("truncated BGZF input")
            data = data + more

    def _refill(self, need=0):
        """inflate more members into buf: as many as hold `need` bytes of text (0: all that are at hand)"""
        new, exhausted = b"", False
        if len(self.pending) < (1 << 17):
            new = self.raw.read(self._next_read)
            self._next_read = min(self._next_read * 4, self.CHUNK)
            exhausted = not new
        data = self.pending + new if new else self.pending
        if not data:
            self.eof = True
            return
        limit, at_stop, last = len(data), False, None
        if self.stop is not None and self.cpos + limit > self.stop[0]:
            limit, at_stop = max(self.stop[0] - self.cpos, 0), True      # the share ends inside (or right before) the member there
        tab, used, _ = bgzf_walk(data, limit, need)
        in_off, in_len, out_len, _crc = tab
        if at_stop and used == limit:
            self.eof = True
            if self.stop[1] > 0:
                last = self._stop_member(data, limit)
        elif exhausted and used == len(data):
            self.eof = True
        elif len(in_off) == 0 and (exhausted or at_stop):
            raise ValueError("truncated BGZF input")
        self.pending = b"" if self.eof else data[used:]
        if len(in_off):
            text = bgzf_inflate(data, tab, self.n_threads)
            # where every member starts in the file: the first at cpos, each of the others behind its predecessor's trailer
            starts = np.empty(len(in_off), dtype=np.int64)
            starts[0] = self.cpos
            starts[1:] = self.cpos + in_off[:-1].astype(np.int64) + in_len[:-1] + 8
            at = self.produced + np.concatenate([[0], np.cumsum(out_len[:-1], dtype=np.int64)])
            self.track.extend(zip(at.tolist(), starts.tolist()))
            self.produced += len(text)
            self.buf += memoryview(text)
        self.cpos += used
        if last is not None:
            self.track.append((self.produced, self.cpos))
            self.produced += len(last)
            self.buf += last
        if not self.keep_track and len(self.track) > 4096:
            consumed = self.produced - len(self.buf)
            k = 0
            while k + 1 < len(self.track) and self.track[k + 1][0] <= consumed:
                k += 1
            del self.track[:k]

    def read_span(self, nbytes):
        """The next block of about nbytes of text as a BgzfSpan (its members still deflated), cut behind its last line feed; bytes
        when nothing compressed is left (the end of the input or of this reader's share: possibly without a final line feed), b""
        at the end.  What follows the block's last line feed stays buffered as the head of the next block; finding it costs one
        member inflated on the host (the last), the block's first line another."""
        import zlib
        head = bytes(self.buf)
        self.buf = bytearray()
        if self.eof:
            return head
        self.pending = b""                               # (compressed bytes an earlier read() fetched but did not inflate are read again)
        want = max(int(nbytes) - len(head), 1 << 16)
        end = self.size if self.stop is None else min(self.stop[0], self.size)
        span = max(int(want / self._ratio * 1.05), 1 << 20)
        fd = self.raw.fileno()
        while True:
            limit = min(self.cpos + span, end)
            # the members go into a buffer of the engine's page-locked pool when there is one (self.alloc): the copy to the device is
            # then one asynchronous DMA out of this very buffer, queued by the ingestion thread while this thread reads the next block
            buf = self.alloc((max(limit - self.cpos, 1),), np.uint8)
            view, got = memoryview(buf).cast("B"), 0
            while got < limit - self.cpos:
                k = os.preadv(fd, [view[got:min(got + (1 << 30), limit - self.cpos)]], self.cpos + got)
                if k <= 0:
                    raise ValueError("truncated BGZF input")
                got += k
            data = view[:got]
            tab, used, text = bgzf_walk(data, None, want)
            if text >= want or limit == end:
                break
            span = max(int(span * max(want / max(text, 1), 1.0) * 1.05), span + (1 << 20))
        in_off, in_len, out_len, crc = tab
        n = len(in_off)
        if text < want and self.cpos + used != end:
            raise ValueError("truncated BGZF input")
        if n:
            self._ratio = max(text / max(used, 1), 1.0)
        self.cpos += used
        tail_extra = b""
        self.raw.seek(self.cpos)
        if self.cpos == end:
            self.eof = True
            if self.stop is not None and self.stop[0] < self.size and self.stop[1] > 0:
                tail_extra = self._stop_member(self.raw.read(1 << 16), 0)
        if n == 0:                                       # nothing compressed left (the end of the input / of the share)
            return head + tail_extra

        def member_text(k):
            a = int(in_off[k])
            return zlib.decompress(data[a:a + int(in_len[k])], wbits=-15)

        # the block ends behind the last line feed of its members' text: walk back from the last member
        tail, k, found = [], n - 1, False
        while k >= 0 and n - k <= 64:
            t = member_text(k)
            cut = t.rfind(b"\n")
            if cut >= 0:
                tail.append(t[cut + 1:])
                found = True
                break
            tail.append(t)
            k -= 1
        if not found:
            # no line ends in the last members (lines of megabytes): this block is inflated on the host and handed on as text
            body = head + bgzf_inflate(data[:used], tab, self.n_threads).tobytes() + tail_extra
            if self.eof:
                return body
            cut = body.rfind(b"\n") + 1
            self.buf = bytearray(body[cut:])
            return body[:cut] if cut else self.read_span(len(body) + int(nbytes))
        tail = b"".join(reversed(tail))
        text_len = len(head) + text - len(tail)
        self.buf = bytearray(tail + tail_extra)
        # the block's first line: in head, or head + the start of the first members' text
        nl = head.find(b"\n")
        if nl >= 0:
            first = head[:nl]
        else:
            first, k = head, 0
            while True:
                t = member_text(k)
                nl = t.find(b"\n")
                if nl >= 0:
                    first += t[:nl]
                    break
                first += t
                k += 1
        return BgzfSpan(head, buf[:used], (in_off, in_len, out_len, crc), text_len, first)

    def set_stop(self, coffset, uoffset):
        """end this reader's share at byte `uoffset` of the member at file offset `coffset` (which may already be buffered)"""
        self.stop = (coffset, uoffset)
        for at, st in self.track:
            if st == coffset:
                consumed = self.produced - len(self.buf)
                keep = max(at + uoffset - consumed, 0)
                if keep < len(self.buf):
                    del self.buf[keep:]
                self.produced = consumed + len(self.buf)
                self.eof = True
                return
        if self.cpos > coffset:                                   # already past it, nothing of it buffered: an empty share
            self.buf = bytearray()
            self.eof = True

    def seek_member(self, guess):
        """continue reading at the first member that starts at or behind file offset `guess` (the end of the file if none)"""
        pos = guess
        self.raw.seek(pos)
        window = b""
        base = pos
        found = None
        while found is None:
            more = self.raw.read(1 << 20)
            window += more
            at = 0
            while True:
                k = window.find(b"\x1f\x8b\x08\x04", at)
                if k < 0 or k + 18 > len(window):
                    break
                if window[k + 10:k + 16] == b"\x06\x00BC\x02\x00":
                    size = int.from_bytes(window[k + 16:k + 18], "little") + 1
                    nxt = base + k + size
                    if nxt == self.size:
                        found = base + k
                        break
                    if nxt + 4 <= self.size:                      # the next member must start where this one says it ends
                        here = self.raw.tell()
                        self.raw.seek(nxt)
                        good = self.raw.read(4) == b"\x1f\x8b\x08\x04"
                        self.raw.seek(here)
                        if good:
                            found = base + k
                            break
                at = k + 1
            if found is None and not more:
                found = self.size
            if found is None and len(window) > (2 << 20):
                cutoff = len(window) - 64
                base += cutoff
                window = window[cutoff:]
        self.raw.seek(found)
        self.pending, self.buf, self.eof = b"", bytearray(), False
        self.cpos, self.produced = found, 0
        self.track = []
        return found

    def virtual_of(self, offset):
        """(member file offset, offset inside the member) of inflated byte `offset` since seek_member (needs track)"""
        import bisect
        if not self.track:
            return self.size, 0
        k = bisect.bisect_right([t[0] for t in self.track], offset) - 1
        if k < 0:
            k = 0
        return self.track[k][1], offset - self.track[k][0]

    def read(self, n=-1):
        if n is None or n < 0:
            while not self.eof:
                self._refill()
            out = bytes(self.buf)
            self.buf = bytearray()
            return out
        while len(self.buf) < n and not self.eof:
            self._refill(n - len(self.buf))
        out = bytes(self.buf[:n])
        del self.buf[:n]
        return out

    def readline(self):
        while True:
            k = self.buf.find(b"\n")
            if k >= 0:
                out = bytes(self.buf[:k + 1])
                del self.buf[:k + 1]
                return out
            if self.eof:
                out = bytes(self.buf)
                self.buf = bytearray()
                return out
            self._refill(1)

    def close(self):
        self.raw.close()


class GzipStream:
    """ONE gzip stream (what `gzip` writes; the reference reads it with gzip.open, genomics.py:1917): inflated serially by zlib in the
    library (pg_gzip_read_lines), straight into the block's buffer -- no Python byte strings on the way.  BGZF files never come here
    (BgzfFile: members in parallel, on the device)."""

    MARGIN = 16 << 20

    def __init__(self, path):
        self._L = _lib.lib()
        h = C.c_void_p()
        check(self._L.pg_gzip_open(os.fsencode(str(path)), C.byref(h)))
        self._h = h
        self.eof = False

    def read_lines(self, want):
        """at least `want` bytes of text (fewer at the end of the input) up to a line feed, as a memoryview of a fresh array;
        want None: everything that is left"""
        if want is None:
            parts = []
            while True:
                b = self.read_lines(256 << 20)
                if len(b) == 0:
                    break
                parts.append(np.frombuffer(b, dtype=np.uint8))
            return memoryview(np.concatenate(parts)) if len(parts) > 1 else memoryview(parts[0]) if parts else b""
        if self.eof:
            return b""
        want = max(int(want), 1)
        parts, total = [], 0
        while True:
            cap = want + self.MARGIN
            buf = np.empty(cap, dtype=np.uint8)
            got, complete, eof = C.c_int64(0), C.c_int(0), C.c_int(0)
            check(self._L.pg_gzip_read_lines(self._h, C.c_void_p(buf.ctypes.data), cap, want, C.byref(got), C.byref(complete), C.byref(eof)))
            if got.value:
                parts.append(buf[:got.value])
                total += got.value
            if eof.value:
                self.eof = True
            if complete.value or eof.value:
                break
            want = 1                                   # (a line longer than the margin: the rest of it)
        if not parts:
            return b""
        return memoryview(parts[0] if len(parts) == 1 else np.concatenate(parts))

    def readline(self):
        return bytes(self.read_lines(1))

    def stats(self):
        """which decoder read the stream: {"decoder", "threads", "batches", "serial_takeovers"}"""
        if self._h is not None:
            a = np.zeros(4, dtype=np.int64)
            check(self._L.pg_gzip_stats(self._h, a))
            self._stats = {"decoder": ("zlib" if a[0] == -2 else "serial (pg_fast_inflate.h)" if a[0] <= 0 else "chunks side by side (pg_par_gunzip.h)"),
                           "threads": int(max(a[0], 1)), "batches": int(a[1]), "serial_takeovers": int(a[2])}
        return getattr(self, "_stats", None)

    def close(self):
        if self._h is not None:
            self.stats()
            self._L.pg_gzip_close(self._h)
            self._h = None


class BlockReader:
    """The input as a sequence of byte blocks that end at line boundaries (gunzipped when the name ends in .gz -- in parallel
    when the file is BGZF, i.e. written by bgzip; stdin when path is None).  read_block(None) returns everything that is
    left."""

    def __init__(self, path):
        self.path = path
        self.stop = None                  # byte offset at which this reader ends (restrict()); None = end of file
        self.mm = None                    # plain files are memory-mapped: blocks are views of the page cache, not copies
        if path is None:
            self.f = STDIN
        elif str(path).endswith(".gz"):
            # (PG_GZIP_NATIVE=0: Python's gzip module, the reader of rounds 3 - 5)
            self.f = BgzfFile(path) if BgzfFile.is_bgzf(path) else (GzipStream(path) if os.environ.get("PG_GZIP_NATIVE", "1") != "0" else gzip.open(path, "rb"))
        else:
            import mmap
            self.f = open(path, "rb")
            if os.path.getsize(path) > 0:
                self.mm = mmap.mmap(self.f.fileno(), 0, access=mmap.ACCESS_READ)
                probe = np.frombuffer(self.mm, dtype=np.uint8)
                self._mm_addr = int(probe.ctypes.data)       # where the mapping begins: file_range() turns a block into (fd, offset)
                del probe
        self.bytes_read = 0
        self.spans = False                # BGZF input: read_block() may hand out BgzfSpan blocks (members still deflated: the device inflates them)

    def read_header(self):
        line = self.mm.readline() if self.mm is not None else self.f.readline()
        self.bytes_read += len(line)
        return line

    def seekable_text(self):
        """plain (uncompressed) file on disk: byte ranges of it can be handed to different ranks"""
        return self.path is not None and not str(self.path).endswith(".gz")

    def tell(self):
        return self.mm.tell() if self.mm is not None else self.f.tell()

    def restrict(self, start, stop):
        """read only the bytes [start, stop) of the file from now on (both at line starts)"""
        (self.mm if self.mm is not None else self.f).seek(start)
        self.stop = stop

    def read_block(self, nbytes=None):
        """the next block of whole lines (about nbytes of them; everything that is left when None); b"" at the end.  For a
        memory-mapped file the block is a memoryview of the mapping: the reader thread copies nothing, the tokenizer's threads
        fault the pages in."""
        if self.mm is not None:
            mm = self.mm
            pos = mm.tell()
            limit = len(mm) if self.stop is None else min(self.stop, len(mm))
            end = limit if nbytes is None else min(pos + nbytes, limit)
            if pos >= limit:
                return b""
            if end < limit:
                nl = mm.find(b"\n", max(end - 1, pos), limit)
                end = limit if nl < 0 else nl + 1
            mm.seek(end)
            self.bytes_read += end - pos
            return memoryview(mm)[pos:end]
        if self.spans and nbytes is not None and isinstance(self.f, BgzfFile):
            data = self.f.read_span(nbytes)
            self.bytes_read += len(data)
            return data
        if self.stop is not None:
            left = max(self.stop - self.f.tell(), 0)
            if nbytes is None or nbytes >= left:
                data = self.f.read(left)
                self.bytes_read += len(data)
                return data
            data = self.f.read(nbytes)
            if data and not data.endswith(b"\n"):
                data += self.f.readline()
            over = self.f.tell() - self.stop
            if over > 0:                                 # cannot happen when stop is a line start; be safe
                data = data[:len(data) - over]
            self.bytes_read += len(data)
            return data
        if isinstance(self.f, GzipStream):
            data = self.f.read_lines(nbytes)
        elif nbytes is None:
            data = self.f.read()
        else:
            data = self.f.read(nbytes)
            if data and not data.endswith(b"\n"):
                data += self.f.readline()
        self.bytes_read += len(data)
        return data

    packed = False

    def file_range(self, block):
        """(file descriptor, file offset) of a block read_block() handed out as a view of the memory-mapped file, else None"""
        if self.mm is None or not isinstance(block, memoryview) or len(block) == 0:
            return None
        off = int(np.frombuffer(block, dtype=np.uint8).ctypes.data) - self._mm_addr
        if off < 0 or off + len(block) > len(self.mm):
            return None
        return self.f.fileno(), off

    def input_size(self):
        return os.path.getsize(self.path) if self.path is not None and os.path.exists(str(self.path)) else -1

    def shard(self, world, comm, wanted, max_share=0.75):
        """Restrict this reader to rank `world.rank`'s slice of the data lines: the byte range between the scaffold-run
        boundaries nearest to the equal split (find_run_boundary; each rank looks for its own start, one all-gather shares
        them).  Returns False, leaving the reader untouched, when the input cannot be split (stdin, gzip) or has too few runs
        for a useful split (some rank would hold more than max_share of the bytes)."""
        if isinstance(self.f, BgzfFile):
            return self._shard_bgzf(world, comm, wanted, max_share)
        if not self.seekable_text():
            return False
        size = os.path.getsize(self.path)
        start = self.tell()                                    # the header line, if any, has been consumed
        mine, scanned = float(start), 0
        if world.rank > 0:
            # the search starts a little before the equal split, so that a boundary sitting exactly on it (equal scaffolds) is
            # not missed by a few bytes
            stride = (size - start) // world.size
            guess = start + stride * world.rank - min(max(stride // 64, 1 << 12), stride // 2)
            cut, scanned = find_run_boundary(self.path, max(guess, start), wanted)
            mine = float(cut)
        cuts = [int(c) for c in comm.allgather(np.array([mine])).ravel()] + [size]
        for r in range(1, len(cuts)):
            cuts[r] = max(cuts[r], cuts[r - 1])
        share = max(cuts[r + 1] - cuts[r] for r in range(world.size)) / max(size - start, 1)
        self.bytes_read += scanned
        if share > max_share:
            return False
        self.restrict(cuts[world.rank], cuts[world.rank + 1])
        return True

    def text_runs(self, world):
        """[(byte offset, scaffold)] of the first data line of every scaffold run among the lines that START in rank
        `world.rank`'s equal share of the data bytes (pg_text_runs over the memory-mapped range); None when the input is not
        plain text on disk.  The ranks' lists, concatenated with a seam run merged into its predecessor of the same name, are
        the runs of the whole file."""
        import ctypes as C
        from . import _lib
        if isinstance(self.f, BgzfFile) or not self.seekable_text() or self.mm is None:
            return None
        size = len(self.mm)
        start = self.tell()
        stride = max((size - start) // world.size, 1)

        def cut(r):
            if r <= 0:
                return start
            if r >= world.size:
                return size
            guess = min(start + stride * r, size)
            nl = self.mm.find(b"\n", max(guess - 1, start), size)
            return size if nl < 0 else nl + 1

        a, b = cut(world.rank), cut(world.rank + 1)
        if b <= a:
            return []
        view = memoryview(self.mm)[a:b]
        ptr, nbytes, _keep = _lib.text_ptr(view)
        L = _lib.lib()
        cap = 1024
        while True:
            starts = np.zeros(cap, dtype=np.int64)
            n = C.c_int64(0)
            _lib.check(L.pg_text_runs(ptr, nbytes, starts, cap, C.byref(n)))
            if n.value <= cap:
                break
            cap = int(n.value)
        out = []
        for off in starts[:n.value]:
            q = a + int(off)
            e = q
            while e < b and self.mm[e] not in (9, 32, 10):
                e += 1
            out.append((q, self.mm[q:e].decode("utf-8", "replace")))
        self.bytes_read += b - a
        del view, _keep, ptr
        return out

    def shard_lines(self, world):
        """Restrict this reader to rank `world.rank`'s share of the data lines, cut at ANY line boundary (sites are independent:
        freq.py, whose reference reads slices of sites in parallel, freq.py:23-28).  Every rank finds its own start and its
        successor's the same way (the first line that starts at or behind the equal split), so no exchange is needed.  False,
        and the reader untouched, when the input is not plain text on disk."""
        if isinstance(self.f, BgzfFile):
            return self._shard_lines_bgzf(world)
        if not self.seekable_text():
            return False
        size = os.path.getsize(self.path)
        start = self.tell()
        stride = max((size - start) // world.size, 1)

        def cut(r):
            if r <= 0:
                return start
            if r >= world.size:
                return size
            guess = min(start + stride * r, size)
            with open(self.path, "rb") as f:
                f.seek(max(guess - 1, start))
                if guess > start:
                    f.readline()                               # to the end of the line that holds byte guess - 1
                return min(f.tell(), size)

        a, b = cut(world.rank), cut(world.rank + 1)
        self.restrict(a, max(a, b))
        return True

    def _shard_lines_bgzf(self, world):
        """the same on bgzip-compressed text: rank r's share starts at the first line that begins in or behind the member at r / N of
        the compressed bytes; a cut is a (member file offset, offset inside the member) pair (every rank finds its own cut and its
        successor's the same way: no exchange)"""
        bz = self.f
        size = bz.size

        def cut(r):
            if r <= 0:
                return (0, 0)
            if r >= world.size:
                return (size, 0)
            probe = BgzfFile(self.path)
            probe.keep_track = True
            probe.seek_member(size * r // world.size)
            skipped = len(probe.readline())                    # the line that straddles into the member belongs to the left
            tok = probe.virtual_of(skipped)                    # (at a member's end: that member and its length -- the same place)
            probe.close()
            return (int(tok[0]), int(tok[1]))

        a, b = cut(world.rank), cut(world.rank + 1)
        self.restrict_virtual(a, max(a, b), first=(world.rank == 0))
        return True

    def _shard_bgzf(self, world, comm, wanted, max_share):
        """BGZF (bgzip) input: the cuts are (member file offset, offset inside the member) pairs; a rank starts by seeking to its
        member and dropping the bytes in front of its first line, and stops inside the member its successor starts in"""
        bz = self.f
        size = bz.size
        mine = (0.0, 0.0)
        scanned = 0
        if world.rank > 0:
            stride = size // world.size
            guess = stride * world.rank - min(max(stride // 64, 1 << 12), stride // 2)
            (c, u), scanned = find_run_boundary_bgzf(self.path, max(guess, 0), wanted)
            mine = (float(c), float(u))
        allc = comm.allgather(np.array(mine)).reshape(world.size, 2)
        cuts = [(int(c), int(u)) for c, u in allc] + [(size, 0)]
        for r in range(2, len(cuts)):
            cuts[r] = max(cuts[r], cuts[r - 1])
        starts = [0] + [c for c, _ in cuts[1:]]
        if max(starts[r + 1] - starts[r] for r in range(world.size)) / max(size, 1) > max_share:
            return False
        self.bytes_read += scanned
        if world.rank > 0:
            bz.seek_member(cuts[world.rank][0])
            bz.read(cuts[world.rank][1])
        if cuts[world.rank] >= cuts[world.rank + 1] and world.rank > 0:
            bz.buf, bz.eof = bytearray(), True                 # an empty share
        elif world.rank + 1 < world.size:
            bz.set_stop(*cuts[world.rank + 1])
        return True

    def restrict_virtual(self, start, end, first=False):
        """BGZF: read only the inflated bytes between the virtual positions start and end, (member file offset, offset inside the
        member) pairs; first: the reader stays where it is (behind the header line) instead of seeking to `start`"""
        bz = self.f
        start, end = (int(start[0]), int(start[1])), (int(end[0]), int(end[1]))
        if not first:
            bz.seek_member(start[0])
            bz.read(start[1])
        if end <= start and not first:
            bz.buf, bz.eof = bytearray(), True                 # an empty share
        elif end[0] < bz.size:
            bz.set_stop(*end)

    def to_geno(self, body, layout, n_threads=0, head_rows=0, pitch=None, alloc=None, keep_packed=False, narrow_ok=False):
        if isinstance(body, BgzfSpan):
            body = bytes(body)
        return encode(body, layout, n_threads, head_rows, pitch, alloc, narrow_ok)

    def close(self):
        if self.mm is not None:
            try:
                self.mm.close()
            except BufferError:            # a block is still referenced somewhere: the mapping goes with its last view
                pass
        if self.f is not STDIN:
            self.f.close()


def _scan_run_boundary(next_chunk, wanted, chunk=8 << 10, max_chunk=16 << 20):
    """Walk whole lines from `next_chunk(nbytes)` (bytes ending at a line boundary, b"" at the end) and return (offset of the first
    data line that starts a new scaffold run whose own scaffold and the preceding one are both wanted -- relative to the start of
    the walk, None if there is none --, bytes walked).  The first data line only names the current run."""
    import re
    scanned, base = 0, 0
    cur, pat = None, None
    while True:
        data = next_chunk(chunk)
        chunk = min(2 * chunk, max_chunk)                # small reads first: the boundary is usually near
        if not data:
            return None, scanned
        scanned += len(data)
        # the pattern needs the line feed in front of a line: behind the first chunk, give the chunk's first line the one that
        # ended the previous chunk (a run that ends exactly at a chunk seam would otherwise be cut one line late)
        lead = 0 if cur is None else 1
        if lead:
            data = b"\n" + data
        at = lead
        while at < len(data):
            if cur is None:                              # first data line of the scan: it only names the current run
                nl = data.find(b"\n", at)
                line = data[at:nl if nl >= 0 else len(data)]
                tok = line.split(None, 1)
                if tok and not line.startswith(b"#"):
                    cur = tok[0]
                    pat = re.compile(rb"\n(?!" + re.escape(cur) + rb"[ \t])")
                at = (nl + 1) if nl >= 0 else len(data)
                continue
            m = pat.search(data, max(at - 1, 0))
            if m is None:
                break
            at = m.end()                                  # start of a line that does not begin with `cur` + blank
            if at >= len(data):
                break
            nl = data.find(b"\n", at)
            line = data[at:nl if nl >= 0 else len(data)]
            tok = line.split(None, 1)
            if not tok or line.startswith(b"#"):          # blank or comment line: not a data row
                at = (nl + 1) if nl >= 0 else len(data)
                continue
            prev, cur = cur, tok[0]
            pat = re.compile(rb"\n(?!" + re.escape(cur) + rb"[ \t])")
            if wanted(prev.decode("utf-8", "replace")) and wanted(cur.decode("utf-8", "replace")):
                return base + at - lead, scanned
            at = (nl + 1) if nl >= 0 else len(data)
        base += len(data) - lead


def find_run_boundary(path, guess, wanted):
    """Byte offset of the first data line at or behind offset `guess` that starts a new scaffold run whose own scaffold and the
    preceding one are both wanted (`wanted(name) -> bool`: --include / --exclude), or the file size when there is none; and
    the number of bytes scanned.  Windows never span scaffold runs, and the window generators carry state across a run
    boundary only around skipped scaffolds (genomics.py:2016-2023), so such a boundary is a place where the input can be
    split between ranks (the slice-parallel ingestion of the reference's freq.py:23-28, here on run boundaries)."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        # start at the line that holds the byte before `guess`: it names the run to the left of the first candidate line
        back = min(guess, 4 << 20)
        f.seek(guess - back)
        head = f.read(back)
        start = guess - back + head.rfind(b"\n", 0, max(back - 1, 0)) + 1
        f.seek(start)

        def next_chunk(n):
            data = f.read(n)
            if data and not data.endswith(b"\n"):
                data += f.readline()
            return data

        rel, scanned = _scan_run_boundary(next_chunk, wanted)
    return (size if rel is None else start + rel), scanned


def find_run_boundary_bgzf(path, guess, wanted):
    """The same on a BGZF file: the walk starts at the first line that begins in or behind the member at compressed offset >=
    `guess`; returns ((member file offset, offset inside the member) of the boundary line -- (file size, 0) if none --,
    inflated bytes walked)."""
    bz = BgzfFile(path)
    bz.keep_track = True
    bz.seek_member(guess)
    skipped = len(bz.readline())                           # the line that straddles into this member belongs to the left

    def next_chunk(n):
        data = bz.read(n)
        if data and not data.endswith(b"\n"):
            data += bz.readline()
        return data

    rel, scanned = _scan_run_boundary(next_chunk, wanted)
    out = (bz.size, 0) if rel is None else bz.virtual_of(skipped + rel)
    bz.close()
    return out, scanned + skipped


# ---- packed `.pgeno` files: a tokenised `.geno` kept on disk ---------------------------------------------------------------
# SURVEY.md 8f row 4.  Layout (little endian): magic, u32 length + JSON header {"names": [...], "ploidy": [...], "codec": ...}
# (file column order), then blocks { u64 n_rows, u32 n_runs, n_runs x (u64 first_row, u16 len, scaffold name),
# payload }, terminated by a block with n_rows = 0; payload = pos (i64 when the header says "pos_bytes": 8, else i32) || u8 cells[n_rows][n_cols], stored raw (codec "none") or as
# independently deflated 4 MiB chunks (codec "zlib", the default: u32 n_chunks, n_chunks x (u32 stored, u32 raw), data; both sides
# work on the chunks with a thread pool).  A cell byte = first allele code | second << 4 (one-hot
# codes A=1 C=2 G=4 T=8, 0 = missing), independent of the text format it came from and of any population layout, so one packed
# file serves every later run.  4x (raw) to ~30x (deflated) smaller than the text, no serial gunzip, no tokenizer: tools/geno_pack.py writes it, the drivers
# read it when the input name ends in .pgeno (-f is then only used for the default ploidy).
PGENO_MAGIC = b"PGENO1\n"


PGENO_CHUNK = 4 << 20


def _pool():
    from concurrent.futures import ThreadPoolExecutor
    return ThreadPoolExecutor(max(1, min(16, _lib.usable_cpus())))


class PackedWriter:
    """pos_bytes: width of the stored positions: 8 (int64: what the text tokenizers carry -- the reference parses Python integers,
    genomics.py:1884-1904, and chromosomes of more than 2^31 bases exist) or 4 (files written before round 5; the header then has
    no "pos_bytes" key)"""

    def __init__(self, path, names, ploidy, codec="zlib", pos_bytes=8):
        import json
        assert codec in ("none", "zlib") and pos_bytes in (4, 8)
        self.f = open(path, "wb")
        self.n_cols = len(names)
        self.codec = codec
        self.pos_bytes = pos_bytes
        head = {"names": list(names), "ploidy": [int(p) for p in ploidy], "codec": codec}
        if pos_bytes != 4:
            head["pos_bytes"] = pos_bytes
        head = json.dumps(head).encode()
        self.f.write(PGENO_MAGIC + len(head).to_bytes(4, "little") + head)

    def write_block(self, data, cells):
        """data: GenoData of the block (pos, run_starts, run_names); cells: uint8 [n_rows][n_cols]."""
        import zlib
        n = int(data.n_sites)
        if n == 0:
            return
        out = [n.to_bytes(8, "little"), len(data.run_names).to_bytes(4, "little")]
        for st, nm in zip(data.run_starts, data.run_names):
            b = nm.encode()
            out += [int(st).to_bytes(8, "little"), len(b).to_bytes(2, "little"), b]
        self.f.write(b"".join(out))
        pos = np.asarray(data.pos)
        if self.pos_bytes == 4 and len(pos) and (int(pos.max()) > 0x7FFFFFFF or int(pos.min()) < -0x80000000):
            raise ValueError("a position beyond 32 bits in a .pgeno file with 4-byte positions")
        payload = np.ascontiguousarray(pos, dtype="<i%d" % self.pos_bytes).tobytes() + np.ascontiguousarray(cells, dtype=np.uint8).tobytes()
        if self.codec == "none":
            self.f.write(payload)
            return
        view = memoryview(payload)
        parts = [view[a:a + PGENO_CHUNK] for a in range(0, len(payload), PGENO_CHUNK)]
        with _pool() as ex:
            comp = list(ex.map(lambda p: zlib.compress(p, 1), parts))          # zlib releases the GIL
        self.f.write(len(parts).to_bytes(4, "little"))
        self.f.write(b"".join(len(c).to_bytes(4, "little") + len(p).to_bytes(4, "little") for c, p in zip(comp, parts)))
        for c in comp:
            self.f.write(c)

    def close(self):
        self.f.write((0).to_bytes(8, "little"))
        self.f.close()


class _PackedBlock:
    """One block of a `.pgeno` file as read from disk: scaffold runs + either the materialised arrays (pos, cells) or the still
    deflated chunks (comp, table), which PackedReader.to_geno inflates straight into their destination (pg_inflate_chunks)."""

    def __init__(self, starts, names, n, n_cols, pos=None, cells=None, comp=None, table=None, fd=None, pos_off=0, cells_off=0, pos_bytes=4):
        self.starts, self.names, self.n, self.n_cols = starts, names, n, n_cols
        self.pos_bytes = pos_bytes        # width of the positions in the file (4 or 8); in memory they are int64
        self.pos, self.cells, self.comp, self.table = pos, cells, comp, table
        # codec "none": the payload stays in the file until somebody wants it -- the device route reads the cells with the staging
        # threads of the tokenizer (pg_stage_file), the host route straight into its destination
        self.fd, self.pos_off, self.cells_off = fd, pos_off, cells_off

    def in_file(self):
        return self.fd is not None and self.pos is None and self.comp is None

    def positions(self):
        if self.in_file():
            return np.frombuffer(os.pread(self.fd, self.pos_bytes * self.n, self.pos_off), dtype="<i%d" % self.pos_bytes).astype(np.int64)
        self.materialise()
        return self.pos

    @staticmethod
    def _read_into(fd, dst, off):
        view = memoryview(dst).cast("B")
        got = 0
        while got < len(view):
            k = os.preadv(fd, [view[got:got + (1 << 30)]], off + got)
            if k <= 0:
                raise ValueError("truncated .pgeno file")
            got += k

    def inflate_into(self, pos_dst, cells_dst, n_threads=0):
        """positions -> pos_dst[n] (int64), cells -> cells_dst[n][n_cols] (uint8, C-contiguous rows)"""
        if self.pos_bytes != 8 and (self.in_file() or self.comp is not None):      # 4-byte positions: through a buffer of their own
            tmp = np.empty(self.n, dtype=np.int32)
            self._inflate_raw(tmp, cells_dst, n_threads)
            pos_dst[...] = tmp
            return
        self._inflate_raw(pos_dst, cells_dst, n_threads)

    def _inflate_raw(self, pos_dst, cells_dst, n_threads=0):
        if self.in_file():
            assert pos_dst.flags.c_contiguous and cells_dst.flags.c_contiguous
            self._read_into(self.fd, pos_dst, self.pos_off)
            self._read_into(self.fd, cells_dst, self.cells_off)
            return
        if self.comp is None:
            pos_dst[...] = self.pos
            cells_dst[...] = self.cells
            return
        assert pos_dst.flags.c_contiguous and cells_dst.flags.c_contiguous
        stored = self.table[:, 0].astype(np.int64)
        off = np.concatenate([[0], np.cumsum(stored)[:-1]]).astype(np.int64)
        src = np.frombuffer(self.comp, dtype=np.uint8)
        check(_lib.lib().pg_inflate_chunks(C.c_void_p(src.ctypes.data), off, np.ascontiguousarray(stored),
                                           np.ascontiguousarray(self.table[:, 1].astype(np.int64)), len(stored),
                                           C.c_void_p(pos_dst.ctypes.data), self.pos_bytes * self.n, C.c_void_p(cells_dst.ctypes.data),
                                           self.n * self.n_cols, n_threads))

    def materialise(self):
        if self.in_file():
            pos, cells = np.empty(self.n, dtype=np.int64), np.empty((self.n, self.n_cols), dtype=np.uint8)
            self.inflate_into(pos, cells)
            self.pos, self.cells = pos, cells
        if self.comp is not None:
            pos, cells = np.empty(self.n, dtype=np.int64), np.empty((self.n, self.n_cols), dtype=np.uint8)
            self.inflate_into(pos, cells)
            self.pos, self.cells, self.comp, self.table = pos, cells, None, None
        return self

    def trim(self, a, b):
        """rows [a, b) of the block (materialised, unless the payload is still in the file: then only the offsets move)"""
        keep = np.flatnonzero((self.starts < b) & (np.append(self.starts[1:], self.n) > a))
        names = [self.names[k] for k in keep]
        starts = np.maximum(self.starts[keep] - a, 0)
        if self.in_file():
            return _PackedBlock(starts, names, b - a, self.n_cols, fd=self.fd, pos_off=self.pos_off + self.pos_bytes * a,
                                cells_off=self.cells_off + a * self.n_cols, pos_bytes=self.pos_bytes)
        self.materialise()
        return _PackedBlock(starts, names, b - a, self.n_cols, self.pos[a:b], self.cells[a:b], pos_bytes=self.pos_bytes)


class PackedReader:
    """Counterpart of BlockReader for `.pgeno` files: read_header() returns a `.geno`-style header line, read_block(nbytes)
    returns the raw blocks (about nbytes of cells) and to_geno() decodes them into slot order (pg_decode_packed)."""
    packed = True

    def __init__(self, path):
        import json
        self.path = path
        self.f = open(path, "rb")
        if self.f.read(len(PGENO_MAGIC)) != PGENO_MAGIC:
            raise ValueError("%s is not a .pgeno file" % path)
        hl = int.from_bytes(self.f.read(4), "little")
        self.head = json.loads(self.f.read(hl).decode())
        self.names, self.ploidy = self.head["names"], np.asarray(self.head["ploidy"], dtype=np.int32)
        self.pos_bytes = int(self.head.get("pos_bytes", 4))           # (files written before round 5: 4-byte positions, no key)
        if self.pos_bytes not in (4, 8):
            raise ValueError("%s: positions of %d bytes" % (path, self.pos_bytes))
        self.codec = self.head.get("codec", "none")
        if self.codec not in ("none", "zlib"):
            raise ValueError("%s: unknown codec %r" % (path, self.codec))
        self.n_cols = len(self.names)
        self._size = os.path.getsize(path)
        self.bytes_read = len(PGENO_MAGIC) + 4 + hl
        self.done = False
        self._rows = None                 # (first, last + 1) global row of this reader's share (shard()); None = everything
        self._g = 0                       # global row of the next block

    def read_header(self):
        return ("#CHROM\tPOS\t" + "\t".join(self.names) + "\n").encode()

    def input_size(self):
        return os.path.getsize(self.path)

    def _index(self):
        """[(file offset, first global row, n_rows, run starts, run names)] of every block, read from the block headers alone
        (payloads are skipped), and the file position restored"""
        f, here, out, g = self.f, self.f.tell(), [], 0
        while True:
            off = f.tell()
            raw = f.read(8)
            n = int.from_bytes(raw, "little") if len(raw) == 8 else 0
            if n == 0:
                break
            n_runs = int.from_bytes(f.read(4), "little")
            starts, names = [], []
            for _ in range(n_runs):
                starts.append(int.from_bytes(f.read(8), "little"))
                ln = int.from_bytes(f.read(2), "little")
                names.append(f.read(ln).decode())
            if self.codec == "none":
                f.seek(self.pos_bytes * n + n * self.n_cols, 1)
            else:
                n_chunks = int.from_bytes(f.read(4), "little")
                table = np.frombuffer(f.read(8 * n_chunks), dtype="<u4").reshape(-1, 2)
                f.seek(int(table[:, 0].sum()), 1)
            out.append((off, g, n, starts, names))
            g += n
        f.seek(here)
        return out, g

    def shard(self, world, comm, wanted, max_share=0.75):
        """Restrict this reader to rank `world.rank`'s rows: the row range between the scaffold-run boundaries (both neighbours
        wanted) nearest to the equal split, found from the block headers alone -- every rank computes the same plan, nothing is
        exchanged.  Blocks that straddle a cut are inflated by both neighbours and trimmed.  False (reader untouched) when some
        rank would hold more than max_share of the rows."""
        blocks, total = self._index()
        if total == 0:
            return False
        run_row, run_name = [], []                       # scaffold runs of the whole file
        for _, g, _, starts, names in blocks:
            for s_, n_ in zip(starts, names):
                if run_name and run_name[-1] == n_ and s_ == 0:
                    continue                             # the run continues across the block seam
                run_row.append(g + s_)
                run_name.append(n_)
        ok = [run_row[k] for k in range(1, len(run_row)) if wanted(run_name[k - 1]) and wanted(run_name[k])]
        cuts = [0]
        for r in range(1, world.size):
            stride = total // world.size
            guess = stride * r - min(max(stride // 64, 1), stride // 2)
            nxt = [c for c in ok if c >= guess]
            cuts.append(max(nxt[0] if nxt else total, cuts[-1]))
        cuts.append(total)
        if max(cuts[r + 1] - cuts[r] for r in range(world.size)) / total > max_share:
            return False
        self.restrict_rows(blocks, cuts[world.rank], cuts[world.rank + 1])
        return True

    def shard_lines(self, world):
        """rank r's share of the rows, cut anywhere (sites are independent: freq.py, `distMat.py --windType cat`); every rank reads
        the block headers and computes the same equal split"""
        blocks, total = self._index()
        self.restrict_rows(blocks, total * world.rank // world.size, total * (world.rank + 1) // world.size)
        return True

    def restrict_rows(self, blocks, a, b):
        """read only the global rows [a, b) from now on (blocks: _index()); blocks that straddle an end are trimmed"""
        self._rows = (int(a), int(b))
        first = [blk for blk in blocks if blk[1] + blk[2] > self._rows[0]]
        if first and self._rows[1] > self._rows[0]:
            self.f.seek(first[0][0])
            self._g = first[0][1]
        else:
            self.done = True

    def positions(self, blocks, a, b):
        """positions of the global rows [a, b): the position arrays are the first pos_bytes * n_rows bytes of the payloads of the blocks that
        overlap the range (the deflated chunks that hold them are inflated, nothing else is read); the file position is restored"""
        import zlib
        f, here, out = self.f, self.f.tell(), []
        cache = self.__dict__.setdefault("_pos_cache", {})
        for off, g, n, starts, names in blocks:
            if g + n <= a or g >= b:
                continue
            if off in cache:
                out.append(cache[off][max(a - g, 0):min(b - g, n)])
                continue
            f.seek(off + 12 + sum(10 + len(nm.encode()) for nm in names))
            if self.codec == "none":
                pos = np.frombuffer(f.read(self.pos_bytes * n), dtype="<i%d" % self.pos_bytes).astype(np.int64)
            else:
                n_chunks = int.from_bytes(f.read(4), "little")
                table = np.frombuffer(f.read(8 * n_chunks), dtype="<u4").reshape(-1, 2)
                raw, k = b"", 0
                while len(raw) < self.pos_bytes * n:
                    raw += zlib.decompress(f.read(int(table[k, 0])))
                    k += 1
                pos = np.frombuffer(raw[:self.pos_bytes * n], dtype="<i%d" % self.pos_bytes).astype(np.int64)
            cache[off] = pos
            out.append(pos[max(a - g, 0):min(b - g, n)])
        f.seek(here)
        return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)

    def _one(self):
        raw = self.f.read(8)
        n = int.from_bytes(raw, "little") if len(raw) == 8 else 0
        if n == 0:
            self.done = True
            return None
        n_runs = int.from_bytes(self.f.read(4), "little")
        starts, names = [], []
        for _ in range(n_runs):
            starts.append(int.from_bytes(self.f.read(8), "little"))
            ln = int.from_bytes(self.f.read(2), "little")
            names.append(self.f.read(ln).decode())
        want = self.pos_bytes * n + n * self.n_cols
        starts = np.asarray(starts, dtype=np.int64)
        if self.codec == "none":
            at = self.f.tell()
            if at + want > self._size:
                raise ValueError("truncated .pgeno file")
            self.f.seek(want, 1)                              # the payload stays where it is: whoever needs it reads it from there
            self.bytes_read += 12 + want
            blk = _PackedBlock(starts, names, n, self.n_cols, fd=self.f.fileno(), pos_off=at, cells_off=at + self.pos_bytes * n,
                               pos_bytes=self.pos_bytes)
        else:
            n_chunks = int.from_bytes(self.f.read(4), "little")
            table = np.frombuffer(self.f.read(8 * n_chunks), dtype="<u4").reshape(-1, 2)
            if table.shape[0] != n_chunks or int(table[:, 1].sum()) != want:
                raise ValueError("damaged .pgeno block")
            comp = self.f.read(int(table[:, 0].sum()))               # still deflated: inflated into place by to_geno
            if len(comp) != int(table[:, 0].sum()):
                raise ValueError("truncated .pgeno file")
            self.bytes_read += 16 + 8 * n_chunks + len(comp)
            blk = _PackedBlock(starts, names, n, self.n_cols, comp=comp, table=table, pos_bytes=self.pos_bytes)
        g0, self._g = self._g, self._g + n
        if self._rows is not None:                       # trim the block to this reader's rows
            a, b = max(self._rows[0] - g0, 0), min(self._rows[1] - g0, n)
            if g0 + n >= self._rows[1]:
                self.done = True
            if b <= a:
                return self._one() if not self.done else None
            if a > 0 or b < n:
                blk = blk.trim(a, b)
        return blk

    def read_block(self, nbytes=None):
        """list of raw blocks ([] at the end of the file); nbytes counts text bytes (about 4 per cell) like BlockReader's"""
        out, got = [], 0
        while not self.done and (nbytes is None or 4 * got < nbytes):
            b = self._one()
            if b is None:
                break
            out.append(b)
            got += b.n * b.n_cols
        return out

    def to_geno(self, raw_blocks, layout, n_threads=0, head_rows=0, pitch=None, alloc=None, keep_packed=False):
        """GenoData of the raw blocks.  keep_packed: the cells stay packed (GenoData.packed, gt = uint8 [n][n_cols]) for
        Engine.upload_packed_async, which expands them on the device; otherwise pg_decode_packed scatters the nibbles into slot
        order on host threads (rows of `pitch` bytes when given).  alloc(shape, dtype): allocator of the big arrays."""
        if len(layout.col_ploidy) != self.n_cols:
            raise ValueError("layout was built for %d columns, the file has %d" % (len(layout.col_ploidy), self.n_cols))
        wanted = layout.col_ploidy > 0
        if np.any(layout.col_ploidy[wanted] != self.ploidy[wanted]):
            bad = int(np.flatnonzero(wanted & (layout.col_ploidy != self.ploidy))[0])
            raise ValueError("sample %s was packed with ploidy %d but ploidy %d is requested" % (
                self.names[bad], int(self.ploidy[bad]), int(layout.col_ploidy[bad])))
        n = sum(b.n for b in raw_blocks)
        alloc = alloc or np.zeros
        width = self.n_cols if keep_packed else (int(pitch) if pitch else layout.n_hap)
        gt_full = alloc((head_rows + max(n, 1), width), np.uint8 if keep_packed else np.int8)
        pos_full = alloc((head_rows + max(n, 1),), np.int64)
        gt, pos = gt_full[head_rows:head_rows + n], pos_full[head_rows:head_rows + n]
        starts, names, row = [], [], 0
        L = _lib.lib()
        for blk in raw_blocks:
            k = blk.n
            if keep_packed:
                blk.inflate_into(pos[row:row + k], gt[row:row + k], n_threads)      # deflated chunks -> their final rows
            else:
                cells = np.empty((k, self.n_cols), dtype=np.uint8)
                blk.inflate_into(pos[row:row + k], cells, n_threads)
                check(L.pg_decode_packed(cells, k, self.n_cols, layout.max_ploidy,
                                         np.ascontiguousarray(layout.col_slot), layout.col_ploidy, width,
                                         gt[row:row + k], n_threads))
            for s_, n_ in zip(blk.starts, blk.names):
                if names and names[-1] == n_ and int(s_) == 0:
                    continue                                     # the run continues across the block seam
                starts.append(row + int(s_))
                names.append(n_)
            row += k
        d = GenoData(gt, pos, np.asarray(starts, dtype=np.int64), names, packed=keep_packed)
        d.spare = (gt_full, pos_full, head_rows)
        return d

    def close(self):
        self.f.close()


def pack_geno(src_path, dst_path, fmt, ploidy_of=None, header_line=None, block_bytes=256 << 20, codec="zlib"):
    """Tokenise a `.geno(.gz)` file once and keep the result (tools/geno_pack.py).  ploidy_of: {sample: 1|2} overrides of the
    format's default (2, or 1 for `haplo`)."""
    from .samples import HapLayout, SampleData
    rd = BlockReader(src_path)
    names = header_line.split()[2:] if header_line else rd.read_header().decode("utf-8", "replace").split()[2:]
    pl = {nm: (1 if fmt == "haplo" else 2) for nm in names}
    pl.update(ploidy_of or {})
    if len(set(names)) != len(names):
        raise ValueError("duplicate sample names in the header")
    if any(pl[nm] not in (1, 2) for nm in names):
        raise ValueError(".pgeno holds ploidy 1 or 2 only")
    lay = HapLayout(SampleData(indNames=list(names), ploidyDict=pl), names, fmt)      # slot order == file order
    wr = PackedWriter(dst_path, names, [pl[nm] for nm in names], codec)
    n_rows = 0
    while True:
        body = rd.read_block(block_bytes)
        if not body:
            break
        d = encode(body, lay)
        cells = np.zeros((d.n_sites, len(names)), dtype=np.uint8)
        for c, nm in enumerate(names):
            sl = lay.col_slot[c]
            cells[:, c] = d.gt[:, sl[0]].view(np.uint8)
            if lay.col_ploidy[c] > 1:
                cells[:, c] |= d.gt[:, sl[1]].view(np.uint8) << 4
        wr.write_block(d, cells)
        n_rows += d.n_sites
    wr.close()
    rd.close()
    return n_rows


def open_input(path):
    """BlockReader for text, PackedReader for `.pgeno`."""
    if path is not None and str(path).endswith(".pgeno"):
        return PackedReader(path)
    return BlockReader(path)


def read_header_names(path):
    """Sample names of the file's first line (popgenWindows.py:284-286, distMat.py:205-206)."""
    if str(path).endswith(".pgeno"):
        rd = PackedReader(path)
        rd.close()
        return list(rd.names)
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as f:
        return f.readline().split()[2:]


def first_row_ploidy(path, fmt, header_line=None):
    """{sample name: ploidy its cell in the first data row implies} (for --inferPloidy); of a piped input the lines up to the first
    data row are read and pushed back"""
    def from_widths(names, line):
        w = [len(c) for c in line.split()[2:]]
        # splitSeq (genomics.py:390-396): phased cells hold their alleles at every other character, pairs one per character
        return {nm: ((x + 1) // 2 if fmt == "phased" else x if fmt == "pairs" else 1 if fmt == "haplo" else 2) for nm, x in zip(names, w)}
    if path is None:
        seen, names = b"", (header_line.split()[2:] if header_line else None)
        try:
            while True:
                raw = STDIN.readline()
                seen += raw
                if not raw:
                    return {nm: (1 if fmt == "haplo" else 2) for nm in (names or [])}
                line = raw.decode("utf-8", "replace")
                if names is None:
                    names = line.split()[2:]
                elif line.strip() and not line.startswith("#"):
                    return from_widths(names, line)
        finally:
            STDIN.unread(seen)
    if str(path).endswith(".pgeno"):
        rd = PackedReader(path)
        rd.close()
        return {nm: int(p) for nm, p in zip(rd.names, rd.ploidy)}
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as f:
        names = header_line.split()[2:] if header_line else f.readline().split()[2:]
        for line in f:
            if line.strip() and not line.startswith("#"):
                return from_widths(names, line)
    return {nm: (1 if fmt == "haplo" else 2) for nm in names}


class PloidySegments:
    """--inferPloidy on a file whose cell widths change: `starts[k]` = index (among the file's data rows) of the first row of
    segment k, `ploidy[k][i]` = what the cells of individual `inds[i]` hold there.  The reference infers the ploidy of a sample
    per window, from the shortest cell the window holds (genoToAlignment with ploidy None, genomics.py:1108-1111; splitSeq zips
    the cells, 390-396): `window_ploidy(a, b)` = the minimum over the segments that rows [a, b) touch."""

    def __init__(self, inds, starts, ploidy):
        self.inds, self.starts, self.ploidy = list(inds), np.asarray(starts, dtype=np.int64), np.asarray(ploidy, dtype=np.int32)

    def max_ploidy(self):
        return {nm: int(v) for nm, v in zip(self.inds, self.ploidy.max(axis=0))}

    def window_ploidy(self, a, b):
        """int32 [n_windows][n_inds] for the global data-row ranges [a[w], b[w]) (an empty range: the segment row a[w] would be in)"""
        a, b = np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64)
        s0 = np.maximum(np.searchsorted(self.starts, a, side="right") - 1, 0)
        s1 = np.maximum(np.searchsorted(self.starts, np.maximum(b - 1, a), side="right") - 1, 0)
        out = self.ploidy[s0].copy()
        for w in np.flatnonzero(s1 > s0):
            out[w] = self.ploidy[s0[w]:s1[w] + 1].min(axis=0)
        return out


def scan_ploidy_segments(path, fmt, inds, header_line=None, block_bytes=256 << 20):
    """PloidySegments of a text input (plain, gzip or BGZF) in the phased / pairs formats: one pass of pg_text_cell_widths over the
    whole file (host threads; the widths of the other columns are not looked at)."""
    rd = BlockReader(path)
    try:
        names = header_line.split()[2:] if header_line else rd.read_header().decode("utf-8", "replace").split()[2:]
        col = {}
        for k, nm in enumerate(names):
            col.setdefault(nm, k)
        for nm in inds:
            if nm not in col:
                raise AssertionError("sample %s is not in the genotype file header" % nm)
        n_cols = len(names)
        watch = np.zeros(n_cols, dtype=np.int32)
        watch[[col[nm] for nm in inds]] = 1
        state = np.full(n_cols, -1, dtype=np.int32)
        L = _lib.lib()
        starts, widths, rows = [], [], 0
        while True:
            body = rd.read_block(block_bytes)
            if len(body) == 0:
                break
            ptr, nbytes, _keep = _lib.text_ptr(body)
            cap = 64
            while True:
                at, w = np.zeros(cap, dtype=np.int64), np.zeros((cap, max(n_cols, 1)), dtype=np.int32)
                n, nr = C.c_int64(0), C.c_int64(0)
                check(L.pg_text_cell_widths(ptr, nbytes, n_cols, watch, state, at, w, cap, C.byref(n), C.byref(nr)))
                if n.value <= cap:
                    break
                cap = int(n.value)
            for k in range(int(n.value)):
                starts.append(rows + int(at[k]))
                widths.append(w[k, [col[nm] for nm in inds]].copy())
            rows += int(nr.value)
            del body
    finally:
        rd.close()
    if not starts:                                           # no data row at all
        return PloidySegments(inds, [0], [[1 if fmt == "haplo" else 2] * len(inds)])
    widths = np.array(widths, dtype=np.int32)
    # splitSeq (genomics.py:390-396): phased cells hold their alleles at every other character, pairs one per character
    ploidy = (widths + 1) // 2 if fmt == "phased" else widths
    keep = [0] + [k for k in range(1, len(starts)) if not np.array_equal(ploidy[k], ploidy[k - 1])]
    return PloidySegments(inds, [starts[k] for k in keep], ploidy[keep])


def split_header(data, header_line=None):
    """(sample names, data bytes after the header).  With --header the file has no header line
    (GenoFileReader.__init__, genomics.py:1917-1919)."""
    if header_line:
        return header_line.split()[2:], data
    nl = data.find(b"\n")
    first = data if nl < 0 else data[:nl]
    rest = b"" if nl < 0 else data[nl + 1:]
    return first.decode("utf-8", "replace").split()[2:], rest


class GenoData:
    """Encoded input: one-hot int8 genotypes in device slot order, positions and scaffold runs."""

    def __init__(self, gt, pos, run_starts, run_names, packed=False):
        self.gt, self.pos, self.run_starts, self.run_names = gt, pos, run_starts, run_names
        self.n_sites = len(pos)
        self.packed = packed              # gt holds packed `.pgeno` cells (uint8 [n][n_cols]) instead of slot-order codes
        self.spare = None                 # (gt_full, pos_full, head_rows): the arrays gt / pos are views of, with spare rows in front


def concat(a, b):
    """Rows of GenoData a followed by those of b (a run that continues across the seam is merged).  When b was encoded
    with head_rows = a.n_sites the rows of a are copied into the spare rows in front of b's arrays (no copy of b)."""
    if a is None or a.n_sites == 0:
        return b
    if b.n_sites == 0:
        return a
    merge = a.run_names[-1] == b.run_names[0]
    starts = np.concatenate([a.run_starts, (b.run_starts[1:] if merge else b.run_starts) + a.n_sites]).astype(np.int64)
    names = list(a.run_names) + list(b.run_names[1:] if merge else b.run_names)
    n = a.n_sites
    if b.spare is not None and b.spare[2] >= n:
        gfull, pfull, head = b.spare
        gfull[head - n:head] = a.gt
        pfull[head - n:head] = a.pos
        d = GenoData(gfull[head - n:head + b.n_sites], pfull[head - n:head + b.n_sites], starts, names, b.packed)
        d.spare = (gfull, pfull, head - n)
        return d
    return GenoData(np.concatenate([a.gt, b.gt]), np.concatenate([a.pos, b.pos]), starts, names, b.packed)


def tail(d, keep_from):
    """Rows [keep_from, n) of GenoData d."""
    if keep_from >= d.n_sites:
        return None
    r = int(np.searchsorted(d.run_starts, keep_from, side="right")) - 1
    starts = np.concatenate([[0], d.run_starts[r + 1:] - keep_from]).astype(np.int64)
    return GenoData(d.gt[keep_from:].copy(), d.pos[keep_from:].copy(), starts, list(d.run_names[r:]), d.packed)


def concat_meta(a, b):
    """concat() of the positions and scaffold runs only (rows that live on the device: gt is None)"""
    if a is None or a.n_sites == 0:
        return b
    if b.n_sites == 0:
        return a
    merge = a.run_names[-1] == b.run_names[0]
    starts = np.concatenate([a.run_starts, (b.run_starts[1:] if merge else b.run_starts) + a.n_sites]).astype(np.int64)
    names = list(a.run_names) + list(b.run_names[1:] if merge else b.run_names)
    return GenoData(None, np.concatenate([a.pos, b.pos]), starts, names)


def tail_meta(d, keep_from):
    """tail() of the positions and scaffold runs only"""
    if keep_from >= d.n_sites:
        return None
    r = int(np.searchsorted(d.run_starts, keep_from, side="right")) - 1
    starts = np.concatenate([[0], d.run_starts[r + 1:] - keep_from]).astype(np.int64)
    return GenoData(None, d.pos[keep_from:].copy(), starts, list(d.run_names[r:]))


def encode(data, layout, n_threads=0, head_rows=0, pitch=None, alloc=None, narrow_ok=False):
    full = []
    gt, pos, soff, slen = encode_text(data, layout, n_threads, head_rows, pitch, alloc, full, narrow_ok)
    n = len(pos)
    L = _lib.lib()
    ptr, _, _keep = _lib.text_ptr(data)
    cap = 1024
    while True:
        starts = np.zeros(cap, dtype=np.int64)
        nr = C.c_int64(0)
        rc = L.pg_scaffold_runs(ptr, soff, slen, n, starts, cap, C.byref(nr))
        if nr.value > cap:
            cap = int(nr.value)
            continue
        check(rc)
        break
    starts = starts[:nr.value]
    names = [bytes(data[int(soff[i]):int(soff[i]) + int(slen[i])]).decode("utf-8", "replace") for i in starts]
    d = GenoData(gt, pos, starts, names)
    d.spare = (full[0][0], full[0][1], head_rows)
    return d
