"""The DEFLATE decoder the GPU runs (genomics_general_amd/csrc/pg_inflate_core.h: one wavefront per BGZF member) held against zlib
WITHOUT a GPU: tests/inflate_emul.cpp compiles the very same source as a lockstep emulation of its 64 lanes (every per-lane statement
of the kernel is one memory operation, executed for lane 0 .. 63 in turn -- the order the hardware gives them).  So an error in the
bit reader, the canonical Huffman decoding by limits, the table builder, the LDS ring, the flush in aligned pieces or the copy of
overlapping matches shows up on every machine; what the GPU adds (tests/test_gpu_inflate.py) is the real thing at scale.

Also here: the host side of the BGZF route -- pg_bgzf_walk, pg_inflate_members, pg_bgzf_compress, genoio.BgzfFile.read_span."""
import ctypes as C
import gzip
import os
import random
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

from genomics_general_amd import genoio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    d = tmp_path_factory.mktemp("emul")
    so = str(d / "libinflate_emul.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "inflate_emul.cpp"), "-o", so])
    L = C.CDLL(so)
    L.pgi_emul_inflate_at.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int]

    L.pgi_emul_nl_setup.argtypes = [C.c_uint32, C.c_uint32]
    L.pgi_emul_nl_result.argtypes = [C.POINTER(C.c_uint16), C.c_uint32]
    L.pgi_emul_nl_result.restype = C.c_uint32

    L.pgi_emul_crc_setup.argtypes = [C.c_int, C.c_uint32]

    def run(stream, out_len, pre=0, post=0, misalign=0, rng=random, nl=None, crc=None):
        """nl = (capacity, limit): also the list of line feeds the decoder keeps beside the text -> (rc, text, count, offsets);
        crc: the CRC-32 the decoder holds the text against while it flushes it (None: no check)"""
        comp = bytes(rng.randrange(256) for _ in range(pre)) + stream + bytes(rng.randrange(256) for _ in range(post))
        dst = C.create_string_buffer(max(out_len, 1) + 8)
        L.pgi_emul_nl_setup(*(nl if nl else (0, 0xFFFFFFFF)))
        L.pgi_emul_crc_setup(0 if crc is None else 1, 0 if crc is None else crc & 0xFFFFFFFF)
        rc = L.pgi_emul_inflate_at(comp, len(comp), pre, len(stream), dst, out_len, misalign)
        if nl:
            buf = (C.c_uint16 * max(nl[0], 1))()
            cnt = L.pgi_emul_nl_result(buf, nl[0])
            return rc, dst.raw[:out_len], cnt, list(buf[:min(cnt, nl[0])])
        return rc, dst.raw[:out_len]
    return run


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    return c.compress(data) + c.flush()


def geno_text(rng, n, ns):
    return ("\n".join("scaf%d\t%d\t" % (i // 1000, i * 37 + 1) + "\t".join(rng.choice(["A/A", "A/T", "T/T", "N/N", "A/A", "A/A"]) for _ in range(ns))
                      for i in range(n)) + "\n").encode()


def test_the_kernel_source_inflates_what_zlib_deflates(emul):
    """every block type (stored, fixed, dynamic), levels 0 - 9, five strategies, two memory levels (memLevel 1: dozens of blocks per
    member), outputs of 0 .. 70 000 bytes, matches of every distance from 1 to 69 (the pattern fill below the wavefront's width, the
    stepwise copy above it), output addresses at every misalignment of the 16-byte flush"""
    rng = random.Random(1)
    cases = [b"", b"a", b"abcabcabcabcabcabcabcabc" * 50, b"\0" * 70000, bytes(rng.randrange(256) for _ in range(65280)),
             bytes(rng.randrange(4) for _ in range(65280)), geno_text(rng, 300, 50)[:65280], geno_text(rng, 80, 200)[:65280]]
    for n in (1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 257, 258, 259, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097):
        cases += [b"x" * n, bytes(rng.randrange(256) for _ in range(n)), (b"ab" * n)[:n]]
    for p in range(1, 70):
        pat = bytes(rng.randrange(256) for _ in range(p))
        cases.append((pat * 400)[:3000 + p])
    n = 0
    for data in cases:
        for level in (0, 1, 4, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                ml = 8 if n % 3 else 1
                rc, out = emul(deflate(data, level, strat, ml), len(data), pre=n % 9, post=n % 5, misalign=n % 16, rng=rng,
                               crc=zlib.crc32(data) if n % 4 else None)
                n += 1
                assert rc == 0 and out == data, (len(data), level, strat, ml, rc)
    assert n > 3500


def test_the_decoder_lists_the_line_feeds_of_the_text_it_writes(emul):
    """round 6: k_inflate reports the offsets of a member's line feeds, found in the registers of its flushes (head bytes, aligned
    1 KiB pieces, tail bytes), so that the tokenizer needs no pass over the text for them: every offset, in order, for every
    misalignment of the output; a limit (the block's text ends inside the member); a list that is too short still counts them all"""
    rng = random.Random(5)
    texts = [geno_text(rng, 300, 50)[:65280], geno_text(rng, 70, 200)[:60001], b"\n" * 3000, b"a\n" * 2000 + b"tail without a line feed",
             b"\n", b"", b"x" * 5000, bytes(rng.choice(b"\nab") for _ in range(20000))]
    n = 0
    for data in texts:
        want = [i for i, ch in enumerate(data) if ch == 10]
        for mis in range(16):
            for level in (0, 1, 6):
                rc, out, cnt, offs = emul(deflate(data, level), len(data), misalign=mis, rng=rng, nl=(70000, 0xFFFFFFFF))
                assert rc == 0 and out == data and cnt == len(want) and offs == want, (len(data), mis, level)
                n += 1
        for lim in (0, 1, len(data) // 2, len(data) - 1, len(data), len(data) + 5):
            lim = max(lim, 0)
            rc, out, cnt, offs = emul(deflate(data), len(data), misalign=lim % 16, rng=rng, nl=(70000, lim))
            assert rc == 0 and offs == [i for i in want if i < lim] and cnt == len(offs)
        rc, out, cnt, offs = emul(deflate(data), len(data), misalign=3, rng=rng, nl=(5, 0xFFFFFFFF))
        assert rc == 0 and out == data and cnt == len(want) and offs == want[:5]
    assert n > 300


def test_the_decoder_checksums_the_text_it_writes(emul):
    """round 6: the member's CRC-32 is taken in the flush (lane i checksums bytes 16 i .. 16 i + 15 of every aligned 1 KiB piece; head
    and last bytes through the byte table; the lanes' registers moved to the end of the text and XORed) instead of by a kernel that
    reads the text again: every size around the piece and chunk boundaries at every misalignment; a wrong CRC is PGI_ERR_CRC (64)"""
    rng = random.Random(9)
    n = 0
    sizes = list(range(0, 70)) + [1023, 1024, 1025, 1039, 1040, 1041, 2047, 2048, 2049, 2063, 2064, 2065, 3071, 3072, 3073, 4095, 4096, 4111, 4112,
                                  5000, 20000, 65279, 65280]
    for size in sizes:
        data = bytes(rng.choice(b"ACGT/\t\nN") for _ in range(size))
        good = zlib.crc32(data)
        for mis in range(16):
            rc, out = emul(deflate(data, 6 if size % 2 else 0), len(data), misalign=mis, rng=rng, crc=good)
            assert rc == 0 and out == data, (size, mis, rc)
            rc, out = emul(deflate(data), len(data), misalign=mis, rng=rng, crc=good ^ (1 << (n % 32)))
            assert rc == 64, (size, mis, rc)
            n += 1
    assert n > 1400


def test_matches_that_reach_behind_the_lds_ring_read_global_memory(emul):
    """distances beyond the 4 KiB ring (zlib looks back up to 32 KiB): the source is text the wavefront has already flushed"""
    rng = random.Random(2)
    block = bytes(rng.randrange(256) for _ in range(300))
    for gap in (3700, 3776, 3777, 4000, 4096, 4097, 5000, 9000, 20000, 32000):
        data = block + bytes(rng.randrange(256) for _ in range(gap - 300)) + block + b"tail" + block[:100]
        for mis in (0, 5, 15):
            rc, out = emul(deflate(data, 9), len(data), misalign=mis, rng=rng)
            assert rc == 0 and out == data, (gap, mis, rc)


def test_damaged_streams_end_in_an_error_and_touch_nothing_outside_their_output(emul):
    """flipped bits and truncations: whenever zlib accepts the stream with the same output size the emulation gives zlib's bytes, otherwise
    it reports an error; the bytes in front of and behind the member's output stay untouched (inflate_emul.cpp checks canaries)"""
    rng = random.Random(7)
    base = [geno_text(rng, 40, 50), bytes(rng.randrange(256) for _ in range(3000)), b"abc" * 1000, bytes(rng.randrange(3) for _ in range(5000))]
    agree = err = 0
    for it in range(2500):
        data = rng.choice(base)
        raw = bytearray(deflate(data, rng.choice([1, 6, 9]), rng.choice([0, 0, zlib.Z_FIXED, zlib.Z_RLE])))
        for _ in range(rng.choice([1, 1, 2, 5])):
            raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        if rng.random() < 0.1:
            raw = raw[:rng.randrange(len(raw))]
        raw = bytes(raw)
        try:
            d = zlib.decompressobj(-15)
            ref = d.decompress(raw)
            ok = d.eof and len(d.unused_data) == 0 and len(ref) == len(data)
        except zlib.error:
            ref, ok = None, False
        rc, out = emul(raw, len(data), misalign=it % 16, rng=rng)
        assert rc < (1 << 20), "the decoder wrote outside its output"
        if ok:
            assert rc == 0 and out == ref
            agree += 1
        else:
            assert rc != 0
            err += 1
    assert agree > 500 and err > 500


# ---- host side --------------------------------------------------------------------------------------------------------------------
def test_bgzf_compress_walk_and_host_inflate_round_trip(tmp_path):
    rng = random.Random(3)
    text = geno_text(rng, 3000, 60)
    bz = genoio.bgzf_compress(text, block=7000)
    assert gzip.decompress(bz.tobytes()) == text                          # a valid multi-member gzip file for everybody else
    tab, used, n_text = genoio.bgzf_walk(bz)
    assert used == len(bz) and n_text == len(text) and int(tab[2][-1]) == 0          # the EOF member
    assert genoio.bgzf_inflate(bz, tab).tobytes() == text
    # the walk stops in front of an incomplete member, and at the text it was asked for
    tab2, used2, _ = genoio.bgzf_walk(bz[:len(bz) - 40])
    assert len(tab2[0]) == len(tab[0]) - 2 and used2 < len(bz) - 40
    tab3, _, text3 = genoio.bgzf_walk(bz, None, 20000)
    assert 20000 <= text3 < 20000 + 7000 and len(tab3[0]) == 3
    # other extra subfields in front of BC, a file name: legal gzip
    comp = deflate(b"hello\n")
    m = (b"\x1f\x8b\x08\x0c\0\0\0\0\0\xff" + struct.pack("<H", 13) + b"XY\x03\x00abc" + b"BC\x02\x00" + struct.pack("<H", 12 + 13 + 2 + len(comp) + 8 - 1) +
         b"a\0" + comp + struct.pack("<II", zlib.crc32(b"hello\n"), 6))
    tab4, used4, _ = genoio.bgzf_walk(m)
    assert used4 == len(m) and genoio.bgzf_inflate(m, tab4).tobytes() == b"hello\n"
    # damage: the host pool names it
    bad = bytearray(bz.tobytes())
    bad[int(tab[0][2]) + 30] ^= 0x20
    with pytest.raises(ValueError, match="damaged BGZF member"):
        genoio.bgzf_inflate(bytes(bad), tab)
    with pytest.raises(ValueError, match="stops being BGZF"):
        genoio.bgzf_walk(b"\x1f\x8b\x08\x00" + bytes(30))


def test_text_that_does_not_deflate_is_stored_and_a_failed_writer_leaves_no_eof_member(tmp_path):
    """(i) a 65280-byte block of random bytes deflates to more than 64 KiB - 26: pg_bgzf_compress stores it (what
    bgzip's block size is made for) instead of failing the call; (ii) BgzfWriter.abort(): a truncated output must not end in the
    end-of-file member"""
    rng = random.Random(11)
    noise = bytes(rng.getrandbits(8) for _ in range(65280 * 2 + 1000)) + b"A/A\tA/T\n" * 20000
    bz = genoio.bgzf_compress(noise)
    assert gzip.decompress(bz.tobytes()) == noise
    tab, used, n_text = genoio.bgzf_walk(bz)
    assert used == len(bz) and n_text == len(noise) and int(tab[1].max()) <= 65536 - 26 and int(tab[1][0]) in (65285, 65300)      # (zlib itself falls back to stored blocks: four of them here)
    assert genoio.bgzf_inflate(bz, tab).tobytes() == noise
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    path = str(tmp_path / "x.gz")
    w = genoio.BgzfWriter(path)
    w.write(b"chr1\t1\tA/A\n" * 10)
    w.close()
    assert open(path, "rb").read().endswith(eof)
    w = genoio.BgzfWriter(path)
    w.write((noise * 60)[:genoio.BgzfWriter.PIECE + 5])
    w.abort()
    w.close()                                                                 # (idempotent after abort)
    raw = open(path, "rb").read()
    assert len(raw) > 0 and not raw.endswith(eof)
    with pytest.raises(RuntimeError):
        with genoio.BgzfWriter(path) as w:
            w.write(b"chr1\t1\tA/A\n")
            raise RuntimeError("the run failed")
    assert open(path, "rb").read() == b""


@pytest.mark.parametrize("blk,want", [(900, 4000), (5000, 30000), (65280, 100000)])
def test_read_span_cuts_blocks_behind_their_last_line_feed(blk, want, tmp_path):
    """BgzfFile.read_span: blocks of deflated members whose text = head + members' text, cut behind the last line feed; what follows
    is the head of the next block; every byte of the input exactly once; first_line is the block's first line"""
    rng = random.Random(blk)
    text = b"#CHROM\tPOS\t" + b"\t".join(b"s%d" % i for i in range(30)) + b"\n" + geno_text(rng, 1500, 30)
    path = str(tmp_path / "x.geno.gz")
    with open(path, "wb") as f:
        f.write(genoio.bgzf_compress(text, block=blk).tobytes())
    rd = genoio.BlockReader(path)
    rd.spans = True
    parts = [rd.read_header()]
    n_spans = 0
    while True:
        b = rd.read_block(want)
        if not len(b):
            break
        if isinstance(b, genoio.BgzfSpan):
            n_spans += 1
            body = bytes(b)
            assert body.endswith(b"\n") and body.split(b"\n", 1)[0] == b.first_line and len(body) == len(b)
            assert bytes(b[:1 << 16]).startswith(b.first_line + b"\n"[:1])
            parts.append(body)
        else:
            parts.append(bytes(b))
    rd.close()
    assert b"".join(parts) == text and n_spans >= 2


def test_a_line_longer_than_the_members_of_a_block_is_handed_on_as_text(tmp_path):
    """lines of 300 kB in members of 20 kB: no member of a block ends a line -- the reader inflates on the host and still delivers whole lines"""
    line = b"chr1\t1\t" + b"\t".join([b"A/T"] * 75000) + b"\n"
    text = b"#h\n" + line + line.replace(b"\t1\t", b"\t2\t") + b"chr1\t3\tA/A\n"
    path = str(tmp_path / "long.geno.gz")
    with open(path, "wb") as f:
        f.write(genoio.bgzf_compress(text, block=20000).tobytes())
    rd = genoio.BlockReader(path)
    rd.spans = True
    got = [rd.read_header()]
    while True:
        b = rd.read_block(100000)
        if not len(b):
            break
        got.append(bytes(b))
        assert got[-1].endswith(b"\n")
    assert b"".join(got) == text


def test_one_gzip_stream_through_the_native_reader(tmp_path):
    """`gzip file.geno` (README.md:106 of the reference; gzip.open there): ONE deflate stream, read by zlib inside the library
    (pg_gzip_read_lines) straight into the block's buffer.  Blocks end at line feeds whatever size is asked for, concatenated members
    (`cat a.gz b.gz`) and an empty member are followed, a last line without a line feed comes through, damage and truncation are
    named."""
    rng = random.Random(4)

    def lines(n):
        return b"".join(b"chr1\t%d\t" % i + b"\t".join(rng.choice([b"A/A", b"A/T", b"T/T"]) for _ in range(rng.randrange(1, 40))) + b"\n"
                        for i in range(n))
    cases = [b"", b"x", b"no newline at end", lines(5), lines(3000), lines(200) + b"tail", b"\n" * 1000, b"a" * 3000000 + b"\n" + lines(10)]
    for k, data in enumerate(cases):
        for multi in (False, True):
            p = str(tmp_path / ("t%d.gz" % k))
            with open(p, "wb") as f:
                if multi and len(data) > 10:
                    h = len(data) // 3
                    f.write(gzip.compress(data[:h]) + gzip.compress(b"") + gzip.compress(data[h:]))
                else:
                    f.write(gzip.compress(data))
            for want in (1, 7, 4096, 70000, None):
                rd = genoio.GzipStream(p)
                got = b""
                while True:
                    b = bytes(rd.read_lines(want))
                    if not b:
                        break
                    assert b.endswith(b"\n") or got + b == data
                    assert want is None or got + b == data or len(b) >= want
                    got += b
                assert got == data, (k, multi, want)
                rd.close()
    # the drivers' reader: header line, then blocks
    p = str(tmp_path / "g.geno.gz")
    text = b"#CHROM\tPOS\ta\tb\n" + lines(4000)
    with open(p, "wb") as f:
        f.write(gzip.compress(text))
    rd = genoio.BlockReader(p)
    assert isinstance(rd.f, genoio.GzipStream) and rd.read_header() == b"#CHROM\tPOS\ta\tb\n"
    body = b""
    while True:
        b = rd.read_block(30000)
        if len(b) == 0:
            break
        body += bytes(b)
    assert body == text[len(b"#CHROM\tPOS\ta\tb\n"):]
    rd.close()
    raw = bytearray(gzip.compress(lines(3000)))
    raw[len(raw) // 2] ^= 0x55
    for blob, what in ((bytes(raw), "invalid deflate data"), (gzip.compress(lines(3000))[:-300], "ends inside a member")):
        with open(p, "wb") as f:
            f.write(blob)
        with pytest.raises(Exception, match=what):
            rd = genoio.GzipStream(p)
            while bytes(rd.read_lines(1000)):
                pass


def test_the_fast_host_decoder_against_zlib(tmp_path):
    """genomics_general_amd/csrc/pg_fast_inflate.h (the decoder behind GzipStream: one long stream, 11-bit tables, eight-byte
    refills and copies, resumable at any output byte): every block type, level, strategy and two memory levels (memLevel 1: many
    short blocks), outputs that straddle the caller's buffers at odd sizes (the saved 32 KiB window, the rest of a match carried
    over), patterns of every short period; then damaged and truncated streams: refused, or exactly zlib's bytes -- never a crash
    (`make asan-test` runs this file under AddressSanitizer)."""
    rng = random.Random(11)
    cases = [b"", b"a", b"abc" * 1000, bytes(rng.randrange(256) for _ in range(60000)), bytes(rng.randrange(4) for _ in range(120000)),
             geno_text(rng, 2000, 50), geno_text(rng, 300, 400), b"\0" * 200000]
    for p in range(1, 24):
        pat = bytes(rng.randrange(256) for _ in range(p))
        cases.append((pat * 3000)[:40000 + p])
    path = str(tmp_path / "f.gz")
    n = 0
    for data in cases:
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                ml = 8 if n % 2 else 1
                c = zlib.compressobj(level, zlib.DEFLATED, 31, ml, strat)
                with open(path, "wb") as f:
                    f.write(c.compress(data) + c.flush())
                for want in (1 << 30, 1000, 33333):
                    rd = genoio.GzipStream(path)
                    assert rd._L.pg_gzip_open is not None
                    got = b""
                    while True:
                        b = bytes(rd.read_lines(want))
                        if not b:
                            break
                        got += b
                    rd.close()
                    assert got == data, (len(data), level, strat, ml, want)
                    n += 1
    assert n > 1400
    base = geno_text(rng, 1500, 60)
    blob0 = gzip.compress(base, 6)
    refused = 0
    for it in range(600):
        raw = bytearray(blob0)
        k = it % 3
        if k == 0:
            raw[rng.randrange(10, len(raw))] ^= 1 << rng.randrange(8)
        elif k == 1:
            raw = raw[:rng.randrange(len(raw))]
        else:
            a = rng.randrange(10, len(raw))
            m = rng.randrange(1, 8)
            raw[a:a + m] = bytes(rng.randrange(256) for _ in range(m))
        with open(path, "wb") as f:
            f.write(raw)
        try:
            rd = genoio.GzipStream(path)
            got = b""
            while True:
                b = bytes(rd.read_lines(50000))
                if not b:
                    break
                got += b
            rd.close()
        except Exception:
            refused += 1
            continue
        assert got == gzip.decompress(bytes(raw)), it
    assert refused > 500


@pytest.mark.parametrize("chunk,threads", [("20000", "8"), ("4096", "3"), ("300000", "16")])
def test_one_gzip_stream_in_chunks_side_by_side(chunk, threads, tmp_path, monkeypatch):
    """genomics_general_amd/csrc/pg_par_gunzip.h: block starts found by trial (dynamic-block headers that parse and decode), every
    chunk decoded without the 32 KiB in front of it (16-bit output with markers), chunks chained by their bit positions, windows
    handed on, markers replaced, CRC-32 per chunk combined in order.  Tiny chunks here (PG_GZIP_CHUNK), so that a file of a few
    megabytes runs through many batches: levels and strategies whose blocks are stored or fixed-code (no start to be found: the
    serial decoder takes over), concatenated members, a stream that ends inside a batch, damaged streams."""
    monkeypatch.setenv("PG_GZIP_CHUNK", chunk)
    monkeypatch.setenv("PG_GZIP_THREADS", threads)
    rng = random.Random(21)
    big = geno_text(rng, 9000, 120)                                       # 4 MB of text
    path = str(tmp_path / "p.gz")
    n = 0
    for data in (big, big[:700000], bytes(rng.randrange(4) for _ in range(600000)), b"ACGT" * 200000):
        for level, strat in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                             (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY)):
            for ml in (8, 2):
                c = zlib.compressobj(level, zlib.DEFLATED, 31, ml, strat)
                blob = c.compress(data) + c.flush()
                with open(path, "wb") as f:
                    f.write(blob if n % 3 else blob + gzip.compress(data[:1000]) + blob)
                want_text = data if n % 3 else data + data[:1000] + data
                for want in (1 << 30, 250000):
                    rd = genoio.GzipStream(path)
                    got = b""
                    while True:
                        b = bytes(rd.read_lines(want))
                        if not b:
                            break
                        got += b
                    rd.close()
                    assert got == want_text, (len(data), level, strat, ml, want)
                n += 1
    blob0 = gzip.compress(big, 6)
    refused = 0
    for it in range(60):
        raw = bytearray(blob0)
        if it % 2:
            raw[rng.randrange(10, len(raw))] ^= 1 << rng.randrange(8)
        else:
            raw = raw[:rng.randrange(len(raw))]
        with open(path, "wb") as f:
            f.write(raw)
        try:
            rd = genoio.GzipStream(path)
            got = b""
            while True:
                b = bytes(rd.read_lines(1 << 30))
                if not b:
                    break
                got += b
            rd.close()
        except Exception:
            refused += 1
            continue
        assert got == gzip.decompress(bytes(raw)), it
    assert refused >= 55


def test_the_carry_less_crc32_equals_zlibs(tmp_path):
    """csrc/pg_crc32_fast.h (PCLMULQDQ folding where the CPU has it) against zlib's crc32 on every length up to 700 bytes at 17
    alignments and a few long buffers, through a small C++ program (the library itself uses it for gzip trailers)"""
    header = os.path.join(ROOT, "genomics_general_amd", "csrc", "pg_crc32_fast.h")
    prog = """
#include "HEADER"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main() {
    std::vector<uint8_t> b(1 << 20);
    srand(3);
    for (auto &x : b) x = (uint8_t)rand();
    int bad = 0;
    for (size_t off = 0; off < 17; ++off)
        for (size_t len = 0; len < 700; ++len) {
            uint32_t seed = (uint32_t)(len * 2654435761u);
            if (pg_crc32(seed, b.data() + off, len) != (uint32_t)crc32_z(seed, b.data() + off, len)) ++bad;
        }
    for (size_t len : {1000u, 4096u, 65537u, 1000003u})
        if (pg_crc32(0, b.data() + 1, len % (1 << 20)) != (uint32_t)crc32_z(0, b.data() + 1, len % (1 << 20))) ++bad;
    printf("%d", bad);
    return bad != 0;
}
""".replace("HEADER", header)
    src = str(tmp_path / "t.cpp")
    with open(src, "w") as f:
        f.write(prog)
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-lz", "-o", exe])
    assert subprocess.check_output([exe]).strip() == b"0"


def test_the_host_compressor_round_trips_through_zlib_and_the_decoders(tmp_path):
    """csrc/pg_fast_deflate.h (BGZF members at the default level: hash chains of 24, one line back first, one lazy step, one dynamic
    block per member): what it writes must inflate to the text with zlib, with the host decoder (GzipStream takes a BGZF file as
    concatenated members when asked to) and, in the -m gpu suite, with k_inflate; texts that do not deflate are stored; the ratio on
    `.geno` text stays within 15 % of zlib's level 6"""
    rng = random.Random(31)
    texts = [b"", b"a", b"ab" * 7, b"x" * 100000, bytes(rng.randrange(256) for _ in range(70000)), bytes(rng.randrange(3) for _ in range(150000)),
             geno_text(rng, 4000, 60), geno_text(rng, 300, 500), b"\n".join(b"%d,%d,%.4f,nan,0.1234" % (i, i * 50000, i / 7.0) for i in range(20000)),
             bytes(range(256)) * 300]
    for p in (1, 2, 3, 5, 8, 13, 255, 256, 257, 4000, 32767, 32768, 32769, 40000):
        pat = bytes(rng.randrange(256) for _ in range(p))
        texts.append((pat * (200000 // p + 2))[:200000])
    for text in texts:
        for block in (65280, 4097, 65535 if len(text) % 2 else 1000):
            bz = genoio.bgzf_compress(text, block=min(block, 65280))
            assert gzip.decompress(bz.tobytes()) == text, (len(text), block)
            tab, used, n_text = genoio.bgzf_walk(bz)
            assert n_text == len(text) and genoio.bgzf_inflate(bz, tab).tobytes() == text
    g = geno_text(rng, 6000, 100)
    own = len(genoio.bgzf_compress(g))
    z6 = sum(len(zlib.compress(g[a:a + 65280], 6)) + 14 for a in range(0, len(g), 65280))
    assert own <= 1.15 * z6, (own, z6)
