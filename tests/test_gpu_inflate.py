"""-m gpu: `.geno.gz` written by bgzip, inflated ON THE DEVICE (csrc/pg_inflate.hip: k_inflate, a wavefront per BGZF member, +
k_crc32) and handed to the device tokenizer where it lies.

  * the kernel against zlib: every block type (stored, fixed, dynamic), every compression level and strategy, members from 0 bytes
    to 65 280, matches of every distance from 1 to beyond the wavefront, incompressible bytes, damaged members (named, never a crash);
  * every golden whose window type streams, as BGZF with members that end in the middle of lines, an empty member in the middle and
    the EOF member, in blocks of a few kilobytes: byte for byte the reference's output;
  * 2 / 3 / 8 ranks on one device, cutting inside members (shardplan, restrict_virtual)."""
import ctypes as C
import os
import random
import struct
import zlib

import numpy as np
import pytest

from golden_util import align_columns
from genomics_general_amd import _lib, genoio
from genomics_general_amd.engine import Engine

import test_gpu_golden as G
from test_cli_cpu import write_bgzf

pytestmark = pytest.mark.gpu
EOF_MEMBER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8, extra=b"", name=None):
    """one BGZF member (optionally with another extra subfield in front of BC and a file name: both legal gzip)"""
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    comp = c.compress(chunk) + c.flush()
    xlen = 6 + len(extra)
    flg = 4 | (8 if name else 0)
    tail = (name + b"\0") if name else b""
    total = 12 + xlen + len(tail) + len(comp) + 8
    return (b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra + b"BC\x02\x00" + struct.pack("<H", total - 1) +
            tail + comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))


def device_inflate(engine, data, check_crc=True):
    tab, used, text = genoio.bgzf_walk(data)
    assert used == len(data)
    in_off, in_len, out_len, crc = tab
    arr = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(max(text, 1), dtype=np.uint8)
    ms = C.c_double(0)
    vp = lambda a: C.c_void_p(a.ctypes.data)                                  # noqa: E731
    _lib.check(_lib.lib().pg_inflate_device(engine._h, vp(arr), len(arr), vp(in_off), vp(in_len), vp(out_len), vp(crc) if check_crc else None,
                                            len(in_off), vp(out), C.byref(ms)))
    return out[:text].tobytes(), ms.value


def geno_text(rng, n, ns):
    rows = []
    for i in range(n):
        rows.append("scaf%d\t%d\t" % (i // 1000, i * 37 + 1) + "\t".join(rng.choice(["A/A", "A/T", "T/T", "N/N", "A/A", "A/A"]) for _ in range(ns)))
    return ("\n".join(rows) + "\n").encode()


@pytest.fixture(scope="module")
def engine():
    e = Engine(0)
    yield e
    e.close()


def test_members_of_every_kind_inflate_to_what_zlib_gives(engine):
    rng = random.Random(1)
    chunks = [b"", b"a", b"abc" * 400, b"\0" * 65280, bytes(rng.randrange(256) for _ in range(56000)),     # (incompressible: the member must stay < 64 KiB)
              bytes(rng.randrange(4) for _ in range(65280)), geno_text(rng, 300, 50)[:65280], geno_text(rng, 80, 200)[:65280]]
    for n in (1, 2, 3, 63, 64, 65, 127, 128, 129, 257, 258, 259, 1000):
        chunks += [b"x" * n, bytes(rng.randrange(256) for _ in range(n)), (b"ab" * n)[:n]]
    for p in range(1, 70):                                                  # matches of every distance from 1 to beyond the wavefront
        pat = bytes(rng.randrange(256) for _ in range(p))
        chunks.append((pat * 400)[:3000 + p])
    parts, want = [], []
    k = 0
    for ch in chunks:
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                k += 1
                parts.append(member(ch, level, strat, 8 if k % 3 else 1, extra=b"XY\x03\x00abc" if k % 5 == 0 else b"",
                                    name=b"x.geno" if k % 7 == 0 else None))
                want.append(ch)
    parts.append(EOF_MEMBER)
    got, ms = device_inflate(engine, b"".join(parts))
    want = b"".join(want)
    assert len(got) == len(want)
    if got != want:
        at = next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)
        raise AssertionError("first difference at byte %d of %d" % (at, len(want)))
    print("[inflate] %d members, %.1f MB of text in %.3f ms" % (len(parts), len(want) / 1e6, ms))


def test_multi_block_members_and_a_gigabyte_scale_batch(engine):
    """members whose deflate stream holds several blocks (memLevel 1: a new block every 128 symbols... up to dozens per member), and a
    batch of 4000 members (260 MB of text) in one launch"""
    rng = random.Random(5)
    text = geno_text(rng, 1200, 100)                                           # ~ 480 kB
    parts, want = [], []
    for k in range(4000):
        a = (k * 7919) % (len(text) - 65280)
        ch = text[a:a + 65280]
        parts.append(member(ch, 1 + k % 9, zlib.Z_DEFAULT_STRATEGY, 1 + k % 9))
        want.append(ch)
    got, ms = device_inflate(engine, b"".join(parts))
    assert got == b"".join(want)
    print("[inflate] 4000 members, %.0f MB of text in %.2f ms = %.1f GB/s of text" % (len(got) / 1e6, ms, len(got) / ms / 1e6))


@pytest.mark.parametrize("damage", ["bit_in_stream", "crc", "isize_short", "isize_long", "truncated_stream", "distance_too_far", "bad_block_type"])
def test_a_damaged_member_is_named_not_inflated(engine, damage):
    rng = random.Random(11)
    good = [member(geno_text(rng, 60, 40)) for _ in range(5)]
    m = bytearray(good[3])
    if damage == "bit_in_stream":
        m[18 + 40] ^= 0x10
    elif damage == "crc":
        m[-8] ^= 1
    elif damage in ("isize_short", "isize_long"):
        n = struct.unpack("<I", m[-4:])[0] + (-1 if damage == "isize_short" else 1)
        m[-4:] = struct.pack("<I", n)
    elif damage == "truncated_stream":
        body = bytes(m[18:-8])[:-20]
        m = bytearray(bytes(m[:16]) + struct.pack("<H", 18 + len(body) + 8 - 1) + body + bytes(m[-8:]))
    elif damage == "distance_too_far":
        # fixed block: literal 'a' (0x61 + 0x30 = 10010001), then a match of length 3 at distance 4 with a single byte of history
        bits = "1" + "10" + "10010001" + "0000001" + "00011" + "0000000"
        bits += "0" * (-len(bits) % 8)
        body = bytes(int(bits[i:i + 8][::-1], 2) for i in range(0, len(bits), 8))
        m = bytearray(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(body) + 8 - 1) + body + struct.pack("<II", 0, 4))
    elif damage == "bad_block_type":
        body = b"\x07\x00"
        m = bytearray(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(body) + 8 - 1) + body + struct.pack("<II", 0, 0))
    data = b"".join(good[:3] + [bytes(m)] + good[4:])
    with pytest.raises(_lib.PopgenError, match="damaged BGZF member \\(member 3 of the block"):
        device_inflate(engine, data)
    got, _ = device_inflate(engine, b"".join(good))                         # the context is as usable as before
    assert zlib.crc32(got) == zlib.crc32(b"".join(zlib.decompress(g[18:-8], wbits=-15) for g in good))


BGZF_CASES = [c for c in G.STREAMABLE if c["fixture"] != "mixed"]


@pytest.mark.parametrize("blk,block", [(700, 3000), (5000, 3000), (3000, 50000), (65280, 30000)])
@pytest.mark.parametrize("case", BGZF_CASES, ids=lambda c: c["name"])
def test_goldens_as_bgzf_inflated_on_the_device(case, blk, block, tmp_path, monkeypatch, capfd):
    import gzip
    import json
    with gzip.open(os.path.join(G.GOLD, case["fixture"] + ".geno.gz"), "rb") as f:
        text = f.read()
    geno = str(tmp_path / (case["fixture"] + ".geno.gz"))
    write_bgzf(geno, text, blk, empty_member_at=2)
    monkeypatch.setenv("PG_STREAM_BYTES", str(block))
    monkeypatch.setenv("PG_TIMING", "1")
    G.test_cli_reproduces_reference_output(case, tmp_path, geno=geno)
    timing = [json.loads(ln[len("PG_TIMING "):]) for ln in capfd.readouterr().err.splitlines() if ln.startswith("PG_TIMING ")]
    if len(text) > 2 * blk + 65536 and not case["fixture"].startswith("ploidyshift"):       # (changing ploidy: host tokenizer, host inflate)
        assert timing and timing[-1].get("bgzf_blocks_inflated_on_device", 0) > 0


@pytest.mark.parametrize("env", [{"PG_BGZF_NL": "0"}, {"PG_BGZF_NL_CAP": "3"}], ids=["passes_over_the_text", "lists_too_short"])
@pytest.mark.parametrize("name", ["c1_popgen", "abba_windows_sites", "sparse_overlap_failed_id"])
def test_line_feeds_by_passes_over_the_text_give_the_same_rows(name, env, tmp_path, monkeypatch, capfd):
    """round 6: k_inflate lists a block's line feeds member by member (the default, what every other BGZF test runs); PG_BGZF_NL=0 keeps
    k_nl_count / k_nl_scan / k_nl_write, and a member with more line feeds than its list holds (PG_BGZF_NL_CAP=3) sends its block
    through those passes after all: the reference's output either way"""
    import gzip
    case = [c for c in G.CASES if c["name"] == name][0]
    with gzip.open(os.path.join(G.GOLD, case["fixture"] + ".geno.gz"), "rb") as f:
        text = f.read()
    geno = str(tmp_path / (case["fixture"] + ".geno.gz"))
    write_bgzf(geno, text, 5000, empty_member_at=2)
    monkeypatch.setenv("PG_STREAM_BYTES", "30000")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    G.test_cli_reproduces_reference_output(case, tmp_path, geno=geno)


def test_goldens_as_plain_gzip_still_take_the_serial_reader(tmp_path, monkeypatch):
    """a single-stream gzip file cannot be inflated in parallel: the gzip module reads it, the device tokenizer gets text"""
    case = [c for c in G.CASES if c["name"] == "c1_popgen"][0]
    monkeypatch.setenv("PG_STREAM_BYTES", "3000")
    G.test_cli_reproduces_reference_output(case, tmp_path)


@pytest.mark.parametrize("name,tool,size", [("one_popgen_overlap_failed_id", "popgenWindows.py", 8), ("four_popgen_id", "popgenWindows.py", 3),
                                            ("four_abba_overlap", "ABBABABAwindows.py", 2), ("holes_distmat_cat_nexus", "distMat.py", 3),
                                            ("bigpos_popgen_coordinate", "popgenWindows.py", 3)])
def test_bgzf_on_several_ranks_cut_inside_members(name, tool, size, tmp_path):
    import gzip
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from cases import CASES
    case = [c for c in CASES if c["name"] == name][0]
    geno = str(tmp_path / (case["fixture"] + ".geno.gz"))
    with gzip.open(os.path.join(G.GOLD, case["fixture"] + ".geno.gz"), "rb") as f:
        write_bgzf(geno, f.read(), 1900)
    out = str(tmp_path / "ranks.out")
    argv = [a.format(geno=geno, dir=G.GOLD, out=out) for a in case["argv"]] + ["-o", out]
    procs = []
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(38000 + (os.getpid() + size) % 1500), PG_COMM="file", PG_COMM_TIMEOUT="90", PG_TIMING="1",
                   PG_STREAM_BYTES="6000", PG_RDZV_FILE=str(tmp_path / "rdzv"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, tool)] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    errs = [p.communicate(timeout=600)[1].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join("--- rank %d (rc %s)\n%s" % (r, p.returncode, e[-1500:])
                                                            for r, (p, e) in enumerate(zip(procs, errs)))
    assert sum('"sharded_input": true' in ln for e in errs for ln in e.splitlines() if ln.startswith("PG_TIMING ")) == size
    with open(out) as f, open(os.path.join(G.GOLD, case["name"] + ".out")) as g:
        got, want = f.read(), g.read()
    G.compare_text(align_columns(got, want), want, G.round_digits(case))


# ---- the VCF drop-in on a bgzipped VCF: members inflated on the device, the text copied back for the host parser -------------------
def _vcf_cases():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_vcf import VCF_CASES
    return [c for c in VCF_CASES if not any(a in ("--simplifyALT", "--expandMulti", "--field", "--missing") for a in c[2])]


@pytest.mark.parametrize("name,src,argv", _vcf_cases(), ids=[c[0] for c in _vcf_cases()])
@pytest.mark.parametrize("member,block", [(700, 3000), (65280, None)])
def test_bgzipped_vcf_inflated_on_the_device(name, src, argv, member, block, tmp_path, monkeypatch):
    import gzip
    from genomics_general_amd import vcf
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vcf")
    if block:
        monkeypatch.setenv("PG_STREAM_BYTES", str(block))
    bg = str(tmp_path / "in.vcf.gz")
    with gzip.open(os.path.join(gold, src + ".vcf.gz"), "rb") as f:
        text = f.read()
    with open(bg, "wb") as f:
        f.write(genoio.bgzf_compress(text, 6, member).tobytes())
    out = str(tmp_path / "out.geno")
    monkeypatch.setenv("PG_VCF_WAIT_FOR_DEVICE", "1")             # (small files are over before the context exists: wait for it here)
    monkeypatch.setenv("PG_VCF_DEVICE", "0")                      # (the host parser on device-inflated text; the device's parser: tests/test_gpu_vcf.py)
    assert vcf.parse_vcf_main(["-i", bg, "-o", out] + [a.format(dir=gold) for a in argv]) in (0, None)
    info = vcf._text_blocks.last_info
    assert info["bgzf"] and info["blocks"] >= 1
    if member == 700:                           # (one member of 65 280 bytes is this whole file: the header read has inflated it)
        assert info["device_inflate"] and info["inflate_kernel_ms"] > 0 and info["blocks_inflated_on_device"] == info["blocks"]
    with open(out, "rb") as f, open(os.path.join(gold, name + ".geno"), "rb") as g:
        assert f.read() == g.read()


def test_engine_inflate_members_into_a_pinned_array_and_a_damaged_member():
    rng = np.random.default_rng(3)
    text = bytes(rng.choice(list(b"ACGT\t\n0123/|."), size=700000, p=None).astype(np.uint8))
    comp = genoio.bgzf_compress(text, 6, 30000)
    tab, used, n_text = genoio.bgzf_walk(comp, None, 1 << 30)
    assert n_text == len(text)
    e = Engine(0)
    dst = e.pinned.empty((len(text) + 100,), np.uint8)
    dst[:] = 0xEE
    ms = e.inflate_members(memoryview(comp)[:used], tab, dst)
    assert ms > 0 and dst[:len(text)].tobytes() == text and (dst[len(text):] == 0xEE).all()
    bad = np.array(comp, copy=True)
    bad[int(tab[0][5]) + 40] ^= 0x10                                            # a byte inside member 5's deflate stream
    with pytest.raises(_lib.PopgenError):
        e.inflate_members(memoryview(bad)[:used], tab, dst)
    with pytest.raises(ValueError):
        e.inflate_members(memoryview(comp)[:used], tab, dst[:1000])
    e.close()
