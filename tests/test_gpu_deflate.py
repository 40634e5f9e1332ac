"""k_deflate (csrc/pg_deflate.hip): text deflated on the device into BGZF members.  Whatever it writes must inflate -- by zlib, by the
library's host decoder, by k_inflate -- to the text it was given, member by member (sizes, CRC-32, the BC field); on `.geno` rows its
ratio stays within 15 % of zlib's level 6 (the reference's `| bgzip`, VCF_processing/README.md:33)."""
import gzip
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from genomics_general_amd import genoio, vcf  # noqa: E402
from genomics_general_amd.engine import Engine  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def eng():
    return Engine(0)


def _check(eng, text):
    comp, ms = eng.bgzf_compress(text)
    comp = comp.tobytes()
    if len(text) == 0:
        assert comp == b""
        return comp
    assert gzip.decompress(comp) == text                          # (every member's CRC-32 and size are checked by the gzip module)
    tab, used, n_text = genoio.bgzf_walk(comp, None, 1 << 40)
    in_off, in_len, out_len, crc = tab
    assert used == len(comp) and n_text == len(text)
    assert len(out_len) == (len(text) + 65279) // 65280 and (out_len[:-1] == 65280).all()
    at = 0
    for k in range(len(out_len)):                                 # raw streams, one by one
        piece = zlib.decompress(comp[int(in_off[k]):int(in_off[k]) + int(in_len[k])], wbits=-15)
        assert piece == text[at:at + int(out_len[k])], k
        at += int(out_len[k])
    # ... and back through k_inflate
    dst = eng.pinned.empty((len(text) + 64,), np.uint8)
    eng.inflate_members(comp, tab, dst)
    assert dst[:len(text)].tobytes() == text
    return comp


def _geno_rows():
    with gzip.open(os.path.join(GOLD, "c1.geno.gz"), "rb") as f:
        return f.read()


@pytest.mark.parametrize("what", ["empty", "1", "15", "16", "17", "65279", "65280", "65281", "random_bytes", "zeros", "acgt", "two_symbols",
                                  "geno_rows", "long_lines", "geno_rows_x40", "csv_floats", "binary_runs"])
def test_members_inflate_to_the_text(eng, what):
    rng = np.random.default_rng(len(what) * 1000 + ord(what[0]))
    if what == "empty":
        text = b""
    elif what.isdigit():
        text = bytes(rng.choice(list(b"ACGT/\t\n"), size=int(what)).astype(np.uint8))
    elif what == "random_bytes":
        text = rng.integers(0, 256, size=300000, dtype=np.uint8).tobytes()
    elif what == "zeros":
        text = bytes(200000)
    elif what == "acgt":
        text = b"ACGT" * 70000
    elif what == "two_symbols":
        text = bytes(rng.choice(list(b"AB"), size=150000).astype(np.uint8))
    elif what == "geno_rows":
        text = _geno_rows()
    elif what == "geno_rows_x40":
        text = _geno_rows() * 40
    elif what == "long_lines":
        text = b"".join(bytes(rng.choice(list(b"ACGTN/|\t"), size=int(n)).astype(np.uint8)) + b"\n" for n in rng.integers(1, 90000, size=12))
    elif what == "csv_floats":
        text = "\n".join(",".join("%.4f" % x for x in row) for row in rng.random((6000, 12))).encode()
    else:
        text = b"".join(bytes([int(b)]) * int(n) for b, n in zip(rng.integers(0, 256, size=3000), rng.integers(1, 600, size=3000)))
    comp = _check(eng, text)
    if what == "random_bytes":
        assert len(comp) <= len(text) + (len(text) // 65280 + 1) * 31          # stored members: five bytes of block header + 26 of gzip
    if what in ("zeros", "acgt"):
        assert len(comp) < len(text) // 100


@pytest.mark.parametrize("seed", range(int(os.environ.get("PG_DEFLATE_FUZZ_SEEDS", "60"))))
def test_random_texts(eng, seed):
    rng = np.random.default_rng(77000 + seed)
    n = int(rng.choice([20, 300, 5000, 65280, 70000, 200000, 1000000]))
    n += int(rng.integers(0, 50))
    kind = seed % 4
    if kind == 0:
        alphabet = rng.integers(0, 256, size=int(rng.integers(1, 40)))
        text = bytes(rng.choice(alphabet, size=n).astype(np.uint8))
    elif kind == 1:                                                # rows that repeat the row above with a few changes
        row = rng.choice(list(b"ACGTN/\t"), size=int(rng.integers(20, 3000))).astype(np.uint8)
        rows = []
        while sum(len(r) for r in rows) < n:
            row = row.copy()
            k = int(rng.integers(0, max(len(row) // 20, 1)))
            row[rng.integers(0, len(row), size=k)] = rng.choice(list(b"ACGT"), size=k)
            rows.append(bytes(row) + b"\n")
        text = b"".join(rows)[:n]
    elif kind == 2:                                                # copies at every distance
        base = rng.integers(0, 256, size=int(rng.integers(1, 40000)), dtype=np.uint8).tobytes()
        text = (base * (n // len(base) + 2))[:n]
    else:
        text = b"".join(bytes([int(b)]) * int(k) for b, k in zip(rng.integers(65, 70, size=n // 3 + 1), rng.integers(1, 7, size=n // 3 + 1)))[:n]
    _check(eng, text)


def test_ratio_on_geno_rows_is_within_fifteen_percent_of_zlib_level_6(eng, tmp_path):
    import vcf_bench
    src = str(tmp_path / "in.vcf")
    vcf_bench.write_vcf(src, 20000, 200)
    out = str(tmp_path / "out.geno")
    os.environ["PG_VCF_DEVICE"] = "0"
    try:
        vcf.parse_vcf_main(["-i", src, "-o", out, "--skipIndels", "--minQual", "30", "--gtf", "flag=DP", "min=8", "--gtf", "flag=GQ", "min=20"])
    finally:
        del os.environ["PG_VCF_DEVICE"]
    with open(out, "rb") as f:
        text = f.read()
    comp = _check(eng, text)
    ref = sum(len(zlib.compress(text[a:a + 65280], 6)) + 14 for a in range(0, len(text), 65280))       # (zlib wrapper 6 bytes, BGZF 26 + 5: about the same)
    assert len(comp) <= 1.15 * ref, (len(comp), ref)
    # the real thing: rows that resemble the rows above (the goldens' c1 file, forty times over)
    text = _geno_rows() * 40
    comp, _ = eng.bgzf_compress(text)
    ref = sum(len(zlib.compress(text[a:a + 65280], 6)) + 14 for a in range(0, len(text), 65280))
    assert len(comp) <= 1.15 * ref, (len(comp), ref)


@pytest.mark.parametrize("bgz_in", [False, True])
def test_vcf_drop_in_writes_geno_gz_deflated_on_the_device(bgz_in, tmp_path, monkeypatch):
    """`parseVCF.py -i x.vcf(.gz) -o out.geno.gz`: rows made AND deflated on the device == the host route's plain output; the file is
    BGZF with its end-of-file member"""
    import vcf_bench
    src = str(tmp_path / "in.vcf")
    vcf_bench.write_vcf(src, 30000, 60)
    if bgz_in:
        with open(src, "rb") as f, open(src + ".gz", "wb") as g:
            g.write(genoio.bgzf_compress(f.read(), 6, 65280).tobytes())
        src += ".gz"
    argv = ["--skipIndels", "--minQual", "30", "--gtf", "flag=DP", "min=8"]
    monkeypatch.setenv("PG_STREAM_BYTES", str(6 << 20))
    monkeypatch.setenv("PG_VCF_DEVICE", "0")
    want = str(tmp_path / "host.geno")
    assert vcf.parse_vcf_main(["-i", src, "-o", want] + argv) in (0, None)
    monkeypatch.setenv("PG_VCF_DEVICE", "1")
    got = str(tmp_path / "dev.geno.gz")
    assert vcf.parse_vcf_main(["-i", src, "-o", got] + argv) in (0, None)
    info = vcf._text_blocks.last_info
    assert info["blocks_parsed_on_device"] >= 2
    with open(got, "rb") as f:
        raw = f.read()
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    with open(want, "rb") as f:
        assert gzip.decompress(raw) == f.read()
    assert genoio.BgzfFile.is_bgzf(got)
    # the same through the host's deflate: the same text
    monkeypatch.setenv("PG_DEFLATE_DEVICE", "0")
    got2 = str(tmp_path / "dev2.geno.gz")
    assert vcf.parse_vcf_main(["-i", src, "-o", got2] + argv) in (0, None)
    with open(got2, "rb") as f:
        assert gzip.decompress(f.read()) == gzip.decompress(raw)


def test_bgzip_tool_on_the_device(tmp_path):
    """tools/bgzip.py --device: a BGZF file (EOF member included) whose members k_deflate wrote"""
    import subprocess
    text = _geno_rows() * 3
    src = str(tmp_path / "a.geno")
    with open(src, "wb") as f:
        f.write(text)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "bgzip.py"), "--device", src])
    assert genoio.BgzfFile.is_bgzf(src + ".gz")
    with gzip.open(src + ".gz", "rb") as f:
        assert f.read() == text
