"""VCF lines parsed on the device (csrc/pg_vcf_dev.hip: k_vcf_heads / k_vcf_cells / k_vcf_scan behind pg_vcf_dev_*) against the outputs
of the UNMODIFIED reference parseVCF.py (tests/golden/vcf) and against the host parser (pg_encode_vcf + pg_vcf_render_rows): byte for
byte, plain and bgzipped input, blocks of a few lines and whole files, lines the device hands to the host, thousands of sample
columns."""
import gzip
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden_vcf import VCF_CASES  # noqa: E402

from genomics_general_amd import genoio, vcf  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "vcf")

DEVICE_CASES = [c for c in VCF_CASES if not any(a in c[2] for a in ("--field", "--simplifyALT", "--expandMulti")) and
                not any(len(c[2][i + 1]) != 1 for i, a in enumerate(c[2]) if a in ("--missing", "--outSep"))]


def _run(src, out, argv, env, monkeypatch, device="1"):
    monkeypatch.setenv("PG_VCF_DEVICE", device)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rc = vcf.parse_vcf_main(["-i", src, "-o", out] + argv)
    assert rc in (0, None)
    info = dict(vcf._text_blocks.last_info)
    eng = vcf._text_blocks.engine
    info["stats"] = eng.vcf_stats() if (eng is not None and device == "1") else (0, 0)
    with open(out, "rb") as f:
        return f.read(), info


@pytest.mark.parametrize("name,src,argv", DEVICE_CASES, ids=[c[0] for c in DEVICE_CASES])
@pytest.mark.parametrize("form", ["plain", "plain_3000", "bgzf_700_3000", "bgzf_65280", "gzip_stream_3000"])
def test_goldens_through_the_device_parser(name, src, argv, form, tmp_path, monkeypatch):
    with gzip.open(os.path.join(GOLD, src + ".vcf.gz"), "rb") as f:
        text = f.read()
    env = {}
    if form.startswith("plain"):
        path = str(tmp_path / "in.vcf")
        with open(path, "wb") as f:
            f.write(text)
    elif form.startswith("gzip_stream"):                          # `gzip in.vcf`: ONE stream, inflated on the host, parsed on the device
        path = str(tmp_path / "in.vcf.gz")
        with gzip.open(path, "wb") as f:
            f.write(text)
    else:
        path = str(tmp_path / "in.vcf.gz")
        with open(path, "wb") as f:
            f.write(genoio.bgzf_compress(text, 6, int(form.split("_")[1])).tobytes())
    if form.endswith("_3000"):
        env["PG_STREAM_BYTES"] = "3000"
    got, info = _run(path, str(tmp_path / "out.geno"), [a.format(dir=GOLD) for a in argv], env, monkeypatch)
    with open(os.path.join(GOLD, name + ".geno"), "rb") as g:
        assert got == g.read()
    blocks, host_blocks = info["stats"]
    # the goldens' files hold nothing the device hands over: every block it was given came back as rows
    assert info["blocks_parsed_on_device"] >= 1 and blocks == info["blocks_parsed_on_device"] and host_blocks == 0, info


@pytest.mark.parametrize("seed", range(int(os.environ.get("PG_VCF_FUZZ_SEEDS", "24"))))
def test_random_files_device_parser_equals_host_parser(seed, tmp_path, monkeypatch):
    from genomics_general_amd._lib import PopgenError
    from test_vcf import _fuzz_vcf
    rng = np.random.default_rng(31000 + seed)
    names, body, argv = _fuzz_vcf(rng, str(tmp_path))
    head = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names).encode() + b"\n"
    bgz = seed % 2 == 1
    path = str(tmp_path / ("in.vcf.gz" if bgz else "in.vcf"))
    with open(path, "wb") as f:
        f.write(genoio.bgzf_compress(head + body, 6, 5000).tobytes() if bgz else head + body)
    env = {"PG_STREAM_BYTES": str([2000, 20000, 1 << 27][seed % 3])}
    try:
        want, _ = _run(path, str(tmp_path / "host.geno"), argv, env, monkeypatch, device="0")
    except PopgenError:
        with pytest.raises(PopgenError):
            _run(path, str(tmp_path / "dev.geno"), argv, env, monkeypatch)
        return
    got, info = _run(path, str(tmp_path / "dev.geno"), argv, env, monkeypatch)
    assert got == want
    assert info["blocks_parsed_on_device"] >= 1 and info["stats"][1] == 0, info


def _bench_vcf(path, n_sites, n_samples):
    import vcf_bench
    return vcf_bench.write_vcf(path, n_sites, n_samples)


@pytest.mark.parametrize("n_samples,argv", [
    (50, ["--skipIndels", "--minQual", "30", "--gtf", "flag=DP", "min=8", "--gtf", "flag=GQ", "min=20"]),
    (50, ["--gtf", "flag=AD", "min=2", "gtTypes=Het", "--addRefTrack", "--keepPartial"]),
    (3, []),
    (4000, ["--skipIndels", "-s", "ind0003,ind3999,ind0000,ind2048"]),          # two lines per block of the cells kernel
    (9000, ["--gtf", "flag=GQ", "min=30"]),                                       # one line per block
    (14000, ["--skipIndels", "--gtf", "flag=DP", "min=8"]),                       # the most sample columns the device keeps tab positions for
])
def test_gatk_style_file_device_equals_host(n_samples, argv, tmp_path, monkeypatch):
    """tools/vcf_bench.py's generator (GT:AD:DP:GQ, indels, tri-allelic sites, missing calls): plain and bgzipped, several blocks"""
    n_sites = 30000 if n_samples <= 50 else (300 if n_samples < 14000 else 120)
    path = str(tmp_path / "in.vcf")
    _bench_vcf(path, n_sites, n_samples)
    if n_samples >= 4000:
        with open(path, "rb") as f:
            text = f.read()
        # (the generator names its samples ind%03d)
        names = text[text.index(b"#CHROM"):].split(b"\n", 1)[0].split(b"\t")[9:]
        argv = [a if not a.startswith("ind") else ",".join(names[int(x[3:])].decode() for x in a.split(",")) for a in argv]
    bgz = path + ".gz"
    with open(path, "rb") as f, open(bgz, "wb") as g:
        g.write(genoio.bgzf_compress(f.read(), 6, 65280).tobytes())
    env = {"PG_STREAM_BYTES": str(8 << 20)}
    want, _ = _run(path, str(tmp_path / "host.geno"), argv, env, monkeypatch, device="0")
    for src in (path, bgz):
        got, info = _run(src, str(tmp_path / "dev.geno"), argv, env, monkeypatch)
        assert got == want, src
        assert info["blocks_parsed_on_device"] >= 1 and info["stats"][1] == 0, info


@pytest.mark.parametrize("what", ["crlf", "space", "pos_leading_zero", "qual_exponent", "no_final_newline", "seventeen_alleles", "wrong_ploidy"])
def test_lines_the_device_does_not_take_go_to_the_host_parser(what, tmp_path, monkeypatch):
    """one irregular line in the middle of a file of several blocks: that block comes out of the host parser, the others out of the
    device's, and the output is the host parser's output of the whole file (an error where the host parser raises one)"""
    from genomics_general_amd._lib import PopgenError
    with gzip.open(os.path.join(GOLD, "main.vcf.gz"), "rb") as f:
        lines = f.read().split(b"\n")
    first = next(i for i, ln in enumerate(lines) if ln and not ln.startswith(b"#"))
    k = first + 150
    f_ = lines[k].split(b"\t")
    argv = ["--skipIndels", "--minQual", "10"]
    if what == "crlf":
        lines[k] += b"\r"
    elif what == "space":
        lines[k] = lines[k].replace(b"\t", b" ", 1)
    elif what == "pos_leading_zero":
        f_[1] = b"00" + f_[1]
        lines[k] = b"\t".join(f_)
    elif what == "qual_exponent":
        f_[5] = b"1e2"
        lines[k] = b"\t".join(f_)
    elif what == "no_final_newline":
        while lines and not lines[-1]:
            lines.pop()
    elif what == "seventeen_alleles":
        f_[3], f_[4] = b"A", b",".join([b"C", b"G", b"T", b"AA"] * 4)
        lines[k] = b"\t".join(f_)
        argv = []
    elif what == "wrong_ploidy":
        f_[9] = b"0/1/1" + f_[9][3:]
        lines[k] = b"\t".join(f_)
    path = str(tmp_path / "in.vcf")
    with open(path, "wb") as f:
        f.write(b"\n".join(lines))
    env = {"PG_STREAM_BYTES": "6000"}
    if what == "wrong_ploidy":
        for dev in ("0", "1"):
            with pytest.raises(PopgenError, match="ploidy"):
                _run(path, str(tmp_path / "x.geno"), argv, env, monkeypatch, device=dev)
        return
    want, _ = _run(path, str(tmp_path / "host.geno"), argv, env, monkeypatch, device="0")
    got, info = _run(path, str(tmp_path / "dev.geno"), argv, env, monkeypatch)
    assert got == want
    blocks, host_blocks = info["stats"]
    assert host_blocks == 1 and blocks >= 3, info


def test_option_sets_the_device_does_not_take_stay_on_the_host(tmp_path, monkeypatch):
    src = str(tmp_path / "in.vcf.gz")
    with gzip.open(os.path.join(GOLD, "main.vcf.gz"), "rb") as f, open(src, "wb") as g:
        g.write(genoio.bgzf_compress(f.read(), 6, 3000).tobytes())
    argv = ["--skipIndels"] + [x for k in range(9) for x in ("--gtf", "flag=DP", "min=%d" % k)]         # nine genotype filters
    got, info = _run(src, str(tmp_path / "o.geno"), argv, {"PG_VCF_WAIT_FOR_DEVICE": "1"}, monkeypatch)
    want, _ = _run(src, str(tmp_path / "h.geno"), argv, {}, monkeypatch, device="0")
    assert got == want and info["blocks_parsed_on_device"] == 0 and "eight genotype filters" in info.get("device_parser_not_taken", "")


def test_exclude_duplicates_over_block_seams_and_host_blocks(tmp_path, monkeypatch):
    """--excludeDuplicates on the device: runs of equal (CHROM, POS) cut by block seams (the device carries the key), with comment
    lines in between, and with a block that goes to the host parser in the middle (the key crosses over both ways)"""
    rng = np.random.default_rng(5)
    lines, pos = [], 0
    for k in range(3000):
        if rng.random() < 0.6:
            pos += int(rng.integers(1, 4))
        if rng.random() < 0.02:
            lines.append(b"# note %d" % k)
        g = [b"0/0", b"0/1", b"1/1", b"./."][int(rng.integers(0, 4))]
        lines.append(b"chr%d\t%d\t.\tA\tC\t50\tPASS\t.\tGT:DP\t%s:7\t0/1:9" % (1 + k // 1500, pos, g))
    lines[1700] = lines[1700].replace(b"\tPASS", b" PASS")                    # a line only the host reads
    head = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\n"
    path = str(tmp_path / "in.vcf")
    with open(path, "wb") as f:
        f.write(head + b"\n".join(lines) + b"\n")
    for block in ("700", "5000", "100000000"):
        env = {"PG_STREAM_BYTES": block}
        want, _ = _run(path, str(tmp_path / "host.geno"), ["--excludeDuplicates"], env, monkeypatch, device="0")
        got, info = _run(path, str(tmp_path / "dev.geno"), ["--excludeDuplicates"], env, monkeypatch)
        assert got == want, block
        assert info["blocks_parsed_on_device"] >= 1 and info["stats"][1] == 1, info
    assert want.count(b"\n") < 2500                              # (duplicates were there to drop)


def test_a_damaged_member_of_a_bgzipped_vcf_is_named(tmp_path, monkeypatch):
    """the device route inflates the members itself: a member whose bytes were changed fails the run with its number (CRC-32 / Huffman
    code), it does not come out as rows"""
    from genomics_general_amd._lib import PopgenError
    with gzip.open(os.path.join(GOLD, "main.vcf.gz"), "rb") as f:
        text = f.read()
    comp = bytearray(genoio.bgzf_compress(text, 6, 4000).tobytes())
    tab, used, _ = genoio.bgzf_walk(bytes(comp), None, 1 << 40)
    k = len(tab[0]) // 2
    comp[int(tab[0][k]) + int(tab[1][k]) // 2] ^= 0x5a
    path = str(tmp_path / "in.vcf.gz")
    with open(path, "wb") as f:
        f.write(comp)
    with pytest.raises((PopgenError, ValueError), match="member"):
        _run(path, str(tmp_path / "o.geno"), ["--skipIndels"], {"PG_STREAM_BYTES": "20000"}, monkeypatch)


def test_a_piped_vcf_reaches_the_device_once_it_proves_large(tmp_path):
    """`bcftools view ... | parseVCF.py`: no size to look at -- the device context is made when a first block of 32 MB has arrived whole, the
    blocks before it go through the host parser; the rows are the host parser's"""
    import json
    import subprocess
    src = str(tmp_path / "in.vcf")
    _bench_vcf(src, 60000, 100)
    assert os.path.getsize(src) > (80 << 20)
    shim = os.path.join(ROOT, "VCF_processing", "parseVCF.py")
    outs = {}
    for dev in ("0", "1"):
        out = str(tmp_path / ("o%s.geno" % dev))
        env = dict(os.environ, PG_TIMING="1")
        env.pop("PG_STREAM_BYTES", None)
        if dev == "0":
            env["PG_VCF_DEVICE"] = "0"
        else:
            env.pop("PG_VCF_DEVICE", None)
            env["PG_VCF_WAIT_FOR_DEVICE"] = "1"                   # (the host parser would finish these 90 MB before the context exists)
        with open(src, "rb") as f:
            r = subprocess.run([sys.executable, shim, "--skipIndels", "--minQual", "30", "-o", out], stdin=f, env=env, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        tm = [json.loads(ln[10:]) for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")][-1]
        with open(out, "rb") as f:
            outs[dev] = (f.read(), tm)
    assert outs["0"][0] == outs["1"][0]
    assert outs["0"][1]["blocks_parsed_on_device"] == 0 and outs["1"][1]["blocks_parsed_on_device"] >= 1, outs["1"][1]


def test_exclude_duplicates_when_a_block_ends_in_a_line_only_the_host_reads(tmp_path, monkeypatch):
    """--excludeDuplicates: the LAST data line of a block has a spelling the device does not take, or tokens its key cannot hold, and the
    next block starts with its duplicate: the first block goes to the host parser for that line, and the second -- whose first line the
    device cannot judge -- follows it with the host's key.  Blocks of one line each (PG_STREAM_BYTES=1)."""
    def line(chrom, pos, sep=b"\t"):
        return chrom + b"\t" + pos + sep + b".\tA\tC\t50\tPASS\t.\tGT\t0/1\t1/1"
    for chrom, sep in ((b"chr1", b" "), (b"c" * 130, b"\t")):
        lines = [line(b"chr1", b"%d" % (100 + k)) for k in range(30)]
        lines[9] = line(chrom, b"109", sep)
        lines[10] = line(chrom, b"109")                           # its duplicate, regular in every other way
        lines[11] = line(chrom, b"109")                           # and another
        head = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\n"
        path = str(tmp_path / "in.vcf")
        with open(path, "wb") as f:
            f.write(head + b"\n".join(lines) + b"\n")
        env = {"PG_STREAM_BYTES": "1"}
        want, _ = _run(path, str(tmp_path / "host.geno"), ["--excludeDuplicates"], env, monkeypatch, device="0")
        got, info = _run(path, str(tmp_path / "dev.geno"), ["--excludeDuplicates"], env, monkeypatch)
        assert got == want and want.count(b"\n") == 1 + 28, want.count(b"\n")     # (header + 30 lines - the two duplicates)
        assert info["stats"][1] >= 2, info                         # the irregular block and the one behind it


def test_exclude_duplicates_on_a_last_line_without_a_line_feed(tmp_path, monkeypatch):
    """the file's last line has no line feed (that block is the host's) and repeats the line before it, which the device parsed in
    another block: the host parser gets the device's key"""
    def line(pos):
        return b"chr1\t%d\t.\tA\tC\t50\tPASS\t.\tGT\t0/1\t1/1" % pos
    lines = [line(100 + k) for k in range(12)] + [line(111)]
    head = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\n"
    path = str(tmp_path / "in.vcf")
    with open(path, "wb") as f:
        f.write(head + b"\n".join(lines))                         # (no line feed behind the last line)
    env = {"PG_STREAM_BYTES": "1"}
    want, _ = _run(path, str(tmp_path / "host.geno"), ["--excludeDuplicates"], env, monkeypatch, device="0")
    got, info = _run(path, str(tmp_path / "dev.geno"), ["--excludeDuplicates"], env, monkeypatch)
    assert got == want and want.count(b"\n") == 1 + 12
    assert info["stats"][1] == 1, info
