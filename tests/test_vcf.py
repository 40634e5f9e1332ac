"""The parseVCF.py drop-in (genomics_general_amd/vcf.py + the native pg_encode_vcf) against the outputs of the UNMODIFIED
reference VCF_processing/parseVCF.py on seeded synthetic VCF files (tests/golden/make_golden_vcf.py): byte for byte, for every
supported flag; and the `--packed` route against the text route."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_vcf import VCF_CASES  # noqa: E402

from genomics_general_amd import genoio, vcf  # noqa: E402
from genomics_general_amd.samples import HapLayout, SampleData  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "vcf")


@pytest.mark.parametrize("name,src,argv", VCF_CASES, ids=[c[0] for c in VCF_CASES])
@pytest.mark.parametrize("block", [None, 3000])
def test_parse_vcf_reproduces_the_reference_byte_for_byte(name, src, argv, block, tmp_path, monkeypatch):
    if block:
        monkeypatch.setenv("PG_STREAM_BYTES", str(block))          # many blocks: duplicate test and runs across block seams
    out = str(tmp_path / "out.geno")
    rc = vcf.parse_vcf_main(["-i", os.path.join(GOLD, src + ".vcf.gz"), "-o", out] + [a.format(dir=GOLD) for a in argv])
    assert rc == 0
    with open(out, "rb") as f, open(os.path.join(GOLD, name + ".geno"), "rb") as g:
        assert f.read() == g.read()


def test_packed_route_equals_tokenising_the_text(tmp_path):
    """VCF -> .pgeno directly == VCF -> .geno text -> tokenizer, with calls longer than one base as missing"""
    src = os.path.join(GOLD, "main.vcf.gz")
    txt, pk = str(tmp_path / "a.geno"), str(tmp_path / "a.pgeno")
    assert vcf.parse_vcf_main(["-i", src, "-o", txt, "--packed", pk, "--skipIndels", "--maxREFlen", "1"]) == 0
    rd = genoio.PackedReader(pk)
    lay = HapLayout(SampleData(indNames=list(rd.names)), rd.names, "phased")
    got = rd.to_geno(rd.read_block(None), lay)
    with open(txt, "rb") as f:
        f.readline()
        want = genoio.encode(f.read(), lay)
    assert np.array_equal(got.gt, want.gt) and np.array_equal(got.pos, want.pos) and got.run_names == want.run_names
    # without --maxREFlen 1 the packed file holds the deletion sites too, their multi-base calls as missing
    pk2 = str(tmp_path / "b.pgeno")
    assert vcf.parse_vcf_main(["-i", src, "--packed", pk2, "--skipIndels", "--packedCodec", "none"]) == 0       # raw cells
    rd2 = genoio.PackedReader(pk2)
    all_rows = rd2.to_geno(rd2.read_block(None), lay)
    with open(os.path.join(GOLD, "main_skipindels.geno")) as f_:
        rows = [ln.split() for ln in f_.readlines()[1:]]
    assert all_rows.n_sites == len(rows)
    lut = {"A": 1, "C": 2, "G": 4, "T": 8}
    for r, row in enumerate(rows):
        for c, cell in enumerate(row[2:]):
            a, b = cell.replace("|", "/").split("/")
            assert all_rows.gt[r, 2 * c] == lut.get(a, 0) and all_rows.gt[r, 2 * c + 1] == lut.get(b, 0), (r, c, cell)


def test_unsupported_flags_and_ploidy_errors_are_loud(tmp_path):
    src = os.path.join(GOLD, "hap.vcf.gz")
    with pytest.raises(KeyError):                                  # records without CIGAR in INFO: the reference raises the same
        vcf.parse_vcf_main(["-i", src, "-o", str(tmp_path / "x"), "--expandMulti"])
    with pytest.raises(SystemExit):
        vcf.parse_vcf_main(["-i", os.path.join(GOLD, "cigar.vcf.gz"), "--packed", str(tmp_path / "x.pgeno"), "--expandMulti"])
    with pytest.raises(ValueError):                                # a haploid call where two alleles are expected
        vcf.parse_vcf_main(["-i", os.path.join(GOLD, "cigarhap.vcf.gz"), "-o", str(tmp_path / "x"), "--simplifyALT"])
    from genomics_general_amd._lib import PopgenError
    with pytest.raises(PopgenError):                               # the reference raises ValueError on the haploid calls
        vcf.parse_vcf_main(["-i", src, "-o", str(tmp_path / "x"), "--skipIndels"])


def test_expanded_rows_are_what_the_engine_tokenises():
    """`--expandMulti` turns multi-base freebayes records into one row per base: every cell of the golden output is one allele
    character per haplotype, so the host tokenizer (and the device tokenizer's regular layout) takes the file as it is"""
    with open(os.path.join(GOLD, "cigar_expand.geno"), "rb") as f:
        names = f.readline().decode().split()[2:]
        body = f.read()
    lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
    d = genoio.encode(body, lay)
    assert d.n_sites == body.count(b"\n") and d.gt.shape[1] == 2 * len(names)
    assert set(np.unique(d.gt)) <= {0, 1, 2, 4, 8} and (d.gt != 0).mean() > 0.5


def test_gz_output_is_bgzf_and_holds_the_same_text(tmp_path):
    """-o out.geno.gz: BGZF (what `parseVCF.py | bgzip` gives): a valid gzip file whose text is the plain output's"""
    import gzip
    from genomics_general_amd import genoio
    plain, gz = str(tmp_path / "x.geno"), str(tmp_path / "x.geno.gz")
    for out in (plain, gz):
        assert vcf.parse_vcf_main(["-i", os.path.join(GOLD, "main.vcf.gz"), "-o", out]) in (0, None)
    with open(plain, "rb") as f, gzip.open(gz, "rb") as g:
        text = f.read()
        assert g.read() == text and len(text) > 1000
    assert genoio.BgzfFile.is_bgzf(gz)
    rd = genoio.BlockReader(gz)
    assert rd.read_header() + rd.read_block(None) == text
    rd.close()


def _bgzf_copy(src_gz, dst, member_text):
    """the text of a gzip file as BGZF with members of member_text bytes (they end anywhere in a line)"""
    import gzip
    with gzip.open(src_gz, "rb") as f:
        text = f.read()
    with open(dst, "wb") as f:
        f.write(genoio.bgzf_compress(text, 6, member_text).tobytes())
    return text


@pytest.mark.parametrize("name,src,argv", VCF_CASES, ids=[c[0] for c in VCF_CASES])
@pytest.mark.parametrize("member,block", [(700, 3000), (4000, 1 << 20), (65280, None)])
def test_bgzipped_vcf_is_read_as_members_and_gives_the_reference_text(name, src, argv, member, block, tmp_path, monkeypatch):
    """a VCF written by bgzip: read as spans of deflated members, inflated into the ring of block buffers (here by the library's
    host threads; on a GPU box by k_inflate: tests/test_gpu_inflate.py), parsed and rendered natively: == the reference's output"""
    if block:
        monkeypatch.setenv("PG_STREAM_BYTES", str(block))
    bg = str(tmp_path / "in.vcf.gz")
    _bgzf_copy(os.path.join(GOLD, src + ".vcf.gz"), bg, member)
    assert genoio.BgzfFile.is_bgzf(bg)
    out = str(tmp_path / "out.geno")
    assert vcf.parse_vcf_main(["-i", bg, "-o", out] + [a.format(dir=GOLD) for a in argv]) in (0, None)
    if not any(a in ("--simplifyALT", "--expandMulti", "--field", "--missing") for a in argv):    # (those are line loops on the host)
        info = vcf._text_blocks.last_info
        assert info["bgzf"] and not info["device_inflate"] and info["blocks"] >= 1
    with open(out, "rb") as f, open(os.path.join(GOLD, name + ".geno"), "rb") as g:
        assert f.read() == g.read()


def _render_py(buf, k, pl, chars, aidx, phase, rflag, pos, coff, clen, roff, rlen, aoff, alen, sep, missing, add_ref):
    """the rows the way the reference prints them (parseVCF.py:151-169, 380-383), from pg_encode_vcf's outputs"""
    rows = []
    for i in range(k):
        pre = [buf[coff[i]:coff[i] + clen[i]], str(int(pos[i])).encode()]
        ref = buf[roff[i]:roff[i] + rlen[i]]
        if add_ref:
            pre.append(ref)
        alleles = [ref]
        alt = buf[aoff[i]:aoff[i] + alen[i]]
        if alt != b".":
            alleles += alt.split(b",")
        cells = []
        for s in range(len(pl)):
            if rflag[i]:
                calls = [alleles[a] if a >= 0 else missing for a in aidx[i, 2 * s:2 * s + pl[s]]]
            else:
                calls = [bytes([c]) for c in chars[i, 2 * s:2 * s + pl[s]]]
            cells.append(bytes([phase[i, s]]).join(calls))
        rows.append(sep.join(pre + cells) + b"\n" if len(pl) else sep.join(pre) + sep + b"\n")
    return b"".join(rows)


@pytest.mark.parametrize("seed", range(6))
def test_render_rows_equals_the_line_by_line_rendering(seed):
    """pg_vcf_render_rows on random parser outputs: plain rows, rows with alleles longer than a base (cells from the REF / ALT
    strings), haploid columns, negative and 18-digit positions, a reference track, one to many host threads"""
    import ctypes as C
    from genomics_general_amd import _lib
    rng = np.random.default_rng(seed)
    L = _lib.lib()
    k, n_sel = int(rng.integers(1, 900)), int(rng.integers(0, 9)) if seed else 0
    pl = rng.integers(1, 3, size=n_sel).astype(np.int32)
    names = [b"chr1", b"scaffold_22", b"x"]
    buf, coff, clen, roff, rlen, aoff, alen, n_alt = bytearray(), [], [], [], [], [], [], []
    for i in range(k):
        nm = names[int(rng.integers(0, 3))]
        ref = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 4))).tolist())
        alts = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 5))).tolist()) for _ in range(int(rng.integers(0, 4)))]
        alt = b",".join(alts) if alts else b"."
        for lst, lens, tok in ((coff, clen, nm), (roff, rlen, ref), (aoff, alen, alt)):
            lst.append(len(buf))
            lens.append(len(tok))
            buf += tok + b"\t"
        n_alt.append(len(alts))
    i64, i32 = (lambda x: np.array(x, dtype=np.int64)), (lambda x: np.array(x, dtype=np.int32))
    coff, roff, aoff, clen, rlen, alen = i64(coff), i64(roff), i64(aoff), i32(clen), i32(rlen), i32(alen)
    pos = rng.integers(-5, 10 ** 7, size=k).astype(np.int64)
    pos[0] = 10 ** 17 + 12345
    chars = rng.choice(list(b"ACGTN"), size=(k, 2 * n_sel)).astype(np.uint8)
    phase = rng.choice(list(b"/|"), size=(k, n_sel)).astype(np.uint8)
    aidx = np.stack([rng.integers(-1, n_alt[i] + 1, size=2 * n_sel) for i in range(k)]).astype(np.int8).reshape(k, 2 * n_sel)
    rflag = (rng.random(k) < 0.3).astype(np.uint8)
    buf = bytes(buf)
    vp = lambda a: C.c_void_p(a.ctypes.data if a.size else 0)                   # noqa: E731
    for sep, missing, add_ref, nt in ((b"\t", b"N", 0, 1), (b" ", b"X", 1, 3), (b"\t", b"N", 1, 0)):
        want = _render_py(buf, k, pl, chars, aidx, phase, rflag, pos, coff, clen, roff, rlen, aoff, alen, sep, missing, add_ref)
        args = (buf, k, n_sel, vp(pl), vp(chars), vp(aidx), vp(phase), vp(rflag), vp(pos), vp(coff), vp(clen), vp(roff), vp(rlen), vp(aoff),
                vp(alen), C.c_char(sep), C.c_char(missing), add_ref)
        size = C.c_int64(0)
        _lib.check(L.pg_vcf_render_rows(*args, None, 0, C.byref(size), nt))
        assert size.value == len(want)
        out = np.full(size.value + 8, 0xEE, dtype=np.uint8)
        _lib.check(L.pg_vcf_render_rows(*args, vp(out), size.value, C.byref(size), nt))
        assert out[:size.value].tobytes() == want and (out[size.value:] == 0xEE).all()
        assert L.pg_vcf_render_rows(*args, vp(out), size.value - 1, C.byref(size), nt) < 0          # an output that is too small: refused
    if n_sel and rflag.any():
        bad = aidx.copy()
        bad[int(np.flatnonzero(rflag)[0]), 0] = 100                             # an index that names no allele
        assert L.pg_vcf_render_rows(buf, k, n_sel, vp(pl), vp(chars), vp(bad), vp(phase), vp(rflag), vp(pos), vp(coff), vp(clen), vp(roff),
                                    vp(rlen), vp(aoff), vp(alen), C.c_char(b"\t"), C.c_char(b"N"), 0, None, 0, C.byref(size), 1) < 0


def test_filter_on_a_format_field_beyond_the_sixteenth_colon(tmp_path):
    """the parser keeps the offsets of a cell's first 16 ':'; a --gtf flag further back takes the general route: the same rows as with
    the field moved to the front of FORMAT"""
    rng = np.random.default_rng(12)
    pad = ["F%02d" % k for k in range(17)]
    lines_far, lines_near = [], []
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\tc\n"
    for i in range(300):
        far, near = [], []
        for s in range(3):
            gt = ["0/0", "0/1", "1|1", "./.", "1/0"][int(rng.integers(0, 5))]
            dp = str(int(rng.integers(0, 30)))
            junk = [str(int(rng.integers(0, 9))) for _ in pad]
            far.append(":".join([gt] + junk + [dp]))
            near.append(":".join([gt, dp] + junk))
        pre = "chr1\t%d\t.\tA\tC\t50\tPASS\t.\t" % (i + 1)
        lines_far.append(pre + ":".join(["GT"] + pad + ["DP"]) + "\t" + "\t".join(far))
        lines_near.append(pre + ":".join(["GT", "DP"] + pad) + "\t" + "\t".join(near))
    outs = []
    for tag, lines in (("far", lines_far), ("near", lines_near)):
        src, out = str(tmp_path / (tag + ".vcf")), str(tmp_path / (tag + ".geno"))
        with open(src, "w") as f:
            f.write(head + "\n".join(lines) + "\n")
        assert vcf.parse_vcf_main(["-i", src, "-o", out, "--gtf", "flag=DP", "min=10"]) in (0, None)
        with open(out, "rb") as f:
            outs.append(f.read())
    assert outs[0] == outs[1] and outs[0].count(b"N/N") > 100 and outs[0].count(b"\n") == 301
