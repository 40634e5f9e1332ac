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


# ---- the device path's per-line / per-cell functions (csrc/pg_vcf_core.h), walked on the host by tests/vcf_emul.cpp ------------------
import ctypes as C  # noqa: E402
import gzip  # noqa: E402
import subprocess  # noqa: E402


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("vcf_emul") / "libvcf_emul.so")
    # PG_EMUL_SANITIZE=1 (run under LD_PRELOAD of libasan, see `make asan-test`): the device's per-line / per-cell functions under
    # AddressSanitizer + UBSan -- every read of the line's bytes checked against the exact end of the test's buffer
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g"] if os.environ.get("PG_EMUL_SANITIZE") else []
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC"] + san + [os.path.join(ROOT, "tests", "vcf_emul.cpp"), "-o", so])
    return C.CDLL(so)


def _split_header(text):
    """(sample names of the #CHROM line, the bytes behind it)"""
    at = 0
    while True:
        nl = text.index(b"\n", at)
        if text[at:nl].startswith(b"#CHROM"):
            return text[at:nl].decode().split()[9:], text[nl + 1:]
        at = nl + 1


def _emul_rows(emul, plan, body, prev=(None, None)):
    """(status, text, rows, line) of the emulated device path over a block of whole lines"""
    out = np.empty(4 * len(body) + 4096 + 64 * plan.n_sel * (body.count(b"\n") + 1), dtype=np.uint8)
    body = bytes(body)                                            # (an exact-size object: a sanitizer build sees every read behind its end)
    n, rows, line, taken = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
    fn = emul.pgv_emul_block
    fn.restype = C.c_int
    rc = fn(body, C.c_int64(len(body)), *plan.site_args(), C.c_char(plan.sep.encode()), 1 if plan.args.addRefTrack else 0,
            C.c_char_p(prev[0]), len(prev[0] or b""), C.c_char_p(prev[1]), len(prev[1] or b""), C.c_void_p(out.ctypes.data), C.c_int64(out.size), C.byref(n), C.byref(rows), C.byref(line), C.byref(taken))
    return rc, out[:n.value].tobytes(), rows.value, line.value, bool(taken.value)


def _host_rows(plan, body, prev=(None, None)):
    from genomics_general_amd import _lib
    ptr, nbytes, keep = _lib.text_ptr(body)
    k, _, A = plan.host_parse(ptr, nbytes, prev[0], prev[1])
    return (plan.host_render(ptr, k, A).tobytes() if k else b""), k


def _plan(argv, names):
    return vcf.Plan(vcf.make_parser().parse_args(argv), names)


_DEVICE_CASES = [c for c in VCF_CASES if not any(a in c[2] for a in ("--field", "--simplifyALT", "--expandMulti")) and
                 not any(len(c[2][i + 1]) != 1 for i, a in enumerate(c[2]) if a in ("--missing", "--outSep"))]


@pytest.mark.parametrize("name,src,argv", _DEVICE_CASES, ids=[c[0] for c in _DEVICE_CASES])
def test_device_functions_give_the_reference_rows(emul, name, src, argv):
    """every golden the one-character cell form covers: the rows of the emulated device path == the reference's output; a block the
    device path hands to the host (status 1) must be one the option set or the text explains"""
    with gzip.open(os.path.join(GOLD, src + ".vcf.gz"), "rb") as f:
        names, body = _split_header(f.read())
    plan = _plan([a.format(dir=GOLD) for a in argv], names)
    rc, text, rows, line, taken = _emul_rows(emul, plan, body)
    with open(os.path.join(GOLD, name + ".geno"), "rb") as g:
        want = g.read()
    if not args_header_off(argv):
        want = want[want.index(b"\n") + 1:]
    assert taken
    if rc == 1:
        # the goldens' files hold genotypes of the wrong ploidy (an error without --ploidyMismatchToMissing) and nothing else irregular
        bad = body.split(b"\n")[line]
        raise AssertionError("line %d went to the host: %r" % (line, bad[:200]))
    assert rc == 0 and text == want and rows == want.count(b"\n")


def args_header_off(argv):
    return "--noHeader" in argv


def _fuzz_vcf(rng, tmp):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import diff_reference_vcf as D
    import make_golden_vcf as MV
    n = int(D.pick(rng, [1, 2, 4, 6, 9, 70]))
    hap = int(rng.integers(0, n)) if rng.random() < 0.3 else None
    wrong = float(D.pick(rng, [0.0, 0.0, 0.03]))
    path = os.path.join(tmp, "f.vcf.gz")
    MV.make_vcf(path, int(rng.integers(1, 1 << 30)), n_samples=n, hap_sample=hap, snps_only=rng.random() < 0.2, wrong_ploidy=wrong)
    with gzip.open(path, "rb") as f:
        names, body = _split_header(f.read())
    argv = []
    if rng.random() < 0.35:
        argv += ["-s", ",".join(str(x) for x in rng.choice(names, size=int(rng.integers(1, n + 1)), replace=False))]
    r = rng.random()
    contigs = [c for c in ("chr1", "chr2", "chr3", "chrX") if rng.random() < 0.5] or ["chr2"]
    if r < 0.2:
        argv += ["--include", ",".join(contigs)]
    elif r < 0.4:
        argv += ["--exclude", ",".join(contigs)]
    if rng.random() < 0.4:
        argv += ["--minQual", str(int(D.pick(rng, [0, 10, 30, 80])))]
    for _ in range(int(D.pick(rng, [0, 0, 1, 1, 2, 3, 4, 7]))):
        g = ["--gtf", "flag=" + D.pick(rng, ["DP", "GQ", "AD", "XX"])]
        if rng.random() < 0.8:
            g += ["min=" + str(D.pick(rng, [1, 5, 20, 50, 7.5]))]
        if rng.random() < 0.3:
            g += ["max=" + str(int(D.pick(rng, [10, 30, 90])))]
        if rng.random() < 0.3:
            g += ["siteTypes=" + ",".join(t for t in ("SNP", "MONO", "INDEL") if rng.random() < 0.6 or t == "SNP")]
        if rng.random() < 0.3:
            g += ["gtTypes=" + ",".join(t for t in ("Het", "HomRef", "HomAlt", "Missing") if rng.random() < 0.5 or t == "Het")]
        if rng.random() < 0.3:
            g += ["samples=" + ",".join(str(x) for x in rng.choice(names, size=int(rng.integers(1, n + 1)), replace=False))]
        argv += g
    if rng.random() < 0.6:
        argv += ["--skipIndels"]
    if rng.random() < 0.25:
        argv += ["--maxREFlen", str(int(D.pick(rng, [1, 2, 3])))]
    if rng.random() < 0.3:
        argv += ["--excludeDuplicates"]
    if hap is not None and rng.random() < 0.8:
        pf = os.path.join(tmp, "f.ploidy")
        with open(pf, "w") as f:
            f.write("s%d 1\n" % hap)
        argv += ["--ploidyFile", pf]
    elif rng.random() < 0.1:
        argv += ["--ploidy", str(int(D.pick(rng, [1, 2])))]
    if rng.random() < 0.7:
        argv += ["--ploidyMismatchToMissing"]
    if rng.random() < 0.3:
        argv += ["--keepPartial"]
    if rng.random() < 0.3:
        argv += ["--addRefTrack"]
    if rng.random() < 0.25:
        argv += ["--missing", D.pick(rng, ["X", "?", ".", "A"])]
    if rng.random() < 0.25:
        argv += ["--outSep", D.pick(rng, [" ", ",", ";"])]
    return names, body, argv


@pytest.mark.parametrize("seed", range(40))
def test_device_functions_equal_the_host_parser_on_random_files(emul, seed, tmp_path):
    """random files x random option sets: wherever the emulated device path answers, its rows are the host parser's; where the host
    parser raises (a genotype of the wrong ploidy without --ploidyMismatchToMissing) the device path must have handed the block over"""
    from genomics_general_amd._lib import PopgenError
    rng = np.random.default_rng(9000 + seed)
    names, body, argv = _fuzz_vcf(rng, str(tmp_path))
    plan = _plan(argv, names)
    rc, text, rows, line, taken = _emul_rows(emul, plan, body)
    assert taken and rc in (0, 1)
    try:
        want, k = _host_rows(plan, body)
    except PopgenError:
        assert rc == 1
        return
    if rc == 0:
        assert text == want and rows == k
    else:
        # handed over although the host parser has no complaint: only what the head of pg_vcf_core.h lists may do that -- the
        # generator's files hold such lines (QUAL in exponent form is not among them; '*' alleles and long REFs are regular)
        bad = body.split(b"\n")[line]
        raise AssertionError("line %d went to the host: %r (%s)" % (line, bad[:300], " ".join(argv)))


_IRREGULAR = [
    (b"\t", b" ", "a space between two columns"),
    (b"\n", b"\r\n", "CRLF line ends"),
    (b"\tGT:", b"\tXX:", "no GT in FORMAT"),
]


@pytest.mark.parametrize("old,new,what", _IRREGULAR, ids=[x[2] for x in _IRREGULAR])
def test_irregular_spellings_go_to_the_host(emul, old, new, what):
    with gzip.open(os.path.join(GOLD, "snps.vcf.gz"), "rb") as f:
        names, body = _split_header(f.read())
    lines = body.split(b"\n")
    k = len(lines) // 2
    if old == b"\n":
        lines[k] += b"\r"
    else:
        at = lines[k].index(old, 40 if old == b"\t" else 0)
        lines[k] = lines[k][:at] + new + lines[k][at + len(old):]
    plan = _plan([], names)
    rc, _, _, line, taken = _emul_rows(emul, plan, b"\n".join(lines))
    assert taken and rc == 1 and line == k


def test_irregular_numbers_and_positions_go_to_the_host_or_agree(emul):
    """spellings of POS, QUAL and a filtered FORMAT value that only the host parser reads (exponent, sign, leading zeros, sixteen
    digits) hand the block over; plain decimals of up to fifteen digits are read on the device exactly as the host reads them"""
    names = ["a", "b"]
    def line(pos=b"100", qual=b"50", dp=b"12"):
        return b"chr1\t" + pos + b"\t.\tA\tC\t" + qual + b"\tPASS\t.\tGT:DP\t0/1:" + dp + b"\t1/1:7\n"
    for kw, host in [({}, False), ({"pos": b"0100"}, True), ({"pos": b"+100"}, True), ({"qual": b"1e3"}, True), ({"qual": b"."}, False),
                     ({"qual": b"29.999999999999"}, False), ({"qual": b"30.0000000000001"}, False), ({"qual": b"0030"}, False),
                     ({"qual": b"1234567890123456"}, True), ({"dp": b"1e1"}, True), ({"dp": b"-3"}, True), ({"dp": b"."}, False),
                     ({"dp": b"9.99999999999999"}, False), ({"dp": b"10.000000000000"}, False), ({"dp": b""}, False), ({"dp": b"1.2.3"}, False),
                     ({"dp": b"inf"}, True), ({"dp": b"012"}, False)]:
        body = line(**kw) + line(pos=b"200")
        plan = _plan(["--minQual", "30", "--gtf", "flag=DP", "min=10"], names)
        rc, text, rows, ln, taken = _emul_rows(emul, plan, body)
        assert taken
        if host:
            assert rc == 1 and ln == 0, kw
        else:
            want, k = _host_rows(plan, body)
            assert rc == 0 and text == want and rows == k, kw


_ODD = {
    "GT second in FORMAT": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tDP:GT\t12:0/1\t7:1/1\n",
    "GT alone": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT\t0/1\t1/1\n",
    "a cell shorter than FORMAT": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:AD:DP\t0/1\t1/1:3,4:12\n",
    "a cell longer than FORMAT": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:12:99:x\t1/1:7\n",
    "phased": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0|1:12\t1|0:17\n",
    "haploid call": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t1:12\t0/1:17\n",
    "triploid call": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1/1:12\t0/1:17\n",
    "missing calls": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t./.:12\t.|.:17\n",
    "half missing": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t./1:12\t0/.:17\n",
    "a lone dot": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t.:12\t.\n",
    "allele index beyond ALT": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/2:12\t1/1:17\n",
    "allele index with a leading zero": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/01:12\t1/1:17\n",
    "twelve alleles": b"chr1\t100\t.\tA\tC,G,T,AA,AC,AG,AT,CA,CC,CG,CT\t50\tPASS\t.\tGT:DP\t10/11:12\t0/9:17\n",
    "twenty alleles": b"chr1\t100\t.\tA\t" + b",".join(b"A" * k for k in range(2, 21)) + b"\t50\tPASS\t.\tGT:DP\t18/19:12\t0/1:17\n",
    "no ALT": b"chr1\t100\t.\tA\t.\t50\tPASS\t.\tGT:DP\t0/0:12\t0/0:17\n",
    "star allele": b"chr1\t100\t.\tA\tC,*\t50\tPASS\t.\tGT:DP\t0/2:12\t1/2:17\n",
    "symbolic allele": b"chr1\t100\t.\tA\t<DEL>\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "lower case bases": b"chr1\t100\t.\ta\tc\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "an insertion": b"chr1\t100\t.\tA\tACG\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "a deletion": b"chr1\t100\t.\tACG\tA\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "an empty last cell": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:12\t\n",
    "an empty cell": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t\t1/1:17\n",
    "a column more than the header": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\t0/0:9\n",
    "a column fewer than the header": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:12\n",
    "no sample columns": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\n",
    "eight columns": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\n",
    "a tab at the end": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\t\n",
    "a blank inside a cell": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:1 2\t1/1:17\n",
    "a blank in INFO": b"chr1\t100\t.\tA\tC\t50\tPASS\tA=1 B=2\tGT:DP\t0/1:12\t1/1:17\n",
    "an empty CHROM": b"\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "an empty POS": b"chr1\t\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "an empty QUAL": b"chr1\t100\t.\tA\tC\t\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "an empty FORMAT": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\t\t0/1:12\t1/1:17\n",
    "an empty ALT": b"chr1\t100\t.\tA\t\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n",
    "a comma at the end of ALT": b"chr1\t100\t.\tA\tC,\t50\tPASS\t.\tGT:DP\t0/1:12\t1/2:17\n",
    "a colon at the end of FORMAT": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP:\t0/1:12:\t1/1:17\n",
    "FORMAT names that start alike": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DPX:DP\t0/1:1:12\t1/1:99:3\n",
    "a separator of its own in GT": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0-1:12\t1/1:17\n",
    "a letter as allele index": b"chr1\t100\t.\tA\tC\t50\tPASS\t.\tGT:DP\tx/1:12\t1/1:17\n",
    "a high byte": b"chr1\t100\t.\tA\tC\t50\tPASS\t\xc3\xa9\tGT:DP\t0/1:12\t1/1:17\n",
    "a NUL byte": b"chr1\t100\t.\tA\tC\t50\tPASS\t\x00\tGT:DP\t0/1:12\t1/1:17\n",
}
_ODD_ARGV = [[], ["--skipIndels"], ["--ploidyMismatchToMissing", "--keepPartial"], ["--gtf", "flag=DP", "min=10", "--ploidyMismatchToMissing"],
             ["--addRefTrack", "--excludeDuplicates", "--minQual", "30", "--ploidyMismatchToMissing"]]


@pytest.mark.parametrize("what", sorted(_ODD))
def test_odd_lines_agree_with_the_host_parser_or_go_to_it(emul, what):
    """lines a real file may hold and the generators do not write: the emulated device path either gives the host parser's rows or
    hands the block over -- and it must hand over whatever the host parser stops at"""
    from genomics_general_amd._lib import PopgenError
    names = ["a", "b"]
    plain = b"chr1\t%d\t.\tG\tT\t50\tPASS\t.\tGT:DP\t0/1:12\t1/1:17\n"
    body = plain % 90 + _ODD[what] + plain % 110
    for argv in _ODD_ARGV:
        plan = _plan(argv, names)
        rc, text, rows, ln, taken = _emul_rows(emul, plan, body)
        assert taken and rc in (0, 1), (what, argv)
        try:
            want, k = _host_rows(plan, body)
        except (PopgenError, ValueError, AssertionError):
            assert rc == 1 and ln == 1, (what, argv, rc, ln)
            continue
        if rc == 0:
            assert text == want and rows == k, (what, argv, text, want)
        else:
            assert ln == 1, (what, argv, ln)


def test_duplicates_across_blocks_and_comment_lines(emul):
    """--excludeDuplicates: a line is held against the DATA line before it -- over '#' lines and empty lines in between, and over the
    seam of two blocks (the key the device carries); block by block the emulated device path == the host parser"""
    names = ["a", "b"]
    def line(chrom, pos, alt=b"C"):
        return chrom + b"\t" + pos + b"\t.\tA\t" + alt + b"\t50\tPASS\t.\tGT\t0/1\t1/1\n"
    blocks = [line(b"chr1", b"10") + line(b"chr1", b"10", b"G") + b"# a comment\n\n" + line(b"chr1", b"10", b"T") + line(b"chr1", b"11"),
              line(b"chr1", b"11", b"G") + line(b"chr2", b"11") + b"##x\n",
              b"#y\n" + line(b"chr2", b"11", b"T") + line(b"chr2", b"110") + line(b"chr2", b"11"),
              line(b"chr2", b"11", b"G")]
    plan = _plan(["--excludeDuplicates"], names)
    prev, kept = (None, None), []
    for body in blocks:
        rc, text, rows, ln, taken = _emul_rows(emul, plan, body, prev)
        want, k = _host_rows(plan, body, prev)
        assert taken and rc == 0 and text == want and rows == k
        kept.append(rows)
        pc, pp = vcf._last_key(body)
        if pc is not None:
            prev = (pc, pp)
    assert kept == [2, 1, 2, 0]
