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
