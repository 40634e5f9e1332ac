"""CPU-only tests (-m "not gpu"): the C-ABI library loads and exports every declared symbol, the native tokenizer
(K0, host code inside the .so) agrees with the oracle's parser, sample layout, host finalisers, error behaviour."""
import gzip
import os
import re

import numpy as np
import pytest

from genomics_general_amd import _lib, genoio
from genomics_general_amd.engine import WindowBatch, encode_text
from genomics_general_amd.samples import HapLayout, SampleData
from oracle import popgen_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "popgen_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libpopgen_hip.so does not export " + n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert L.pg_abi_version() == 1


def test_no_gpu_is_a_loud_error_not_a_fallback():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    from genomics_general_amd.engine import Engine
    with pytest.raises(_lib.PopgenError) as ei:
        Engine(0)
    assert ei.value.code == _lib.PG_ERR_NODEV


FIX = [("c1", "phased", 8), ("sparse", "phased", 12), ("abba_pairs", "pairs", 16), ("abba_diplo", "diplo", 16), ("haplo", "haplo", 10)]


@pytest.mark.parametrize("name,fmt,n", FIX)
def test_tokenizer_matches_oracle_parser(name, fmt, n):
    path = os.path.join(GOLD, name + ".geno.gz")
    raw = genoio.read_all(path)
    names, body = genoio.split_header(raw)
    assert len(names) == n
    pl = 1 if fmt == "haplo" else 2
    sd = SampleData(indNames=list(names), ploidyDict={nm: pl for nm in names})
    lay = HapLayout(sd, names, fmt)
    data = genoio.encode(body, lay)
    with gzip.open(path, "rt") as fh:
        onames, sites = orc.read_sites(fh)
    assert onames == names and data.n_sites == len(sites)
    assert np.array_equal(data.pos, [s[1] for s in sites])
    # scaffold runs
    runs = [0] + [i for i in range(1, len(sites)) if sites[i][0] != sites[i - 1][0]]
    assert list(data.run_starts) == runs and data.run_names == [sites[i][0] for i in runs]
    # codes: slot order == file order here (no populations)
    lut = {"A": 1, "C": 2, "G": 4, "T": 8}
    for row in (0, 1, len(sites) // 2, len(sites) - 1):
        want = []
        for cell in sites[row][2]:
            want += [lut.get(a, 0) for a in orc.split_cell(cell, fmt, pl)]
        assert list(data.gt[row]) == want


@pytest.mark.parametrize("name", ["sparse", "abba"])
def test_block_wise_encoding_with_carried_rows_equals_whole_file(name, tmp_path):
    """genoio.BlockReader + encode(head_rows) + concat + tail: the streaming plumbing of cli.Run, without a GPU"""
    path = os.path.join(GOLD, name + ".geno.gz")
    raw = genoio.read_all(path)
    names, body = genoio.split_header(raw)
    lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
    whole = genoio.encode(body, lay)
    rd = genoio.BlockReader(path)
    assert rd.read_header().decode().split()[2:] == names
    carry, seen, first_global = None, 0, 0
    rows_gt, rows_pos = [], []
    k = 0
    while True:
        blk = rd.read_block(7000)
        if not blk:
            break
        assert bytes(blk[-1:]) == b"\n"
        b = genoio.encode(blk, lay, head_rows=carry.n_sites if carry is not None else 0)
        buf = genoio.concat(carry, b)
        n_carry = carry.n_sites if carry is not None else 0
        assert buf.n_sites == n_carry + b.n_sites
        # the buffer is rows [first_global, first_global + n) of the whole file, runs included
        a0 = first_global
        assert np.array_equal(buf.gt, whole.gt[a0:a0 + buf.n_sites]) and np.array_equal(buf.pos, whole.pos[a0:a0 + buf.n_sites])
        r0 = int(np.searchsorted(whole.run_starts, a0, side="right")) - 1
        want_starts = [0] + [int(x) - a0 for x in whole.run_starts[r0 + 1:] if x < a0 + buf.n_sites]
        assert list(buf.run_starts) == want_starts
        assert buf.run_names == whole.run_names[r0:r0 + len(want_starts)]
        keep = (buf.n_sites * (3 + k % 5)) // 8            # carry a varying tail
        carry = genoio.tail(buf, keep)
        first_global = a0 + keep
        k += 1
    assert first_global + (carry.n_sites if carry is not None else 0) == whole.n_sites
    rd.close()


def _bgzf_write(path, data, blk=60000):
    import struct
    import zlib
    with open(path, "wb") as f:
        for a in range(0, len(data), blk):
            chunk = data[a:a + blk]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = c.compress(chunk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                    struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))      # bgzip's EOF member


def test_bgzf_input_is_inflated_member_wise_and_equals_gzip(tmp_path):
    """a .geno.gz written by bgzip (BGZF) goes through the parallel reader; a plain gzip file through the gzip module"""
    raw = genoio.read_all(os.path.join(GOLD, "abba.geno.gz"))
    path = str(tmp_path / "x.geno.gz")
    _bgzf_write(path, raw, blk=7000)
    assert genoio.BgzfFile.is_bgzf(path) and not genoio.BgzfFile.is_bgzf(os.path.join(GOLD, "abba.geno.gz"))
    with gzip.open(path, "rb") as f:
        assert f.read() == raw                               # a valid multi-member gzip file for everybody else
    rd = genoio.BlockReader(path)
    assert isinstance(rd.f, genoio.BgzfFile)
    parts = [rd.read_header()]
    while True:
        b = rd.read_block(50000)
        if not b:
            break
        assert bytes(b[-1:]) == b"\n"
        parts.append(b)
    rd.close()
    assert b"".join(parts) == raw
    rd = genoio.BlockReader(path)
    assert rd.read_block(None) == raw                        # whole-input mode
    rd.close()
    with open(path, "rb") as f:
        cut = f.read()[:-40]
    with open(path, "wb") as f:
        f.write(cut)
    rd = genoio.BlockReader(path)
    with pytest.raises(ValueError):
        rd.read_block(None)


def test_tokenizer_thread_count_does_not_change_output():
    raw = genoio.read_all(os.path.join(GOLD, "abba.geno.gz"))
    names, body = genoio.split_header(raw)
    lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
    a = encode_text(body, lay, n_threads=1)
    b = encode_text(body, lay, n_threads=7)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_tokenizer_column_selection_and_slot_order():
    text = b"chr1 5 A/C G/T N/N T/T\n#comment\nchr1 9 C|C a/T G/G A/N\n\nchr2 2 T/T T/G C/A G/C\n"
    names = ["w", "x", "y", "z"]
    sd = SampleData(popNames=["P", "Q"], popInds=[["z"], ["x", "w"]])          # slot order: z | x w ; y unused
    lay = HapLayout(sd, names, "phased")
    assert lay.ind_order == ["z", "x", "w"] and lay.hap_names == ["z_A", "z_B", "x_A", "x_B", "w_A", "w_B"]
    assert list(lay.hap_pop) == [0, 0, 1, 1, 1, 1]
    gt, pos, soff, slen = encode_text(text, lay)
    assert list(pos) == [5, 9, 2]
    assert gt.tolist() == [[8, 8, 4, 8, 1, 2], [1, 0, 0, 8, 2, 2], [4, 2, 8, 4, 8, 8]]   # lower case 'a' = missing
    assert [text[o:o + n] for o, n in zip(soff, slen)] == [b"chr1", b"chr1", b"chr2"]
    assert list(np.array(lay.hap_names)[lay.ref_order]) == sorted(lay.hap_names)


@pytest.mark.parametrize("text,fmt,msg", [
    (b"chr1 5 A/C G\n", "phased", "ploidy"),
    (b"chr1 5 A/C\n", "phased", "fewer genotype columns"),
    (b"chr1 x A/C G/T\n", "phased", "position"),
    (b"chr1 5 AC GTT\n", "pairs", "ploidy"),
])
def test_tokenizer_errors(text, fmt, msg):
    lay = HapLayout(SampleData(indNames=["a", "b"]), ["a", "b"], fmt)
    with pytest.raises(_lib.PopgenError) as ei:
        encode_text(text, lay)
    assert ei.value.code == _lib.PG_ERR_PARSE and msg in str(ei.value)


def test_positions_with_leading_zeros_and_nineteen_digits():
    """int() of the reference (genomics.py:1884-1904) takes a zero-padded position; nineteen significant digits are
    refused before the multiplication that would leave int64"""
    lay = HapLayout(SampleData(indNames=["a", "b"]), ["a", "b"], "phased")
    data = genoio.encode(b"chr1 0000000000000000000012 A/C G/T\nchr1 000 A/C G/T\nchr1 -0999999999999999999 A/A T/T\nchr1 +007 A/A T/T\n", lay)
    assert list(data.pos) == [12, 0, -999999999999999999, 7]
    for bad in (b"chr1 9999999999999999999 A/C G/T\n", b"chr1 01234567890123456789 A/C G/T\n"):
        with pytest.raises(_lib.PopgenError, match="at most 18 digits"):
            encode_text(bad, lay)


def test_the_documented_numpy_order_limit_is_the_compiled_one():
    """PG_NP_MAX_SITES moved from 4096 to 256 and the header, engine.py and cli.py kept the old number.  The
    header is the contract: whatever it, the Python mirror and the docstrings say must be the number the library was compiled with."""
    import re
    with open(os.path.join(ROOT, "genomics_general_amd", "csrc", "pg_internal.h")) as f:
        compiled = int(re.search(r"#define PG_NP_MAX_SITES (\d+)", f.read()).group(1))
    from genomics_general_amd import cli
    assert cli.NP_MAX_SITES == compiled
    for rel, pat in (("include/popgen_hip.h", r"up to (\d+) sites \(PG_NP_MAX_SITES\)"),
                     ("genomics_general_amd/engine.py", r"up to (\d+) sites \(PG_NP_MAX_SITES\)"),
                     ("genomics_general_amd/cli.py", r"more than NP_MAX_SITES \((\d+)\)")):
        with open(os.path.join(ROOT, rel)) as f:
            found = re.findall(pat, f.read())
        assert found and all(int(x) == compiled for x in found), (rel, found, compiled)


def test_sampledata_mirror():
    sd = SampleData(indNames=["q"], popNames=["A", "B"], popInds=[["x", "y"], ["y", "z"]], ploidyDict={"q": 1, "x": 2, "y": 2, "z": 2})
    assert sd.indNames == ["q", "x", "y", "z"]
    assert sd.getPop("x") == "A" and sd.getPop("y") == ("A", "B") and sd.getPop("q") is None
    assert sd.popInds[0] == ["x", "y"] and sd.ploidy["q"] == 1
    with pytest.raises(ValueError):
        HapLayout(sd, ["q", "x", "y", "z"], "phased")            # y in two populations
    with pytest.raises(KeyError):
        HapLayout(SampleData(indNames=["nope"]), ["a"], "phased")


class FakeLib:
    """Stands in for libpopgen_hip.so in the finaliser test: fills the C-ABI outputs from oracle integer counts."""

    def __init__(self, codes, lay):
        self.codes, self.lay = codes, lay

    def _counts(self, a, b):
        aln, _ = orc.aln_from_codes(self.codes[a:b], [str(i).zfill(4) for i in range(self.lay.n_hap)],
                                    self.lay.hap_sample_name, self.lay.hap_group)
        return orc.pair_counts_gemm(aln)

    def pg_popdist(self, h, lo, hi, n, ms, sums, cnts):
        ps = [0] + list(np.cumsum(self.lay.pop_sizes))
        thr = max(ms, 1)
        for w in range(n):
            D, C = self._counts(lo[w], hi[w])
            k = 0
            for x in range(self.lay.n_pops):
                for y in range(x, self.lay.n_pops):
                    s, c = 0.0, 0
                    for i in range(ps[x], ps[x + 1]):
                        for j in range(ps[y], ps[y + 1]):
                            if (x == y and i >= j) or C[i, j] < thr:
                                continue
                            s += D[i, j] / C[i, j]
                            c += 1
                    sums[w, k], cnts[w, k] = s, c
                    k += 1
        return 0


    def pg_popdist_stats(self, h, lo, hi, n, ms, min_data, do_pairs, tab):
        """CPU emulation of k_popstats (same expression order) on top of the oracle-derived sums."""
        P = self.lay.n_pops
        npairs = P * (P + 1) // 2
        sums = np.zeros((n, npairs)); cnts = np.zeros((n, npairs), dtype=np.int64)
        self.pg_popdist(h, lo, hi, n, ms, sums, cnts)
        size = self.lay.pop_sizes
        kd = lambda p: p * P - p * (p - 1) // 2

        def nmm(total, nv, sz):
            if sz == 0 or not (1 - (1. * (sz - nv) / sz) >= min_data) or nv <= 0:
                return np.nan
            return total / float(nv)
        npo = P * (P - 1) // 2
        for w in range(n):
            pi = [nmm(2 * sums[w, kd(x)], 2 * cnts[w, kd(x)], size[x] * size[x]) for x in range(P)]
            tab[w, :P] = pi
            k = 0
            for x in range(P - 1):
                for y in range(x + 1, P):
                    if do_pairs:
                        kxy = kd(x) + (y - x)
                        tab[w, P + k] = nmm(sums[w, kxy], cnts[w, kxy], size[x] * size[y])
                        wt = 1. * size[x] / (size[x] + size[y])
                        pi_s = wt * pi[x] + (1 - wt) * pi[y]
                        tot = 2 * sums[w, kd(x)] + 2 * sums[w, kd(y)] + 2 * sums[w, kxy]
                        cnt = 2 * cnts[w, kd(x)] + 2 * cnts[w, kd(y)] + 2 * cnts[w, kxy]
                        pi_t = nmm(tot, cnt, (size[x] + size[y]) ** 2)
                        with np.errstate(divide="ignore", invalid="ignore"):
                            tab[w, P + npo + k] = 1 - np.float64(pi_s) / np.float64(pi_t)
                    k += 1
        return 0


class FakeEngine:
    def __init__(self, lib, lay):
        self._L, self._h, self.layout = lib, None, lay


@pytest.mark.parametrize("min_sites,min_data", [(1, 0.01), (30, 0.6), (400, 0.01)])
def test_host_finalisers_match_reference_formulas(min_sites, min_data):
    from genomics_general_amd import synth
    n_dip, n_pops, L = 9, 3, 600
    names = ["s%d" % d for d in range(n_dip)]
    sd = SampleData(popNames=["a", "b", "c"], popInds=[names[0:2], names[2:6], names[6:9]])
    lay = HapLayout(sd, names, "phased")
    sid, pos = synth.dense_sites(L, 1)
    codes = synth.gen_codes(3, sid, pos, n_dip, n_pops, var_thr=40000, miss_thr=25000)
    wb = WindowBatch(FakeEngine(FakeLib(codes, lay), lay), [0, 300], [300, 600])
    got = wb.groupDistStats(True, min_sites, min_data)
    for w, (a, b) in enumerate([(0, 300), (300, 600)]):
        aln, _ = orc.aln_from_codes(codes[a:b], lay.hap_names, lay.hap_sample_name, lay.hap_group)
        D, C = orc.pair_counts_gemm(aln)
        want, _ = orc.group_dist_stats(aln, D, C, True, min_sites, min_data)
        assert set(want) == set(got)
        for k, v in want.items():
            g = got[k][w]
            assert (abs(g - v) < 1e-12) or (g != g and v != v), (k, g, v)


def test_cli_asserts_like_the_reference():
    from genomics_general_amd import cli
    with pytest.raises(AssertionError, match="Window size must be provided"):
        cli.popgen_main(["-f", "phased", "-g", "x.geno"])
    with pytest.raises(AssertionError, match="Overlap does not apply to coordinate windows"):
        cli.popgen_main(["-f", "phased", "-g", "x.geno", "-w", "100", "-O", "5"])
    with pytest.raises(AssertionError, match="between 0 and 1"):
        cli.abbababa_main(["-f", "phased", "-g", "x.geno", "-w", "100", "--minData", "2", "-P1", "a", "s1", "-P2", "b", "s2",
                           "-P3", "c", "s3", "-O", "d", "s4"])


def test_oracle_gemm_counts_equal_reference_pair_loop():
    from genomics_general_amd import synth
    sid, pos = synth.dense_sites(400, 1)
    codes = synth.gen_codes(9, sid, pos, 7, 2, var_thr=40000, miss_thr=20000)
    names = ["h%02d" % i for i in range(14)]
    aln, _ = orc.aln_from_codes(codes, names, names, ["g"] * 14)
    D1, C1 = orc.pair_counts_loop(aln)
    D2, C2 = orc.pair_counts_gemm(aln)
    assert np.array_equal(D1, D2) and np.array_equal(C1, C2)


def test_tokenizer_throughput_report(capsys):
    """Tier T2 evidence: native tokenizer rate on this host (printed; only a loose floor is asserted)."""
    import time
    n_dip, n_lines = 100, 60000
    names = ["s%d" % d for d in range(n_dip)]
    rng = np.random.default_rng(1)
    cells = np.array(["A/A", "A/C", "C/C", "G/T", "N/N", "T/T"])
    rows = cells[rng.integers(0, len(cells), size=(512, n_dip))]
    block = ["\t".join(r) for r in rows]
    text = "".join("chr1\t%d\t%s\n" % (i + 1, block[i % 512]) for i in range(n_lines)).encode()
    lay = HapLayout(SampleData(indNames=list(names)), names, "phased")
    encode_text(text[:100000].rsplit(b"\n", 1)[0] + b"\n", lay)          # warm up
    t0 = time.perf_counter()
    gt, pos, _, _ = encode_text(text, lay)
    dt = time.perf_counter() - t0
    assert len(pos) == n_lines and gt.shape == (n_lines, 2 * n_dip)
    rate = len(text) / dt / 1e6
    with capsys.disabled():
        print("\n[tokenizer] %.0f MB/s, %.2f M sites/s (%d diploids, %d host threads)" % (rate, n_lines / dt / 1e6, n_dip, os.cpu_count()))
    assert rate > 20


@pytest.mark.parametrize("name,fmt,haploid", [("c1", "phased", ()), ("abba_diplo", "diplo", ()), ("abba_pairs", "pairs", ()),
                                              ("haplo", "haplo", ()), ("mixed", "phased", ("s1", "s6", "s9")), ("holes", "phased", ())])
def test_packed_pgeno_round_trip_equals_the_tokenizer(name, fmt, haploid, tmp_path, monkeypatch):
    monkeypatch.setattr(genoio, "PGENO_CHUNK", 1000)                          # several deflate chunks per block
    """genoio.pack_geno -> PackedReader.to_geno (pg_decode_packed) == pg_encode_text of the text, for a layout that reorders,
    drops and regroups samples; block seams inside scaffolds; carried rows in front"""
    path = os.path.join(GOLD, name + ".geno.gz")
    names, body = genoio.split_header(genoio.read_all(path))
    pl = {nm: (1 if (fmt == "haplo" or nm in haploid) else 2) for nm in names}
    out = str(tmp_path / "x.pgeno")
    n = genoio.pack_geno(path, out, fmt, {nm: 1 for nm in haploid}, block_bytes=5000, codec="none" if name == "holes" else "zlib")
    assert genoio.read_header_names(out) == list(names)
    sub = list(names[::-1][: max(2, len(names) - 3)])                         # reversed, three samples dropped
    sd = SampleData(popNames=["x", "y"], popInds=[sub[1::2], sub[0::2]], ploidyDict=pl)
    lay = HapLayout(sd, names, fmt)
    whole = genoio.encode(body, lay)
    assert n == whole.n_sites
    for block_bytes in (None, 3000, 40000):
        rd = genoio.open_input(out)
        assert rd.packed and rd.read_header().decode().split()[2:] == list(names)
        got, carry = None, None
        while True:
            raw = rd.read_block(block_bytes)
            blk = rd.to_geno(raw, lay, n_threads=3, head_rows=got.n_sites if got is not None else 0)
            got = genoio.concat(got, blk)
            if block_bytes is None or not raw:
                break
        rd.close()
        assert np.array_equal(got.gt, whole.gt) and np.array_equal(got.pos, whole.pos)
        assert list(got.run_starts) == list(whole.run_starts) and got.run_names == whole.run_names
    # a ploidy that differs from the packed one is refused, so is a file that is not .pgeno
    if fmt == "phased" and not haploid:
        bad = dict(pl)
        bad[sub[0]] = 1
        lay1 = HapLayout(SampleData(popNames=["x"], popInds=[sub], ploidyDict=bad), names, fmt)
        rd = genoio.open_input(out)
        with pytest.raises(ValueError):
            rd.to_geno(rd.read_block(None), lay1)
    notp = str(tmp_path / "y.pgeno")
    with open(notp, "wb") as f:
        f.write(b"#CHROM\tPOS\ta\n")
    with pytest.raises(ValueError):
        genoio.open_input(notp)


class _FakeComm:
    """all ranks' contributions are computed in this process: allgather returns what the ranks would have sent"""

    def __init__(self, rows):
        self.rows = rows

    def allgather(self, arr):
        return np.array(self.rows, dtype=np.float64)


def test_bgzf_input_shards_at_scaffold_runs(tmp_path):
    """BlockReader.shard on a bgzip file: cuts are (member offset, offset inside the member) pairs; the ranks' slices
    concatenate to the data lines of the whole file, each slice starts on a scaffold-run boundary"""
    from genomics_general_amd import dist
    raw = genoio.read_all(os.path.join(GOLD, "sparse.geno.gz"))
    path = str(tmp_path / "s.geno.gz")
    _bgzf_write(path, raw, blk=3000)
    body = raw[raw.index(b"\n") + 1:]
    for n_ranks in (2, 3):
        size = os.path.getsize(path)
        mine = [(0.0, 0.0)]
        for r in range(1, n_ranks):
            stride = size // n_ranks
            guess = stride * r - min(max(stride // 64, 1 << 12), stride // 2)
            (c, u), _ = genoio.find_run_boundary_bgzf(path, max(guess, 0), lambda nm: True)
            mine.append((float(c), float(u)))
        parts = []
        for r in range(n_ranks):
            rd = genoio.BlockReader(path)
            rd.read_header()
            assert rd.shard(dist.World(r, n_ranks, r), _FakeComm(mine), lambda nm: True, max_share=0.95)
            got = b""
            while True:
                b = rd.read_block(4000)
                if not b:
                    break
                got += bytes(b)
            rd.close()
            parts.append(got)
        assert b"".join(parts) == body
        firsts = [p.split(None, 1)[0] for p in parts if p]
        assert len(set(firsts)) == len(firsts) and all(p.endswith(b"\n") for p in parts if p)


def test_run_boundary_at_a_chunk_seam(tmp_path):
    """A scaffold run that ends exactly where a scan chunk ends: the cut is the FIRST line of the next scaffold (the pattern needs
    the line feed in front of a line; the chunk's first line must still be tested), for plain text and for BGZF input."""
    line = lambda sc, k: ("%s\t%d\tA/A\tC/C\n" % (sc, k)).encode()
    for n_a in (1, 4, 37):
        a = b"".join(line("scafA", k + 1) for k in range(n_a))
        b = b"".join(line("scafB", k + 1) for k in range(50))
        text = a + b
        # chunks that end exactly at the seam, one line before it and one line behind it
        for first in (len(a), len(a) - len(line("scafA", n_a)) if n_a > 1 else len(a), len(a) + len(line("scafB", 1))):
            pos = [0]

            def next_chunk(n, first=first):
                if pos[0] == 0:
                    out = text[:first]
                else:
                    end = text.find(b"\n", min(pos[0] + 40, len(text) - 1)) + 1
                    out = text[pos[0]:end if end > 0 else len(text)]
                pos[0] += len(out)
                return out

            rel, scanned = genoio._scan_run_boundary(next_chunk, lambda nm: True)
            assert rel == len(a), (n_a, first, rel)
    # through the file front ends: every guess up to the seam finds the seam
    header = b"#CHROM\tPOS\ts1\ts2\n"
    a = b"".join(line("scafA", k + 1) for k in range(300))
    b = b"".join(line("scafB", k + 1) for k in range(300))
    path = str(tmp_path / "seam.geno")
    with open(path, "wb") as f:
        f.write(header + a + b)
    seam = len(header) + len(a)
    for guess in (len(header) + 5, seam - 8192, seam - 8192 - 17, seam - 1, seam):
        off, _ = genoio.find_run_boundary(path, max(guess, len(header)), lambda nm: True)
        assert off == seam, (guess, off)
    gz = str(tmp_path / "seam.geno.gz")
    _bgzf_write(gz, header + a + b, blk=len(header) + len(a))          # the first member ends exactly at the seam
    for guess in (0,):
        (c, u), _ = genoio.find_run_boundary_bgzf(gz, guess, lambda nm: True)
        bz = genoio.BgzfFile(gz)
        bz.seek_member(int(c))
        data = bz.read(1 << 20)
        assert data[int(u):].startswith(b"scafB\t1\t"), (guess, c, u)


def test_text_runs_of_raw_lines(tmp_path):
    """pg_text_runs / BlockReader.text_runs: first data line of every run of lines sharing their first field; comment and empty
    lines skipped, a last line without a line feed counted, the ranks' lists tile the file"""
    import ctypes as C
    from genomics_general_amd import _lib, dist
    txt = b"#CHROM\tPOS\ta\nchr1\t1\tA/A\nchr1\t2\tA/A\n#note\n\nchr2\t1\tA/A\nchr10\t5\tA/A\nchr1\t9\tA/A\nchr1 10 A/A"
    L = _lib.lib()
    ptr, n, keep = _lib.text_ptr(txt)
    for cap in (0, 2, 16):
        starts = np.zeros(max(cap, 1), dtype=np.int64)
        cnt = C.c_int64(0)
        _lib.check(L.pg_text_runs(ptr, n, starts, cap, C.byref(cnt)))
        assert cnt.value == 4
        if cap >= 4:
            assert [txt[o:o + 5] for o in starts[:4]] == [b"chr1\t", b"chr2\t", b"chr10", b"chr1\t"]
    path = str(tmp_path / "runs.geno")
    body = b"".join(b"chr%d\t%d\tA/A\n" % (1 + i // 37, i) for i in range(400))
    with open(path, "wb") as f:
        f.write(b"#CHROM\tPOS\ta\n" + body)
    whole = None
    for size in (1, 2, 3, 7):
        runs = []
        for r in range(size):
            rd = genoio.open_input(path)
            rd.read_header()
            for off, name in rd.text_runs(dist.World(r, size, r)):
                if not runs or runs[-1][1] != name:
                    runs.append((off, name))
            rd.close()
        whole = whole or runs
        assert runs == whole and [nm for _, nm in runs] == ["chr%d" % k for k in range(1, 12)]


def test_text_seek_pos_and_skip_rows_walk_data_lines_only():
    """pg_text_seek_pos / pg_text_skip_rows (the window-range cuts of the multi-GPU plan, genomics_general_amd/shardplan.py): comment
    and blank lines are not rows, the walk stops at the first line of another scaffold, an unterminated last line is a line only
    when the caller says the buffer is the whole text"""
    import ctypes as C
    from genomics_general_amd import _lib
    L = _lib.lib()
    text = (b"chr1\t5\tA/C\n#note\nchr1\t9\tA/C\n\nchr1\t9\tC/C\n  chr1 \t 40\tA/A\nchr10\t2\tA/A\nchr1\t50\tA/A")

    def seek(buf, scaf, pos_min, whole=1):
        off, st, ps, rows = C.c_int64(0), C.c_int32(0), C.c_int64(0), C.c_int64(0)
        _lib.check(L.pg_text_seek_pos(buf, len(buf), whole, scaf, len(scaf), pos_min, C.byref(off), C.byref(st), C.byref(ps), C.byref(rows)))
        return off.value, st.value, ps.value, rows.value

    assert seek(text, b"chr1", 1) == (0, 1, 5, 0)
    assert seek(text, b"chr1", 9) == (text.index(b"chr1\t9"), 1, 9, 1)                  # the FIRST of two lines at position 9
    assert seek(text, b"chr1", 10) == (text.index(b"  chr1"), 1, 40, 3)                  # leading blanks, blank-padded fields
    assert seek(text, b"chr1", 41) == (text.index(b"chr10"), 0, 0, 4)                    # chr10 is not chr1: the run is over
    assert seek(text, b"chr10", 1) == (0, 0, 0, 0)
    tail = text[text.index(b"chr1\t50"):]
    assert seek(tail, b"chr1", 50, whole=1) == (0, 1, 50, 0)                              # unterminated last line: a line of the whole text,
    assert seek(tail, b"chr1", 50, whole=0) == (0, -1, 0, 0)                              # left for the next call otherwise
    assert seek(b"", b"chr1", 1) == (0, -1, 0, 0)
    with pytest.raises(_lib.PopgenError):
        seek(b"chr1\tx12\tA/A\n", b"chr1", 1)

    def skip(buf, n):
        off, rows = C.c_int64(0), C.c_int64(0)
        _lib.check(L.pg_text_skip_rows(buf, len(buf), n, C.byref(off), C.byref(rows)))
        return off.value, rows.value

    assert skip(text, 0) == (0, 0)
    assert skip(text, 1) == (text.index(b"chr1\t9"), 1)                                  # (the comment line is walked over, not counted)
    assert skip(text, 2) == (text.index(b"chr1\t9\tC/C"), 2)
    assert skip(text, 6) == (len(text), 6) and skip(text, 99) == (len(text), 6)
    assert 1 <= _lib.usable_cpus() <= (os.cpu_count() or 1)


def test_values_within_reach_of_a_rounding_tie_are_recognised():
    """cli._near_rounding_tie: which statistics of a long window (fixed-tree sums) are computed again in NumPy's order"""
    import numpy as np
    from genomics_general_amd import cli
    v = np.array([0.12345, 0.1234, 0.0, -1e-17, 0.5, 0.30000000000000004, np.nan, 0.99995, 0.123449999])
    assert cli._near_rounding_tie(v, 4).tolist() == [True, False, True, True, False, False, False, True, False]
    assert cli._near_rounding_tie(np.array([0.125]), 2).tolist() == [True] and cli._near_rounding_tie(np.array([0.125]), 3).tolist() == [False]
    r = np.array([1e15, np.inf, -np.inf, 3.0, np.nan])
    assert cli._near_rounding_tie(r, 4, ratio=True).tolist() == [True, True, True, False, False]
    assert cli._near_rounding_tie(np.array([0.3141592653589]), 12).tolist() == [True]          # 12 digits: beyond what the trees agree on
    # Fst = 1 - pi_s / pi_t: an absolute error bound (1e-11) on a value that may be small
    f = np.array([0.000123454996])
    assert cli._near_rounding_tie(f, 8).tolist() == [False] and cli._near_rounding_tie(f, 8, difference=True).tolist() == [True]
    assert cli._near_rounding_tie(np.array([0.0123]), 4, difference=True).tolist() == [False]


def test_the_pairwise_summation_tree_adds_up_like_numpy():
    """pg_np_tree (pg_abi.cpp np_tree: what k_popdist_np walks): values added up along it -- a run as eight interleaved partial sums
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus its tail, the inner nodes in table order -- equal np.sum() to the last bit for lengths
    around every boundary of NumPy's algorithm (8, 128, the halving rule, the 8192-value pieces of the ufunc buffer)"""
    import ctypes as C
    import numpy as np
    from genomics_general_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)

    def run_sum(a):
        n = len(a)
        if n < 8:
            res = 0.0
            for x in a:
                res = res + x
            return res
        r = [a[j] for j in range(8)]
        m8 = n - n % 8
        for i in range(8, m8, 8):
            for j in range(8):
                r[j] = r[j] + a[i + j]
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(m8, n):
            res = res + a[i]
        return res

    for n in (1, 2, 7, 8, 9, 15, 16, 127, 128, 129, 255, 256, 257, 1000, 4097, 8191, 8192, 8193, 8200, 10000, 16384, 16385, 21316, 40000, 131072):
        blob = np.zeros(4 * n // 64 + 64, dtype=np.int32)
        ln = C.c_int64(0)
        _lib.check(L.pg_np_tree(n, blob, len(blob), C.byref(ln)))
        assert ln.value <= len(blob)
        nl, ni, nlev = int(blob[0]), int(blob[1]), int(blob[2])
        off = blob[3:3 + nl + 1]
        left, right = blob[4 + nl:4 + nl + ni], blob[4 + nl + ni:4 + nl + 2 * ni]
        assert off[0] == 0 and off[nl] == n and np.all(np.diff(off) <= 128) and np.all(np.diff(off) >= 1)
        a = (rng.random(n) * 0.1).tolist()
        slots = [run_sum(a[off[k]:off[k + 1]]) for k in range(nl)] + [None] * ni
        for k in range(ni):
            slots[nl + k] = slots[left[k]] + slots[right[k]]
        assert 0.0 + slots[-1] == float(np.sum(np.array(a))), n


@pytest.mark.parametrize("round_to", [4, 10, 0, -1, 15])
def test_format_float_rows_prints_what_numpy_prints(round_to):
    """pg_format_float_rows (the matrix text of distMat.py) against `M.round(r).astype(str)` -- genomics.py:2288-2306 -- on special
    values and on random doubles of every exponent: the same characters"""
    from genomics_general_amd import cli
    rng = np.random.default_rng(round_to + 7)
    special = [0.0, -0.0, float("nan"), float("inf"), -float("inf"), 1e-5, 9.999e-5, 1e-4, 0.00012345, 1e15, 1e16, 9999999999999998.0,
               123456789012345678.0, 1e22, 1e23, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 0.1, 0.5, 1.5, 2.5, 1.0, 100.0,
               1e-7, 123.456, 0.30000000000000004]
    special += [-x for x in special if x == x]
    for vals in (special + [0.0] * (-len(special) % 6), rng.integers(0, 2 ** 64 - 1, size=60000, dtype=np.uint64).view(np.float64),
                 rng.random(60000) * rng.choice([1e-9, 1e-4, 1e-2, 1, 1e3, 1e8, 1e15, 1e17], size=60000), np.round(rng.random(6000), 4)):
        M = np.asarray(vals, dtype=np.float64).reshape(-1, 6)
        with np.errstate(all="ignore"):
            want = "".join(" ".join(row) + "\n" for row in (M.round(round_to) if round_to >= 0 else M).astype(str))
        assert cli._float_rows(M, round_to) == want
    M = rng.random((5, 3))
    pre = ["a  ", "[2] 'bb'    ", "", "x", "yy "]
    want = "".join(p + " ".join(row) + "\n" for p, row in zip(pre, M.round(4).astype(str)))
    assert cli._float_rows(M, 4, pre) == want
