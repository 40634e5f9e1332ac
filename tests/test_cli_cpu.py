"""The command-line drivers (genomics_general_amd/cli.py: arguments, windows, streaming in small blocks, packed input, output
formatting) end to end WITHOUT a GPU: the engine is replaced by tests/cpu_engine.CpuEngine, whose per-window numbers come from
the oracle.  The same reference goldens as the -m gpu run, so a regression in the host logic shows up on every machine; the HIP
kernels and engine.WindowBatch are what tests/test_gpu_golden.py adds on an MI355X."""
import os

import pytest

from cases import CASES
from cpu_engine import CpuEngine
from golden_util import align_columns
from genomics_general_amd import cli, genoio

import test_gpu_golden as G

GOLD = G.GOLD


def run_case(case, tmp_path, monkeypatch, geno=None):
    monkeypatch.setattr(cli, "Engine", CpuEngine)
    geno = geno or os.path.join(GOLD, case["fixture"] + ".geno.gz")
    out = str(tmp_path / (case["name"] + ".out"))
    argv = [a.format(geno=geno, dir=GOLD, out=out) for a in case["argv"]] + ["-o", out]
    G.MAINS[case["tool"]](argv)
    with open(out) as f:
        got = f.read()
    with open(os.path.join(GOLD, case["name"] + ".out")) as f:
        want = f.read()
    assert G.compare_text(align_columns(got, want), want, G.round_digits(case)) == 0         # (compare_text asserts on any difference)
    side = os.path.join(GOLD, case["name"] + ".out.windows")
    if os.path.exists(side):
        with open(out + ".windows") as f, open(side) as g:
            assert f.read() == g.read()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_drivers_reproduce_reference_output_on_the_cpu_engine(case, tmp_path, monkeypatch):
    run_case(case, tmp_path, monkeypatch)


@pytest.mark.parametrize("case", G.STREAMABLE, ids=[c["name"] for c in G.STREAMABLE])
def test_drivers_streaming_in_small_blocks_on_the_cpu_engine(case, tmp_path, monkeypatch):
    monkeypatch.setenv("PG_STREAM_BYTES", "4000")
    run_case(case, tmp_path, monkeypatch)


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("c1_popgen", "sparse_sites_windows", "mixed_haploid_flag",
                                                                     "abba_windows_diplo", "multi_distmat", "haplo_popgen")],
                         ids=lambda c: c["name"])
@pytest.mark.parametrize("codec", ["zlib", "none"])
def test_drivers_on_packed_input_on_the_cpu_engine(case, codec, tmp_path, monkeypatch):
    """`.pgeno` input: deflated cells are inflated by host threads and uploaded packed (chunks()); raw cells (codec none) go from the
    file to the device through the staging interface and are expanded there (cli.Run._chunks_device, packed route)"""
    argv = case["argv"]
    fmt = argv[argv.index("-f") + 1] if "-f" in argv else "phased"
    haploid = {"s1": 1, "s6": 1, "s9": 1} if case["fixture"] == "mixed" else {}
    packed = str(tmp_path / (case["fixture"] + ".pgeno"))
    genoio.pack_geno(os.path.join(GOLD, case["fixture"] + ".geno.gz"), packed, "pairs" if fmt == "alleles" else fmt, haploid,
                     block_bytes=20000, codec=codec)
    monkeypatch.setenv("PG_STREAM_BYTES", "30000")
    run_case(case, tmp_path, monkeypatch, geno=packed)


@pytest.mark.parametrize("block", ["4000", "70000"])
@pytest.mark.parametrize("case", [c for c in G.STREAMABLE if c["fixture"] != "mixed" and c["tool"] != "freq.py" and not c["fixture"].startswith("ploidyshift")],
                         ids=lambda c: c["name"])
def test_drivers_with_the_device_tokenizer_interface(case, block, tmp_path, monkeypatch):
    """the default (cli.Run._chunks_device): rows tokenised behind the carried rows, carried rows moved to the front, growth of the
    reservation through the host -- on the stand-in engine, whose tokenize_text is the host tokenizer"""
    monkeypatch.setenv("PG_STREAM_BYTES", block)
    before = CpuEngine.tokenizer_calls
    run_case(case, tmp_path, monkeypatch)
    assert CpuEngine.tokenizer_calls > before, "the driver did not take the device-tokenizer path"


def write_bgzf(path, text, blk, empty_member_at=None):
    """text as a BGZF file with members of blk bytes of text (they end anywhere in a line), optionally an empty member in the middle,
    and bgzip's EOF member at the end"""
    import struct
    import zlib
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    with open(path, "wb") as f:
        for k, a in enumerate(range(0, len(text), blk)):
            chunk = text[a:a + blk]
            c = zlib.compressobj(1 + k % 9, zlib.DEFLATED, -15, 8, zlib.Z_FIXED if k % 7 == 3 else zlib.Z_DEFAULT_STRATEGY)
            comp = c.compress(chunk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                    struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
            if empty_member_at == k:
                f.write(eof)
        f.write(eof)


@pytest.mark.parametrize("blk,block", [(700, "4000"), (5000, "4000"), (3000, "70000"), (65280, "30000")])
@pytest.mark.parametrize("case", [c for c in G.STREAMABLE if c["fixture"] != "mixed"], ids=lambda c: c["name"])
def test_drivers_on_bgzf_input_as_deflated_blocks(case, blk, block, tmp_path, monkeypatch):
    """`.geno.gz` written by bgzip: the blocks reach the engine as genoio.BgzfSpan (members still deflated, inflated by
    tokenize_submit_bgzf -- on the device in the real engine), cut behind their last line feed whatever the members' ends; members
    that end in the middle of a line, an empty member in the middle, the EOF member"""
    import gzip
    with gzip.open(os.path.join(GOLD, case["fixture"] + ".geno.gz"), "rb") as f:
        text = f.read()
    geno = str(tmp_path / (case["fixture"] + ".geno.gz"))
    write_bgzf(geno, text, blk, empty_member_at=2)
    monkeypatch.setenv("PG_STREAM_BYTES", block)
    before = CpuEngine.bgzf_spans
    run_case(case, tmp_path, monkeypatch, geno=geno)
    # (the member that holds the header line is inflated by the reader itself; --inferPloidy on a file whose ploidy changes: the
    # host tokenizer takes the text, inflated by the host pool)
    if len(text) > 2 * blk and not case["fixture"].startswith("ploidyshift"):
        assert CpuEngine.bgzf_spans > before, "no block arrived deflated"


@pytest.mark.parametrize("block", ["4000", "70000", None])
@pytest.mark.parametrize("case", G.STREAMABLE, ids=lambda c: c["name"])
def test_drivers_with_the_host_tokenizer_pipeline(case, block, tmp_path, monkeypatch):
    """PG_GPU_TOKENIZER=0: reader || tokenizer || asynchronous uploads into alternating halves of the resident rows"""
    if block:
        monkeypatch.setenv("PG_STREAM_BYTES", block)
    monkeypatch.setenv("PG_GPU_TOKENIZER", "0")
    before = CpuEngine.tokenizer_calls
    run_case(case, tmp_path, monkeypatch)
    assert CpuEngine.tokenizer_calls == before


def test_device_tokenizer_path_falls_back_block_by_block(tmp_path, monkeypatch):
    """a comment line in the middle of the input: that block goes through the host tokenizer, the others through the device
    interface, the output is that of the clean input"""
    import gzip
    case = [c for c in CASES if c["name"] == "c1_popgen"][0]
    raw = gzip.open(os.path.join(GOLD, "c1.geno.gz"), "rb").read().split(b"\n")
    dirty = str(tmp_path / "dirty.geno")
    with open(dirty, "wb") as f:
        f.write(b"\n".join(raw[:300] + [b"# not a site"] + raw[300:]))
    monkeypatch.setenv("PG_STREAM_BYTES", "20000")
    before = CpuEngine.tokenizer_calls
    run_case(case, tmp_path, monkeypatch, geno=dirty)
    assert CpuEngine.tokenizer_calls > before + 2


@pytest.mark.parametrize("host_tokenizer", [False, True])
def test_infer_ploidy_follows_a_file_whose_cell_widths_change(host_tokenizer, tmp_path, monkeypatch):
    """`--inferPloidy`: the reference infers the ploidy window by window from the shortest cell of the window (genomics.py:1108-1111,
    390-396).  Until round 5 this engine took it once from the first data row and refused a file in which a later window holds a cell
    of another width; now (genoio.PloidySegments, cli.MultiLayoutBatch) such a window is computed under its own ploidies.  Here: a
    diploid sample's cell written with one allele, far behind the first row -- ONE window sees that sample as haploid (all its other
    cells lose their second allele there).  Expected: the oracle's command line (pinned to the reference by the ploidyshift goldens)."""
    import gzip
    import oracle_cli
    case = [c for c in CASES if c["name"] == "mixed_inferploidy"][0]
    lines = gzip.open(os.path.join(GOLD, "mixed.geno.gz"), "rb").read().split(b"\n")
    row = 1500                                                       # a row of a later window, far behind the first block
    cells = lines[row].split(b"\t")
    assert len(cells[2]) == 3                                        # s0 is diploid: `A/C`
    cells[2] = cells[2][:1]
    odd = str(tmp_path / "odd.geno")
    with open(odd, "wb") as f:
        f.write(b"\n".join(lines[:row] + [b"\t".join(cells)] + lines[row + 1:]))
    monkeypatch.setenv("PG_STREAM_BYTES", "20000")
    if host_tokenizer:
        monkeypatch.setenv("PG_GPU_TOKENIZER", "0")
    monkeypatch.setattr(cli, "Engine", CpuEngine)
    out = str(tmp_path / "odd.out")
    argv = [a.format(geno=odd, dir=GOLD, out=out) for a in case["argv"]]
    G.MAINS[case["tool"]](argv + ["-o", out])
    want = oracle_cli.run(case["tool"], argv)
    with open(out) as f:
        got = f.read()
    assert align_columns(got, want) == want
    with open(os.path.join(GOLD, case["name"] + ".out")) as f:
        assert f.read() != want                                      # (the odd cell does change a window)
    run_case(case, tmp_path, monkeypatch)                            # the regular file: the golden of the reference


@pytest.mark.parametrize("name", ["ploidyshift_popgen_sliding_ind", "ploidyshift_abba", "ploidyshift_popgen_sites_pairs"])
def test_infer_ploidy_with_changing_ploidy_on_a_piped_input(name, tmp_path):
    """the cell widths of the WHOLE input decide (one pass), then the run reads it: a piped input is spooled to a temporary file"""
    import gzip
    import subprocess
    import sys
    import test_dist
    case = [c for c in CASES if c["name"] == name][0]
    out = str(tmp_path / "piped.out")
    argv = [a.format(geno="", dir=GOLD, out=out) for a in case["argv"]]
    k = argv.index("-g")
    argv = argv[:k] + argv[k + 2:] + ["-o", out]
    text = gzip.open(os.path.join(GOLD, case["fixture"] + ".geno.gz"), "rb").read()
    r = subprocess.run([sys.executable, "-c", test_dist.CLI_WORKER, case["tool"]] + argv, input=text, capture_output=True, timeout=300,
                       env=dict(os.environ, TMPDIR=str(tmp_path), PG_STREAM_BYTES="9000"))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    with open(out) as f, open(os.path.join(GOLD, case["name"] + ".out")) as g:
        got, want = f.read(), g.read()
    assert align_columns(got, want) == want
    assert [n for n in os.listdir(str(tmp_path)) if n.startswith("pg_stdin_")] == [], "the spooled copy of stdin was left behind"


def test_infer_ploidy_of_a_piped_input(tmp_path):
    """--inferPloidy with the genotypes on stdin: the first data row is looked at and pushed back (genoio.STDIN) -- the golden of the
    same command line with -g FILE"""
    import gzip
    import subprocess
    import sys
    import test_dist
    case = [c for c in CASES if c["name"] == "mixed_inferploidy"][0]
    out = str(tmp_path / "piped.out")
    argv = [a.format(geno="", dir=GOLD, out=out) for a in case["argv"]]
    k = argv.index("-g")
    argv = argv[:k] + argv[k + 2:] + ["-o", out]
    text = gzip.open(os.path.join(GOLD, case["fixture"] + ".geno.gz"), "rb").read()
    r = subprocess.run([sys.executable, "-c", test_dist.CLI_WORKER, case["tool"]] + argv, input=text, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    with open(out) as f, open(os.path.join(GOLD, case["name"] + ".out")) as g:
        got, want = f.read(), g.read()
    assert align_columns(got, want) == want


@pytest.mark.parametrize("tool", ["popgenWindows.py", "ABBABABAwindows.py"])
def test_long_windows_take_the_refinement_pass_on_the_cpu_engine(tool, tmp_path, monkeypatch, capfd):
    """windows of more than cli.NP_MAX_SITES (256) sites: cli._refine_long_windows looks for values within reach of a rounding tie and computes those
    windows again (on the stand-in engine: the same numbers) -- the bookkeeping of that second pass (masks, replaced rows, timing
    field) without a GPU; text == the oracle's command line"""
    import json
    import numpy as np
    import oracle_cli
    from genomics_general_amd import synth
    monkeypatch.setattr(cli, "Engine", CpuEngine)
    n_dip = 8
    names = ["s%d" % d for d in range(n_dip)]
    sid, pos = synth.dense_sites(11000, 1)
    codes = synth.gen_codes(31, sid, pos, n_dip, 4, var_thr=20000, miss_thr=3000)
    geno = str(tmp_path / "long.geno")
    synth.write_geno(geno, ["chr1"], sid, pos, codes, names)
    if tool == "popgenWindows.py":
        argv = ["-g", geno, "-f", "phased", "-w", "5000", "-m", "10", "--roundTo", "12", "-p", "a", "s0,s1,s2,s3", "-p", "b", "s4,s5,s6,s7"]
    else:
        argv = ["-g", geno, "-f", "phased", "-w", "5000", "-m", "10", "--minData", "0.5",
                "-P1", "a", "s0,s1", "-P2", "b", "s2,s3", "-P3", "c", "s4,s5", "-O", "o", "s6,s7"]
    want = oracle_cli.run(tool, argv)
    out = str(tmp_path / "long.out")
    monkeypatch.setenv("PG_TIMING", "1")
    G.MAINS[tool](argv + ["-o", out])
    with open(out) as f:
        assert f.read() == want
    if tool == "popgenWindows.py":
        t = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("PG_TIMING ")]
        assert json.loads(t[-1][len("PG_TIMING "):]).get("windows_recomputed_in_numpy_order", 0) == 3       # 5000 + 5000 + 1000 sites: all beyond 256


def test_a_large_single_stream_gzip_input_gets_the_bgzip_hint(tmp_path, monkeypatch, capfd):
    """a plain gzip file is inflated serially; beyond PG_GZIP_HINT_BYTES the driver says so once and names bgzip"""
    case = [c for c in CASES if c["name"] == "c1_popgen"][0]
    monkeypatch.setenv("PG_GZIP_HINT_BYTES", "1000")
    run_case(case, tmp_path, monkeypatch)
    err = capfd.readouterr().err
    assert err.count("single gzip stream") == 1 and "bgzip" in err


def test_rows_of_float_columns_only_formatted_natively(tmp_path, monkeypatch):
    """popgenWindows.py --analysis indPairDist / popDist: rows of nothing but float columns take pg_format_float_rows when they are wide
    (cli.WIDE_ROW_COLS); forced here on the goldens' narrow tables: the reference's text, nan and tiny values included"""
    used = []
    real = cli._float_rows
    monkeypatch.setattr(cli, "WIDE_ROW_COLS", 1)
    monkeypatch.setattr(cli, "_float_rows", lambda *a, **k: (used.append(1), real(*a, **k))[1])
    n = 0
    for case in CASES:
        an = case["argv"][case["argv"].index("--analysis") + 1:] if "--analysis" in case["argv"] else ["popDist", "popPairDist"]
        an = [a for a in an if not a.startswith("-") and a in ("popDist", "popPairDist", "indPairDist", "popFreq", "indHet", "hapStats")]
        if case["tool"] != "popgenWindows.py" or not set(an) <= {"popDist", "popPairDist", "indPairDist"}:
            continue
        before = len(used)
        run_case(case, tmp_path, monkeypatch)
        n += len(used) > before
    assert n >= 10, n


@pytest.mark.parametrize("name", ["c1_popgen", "abba_windows", "multi_distmat", "abba_freq_derived"])
def test_gz_outputs_are_bgzf_holding_the_plain_output(name, tmp_path, monkeypatch):
    """`-o out.gz` ("If you add `.gz` it will be gzipped", the reference's README on freq.py; popgenWindows.py:316): a gzip file any
    reader takes -- here BGZF, deflated by the library's host threads instead of the gzip module at level 9 on the calling thread --
    whose text is the plain output's"""
    import gzip
    case = next(c for c in CASES if c["name"] == name)
    monkeypatch.setattr(cli, "Engine", CpuEngine)
    geno = os.path.join(GOLD, case["fixture"] + ".geno.gz")
    plain, gz = str(tmp_path / "o.out"), str(tmp_path / "o.out.gz")
    for out in (plain, gz):
        G.MAINS[case["tool"]]([a.format(geno=geno, dir=GOLD, out=out) for a in case["argv"]] + ["-o", out])
    with open(plain, "rb") as f, gzip.open(gz, "rb") as g:
        assert f.read() == g.read()
    assert genoio.BgzfFile.is_bgzf(gz)
    with open(gz, "rb") as f:
        assert f.read().endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def test_names_missing_from_the_header_matter_only_where_a_line_is_read(tmp_path, monkeypatch):
    """A headerless file read with --header and without -p / --samples: the reference takes the sample names from the file's FIRST
    DATA LINE (popgenWindows.py:284-286) and looks them up in the lines it adds to a window (genomics.py:1993) -- a KeyError at the
    first such line, and a header row and nothing else when the contig lists leave no line (its output for this file, run here:
    the expected text below; tools/diff_reference_fuzz.py 300 808, case 69)"""
    import random
    monkeypatch.setattr(cli, "Engine", CpuEngine)
    rnd = random.Random(1)
    geno = tmp_path / "f.geno"
    with open(geno, "w") as f:
        for p in range(1, 400):
            f.write("scaffold_1\t%d\t%s\n" % (p, "\t".join(rnd.choice(["A/A", "A/T", "T/T", "N/N", "G/G"]) for _ in range(3))))
    (tmp_path / "other.scafs").write_text("other\n")
    (tmp_path / "s1.scafs").write_text("scaffold_1\n")
    argv = ["-g", str(geno), "-f", "phased", "-w", "100", "-m", "10", "--writeFailedWindows", "--header", "#CHROM\tPOS\ts0\ts1\ts2",
            "--analysis", "popDist", "indPairDist", "indHet", "--roundTo", "2"]
    out = str(tmp_path / "o.csv")
    assert cli.popgen_main.__wrapped__(argv + ["--include", str(tmp_path / "other.scafs"), "-o", out]) in (0, None)
    with open(out) as f:
        got = f.read().rstrip("\n").split(",")
    want = ("scaffold,start,end,mid,sites,pi_all,d_A/A_A/A,d_A/A_A/T,d_A/A_G/G,d_A/T_A/T,d_A/T_G/G,d_G/G_G/G,"
            "het_A/T,het_G/G,het_A/A").split(",")
    assert got[:12] == want[:12] and sorted(got[12:]) == sorted(want[12:])                  # (het_*: hash order in the reference)
    with pytest.raises(KeyError, match="not in the genotype file header"):
        cli.popgen_main.__wrapped__(argv + ["--include", str(tmp_path / "s1.scafs"), "-o", str(tmp_path / "o2.csv")])
    with pytest.raises(KeyError, match="not in the genotype file header"):
        cli.popgen_main.__wrapped__(argv + ["-o", str(tmp_path / "o3.csv")])


def test_a_sample_without_a_population_matters_only_in_a_window_that_is_computed(tmp_path, monkeypatch):
    """--samples naming an individual of no population beside population statistics: the reference fails in the first window it
    computes statistics for (a TypeError or a hang, genomics.py) and not before -- a run all of whose windows fail --minSites writes
    its rows of nan (tools/diff_reference_fuzz.py 300 909, cases 173 and 200: `-m 0` means the window size)"""
    import random
    monkeypatch.setattr(cli, "Engine", CpuEngine)
    rnd = random.Random(2)
    geno = tmp_path / "f.geno"
    with open(geno, "w") as f:
        f.write("#CHROM\tPOS\ts0\ts1\ts2\ts3\n")
        for p in range(1, 300, 2):
            f.write("chr1\t%d\t%s\n" % (p, "\t".join(rnd.choice(["A/A", "A/T", "T/T", "N/N"]) for _ in range(4))))
    argv = ["-g", str(geno), "-f", "phased", "-w", "100", "--writeFailedWindows", "--analysis", "popDist", "popPairDist", "-p", "A", "s0", "-p", "B", "s1",
            "--samples", "s0,s3,s1"]
    out = str(tmp_path / "o.csv")
    assert cli.popgen_main.__wrapped__(argv + ["-m", "0", "-o", out]) in (0, None)          # minSites = 100 > the 50 sites of a window
    with open(out) as f:
        rows = f.read().splitlines()
    assert rows[0] == "scaffold,start,end,mid,sites,pi_A,pi_B,dxy_A_B,Fst_A_B" and rows[1:] == [
        "chr1,1,100,50,50,nan,nan,nan,nan", "chr1,101,200,150,50,nan,nan,nan,nan", "chr1,201,300,250,50,nan,nan,nan,nan"], rows
    with pytest.raises(AssertionError, match="without a population"):
        cli.popgen_main.__wrapped__(argv + ["-m", "10", "-o", str(tmp_path / "o2.csv")])


def test_h2_that_rounds_to_zero_is_a_float_and_h2_of_one_cluster_an_integer(tmp_path, monkeypatch):
    """H12stats answers the integer `H2 = 0` for a population that is one cluster (genomics.py:1092-1093: printed "0") and a sum of
    squares otherwise -- which may ROUND to zero and is then printed "0.0" (clusters of 22, 1 and 1 haplotypes at --roundTo 2; the
    expected text is the reference's output for this file; tools/diff_reference_fuzz.py 300 1010, cases 89 and 106)"""
    monkeypatch.setattr(cli, "Engine", CpuEngine)
    geno = tmp_path / "f.geno"
    with open(geno, "w") as f:
        f.write("#CHROM\tPOS\t" + "\t".join("h%d" % k for k in range(24)) + "\n")
        for p in range(1, 41):
            row = ["A"] * 24
            if p == 5:
                row[22] = "T"
            if p == 9:
                row[23] = "T"
            f.write("chr1\t%d\t%s\n" % (p, "\t".join(row)))
        for p in range(101, 141):
            f.write("chr1\t%d\t%s\n" % (p, "\t".join(["C"] * 24)))
    out = str(tmp_path / "o.csv")
    cli.popgen_main.__wrapped__(["-g", str(geno), "-f", "haplo", "-w", "100", "-m", "10", "--analysis", "hapStats", "--roundTo", "2", "-o", out])
    with open(out) as f:
        assert f.read() == ("scaffold,start,end,mid,sites,H1_all,H12_all,H2_all\n"
                            "chr1,1,100,20,40,0.84,0.92,0.0\n"
                            "chr1,101,200,120,40,1.0,1.0,0\n")
