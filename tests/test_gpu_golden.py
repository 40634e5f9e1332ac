"""-m gpu: the drop-in command lines, end to end (.geno text -> K0 -> HIP kernels -> CSV), against the committed
outputs of the unmodified reference (tests/golden/, made by tests/golden/make_golden.py).

Every cell must be the reference's TEXT, with no rounding-tie allowance (compare_text asserts on any difference; its `ties=True`
mode, one unit of the last digit, is not used by the golden tests).  Windows of up to cli.NP_MAX_SITES (256) sites form their
float64 sums in NumPy's own order on the device; the goldens' longer windows (c1: 1000 sites, abba: ~600) take the fixed reduction
trees and, where a printed value is within reach of a rounding tie, cli._refine_long_windows computes the window again in NumPy's
order -- so the golden tests exercise both routes and the refinement's bookkeeping."""
import os

import pytest

from cases import CASES
from golden_util import align_columns
from genomics_general_amd import cli

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAINS = {"popgenWindows.py": cli.popgen_main, "ABBABABAwindows.py": cli.abbababa_main, "distMat.py": cli.distmat_main,
         "freq.py": cli.freq_main, "fourPopWindows.py": cli.fourpop_main}


def round_digits(case):
    if case["tool"] in ("ABBABABAwindows.py", "fourPopWindows.py"):
        return 4
    if "--roundTo" in case["argv"]:
        return int(case["argv"][case["argv"].index("--roundTo") + 1])
    return 4


def cells(text):
    return [ln.replace(",", " ").split() for ln in text.splitlines()]


def compare_text(got, want, digits, inexact=None, ties=False):
    """Cells must be the same text.  ties=True (outputs with windows of more than 4096 sites, whose float64 sums are formed in a fixed
    tree instead of NumPy's order: pg_popdist_stats): a floating-point cell may sit on the other side of a rounding tie -- then the
    two printed values differ by exactly ONE unit of the rounding digit (anything else is an error, however small).  Returns the
    number of such cells; `inexact` collects (row, column, got, want)."""
    g, w = cells(got), cells(want)
    assert len(g) == len(w), "row count %d != %d" % (len(g), len(w))
    unit = 10.0 ** (-digits)
    n_inexact = 0
    for r, (gr, wr) in enumerate(zip(g, w)):
        assert len(gr) == len(wr), "row %d: %r vs %r" % (r, gr, wr)
        for c, (gc, wc) in enumerate(zip(gr, wr)):
            if gc == wc:
                continue
            try:
                gv, wv = float(gc), float(wc)
            except ValueError:
                raise AssertionError("row %d: %r != %r" % (r, gc, wc))
            assert ties, "row %d column %d: %r != %r" % (r, c, gc, wc)
            assert abs(abs(gv - wv) - unit) <= 1e-6 * unit, "row %d: %r vs %r is not one unit of the rounding digit" % (r, gc, wc)
            n_inexact += 1
            if inexact is not None:
                inexact.append((r, c, gv, wv))
    return n_inexact


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_cli_reproduces_reference_output(case, tmp_path, geno=None):
    geno = geno or os.path.join(GOLD, case["fixture"] + ".geno.gz")
    out = str(tmp_path / (case["name"] + ".out"))
    argv = [a.format(geno=geno, dir=GOLD, out=out) for a in case["argv"]] + ["-o", out]
    MAINS[case["tool"]](argv)
    with open(out) as f:
        got = f.read()
    with open(os.path.join(GOLD, case["name"] + ".out")) as f:
        want = f.read()
    # the text is the reference's, cell for cell -- no rounding-tie allowance (compare_text words the difference): short windows
    # through k_popdist_np / k_quartet_np / k_popfreq_ordered, longer ones through the fixed trees + cli._refine_long_windows
    n_inexact = compare_text(align_columns(got, want), want, round_digits(case))
    assert n_inexact == 0, "%d cells differ in the last digit" % n_inexact
    side = os.path.join(GOLD, case["name"] + ".out.windows")
    if os.path.exists(side):
        with open(out + ".windows") as f, open(side) as g:
            assert f.read() == g.read()


def _streamable(c):
    if c["tool"] not in ("popgenWindows.py", "ABBABABAwindows.py", "fourPopWindows.py", "distMat.py", "freq.py"):
        return False
    if "--windType" not in c["argv"]:
        return True
    return c["argv"][c["argv"].index("--windType") + 1] in ("coordinate", "sites", "predefined")


STREAMABLE = [c for c in CASES if _streamable(c)]


@pytest.mark.parametrize("case", STREAMABLE, ids=[c["name"] for c in STREAMABLE])
@pytest.mark.parametrize("block", [3000, 50000])
def test_cli_streaming_in_small_blocks_reproduces_reference_output(case, block, tmp_path, monkeypatch):
    """the same goldens with the input consumed in blocks of a few kilobytes (windows.CoordWindowStream / SitesWindowStream +
    carried rows)"""
    monkeypatch.setenv("PG_STREAM_BYTES", str(block))
    test_cli_reproduces_reference_output(case, tmp_path)


@pytest.mark.parametrize("case", STREAMABLE, ids=[c["name"] for c in STREAMABLE])
@pytest.mark.parametrize("block", [3000, 50000, None])
def test_cli_with_the_host_tokenizer_reproduces_reference_output(case, block, tmp_path, monkeypatch):
    """the same goldens with K0 on the host (PG_GPU_TOKENIZER=0: reader || tokenizer threads || asynchronous uploads into
    alternating halves of the resident rows); by default pg_tokenize_text writes the resident rows on the device, and only the
    mixed-ploidy fixtures take the host path, by layout"""
    if block:
        monkeypatch.setenv("PG_STREAM_BYTES", str(block))
    monkeypatch.setenv("PG_GPU_TOKENIZER", "0")
    test_cli_reproduces_reference_output(case, tmp_path)


@pytest.mark.parametrize("codec", ["zlib", "none"])
@pytest.mark.parametrize("case", [c for c in CASES if not c["fixture"].startswith("ploidyshift")], ids=lambda c: c["name"])   # (a `.pgeno` file has ONE ploidy per sample)
def test_cli_on_packed_pgeno_input_reproduces_reference_output(case, codec, tmp_path, monkeypatch):
    """every golden again with the text tokenised once into a `.pgeno` file (genoio.pack_geno, tools/geno_pack.py) and the driver
    reading that; streamed in small blocks where the window type streams.  Deflated cells: host threads inflate them, the cells are
    uploaded packed and expanded on the device; raw cells (codec none): the staging threads read them from the file
    (pg_stage_file) and k_unpack expands them (pg_unpack_staged), two blocks in flight"""
    from genomics_general_amd import genoio
    argv = case["argv"]
    fmt = argv[argv.index("-f") + 1] if "-f" in argv else "phased"
    fmt = "pairs" if fmt == "alleles" else fmt
    haploid = {"s1": 1, "s6": 1, "s9": 1} if case["fixture"] == "mixed" else {}
    packed = str(tmp_path / (case["fixture"] + ".pgeno"))
    genoio.pack_geno(os.path.join(GOLD, case["fixture"] + ".geno.gz"), packed, fmt, haploid, block_bytes=20000, codec=codec)
    if _streamable(case):
        monkeypatch.setenv("PG_STREAM_BYTES", "30000")
    test_cli_reproduces_reference_output(case, tmp_path, geno=packed)


def test_cli_reads_stdin_and_writes_gzip_and_stdout(tmp_path):
    """-g absent = stdin, -o absent = stdout, -o *.gz = gzip (popgenWindows.py:311-316)"""
    import gzip
    import subprocess
    import sys
    case = [c for c in CASES if c["tool"] == "popgenWindows.py" and c["fixture"] == "c1"][0]
    root = os.path.dirname(os.path.dirname(GOLD))
    with gzip.open(os.path.join(GOLD, "c1.geno.gz"), "rb") as f:
        text = f.read()
    argv = [a for a in case["argv"] if a not in ("-g", "{geno}")]
    with open(os.path.join(GOLD, case["name"] + ".out")) as f:
        want = f.read()
    r = subprocess.run([sys.executable, os.path.join(root, "popgenWindows.py")] + argv, input=text, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, cwd=root, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    compare_text(align_columns(r.stdout.decode(), want), want, round_digits(case))
    out = str(tmp_path / "o.csv.gz")
    r = subprocess.run([sys.executable, os.path.join(root, "popgenWindows.py")] + argv + ["-o", out], input=text,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root, timeout=300)
    assert r.returncode == 0 and r.stdout == b"", r.stderr.decode()[-500:]
    with gzip.open(out, "rt") as f:
        compare_text(align_columns(f.read(), want), want, round_digits(case))


@pytest.mark.parametrize("name", ["mixed_haploid_flag", "mixed_ploidyfile_distmat", "abba_freq_counts", "abba_freq_derived"])
def test_mixed_ploidy_and_freq_are_tokenised_on_the_device(name, tmp_path):
    """files of mixed ploidy (narrower cells for the haploid samples) and freq.py's site blocks go through pg_tokenize_text like
    everything else: no block falls back to the host tokenizer, and the output is the reference's"""
    import json
    import subprocess
    import sys
    case = [c for c in CASES if c["name"] == name][0]
    root = os.path.dirname(os.path.dirname(GOLD))
    out = str(tmp_path / (name + ".out"))
    geno = os.path.join(GOLD, case["fixture"] + ".geno.gz")
    argv = [a.format(geno=geno, dir=GOLD, out=out) for a in case["argv"]] + ["-o", out]
    env = dict(os.environ, PG_TIMING="1", PG_STREAM_BYTES="40000")
    r = subprocess.run([sys.executable, os.path.join(root, case["tool"])] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    tm = [json.loads(ln[len("PG_TIMING "):]) for ln in r.stderr.decode().splitlines() if ln.startswith("PG_TIMING ")]
    assert tm and tm[-1].get("device_tokenizer") == 1 and tm[-1]["host_tokenized_blocks"] == 0, tm
    assert tm[-1].get("chunks", tm[-1].get("blocks", 0)) > 1, tm                 # several blocks: carried rows, run seams
    with open(out) as f, open(os.path.join(GOLD, name + ".out")) as g:
        got, want = f.read(), g.read()
    compare_text(align_columns(got, want), want, round_digits(case))
