"""Command-line helpers under tools/ that need no GPU."""
import os
import subprocess
import sys

import numpy as np

from genomics_general_amd import genoio
from genomics_general_amd.samples import HapLayout, SampleData

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_geno_pack_tool_writes_a_file_the_reader_decodes(tmp_path):
    """tools/geno_pack.py end to end: mixed ploidy through --ploidyFile, raw and deflated blocks give the same rows"""
    src = os.path.join(GOLD, "mixed.geno.gz")
    names, body = genoio.split_header(genoio.read_all(src))
    pl = {nm: (1 if nm in ("s1", "s6", "s9") else 2) for nm in names}
    lay = HapLayout(SampleData(indNames=list(names), ploidyDict=pl), names, "phased")
    want = genoio.encode(body, lay)
    sizes = {}
    for codec in ("zlib", "none"):
        out = str(tmp_path / ("m_%s.pgeno" % codec))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "geno_pack.py"), "-g", src, "-o", out, "-f", "phased",
                            "--ploidyFile", os.path.join(GOLD, "mixed_ploidy.txt"), "--codec", codec, "--blockMiB", "1"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path), timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-400:]
        assert ("%d sites" % want.n_sites) in r.stderr.decode()
        rd = genoio.open_input(out)
        assert rd.codec == codec and list(rd.ploidy) == [pl[nm] for nm in names]
        got = rd.to_geno(rd.read_block(None), lay)
        rd.close()
        assert np.array_equal(got.gt, want.gt) and np.array_equal(got.pos, want.pos) and got.run_names == want.run_names
        sizes[codec] = os.path.getsize(out)
    assert sizes["zlib"] < sizes["none"]
    # --haploid gives the same file as the ploidy file
    out2 = str(tmp_path / "m2.pgeno")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "geno_pack.py"), "-g", src, "-o", out2, "-f", "phased",
                        "--haploid", "s1,s6,s9", "--codec", "none", "--blockMiB", "1"], stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    with open(out2, "rb") as f, open(str(tmp_path / "m_none.pgeno"), "rb") as g:
        assert f.read() == g.read()


def test_packed_reader_row_shards_concatenate_to_the_whole(tmp_path):
    """PackedReader.shard: run-aligned row ranges from the block headers alone; blocks straddling a cut are trimmed"""
    import numpy as np
    from genomics_general_amd import dist, genoio
    from genomics_general_amd.samples import HapLayout, SampleData
    src = os.path.join(ROOT, "tests", "golden", "sparse.geno.gz")
    for codec in ("zlib", "none"):
        dst = str(tmp_path / ("s_%s.pgeno" % codec))
        genoio.pack_geno(src, dst, "phased", block_bytes=9000, codec=codec)
        rd = genoio.PackedReader(dst)
        lay = HapLayout(SampleData(indNames=list(rd.names)), rd.names, "phased")
        whole = rd.to_geno(rd.read_block(None), lay)
        for n_ranks in (2, 3):
            parts, read = [], 0
            for r in range(n_ranks):
                q = genoio.PackedReader(dst)
                assert q.shard(dist.World(r, n_ranks, r), None, lambda nm: True, max_share=0.9)
                blocks = []
                while True:
                    b = q.read_block(5000)
                    if not b:
                        break
                    blocks += b
                parts.append(q.to_geno(blocks, lay))
                read += q.bytes_read
            assert np.array_equal(np.concatenate([p.gt for p in parts]), whole.gt)
            assert np.array_equal(np.concatenate([p.pos for p in parts]), whole.pos)
            assert sum([p.run_names for p in parts], []) == whole.run_names
            assert read < 1.6 * os.path.getsize(dst)
        # a skipped scaffold next to every cut candidate: no split
        q = genoio.PackedReader(dst)
        assert not q.shard(dist.World(1, 2, 1), None, lambda nm: nm != "chr2")


def test_generated_pair_kernel_loops_are_what_the_generator_writes():
    """csrc/pg_pairc_big.inc (the hand-scheduled main loops of k_pairC_big) is committed; it must be the output of
    csrc/gen_pairc_big.py as committed, and every loop must keep the hazards the generator promises: a register written by a
    VALU operation is read by a matrix instruction no sooner than two instructions later"""
    import re
    csrc = os.path.join(ROOT, "genomics_general_amd", "csrc")
    env = {k: v for k, v in os.environ.items() if k != "PG_CBIG_VARIANT"}
    out = subprocess.run([sys.executable, os.path.join(csrc, "gen_pairc_big.py")], capture_output=True, text=True, env=env, check=True).stdout
    with open(os.path.join(csrc, "pg_pairc_big.inc")) as f:
        assert f.read() == out
    lines = [m.group(1) for m in re.finditer(r'^\s+"(.*?)\\n\\t" \\$', out, re.M)]
    assert len(lines) > 5000
    recent = []                                            # registers written by the last two instructions
    n_mfma = 0
    for ins in lines:
        m = re.match(r"v_mfma\S* (\S+), v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", ins)
        if m:
            n_mfma += 1
            used = set(range(int(m.group(2)), int(m.group(3)) + 1)) | set(range(int(m.group(4)), int(m.group(5)) + 1))
            for w in recent[-2:]:
                assert not (used & w), ins
            recent.append(set())
            continue
        if ins.startswith("s_nop"):
            recent += [set()] * (int(ins.split()[1]) + 1)
            continue
        w = re.match(r"v_\w+ v(\d+),", ins)
        recent.append({int(w.group(1))} if w else set())
    assert n_mfma == 8 * sum(t * (t + 1) // 2 for t in range(1, 8))        # two unrolled pairs x four K steps x the tiles of T = 1 .. 7


def test_roofline_traffic_comes_from_the_newest_profile_round():
    """bench.py's roofline.traffic is the committed figure of the last `rocprofv3 --pmc` passes (profiles/hbm_traffic.json), not a
    live counter: it must not go stale silently -- every workload's `_source` names the newest profiles/rNN directory, and that
    directory holds the kernel statistics the figure belongs to"""
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rounds = sorted(d for d in os.listdir(os.path.join(root, "profiles")) if re.fullmatch(r"r\d\d", d))
    with open(os.path.join(root, "profiles", "hbm_traffic.json")) as f:
        t = json.load(f)
    assert rounds and set(t["_source"].values()) == {rounds[-1]}, (rounds, t["_source"])
    for wl in t["_source"]:
        assert os.path.exists(os.path.join(root, "profiles", rounds[-1], wl + "_kernel_stats.csv"))
        assert os.path.exists(os.path.join(root, "profiles", rounds[-1], wl + "_pmc_summary.json"))
