#!/usr/bin/env python
"""k_deflate alone: `.geno` rows (the VCF drop-in's output on tools/vcf_bench.py's synthetic VCF) deflated on the device -- kernel time,
GB/s of text, ratio against zlib level 6 and against the library's host compressor.   python tools/deflate_bench.py [n_sites] [n_samples]"""
import json
import os
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import vcf_bench
    from genomics_general_amd import genoio, vcf
    from genomics_general_amd.engine import Engine
    n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    tmp = tempfile.mkdtemp(prefix="pg_dfl_", dir=os.environ.get("PG_BENCH_TMP", "/tmp"))
    src, out = os.path.join(tmp, "in.vcf"), os.path.join(tmp, "out.geno")
    vcf_bench.write_vcf(src, n_sites, n_samples)
    vcf.parse_vcf_main(["-i", src, "-o", out, "--skipIndels", "--minQual", "30", "--gtf", "flag=DP", "min=8", "--gtf", "flag=GQ", "min=20"])
    with open(out, "rb") as f:
        text = f.read()
    for fn in (src, out):
        os.remove(fn)
    os.rmdir(tmp)
    eng = Engine(0)
    eng.bgzf_compress(text[:1 << 20])
    best = None
    for _ in range(3):
        comp, ms = eng.bgzf_compress(text)
        best = ms if best is None else min(best, ms)
    assert zlib.decompress(comp.tobytes()[:0] or b"x\x9c\x03\x00\x00\x00\x00\x01") == b""
    import gzip
    assert gzip.decompress(comp.tobytes()) == text
    sample = text[:64 << 20]
    t0 = time.perf_counter()
    z6 = sum(len(zlib.compress(sample[a:a + 65280], 6)) + 14 for a in range(0, len(sample), 65280))
    t_z6 = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = len(genoio.bgzf_compress(sample, 6, 65280, eof_marker=False))
    t_host = time.perf_counter() - t0
    dev_sample = len(eng.bgzf_compress(sample)[0])
    print(json.dumps({"text_bytes": len(text), "k_deflate_chain_ms": round(best, 3), "text_GBps": round(len(text) / best / 1e6, 2),
                      "members_bytes": int(len(comp)), "ratio": round(len(text) / len(comp), 3),
                      "sample_64MiB": {"zlib_level_6_bytes": z6, "zlib_level_6_one_thread_s": round(t_z6, 2), "host_compressor_bytes": host,
                                       "host_compressor_all_threads_s": round(t_host, 3), "k_deflate_bytes": dev_sample,
                                       "k_deflate_over_zlib6": round(dev_sample / z6, 4)}}))


if __name__ == "__main__":
    main()
