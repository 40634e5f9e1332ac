#!/bin/bash
# round 6, call ah: the default k_deflate (two places per 4-byte hash + eight per 12-byte hash, three waves per SIMD): tests, fuzz, rate,
# k_inflate on its files, the VCF drop-in end to end
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ah; mkdir -p $O
PG_DEFLATE_FUZZ_SEEDS=3000 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 2>&1 | tail -1 | tee $O/deflate_fuzz_3000_seeds.txt
timeout 600 python tools/deflate_bench.py 400000 200 > $O/deflate_bench.json 2>/dev/null; cat $O/deflate_bench.json
timeout 600 python tools/inflate_by_writer.py 1200000 200 > $O/inflate_by_writer.json 2>/dev/null; cat $O/inflate_by_writer.json
VCF_LEGS=0,2 VCF_REPS=3 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2> $O/vcf_bench.err; cut -c1-900 $O/vcf_bench_6GB.json; echo
timeout 600 python -m pytest tests/test_gpu_vcf.py -q -n 6 2>&1 | tail -1
