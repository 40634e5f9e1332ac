#!/bin/bash
# round 6, call af: k_deflate with two tables of recent places (four per 4-byte hash + eight per 12-byte hash) -- tests, fuzz, ratio and
# rate at two / three waves per SIMD, k_inflate on what it writes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06af; mkdir -p $O
( cd genomics_general_amd/csrc && make OUT=/tmp/libpopgen_w3.so EXTRA="-DPGD_WAVES=3" > /tmp/build_w3.log 2>&1 ) &
PG_DEFLATE_FUZZ_SEEDS=1500 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 2>&1 | tail -2
wait
for k in 1 2; do
  timeout 600 python tools/deflate_bench.py 400000 200 2>/dev/null | sed -e "s/^/w2 /"
  PG_LIBRARY=/tmp/libpopgen_w3.so timeout 600 python tools/deflate_bench.py 400000 200 2>/dev/null | sed -e "s/^/w3 /"
done | tee $O/deflate_bench_two_tables.txt
timeout 600 python tools/inflate_by_writer.py 1200000 200 | tee $O/inflate_by_writer.json
