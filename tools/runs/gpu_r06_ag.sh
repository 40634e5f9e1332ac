#!/bin/bash
# round 6, call ag: k_deflate's two tables at several depths (places per 4-byte hash / per 12-byte hash / waves per SIMD)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ag; mkdir -p $O
( cd genomics_general_amd/csrc
  make OUT=/tmp/lib_a2b8w3.so EXTRA="-DPGD_WA=2 -DPGD_WB=8 -DPGD_WAVES=3" > /tmp/b1.log 2>&1 &
  make OUT=/tmp/lib_a2b4w3.so EXTRA="-DPGD_WA=2 -DPGD_WB=4 -DPGD_WAVES=3" > /tmp/b2.log 2>&1 &
  make OUT=/tmp/lib_a4b8w2.so EXTRA="-DPGD_WA=4 -DPGD_WB=8 -DPGD_WAVES=2" > /tmp/b3.log 2>&1 &
  wait )
PG_DEFLATE_FUZZ_SEEDS=600 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 2>&1 | tail -1
for k in 1 2; do
  timeout 600 python tools/deflate_bench.py 400000 200 2>/dev/null | cut -c1-330 | sed -e "s/^/a2b8w2 /"
  for v in a2b8w3 a2b4w3 a4b8w2; do PG_LIBRARY=/tmp/lib_$v.so timeout 600 python tools/deflate_bench.py 400000 200 2>/dev/null | cut -c1-330 | sed -e "s/^/$v /"; done
done | tee $O/deflate_bench_table_depths.txt
for v in a2b8w3 a2b4w3; do PG_LIBRARY=/tmp/lib_$v.so PG_DEFLATE_FUZZ_SEEDS=300 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 2>&1 | tail -1; done
for v in a2b8w3 a2b4w3; do echo $v; PG_LIBRARY=/tmp/lib_$v.so timeout 600 python tools/inflate_by_writer.py 1200000 200 | python -c "import sys,json; d=json.load(sys.stdin); print(d['writers']['k_deflate'])"; done | tee $O/inflate_by_writer_variants.txt
