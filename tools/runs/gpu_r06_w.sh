#!/bin/bash
# round 6, call w: k_deflate at three / four / five waves per SIMD (168 / 128 / 96 registers, 2 / 13 / 24 spilled)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06w; mkdir -p $O
for w in 4 5; do ( cd genomics_general_amd/csrc && make OUT=/tmp/libpopgen_w$w.so EXTRA="-DPGD_WAVES=$w" > /tmp/build_w$w.log 2>&1 ) & done; wait
for k in 1 2; do
  timeout 600 python tools/deflate_bench.py 400000 200 2>/dev/null | cut -c1-140 | sed -e "s/^/w3 /"
  for w in 4 5; do PG_LIBRARY=/tmp/libpopgen_w$w.so timeout 600 python tools/deflate_bench.py 400000 200 2>/dev/null | cut -c1-140 | sed -e "s/^/w$w /"; done
done | tee $O/deflate_bench_waves_per_simd.txt
PG_LIBRARY=/tmp/libpopgen_w4.so PG_DEFLATE_FUZZ_SEEDS=200 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 2>&1 | tail -1
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import vcf_bench
os.makedirs('/tmp/vb', exist_ok=True)
vcf_bench.write_vcf('/tmp/vb/in.vcf', 2000000, 200)
subprocess.check_call([sys.executable, 'tools/bgzip.py', '/tmp/vb/in.vcf', '/tmp/vb/in.vcf.gz'], env=dict(os.environ, PG_BGZF_ZLIB='1'))
PY
OPTS="--skipIndels --minQual 30 --gtf flag=DP min=8 --gtf flag=GQ min=20"
run() { for k in 1 2 3; do env "$@" PG_TIMING=1 python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS 2>&1 | grep -E "PG_TIMING" | sed -e 's/"bgzf".*//' | sed -e 's/.*device_submit_s/device_submit_s/'; done; }
( echo "== w3"; run A=1; echo "== w4"; run PG_LIBRARY=/tmp/libpopgen_w4.so; echo "== w5"; run PG_LIBRARY=/tmp/libpopgen_w5.so ) | tee $O/vcf_6GB_gz_to_gz_deflate_waves.txt
