#!/bin/bash
# round 6, call p: k_deflate at two / three waves per SIMD (1024 / 512 buckets), the kernels of the VCF drop-in's chain by rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06p; mkdir -p $O
( cd genomics_general_amd/csrc && make OUT=/tmp/libpopgen_hb9w3.so EXTRA="-DPGD_HB=9 -DPGD_WAVES=3" > /tmp/build_v.log 2>&1; tail -2 /tmp/build_v.log )
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -2
for k in 1 2; do
timeout 600 python tools/deflate_bench.py 400000 200 > $O/deflate_bench_hb10_w2_$k.json 2> $O/err.txt; cat $O/deflate_bench_hb10_w2_$k.json
PG_LIBRARY=/tmp/libpopgen_hb9w3.so timeout 600 python tools/deflate_bench.py 400000 200 > $O/deflate_bench_hb9_w3_$k.json 2>> $O/err.txt; cat $O/deflate_bench_hb9_w3_$k.json
done
PG_LIBRARY=/tmp/libpopgen_hb9w3.so timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -2
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import vcf_bench
os.makedirs('/tmp/vb', exist_ok=True)
vcf_bench.write_vcf('/tmp/vb/in.vcf', 2000000, 200)
subprocess.check_call([sys.executable, 'tools/bgzip.py', '/tmp/vb/in.vcf', '/tmp/vb/in.vcf.gz'], env=dict(os.environ, PG_BGZF_ZLIB='1'))
PY
OPTS="--skipIndels --minQual 30 --gtf flag=DP min=8 --gtf flag=GQ min=20"
for k in 1 2 3; do PG_TIMING=1 python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS 2>&1 | grep PG_TIMING; done | tee $O/vcf_6GB_gz_to_gz_timing.txt
for k in 1 2; do PG_LIBRARY=/tmp/libpopgen_hb9w3.so PG_TIMING=1 python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS 2>&1 | grep PG_TIMING; done | tee $O/vcf_6GB_gz_to_gz_timing_hb9w3.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_vcf -o vcf -- python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS > /dev/null 2>&1
find /tmp/prof_vcf -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vcf_gz_to_gz_kernel_stats.csv; head -12 $O/vcf_gz_to_gz_kernel_stats.csv
ls -la /tmp/vb
