#!/bin/bash
# round 6, call r: k_deflate with the candidates' loads side by side -- tests, rate, the VCF chain's kernels again
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -2
timeout 600 python tools/deflate_bench.py 400000 200 > $O/deflate_bench.json 2> $O/err.txt; cat $O/deflate_bench.json
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
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o vcf --output-format csv -- python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vcf_gz_to_gz_kernel_stats.csv; head -6 $O/vcf_gz_to_gz_kernel_stats.csv | cut -c1-60,150-400
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.csv" -size +2M -delete
