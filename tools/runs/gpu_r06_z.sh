#!/bin/bash
# round 6, call z: the kernels of the VCF chain at the last HEAD (rocprofv3 needs the process to end the long way: the shim's os._exit is
# off under a profiler)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06z; mkdir -p $O
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import vcf_bench
os.makedirs('/tmp/vb', exist_ok=True)
vcf_bench.write_vcf('/tmp/vb/in.vcf', 2000000, 200)
subprocess.check_call([sys.executable, 'tools/bgzip.py', '/tmp/vb/in.vcf', '/tmp/vb/in.vcf.gz'], env=dict(os.environ, PG_BGZF_ZLIB='1'))
PY
OPTS="--skipIndels --minQual 30 --gtf flag=DP min=8 --gtf flag=GQ min=20"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o vcf --output-format csv -- python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vcf_gz_to_gz_kernel_stats.csv; head -8 $O/vcf_gz_to_gz_kernel_stats.csv | cut -c1-50,180-400
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.csv" -size +2M -delete
