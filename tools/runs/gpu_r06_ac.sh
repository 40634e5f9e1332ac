#!/bin/bash
# round 6, call ac: the device's VCF parser behind ONE gzip stream and behind a pipe
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ac; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vcf.py -q -n 6 2>&1 | tail -3
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import vcf_bench
os.makedirs('/tmp/vb', exist_ok=True)
vcf_bench.write_vcf('/tmp/vb/in.vcf', 1000000, 200)
subprocess.check_call("gzip -6 -c /tmp/vb/in.vcf > /tmp/vb/in_stream.vcf.gz", shell=True)
PY
OPTS="--skipIndels --minQual 30 --gtf flag=DP min=8 --gtf flag=GQ min=20"
( for v in 1 0; do for k in 1 2; do echo "== ONE gzip stream (3 GB of VCF), PG_VCF_DEVICE=$v"; PG_VCF_DEVICE=$( [ $v = 1 ] && echo "" || echo 0 ) PG_TIMING=1 python VCF_processing/parseVCF.py -i /tmp/vb/in_stream.vcf.gz -o /tmp/vb/o.geno.gz $OPTS 2>&1 | grep PG_TIMING | cut -c1-420; done; done
  for v in 1 0; do for k in 1 2; do echo "== a pipe (cat in.vcf |), PG_VCF_DEVICE=$v"; cat /tmp/vb/in.vcf | PG_VCF_DEVICE=$( [ $v = 1 ] && echo "" || echo 0 ) PG_TIMING=1 python VCF_processing/parseVCF.py -o /tmp/vb/o2.geno.gz $OPTS 2>&1 | grep PG_TIMING | cut -c1-420; done; done ) | tee $O/vcf_3GB_gzip_stream_and_pipe.txt
