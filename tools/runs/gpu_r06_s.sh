#!/bin/bash
# round 6, call s: the VCF chain with k_deflate on the second stream and blocks of 512 MB; A/B of both
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06s; mkdir -p $O
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import vcf_bench
os.makedirs('/tmp/vb', exist_ok=True)
vcf_bench.write_vcf('/tmp/vb/in.vcf', 2000000, 200)
subprocess.check_call([sys.executable, 'tools/bgzip.py', '/tmp/vb/in.vcf', '/tmp/vb/in.vcf.gz'], env=dict(os.environ, PG_BGZF_ZLIB='1'))
PY
OPTS="--skipIndels --minQual 30 --gtf flag=DP min=8 --gtf flag=GQ min=20"
run() { for k in 1 2 3; do t0=$(date +%s.%N); env "$@" PG_TIMING=1 python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS 2>&1 | grep -E "PG_TIMING" | sed -e 's/"bgzf".*//' ; t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s"; done; }
( echo "== default (512 MB blocks, k_deflate on the second stream)"; run A=1
  echo "== PG_DEFLATE_STREAM=0"; run PG_DEFLATE_STREAM=0
  echo "== PG_VCF_DEVICE_BYTES=128 MB"; run PG_VCF_DEVICE_BYTES=134217728
  echo "== PG_VCF_DEVICE_BYTES=128 MB PG_DEFLATE_STREAM=0"; run PG_VCF_DEVICE_BYTES=134217728 PG_DEFLATE_STREAM=0
  echo "== PG_VCF_DEVICE_BYTES=256 MB"; run PG_VCF_DEVICE_BYTES=268435456
  echo "== PG_VCF_DEVICE_BYTES=1 GB"; run PG_VCF_DEVICE_BYTES=1073741824 ) 2>&1 | tee $O/vcf_6GB_gz_to_gz_block_size_and_stream_ab.txt
python -c "
import gzip,sys
a=gzip.open('/tmp/vb/o.geno.gz','rb').read()
print(len(a), a.count(b'\n'))
"
echo "== plain vcf -> geno"; for k in 1 2; do t0=$(date +%s.%N); env PG_TIMING=1 python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf -o /tmp/vb/o.geno $OPTS 2>&1 | grep -E "PG_TIMING" | sed -e 's/"bgzf".*//'; t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s"; done | tee $O/vcf_6GB_plain_to_plain.txt
