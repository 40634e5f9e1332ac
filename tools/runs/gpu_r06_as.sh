#!/bin/bash
# round 6, call t (run again as call y at the last HEAD): evidence at the HEAD with the device's VCF parser and k_deflate: the whole -m gpu suite, smoke(), the default bench
# line (now with a VCF leg), the VCF drop-in on 6 GB of VCF (all legs; the host parser beside it), k_deflate alone, the kernels of the
# VCF chain, the drivers end to end
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06as; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --timeout=300 > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -1 | tee $O/smoke.txt
timeout 600 python tools/deflate_bench.py 400000 200 > $O/deflate_bench.json 2> $O/deflate_bench.err; cat $O/deflate_bench.json
VCF_REPS=2 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2> $O/vcf_bench.err; tail -c 3000 $O/vcf_bench_6GB.json; echo
PG_VCF_DEVICE=0 VCF_LEGS=0,2 VCF_REPS=1 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB_host_parser.json 2>> $O/vcf_bench.err; tail -c 900 $O/vcf_bench_6GB_host_parser.json; echo
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import vcf_bench
os.makedirs('/tmp/vb', exist_ok=True)
vcf_bench.write_vcf('/tmp/vb/in.vcf', 2000000, 200)
subprocess.check_call([sys.executable, 'tools/bgzip.py', '/tmp/vb/in.vcf', '/tmp/vb/in.vcf.gz'], env=dict(os.environ, PG_BGZF_ZLIB='1'))
PY
OPTS="--skipIndels --minQual 30 --gtf flag=DP min=8 --gtf flag=GQ min=20"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o vcf --output-format csv -- env PG_FAST_EXIT=0 python VCF_processing/parseVCF.py -i /tmp/vb/in.vcf.gz -o /tmp/vb/o.geno.gz $OPTS > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vcf_gz_to_gz_kernel_stats.csv; head -8 $O/vcf_gz_to_gz_kernel_stats.csv | cut -c1-50,180-400
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.csv" -size +2M -delete
rm -rf /tmp/vb
timeout 900 python tools/drivers_bench.py 5000000 200 > $O/drivers_bench_gpu.json 2> $O/drivers_bench.err; tail -c 300 $O/drivers_bench_gpu.json; echo
timeout 1500 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; tail -c 1800 $O/bench_northstar_default.json; echo
du -sh $O
