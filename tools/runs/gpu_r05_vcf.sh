#!/bin/bash
# round 5: the VCF drop-in on a bgzipped VCF with the members inflated on the device: its GPU tests, then tools/vcf_bench.py
# (no reference on the GPU box: the reference leg of the tool runs in the build container)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05vcf; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q -k "vcf or inflate_members" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/vcf_bench.py ${VCF_SITES:-400000} 200 > $O/vcf_bench.json 2> $O/vcf_bench.err; tail -c 3000 $O/vcf_bench.json; tail -3 $O/vcf_bench.err
for mb in 32 64; do PG_STREAM_BYTES=$((mb<<20)) timeout 900 python tools/vcf_bench.py ${VCF_SITES:-400000} 200 > $O/vcf_bench_${mb}MiB.json 2>> $O/vcf_bench.err; python - $O/vcf_bench_${mb}MiB.json $mb <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print("blocks of", sys.argv[2], "MiB:", {k[:40]:(v["seconds"], v["timing"].get("blocks_inflated_on_device")) for k,v in d["legs"].items()})
P
done
