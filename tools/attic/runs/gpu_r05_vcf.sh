#!/bin/bash
# round 5: the VCF drop-in on a bgzipped VCF (members inflated on the device / by the host threads): its GPU tests, then
# tools/vcf_bench.py on 1.2 GB and on 6 GB of VCF text (no reference on the GPU box: the reference leg of the tool runs in the build
# container)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05vcf; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q -k "vcf or inflate_members" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/vcf_bench.py 400000 200 > $O/vcf_bench.json 2> $O/vcf_bench.err; tail -c 3000 $O/vcf_bench.json; tail -3 $O/vcf_bench.err
VCF_LEGS=0,1 VCF_REPS=2 timeout 900 python tools/vcf_bench.py 2000000 200 > $O/vcf_bench_6GB.json 2>> $O/vcf_bench.err; cat $O/vcf_bench_6GB.json
