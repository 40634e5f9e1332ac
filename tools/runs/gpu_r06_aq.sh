#!/bin/bash
# round 6, call aq: parity at scale at the round's last kernels -- the 1000-seed fuzz of the drivers (HIP engine == stand-in), 60 000 random
# members through k_inflate against zlib, 800 random VCF files, 3000 random texts through k_deflate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06aq; mkdir -p $O
PG_FUZZ_SEEDS=1000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 --timeout=600 2>&1 | tail -1 | tee $O/gpu_fuzz_1000_seeds.txt
timeout 900 python tools/inflate_fuzz.py 60000 7 2>&1 | tail -2 | tee $O/inflate_fuzz_60000_members.txt
PG_VCF_FUZZ_SEEDS=800 timeout 900 python -m pytest tests/test_gpu_vcf.py -q -n 8 --timeout=300 2>&1 | tail -1 | tee $O/vcf_device_parser_fuzz_800_seeds.txt
PG_DEFLATE_FUZZ_SEEDS=3000 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 --timeout=300 2>&1 | tail -1 | tee $O/deflate_fuzz_3000_seeds.txt
