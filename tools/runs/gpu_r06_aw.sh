#!/bin/bash
# round 6, call aw: the parity sweeps once more at the round's last commit, several times as many seeds -- 3000 random command lines of the
# drivers (HIP engine == stand-in), 200 000 random members through k_inflate against zlib (another seed), 5000 random VCF files through the
# device parser against the host parser, 20 000 random texts through k_deflate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06aw; mkdir -p $O
PG_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 --timeout=600 2>&1 | tail -1 | tee $O/gpu_fuzz_3000_seeds.txt
timeout 900 python tools/inflate_fuzz.py 200000 11 2>&1 | tail -2 | tee $O/inflate_fuzz_200000_members.txt
PG_VCF_FUZZ_SEEDS=5000 timeout 900 python -m pytest tests/test_gpu_vcf.py -q -n 8 --timeout=300 2>&1 | tail -1 | tee $O/vcf_device_parser_fuzz_5000_seeds.txt
PG_DEFLATE_FUZZ_SEEDS=20000 timeout 900 python -m pytest tests/test_gpu_deflate.py -q -n 8 --timeout=300 2>&1 | tail -1 | tee $O/deflate_fuzz_20000_seeds.txt
