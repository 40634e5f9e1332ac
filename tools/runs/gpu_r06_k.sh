#!/bin/bash
# round 6, call k: a direct table for codes of up to eight bits in k_inflate (PGI_LUT) -- the default build (6 waves per SIMD, one VGPR
# spilled), the same at 5 waves per SIMD (no spill), and without the table -- on 1 GiB of north-star text; 20 000 random members
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06k; mkdir -p $O
S=/tmp/pg_r06k; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> /dev/null
for k in 1 2 3; do
  echo -n "table, 6 waves (1 spill)  "; python tools/inflate_bench.py --file $S/sample.geno.gz | tail -1
  echo -n "table, 5 waves            "; PG_LIBRARY=$R/tools/variants/libpopgen_lut_w5.so python tools/inflate_bench.py --file $S/sample.geno.gz | tail -1
  echo -n "no table, 6 waves         "; PG_LIBRARY=$R/tools/variants/libpopgen_nolut.so python tools/inflate_bench.py --file $S/sample.geno.gz | tail -1
done 2>&1 | tee $O/inflate_lut_ab.txt
rm -rf $S
