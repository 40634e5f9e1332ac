#!/bin/bash
# round 3, call x: `--windType cat` with more ranks than lines on the GPU (ranks without rows still take part in the exchanges)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03x; mkdir -p $O
zcat tests/golden/holes.geno.gz | head -3 > $O/two.geno
python distMat.py -g $O/two.geno -f phased --windType cat --outFormat raw -o $O/one.out 2> $O/one.err
for r in 0 1 2; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=3 MASTER_ADDR=127.0.0.1 MASTER_PORT=39555 PG_COMM=file PG_RDZV_FILE=$PWD/$O/rdzv PG_COMM_TIMEOUT=60 \
    python distMat.py -g $O/two.geno -f phased --windType cat --outFormat raw -o $O/three.out 2> $O/three_$r.err &
done
wait
cmp $O/one.out $O/three.out && echo "identical: $(wc -c < $O/three.out) bytes"; tail -2 $O/three_1.err | cut -c1-200
