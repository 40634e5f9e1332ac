#!/bin/bash
# Same-box A/B of library variants (box-to-box and allocation-to-allocation spread of the HBM-bound kernels is +-4 %, so variants
# are only comparable inside one gpurun call).  Build here, run there:
#
#   tools/ab_variants.sh build  NAME1="-DFOO=1" NAME2="-DFOO=2 -DBAR"      # -> ab/libNAME.so (+ ab/libBASE.so = current sources)
#   gpurun -- 'tools/ab_variants.sh run "BASE NAME1 NAME2" "--workload c2 --steps 20 --warmup 3" [ENV=VAL ...]'
#
# `run` swaps each variant into genomics_general_amd/libpopgen_hip.so, runs bench.py (no CPU baseline) REPS times (default 3,
# interleaved) and prints ms/step and the per-kernel breakdown; the original library is restored afterwards.  ab/ is scratch.
set -e
cd "$(dirname "$0")/.."
CS=genomics_general_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -x hip --offload-arch=gfx950"
SRCS="pg_kernels.hip pg_pair2.hip pg_pair_mfma.hip pg_tokenize.hip pg_abi.cpp pg_encode.cpp pg_vcf.cpp pg_comm.cpp"
case "$1" in
build)
  shift; mkdir -p ab
  (cd $CS && /opt/rocm/bin/hipcc $FLAGS $SRCS -shared -o ../../ab/libBASE.so -ldl -lpthread -lz)
  for spec in "$@"; do
    name="${spec%%=*}"; defs="${spec#*=}"
    (cd $CS && /opt/rocm/bin/hipcc $FLAGS $defs $SRCS -shared -o ../../ab/lib$name.so -ldl -lpthread -lz)
    echo "built ab/lib$name.so  ($defs)"
  done ;;
run)
  variants="$2"; args="$3"; shift 3 || true
  cp genomics_general_amd/libpopgen_hip.so /tmp/pg_orig.so
  trap 'cp /tmp/pg_orig.so genomics_general_amd/libpopgen_hip.so' EXIT
  for rep in $(seq 1 ${REPS:-3}); do
    for v in $variants; do
      cp ab/lib$v.so genomics_general_amd/libpopgen_hip.so
      echo -n "== $v rep $rep: "
      env "$@" timeout 150 python bench.py $args --no-cpu-baseline 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('ms_per_step', d['ms_per_step'], d.get('kernel_ms_per_step'))
    elif 'rror' in ln: print(ln.strip())
"
    done
  done ;;
*) sed -n 2,10p "$0" ;;
esac
