#!/bin/bash
# round 3, call c: balanced slot programs; pack || pair overlap re-measured with the LDS-staged C kernel (PG_OVERLAP=1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03c; mkdir -p $O
PG_PAIR_TILE=cd timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { tag=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/$tag.json 2> $O/$tag.err
  python - "$O" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('%s/%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print("%-26s ms_per_step %.4f  kernels %s" % (sys.argv[2], d["ms_per_step"], d.get("kernel_ms_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e, open('%s/%s.err' % (sys.argv[1], sys.argv[2])).read()[-800:])
PY
}
run ns_c          northstar PG_PAIR_TILE=c
run ns_cd         northstar PG_PAIR_TILE=cd
run ns_none       northstar PG_PAIR_TILE=none
run ns_c_ovl      northstar PG_PAIR_TILE=c PG_OVERLAP=1
run ns_cd_ovl     northstar PG_PAIR_TILE=cd PG_OVERLAP=1
run ns_none_ovl   northstar PG_PAIR_TILE=none PG_OVERLAP=1
run ns_c_2        northstar PG_PAIR_TILE=c
run ns_c_ovl_2    northstar PG_PAIR_TILE=c PG_OVERLAP=1
run c2_c          c2 PG_PAIR_TILE=c
run c2_none       c2 PG_PAIR_TILE=none
