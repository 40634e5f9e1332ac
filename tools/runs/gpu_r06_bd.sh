#!/bin/bash
# round 6, call bd: the other single-GPU configurations of BASELINE.json (C2 popgen, C3 ABBA-BABA, C4 distMat) as bench lines at the
# round's last commit
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06bd; mkdir -p $O
for w in c2 c3 c4; do
  timeout 600 python bench.py --workload $w --no-tiers --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python - $O/bench_$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['config'].get('workload'), d['value'], d['unit'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))
PY
done
