#!/bin/bash
# round 6, call ay: the planes chosen on the filled rows (pg_tune_planes from bench.py): the driver's line without tiers, alternating
# PG_PLANE_TRIALS=1 (no choice of planes) and the default, three fresh processes each; then the tests that touch it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ay; mkdir -p $O
for k in 1 2 3; do
  for v in 1 4; do
    PG_PLANE_TRIALS=$v timeout 300 python bench.py --no-tiers --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('placement_trials',{})
print(json.dumps({'PG_PLANE_TRIALS': $v, 'ms_per_step': d['ms_per_step'], 'k_pack3_ms': d['roofline']['avg_launch_ms'], 'frac': d['roofline']['frac'], 'rows_probe_ms': p.get('probe_ms'), 'planes_probe_ms_on_empty_rows': p.get('planes_probe_ms_on_empty_rows'), 'planes_probe_ms': p.get('planes_probe_ms'), 'planes_kept': p.get('planes_kept')}))" | tee -a $O/bench_planes_ab.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x --timeout=600 2>&1 | tail -2
