#!/bin/bash
# round 6, call ab: the drivers' `-o out.gz` as BGZF by the library's host threads (freq.py's per-site table against the gzip module)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -n 6 2>&1 | tail -2
timeout 900 python tools/drivers_bench.py 5000000 200 > $O/drivers_bench_gpu.json 2> $O/drivers_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06ab/drivers_bench_gpu.json'))
for k, v in d['drivers'].items():
    print(k[:70].ljust(72), v.get('total_s'), v.get('process_wall_s'), v.get('timing', {}).get('write_s'))
PY
