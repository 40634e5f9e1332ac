#!/bin/bash
# round 6, call bb: the -m gpu suite, smoke() and the default bench line at the round.s last commit
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06bb; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --timeout=300 > $O/pytest.log 2>&1; grep -E "passed|failed|Error|^E " $O/pytest.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -1 | tee $O/smoke.txt
timeout 1500 python bench.py > $O/bench_northstar_default.json 2> $O/bench_northstar_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06bb/bench_northstar_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','dtype')})
print(d['roofline'])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['cpu_baseline']['cores'])
t2=d['t2']; print('t2', t2.get('text_GBps'), t2['bgzf'].get('text_GBps'), t2['bgzf_whole_workload']['seconds'], t2['bgzf_whole_workload']['windows_per_sec'])
print('vcf', d['vcf'].get('device_parser'), d['vcf'].get('device_over_host_parser'))
PY
