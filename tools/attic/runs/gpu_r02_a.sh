#!/bin/bash
# GPU call A of round 2: instruction-rate microbenchmark, the -m gpu suite, same-box A/B of k_pack2 vs k_pack3, one full bench line.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
nproc > $O/host.txt; free -g >> $O/host.txt; rocm-smi --showmeminfo vram >> $O/host.txt 2>&1
tools/valu_rate > $O/valu_rate.txt 2>&1
tail -32 $O/valu_rate.txt
( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu.log 2>&1
tail -30 $O/pytest_gpu.log
for rep in 1 2 3; do
  for v in PG_NONE=1 PG_PACK2=1; do
    echo "== c2 $v rep $rep"
    env $v timeout 200 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tee -a $O/ab_c2.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d.get('kernel_ms_per_step'))
    elif 'rror' in ln: print(ln.strip())
"
  done
done
for rep in 1 2; do
  for v in PG_NONE=1 PG_PACK2=1; do
    echo "== northstar $v rep $rep"
    env $v timeout 300 python bench.py --workload northstar --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tee -a $O/ab_ns.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d.get('kernel_ms_per_step'))
    elif 'rror' in ln: print(ln.strip())
"
  done
done
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
