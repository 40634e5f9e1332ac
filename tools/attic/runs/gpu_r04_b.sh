#!/bin/bash
# round 4, call b: what the host of the GPU box really offers (CPUs, quota, memory, /tmp), the device tokenizer with self-paced
# staging threads (pread), the three-stage T2 pipeline, the bench line with the 20 GB T2 sample and the CPU worker sweep
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; mkdir -p $O
{ echo "nproc: $(nproc)  nproc --all: $(nproc --all)"; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())";
  echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /proc/pressure/cpu 2>/dev/null; free -g | head -2; df -h /tmp | tail -1;
  lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"; } > $O/host.txt 2>&1; cat $O/host.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 python tools/tok_bench2.py 2500000 200 > $O/tok_bench2.txt 2>&1; cat $O/tok_bench2.txt
timeout 1500 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04b/bench_default.json").read().strip().splitlines()[-1])
print(json.dumps({k: d.get(k) for k in ("value", "ms_per_step", "t2", "t2_vs_cpu")}, indent=1)[:5000])
c = d["cpu_baseline"]; print(c["value"], c["cores"], c["host_cpus"], json.dumps(c["worker_sweep"]))
PY
