#!/bin/bash
# round 4, call c: host thread pools sized from the CPUs the container may use (cgroup quota), eight staging threads at most, the
# windows' mid positions computed by the formatting thread: T2 through the bench and tok_bench2 again
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04c; mkdir -p $O
python -c "from genomics_general_amd import _lib; print('usable cpus', _lib.usable_cpus())"
timeout 600 python tools/tok_bench2.py 2500000 200 > $O/tok_bench2.txt 2>&1; cat $O/tok_bench2.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_t2_$rep.json 2> $O/bench_t2_$rep.err; tail -c 300 $O/bench_t2_$rep.err
python - $O/bench_t2_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["t2"]; print({k: t.get(k) for k in ("text_GBps", "without_context_creation", "tokenizer_text_GBps", "tokenizer_h2d_GBps", "stages_overlap", "matches_t0")}); print(t.get("seconds"), t.get("error"))
PY
done
