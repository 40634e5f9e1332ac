#!/bin/bash
# round 4, call e: the tokenizer in three steps (parse(k) -> submit(k+1) -> collect(k)), four copy streams: tests, then T2 twice
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -15
timeout 600 python tools/tok_bench2.py 2500000 200 > $O/tok_bench2.txt 2>&1; cat $O/tok_bench2.txt
for rep in 1 2 3; do
timeout 1500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_t2_$rep.json 2> $O/bench_t2_$rep.err; tail -c 300 $O/bench_t2_$rep.err
python - $O/bench_t2_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["t2"]; print({k: t.get(k) for k in ("text_GBps", "without_context_creation", "tokenizer_text_GBps", "tokenizer_h2d_GBps", "stages_overlap", "matches_t0")}); print(t.get("seconds"), t.get("error"))
PY
done
