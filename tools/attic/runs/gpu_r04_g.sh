#!/bin/bash
# round 4, call g: bench with eight ranks on one GPU (c5_share behind the headline, small total), the strong mode, and the default line
# with the two-rank T2 run
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "bench_starts" > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -12
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04g/bench.json").read().strip().splitlines()[-1])
t = d["t2"]; print(t.get("text_GBps"), t.get("without_context_creation"), t.get("two_ranks_one_gpu"), t.get("error"))
PY
