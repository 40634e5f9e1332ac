#!/bin/bash
# round 4, call n: the whole -m gpu suite, smoke, and the driver's bench command as it stands at the end of the round
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1200 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04n/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "vs_baseline", "dtype", "data")})
print(d["roofline"]); c = d["cpu_baseline"]; print(c["value"], c["cores"], c["kind"], c.get("reference_calibration"))
t = d["t2"]; print(t.get("text_GBps"), t.get("without_context_creation"), t.get("packed", {}).get("sites_per_sec"), t.get("two_ranks_one_gpu", {}).get("rank_bytes_share"), d.get("t2_vs_cpu"))
PY
