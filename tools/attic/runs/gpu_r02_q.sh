#!/bin/bash
# round 2, call q: where the device tokenizer's time goes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02q; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_kernels.py -m gpu -x -q -k "device_tokenizer" > gpurun_out/r02q/pytest.log 2>&1; tail -5 gpurun_out/r02q/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02q/tok -o tok -- python tools/tok_bench.py 2500000 100 > gpurun_out/r02q/tok.log 2>&1; grep -v "^[WEI]2026" gpurun_out/r02q/tok.log | tail -25
python - <<'PY'
import sqlite3
db = sqlite3.connect('gpurun_out/r02q/tok/tok_results.db')
for r in db.execute("select * from top_kernels limit 8"): print(r)
PY
