#!/bin/bash
# round 4, call a: window-range sharding on the GPU (multi-rank drivers through PG_COMM=file), bench --strong, then the whole suite
# and the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest_e2e.log 2>&1; tail -15 $O/pytest_e2e.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_e2e.py > $O/pytest.log 2>&1; tail -5 $O/pytest.log
PG_COMM=file timeout 600 python bench.py --gpus 2 --strong --steps 5 --warmup 2 --no-cpu-baseline --no-tiers > $O/bench_strong2.json 2> $O/bench_strong2.err; tail -c 1500 $O/bench_strong2.json
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
