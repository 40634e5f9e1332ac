#!/bin/bash
# round 4, call l: interpreter switch interval of the driver while the ingestion thread runs (50 us against 500 us): T2 from packed cells and from text
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04l; mkdir -p $O
for si in 0.00005 0.0005 0.000005; do
  echo "== PG_SWITCH_INTERVAL=$si" | tee -a $O/t2.txt
  PG_SWITCH_INTERVAL=$si PG_BENCH_PGENO_ONLY_NONE=1 timeout 900 python tools/t2_pgeno_bench.py 25000000 200 2>&1 | grep "none run" | tee -a $O/t2.txt
done
for si in 0.00005 0.0005; do
  echo "== text, PG_SWITCH_INTERVAL=$si" | tee -a $O/t2.txt
  PG_SWITCH_INTERVAL=$si timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$si.json 2> $O/bench_$si.err
  python - $O/bench_$si.json <<'PY' | tee -a $O/t2.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d["t2"]
print(t.get("text_GBps"), t.get("without_context_creation"), t.get("seconds")); print("packed", t.get("packed", {}).get("sites_per_sec"), t.get("packed", {}).get("without_context_creation"), t.get("packed", {}).get("error"))
PY
done
