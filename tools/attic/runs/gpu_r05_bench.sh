#!/bin/bash
# round 5: the driver's command (`python bench.py`) with the whole-workload BGZF leg; wall time of the command
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05bench; mkdir -p $O
S=$(date +%s.%N)
python bench.py > $O/bench.json 2> $O/bench.err
E=$(date +%s.%N)
echo "wall seconds of python bench.py: $(echo "$E - $S" | bc)" | tee $O/wall.txt
wc -l $O/bench.json
python - <<'P'
import json
b=json.loads(open("gpurun_out/r05bench/bench.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"]["frac"])
t=b["tiers"]["t2"] if "tiers" in b else b["t2"]
print(json.dumps(t.get("bgzf_whole_workload"))[:1500])
print({k:(v.get("text_GBps") if isinstance(v,dict) else v) for k,v in t.items() if k in ("bgzf","gz","packed")}, t.get("text_GBps"))
P
tail -3 $O/bench.err
