#!/bin/bash
# round 4: what NumPy's summation order costs a DRIVER run: 8.1 GB of `.geno` text (10^7 sites x 200 diploids) through popgenWindows.py
# in 50 kb windows (fixed-tree finisher) and in 2 kb windows (k_popdist_np), and the latter with the fixed trees forced
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04t2w; mkdir -p $O
CMD=$(python tools/t2_write_sample.py /tmp/t2w.geno 10000000 200 2> $O/sample.txt); cat $O/sample.txt
for rep in 1 2; do
for w in 50000 2000; do
  C=$(echo "$CMD" | sed "s/-w 50000/-w $w/")
  PG_TIMING=1 PG_PLACE_TRIALS=1 $C 2> $O/t_$w.txt; python - <<PY
import json
ln=[l for l in open("$O/t_$w.txt") if l.startswith("PG_TIMING ")][-1]
t=json.loads(ln[len("PG_TIMING "):]); print("-w $w", {k: round(t[k],3) for k in ("total_s","compute_and_write_s","tokenize_s","context_s") if k in t}, "windows", t.get("windows"))
PY
done
C=$(echo "$CMD" | sed "s/-w 50000/-w 2000/")
PG_POPDIST_TREE=0 PG_TIMING=1 PG_PLACE_TRIALS=1 $C 2> $O/t_2000_fixed.txt; python - <<PY
import json
ln=[l for l in open("$O/t_2000_fixed.txt") if l.startswith("PG_TIMING ")][-1]
t=json.loads(ln[len("PG_TIMING "):]); print("-w 2000 fixed trees", {k: round(t[k],3) for k in ("total_s","compute_and_write_s","tokenize_s","context_s") if k in t})
PY
done
