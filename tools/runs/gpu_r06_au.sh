#!/bin/bash
# round 6, call at: small windows -- popgenWindows.py on 5e6 sites x 200 diploids with windows of 50 kb ... 100 bp (and 100-site windows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06au; mkdir -p $O
python tools/t2_write_sample.py /tmp/s.geno 5000000 200 > /dev/null
PG_BGZF_ZLIB=1 python tools/bgzip.py /tmp/s.geno /tmp/s.geno.gz; rm /tmp/s.geno
POPS=$(python - <<'PY'
names=["s%d"%d for d in range(200)]
print(" ".join("-p pop%d %s" % (k, ",".join(names[k*50:(k+1)*50])) for k in range(4)))
PY
)
for w in 50000 5000 1000 200 100; do
  echo "== coordinate windows of $w"; PG_TIMING=1 python popgenWindows.py -g /tmp/s.geno.gz -o /tmp/o.csv -f phased -w $w -m 10 $POPS 2>&1 | grep PG_TIMING | python -c "
import sys, json
t = json.loads(sys.stdin.read().split('PG_TIMING ', 1)[1])
print({k: (round(t[k], 4) if isinstance(t[k], float) else t[k]) for k in ('total_s', 'context_s', 'tokenize_s', 'windows_s', 'prep_wait_s', 'main_stats_s', 'main_refine_s', 'main_format_s', 'windows', 'windows_recomputed_in_numpy_order') if k in t})"
done | tee $O/small_windows.txt
echo "== sites windows of 100"; PG_TIMING=1 python popgenWindows.py -g /tmp/s.geno.gz -o /tmp/o.csv -f phased --windType sites -w 100 -m 10 $POPS 2>&1 | grep PG_TIMING | cut -c1-600 | tee -a $O/small_windows.txt
wc -l /tmp/o.csv
