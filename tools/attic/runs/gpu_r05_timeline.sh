#!/bin/bash
# round 5: where a block's 8 - 9 ms go in the T2 run on bgzipped text (8.1 GB of text): PG_TIMELINE of the three threads + the tokenizer's own laps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05tl; mkdir -p $O
S=/tmp/pg_r05_sample; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 10000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> /dev/null
CMDZ=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/sample.geno.gz #; s#$S/sample.geno.csv#$S/out_gz.csv#")
PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDZ > /dev/null 2>&1
PG_TIMELINE=1 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDZ 2> $O/timeline.err > /dev/null
PG_TOK_TRACE=1 PG_TIMING=1 PG_PLACE_TRIALS=1 $CMDZ 2> $O/toktrace.err > /dev/null
python - $O/timeline.err <<'P'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith("PG_TIMELINE "):
        ev=json.loads(ln[12:])
        t0=min(e[2] for e in ev)
        for th,lab,a,b in sorted(ev,key=lambda e:e[2]):
            print("%-10s %-16s %8.2f %8.2f  %6.2f ms"%(th,lab,(a-t0)*1e3,(b-t0)*1e3,(b-a)*1e3))
P
grep PG_TOK_TRACE $O/toktrace.err | head -40
rm -rf $S
