#!/bin/bash
# round 6, call f: the runtime's start-up beside the imports, as wall-clock of whole processes (alternating, ten each, on the mid-size
# `.geno.gz` of bench.py's t2.bgzf leg); ONE gzip stream through the library's own decoder against zlib's and the gzip module's; the
# whole north star with k_crc32 back on the chain
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06f; mkdir -p $O
S=/tmp/pg_r06f; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 5000000 200 > $S/cmd.txt 2> $S/write.err
python tools/bgzip.py $S/sample.geno $S/sample.geno.gz 2> $O/bgzip.txt
CMD=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/sample.geno.gz #")
wall() { local a=$EPOCHREALTIME; "$@" > /dev/null 2>&1; local b=$EPOCHREALTIME; echo "$a $b" | awk '{printf "%.3f\n", $2 - $1}'; }
$CMD > /dev/null 2>&1
for k in 1 2 3 4 5 6 7 8 9 10; do
  e=$(wall $CMD); n=$(PG_EARLY_INIT=0 wall $CMD 2>/dev/null); echo "$e $n"
done > $O/early_init_wall_seconds.txt
export PG_EARLY_INIT
awk '{a+=$1; b+=$2; if (NR==1 || $1<ma) ma=$1; if (NR==1 || $2<mb) mb=$2} END {printf "runtime start-up beside the imports: mean %.3f s (best %.3f); at the first HIP call: mean %.3f s (best %.3f); %d runs each, alternating\n", a/NR, ma, b/NR, mb, NR}' $O/early_init_wall_seconds.txt | tee -a $O/early_init_wall_seconds.txt
# one gzip stream: 5e6 sites x 200 diploids = 4.06 GB of text
gzip -6 -k -c $S/sample.geno > $S/plain.geno.gz
CMDG=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/plain.geno.gz #")
for mode in "PG_GZIP_FAST=1" "PG_GZIP_FAST=0" "PG_GZIP_NATIVE=0"; do
  for k in 1 2; do echo -n "$mode "; env $mode PG_TIMING=1 $CMDG 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"read_s": [0-9.]*\|"text_bytes": [0-9]*\|"context_s": [0-9.]*' | paste - - - -; done
done > $O/gzip_stream_reader.txt; cat $O/gzip_stream_reader.txt
cmp <(PG_GZIP_FAST=1 $CMDG 2>/dev/null; cat $S/sample.geno.csv) <(PG_GZIP_NATIVE=0 $CMDG 2>/dev/null; cat $S/sample.geno.csv) && echo "csv identical" >> $O/gzip_stream_reader.txt
rm -rf $S
PG_NS_KEEP=/tmp/pg_ns_cmd.txt timeout 900 python tools/t2_northstar_bgzf.py 100000000 4 > $O/whole.json 2> $O/whole.err; tail -c 1200 $O/whole.json; echo
rm -rf /tmp/pg_northstar_* /tmp/pg_ns_cmd.txt
du -sh $O
