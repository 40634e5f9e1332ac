#!/bin/bash
# round 6, call g: ONE gzip stream -- chunks side by side (pg_par_gunzip.h) against the serial decoder, zlib and the gzip module, as the
# reader alone and through popgenWindows.py; 4.06 GB and 16 GB of text
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06g; mkdir -p $O
S=/tmp/pg_r06g; mkdir -p $S
python tools/t2_write_sample.py $S/sample.geno 20000000 200 > $S/cmd.txt 2> $S/write.err
gzip -6 -k -c $S/sample.geno > $S/plain.geno.gz; ls -l $S | tee $O/files.txt
cat > $S/reader.py <<'PY'
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from genomics_general_amd import _lib
L = _lib.lib()
buf = np.zeros((1 << 30) + (16 << 20), dtype=np.uint8)
best = 1e9
for rep in range(2):
    h = C.c_void_p(); _lib.check(L.pg_gzip_open(sys.argv[1].encode(), C.byref(h)))
    got, comp, eof = C.c_int64(0), C.c_int(0), C.c_int(0); n = 0; t = time.time()
    while True:
        _lib.check(L.pg_gzip_read_lines(h, C.c_void_p(buf.ctypes.data), buf.size, 1 << 30, C.byref(got), C.byref(comp), C.byref(eof)))
        if got.value == 0: break
        n += got.value
    dt = time.time() - t; L.pg_gzip_close(h); best = min(best, dt)
print("%-34s reader alone: %.2f s = %.2f GB/s of text (%d bytes)" % (os.environ.get("TAG", ""), best, n / best / 1e9, n))
PY
for mode in "PG_GZIP_THREADS=16" "PG_GZIP_THREADS=8" "PG_GZIP_THREADS=4" "PG_GZIP_THREADS=1" "PG_GZIP_FAST=0"; do env $mode TAG="$mode" python $S/reader.py $S/plain.geno.gz; done 2>&1 | tee $O/gzip_reader_alone.txt
PG_GZIP_TRACE=1 PG_GZIP_THREADS=16 python $S/reader.py $S/plain.geno.gz 2>&1 | grep TRACE | head -12 | tee $O/gzip_trace_16_threads.txt
CMDG=$(cat $S/cmd.txt | sed "s#$S/sample.geno #$S/plain.geno.gz #")
for mode in "PG_GZIP_THREADS=16" "PG_GZIP_THREADS=1" "PG_GZIP_FAST=0" "PG_GZIP_NATIVE=0"; do
  for k in 1 2; do echo -n "$mode "; env $mode PG_TIMING=1 $CMDG 2>&1 | grep PG_TIMING | grep -o '"total_s": [0-9.]*\|"read_s": [0-9.]*\|"text_bytes": [0-9]*\|"context_s": [0-9.]*\|"tokenize_s": [0-9.]*' | paste - - - - -; done
done | tee $O/gzip_through_popgenwindows.txt
PG_GZIP_THREADS=16 $CMDG 2>/dev/null; cp $S/sample.geno.csv $S/a.csv; PG_GZIP_NATIVE=0 $CMDG 2>/dev/null; cmp $S/a.csv $S/sample.geno.csv && echo "csv identical (chunks side by side vs the gzip module)" | tee -a $O/gzip_through_popgenwindows.txt
rm -rf $S
du -sh $O
