#!/bin/bash
# round 6, call ae: k_inflate on `.geno` text as zlib, the host compressor and k_deflate write it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r06ae; mkdir -p $O
timeout 600 python tools/inflate_by_writer.py 1200000 200 > $O/inflate_by_writer.json 2> $O/err.txt; cat $O/inflate_by_writer.json; tail -3 $O/err.txt
