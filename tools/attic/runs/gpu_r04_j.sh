#!/bin/bash
# round 4, call j: the pack kernel's bare traffic with its store bursts gathered into chip-wide time slots (tools/ubench/pack_rw.hip)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04j; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/ubench/pack_rw.hip -o /tmp/pack_rw 2>/dev/null || cp gpurun_variants/pack_rw /tmp/pack_rw
for k in 1 2; do timeout 300 /tmp/pack_rw | tee -a $O/pack_rw.txt; done
