cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
for wl in c2 c3 northstar; do
  ST=5; [ $wl = northstar ] && ST=2
  B="python bench.py --workload $wl --steps $ST --warmup 2 --no-cpu-baseline"
  timeout 150 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o $wl --output-format csv -- $B > gpurun_out/bench_prof_$wl.log 2>&1
  tail -1 gpurun_out/bench_prof_$wl.log | cut -c1-150
  if true; then
    timeout 150 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o $wl --output-format csv -- $B > gpurun_out/pmc_fetch_$wl.log 2>&1
    timeout 150 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o $wl --output-format csv -- $B > gpurun_out/pmc_write_$wl.log 2>&1
    timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_sq -o $wl --output-format csv -- $B > gpurun_out/pmc_sq_$wl.log 2>&1
  fi
done
for wl in c2 c3 northstar c4; do
  timeout 200 python bench.py --workload $wl $( [ $wl = northstar ] && echo "--steps 3 --warmup 1" ) > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err
  tail -c 600 gpurun_out/bench_$wl.json
done
ls gpurun_out/prof_stats gpurun_out/pmc_fetch | head -30
