# Debug aid: two ranks of bench.py on ONE GPU.  RCCL refuses (or stalls on) duplicate devices, so this only shows how far the
# rendezvous and the bootstrap get (NCCL_DEBUG=INFO); the real N>1 run needs one GPU per rank.
cd /root/repo
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29911 WORLD_SIZE=2 NCCL_DEBUG=INFO
T=${1:-40}
(RANK=0 LOCAL_RANK=0 timeout $T python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --workload tiny > gpurun_out/two_r0.log 2>&1; echo "r0 rc=$?" >> gpurun_out/two_r0.log) &
(RANK=1 LOCAL_RANK=1 timeout $T python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --workload tiny > gpurun_out/two_r1.log 2>&1; echo "r1 rc=$?" >> gpurun_out/two_r1.log) &
wait
tail -25 gpurun_out/two_r0.log | cut -c1-250; echo ----; tail -12 gpurun_out/two_r1.log | cut -c1-250
