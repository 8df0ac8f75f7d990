cd /root/repo; mkdir -p gpurun_out/rccl2
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 WORLD_SIZE=2 THETA_BENCH_NDEV=1 THETA_COMM_TIMEOUT_S=60 NCCL_DEBUG=WARN
for rk in 0 1; do
  RANK=$rk LOCAL_RANK=$rk timeout 240 python bench.py --gpus 2 --steps 3 --warmup 1 --batch $((1<<29)) > gpurun_out/rccl2/out$rk.txt 2> gpurun_out/rccl2/err$rk.txt &
done
wait
for rk in 0 1; do echo "== rank $rk"; tail -c 1500 gpurun_out/rccl2/out$rk.txt; tail -15 gpurun_out/rccl2/err$rk.txt; done
