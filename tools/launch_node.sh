#!/bin/bash
# One bench.py (or any script reading RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) per GPU of this node, without torchrun:
#   tools/launch_node.sh 8 bench.py --gpus 8 --steps 20 --warmup 5
# Rank 0's stdout is the job's (the JSON line); the other ranks' stdout is dropped, every rank's stderr is kept.
# (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` does the same; nothing in the path imports torch.)
set -u
N=$1; shift
cd "$(dirname "$0")/.."
export MASTER_ADDR=${MASTER_ADDR:-127.0.0.1} MASTER_PORT=${MASTER_PORT:-29511} WORLD_SIZE=$N HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 1; r < N; r++)); do
    RANK=$r LOCAL_RANK=$r python "$@" > /dev/null &
    pids+=($!)
done
RANK=0 LOCAL_RANK=0 python "$@"
rc=$?
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
