#!/bin/bash
# the whole `-m gpu` suite as the driver runs it, on the tree as it stands -> gpurun_out/suite/pytest_gpu_final.log
cd "$(dirname "$0")/.."
OUT=gpurun_out/suite; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rxXf --durations=15 --timeout 600 > $OUT/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_final.log
tail -5 $OUT/pytest_gpu_final.log
