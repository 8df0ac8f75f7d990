#!/bin/bash
# the whole GPU suite as the driver runs it, then the measurement set of the round
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/full
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -rxXf --durations=15 --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout 2400 tools/profile_round.sh ${1:-r3} > $OUT/profile_round.log 2>&1
tail -c 1500 $ROOT/gpurun_out/prof_${1:-r3}/bench_n1.json
