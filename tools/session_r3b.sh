#!/bin/bash
# round 3, GPU session B: the FP64 sieve (parity with the shipped search), the new bench line, the render generator's rate
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_zzz_render.py "tests/test_gpu_round2.py::test_sieve_and_fused_search_kernels_return_identical_lists" tests/test_gpu_wide.py -m gpu -q -x -rxXf --durations=10 --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
THETA_BENCH_VERBOSE=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_short.json 2> $OUT/bench_short.err
tail -c 2500 $OUT/bench_short.json; tail -30 $OUT/bench_short.err
timeout 200 python tools/enum_profile.py > $OUT/enumerate_render.json 2> $OUT/enumerate_render.err
grep "n2_" $OUT/enumerate_render.err
