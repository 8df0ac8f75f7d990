#!/bin/bash
# quick GPU iteration: the sieve parity tests + the bench (no CPU baseline / traffic / extras)
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/quick
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py::test_fp64_sieve_and_full_solve_modes_return_the_lists_of_the_shipped_search "tests/test_gpu_round2.py::test_sieve_and_fused_search_kernels_return_identical_lists" tests/test_gpu_round2.py::test_sieve_end_to_end_against_the_fused_driver -m gpu -q -x -rxXf --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
THETA_BENCH_VERBOSE=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras $BENCH_ARGS > $OUT/bench_short.json 2> $OUT/bench_short.err
python - <<PY
import json
d=json.load(open("$OUT/bench_short.json"))
print(d['value'], d['ms_per_step'], d['dtype'])
for k,l in d['roofline']['legs'].items():
    print(k,{k2:l[k2] for k2 in ('value','kernel_ms_per_launch','step_kernel_ms','survivors','fallback_candidates','redo_kernel_ms','flop_per_candidate','achieved','frac','newton_iters_per_candidate','dismissed_fraction')})
PY
