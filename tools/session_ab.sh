#!/bin/bash
# quick tests on the main build, then the bench's three legs on the main build and on every build_ab/lib*.so named in $ABS
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/ab
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$NO_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_round3.py::test_fp64_sieve_and_full_solve_modes_return_the_lists_of_the_shipped_search "tests/test_gpu_round2.py::test_sieve_and_fused_search_kernels_return_identical_lists" tests/test_gpu_round2.py::test_sieve_end_to_end_against_the_fused_driver -m gpu -q -x -rxXf --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
[ -n "$NO_TESTS" ] || tail -4 $OUT/pytest_gpu.log
for lib in main $ABS; do
  if [ $lib = main ]; then unset THETA_HIP_LIB; else export THETA_HIP_LIB=$ROOT/build_ab/lib$lib.so; fi
  THETA_BENCH_VERBOSE=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err
  echo "== $lib"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$lib.json"))
print(d['value'], d['ms_per_step'], d['dtype'])
for k,l in d['roofline']['legs'].items():
    print(k,{k2:l[k2] for k2 in ('value','kernel_ms_per_launch','step_kernel_ms','redo_kernel_ms','flop_per_candidate','frac')})
PY
  grep "^step" $OUT/bench_$lib.err | head -14 | tail -3
done
