#!/bin/bash
# phase cycles (build_ab/libprof.so: -DSV_PROF) and PMC counters of the sieve kernel, headline leg and the other legs
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
for leg in full_solve_f64 full_solve_f32 search; do
  THETA_HIP_LIB=$ROOT/build_ab/libprof.so THETA_BENCH_VERBOSE=1 timeout 300 python bench.py --steps 6 --warmup 2 --leg $leg --no-legs --no-cpu-baseline --no-traffic --no-extras > $OUT/phase_$leg.json 2> $OUT/phase_$leg.err
  grep "^step" $OUT/phase_$leg.err | tail -6
done
tools/pmc_kernel.sh gpurun_out/prof/pmc_f64 n3_sieve_kernel > $OUT/pmc_f64.log 2>&1; tail -2 $OUT/pmc_f64.log
