#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for leg in search full_solve_f32; do
  PMC_BENCH_ARGS="--leg $leg" tools/pmc_kernel.sh gpurun_out/pmc/$leg n3_sieve_kernel > gpurun_out/pmc/$leg.log 2>&1
  tail -1 gpurun_out/pmc/$leg.log
done
