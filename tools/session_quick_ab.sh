#!/bin/bash
# bench legs on the main build and on build_ab/lib<name>.so for every name in $ABS (no tests): value, kernel ms, evaluations per candidate
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/qab
mkdir -p $OUT
for lib in main $ABS; do
  if [ $lib = main ]; then unset THETA_HIP_LIB; else export THETA_HIP_LIB=$ROOT/build_ab/lib$lib.so; fi
  timeout 500 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err
  echo "== $lib"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$lib.json"))
for k,l in d['roofline']['legs'].items():
    print("%-22s %.3e cand/s  kernel %.2f ms  evals/cand %.4f  flop/cand %.1f frac %.3f" % (k, l['value'], l['kernel_ms_per_launch'], l['newton_iters_per_candidate'], l['flop_per_candidate'], l['frac']))
PY
done
