#!/bin/bash
# bench legs of one library: tools/ab_one.sh NAME  (NAME = main or a build_ab/libNAME.so)
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/ab; mkdir -p $OUT; export TMPDIR=/tmp
for lib in "$@"; do
  if [ $lib = main ]; then unset THETA_HIP_LIB; else export THETA_HIP_LIB=$ROOT/build_ab/lib$lib.so; fi
  THETA_BENCH_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err
  echo "== $lib"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$lib.json"))
print(d['value'], d['ms_per_step'], d['dtype'], d.get('setup_ms_per_step'))
for k,l in d['roofline']['legs'].items():
    print(k,{k2:l[k2] for k2 in ('value','kernel_ms_per_launch','step_kernel_ms','flop_per_candidate','frac')})
PY
done
