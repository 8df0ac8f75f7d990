#!/bin/bash
# A/B of one environment switch on the GPU box:  tools/session_env_ab.sh VAR VALUE_A VALUE_B [bench args]  (driver layout: 20 steps, 5 warm-up)
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/envab
mkdir -p $OUT
export TMPDIR=/tmp
VAR=$1; A=$2; B=$3; shift 3
for v in $A $B; do
  env $VAR=$v THETA_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras "$@" > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  echo "== $VAR=$v"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$v.json"))
print(d['value'], d['ms_per_step'], d['dtype'])
for k,l in d['roofline']['legs'].items():
    print(k,{k2:l[k2] for k2 in ('value','kernel_ms_per_launch','step_kernel_ms','redo_kernel_ms','survivors','frac')})
PY
done
