#!/bin/bash
# bench legs of the main library under environment settings: tools/ab_env.sh "VAR=val" "VAR=val VAR2=val" ...
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/ab; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting THETA_BENCH_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_env$i.json 2> $OUT/bench_env$i.err
  echo "== $setting"
  python - <<PY
import json
d=json.load(open("$OUT/bench_env$i.json"))
print(d['value'], d['ms_per_step'], d['dtype'], d.get('setup_ms_per_step'))
for k,l in d['roofline']['legs'].items():
    print(k,{k2:l[k2] for k2 in ('value','kernel_ms_per_launch','wall_ms_per_launch','frac')})
PY
done
