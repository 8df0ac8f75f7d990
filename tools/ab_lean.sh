#!/bin/bash
# round 5: the bench's full-solve legs for the regular build and build_ab/lib{lean0,third48,third32}.so (UNIT=n3_sieve tools/ab_build.sh lean0 -DSV_LEAN_FIRST=0 ...)
cd "$(dirname "$0")/.."
for lib in "" build_ab/liblean0.so build_ab/libthird48.so build_ab/libthird32.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  THETA_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 12 --warmup 3 --no-traffic --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-regular}', 'value %.4g' % d['value'])
for k,v in d['roofline']['legs'].items(): print('   ', k, round(v['kernel_ms_per_launch'],2), round(v['newton_iters_per_candidate'],3), v['survivors'])
"
done
