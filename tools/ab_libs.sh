#!/bin/bash
# the bench's headline leg (and the coarse one) for the regular build and every build_ab/lib<NAME>.so given:  tools/ab_libs.sh unr1 unr3 ...
cd "$(dirname "$0")/.."
for name in "" "$@"; do
  lib=${name:+build_ab/lib$name.so}
  THETA_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 12 --warmup 3 --no-traffic --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
l=d['roofline']['legs']
print('${name:-regular}', 'value %.4g' % d['value'], ' '.join('%s %.2f' % (k, v['kernel_ms_per_launch']) for k, v in l.items()))
"
done
