#!/bin/bash
# A/B of bench legs across library builds, ON THE GPU BOX:  tools/ab_legs.sh "leg1 leg2" "lib1.so lib2.so"  (paths relative to the repo)
cd "$(dirname "$0")/.."
for leg in $1; do
  for lib in $2; do
    echo "== $leg $lib"
    THETA_HIP_LIB=$PWD/$lib tools/guard.sh 120 40 python bench.py --steps 20 --warmup 5 --leg $leg --no-legs --no-cpu-baseline --no-traffic --no-extras 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); g=d['roofline']['legs'][d['config']['leg']]
        print('%.4g cand/s  %.2f ms/step  evaluations per candidate %.3f  survivors %d' % (d['value'], d['ms_per_step'], g['newton_iters_per_candidate'], g['survivors']))
    elif 'Error' in l or 'error' in l: print(l.strip())
"
  done
done
