for v in 0 1; do
THETA_N3_SECOND=$v python bench.py --steps 10 --warmup 3 --no-traffic --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['roofline']['legs'].items(): print('second=$v', k, round(v['kernel_ms_per_launch'],2), round(v['newton_iters_per_candidate'],3), v['survivors'])
"
done
