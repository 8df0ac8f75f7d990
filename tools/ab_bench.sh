#!/bin/bash
# bench.py (no CPU baseline) against every build_ab/lib*.so
cd "$(dirname "$0")/.."
for f in build_ab/lib*.so; do echo -n "$f: "; THETA_HIP_LIB=$PWD/$f python bench.py --no-cpu-baseline --steps 6 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.4g kernel_ms %.2f'%(d['value'], r['kernel_ms_per_launch']))"; done
