#!/bin/bash
# round 3, GPU session A: the whole GPU suite (un-staged), render-generator dump, operator probes, the bench as the driver runs it
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/box.txt; free -g | head -2 >> $OUT/box.txt
timeout 1500 python -m pytest tests -m gpu -q -rxXf --durations=20 --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -60 $OUT/pytest_gpu.log
timeout 300 python tools/render_debug.py > $OUT/render_debug.txt 2>&1
cat $OUT/render_debug.txt
timeout 200 python tools/enum_profile.py > $OUT/enumerate_default.json 2> $OUT/enumerate_default.err
for shape in "131072 512 200" "65536 64 200"; do timeout 100 python tools/scorer_probe.py $shape; done > $OUT/scorer_probe.txt 2>&1
timeout 200 python tools/device_chain.py 26 > $OUT/device_chain.json 2> $OUT/device_chain.err
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 3000 $OUT/bench_n1.json
