#!/bin/bash
# First GPU session of a round, run ON THE GPU BOX in ONE gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'tools/first_session.sh r3'
# Every step has its own timeout and writes under gpurun_out/first_$R/, so that a step that fails or hangs costs its own
# output only.  1. the GPU suite as the driver runs it (staged tests report XPASS / XFAIL), 2. the n=2 generator with and
# without the render kernel, 3. the plain / masked scorer probes, 4. the measurement set of the round (tools/profile_round.sh).
R=${1:-r3}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/first_$R
mkdir -p $OUT
export TMPDIR=/tmp
free -g | head -2 > $OUT/box.txt; nproc >> $OUT/box.txt
timeout 1200 python -m pytest tests -m gpu -q -rxX --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
timeout 200 python tools/enum_profile.py > $OUT/enumerate_default.json 2> $OUT/enumerate_default.err
THETA_N2_ENUM_RENDER=1 timeout 200 python tools/enum_profile.py > $OUT/enumerate_render.json 2> $OUT/enumerate_render.err
grep "n2_" $OUT/enumerate_default.err $OUT/enumerate_render.err
for shape in "131072 512 200" "65536 64 200" "16384 512 200"; do timeout 100 python tools/scorer_probe.py $shape; done > $OUT/scorer_probe.txt 2>&1
timeout 200 python tools/device_chain.py 26 > $OUT/device_chain.json 2> $OUT/device_chain.err
cat $OUT/scorer_probe.txt
timeout 1500 tools/profile_round.sh $R > $OUT/profile_round.log 2>&1
tail -3 $ROOT/gpurun_out/prof_$R/bench_n1.json
