#!/bin/bash
# PC sampling of tools/phase_profile.py (run ON THE GPU BOX): tools/pc_sample.sh [method] [unit] [interval]
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/pcs
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${1:-host_trap} --pc-sampling-unit ${2:-time} --pc-sampling-interval ${3:-100} \
    --kernel-trace --output-format csv -d $OUT -o pcs -- python $ROOT/tools/phase_profile.py > $OUT/run.log 2>&1
echo rc=$?
tail -5 $OUT/run.log
find $OUT -type f | head -20
for f in $(find $OUT -name '*pc_sampling*.csv'); do echo $f; wc -l $f; head -5 $f; done
