#!/bin/bash
# Measurements of the materialised generator, run ON THE GPU BOX: throughput table, rocprofv3 kernel stats and the
# HBM write counter of the same command -> gpurun_out/enum_$1/
R=${1:-r1}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/enum_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 5 300 python $ROOT/tools/enum_profile.py > $OUT/enumerate.json 2> $OUT/enumerate.err
CMD="python $ROOT/tools/enum_burst_probe.py 50 6 28"     # 3 launches of 2^28 candidates
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/probe_under_rocprof.txt 2> $OUT/kt.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/enum_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kt
for grp in WRITE_SIZE FETCH_SIZE; do
    timeout -k 5 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$grp -o pmc -- $CMD > /dev/null 2> $OUT/pmc_$grp.err
    f=$(find $OUT/pmc_$grp -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $ROOT/tools/pmc_summary.py $f n3_enumerate_burst_kernel > $OUT/pmc_$grp.json
    rm -rf $OUT/pmc_$grp
done
ls -la $OUT
