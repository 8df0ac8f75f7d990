#!/bin/bash
# Measurement set of one round, run ON THE GPU BOX:  tools/profile_round.sh r1
#   1. bench.py                                    -> gpurun_out/prof_$R/bench_n1.json
#   2. rocprofv3 --kernel-trace --stats (same cmd) -> gpurun_out/prof_$R/kernel_stats.csv
#   3. PMC passes, one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass)      -> gpurun_out/prof_$R/pmc_*.csv  (summarised by tools/pmc_summary.py)
R=${1:-r1}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 8 --warmup 2"
cd /tmp
if [ -z "$PMC_ONLY" ]; then
timeout 600 $BENCH > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -o pmc -- $BENCH --no-cpu-baseline > /dev/null 2> $OUT/pmc$i.err
    f=$(find $OUT/pmc$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $ROOT/tools/pmc_summary.py $f n3_search_kernel > $OUT/pmc_$i.json
    rm -rf $OUT/pmc$i
done
rm -rf $OUT/kt
python - <<PY
import json,glob
m={}
for f in sorted(glob.glob("$OUT/pmc_*.json")):
    try: m.update(json.load(open(f)))
    except Exception as e: print(f, e)
json.dump(m, open("$OUT/pmc_n3_search_kernel.json","w"), indent=1)
PY
rm -f $OUT/pmc_[0-9].json $OUT/pmc[0-9].err
ls -la $OUT
