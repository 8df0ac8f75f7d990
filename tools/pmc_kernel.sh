#!/bin/bash
# PMC passes for one kernel of the bench, run ON THE GPU BOX:  tools/pmc_kernel.sh <out-dir> <kernel-substring> [env assignments...]
#   e.g. tools/pmc_kernel.sh gpurun_out/pmc_sieve n3_sieve_kernel THETA_N3_SIEVE=1
# One counter group per rocprofv3 run (--pmc only, no trace options); summarised per launch by tools/pmc_summary.py.
OUT=$1; KERNEL=$2; shift 2
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
BENCH="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-legs --no-traffic --no-extras $PMC_BENCH_ARGS"
cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/p$i -o pmc -- $BENCH > /dev/null 2> $ROOT/$OUT/p$i.err
    f=$(find $ROOT/$OUT/p$i -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $ROOT/tools/pmc_summary.py $f $KERNEL > $ROOT/$OUT/pmc_$i.json
    rm -rf $ROOT/$OUT/p$i
done
python - <<PY
import json,glob
m={}
for f in sorted(glob.glob("$ROOT/$OUT/pmc_*.json")):
    try: m.update(json.load(open(f)))
    except Exception as e: print(f, e)
json.dump(m, open("$ROOT/$OUT/pmc.json","w"), indent=1)
print(json.dumps({k:(v.get("mean_per_launch") if isinstance(v,dict) else v) for k,v in m.items()}))
PY
rm -f $ROOT/$OUT/pmc_[0-9].json
