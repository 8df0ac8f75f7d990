#!/bin/bash
# PMC passes of one kernel of an arbitrary command, run ON THE GPU BOX:  tools/pmc_generic.sh OUT KERNEL -- command...
OUT=$1; KERNEL=$2; shift 3
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
cd /tmp
i=0
# (PMC_GROUPS="A B;C D" replaces the default counter groups: one rocprofv3 run per group)
DEFAULT_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS;SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR;SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD;FETCH_SIZE;WRITE_SIZE"
IFS=';' read -ra GRPS <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for grp in "${GRPS[@]}"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/p$i -o pmc -- "$@" > /dev/null 2> $ROOT/$OUT/p$i.err
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
print(json.dumps({k:(v.get("max") if isinstance(v,dict) else v) for k,v in m.items()}))
PY
rm -f $ROOT/$OUT/pmc_[0-9].json
