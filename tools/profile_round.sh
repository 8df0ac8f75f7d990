#!/bin/bash
# Measurement set of one round, run ON THE GPU BOX:  tools/profile_round.sh r2   -> gpurun_out/prof_$R/ (copy what is to be judged
# into profiles/$R/).
#   1. bench.py (the driver's command)                         -> bench_n1.json
#   2. rocprofv3 --kernel-trace --stats of the same command     -> bench_kernel_stats.csv   (all three legs)
#   3. PMC passes of the dominant kernel, one group per run     -> pmc_n3_sieve_kernel.json (tools/pmc_kernel.sh)
#   4. other configs, materialised operators                    -> other_configs.json, enumerate.json, device_chain.json, batch_ops.json
R=${1:-r2}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
# every launch of the search kernels, in order (the averages of the stats file mix the short bootstrap launches of a job's first
# step with the timed 2^31-candidate launches)
python - <<PY
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "n3_sieve_kernel" in r["Kernel_Name"] or "n3_finish_kernel" in r["Kernel_Name"] or "n3_search_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open("$OUT/bench_search_launches.csv", "w") as o:
        o.write("kernel,grid_size,duration_ms\n")
        for r in rows:
            o.write("%s,%s,%.4f\n" % (r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $OUT/kt
$ROOT/tools/pmc_kernel.sh gpurun_out/prof_$R/pmc_sieve n3_sieve_kernel > $OUT/pmc_sieve.log 2>&1
cp $OUT/pmc_sieve/pmc.json $OUT/pmc_n3_sieve_kernel.json 2>/dev/null
timeout 300 python $ROOT/tools/bench_configs.py > $OUT/other_configs.json 2> $OUT/other_configs.err
timeout 300 python $ROOT/tools/enum_profile.py > $OUT/enumerate.json 2> $OUT/enumerate.err
timeout 300 python $ROOT/tools/device_chain.py 26 > $OUT/device_chain.json 2> $OUT/device_chain.err
timeout 300 python $ROOT/tools/batch_profile.py > $OUT/batch_ops.json 2> $OUT/batch_ops.err
ls -la $OUT
