#!/bin/bash
# Measurement set of one round, run ON THE GPU BOX:  tools/profile_round.sh r3   -> gpurun_out/prof_$R/ (copy what is to be judged
# into profiles/$R/).
#   1. bench.py as the driver runs it (--steps 20 --warmup 5)   -> bench_n1.json
#   2. rocprofv3 --kernel-trace --stats of the same command      -> bench_kernel_stats.csv (all legs), bench_search_launches.csv
#   3. PMC passes of the dominant kernel, one group per run      -> pmc_n3_sieve_kernel.json (tools/pmc_kernel.sh)
#   4. phase cycles of the sieve kernel (build_ab/libprof.so)    -> phase_cycles.txt
#   5. the riders, and rocprofv3 --kernel-trace --stats of them  -> riders.json, riders_kernel_stats.csv
R=${1:-r3}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
# every launch of the search kernels, in order (the averages of the stats file mix the short bootstrap / first-slice launches
# with the full-size ones)
python - <<PY
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "n3_sieve_kernel" in r["Kernel_Name"] or "n3_finish_kernel" in r["Kernel_Name"] or "n3_search_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open("$OUT/bench_search_launches.csv", "w") as o:
        o.write("kernel,grid_size,duration_ms\n")
        for r in rows:
            o.write("%s,%s,%.4f\n" % (r["Kernel_Name"].split("(")[0].replace("void ", "").replace(",", ";"), r.get("Grid_Size", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $OUT/kt
$ROOT/tools/pmc_kernel.sh gpurun_out/prof_$R/pmc_sieve n3_sieve_kernel > $OUT/pmc_sieve.log 2>&1
cp $OUT/pmc_sieve/pmc.json $OUT/pmc_n3_sieve_kernel.json 2>/dev/null
if [ -f $ROOT/build_ab/libprof.so ]; then
  for leg in full_solve_f64 full_solve_f64_tight full_solve_f64_tight_certified full_solve_f64_l2_certified full_solve_f32 search; do
    echo "== $leg (cycles summed over waves: 0 group tile, 1 parent phase, 2 children phase, 3 queue drain, 4 prefix successor, 5 whole wave, 6 the last level's own expansion; 7 = prefixes walked)"
    THETA_HIP_LIB=$ROOT/build_ab/libprof.so THETA_BENCH_VERBOSE=1 timeout 300 python $ROOT/bench.py --steps 6 --warmup 2 --leg $leg --no-legs --no-cpu-baseline --no-traffic --no-extras 2>&1 >/dev/null | grep "^step"
  done > $OUT/phase_cycles.txt
fi
# round 5: the branch and bound on BASELINE configs 3 and 4 (end to end through do_optimization_single), and its kernels under rocprofv3
timeout 300 python -u $ROOT/tools/bnb_run.py c3 c4 c5 > $OUT/bnb_configs_3_4.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kb -o kb -- python $ROOT/tools/bnb_run.py c3 c4 c5 > /dev/null 2> $OUT/kb.err
cp $(find $OUT/kb -name '*kernel_stats.csv' | head -1) $OUT/bnb_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kb
timeout 600 python $ROOT/tools/riders.py > $OUT/riders.json 2> $OUT/riders.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kr -o kr -- python $ROOT/tools/riders.py > /dev/null 2> $OUT/kr.err
cp $(find $OUT/kr -name '*kernel_stats.csv' | head -1) $OUT/riders_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kr
for shape in "131072 512 200" "65536 64 200"; do timeout 100 python $ROOT/tools/scorer_probe.py $shape; done > $OUT/scorer_probe.txt 2>&1
# round 4: the n=2 search alone (with and without the dismissal by the lower bound) and its PMC passes; where the sieve's write traffic
# comes from, launch by launch; the plain scorer over record lengths
(timeout 100 python $ROOT/tools/n2_search_run.py; echo "-- THETA_N2_NO_DISMISS=1"; THETA_N2_NO_DISMISS=1 timeout 100 python $ROOT/tools/n2_search_run.py) > $OUT/n2_search.txt 2>&1
bash $ROOT/tools/pmc_generic.sh gpurun_out/prof_$R/pmc_n2 n2_search_kernel -- python $ROOT/tools/n2_search_run.py m100_k5 > /dev/null 2>&1
cp $OUT/pmc_n2/pmc.json $OUT/pmc_n2_search.json 2>/dev/null
THETA_N2_NO_DISMISS=1 bash $ROOT/tools/pmc_generic.sh gpurun_out/prof_$R/pmc_n2nd n2_search_kernel -- python $ROOT/tools/n2_search_run.py m100_k5 > /dev/null 2>&1
cp $OUT/pmc_n2nd/pmc.json $OUT/pmc_n2_search_no_dismiss.json 2>/dev/null
PMC_GROUPS="WRITE_SIZE;FETCH_SIZE;TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum;SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD;TCC_EA0_ATOMIC_sum TCC_WRITE_sum" bash $ROOT/tools/pmc_generic.sh gpurun_out/prof_$R/pmc_w n3_sieve_kernel -- python $ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-legs --no-traffic --no-extras > /dev/null 2>&1
cp $OUT/pmc_w/pmc.json $OUT/pmc_sieve_writes_per_launch.json 2>/dev/null
timeout 200 python $ROOT/tools/plain_shapes.py > $OUT/plain_shapes.txt 2>&1
rm -rf $OUT/pmc_n2 $OUT/pmc_n2nd $OUT/pmc_w $OUT/pmc_sieve
ls -la $OUT
