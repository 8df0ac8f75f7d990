#!/bin/bash
# one gpurun call on the final tree of a round: the whole `-m gpu` suite as the driver runs it, then the part of the measurement set
# that depends on the sieve kernel (bench in the driver's layout, rocprofv3 kernel stats of the same command, PMC passes of the dominant
# kernel, phase cycles of the profiling build)  ->  gpurun_out/final/   (tools/profile_round.sh is the full set; this is ~20 min)
R=${1:-r4}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -rxXf --durations=15 --timeout 600 > $OUT/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_final.log
tail -22 $OUT/pytest_gpu_final.log
cd /tmp
timeout 600 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
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
$ROOT/tools/pmc_kernel.sh gpurun_out/final/pmc_sieve n3_sieve_kernel > $OUT/pmc_sieve.log 2>&1
cp $OUT/pmc_sieve/pmc.json $OUT/pmc_n3_sieve_kernel.json 2>/dev/null
rm -rf $OUT/pmc_sieve
if [ -f $ROOT/build_ab/libprof.so ]; then
  for leg in full_solve_f64 full_solve_f64_tight full_solve_f32 search; do
    echo "== $leg (cycles summed over waves: 0 group tile, 1 parent phase, 2 children phase, 3 queue drain, 4 prefix successor, 5 whole wave, 6 the last level's own expansion; 7 = prefixes walked)"
    THETA_HIP_LIB=$ROOT/build_ab/libprof.so THETA_BENCH_VERBOSE=1 timeout 200 python $ROOT/bench.py --steps 6 --warmup 2 --leg $leg --no-legs --no-cpu-baseline --no-traffic --no-extras 2>&1 >/dev/null | grep "^step"
  done > $OUT/phase_cycles.txt
fi
tail -c 600 $OUT/bench_n1.json
