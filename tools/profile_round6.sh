#!/bin/bash
# Round 6's measurement set, run ON THE GPU BOX (tools/profile_round.sh without the parts whose code did not change this round: the
# n=2 search's PMC passes, the sieve's writes per launch, the plain scorer's shapes -- profiles/r5/ holds those):
#   tools/profile_round6.sh  -> gpurun_out/prof_r6/ (copy what is to be judged into profiles/r6/)
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r6
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
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
$ROOT/tools/pmc_kernel.sh gpurun_out/prof_r6/pmc_sieve n3_sieve_kernel > $OUT/pmc_sieve.log 2>&1
cp $OUT/pmc_sieve/pmc.json $OUT/pmc_n3_sieve_kernel.json 2>/dev/null
rm -rf $OUT/pmc_sieve
if [ -f $ROOT/build_ab/libprof.so ]; then
  for leg in full_solve_f64_tight_certified full_solve_f64_l2_certified full_solve_f64; do
    echo "== $leg (cycles summed over waves: 0 group tile, 1 parent phase, 2 children phase, 3 queue drain, 4 prefix successor, 5 whole wave, 6 the last level's own expansion; 7 = prefixes walked)"
    THETA_HIP_LIB=$ROOT/build_ab/libprof.so THETA_BENCH_VERBOSE=1 timeout 200 python $ROOT/bench.py --steps 6 --warmup 2 --leg $leg --no-legs --no-cpu-baseline --no-traffic --no-extras 2>&1 >/dev/null | grep "^step"
  done > $OUT/phase_cycles.txt
fi
# the whole-space search (configs 3 / 4 / 5's shape, four small spaces against the walk), and its kernels under rocprofv3
timeout 200 python -u $ROOT/tools/mix_lines_probe.py all > $OUT/mix_whole_spaces.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/km -o km -- python $ROOT/tools/mix_trace.py > $OUT/mix_trace.txt 2> $OUT/km.err
cp $(find $OUT/km -name '*kernel_stats.csv' | head -1) $OUT/mix_kernel_stats.csv 2>/dev/null
rm -rf $OUT/km
timeout 600 python $ROOT/tools/riders.py > $OUT/riders.json 2> $OUT/riders.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kr -o kr -- python $ROOT/tools/riders.py > /dev/null 2> $OUT/kr.err
cp $(find $OUT/kr -name '*kernel_stats.csv' | head -1) $OUT/riders_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kr
ls -la $OUT
