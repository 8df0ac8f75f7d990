#!/bin/bash
# after a change of bench.py only: the modes test, smoke(), the bench in the driver's layout and the rocprofv3 kernel stats of the same command -> gpurun_out/cert/
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/cert; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "fp64_sieve_and_full_solve_modes" > $OUT/pytest_modes.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_modes.log
tail -4 $OUT/pytest_modes.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
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
if [ -f $ROOT/build_ab/libprof.so ]; then
  for leg in full_solve_f64_tight_certified; do
    echo "== $leg (cycles summed over waves: 0 group tile, 1 parent phase, 2 children phase, 3 queue drain, 4 prefix successor, 5 whole wave, 6 the last level's own expansion; 7 = prefixes walked)"
    THETA_HIP_LIB=$ROOT/build_ab/libprof.so THETA_BENCH_VERBOSE=1 timeout 200 python $ROOT/bench.py --steps 6 --warmup 2 --leg $leg --no-legs --no-cpu-baseline --no-traffic --no-extras 2>&1 >/dev/null | grep "^step"
  done > $OUT/phase_cycles_certified.txt
fi
python - <<PY
import json
d=json.load(open("$OUT/bench_n1.json"))
print(d["value"], d["ms_per_step"])
for k,l in d['roofline']['legs'].items():
    print("%-32s %.3e cand/s  kernel %.2f ms  evals/cand %.4f  flop/cand %.1f frac %.3f" % (k, l['value'], l['kernel_ms_per_launch'], l['newton_iters_per_candidate'], l['flop_per_candidate'], l['frac']))
PY
