#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/pb; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_round4.py::test_prefixes_finished_by_the_prefix_bound_change_no_list tests/test_gpu_round3.py::test_fp64_sieve_and_full_solve_modes_return_the_lists_of_the_shipped_search tests/test_gpu_round2.py::test_sieve_and_fused_search_kernels_return_identical_lists tests/test_gpu_round2.py::test_sieve_end_to_end_against_the_fused_driver -m gpu -q -x -rxXf --timeout 900 > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
for pb in 1 0; do
  echo "== prefix bound $pb"
  THETA_N3_PREFIX_BOUND=$pb THETA_BENCH_VERBOSE=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $OUT/bench_$pb.json 2> $OUT/bench_$pb.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$pb.json"))
for k,l in d['roofline']['legs'].items():
    print("%-22s %.3e cand/s  kernel %.2f ms  evals/cand %.4f" % (k, l['value'], l['kernel_ms_per_launch'], l['newton_iters_per_candidate']))
print(d.get('riders',{}).get('config5_search'))
PY
done
grep "^step" $OUT/bench_1.err | tail -3
