cd /root/repo; mkdir -p gpurun_out/cnt
for lib in cnt prof; do
for leg in full_solve_f64 search; do
  echo "== $lib $leg"
  THETA_HIP_LIB=$PWD/build_ab/lib$lib.so THETA_BENCH_VERBOSE=1 timeout 300 python bench.py --steps 4 --warmup 2 --leg $leg --no-legs --no-cpu-baseline --no-traffic --no-extras 2>&1 >/dev/null | grep "^step"
done; done | tee gpurun_out/cnt/counts.txt
