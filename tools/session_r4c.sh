export TMPDIR=/tmp
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round3.py::test_fp64_sieve_and_full_solve_modes_return_the_lists_of_the_shipped_search -m gpu -x -q --timeout 800 > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
rocprofv3 -L 2>/dev/null | grep -io "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*WRITEBACK[A-Z_]*\|TCC_[A-Z0-9_]*ATOMIC[A-Z0-9_]*\|TCC_NORMAL_[A-Z_]*\|TCC_WRITE[A-Z_0-9]*" | sort -u > $OUT/tcc_counters.txt; wc -l $OUT/tcc_counters.txt
python tools/n2_search_run.py > $OUT/n2_run.txt 2>&1; cat $OUT/n2_run.txt
bash tools/pmc_generic.sh $OUT/pmc_n2_m100 n2_search_kernel -- python $PWD/tools/n2_search_run.py m100_k5 | tail -1
bash tools/pmc_generic.sh $OUT/pmc_n2_m200 n2_search_kernel -- python $PWD/tools/n2_search_run.py m200_k7 | tail -1
