# one gpurun call: the new round-2 tests, then bench lines (cfg2 with the per-shape table, cfg4 SR part)
set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_bench_mode_gpu.py tests/test_full_size_gpu.py tests/test_dp_gpu.py tests/test_ewc_gpu.py -m gpu -x -q -s > gpurun_out/r2_tests_a.log 2>&1 && echo TESTS_OK && \
python bench.py --detail > gpurun_out/r2_bench_cfg2.json 2> gpurun_out/r2_bench_cfg2.err && echo BENCH2_OK && \
python bench.py --window 2 --scale 4 --height 270 --width 480 --detail > gpurun_out/r2_bench_cfg4.json 2> gpurun_out/r2_bench_cfg4.err && echo BENCH4_OK
tail -5 gpurun_out/r2_tests_a.log
