set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_sr_parity_gpu.py tests/test_bench_mode_gpu.py -m gpu -q -x > gpurun_out/r2_tests_k.log 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/r2_tests_k.log
bash tools/r2_prof_cfg2.sh
