set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_sr_parity_gpu.py tests/test_bench_mode_gpu.py -m gpu -q -x > gpurun_out/r2_conv_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r2_conv_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r2_ab.sh $1
