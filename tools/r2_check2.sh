# one gpurun call: the whole GPU suite, then the cfg4 bench line with the recovery head
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r2_tests_all.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r2_tests_all.log
timeout -k 10 600 python bench.py --window 2 --scale 4 --height 270 --width 480 --recovery --detail > gpurun_out/r2_bench_cfg4_full.json 2> gpurun_out/r2_bench_cfg4_full.err; echo "bench rc=$?"; tail -3 gpurun_out/r2_bench_cfg4_full.err
