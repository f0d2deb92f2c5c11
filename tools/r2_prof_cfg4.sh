set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests/test_frame_recovery_gpu.py tests/test_harness_gpu.py "tests/test_ewc_gpu.py::test_adaptive_engine_strength_mode_and_names" -m gpu -q -s > gpurun_out/r2_fr2.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FR bf16|FAILED" gpurun_out/r2_fr2.log
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg4f -o p -- python3 $R/bench.py --window 2 --scale 4 --height 270 --width 480 --recovery --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/cfg4f_prof_bench.json 2> $R/gpurun_out/cfg4f_prof.err && echo PROF_OK
rm -f $R/gpurun_out/prof_cfg4f/p_kernel_trace.csv
