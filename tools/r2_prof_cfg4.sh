set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q > gpurun_out/r2_ops.log 2>&1; echo "ops tests rc=$?"; tail -1 gpurun_out/r2_ops.log
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg4f -o p -- python3 $R/bench.py --window 2 --scale 4 --height 270 --width 480 --recovery --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/cfg4f_prof_bench.json 2> $R/gpurun_out/cfg4f_prof.err && echo PROF_OK
rm -f $R/gpurun_out/prof_cfg4f/p_kernel_trace.csv
