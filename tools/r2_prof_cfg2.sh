set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2b -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/cfg2b_prof_bench.json 2> $R/gpurun_out/cfg2b_prof.err && echo PROF_OK
rm -f $R/gpurun_out/prof_cfg2b/p_kernel_trace.csv
