# GPU box: rocprofv3 kernel statistics of the cfg5 step (bf16, graph replay, batch 8) in this tree and in a second tree (_r3)
set -o pipefail
R=$GRAFT_REPO_ROOT
export CFG5_ONLY=bf16,auto,8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg5_head -o p -- python3 $R/tools/cfg5_step.py > $R/gpurun_out/cfg5_head.txt 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg5_r3 -o p -- python3 $R/_r3/tools/cfg5_step.py > $R/gpurun_out/cfg5_r3.txt 2>&1 && \
rm -f $R/gpurun_out/cfg5_head/p_kernel_trace.csv $R/gpurun_out/cfg5_r3/p_kernel_trace.csv; tail -1 $R/gpurun_out/cfg5_head.txt $R/gpurun_out/cfg5_r3.txt
