# BASELINE configs[4] on one GPU: the whole train_continual.py --strategy ewc task sequence (wall time), then the same under
# rocprofv3 (short: 2 tasks x 1 epoch) to count what autograd adds around the fused penalty gradient
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/cfg5 && cd /tmp/cfg5
T0=$(date +%s.%N)
python $R/experiments/train_continual.py --strategy ewc > $R/gpurun_out/r02_cfg5_ewc.log 2> $R/gpurun_out/r02_cfg5_ewc.err; echo "ewc rc=$?"
T1=$(date +%s.%N); echo "wall $(python -c "print(round($T1 - $T0, 1))") s (4 tasks x 5 epochs x 200 samples, batch 16, F=64 N=8, 64x64, incl. process start-up)" | tee -a $R/gpurun_out/r02_cfg5_ewc.log
tail -4 $R/gpurun_out/r02_cfg5_ewc.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg5 -o p -- python3 $R/experiments/train_continual.py --strategy ewc --tasks 2 --epochs 1 > $R/gpurun_out/r02_cfg5_prof.log 2>&1; echo "prof rc=$?"
rm -f $R/gpurun_out/prof_cfg5/p_kernel_trace.csv
