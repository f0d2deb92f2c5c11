# one gpurun call: whole GPU suite, smoke, the fp32 parity-mode line, a 2-rank gloo rehearsal of bench.py's DP path on the one GPU
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02_tests_all.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02_tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r02_smoke.log
timeout -k 10 600 python bench.py --math f32 --batch 2 --no-cpu-extras > gpurun_out/r02_fp32_bench.json 2> gpurun_out/r02_fp32_bench.err; echo "fp32 bench rc=$?"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --batch 2 --steps 3 --warmup 1 --no-kernel-timer > gpurun_out/r02_dp2_gloo.json 2> gpurun_out/r02_dp2_gloo.err; echo "dp2 rc=$?"; tail -c 600 gpurun_out/r02_dp2_gloo.json
