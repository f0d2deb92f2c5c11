set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py tests/test_frame_recovery_gpu.py tests/test_harness_gpu.py -m gpu -q -s > gpurun_out/r2_fr3.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FR bf16|FR losses|FAILED" gpurun_out/r2_fr3.log
timeout -k 10 600 python bench.py --window 2 --scale 4 --height 270 --width 480 --recovery --no-cpu-baseline > gpurun_out/r2_bench_cfg4_full2.json 2> gpurun_out/r2_bench_cfg4_full2.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('gpurun_out/r2_bench_cfg4_full2.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
