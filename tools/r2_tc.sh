set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_frame_recovery_gpu.py -m gpu -q -x > gpurun_out/r2_tc_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r2_tc_tests.log
[ $rc -eq 0 ] || exit $rc
for tic in 1 0; do NVQ_FR_TIME_IN_CHANNELS=$tic python bench.py --window 2 --scale 4 --height 270 --width 480 --recovery --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg4 tic=$tic', d['value'], d['ms_per_step'])"; done
