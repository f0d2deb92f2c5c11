set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests_q.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r2_tests_q.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg2', d['value'], d['ms_per_step'])"; done
python bench.py --window 2 --scale 4 --height 270 --width 480 --recovery --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg4', d['value'], d['ms_per_step'])"
