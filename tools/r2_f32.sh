# fp32-mode A/B: $1 = the other library
set -o pipefail
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/$1
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_sr_parity_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -2
for which in old new; do
  if [ $which = old ]; then export NVQ_LIB=$OLD; else unset NVQ_LIB; fi
  python bench.py --math f32 --batch 2 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('fp32 cfg2 $which', round(d['value'],2), round(d['ms_per_step'],2))"
  echo "== $which small frames"; timeout -k 10 200 python tools/graph_bench.py 2>&1 | grep -v amdgpu.ids
done
