# same-box A/B of two builds of libnvq.so: $1 = path of the other library (relative to the repo root)
set -o pipefail
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/$1
for i in 1 2; do
  for which in old new; do
    if [ $which = old ]; then export NVQ_LIB=$OLD; else unset NVQ_LIB; fi
    python bench.py --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg2 $which', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
for which in old new; do
  if [ $which = old ]; then export NVQ_LIB=$OLD; else unset NVQ_LIB; fi
  echo "== $which"; PH_N=8 PH_LD=${PH_LD:-256} timeout -k 10 300 python tools/kernel_phases.py 2>&1 | grep -v amdgpu.ids
done
