# same-box A/B of two builds of libnvq.so with per-kernel times: $1 = the other library, $2.. = kernel-name substrings to print
set -o pipefail
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/$1; shift
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "corr or warp" 2>&1 | tail -2
for i in 1 2; do
  for which in old new; do
    if [ $which = old ]; then export NVQ_LIB=$OLD; else unset NVQ_LIB; fi
    python bench.py --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cfg2 $which', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
cd /tmp && export TMPDIR=/tmp
for which in old new; do
  if [ $which = old ]; then export NVQ_LIB=$OLD; else unset NVQ_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$which -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
  echo "== $which"; python3 - "$which" "$@" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"/tmp/prof_{sys.argv[1]}/p_kernel_stats.csv")))
for r in rows:
    if any(p in r["Name"] for p in sys.argv[2:]):
        print(f"  {r['Name'][:80]:80s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
