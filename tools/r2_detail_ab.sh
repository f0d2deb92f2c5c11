set -o pipefail
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/$1
NVQ_LIB=$OLD python bench.py --no-cpu-baseline --detail --steps 5 --warmup 2 2> gpurun_out/detail_old.txt | tail -1 | cut -c1-150
python bench.py --no-cpu-baseline --detail --steps 5 --warmup 2 2> gpurun_out/detail_new.txt | tail -1 | cut -c1-150
python - <<'PY'
import re
def load(p):
    d={}
    for line in open(p):
        m=re.match(r"^(\S+)\s+(n\d+ .*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s*$", line.rstrip())
        if m: d[(m.group(1), m.group(2).strip())]=(int(m.group(3)), float(m.group(4)))
    return d
a,b=load("gpurun_out/detail_old.txt"),load("gpurun_out/detail_new.txt")
keys=sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0))[1]+b.get(k,(0,0))[1]))
ta=tb=0
for k in keys[:45]:
    x,y=a.get(k,(0,0))[1],b.get(k,(0,0))[1]; ta+=x; tb+=y
    print(f"{k[0][:26]:26s} {k[1][:38]:38s} old {x:7.3f} new {y:7.3f}  {('%+.1f%%' % ((y/x-1)*100)) if x and y else ''}")
print("sum top45 old", round(ta,2), "new", round(tb,2))
PY
