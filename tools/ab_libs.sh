# GPU box: bench.py alternated between two builds of the library (continual-learning-..._amd/libnvq_old.so and libnvq.so), same box
P=$GRAFT_REPO_ROOT/continual-learning-for-dynamic-video-quality-enhancement_amd
for l in libnvq_old.so libnvq.so libnvq_old.so libnvq.so; do
  NVQ_LIB=$P/$l python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],2), ' '.join(f'{n[:22]}={v}' for n, v in list(k.items())[:5]))" $l
done
