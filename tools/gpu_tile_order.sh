cd $GRAFT_REPO_ROOT
for m in 0 1 2; do
  for g in 80 144; do
  BLACKLIGHT_AMD_TILE_ORDER=$m timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --scratch-gib $g 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('order $m scratch $g: Mrays/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'kernels', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'chunks', d['config']['chunks_per_step'])
"
  done
done
