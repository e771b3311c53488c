cd $GRAFT_REPO_ROOT
for g in 40 80 136; do
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --scratch-gib $g 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('scratch $g GiB: Mrays/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'kernels', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'chunks', d['config']['chunks_per_step'])
"
done
