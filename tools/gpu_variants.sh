# A/B of library builds under variants/*.so (built with BLACKLIGHT_AMD_EXTRA_FLAGS=-D...): one bench line each
cd $GRAFT_REPO_ROOT
for lib in variants/*.so; do
  BLACKLIGHT_AMD_LIB=$PWD/$lib timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --arithmetic ${ARITH:-tolerant} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', 'Mrays/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})
"
done
