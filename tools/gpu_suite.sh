# the whole -m gpu suite, log under gpurun_out/, then one short bench line
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/gpu_suite.log 2>&1
tail -8 gpurun_out/gpu_suite.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('Mrays/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})
"
