# quick iteration on the GPU box: GPU tests + short bench (no CPU baseline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('Mrays/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), 'kernels', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'roofline frac', round(d['roofline']['frac'],3))
"
