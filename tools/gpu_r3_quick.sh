# quick iteration: parity subset + short bench (tolerant + exact)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tolerant.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/r3_quick.err | tee gpurun_out/r3_quick.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('Mrays/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'chunks', d['config']['chunks_per_step'], 'exact', round(d['exact_tier']['value'],2), {k: round(v,1) for k,v in d['exact_tier']['kernel_ms_per_step'].items()})
"
