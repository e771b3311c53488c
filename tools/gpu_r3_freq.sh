cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_tolerant.py tests/test_gpu_true_color.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/gpu_other_configs.py > gpurun_out/r3_other_configs.json 2> gpurun_out/r3_other_configs.err
python -c "
import json; d=json.load(open('gpurun_out/r3_other_configs.json'))
for k,v in d.items(): print(k, {kk: (round(vv,2) if isinstance(vv,float) else vv) for kk,vv in v.items()})
"
