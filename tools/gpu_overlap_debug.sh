cd "$GRAFT_REPO_ROOT"
export BLACKLIGHT_AMD_DEBUG_COUNTERS=1 WORLDS=8 REPS=1
echo tolerant share; timeout -k 10 200 python3 tools/gpu_tiled_emulation.py gpurun_out/x.json 2>&1 | grep "tail overlap" | tail -2
echo exact share; ARITH=exact timeout -k 10 200 python3 tools/gpu_tiled_emulation.py gpurun_out/x.json 2>&1 | grep "tail overlap" | tail -2
echo whole frame forced; BLACKLIGHT_AMD_TAIL_OVERLAP=1 timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "tail overlap" | tail -3
