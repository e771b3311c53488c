#!/bin/bash
# Timeline of a render with BL_SWITCH_TAIL_OVERLAP (printed by the library under BLACKLIGHT_AMD_DEBUG_COUNTERS): an eighth of the
# benchmark frame in both tiers, configuration 2, the whole benchmark frame.   gpurun -- 'bash tools/gpu_overlap_debug.sh'
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export BLACKLIGHT_AMD_DEBUG_COUNTERS=1 BLACKLIGHT_AMD_TAIL_OVERLAP=1 WORLDS=8 REPS=1
echo "eighth of the benchmark frame, tolerant tier (tools/gpu_tiled_emulation.py, WORLDS=8):"; timeout -k 10 200 python3 tools/gpu_tiled_emulation.py gpurun_out/x.json 2>&1 | grep "tail overlap" | tail -2
echo "... exact tier:"; ARITH=exact timeout -k 10 200 python3 tools/gpu_tiled_emulation.py gpurun_out/x.json 2>&1 | grep "tail overlap" | tail -2
echo "configuration 2 (bench.py --workload formula512):"; timeout -k 10 200 python3 bench.py --workload formula512 --steps 2 --warmup 1 2>&1 | grep "tail overlap" | tail -2
echo "whole benchmark frame (bench.py):"; timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "tail overlap" | tail -2
