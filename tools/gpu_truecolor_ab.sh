#!/bin/bash
# bench.py --workload truecolor1024x64 with bl_shade_fused2_kernel's factors instantiation (default) and with the general fused kernel
# (BLACKLIGHT_AMD_GENERAL_FUSED=1), one line each.   gpurun -- 'bash tools/gpu_truecolor_ab.sh'
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python3 bench.py --workload truecolor1024x64 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'switches', d.get('switches'))"; }
run default
BLACKLIGHT_AMD_GENERAL_FUSED=1 run general_fused
for lib in variants/*.so; do
  [ -e "$lib" ] || continue
  BLACKLIGHT_AMD_LIB="$PWD/$lib" run "$lib"
done
