#!/bin/bash
# bench.py --workload polarized1024 once for the library and once per variants/*.so (tools/build_variant.sh), one line each; further
# environment passes through (BLACKLIGHT_AMD_POLCOEF_BLOCKS, switches).   gpurun -- 'bash tools/gpu_polarized_variants.sh'
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python3 bench.py --workload polarized1024 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
run library
for lib in variants/*.so; do
  [ -e "$lib" ] || continue
  BLACKLIGHT_AMD_LIB="$PWD/$lib" run "$lib"
  BLACKLIGHT_AMD_POLCOEF_BLOCKS=12 BLACKLIGHT_AMD_LIB="$PWD/$lib" run "$lib, 12 workgroups per CU"
done
