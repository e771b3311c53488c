#!/bin/bash
# The emulated eight-rank share (tools/gpu_tiled_emulation.py, WORLDS=1,8) once per library under variants/*.so and once for the
# library in the tree, on one box:   gpurun -- 'bash tools/gpu_emu_ab.sh [ENV=VALUE ...]'   (the assignments apply to every run)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
emu() {
  WORLDS=1,8 REPS=7 timeout -k 10 300 python3 tools/gpu_tiled_emulation.py gpurun_out/quad_emu.json > /dev/null 2> gpurun_out/quad_emu.err
  python3 -c "
import json
d = json.load(open('gpurun_out/quad_emu.json'))
for w in (1, 8):
    x = d['world_%d' % w]; r = max(x['ranks'], key=lambda r: r['median_ms'])
    print('$1', 'emulated ranks', w, 'median ms', round(x['frame_ms_median'], 2), 'geodesic', round(r['geodesic'], 2), 'coefficient', round(r['shade'], 2))
" | tee -a gpurun_out/emu_ab.txt
}
: > gpurun_out/emu_ab.txt
for lib in variants/*.so; do
  [ -e "$lib" ] || continue
  BLACKLIGHT_AMD_LIB="$PWD/$lib" emu "$lib"
done
emu tree
