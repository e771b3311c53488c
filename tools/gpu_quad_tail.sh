#!/bin/bash
# The quad tail (bl_geodesic_quad_kernel) measured on one box: configuration 2, the benchmark frame and the emulated eight-rank
# share - as shipped, with the quad tail (BLACKLIGHT_AMD_QUAD_TAIL), with the coefficient kernel beside the last rays
# (BLACKLIGHT_AMD_TAIL_OVERLAP), and for every library under variants/*.so.
#   gpurun -- 'bash tools/gpu_quad_tail.sh [ENV=VALUE ...]'      e.g. BLACKLIGHT_AMD_PARK_BELOW=32 AFTER="0 8 32" (values of BLACKLIGHT_AMD_PARK_AFTER)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out
OUT=gpurun_out/quad_tail.txt
: > "$OUT"
line() {
  python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['config']['workload'][:30], 'Mrays/s', round(d['value'], 3), 'ms', round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})
" | tee -a "$OUT"
}
emu() {
  WORLDS=1,8 REPS=5 timeout -k 10 300 python3 tools/gpu_tiled_emulation.py gpurun_out/quad_emu.json > /dev/null 2> gpurun_out/quad_emu.err
  python3 -c "
import json
d = json.load(open('gpurun_out/quad_emu.json'))
for w in (1, 8):
    x = d['world_%d' % w]; r = max(x['ranks'], key=lambda r: r['median_ms'])
    print('$1', 'emulated ranks', w, 'median ms', round(x['frame_ms_median'], 2), 'efficiency', round(x['strong_scaling_efficiency'], 3), 'geodesic', round(r['geodesic'], 2), 'coefficient', round(r['shade'], 2))
" | tee -a "$OUT"
}
all() {
  timeout -k 10 300 python3 bench.py --workload formula512 --steps 5 --warmup 2 2>/dev/null | line "$1"
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line "$1"
  emu "$1"
}
all "no-tail"
for after in ${AFTER:-32}; do
  BLACKLIGHT_AMD_QUAD_TAIL=1 BLACKLIGHT_AMD_PARK_AFTER=$after all "quad tail after=$after"
  BLACKLIGHT_AMD_TAIL_OVERLAP=1 BLACKLIGHT_AMD_PARK_AFTER=$after all "overlap after=$after"
done
for lib in variants/*.so; do
  [ -e "$lib" ] || continue
  BLACKLIGHT_AMD_LIB="$PWD/$lib" all "$lib"
done
