#!/bin/bash
# BL_SWITCH_SPLIT_LONG (rays predicted long stepped by the quad kernel beside the others) against the shipped path, one box:
# the emulated shares of the benchmark frame, configuration 2, the benchmark frame.   gpurun -- 'bash tools/gpu_split_long.sh [band ...]'
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export REPS=5
OUT=gpurun_out/split_long.txt
: > "$OUT"
emu() {
  timeout -k 10 300 python3 tools/gpu_tiled_emulation.py gpurun_out/emu_x.json > /dev/null 2> gpurun_out/emu_x.err
  python3 -c "
import json
d = json.load(open('gpurun_out/emu_x.json'))
print('$1', [(w, round(d['world_%d' % w]['frame_ms_median'], 2), round(max(r['geodesic'] for r in d['world_%d' % w]['ranks']), 2)) for w in (1, 2, 4, 8)])" | tee -a "$OUT"
}
line() {
  python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['config']['workload'][:24], 'Mrays/s', round(d['value'], 3), 'ms', round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})" | tee -a "$OUT"
}
emu shipped
timeout -k 10 200 python3 bench.py --workload formula512 --steps 5 --warmup 2 2>/dev/null | line shipped
for band in "${@:-0.07}"; do
  export BLACKLIGHT_AMD_SPLIT_LONG=1 BLACKLIGHT_AMD_LONG_BAND=$band
  emu "split band=$band"
  timeout -k 10 200 python3 bench.py --workload formula512 --steps 5 --warmup 2 2>/dev/null | line "split band=$band"
  unset BLACKLIGHT_AMD_SPLIT_LONG
done
