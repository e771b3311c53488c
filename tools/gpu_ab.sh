#!/bin/bash
# A/B on one GPU box, so that clock differences between boxes do not enter the comparison: bench.py once per measurement
# switch (BLACKLIGHT_AMD_<NAME>, include/blacklight_amd.h; "-" = none) and once per library build under variants/*.so
# (tools/build_variant.sh: the current library with one source rebuilt under extra flags).
#   gpurun -- 'bash tools/gpu_ab.sh [steps] [switches ...]'     e.g.  bash tools/gpu_ab.sh 10 - NO_FUSED_LOCATE SAMPLE_RECORDS
#   ARITH=exact for the exact tier, ROUNDS=2 for two passes
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
STEPS="${1:-10}"
shift || true
SWITCHES=("$@")
[ "${#SWITCHES[@]}" -gt 0 ] || SWITCHES=("-")
OUT="gpurun_out/ab_$(date +%H%M%S).txt"
report() {
  python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
e = d.get('exact_tier') or {}
print('$1', 'Mrays/s', round(d['value'], 3), 'ms/step', round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()},
      d['roofline']['kernel'], 'switches', d.get('switches'), 'vs exact', (d.get('tolerant_vs_exact') or {}).get('image_linf_over_max'),
      'exact', round(e.get('value', 0.0), 2), {k: round(v, 2) for k, v in (e.get('kernel_ms_per_step') or {}).items()})
" | tee -a "$OUT"
}
for round in $(seq 1 "${ROUNDS:-1}"); do
  for sw in "${SWITCHES[@]}"; do
    if [ "$sw" = "-" ]; then
      timeout 600 python3 bench.py --steps "$STEPS" --warmup 3 --no-cpu-baseline --arithmetic "${ARITH:-tolerant}" 2>gpurun_out/ab_err.txt | report "default"
    else
      env "BLACKLIGHT_AMD_$sw=1" timeout 600 python3 bench.py --steps "$STEPS" --warmup 3 --no-cpu-baseline --arithmetic "${ARITH:-tolerant}" 2>gpurun_out/ab_err.txt | report "$sw"
    fi
  done
  for lib in variants/*.so; do
    [ -e "$lib" ] || continue
    BLACKLIGHT_AMD_LIB="$PWD/$lib" timeout 600 python3 bench.py --steps "$STEPS" --warmup 3 --no-cpu-baseline --arithmetic "${ARITH:-tolerant}" 2>gpurun_out/ab_err.txt | report "$lib"
  done
done
