#!/bin/bash
# BL_SWITCH_SPLIT_LONG (rays predicted long on compute units of their own) on the emulated eighth of the
# benchmark frame: the default, then bands and CU counts.   gpurun -- 'bash tools/gpu_split_long.sh'  -> gpurun_out/split_long.txt
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp WORLDS="${WORLDS:-8}" REPS="${REPS:-9}"
OUT=gpurun_out/split_long.txt
: > "$OUT"
run() {   # label, then environment assignments
  local label="$1"; shift
  env "$@" timeout -k 10 280 python3 tools/gpu_tiled_emulation.py gpurun_out/split_tmp.json > /dev/null 2> gpurun_out/split_tmp.err
  local rc=$?
  echo "== $label (exit $rc)" | tee -a "$OUT"
  grep -h "^world_\|split long" gpurun_out/split_tmp.err | sort | uniq -c | sort -rn | head -4 | tee -a "$OUT"
}
run "BL_TAIL_WIDE (one stepper)" BLACKLIGHT_AMD_TAIL_POLICY=wide
run "BL_TAIL_AUTO (default)" BLACKLIGHT_AMD_DEBUG_COUNTERS=1
for spec in ${SPECS:-}; do
  cus="${spec%%:*}"; band="${spec##*:}"
  run "split: $cus CUs, band +-$band M" BLACKLIGHT_AMD_SPLIT_LONG=1 BLACKLIGHT_AMD_SPLIT_CUS="$cus" BLACKLIGHT_AMD_SPLIT_BAND="$band" BLACKLIGHT_AMD_DEBUG_COUNTERS=1
done
