#!/bin/bash
# The measurements of DESIGN.md section 5j in one file (profiles/r04_tail_experiments.txt):   gpurun -- 'bash tools/gpu_tail_experiments.sh'
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tail_experiments.txt
{
  echo "== tools/gpu_ray_length_by_radius.py: a ray's time by impact parameter (64 rays = one wave of bl_geodesic_kernel per line)"
  timeout -k 10 300 python3 tools/gpu_ray_length_by_radius.py 2>&1 | grep "^b ="
  echo
  echo "== tools/gpu_quad_latency.py: rays alone, a ray per lane against a ray per quad (BL_SWITCH_QUAD_EVERY_RAY)"
  timeout -k 10 300 python3 tools/gpu_quad_latency.py 2>&1 | grep "^spin"
  echo
  echo "== tools/gpu_quad_tail.sh AFTER=0: as shipped / BL_SWITCH_QUAD_TAIL / BL_SWITCH_TAIL_OVERLAP (every wave parks its rays once the queue is dry)"
  bash tools/gpu_quad_tail.sh "AFTER=0" > /dev/null 2>&1
  cat gpurun_out/quad_tail.txt
  echo
  echo "== tools/gpu_overlap_debug.sh: timeline of renders with BL_SWITCH_TAIL_OVERLAP"
  bash tools/gpu_overlap_debug.sh 2>&1
} > "$OUT" 2>&1
tail -5 "$OUT"
