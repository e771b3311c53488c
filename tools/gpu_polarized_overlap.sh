#!/bin/bash
# The transport matrices on a second stream beside the per-frequency coefficient kernel (BLACKLIGHT_AMD_POLARIZED_OVERLAP=1), for several
# sizes of the coefficient kernel's grid: bench.py --workload polarized1024, one line each, and the kernel timeline of one of them.
#   gpurun -- 'bash tools/gpu_polarized_overlap.sh'
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python3 bench.py --workload polarized1024 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
BLACKLIGHT_AMD_POLARIZED_OVERLAP=0 run "serial, 20 workgroups per CU"
for b in 20 12 8 6; do
  BLACKLIGHT_AMD_POLARIZED_OVERLAP=1 BLACKLIGHT_AMD_POLCOEF_BLOCKS=$b run "overlap, $b workgroups per CU"
done
python3 -m pytest tests/test_gpu_tolerant.py -q -x -k polar 2>&1 | tail -1
BLACKLIGHT_AMD_POLARIZED_OVERLAP=1 python3 -m pytest tests/test_gpu_tolerant.py -q -x -k polar 2>&1 | tail -1
