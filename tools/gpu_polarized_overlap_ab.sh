#!/bin/bash
# serial against BLACKLIGHT_AMD_POLARIZED_OVERLAP=1, alternating, three times each (bench.py --workload polarized1024 and adaptive2048 once)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python3 bench.py --workload "${W:-polarized1024}" --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
for i in 1 2 3; do
  BLACKLIGHT_AMD_POLARIZED_OVERLAP=0 run "serial"
  BLACKLIGHT_AMD_POLARIZED_OVERLAP=1 run "overlap"
done
W=adaptive2048 BLACKLIGHT_AMD_POLARIZED_OVERLAP=0 run "adaptive2048 serial"
W=adaptive2048 BLACKLIGHT_AMD_POLARIZED_OVERLAP=1 run "adaptive2048 overlap"
