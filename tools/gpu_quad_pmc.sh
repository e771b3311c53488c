#!/bin/bash
# Instruction and cycle counters of configuration 2's geodesic stage with and without the quad tail
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for mode in notail tail; do
  rm -rf gpurun_out/qpmc_$mode
  if [ $mode = tail ]; then export BLACKLIGHT_AMD_QUAD_TAIL=1; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/qpmc_$mode -o q --output-format csv -- python3 bench.py --workload formula512 --steps 1 --warmup 0 > gpurun_out/qpmc_$mode.log 2>&1
done
