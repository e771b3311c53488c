#!/bin/bash
# PMC passes over one polarized 1024^2 frame (bench.py --workload polarized1024 --steps 1 --warmup 0), each counter set in a run of its
# own with nothing but --pmc beside it (as the pool requires): traffic, instruction mix, issue and wait cycles per kernel.
#   gpurun -- 'bash tools/gpu_polarized_pmc.sh'   -> gpurun_out/prof_r5/config_polarized1024_pmc_summary.txt
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
REPO="$GRAFT_REPO_ROOT"
OUT="$REPO/gpurun_out/prof_r5"
mkdir -p "$OUT"
cd "$REPO"
rm -rf "$OUT"/ppmc_*
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum" "TCC_REQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE"; do
  name=$(echo "$set" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d "$OUT/ppmc_$name" -- python3 bench.py --workload polarized1024 --steps 1 --warmup 0 > /dev/null 2> "$OUT/ppmc_$name.err"
  echo "pmc $name done"
done
python3 tools/summarise_pmc.py "$OUT" "ppmc_*" "$OUT/config_polarized1024_pmc_summary.txt" "PMC totals per kernel over \`python3 bench.py --workload polarized1024 --steps 1 --warmup 0\` (one 1024^2 full-Stokes frame, tolerant tier, two chunks: two launches of each kernel), separate rocprofv3 --pmc passes" "" > /dev/null
head -60 "$OUT/config_polarized1024_pmc_summary.txt"
