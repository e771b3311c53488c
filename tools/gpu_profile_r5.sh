#!/bin/bash
# Round-5 profile set, one call (output under gpurun_out/prof_r5; tools/collect_profiles.py r05 copies the summaries to profiles/r05_*):
#  (1) the default bench line (with the CPU baseline), (2) rocprofv3 kernel trace + stats of the bench command, (3) PMC passes in runs
#  of their own (--pmc only beside kernel-trace / stats, as the pool requires): HBM traffic, cache, instruction mix, issue and wait
#  cycles, (4) BASELINE's other configurations through bench.py --workload, each with a kernel trace, and instruction counters for the
#  64-frequency frame's per-frequency transfer kernel, (5) the emulated tiled strong-scaling run, (6) the static ISA mix of the
#  coefficient kernels (tools/isa_profile.py needs no GPU, but lives next to the numbers it explains).
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
REPO="$GRAFT_REPO_ROOT"
OUT="$REPO/gpurun_out/prof_r5"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench done"; tail -c 400 "$OUT/bench_default.json"; echo
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
python3 tools/summarise_trace.py "$OUT/trace" "$OUT/kernel_trace_summary.txt" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline  (4 tolerant-tier frames, 4 exact-tier frames, 1 more tolerant; one launch of each kernel per frame)" > /dev/null
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum" "TCC_REQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  name=$(echo "$set" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$name.err"
  echo "pmc $name done"
done
python3 tools/summarise_pmc.py "$OUT" "pmc_*" "$OUT/pmc_summary.txt" "PMC totals per kernel over \`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline\` (tolerant-tier frames, one exact-tier frame: see the launch counts; one launch per kernel and frame), separate rocprofv3 --pmc passes" "$OUT/hbm_traffic_raw.json" > /dev/null
# ---- BASELINE's other configurations at size (bench.py --workload): one line + one kernel trace each
for w in formula512 polarized1024 truecolor1024x64 adaptive2048; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 bench.py --workload "$w" --steps 2 --warmup 1 > "$OUT/config_$w.json" 2> "$OUT/trace_$w.err"
  python3 tools/summarise_trace.py "$OUT/trace_$w" "$OUT/config_${w}_kernel_trace_summary.txt" "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 2 --warmup 1  (tolerant tier, 3 renders)" > /dev/null
  tail -c 300 "$OUT/config_$w.json"; echo
done
python3 bench.py --workload formula512 --arithmetic exact --steps 2 --warmup 1 > "$OUT/config_formula512_exact.json" 2> /dev/null
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo "$set" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d "$OUT/tpmc_$name" -- python3 bench.py --workload truecolor1024x64 --steps 1 --warmup 0 > /dev/null 2> "$OUT/tpmc_$name.err"
done
python3 tools/summarise_pmc.py "$OUT" "tpmc_*" "$OUT/config_truecolor1024x64_pmc_summary.txt" "PMC totals per kernel over \`python3 bench.py --workload truecolor1024x64 --steps 1 --warmup 0\` (one 1024^2 x 64-frequency frame, tolerant tier), separate rocprofv3 --pmc passes" "" > /dev/null
# ---- SURVEY.md 8(f) rows at size
for w in refined256 blockinterp256 slowlight10; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 bench.py --workload "$w" --steps 2 --warmup 1 > "$OUT/config_$w.json" 2> "$OUT/trace_$w.err"
  python3 tools/summarise_trace.py "$OUT/trace_$w" "$OUT/config_${w}_kernel_trace_summary.txt" "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 2 --warmup 1  (3 renders)" > /dev/null
  tail -c 300 "$OUT/config_$w.json"; echo
done
# ---- strong scaling, emulated on the one GPU: the default (BL_TAIL_AUTO), and the one-stepper path for comparison
python3 tools/gpu_tiled_emulation.py "$OUT/tiled_emulation.json" 2>&1 | grep world || true
BLACKLIGHT_AMD_TAIL_POLICY=wide WORLDS=8 python3 tools/gpu_tiled_emulation.py "$OUT/tiled_emulation_wide_stepper_only.json" 2>&1 | grep world || true
BLACKLIGHT_AMD_DEBUG_COUNTERS=1 WORLDS=8 REPS=3 python3 tools/gpu_tiled_emulation.py "$OUT/tmp_split.json" 2>&1 | grep "split long" | head -8 > "$OUT/tail_split_timeline.txt" || true
rm -f "$OUT/tmp_split.json"
# ---- static ISA mix of the coefficient kernels' main loops
{
  python3 tools/isa_profile.py bl_shade_fused.hip fused2_kernelILb1ELb1
  python3 tools/isa_profile.py bl_shade_fast.hip fused_kernelILb1
  python3 tools/isa_profile.py bl_shade.hip bl_shade_exact_kernelILb1
} > "$OUT/isa_mix.txt" 2>&1 || true
ls "$OUT"
