#!/bin/bash
# Round-6 profile set, ONE run on the final library (output under gpurun_out/prof_r6; tools/collect_profiles.py r06 copies the summaries
# to profiles/r06_* and writes profiles/hbm_traffic.json):
#  (1) the default bench line (with the CPU baseline); (2) rocprofv3 kernel trace + stats of the bench command; (3) PMC passes in runs of
#  their own (--pmc only beside kernel-trace / stats, as the pool requires): fabric traffic, cache hits, instruction mix, issue cycles;
#  (4) the series workloads (geodesics once per series; located samples kept; next snapshot staged beside the render); (5) BASELINE's
#  other configurations and the SURVEY 8(f) rows through bench.py --workload, each with a kernel trace; fabric traffic of the polarized
#  frame; (6) the emulated tiled strong-scaling run.
# PART=1: (1)-(4); PART=2: (5)-(6)   (a gpurun call is at most 20 minutes)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
REPO="$GRAFT_REPO_ROOT"
OUT="$REPO/gpurun_out/prof_r6"
PART=${PART:-1}
mkdir -p "$OUT"
cd "$REPO"
trace() {   # name, description, command ...
  name=$1; what=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- "$@" > "$OUT/config_$name.json" 2> "$OUT/trace_$name.err"
  python3 tools/summarise_trace.py "$OUT/trace_$name" "$OUT/config_${name}_kernel_trace_summary.txt" "$what" > /dev/null
  rm -rf "$OUT/trace_$name"
  python3 - "$OUT/config_$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"], 3), d["unit"], "ms_per_step", round(d["ms_per_step"], 2))
PY
}
if [ "$PART" = 1 ]; then
  rocminfo | grep -m1 "Marketing Name" > "$OUT/box.txt" || true
  rocm-smi --showclocks --showpower 2>/dev/null | head -30 >> "$OUT/box.txt" || true
  python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
  echo "bench done"; tail -c 300 "$OUT/bench_default.json"; echo
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
  cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
  python3 tools/summarise_trace.py "$OUT/trace" "$OUT/kernel_trace_summary.txt" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline  (4 tolerant-tier frames, 4 exact-tier frames, 1 more tolerant for the per-pixel comparison, then the line's series leg: 1 fresh frame and a series of 4, frames 2 - 4 of it over resident records - the coefficient kernel's 30.4 ... 30.9 ms launches; one launch of each kernel per frame)" > /dev/null
  rm -rf "$OUT/trace"
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum" "TCC_REQ_sum" \
             "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
    name=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$name.err"
    echo "pmc $name done"
  done
  python3 tools/summarise_pmc.py "$OUT" "pmc_*" "$OUT/pmc_summary.txt" "PMC totals per kernel over \`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline\` (tolerant-tier frames, one exact-tier frame: see the launch counts; one launch per kernel and frame), separate rocprofv3 --pmc passes" "$OUT/hbm_traffic_raw.json" > /dev/null
  rm -rf "$OUT"/pmc_*/
  # ---- the reference's production mode: a series of snapshots
  python3 bench.py --workload series8 > "$OUT/series.json" 2> /dev/null
  python3 bench.py --workload series8 --arithmetic exact > "$OUT/series_exact.json" 2> /dev/null
  python3 bench.py --workload series8_refined > "$OUT/series_refined.json" 2> /dev/null
  python3 bench.py --workload series8_pipelined > "$OUT/series_pipelined.json" 2> /dev/null
  trace series8 "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload series8  (two series of 8 frames: 2 geodesic launches, 16 coefficient launches, + 1 fresh render)" python3 bench.py --workload series8
  python3 -c "
import json
for n in ('series', 'series_exact', 'series_refined', 'series_pipelined'):
    d = json.load(open('$OUT/' + n + '.json'))
    print(n, round(d['value'], 2), 'Mrays/s', {k: d[k] for k in ('frame_1', 'whole_series') if k in d})
"
fi
if [ "$PART" = 2 ]; then
  for w in formula512 polarized1024 truecolor1024x64 adaptive2048 refined256 blockinterp256 slowlight10; do
    trace $w "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 2 --warmup 1  (3 renders)" python3 bench.py --workload "$w" --steps 2 --warmup 1
  done
  python3 bench.py --workload formula512 --arithmetic exact --steps 2 --warmup 1 > "$OUT/config_formula512_exact.json" 2> /dev/null
  # (the mesh rows in the exact tier, and configuration 4's physics over the refined mesh: lines only)
  for w in refined256 blockinterp256 slowlight10; do
    python3 bench.py --workload $w --arithmetic exact --steps 2 --warmup 1 > "$OUT/config_${w}_exact.json" 2> /dev/null
  done
  python3 bench.py --workload polarized_refined1024 --steps 2 --warmup 1 > "$OUT/config_polarized_refined1024.json" 2> /dev/null
  for tier in tolerant exact; do   # (the mesh of refined256 in 16^3-cell blocks: the table sizes of a deep hierarchy)
    python3 bench.py --workload refined256_deep --arithmetic $tier --steps 2 --warmup 1 > "$OUT/config_refined256_deep_$tier.json" 2> /dev/null
  done
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --output-format csv -d "$OUT/ppmc_$name" -- python3 bench.py --workload polarized1024 --steps 1 --warmup 0 > /dev/null 2> "$OUT/ppmc_$name.err"
  done
  python3 tools/summarise_pmc.py "$OUT" "ppmc_*" "$OUT/config_polarized1024_pmc_summary.txt" "PMC totals per kernel over \`python3 bench.py --workload polarized1024 --steps 1 --warmup 0\` (one 1024^2 full-Stokes frame, tolerant tier, 2 chunks), separate rocprofv3 --pmc passes" "$OUT/polarized_traffic_raw.json" > /dev/null
  rm -rf "$OUT"/ppmc_*/
  python3 tools/gpu_tiled_emulation.py "$OUT/tiled_emulation.json" 2>&1 | grep world || true
fi
ls "$OUT" | head -60
