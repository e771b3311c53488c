#!/bin/bash
# Hardware counters of one tolerant-tier benchmark frame per kernel, once per measurement switch ("-" = none), each counter in a
# pass of its own (rocprofv3 --pmc alone: no trace domains beside it):
#   gpurun -- 'COUNTERS="FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_REQ_sum" bash tools/gpu_counters_ab.sh name - NO_FUSED_LOCATE'
#   (BENCH_ARGS="--workload blockinterp256 --steps 1 --warmup 0": another workload's frame)
#   -> gpurun_out/counters_<name>.txt     (FETCH_SIZE is printed x 2-corrected as MI355X_MICROARCH.md prescribes for gfx950)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
NAME="${1:-run}"
shift || true
SWITCHES=("$@")
[ "${#SWITCHES[@]}" -gt 0 ] || SWITCHES=("-")
DST="gpurun_out/counters_$NAME.txt"
: > "$DST"
for sw in "${SWITCHES[@]}"; do
  rm -rf gpurun_out/cab; mkdir -p gpurun_out/cab
  for c in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
    if [ "$sw" = "-" ]; then
      timeout -k 10 300 rocprofv3 --pmc "$c" --output-format csv -d "gpurun_out/cab/$c" -o t -- python3 bench.py ${BENCH_ARGS:---steps 1 --warmup 0 --no-cpu-baseline} --arithmetic "${ARITH:-tolerant}" > /dev/null 2> "gpurun_out/cab/$c.err" || tail -3 "gpurun_out/cab/$c.err"
    else
      export "BLACKLIGHT_AMD_$sw=1"
      timeout -k 10 300 rocprofv3 --pmc "$c" --output-format csv -d "gpurun_out/cab/$c" -o t -- python3 bench.py ${BENCH_ARGS:---steps 1 --warmup 0 --no-cpu-baseline} --arithmetic "${ARITH:-tolerant}" > /dev/null 2> "gpurun_out/cab/$c.err" || tail -3 "gpurun_out/cab/$c.err"
      unset "BLACKLIGHT_AMD_$sw"
    fi
  done
  python3 - "$sw" <<'PY' | tee -a "$DST"
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/cab/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
print('switch', sys.argv[1])
for k, v in agg.items():
    if 'bl_' not in k or 'init' in k: continue
    parts = []
    for c, vals in sorted(v.items()):
        mean = sum(vals) / len(vals)
        if c == 'FETCH_SIZE': parts.append('fetch GB/launch (x2-corrected) %.2f' % (mean * 1024 * 2 / 1e9))
        elif c == 'WRITE_SIZE': parts.append('write GB/launch %.2f' % (mean * 1024 / 1e9))
        else: parts.append('%s/launch %.4e' % (c, mean))
    print('  ', k, 'launches', len(next(iter(v.values()))), '; '.join(parts))
PY
done
