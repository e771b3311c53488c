#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one tolerant-tier frame per kernel (separate passes), and a bench line:
#   gpurun -- 'bash tools/gpu_traffic_quick.sh'   -> gpurun_out/traffic_quick.txt
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rm -rf gpurun_out/tq; mkdir -p gpurun_out/tq
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/tq/$c -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> gpurun_out/tq/$c.err
done
python3 - <<'PY' | tee gpurun_out/traffic_quick.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/tq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'bl_' not in k: continue
    f = v.get('FETCH_SIZE', [0]); w = v.get('WRITE_SIZE', [0])
    print(k, 'launches', len(f), 'fetch GB/launch x2-corrected %.2f' % (sum(f) / len(f) * 1024 * 2 / 1e9), 'write GB/launch %.2f' % (sum(w) / len(w) * 1024 / 1e9))
PY
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()}, d['tolerant_vs_exact'])" | tee -a gpurun_out/traffic_quick.txt
