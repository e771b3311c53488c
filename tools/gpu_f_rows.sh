#!/bin/bash
# SURVEY.md 8(f) rows at the benchmark's size: bench.py --workload refined256 / blockinterp256 / slowlight10 under a kernel trace
# (rocprofv3 --kernel-trace --stats), one summary per workload.   gpurun -- 'bash tools/gpu_f_rows.sh [tag]'  -> gpurun_out/f_<workload>_<tag>.txt
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG="${1:-run}"
for w in ${WORKLOADS:-refined256 blockinterp256 slowlight10}; do
  rm -rf gpurun_out/ftrace; mkdir -p gpurun_out/ftrace
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ftrace -o t -- python3 bench.py --workload "$w" --steps 2 --warmup 1 > gpurun_out/ftrace/bench.json 2> gpurun_out/ftrace/err.txt || tail -5 gpurun_out/ftrace/err.txt
  python3 - "$w" > "gpurun_out/f_${w}_$TAG.txt" <<'PY'
import csv, glob, collections, sys
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --workload", sys.argv[1], "--steps 2 --warmup 1   (3 renders)")
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/ftrace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:72s} launches {len(v):3d}  avg {sum(v)/len(v):9.3f} ms  total {sum(v):9.2f} ms')
PY
  tail -1 gpurun_out/ftrace/bench.json >> "gpurun_out/f_${w}_$TAG.txt"
  head -7 "gpurun_out/f_${w}_$TAG.txt"; tail -1 gpurun_out/ftrace/bench.json | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('   ->', round(d['ms_per_step'],1), 'ms per frame,', round(d['value'],3), 'Mrays/s', {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'chunks', d['config']['chunks_per_step'])
except Exception as e: print('no bench line', e)"
done
