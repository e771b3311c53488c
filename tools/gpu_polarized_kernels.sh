#!/bin/bash
# Kernel trace of the polarized 1024^2 frame (configuration 4's physics): per-kernel times of one bench.py --workload polarized1024 run.
#   gpurun -- 'bash tools/gpu_polarized_kernels.sh [name]'  -> gpurun_out/pol_trace_<name>.txt    (environment passes through)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
NAME="${1:-run}"
rm -rf gpurun_out/ptrace; mkdir -p gpurun_out/ptrace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ptrace -o t -- python3 bench.py --workload "${WORKLOAD:-polarized1024}" --steps 2 --warmup 1 --arithmetic "${ARITH:-tolerant}" > gpurun_out/ptrace/bench.json 2> gpurun_out/ptrace/err.txt
python3 - <<'PY' > "gpurun_out/pol_trace_$NAME.txt"
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/ptrace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:62s} launches {len(v):3d}  avg {sum(v)/len(v):8.3f} ms  total {sum(v):9.2f} ms')
PY
tail -1 gpurun_out/ptrace/bench.json >> "gpurun_out/pol_trace_$NAME.txt"
head -14 "gpurun_out/pol_trace_$NAME.txt"
