#!/bin/bash
# Start and end of every kernel of the last polarized 1024^2 frame of a bench.py run, relative to the frame's first kernel (ms), from a
# rocprofv3 kernel trace: which kernels run beside which.   gpurun -- 'bash tools/gpu_polarized_timeline.sh [name]'   (environment passes through)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
NAME="${1:-run}"
rm -rf gpurun_out/ptl; mkdir -p gpurun_out/ptl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ptl -o t -- python3 bench.py --workload polarized1024 --steps 1 --warmup 1 > gpurun_out/ptl/bench.json 2> gpurun_out/ptl/err.txt
python3 - <<'PY' > "gpurun_out/pol_timeline_$NAME.txt"
import csv, glob
rows = []
for f in glob.glob('gpurun_out/ptl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48]))
rows.sort()
# the last frame: from the last bl_ray_init_kernel on
starts = [i for i, r in enumerate(rows) if 'ray_init' in r[2]]
rows = rows[starts[-1]:]
t0 = rows[0][0]
for s, e, n in rows:
    if (e - s) > 100000:
        print(f'{(s - t0) / 1e6:9.2f} ... {(e - t0) / 1e6:9.2f} ms  ({(e - s) / 1e6:7.2f})  {n}')
PY
cat "gpurun_out/pol_timeline_$NAME.txt"
