# Polarized 1024^2 frame (config 4's physics at size on one GPU) under rocprofv3 --kernel-trace: per-kernel times and resources.
# ARITH=exact|tolerant. Output: gpurun_out/prof_pol_$ARITH/kernel_trace_summary.txt
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
ARITH=${ARITH:-exact}
export ARITH
OUT=$REPO/gpurun_out/prof_pol_$ARITH
rm -rf $OUT; mkdir -p $OUT
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/gpu_polarized_at_size.py 1024 > $OUT/run.json 2> $OUT/trace.err
python3 - <<'PY'
import csv, glob, os, collections
arith = os.environ['ARITH']
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_pol_' + arith
f = glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)[0]
dur = collections.defaultdict(list); regs = {}
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name'].split('(')[0]
    dur[k].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
    regs[k] = (row.get('VGPR_Count'), row.get('Accum_VGPR_Count'), row.get('SGPR_Count'), row.get('LDS_Block_Size'), row.get('Scratch_Size'))
with open(out + '/kernel_trace_summary.txt', 'w') as g:
    g.write(f'rocprofv3 --kernel-trace --stats -- python3 tools/gpu_polarized_at_size.py 1024  (ARITH={arith}; two 1024^2 polarized + tau frames)\n')
    g.write('kernel, launches, avg_ms, min_ms, max_ms, total_ms, (VGPR, AGPR, SGPR, LDS, scratch)\n')
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        g.write(f'{k}, {len(v)}, {sum(v)/len(v):.3f}, {min(v):.3f}, {max(v):.3f}, {sum(v):.2f}, {regs[k]}\n')
print(open(out + '/kernel_trace_summary.txt').read())
PY
tail -30 $OUT/run.json
