# Round-2 profile: rocprofv3 kernel trace + stats of the default bench command (tolerant tier timed first, exact tier
# after it), then PMC passes in runs of their own (--pmc only with kernel-trace / stats domains, as the pool requires).
# Output under gpurun_out/prof_r2; copy the summaries to profiles/r02_*.
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_r2'
f = glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)[0]
dur = collections.defaultdict(list); regs = {}
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name'].split('(')[0]
    dur[k].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
    regs[k] = (row.get('VGPR_Count'), row.get('Accum_VGPR_Count'), row.get('SGPR_Count'), row.get('LDS_Block_Size'), row.get('Scratch_Size'), row.get('Grid_Size'), row.get('Workgroup_Size'))
with open(out + '/kernel_trace_summary.txt', 'w') as g:
    g.write('rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline  (4 tolerant-tier frames, then 4 exact-tier frames + 1 tolerant; 2 launches per frame)\n')
    g.write('kernel, launches, avg_ms, min_ms, max_ms, total_ms, (VGPR, AGPR, SGPR, LDS, scratch, grid, wg)\n')
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        g.write(f'{k}, {len(v)}, {sum(v)/len(v):.3f}, {min(v):.3f}, {max(v):.3f}, {sum(v):.2f}, {regs[k]}\n')
print(open(out + '/kernel_trace_summary.txt').read())
PY
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE" "TA_BUSY_sum TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$name.err
  echo "pmc $name done"
done
python3 - <<'PY'
import csv, glob, collections, os, json
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_r2'
agg = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); launches[k][row['Counter_Name']] += 1
with open(out + '/pmc_summary.txt', 'w') as g:
    g.write('PMC totals per kernel over `python3 bench.py --steps 1 --warmup 0` (one tolerant-tier frame, one exact-tier frame, one more tolerant: '
            'see the launch counts; 2 launches per frame), separate rocprofv3 --pmc passes\n')
    for k, v in agg.items():
        if 'bl_' not in k: continue
        g.write(k + '\n')
        for c, val in sorted(v.items()):
            g.write(f'    {c} {val:.6e} (launches {launches[k][c]})\n')
print(open(out + '/pmc_summary.txt').read())
# HBM-side traffic per launch, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KiB) counts 64 B per 128-B
# request on gfx950 -> x2; WRITE_SIZE (KiB) as reported
traffic = {}
for k, v in agg.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        n = launches[k]['FETCH_SIZE']
        traffic[k] = {'launches': n, 'fetch_bytes_per_launch_x2_corrected': v['FETCH_SIZE'] * 1024 * 2 / n,
                      'write_bytes_per_launch': v['WRITE_SIZE'] * 1024 / n}
json.dump(traffic, open(out + '/hbm_traffic_raw.json', 'w'), indent=1)
print(json.dumps(traffic, indent=1))
PY
