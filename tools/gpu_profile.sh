# rocprofv3 kernel trace + stats, then PMC passes (separate runs, as the pool requires)
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats*" | head -3
cat $(find $OUT/trace -name "*kernel_stats.csv" | head -1)
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$name.err
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/prof'
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:40]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, v in agg.items():
    print(k)
    for c, val in sorted(v.items()):
        print('   ', c, val)
PY
