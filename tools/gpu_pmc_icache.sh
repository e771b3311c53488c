# instruction-cache counters of the benchmark kernels (one rocprofv3 --pmc pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
cd $REPO
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/err.txt
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_icache'
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'].split('(')[0]][row['Counter_Name']] += float(row['Counter_Value'])
for k, v in agg.items():
    if 'bl_' in k: print(k, {c: f'{x:.4g}' for c, x in sorted(v.items())})
PY
tail -3 $OUT/err.txt
