# per-kernel durations of one short bench run (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/traceq
rm -rf $OUT; mkdir -p $OUT
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import csv, glob, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/traceq'
for f in glob.glob(out + '/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        print(row['Name'][:50], 'calls', row['Calls'], 'avg_us', round(float(row['AverageNs']) / 1e3, 1), 'pct', row['Percentage'])
PY
tail -c 600 $OUT/bench.json
