# PMC passes with caller-supplied counter sets: bash tools/gpu_pmc_sets.sh "SET1 COUNTERS" "SET2 COUNTERS" ...
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmcs
rm -rf $OUT; mkdir -p $OUT
cd $REPO
n=0
for set in "$@"; do
  n=$((n+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$n -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$n.err
  tail -3 $OUT/pmc_$n.err | grep -i "error\|invalid\|fail" | head -3
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/pmcs'
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:40]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, v in agg.items():
    if 'bl_' not in k: continue
    print(k, {c: f'{val:.3e}' for c, val in sorted(v.items())})
PY
