# quick PMC pass for the shade / geodesic kernels
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
cd $REPO
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$name.err
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/pmcq'
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:40]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, v in agg.items():
    if 'bl_' not in k: continue
    print(k, {c: f'{val:.3e}' for c, val in sorted(v.items())})
PY
