#!/bin/bash
# quick PMC passes over one tolerant-tier frame: instruction counts, issue and wait cycles per kernel
#   gpurun -- 'bash tools/gpu_pmc_quick.sh [name]'   -> gpurun_out/pmcq_<name>.txt
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
REPO="$GRAFT_REPO_ROOT"
NAME="${1:-run}"
OUT="$REPO/gpurun_out/pmcq"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --arithmetic "${ARITH:-tolerant}" > /dev/null 2> "$OUT/pmc_$i.err" || tail -3 "$OUT/pmc_$i.err"
done
python3 - "$OUT" "$REPO/gpurun_out/pmcq_$NAME.txt" <<'PY'
import csv, glob, collections, sys
out, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:48]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        launches[k][row['Counter_Name']] += 1
with open(dst, 'w') as g:
    for k, v in agg.items():
        if 'bl_' not in k or 'init' in k: continue
        g.write(k + '\n')
        for c, val in sorted(v.items()):
            g.write(f'    {c:28s} per launch {val / launches[k][c]:.4e}  ({launches[k][c]} launches)\n')
print(open(dst).read())
PY
