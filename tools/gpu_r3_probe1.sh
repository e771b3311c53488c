# round 3, first probe: today's baseline + lane-occupancy counters of the geodesic kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_base.json 2> gpurun_out/r3_base.err
cat gpurun_out/r3_base.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('Mrays/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()}, 'exact', round(d['exact_tier']['value'],2))
"
(cd /tmp && TMPDIR=/tmp rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r3_counters_avail.txt 2>&1) || true
bash tools/gpu_pmc_sets.sh "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_SENDMSG SQ_IFETCH" > gpurun_out/r3_pmc_occupancy.txt 2>&1
cat gpurun_out/r3_pmc_occupancy.txt
