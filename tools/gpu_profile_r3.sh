# Round-3 profile set, one call: (1) the default bench line, (2) rocprofv3 kernel trace + stats of the bench command,
# (3) PMC passes in runs of their own (--pmc only beside kernel-trace / stats, as the pool requires): HBM traffic, cache,
# instruction mix, (4) configuration 2 (formula mode, 512^2) kernel trace + instruction counters, (5) the emulated tiled
# strong-scaling run (8 ranks' shares of one frame, one after another on the one GPU).
# Output under gpurun_out/prof_r3; tools/collect_profiles_r3.py copies the summaries to profiles/r03_*.
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_r3
rm -rf $OUT; mkdir -p $OUT
cd $REPO
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done"; tail -c 600 $OUT/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/summarise_trace.py $OUT/trace $OUT/kernel_trace_summary.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline  (4 tolerant-tier frames, then 4 exact-tier frames + 1 tolerant; one launch of each kernel per frame)"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$name.err
  echo "pmc $name done"
done
python3 tools/summarise_pmc.py $OUT "pmc_*" $OUT/pmc_summary.txt "PMC totals per kernel over \`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline\` (tolerant-tier frames, one exact-tier frame: see the launch counts; one launch per kernel and frame), separate rocprofv3 --pmc passes" $OUT/hbm_traffic_raw.json
# ---- configuration 2: formula mode at 512^2, both tiers
for tier in tolerant exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/formula_trace_$tier -- python3 tools/gpu_formula_frame.py $tier 3 > $OUT/formula_$tier.json 2> $OUT/formula_trace_$tier.err
  python3 tools/summarise_trace.py $OUT/formula_trace_$tier $OUT/formula_${tier}_kernel_trace_summary.txt "rocprofv3 --kernel-trace --stats -- python3 tools/gpu_formula_frame.py $tier 3  (BASELINE configuration 2: example_formula at 512^2; 4 frames)"
  cat $OUT/formula_$tier.json
done
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/fpmc_$name -- python3 tools/gpu_formula_frame.py tolerant 1 > /dev/null 2> $OUT/fpmc_$name.err
  echo "formula pmc $name done"
done
python3 tools/summarise_pmc.py $OUT "fpmc_*" $OUT/formula_pmc_summary.txt "PMC totals per kernel over \`python3 tools/gpu_formula_frame.py tolerant 1\` (2 frames of BASELINE configuration 2 in the tolerant tier), separate rocprofv3 --pmc passes" ""
# ---- strong scaling, emulated: the 8 ranks' tile sets of one 1024^2 frame one after another (max over ranks = the frame's time)
python3 tools/gpu_tiled_emulation.py > $OUT/tiled_emulation.json 2> $OUT/tiled_emulation.err
python3 -c "
import json; d=json.load(open('$OUT/tiled_emulation.json'))
for k,v in d.items(): print(k, round(v['max_ms'],2), 'ms', round(v['mrays_per_s'],1), 'Mrays/s')
"
