cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/prof_r5b; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
python3 tools/summarise_trace.py "$OUT/trace" "$OUT/kernel_trace_summary.txt" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline  (4 tolerant-tier frames, 4 exact-tier frames, 1 more tolerant; one launch of each kernel per frame)" | head -6
