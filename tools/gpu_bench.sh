set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1500 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
