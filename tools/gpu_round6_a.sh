# round 6, first GPU call: the -m gpu suite, the bench line, the series workloads
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/gpu_suite.log 2>&1
echo "suite rc $?"; tail -15 gpurun_out/gpu_suite.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --workload series8 > gpurun_out/series8.json 2> gpurun_out/series8.err; echo "series rc $?"
timeout -k 10 300 python bench.py --workload series8 --arithmetic exact > gpurun_out/series8_exact.json 2> gpurun_out/series8_exact.err; echo "series exact rc $?"
timeout -k 10 300 python bench.py --workload series8_refined > gpurun_out/series8_refined.json 2> gpurun_out/series8_refined.err; echo "series refined rc $?"
tail -c 1500 gpurun_out/series8.json; tail -c 600 gpurun_out/series8.err
