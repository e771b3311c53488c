set -u
cd "$GRAFT_REPO_ROOT"
S=${1:-600000}
run() { name=$1; shift; "$@" > gpurun_out/fuzz_all_$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/fuzz_all_$name.log | cut -c1-330)"; }
run tiers timeout -k 10 280 python3 tools/gpu_fuzz_tiers.py 1500 $S 10
FUZZ_RES=128,256 FUZZ_CHUNKS=1 run tiers_large timeout -k 10 330 python3 tools/gpu_fuzz_tiers.py 60 $S 20
run wide timeout -k 10 450 python3 tools/gpu_fuzz_wide.py 700 $S
