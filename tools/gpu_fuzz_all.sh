# Every sweep tool once (docs/notebook.md section 5h), seeds from $1 (default 100000): a final check of a library before a release.
# $2 = 1 or 2 runs one half (a gpurun call is at most 20 minutes: the two halves take ~5 and ~10).
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
S=${1:-100000}
PART=${2:-all}
run() { name=$1; shift; "$@" > gpurun_out/fuzz_all_$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/fuzz_all_$name.log | cut -c1-330)"; }
if [ "$PART" != 2 ]; then
  run tiers timeout -k 10 300 python3 tools/gpu_fuzz_tiers.py 1500 $S 10
  FUZZ_RES=128,256 FUZZ_CHUNKS=1 run tiers_large timeout -k 10 400 python3 tools/gpu_fuzz_tiers.py 60 $S 20
  run wide timeout -k 10 500 python3 tools/gpu_fuzz_wide.py 700 $S
  FUZZ_LAYOUT=5 FUZZ_RES=48,64 FUZZ_CHUNKS=1 run refined timeout -k 10 300 python3 tools/gpu_fuzz_tiers.py 1500 $S 50   # (two-level meshes: the locate step inside)
  FUZZ_LAYOUT=7 FUZZ_RES=48,64 run refined_small_blocks timeout -k 10 300 python3 tools/gpu_fuzz_tiers.py 1600 $S 50   # (... in up to 64 times as many blocks)
  FUZZ_LAYOUT=6 run refined_warped timeout -k 10 300 python3 tools/gpu_fuzz_tiers.py 1500 $S 50   # (... over unevenly spaced angles: the locate kernel)
fi
if [ "$PART" != 1 ]; then
  FUZZ_POLARIZED=1 run polarized timeout -k 10 300 python3 tools/gpu_fuzz_wide.py 300 $S
  FUZZ_ADAPTIVE=1 run adaptive timeout -k 10 300 python3 tools/gpu_fuzz_wide.py 150 $S
  run fmks timeout -k 10 300 python3 tools/gpu_fuzz_fmks.py 150 $S
  run slow timeout -k 10 300 python3 tools/gpu_fuzz_slow.py 120 $S
  run checkpoint timeout -k 10 300 python3 tools/gpu_fuzz_checkpoint.py 200 $S
fi
