# per-row tolerant-vs-exact distances of polarized draws under library variants (variants/*.so): which function carries a difference
cd $GRAFT_REPO_ROOT
for lib in "" variants/*.so; do
  echo "### ${lib:-default}"
  if [ -n "$lib" ]; then export BLACKLIGHT_AMD_LIB=$PWD/$lib; fi
  timeout -k 10 200 python3 tools/gpu_fuzz_detail3.py "$@" 2>&1 | grep "default\] per-row"
done
