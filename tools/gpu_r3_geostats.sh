# lane occupancy of the geodesic kernel (library built with -DBL_GEO_STATS) and a quick counter pass over the current kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BLACKLIGHT_AMD_DEBUG_COUNTERS=1 BLACKLIGHT_AMD_LIB=$PWD/variants/geostats.so timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r3_geostats.json 2> gpurun_out/r3_geostats.err
grep "debug counters" gpurun_out/r3_geostats.err
bash tools/gpu_pmc_quick.sh > gpurun_out/r3_pmc_quick.txt 2>&1
cat gpurun_out/r3_pmc_quick.txt
