# which draws of tools/gpu_fuzz_tiers.py leave the host heap corrupted at exit: one process per subset
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout -k 10 300 python3 tools/gpu_fuzz_tiers.py 200 0 ${ORACLE:-0} > gpurun_out/fuzz_$name.log 2>&1; echo "$name rc=$? $(tail -c 200 gpurun_out/fuzz_$name.log | tr '\n' ' ' | cut -c1-160)"; }
run no_oracle A=1
ORACLE=5 run with_oracle A=1
for l in 0 1 2 3 4; do run layout$l FUZZ_LAYOUT=$l; done
