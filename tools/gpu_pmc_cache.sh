# cache-side counters of the coefficient kernels (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_cache
rm -rf $OUT; mkdir -p $OUT
cd $REPO
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/c_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --arithmetic tolerant > /dev/null 2> $OUT/c_$name.err
  echo "pmc $name done"; tail -2 $OUT/c_$name.err
done
python3 tools/summarise_pmc.py $OUT "c_*" $OUT/summary.txt "cache counters" "" | grep -A12 "fused\|geodesic_kernel\|transfer_kernel"
