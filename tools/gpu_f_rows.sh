#!/bin/bash
# The SURVEY 8(f) rows at size, one line each: gpurun -- 'bash tools/gpu_f_rows.sh [tests]' -> gpurun_out/f_rows.txt
# (tests: the whole GPU suite first)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
if [ "${1:-}" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_full.log 2>&1
  tail -4 gpurun_out/t_full.log
fi
: > gpurun_out/f_rows.txt
for w in ${WORKLOADS:-refined256 blockinterp256 slowlight10}; do
  for tier in ${TIERS:-tolerant}; do
    timeout -k 10 200 python bench.py --workload "$w" --arithmetic "$tier" --steps 3 --warmup 1 > "gpurun_out/b_${w}_${tier}.json" 2> "gpurun_out/b_${w}_${tier}.err" || tail -3 "gpurun_out/b_${w}_${tier}.err"
    python - "$w" "$tier" <<'PY' | tee -a gpurun_out/f_rows.txt
import json, sys
d = json.load(open("gpurun_out/b_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print(sys.argv[1], sys.argv[2], "%.2f ms" % d["ms_per_step"], {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
PY
  done
done
