set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/gpu_suite.log 2>&1
echo "suite rc $?"; tail -15 gpurun_out/gpu_suite.log
for w in blockinterp256 slowlight10 refined256; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 > gpurun_out/f_$w.json 2> gpurun_out/f_$w.err
  python - "$w" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/f_{sys.argv[1]}.json"))
print(sys.argv[1], d["config"]["arithmetic"], "ms_per_step", round(d["ms_per_step"], 1), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
PY
done
