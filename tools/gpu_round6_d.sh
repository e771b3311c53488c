set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tolerant.py tests/test_gpu_parity.py tests/test_gpu_slow_light.py tests/test_gpu_fmks.py tests/test_gpu_series.py tests/test_gpu_checkpoint.py -m gpu -q --timeout 600 > gpurun_out/gpu_d_tests.log 2>&1
echo "tests rc $?"; tail -6 gpurun_out/gpu_d_tests.log
for w in blockinterp256 slowlight10 refined256; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 > gpurun_out/f_$w.json 2> gpurun_out/f_$w.err
  python - "$w" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/f_{sys.argv[1]}.json"))
print(sys.argv[1], d["config"]["arithmetic"], "ms_per_step", round(d["ms_per_step"], 1), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
PY
done
