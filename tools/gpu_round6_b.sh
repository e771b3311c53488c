# round 6, configuration 5 end to end: tests that touch the download paths, the 64-frequency frame, the frame at size
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_true_color.py tests/test_gpu_configs_at_size.py tests/test_gpu_adaptive_cli.py tests/test_gpu_window_1024.py tests/test_gpu_series.py -m gpu -q -x --timeout 600 > gpurun_out/gpu_b_tests.log 2>&1
echo "tests rc $?"; tail -5 gpurun_out/gpu_b_tests.log
timeout -k 10 300 python bench.py --workload truecolor1024x64 --steps 3 --warmup 1 > gpurun_out/truecolor.json 2> gpurun_out/truecolor.err; echo "truecolor rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/truecolor.json"))
print("truecolor1024x64 ms_per_step", round(d["ms_per_step"], 1), "kernel wall", round(d["kernel_ms_per_step"]["wall"], 1), "exposed", round(d["ms_per_step"] - d["kernel_ms_per_step"]["wall"], 1), "chunks", d["config"]["chunks_per_step"])
PY
timeout -k 10 600 python tools/gpu_config5_full.py gpurun_out/config5_full.json 2>&1 | grep -v amdgpu.ids | tail -8
