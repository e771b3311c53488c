set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -k "polar or adaptive or config_4 or window_2048 or tolerant or series" > gpurun_out/gpu_e_tests.log 2>&1
echo "tests rc $?"; tail -6 gpurun_out/gpu_e_tests.log
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_e"; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --workload polarized1024 --steps 2 --warmup 1 > gpurun_out/polarized1024.json 2> "$OUT/trace.err"
python3 tools/summarise_trace.py "$OUT/trace" gpurun_out/polarized1024_trace.txt "polarized1024" > /dev/null
head -14 gpurun_out/polarized1024_trace.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/polarized1024.json"))
print("polarized1024 ms_per_step", round(d["ms_per_step"], 1), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()}, "chunks", d["config"]["chunks_per_step"])
PY
