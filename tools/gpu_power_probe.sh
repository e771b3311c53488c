#!/bin/bash
# Package power and engine clock while frames are rendered back to back (rocm-smi sampled twice a second beside a render loop):
#   gpurun -- 'bash tools/gpu_power_probe.sh'    -> gpurun_out/power_probe.txt
# Workloads: the benchmark frame, the polarized frame with its matrices beside the coefficients and one after the other.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt
: > "$OUT"
probe() {   # $1 label, $2 workload, further environment through env
  local label="$1" workload="$2"
  python3 - "$workload" > gpurun_out/power_loop.txt 2>&1 <<'PY' &
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, blacklight_amd as bl
from blacklight_amd import mock
params = dict(bench.WORKLOAD)
if sys.argv[1] == 'polarized':
    params.update(image_polarization=True, image_tau=True)
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
with bl.Context(bl.Params.from_dict(params)) as ctx:
    ctx.set_grid(grid); ctx.set_arithmetic('tolerant')
    ctx.render()
    print('ready', flush=True)
    t0 = time.time(); n = 0
    while time.time() - t0 < 14.0:
        ctx.render(); n += 1
    print('frames', n, 'ms per frame', 1000.0 * (time.time() - t0) / n, flush=True)
PY
  local pid=$!
  while ! grep -q ready gpurun_out/power_loop.txt 2>/dev/null; do sleep 0.5; kill -0 $pid 2>/dev/null || break; done
  sleep 2
  : > gpurun_out/power_samples.txt
  for i in $(seq 1 16); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk" >> gpurun_out/power_samples.txt
    sleep 0.5
  done
  wait $pid
  python3 - "$label" >> "$OUT" <<'PY'
import re, sys, statistics
power, clock = [], []
for line in open('gpurun_out/power_samples.txt'):
    m = re.search(r'Power \(W\): ([\d.]+)', line)
    if m: power.append(float(m.group(1)))
    m = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', line)
    if m: clock.append(float(m.group(1)))
frames = [l.strip() for l in open('gpurun_out/power_loop.txt') if l.startswith('frames')]
print(sys.argv[1], '| power W median', statistics.median(power) if power else None, 'max', max(power) if power else None,
      '| sclk MHz median', statistics.median(clock) if clock else None, 'min', min(clock) if clock else None, '|', frames[0] if frames else 'no frames')
PY
}
probe "benchmark frame" benchmark
probe "polarized, matrices beside coefficients" polarized
cat "$OUT"
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -4
