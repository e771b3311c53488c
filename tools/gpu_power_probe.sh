# sample power / clocks with rocm-smi while the bench runs (diagnostic only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python bench.py --steps 400 --warmup 2 --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null) &
BP=$!
sleep 30
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction)" | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/power_bench.json')); print('Mrays/s', d['value'], d['kernel_ms_per_step'])"
