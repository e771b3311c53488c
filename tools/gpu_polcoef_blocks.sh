cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --workload polarized1024 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
for b in 8 12 16 20 24; do BLACKLIGHT_AMD_POLCOEF_BLOCKS=$b run "blocks_per_cu=$b"; done
BLACKLIGHT_AMD_TOLERANT_POLARIZED_COEFFICIENTS=1 BLACKLIGHT_AMD_POLCOEF_BLOCKS=16 run "tolerant coefficients, 16"
BLACKLIGHT_AMD_TOLERANT_POLARIZED_COEFFICIENTS=1 BLACKLIGHT_AMD_POLCOEF_BLOCKS=24 run "tolerant coefficients, 24"
