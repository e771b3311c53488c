cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --workload polarized1024 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items()})"; }
run exact_coefficients
export BLACKLIGHT_AMD_TOLERANT_POLARIZED_COEFFICIENTS=1
run tolerant_coefficients
for v in v_pow v_pow_log_exp v_nocontract; do BLACKLIGHT_AMD_LIB=$PWD/variants/$v.so run tolerant+$v; done
