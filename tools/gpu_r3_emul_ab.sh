cd $GRAFT_REPO_ROOT
for lib in variants/a_nocap.so variants/b_cap.so variants/a_nocap.so variants/b_cap.so variants/a_nocap.so variants/b_cap.so; do
  BLACKLIGHT_AMD_LIB=$PWD/$lib WORLDS=8 timeout 300 python tools/gpu_tiled_emulation.py 2>/dev/null | python -c "
import sys, json
d=json.load(sys.stdin)['world_8']
print('$lib', round(d['max_ms'],2), {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['ranks'][0].items()})
"
done
