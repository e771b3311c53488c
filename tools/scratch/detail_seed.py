import os, sys
os.environ["FUZZ_RES"]="256,384,512"
sys.argv=['x']
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests'); sys.path.insert(0,R+'/tools')
import importlib.util, numpy as np
spec=importlib.util.spec_from_file_location('fz',os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),'tools','gpu_fuzz_tiers.py')); fz=importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
import blacklight_amd as bl
params, grid, what = fz.draw(60059)
print(what, {k:params[k] for k in ('camera_resolution','camera_r','camera_th','camera_width','camera_type','simulation_a','image_num_frequencies','image_frequency','cut_sigma_max','cut_theta_e_max','cut_beta_inverse_min','fallback_nan','plasma_use_p')})
p=bl.Params.from_dict(params)
with bl.Context(p) as ctx:
    ctx.set_grid(grid)
    e=ctx.render(); ctx.set_arithmetic("tolerant"); t=ctx.render()
    os.environ["BLACKLIGHT_AMD_NO_FUSED_LOCATE"]="1"
    nf=ctx.render()
    del os.environ["BLACKLIGHT_AMD_NO_FUSED_LOCATE"]
    ctx.debug_set_guard_band(1e30); w=ctx.render()
mx0=np.nanmax(np.abs(e["image"][0])); print("non-fused fast kernel: worst", np.nanmax(np.abs(nf["image"][0]-e["image"][0]))/mx0, "launches_locate", nf["stats"].launches_locate, "pixel 48496:", nf["image"][0,48496], "S_in", e["stats"].n_gathers, t["stats"].n_gathers, nf["stats"].n_gathers)
img_e, img_t, img_w = e["image"], t["image"], w["image"]
for r in range(img_e.shape[0]):
    mx=np.nanmax(np.abs(img_e[r])); d=np.abs(img_t[r]-img_e[r]); c=int(np.nanargmax(d))
    dw=np.abs(img_w[r]-img_e[r])
    print('row',r,'max',mx,'worst',np.nanmax(d)/mx,'at',c,'exact',img_e[r,c],'tol',img_t[r,c],'rel at pixel',d[c]/abs(img_e[r,c]),'samples',e["sample_num"][c], 'all-deferred worst', np.nanmax(dw)/mx)
    order=np.argsort(-np.nan_to_num(d))[:5]; print('   top', [(int(i), float(d[i]/mx)) for i in order])
print('deferred', t["stats"].n_deferred, w["stats"].n_deferred)
