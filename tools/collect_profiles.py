"""Copy the summaries of tools/gpu_profile_rN.sh (gpurun_out/prof_rN) to profiles/r0N_* and write profiles/hbm_traffic.json (what
bench.py's roofline.traffic and .traffic_whole_pipeline read) with the hash of the kernel sources the numbers were measured on.
    python3 tools/collect_profiles.py r06"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(REPO, "gpurun_out", "prof_r" + tag[1:].lstrip("0"))
dst = os.path.join(REPO, "profiles")
for name in sorted(os.listdir(src)):
    path = os.path.join(src, name)
    if os.path.isfile(path) and name.endswith((".json", ".txt", ".csv")) and os.path.getsize(path) > 0 and not name.endswith(".err"):
        shutil.copy(path, os.path.join(dst, f"{tag}_{name}"))
raw = json.load(open(os.path.join(src, "hbm_traffic_raw.json")))
line = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
source = (f"profiles/{tag}_hbm_traffic_raw.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` "
          f"in separate passes (tools/gpu_profile_{tag[0]}{tag[1:].lstrip('0')}.sh); FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies 64 B per 128-B request), WRITE_SIZE (KiB) x 1024; "
          "L2-to-fabric traffic, Infinity-Cache hits included")
record = {"csrc_sha256_16": bench.kernel_source_hash(),
          "launch": f"one launch per frame of the 1024^2 / 256^3 benchmark; {line['roofline']['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic "
                    "per launch (256 B x gathered samples + 13 B x rays)"}
for tier, prefix in (("tolerant", "void " + line["roofline"]["kernel"]), ("exact", "void bl_shade_exact")):
    for k, v in raw.items():
        if k.startswith(prefix):
            record[tier] = {"kernel": k, "coefficient_kernel_bytes_per_launch": v["fetch_bytes_per_launch_x2_corrected"] + v["write_bytes_per_launch"],
                            "fetch_bytes_per_launch": v["fetch_bytes_per_launch_x2_corrected"], "write_bytes_per_launch": v["write_bytes_per_launch"],
                            "source": source}
# the whole pipeline of a frame, fabric bytes: ray start + stepper + coefficient kernel (+ the tolerant tier's exact second pass) + transfer
def per_launch(prefix):
    return sum(v["fetch_bytes_per_launch_x2_corrected"] + v["write_bytes_per_launch"] for k, v in raw.items() if k.startswith(prefix))
common = per_launch("void bl_ray_init_kernel") + per_launch("void bl_geodesic_kernel")
if "tolerant" in record:
    record["tolerant"]["whole_pipeline_bytes_per_frame"] = (common + record["tolerant"]["coefficient_kernel_bytes_per_launch"]
                                                            + per_launch("void bl_shade_kernel<0, false, false, true, false, true, true>") + per_launch("bl_transfer_composed_kernel"))
if "exact" in record:
    record["exact"]["whole_pipeline_bytes_per_frame"] = common + record["exact"]["coefficient_kernel_bytes_per_launch"] + per_launch("void bl_transfer_kernel<false>")
record["all_kernels"] = raw
json.dump(record, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in record.items() if k != "all_kernels"}, indent=1))
