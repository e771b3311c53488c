#!/usr/bin/env python3
"""BASELINE.json's configuration 5 IN ONE PIECE on one GPU: example_true_color.input's 64 frequencies (lin_wave, 1.5e11 ... 3.3e11 Hz)
at 4096^2 over the 256^3 mock - 16.7 M rays, an 8.6 GB image. Renders the frame (tolerant tier; ARITH=exact for the other), reports time,
chunks and per-kernel milliseconds, writes the .npz through the library's ZIP64 writer and reads it back, and spot-checks three rows
against the windows tests/test_gpu_configs_at_size.py::test_config_5_windows_of_the_4096_lattice renders on their own.

    python3 tools/gpu_config5_full.py gpurun_out/config5_full.json         RES=4096 NFREQ=64 ARITH=tolerant KEEP_NPZ=0
"""
import json
import os
os.environ.setdefault("BLACKLIGHT_AMD_ARITHMETIC", "exact")   # (a context starts in this tier; the tool names the tolerant one where it wants it)
import sys
import time
import zipfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import blacklight_amd as bl
from blacklight_amd import mock

res, n_freq = int(os.environ.get("RES", "4096")), int(os.environ.get("NFREQ", "64"))
tier = os.environ.get("ARITH", "tolerant")
out_path = os.environ.get("NPZ", "/tmp/config5_full.npz")
params = dict(bench.WORKLOAD, camera_resolution=res, image_num_frequencies=n_freq, image_frequency_start=1.5e11, image_frequency_end=3.3e11,
              image_frequency_spacing="lin_wave", output_file=out_path)
params.pop("image_frequency", None)
grid = mock.generate(n_r=256, n_th=256, n_ph=256)
report = {"workload": f"{res}^2 x {n_freq} frequencies over the 256^3 mock, one MI355X", "arithmetic": tier}
with bl.Context(bl.Params.from_dict(params)) as ctx:
    ctx.set_geodesic_reuse(False)   # a measurement of whole renders: every one integrates its geodesics
    ctx.set_grid(grid)
    ctx.set_arithmetic(tier)
    t0 = time.perf_counter()
    frame = ctx.render()
    report["first_render_s"] = time.perf_counter() - t0   # (with the scratch allocation)
    print("first render", round(report["first_render_s"], 2), "s", flush=True)
    t0 = time.perf_counter()
    frame = ctx.render()
    sec = time.perf_counter() - t0
    st = frame["stats"]
    report.update(render_s=sec, mrays_per_s=res * res / sec / 1e6, chunks=st.n_chunks, tier_ran="tolerant" if st.arithmetic == 1 else "exact",
                  samples_per_ray=st.n_samples / (res * res), image_gb=frame["image"].nbytes / 1e9,
                  kernel_ms=dict(geodesic=st.ms_geodesic, locate=st.ms_locate, coefficient=st.ms_shade, transfer=st.ms_transfer, wall=st.ms_wall),
                  finite_fraction=float(np.isfinite(frame["image"]).mean()), flagged_rays=int(st.n_flagged))
    print("render", round(sec, 2), "s", report["kernel_ms"], "chunks", st.n_chunks, flush=True)
    # three windows of the lattice on their own (what the test renders): the same pixels of the full frame, bit for bit
    # (many frequencies: the per-frequency transfer kernel, whose arithmetic does not depend on how a frame is cut)
    windows = []
    for v0, u0 in ((1930 * res // 4096, 1600 * res // 4096), (2300 * res // 4096, 900 * res // 4096), (8, 16)):
        iv, iu = np.mgrid[v0:v0 + 48, u0:u0 + 48]
        windows.append((iv * res + iu).reshape(-1).astype(np.int32))
    window = np.concatenate(windows)
    part = ctx.render(pixel_map=window)
    rows = (0, n_freq // 4 + 1, n_freq - 1)
    report["window_rows_checked"] = list(rows)
    report["windows_equal_full_frame_bit_for_bit"] = bool(all(
        np.array_equal(part["image"][l].view(np.uint64), frame["image"][l][window].view(np.uint64)) for l in rows))
    report["window_sample_num_equal"] = bool(np.array_equal(part["sample_num"], frame["sample_num"][window]))
    # the output file: ZIP64 where 32 bits do not hold a size or an offset (include/blacklight_amd.h, bl_write_output)
    t0 = time.perf_counter()
    ctx.write_output([frame], path=out_path)
    report["write_s"] = time.perf_counter() - t0
    report["npz_gb"] = os.path.getsize(out_path) / 1e9
    print("written", round(report["npz_gb"], 2), "GB in", round(report["write_s"], 1), "s", flush=True)
t0 = time.perf_counter()
with zipfile.ZipFile(out_path) as z:
    names = z.namelist()
    report["npz_records"] = names
with np.load(out_path) as z:
    back = z["I_nu"]
    report["read_back_shape"] = list(back.shape)
    flat = back.reshape(n_freq, -1)
    report["read_back_rows_equal"] = bool(all(np.array_equal(flat[l].view(np.uint64), frame["image"][l].view(np.uint64)) for l in rows))
report["read_back_s"] = time.perf_counter() - t0
if not os.environ.get("KEEP_NPZ"):
    os.remove(out_path)
with open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w") as f:
    json.dump(report, f, indent=1)
print(json.dumps({k: v for k, v in report.items() if k != "npz_records"}, indent=1))
