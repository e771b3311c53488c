#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's metric:
Mrays/s (+ achieved HBM GB/s) for a 1024^2 camera over a 256^3 mock GRMHD snapshot.

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete render (camera -> geodesics -> sampling -> coefficients -> transfer ->
image in HBM) of the 1024^2 example_simulation camera over the 256^3 mock snapshot, both synthetic
and generated natively (blacklight_amd.mock restates the reference's generator). The grid is staged
into HBM once, before the timed region. Every step integrates its geodesics (bl_set_geodesic_reuse is
switched off here): what a series of snapshots gains from integrating them once, as the reference
does, is `--workload series8`, a line of its own.

Multi-GPU: one process per GPU over torch.distributed (nccl = RCCL). Started as the driver does
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) the ranks come from the environment;
started plainly (`python bench.py --gpus N`, no WORLD_SIZE) the parent starts that same launcher as a child process
BEFORE anything touches the GPU and exits with its code. --gpus must equal the number of ranks, and a box with fewer
GPUs than that fails loudly. For N > 1 the job is north_star's: ONE 1024^2 frame cut into 32 x 32-pixel tiles dealt
centre-first round-robin to the ranks (every rank gets the same mix of long and short rays), grid replicated, the
image RCCL-gathered on rank 0 inside the timed region, rank 0 de-tiles - strong scaling, value = rays of the frame /
max-over-ranks time. `--mode frames` is the weak-scaling variant (one full frame per rank, views at 360 r / N degrees).

Arithmetic tiers (include/blacklight_amd.h, bl_set_arithmetic): `value` is measured in the TOLERANT tier - the fp64
tolerance north_star grants for intensities (per-pixel L-infinity < 1e-6; measured ~1e-14), ray-step counts, flags and
cut decisions bit-exact - and the same K steps are then timed in the EXACT tier (bit-identical to the reference with
the pinned math library) and reported beside it as "exact_tier". `--arithmetic exact` times only that tier as `value`.

The JSON line also carries
  roofline     HBM-read roofline of the dominant kernel (the coefficient kernel, which issues the grid reads):
               algorithmic bytes (256 B per gathered sample + 13 B per ray) over its HIP-event time
  cpu_baseline the CPU oracle (a port of the reference's algorithm, oracle/) timed on this host's
               cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md; 6290 GB/s measured copy ceiling)
TILE = 32

# input/example_simulation.input of the reference with camera_resolution = 1024
WORKLOAD = dict(
    model_type="simulation", num_threads=1, output_format="npz", output_file="bench.npz",
    output_camera=False, checkpoint_geodesic_save=False, checkpoint_geodesic_load=False,
    checkpoint_sample_save=False, checkpoint_sample_load=False, simulation_format="athena",
    simulation_file="mock", simulation_multiple=False, simulation_coord="sks", simulation_a=0.0,
    simulation_m_msun=4.152e6, simulation_rho_cgs=1.0e-16, simulation_interp=True,
    simulation_block_interp=False, camera_type="plane", camera_r=50.0, camera_th=45.0,
    camera_ph=0.0, camera_urn=0.0, camera_uthn=0.0, camera_uphn=0.0, camera_k_r=1.0,
    camera_k_th=0.0, camera_k_ph=0.0, camera_rotation=0.0, camera_width=24.0,
    camera_resolution=1024, ray_flat=False, ray_terminate="multiplicative", ray_factor=1.005,
    ray_integrator="dp", ray_step=0.01, ray_max_steps=2000, ray_max_retries=20, ray_tol_abs=1.0e-8,
    ray_tol_rel=1.0e-8, image_light=True, image_num_frequencies=1, image_frequency=2.3e11,
    image_normalization="infinity", image_polarization=False, image_rotation_split=False,
    image_time=False, image_length=False, image_lambda=False, image_emission=False, image_tau=False,
    image_lambda_ave=False, image_emission_ave=False, image_tau_int=False, image_crossings=False,
    render_num_images=0, slow_light_on=False, adaptive_max_level=0, plasma_mu=0.5, plasma_ne_ni=1.0,
    plasma_model="ti_te_beta", plasma_use_p=True, plasma_rat_low=1.0, plasma_rat_high=10.0,
    plasma_power_frac=0.0, plasma_kappa_frac=0.0, cut_rho_min=-1.0, cut_rho_max=-1.0,
    cut_n_e_min=-1.0, cut_n_e_max=-1.0, cut_p_gas_min=-1.0, cut_p_gas_max=-1.0,
    cut_theta_e_min=-1.0, cut_theta_e_max=-1.0, cut_b_min=-1.0, cut_b_max=-1.0, cut_sigma_min=-1.0,
    cut_sigma_max=1.0, cut_beta_inverse_min=-1.0, cut_beta_inverse_max=-1.0, cut_omit_near=False,
    cut_omit_far=False, cut_omit_in=-1.0, cut_omit_out=-1.0, cut_midplane_theta=0.0,
    cut_midplane_z=0.0, cut_plane=False, fallback_nan=True,
)


def cpu_baseline(params_dict, grid, resolution, stride):
    """Time the CPU oracle (port of the reference algorithm) on a regular sub-lattice of the camera."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_api
    import blacklight_amd as bl
    from blacklight_amd import _capi
    p = bl.Params.from_dict(params_dict)
    idx = np.arange(stride // 2, resolution, stride)
    pixels = (idx[:, None] * resolution + idx[None, :]).reshape(-1).astype(np.int32)
    desc = grid.desc()
    cores = os.cpu_count() or 1
    out = oracle_api.render(p.ptr, desc, _capi.RenderDesc, _capi.CameraFrame, n_rays=pixels.size,
                            pixel_map=pixels, num_threads=cores)
    return pixels, out, cores


# BASELINE.json's configurations 2, 4 and 5 at size on ONE GPU (`--workload`): parity for them is in tests/ (reference windows and split
# properties at these sizes); these lines say how fast they run and what bounds them. Parameters: the reference's example inputs as
# the golden cases hold them (tests/golden), or the benchmark's workload with that configuration's physics switched on.
OTHER_WORKLOADS = {
    "formula512": "configuration 2: example_formula.input (formula mode, a = 0.9, camera at r = 1000, ray_max_steps = 7000), 512^2 camera",
    "polarized1024": "configuration 4's physics: full-Stokes polarized transfer + image_tau, 1024^2 plane camera over the 256^3 mock",
    "polarized_refined1024": "configuration 4's physics over the two-level refined mesh of refined256 (what real Athena++ / AthenaK dumps are)",
    "adaptive2048": "configuration 4: example_adaptive.input's refinement (8 x 8 blocks, one level, relative Laplacian) over a 2048^2 root camera, "
                    "full-Stokes polarized transfer + image_tau, 256^3 mock, whole adaptive loop",
    "truecolor1024x64": "configuration 5's physics: example_true_color.input's 64 frequencies (lin_wave, 1.5e11 ... 3.3e11 Hz), 1024^2 camera, 256^3 mock",
    # SURVEY.md 8(f) rows at the benchmark's size (parity for them: tests/; these lines say how fast they run)
    "refined256": "8(f)1 mesh refinement: the 256^3 mock as a two-level mesh (4 coarse + 32 fine MeshBlocks of 64^3 cells, scrambled), 1024^2 camera, "
                  "benchmark physics (simulation_sampling.cpp:352-394: block search per sample)",
    "refined256_deep": "the mesh of refined256 cut into 16^3-cell MeshBlocks (2 304 blocks, 24 distinct coordinate rows per axis, 16^3 boxes: the "
                       "table sizes of a deep hierarchy - beyond the LDS budgets of the kernels that search from LDS)",
    "blockinterp256": "8(f)1 inter-block interpolation: the 256^3 mock as 4 x 4 x 4 MeshBlocks of 64^3 cells with simulation_block_interp = true "
                      "(simulation_sampling.cpp:1068-1321; samples at the upper edge of the file's last block use the edge cell, BL_UNDEFINED_EDGE), 1024^2 camera",
    "slowlight10": "8(f)3 slow light: a window of 10 time slices of the 256^3 mock (5.4 GB of cells resident), interpolation in time, 1024^2 camera "
                   "(simulation_sampling.cpp:296-349, :736-912)",
    # the reference's production mode (simulation_multiple): geodesics once per series (blacklight.cpp:93-94 against the run loop :178-250)
    "series8": "a series of 8 snapshots of the 256^3 mock (cells differ from snapshot to snapshot, geometry the same), 1024^2 benchmark camera: "
               "frame 1 integrates the geodesics and leaves their sample records in HBM, frames 2 ... 8 shade them again (bl_set_geodesic_reuse)",
    "series8_refined": "the same series over the two-level refined mesh of refined256 (locate step inside the coefficient kernel, as on the "
                       "benchmark's grid; --arithmetic exact: frames 2 ... 8 also keep the located samples - the reference's first_time "
                       "sampling, radiation_integrator.cpp:693-704)",
    "series8_pipelined": "series8 with the next snapshot staged (bl_set_grid on a second host thread, into the second cell array) while the current one "
                         "renders: value = rays per second of the whole series' wall time, staging included",
}


def relative_distance(got, want):
    """north_star's bar, per pixel: max_m |got_m - want_m| / |want_m| over finite pixels with want_m > 0, the number of pixels above
    1e-6, and whether both frames are non-finite in the same places. Arrays of one shape (numpy)."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    use = np.isfinite(want) & np.isfinite(got) & (want > 0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(got[use] - want[use]) / want[use]
    return {"per_pixel_rel_linf": float(rel.max()) if rel.size else 0.0, "pixels_above_1e-6": int((rel > 1.0e-6).sum()),
            "pixels_compared": int(use.sum()),
            "same_nonfinite_and_nonpositive_pixels": bool(np.array_equal(np.isfinite(got) & (got > 0.0), np.isfinite(want) & (want > 0.0)))}


def pipelined_series(args, params, grids, image, sample_num, flags, device):
    """--workload series8_pipelined: frame n renders on this thread while a second host thread hands snapshot n + 1 to bl_set_grid (same
    geometry: its cells go up beside the render and take effect with the next one). The clock runs from the first bl_set_grid to the
    last image; frames are checked against the unpipelined series."""
    import threading
    import torch
    import blacklight_amd as bl
    n_frames, n_rays = len(grids), image.shape[1]
    line = None
    with bl.Context(bl.Params.from_dict(params), device=0) as ctx:
        ctx.set_arithmetic(args.arithmetic)
        ctx.set_reproducible(True)   # (so that the check below can ask for the same bits)
        ctx.follow_torch_stream(device)
        reference = []
        for rep in range(args.warmup + 1):
            ctx.set_geodesic_reuse(False)
            ctx.set_geodesic_reuse(True)
            images, stats = [], []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.set_grid(grids[0])
            staged = None

            def stage(grid, go):
                go.wait()           # (set on the way into bl_render, which takes the context's render lock first thing; this thread gets the
                ctx.set_grid(grid)  # interpreter when that call releases it: frame n is rendered from snapshot n, n + 1 goes up beside it)

            for n in range(n_frames):
                if staged is not None:
                    staged.join()          # snapshot n is in place
                go = threading.Event()
                if n + 1 < n_frames:
                    staged = threading.Thread(target=stage, args=(grids[n + 1], go))
                    staged.start()         # ... and n + 1 goes up while n renders
                else:
                    staged = None
                go.set()
                st = ctx.render_device(image.data_ptr(), n_rays, sample_num_ptr=sample_num.data_ptr(), sample_flags_ptr=flags.data_ptr())
                stats.append((int(st.geodesics_reused), st.ms_shade, st.ms_geodesic))
                if rep == args.warmup:
                    images.append(image.clone())
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        # the same series one step after the other: the same frames?
        ctx.set_geodesic_reuse(False)
        ctx.set_geodesic_reuse(True)
        differing = []
        for n in range(n_frames):
            ctx.set_grid(grids[n])
            ctx.render_device(image.data_ptr(), n_rays, sample_num_ptr=sample_num.data_ptr(), sample_flags_ptr=flags.data_ptr())
            torch.cuda.synchronize()
            differing.append(int((image.view(torch.int64) != images[n].view(torch.int64)).sum().item()))
        same = not any(differing)
        line = {
            "metric": "Mrays/sec of a whole series of 8 snapshots, grid staging included, next snapshot staged beside the render",
            "value": n_frames * n_rays / elapsed / 1.0e6, "unit": "Mrays/s", "n_gpus": 1, "steps": n_frames, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / n_frames, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": OTHER_WORKLOADS["series8_pipelined"], "arithmetic": args.arithmetic, "rays_per_step": n_rays, "parallelism": "1 GPU, 2 host threads"},
            "frames_reused_geodesics": [s[0] for s in stats], "shade_ms": [round(s[1], 2) for s in stats], "geodesic_ms": [round(s[2], 2) for s in stats],
            "frames_equal_the_unpipelined_series_bit_for_bit": same, "differing_values_per_frame": differing,
        }
    print(json.dumps(line), flush=True)


def series_workload(args):
    """--workload series8[_refined]: the reference's loop over snapshots (blacklight.cpp:178-250) on one GPU. Per frame: bl_set_grid of
    the snapshot's cells (timed apart: it is the reader's side of the boundary), then bl_render into HBM (timed). One JSON line:
    value = rays / s over frames 2 ... 8; frame 1 and the whole series beside it."""
    import dataclasses
    import torch
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import blacklight_amd as bl
    from blacklight_amd import mock
    name = args.workload
    n_frames = 8
    res = args.resolution
    params = dict(WORKLOAD, camera_resolution=res, simulation_multiple=True, simulation_start=0, simulation_end=n_frames - 1)
    base = mock.generate(n_r=args.grid, n_th=args.grid, n_ph=args.grid)
    if name == "series8_refined":
        import golden_util as gu
        base = gu.refined_grid(base, block=(args.grid // 4,) * 3)

    def snapshot(n):
        # density and pressure of snapshot n (the fields a movie changes most); geometry, velocities and field lines as in snapshot 0
        prim = base.prim.copy()
        prim[0:2] *= np.float32(1.0 + 0.07 * n)
        return dataclasses.replace(base, prim=prim)

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_rays = res * res
    image = torch.zeros((1, n_rays), dtype=torch.float64, device=device)
    sample_num = torch.zeros(n_rays, dtype=torch.int32, device=device)
    flags = torch.zeros(n_rays, dtype=torch.uint8, device=device)
    if name == "series8_pipelined":
        return pipelined_series(args, params, [snapshot(n) for n in range(n_frames)], image, sample_num, flags, device)
    frames = []
    with bl.Context(bl.Params.from_dict(params), device=0) as ctx:
        ctx.set_arithmetic(args.arithmetic)
        ctx.set_reproducible(args.reproducible)
        ctx.follow_torch_stream(device)
        for rep in range(args.warmup + 1):   # (warm-up series first: allocations, first use of the kernels; the last series is the one reported)
            frames = []
            ctx.set_geodesic_reuse(False)    # forget what an earlier series left
            ctx.set_geodesic_reuse(True)
            for n in range(n_frames):
                grid = snapshot(n)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.set_grid(grid)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                st = ctx.render_device(image.data_ptr(), n_rays, sample_num_ptr=sample_num.data_ptr(), sample_flags_ptr=flags.data_ptr())
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                frames.append(dict(ms_set_grid=1000.0 * (t1 - t0), ms_render=1000.0 * (t2 - t1), geodesics_reused=int(st.geodesics_reused),
                                   sampling_reused=int(st.sampling_reused), launches_geodesic=int(st.launches_geodesic), launches_locate=int(st.launches_locate),
                                   kernel_ms={"geodesic": st.ms_geodesic, "locate": st.ms_locate, "shade": st.ms_shade, "transfer": st.ms_transfer},
                                   n_gathers=int(st.n_gathers), n_samples=int(st.n_samples)))
        last_stats = st
        # the last frame again with its geodesics integrated afresh: the same frame?
        reused_image, reused_num = image.clone(), sample_num.clone()
        ctx.set_geodesic_reuse(False)
        fresh = ctx.render_device(image.data_ptr(), n_rays, sample_num_ptr=sample_num.data_ptr(), sample_flags_ptr=flags.data_ptr())
        torch.cuda.synchronize()
        check = relative_distance(reused_image.cpu().numpy(), image.cpu().numpy())
        check.update(sample_num_equal=bool((reused_num == sample_num).all().item()), bit_identical=bool(torch.equal(reused_image.view(torch.int64), image.view(torch.int64))),
                     fresh_render_launches_geodesic=int(fresh.launches_geodesic), composed_maps=int(fresh.composed_maps))
    later = frames[1:]
    ms_later = sum(f["ms_render"] for f in later) / len(later)
    ms_all = sum(f["ms_render"] for f in frames)
    gathers = later[-1]["n_gathers"]
    line = {
        "metric": "Mrays/sec, frames 2 ... 8 of a series of snapshots (geodesics once per series), 1024^2 camera over 256^3 GRMHD grid",
        "value": n_rays / (ms_later * 1.0e-3) / 1.0e6, "unit": "Mrays/s", "n_gpus": 1, "steps": len(later), "warmup": args.warmup,
        "ms_per_step": ms_later, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": OTHER_WORKLOADS[name] + "; image rows stay in HBM; bl_set_grid of each snapshot timed apart",
                   "arithmetic": "tolerant" if last_stats.arithmetic == 1 else "exact", "rays_per_step": n_rays,
                   "samples_per_ray": later[-1]["n_samples"] / n_rays, "parallelism": "1 GPU", "fused_variant": int(last_stats.fused_variant)},
        "frame_1": {"ms_render": frames[0]["ms_render"], "mrays_per_s": n_rays / (frames[0]["ms_render"] * 1.0e-3) / 1.0e6, "launches_geodesic": frames[0]["launches_geodesic"],
                    "kernel_ms": frames[0]["kernel_ms"]},
        "frames_2_to_8": {"ms_render_mean": ms_later, "ms_render_min": min(f["ms_render"] for f in later), "ms_render_max": max(f["ms_render"] for f in later),
                          "all_reused_geodesics": all(f["geodesics_reused"] == 1 and f["launches_geodesic"] == 0 for f in later),
                          "all_reused_located_samples": all(f["sampling_reused"] == 1 for f in later),
                          "launches_locate": [f["launches_locate"] for f in later], "kernel_ms_mean": {k: sum(f["kernel_ms"][k] for f in later) / len(later) for k in later[0]["kernel_ms"]}},
        "whole_series": {"frames": n_frames, "ms_render_total": ms_all, "mrays_per_s": n_frames * n_rays / (ms_all * 1.0e-3) / 1.0e6,
                         "ms_set_grid_mean": sum(f["ms_set_grid"] for f in frames) / n_frames},
        "reused_vs_fresh_last_frame": check,
        "hbm_gbs_algorithmic_whole_pipeline": (256.0 * gathers + 13.0 * n_rays) / (ms_later * 1.0e-3) / 1.0e9,
        "hbm_frac_algorithmic_whole_pipeline": (256.0 * gathers + 13.0 * n_rays) / (ms_later * 1.0e-3) / 1.0e9 / HBM_PEAK_GBS,
        "switches": int(last_stats.switches),
    }
    shade_ms = line["frames_2_to_8"]["kernel_ms_mean"]["shade"]
    line["roofline"] = {"bound": "hbm", "kernel": "coefficient kernel of frames 2 ... 8", "achieved": (256.0 * gathers + 13.0 * n_rays) / (shade_ms * 1.0e-3) / 1.0e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (256.0 * gathers + 13.0 * n_rays) / (shade_ms * 1.0e-3) / 1.0e9 / HBM_PEAK_GBS, "traffic": None}
    print(json.dumps(line), flush=True)


def other_workload(args):
    """One of OTHER_WORKLOADS on one GPU: W warm-up renders, K timed ones, one JSON line in bench.py's schema."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import blacklight_amd as bl
    from blacklight_amd import mock
    name = args.workload
    grid = None
    if name == "formula512":
        import golden_util as gu
        params = dict(gu.load_case("formula_dp")[1], camera_resolution=512)
    else:
        params = dict(WORKLOAD)
        grid = mock.generate(n_r=args.grid, n_th=args.grid, n_ph=args.grid)
        if name in ("polarized1024", "adaptive2048", "polarized_refined1024"):
            params.update(image_polarization=True, image_tau=True)
        if name == "polarized_refined1024":
            import golden_util as gu
            grid = gu.refined_grid(grid, block=(args.grid // 4,) * 3)
        if name == "adaptive2048":
            params.update(camera_resolution=2048, adaptive_max_level=1, adaptive_block_size=8, adaptive_frequency_num=1, adaptive_val_cut=0.0,
                          adaptive_val_frac=-1.0, adaptive_abs_grad_cut=0.0, adaptive_abs_grad_frac=-1.0, adaptive_rel_grad_cut=0.0,
                          adaptive_rel_grad_frac=-1.0, adaptive_abs_lapl_cut=0.0, adaptive_abs_lapl_frac=-1.0, adaptive_rel_lapl_cut=1.0,
                          adaptive_rel_lapl_frac=0.25, adaptive_num_regions=0)
        if name == "truecolor1024x64":
            params.update(image_num_frequencies=64, image_frequency_start=1.5e11, image_frequency_end=3.3e11, image_frequency_spacing="lin_wave")
        if name in ("refined256", "refined256_deep"):
            import golden_util as gu
            grid = gu.refined_grid(grid, block=(args.grid // 4,) * 3)
            if name == "refined256_deep":
                grid = gu.subdivide_blocks(grid, 4)
        if name == "blockinterp256":
            import golden_util as gu
            grid = gu.split_grid(grid, 4, 4, 4)
            params.update(simulation_block_interp=True)
        if name == "slowlight10":
            # the window's slices 20 M apart, the latest at the camera's time: the rays' samples (coordinate times 0 ... -110 M from the
            # camera at r = 50) spread over the first six slices
            params.update(slow_light_on=True, slow_interp=True, slow_chunk_size=10, slow_t_start=180.0, slow_dt=20.0, slow_num_images=1, slow_offset=0,
                          simulation_multiple=True, simulation_start=0, simulation_end=9)
    with bl.Context(bl.Params.from_dict(params)) as ctx:
        ctx.set_geodesic_reuse(False)   # every step a complete render: geodesics included
        if name == "slowlight10":
            for n in range(10):   # slice n: n = 0 the latest file (simulation_reader.cpp:211-303)
                ctx.set_grid_slice(n, grid, 180.0 - 20.0 * n)
            ctx.set_snapshot(0)
        elif grid is not None:
            if name == "blockinterp256":
                ctx.set_undefined_policy("edge")
            ctx.set_grid(grid)
        ctx.set_arithmetic(args.arithmetic)
        if name == "adaptive2048":
            render = ctx.render_adaptive
        else:
            # the frame loop's own result buffers, pinned (bl_host_alloc) and used again for every step: what a series of frames does
            n_pix = int(params["camera_resolution"]) ** 2
            buffers = dict(image=ctx.pinned_array((ctx.num_quantities, n_pix)), sample_num=ctx.pinned_array(n_pix, np.int32),
                           sample_flags=ctx.pinned_array(n_pix, np.uint8))

            def render():
                return ctx.render(out=buffers)
        for _ in range(args.warmup):
            render()
        ms = dict(geodesic=0.0, locate=0.0, shade=0.0, transfer=0.0, wall=0.0)
        rays = gathers = samples = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = render()
            for st in ([lv["stats"] for lv in out] if name == "adaptive2048" else [out["stats"]]):
                ms["geodesic"] += st.ms_geodesic
                ms["locate"] += st.ms_locate
                ms["shade"] += st.ms_shade
                ms["transfer"] += st.ms_transfer
                ms["wall"] += st.ms_wall
                rays += st.n_rays
                gathers += st.n_gathers
                samples += st.n_samples
        elapsed = time.perf_counter() - t0
        st = (out[0] if name == "adaptive2048" else out)["stats"]
    rays_per_step = rays / args.steps
    line = {
        "metric": "Mrays/sec + achieved HBM GB/s", "value": rays / elapsed / 1.0e6, "unit": "Mrays/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": OTHER_WORKLOADS[name] + "; results on the host, in pinned buffers the loop reuses (PCIe download of the image rows inside the time)",
                   "arithmetic": "tolerant" if st.arithmetic == 1 else "exact", "rays_per_step": rays_per_step,
                   "samples_per_ray": samples / max(rays, 1), "parallelism": "1 GPU", "chunks_per_step": st.n_chunks},
        "kernel_ms_per_step": {k: v / args.steps for k, v in ms.items()},
        "switches": int(st.switches),
    }
    if gathers > 0 and ms["shade"] > 0.0:
        algorithmic = 256.0 * gathers + 13.0 * rays
        line["hbm_gbs_algorithmic_whole_pipeline"] = algorithmic / elapsed / 1.0e9
        line["roofline"] = {"bound": "hbm", "kernel": "coefficient kernels of the run (see profiles/ for the kernel trace)",
                            "achieved": algorithmic / (ms["shade"] * 1.0e-3) / 1.0e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": algorithmic / (ms["shade"] * 1.0e-3) / 1.0e9 / HBM_PEAK_GBS, "traffic": None}
    else:
        line["roofline"] = None   # no grid is read (formula mode): what bounds the frame is instruction issue, see profiles/README.md
    print(json.dumps(line), flush=True)


def visible_gpus():
    """GPUs of this box as the kernel driver lists them (KFD topology nodes with SIMDs), without initialising HIP or importing
    torch in the launching process: 0 without the driver, None when the topology is there but cannot be read (the ranks
    themselves then report a shortfall)."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(nodes):
        return 0
    try:
        count = 0
        for node in os.listdir(nodes):
            with open(os.path.join(nodes, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
        return count
    except OSError:
        return None


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process and return its exit
    code. The launching process neither imports torch nor touches HIP: it holds no GPU context while the ranks run."""
    visible = visible_gpus()
    if "--rehearse" in argv:
        visible = None
    if visible is not None and visible < n_gpus:
        raise SystemExit(f"bench.py --gpus {n_gpus}: only {visible} GPU(s) visible on this box - refusing to run a smaller job "
                         "under that name")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def kernel_source_hash():
    """Hash of the sources the kernels and their launches are built from (blacklight_amd/csrc: every .hip file - kernels, planning
    and launch code - and every header; not the host-only readers, parser, writers and command line, which cannot move a kernel's
    traffic): profiles/hbm_traffic.json carries the hash of the tree its counters were measured on, and the bench line says when
    the two differ."""
    import hashlib
    csrc = os.path.join(REPO, "blacklight_amd", "csrc")
    digest = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(csrc, name), "rb") as f:
                digest.update(name.encode() + b"\0" + f.read())
    return digest.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["frames", "tiled"], default="tiled", help="N > 1: one frame tiled over the ranks (default) or one frame per rank")
    ap.add_argument("--arithmetic", choices=["tolerant", "exact"], default="tolerant",
                    help="tier `value` is measured in; with tolerant the exact tier is timed afterwards and reported as exact_tier")
    ap.add_argument("--reproducible", action="store_true",
                    help="tolerant tier: bl_set_reproducible (one transfer record per sample: images bit-identical from run to run and between a "
                         "frame and its tiles) instead of composed maps")
    ap.add_argument("--resolution", type=int, default=1024)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scratch-gib", type=float, default=0.0, help="override the library's scratch budget (GiB); 0 = default")
    ap.add_argument("--cpu-stride", type=int, default=2, help="CPU baseline traces every stride-th pixel per axis")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 without N GPUs: the ranks share GPU 0 and gather over gloo (CPU tensors); rank 0 also renders the whole frame "
                         "alone and reports whether the assembled frame equals it bit for bit. A rehearsal of the code path, not a measurement")
    ap.add_argument("--workload", choices=["benchmark"] + sorted(OTHER_WORKLOADS), default="benchmark",
                    help="BASELINE.json's other configurations at their own sizes on one GPU (same JSON schema, not the driver's line)")
    args = ap.parse_args()
    if args.workload != "benchmark":
        if args.gpus != 1:
            raise SystemExit("bench.py --workload: the other configurations are timed on one GPU")
        return series_workload(args) if args.workload.startswith("series8") else other_workload(args)
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE = {world}: the two must agree")

    import torch
    import blacklight_amd as bl
    from blacklight_amd import distributed as bd
    from blacklight_amd import mock

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    rehearse = distributed and args.rehearse
    if rehearse:
        local_rank = 0   # every rank on the one GPU
    if torch.cuda.device_count() < (world if distributed and not rehearse else 1):
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    res = args.resolution
    tiled = distributed and args.mode == "tiled"
    params_dict = dict(WORKLOAD)
    params_dict["camera_resolution"] = res
    if distributed and not tiled:
        params_dict["camera_ph"] = 360.0 * rank / world
    params = bl.Params.from_dict(params_dict)
    grid = mock.generate(n_r=args.grid, n_th=args.grid, n_ph=args.grid)

    ctx = bl.Context(params, device=local_rank)
    ctx.set_grid(grid)          # staged into HBM once, outside the timed region
    ctx.set_geodesic_reuse(False)   # every step is a complete render: its geodesics are integrated, not taken from the step before
    ctx.set_reproducible(args.reproducible)
    if args.scratch_gib > 0.0:
        ctx.set_scratch_limit(int(args.scratch_gib * (1 << 30)))

    if tiled:
        layout = bd.frame_layout(res, world, TILE)     # built once: per-rank pixel lists and, on rank 0's GPU, the way back
        pixels = layout.pixels[rank]
        n_rays = int(pixels.size)
        n_padded = layout.n_padded                     # equal shares for the gather, whatever the world size
    else:
        layout, pixels = None, None
        n_rays = n_padded = res * res
    image = torch.zeros((1, n_padded), dtype=torch.float64, device=device)
    sample_num = torch.zeros(n_padded, dtype=torch.int32, device=device)
    flags = torch.zeros(n_padded, dtype=torch.uint8, device=device)
    # The library renders on streams of its own. bl_set_caller_stream orders every render behind what torch's current stream
    # holds at the call - the fills above, and the RCCL gather of the previous frame (torch makes its current stream wait for a
    # collective it has issued): a rank's next frame cannot write `image` while the gather still reads it. Waits on the device.
    ctx.follow_torch_stream(device)
    comm = bd.Comm(device=torch.device("cpu") if rehearse else device) if distributed else None

    def step():
        return ctx.render_device(image.data_ptr(), n_rays, pixel_map=pixels,
                                 sample_num_ptr=sample_num.data_ptr(), sample_flags_ptr=flags.data_ptr())

    def gather_image():
        """Final image(s) to rank 0 over RCCL (part of the job, inside the timed region): one gather into a buffer rank 0 keeps,
        one index_select through the layout's cached permutation on rank 0's GPU. No numpy, no upload, no loop over ranks."""
        if not distributed:
            return None
        gathered = comm.gather_flat(image.cpu().reshape(-1) if rehearse else image.reshape(-1), dst=0)
        if gathered is not None and tiled:
            return layout.detile(gathered, 1)         # rank 0 de-tiles into (n_q, res*res)
        return gathered

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(tier):
        """W warm-up steps, then exactly K steps between two fences; max over ranks."""
        ctx.set_arithmetic(tier)
        for _ in range(args.warmup):
            step()
            gather_image()
        fence()
        t0 = time.perf_counter()
        ms = dict(geodesic=0.0, locate=0.0, shade=0.0, transfer=0.0, wall=0.0)
        launches = 0
        stats = None
        for _ in range(args.steps):
            stats = step()
            ms["geodesic"] += stats.ms_geodesic
            ms["locate"] += stats.ms_locate
            ms["shade"] += stats.ms_shade
            ms["wall"] += stats.ms_wall
            ms["transfer"] += stats.ms_transfer
            launches += stats.launches_shade
            gather_image()
        fence()
        elapsed = time.perf_counter() - t0
        per_rank = [elapsed]
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank = [float(x.item()) for x in every]
            elapsed = max(per_rank)
            counts = torch.tensor([float(n_rays), float(stats.n_gathers), float(stats.n_samples)], dtype=torch.float64, device="cpu" if rehearse else device)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
            totals = [float(x) for x in counts.tolist()]
        else:
            totals = [float(n_rays), float(stats.n_gathers), float(stats.n_samples)]
        shade_ms_per_launch = ms["shade"] / max(launches, 1)
        bytes_per_launch = stats.algorithmic_bytes / max(stats.launches_shade, 1)
        return dict(elapsed=elapsed, per_rank=per_rank, ms=ms, stats=stats, totals=totals, shade_ms_per_launch=shade_ms_per_launch,
                    bytes_per_launch=bytes_per_launch, tier_ran="tolerant" if stats.arithmetic == 1 else "exact")

    main_run = timed(args.arithmetic)
    exact_run = timed("exact") if args.arithmetic == "tolerant" else None
    tier_distance = None
    if exact_run is not None:
        torch.cuda.synchronize()
        exact_image = image.clone()
        exact_num = sample_num.clone()
        ctx.set_arithmetic("tolerant")
        step()   # leave the headline tier's frame in `image` for the parity cross-check below
        torch.cuda.synchronize()
        # the two tiers against each other over the whole frame of this rank (the exact tier is bit-identical to the reference)
        both_nan = torch.isnan(image) & torch.isnan(exact_image)
        diff = torch.where(both_nan, torch.zeros_like(image), (image - exact_image).abs())
        peak = torch.nan_to_num(exact_image.abs(), nan=0.0).max()
        tier_distance = {"image_linf_over_max": float((torch.nan_to_num(diff, nan=float("inf")).max() / peak).item()),
                         "nan_mask_equal": bool((torch.isnan(image) == torch.isnan(exact_image)).all().item()),
                         "sample_num_equal": bool((sample_num == exact_num).all().item()), "pixels": int(n_rays)}
        # ... and the way north_star words the bar: per pixel, relative to that pixel's own intensity
        tier_distance.update(relative_distance(image[0, :n_rays].cpu().numpy(), exact_image[0, :n_rays].cpu().numpy()))

    # A short series on the side (N = 1): the reference's production mode integrates the geodesics once per series - frames 2 ... 4 of
    # four snapshots shade the records frame 1 left (`--workload series8` is the full measurement; `value` above is a complete render)
    series = None
    if not distributed:
        import dataclasses
        ctx.set_arithmetic(args.arithmetic)
        ctx.set_geodesic_reuse(True)
        frame_ms, staged_ms, reused = [], [], []
        for n in range(4):
            prim = grid.prim.copy()
            prim[0:2] *= np.float32(1.0 + 0.07 * n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.set_grid(dataclasses.replace(grid, prim=prim))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st = step()
            torch.cuda.synchronize()
            frame_ms.append(1000.0 * (time.perf_counter() - t1))
            staged_ms.append(1000.0 * (t1 - t0))
            reused.append(int(st.geodesics_reused))
        ctx.set_geodesic_reuse(False)
        ctx.set_grid(grid)
        step()   # (the benchmark's own frame back in `image` for the cross-check against the oracle below)
        torch.cuda.synchronize()
        later = frame_ms[1:]
        series = {"frames": 4, "frame_1_ms": frame_ms[0], "frames_2_to_4_ms": sum(later) / len(later), "geodesics_reused": reused,
                  "mrays_per_s_frames_2_to_4": n_rays / (sum(later) / len(later) * 1.0e-3) / 1.0e6, "ms_set_grid": sum(staged_ms) / len(staged_ms)}

    rehearsal = None
    if rehearse and tiled:
        # the frame the ranks put together against the frame one rank renders alone: same bits, or the tiling / padding /
        # gather / de-tiling of this very code path is wrong somewhere
        ctx.set_arithmetic(args.arithmetic)
        step()
        torch.cuda.synchronize()
        assembled = gather_image()
        gathered_nums = bd.gather_rows(sample_num.cpu(), dst=0)
        if rank == 0:
            alone = torch.zeros((1, res * res), dtype=torch.float64, device=device)
            num_alone = torch.zeros(res * res, dtype=torch.int32, device=device)
            torch.cuda.synchronize()
            ctx.render_device(alone.data_ptr(), res * res, sample_num_ptr=num_alone.data_ptr())
            torch.cuda.synchronize()
            nums = layout.detile(torch.stack([part.reshape(-1) for part in gathered_nums]), 1).reshape(-1)
            x, y = assembled.cpu().numpy(), alone.cpu().numpy()
            a, b = x.view(np.uint64), y.view(np.uint64)
            with np.errstate(invalid="ignore"):
                distance = float(np.nanmax(np.where(np.isnan(x) & np.isnan(y), 0.0, np.abs(x - y))) / np.nanmax(np.abs(y)))
            rehearsal = {"assembled_frame_equals_single_rank_frame_bit_for_bit": bool(np.array_equal(a, b)), "pixels": int(res * res),
                         "differing_pixels": int((a != b).sum()), "image_linf_over_max": distance,
                         "nan_mask_equal": bool(np.array_equal(np.isnan(x), np.isnan(y))),
                         "sample_num_equal": bool(np.array_equal(nums.cpu().numpy(), num_alone.cpu().numpy())),
                         "ranks_on_one_gpu": world, "collectives": "gloo"}
        dist.barrier()
    if rank == 0:
        stats = main_run["stats"]
        total_rays, total_gathers, total_samples = main_run["totals"]
        elapsed = main_run["elapsed"]
        ms_per_step = 1000.0 * elapsed / args.steps
        value = total_rays / (elapsed / args.steps) / 1.0e6
        achieved = main_run["bytes_per_launch"] / (main_run["shade_ms_per_launch"] * 1.0e-3) / 1.0e9
        traffic, traffic_source, traffic_stale, traffic_whole = None, None, None, None
        traffic_file = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file) and not distributed:
            with open(traffic_file) as f:
                recorded = json.load(f)
            entry = recorded.get(main_run["tier_ran"], {})
            traffic = entry.get("coefficient_kernel_bytes_per_launch")
            traffic_whole = entry.get("whole_pipeline_bytes_per_frame")
            traffic_source = entry.get("source")
            traffic_stale = recorded.get("csrc_sha256_16") != kernel_source_hash()
        tier_text = {"tolerant": "tolerant arithmetic tier (intensities within north_star's fp64 tolerance, counts / flags / cuts bit-exact)",
                     "exact": "exact arithmetic tier (bit-identical to the reference with the pinned math library)"}
        line = {
            "metric": "Mrays/sec + achieved HBM GB/s, 1024^2 camera over 256^3 GRMHD grid",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if args.mode == "frames" else "strong",   # (the N = 1 point of the tiled curve is that curve's)
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{res}x{res} plane camera (example_simulation.input) over {args.grid}^3 mock Athena++ GRMHD "
                            "snapshot, DP integrator, trilinear sampling, thermal synchrotron 230 GHz, unpolarized; "
                            + tier_text[main_run["tier_ran"]],
                "arithmetic": main_run["tier_ran"],
                "rays_per_step": total_rays, "samples_per_ray": total_samples / total_rays,
                "parallelism": ("1 GPU" if not distributed else
                                (f"{world} GPUs, one {res}^2 frame in {TILE}x{TILE} tiles dealt centre-first round-robin, grid replicated, image gathered over RCCL"
                                 if tiled else
                                 f"{world} GPUs, one {res}^2 frame per GPU (views at 360/N deg), grid replicated, frames gathered over RCCL")),
                "chunks_per_step": stats.n_chunks,
                "geodesics_integrated_per_step": int(stats.launches_geodesic),   # 1: every step integrates its rays (no reuse of an earlier step's)
                # bit-identical images from run to run and however the frame is cut? (exact tier: always; tolerant tier: with
                # --reproducible; otherwise equal to rounding, ~1e-15 of the maximum - bl_stats.composed_maps)
                "bit_reproducible": not bool(stats.composed_maps),
            },
            "ranks_seen_by_rccl": world,
            "ms_per_step_per_rank": {"min": 1000.0 * min(main_run["per_rank"]) / args.steps, "max": 1000.0 * max(main_run["per_rank"]) / args.steps},
            "kernel_ms_per_step": {k: v / args.steps for k, v in main_run["ms"].items()},
            "hbm_gbs_algorithmic_whole_pipeline": (256.0 * total_gathers + 13.0 * total_rays) / (elapsed / args.steps) / 1.0e9,
            "hbm_frac_algorithmic_whole_pipeline": (256.0 * total_gathers + 13.0 * total_rays) / (elapsed / args.steps) / 1.0e9 / HBM_PEAK_GBS,
            # the coefficient kernel of the tier that ran: with no launch of a locate kernel it is the one that locates as well
            "roofline": {"bound": "hbm", "kernel": (("bl_shade_fused2_kernel" if stats.fused_variant == 2 else "bl_shade_fast_kernel")
                                                    if main_run["tier_ran"] == "tolerant" else ("bl_shade_exact2_kernel" if stats.fused_variant == 3 else "bl_shade_exact_kernel")),
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_whole_pipeline": traffic_whole, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": main_run["bytes_per_launch"], "ms_per_launch": main_run["shade_ms_per_launch"]},
        }
        if series is not None:
            line["series"] = series
        if rehearsal is not None:
            line["rehearsal"] = rehearsal
            line["data"] = "synthetic; REHEARSAL: the ranks shared one GPU and gathered over gloo - times mean nothing"
        line["switches"] = int(stats.switches)   # BL_SWITCH_* measurement switches active in this run (0: none)
        if tier_distance is not None:
            line["tolerant_vs_exact"] = tier_distance
        if exact_run is not None:
            e_elapsed = exact_run["elapsed"]
            e_achieved = exact_run["bytes_per_launch"] / (exact_run["shade_ms_per_launch"] * 1.0e-3) / 1.0e9
            line["exact_tier"] = {
                "value": exact_run["totals"][0] / (e_elapsed / args.steps) / 1.0e6, "unit": "Mrays/s", "ms_per_step": 1000.0 * e_elapsed / args.steps,
                "kernel_ms_per_step": {k: v / args.steps for k, v in exact_run["ms"].items()},
                "roofline": {"kernel": "bl_shade_exact2_kernel" if exact_run["stats"].fused_variant == 3 else "bl_shade_exact_kernel", "achieved": e_achieved, "frac": e_achieved / HBM_PEAK_GBS,
                             "ms_per_launch": exact_run["shade_ms_per_launch"]},
            }
        if not distributed and not args.no_cpu_baseline:
            pixels_cpu, cpu, cores = cpu_baseline(params_dict, grid, res, args.cpu_stride)
            line["cpu_baseline"] = {
                "value": pixels_cpu.size / cpu["seconds"] / 1.0e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": f"one pixel in {args.cpu_stride} per axis of the same {res}^2 camera "
                          f"({pixels_cpu.size} rays, {cpu['seconds']:.1f} s), same grid, OpenMP over all host threads",
            }
            # cross-check while we are here: the GPU frame (of the tier `value` was measured in) against the oracle
            gpu_img = image[0].cpu().numpy()[pixels_cpu]
            gpu_num = sample_num.cpu().numpy()[pixels_cpu]
            same = (gpu_img == cpu["image"][0]) | (np.isnan(gpu_img) & np.isnan(cpu["image"][0]))
            with np.errstate(invalid="ignore"):
                distance = float(np.nanmax(np.abs(gpu_img - cpu["image"][0])) / np.nanmax(np.abs(cpu["image"][0])))
            line["parity_vs_oracle_on_sample"] = {
                "pixels": int(pixels_cpu.size), "image_bit_exact": bool(same.all()), "image_linf_over_max": distance,
                "nan_mask_equal": bool(np.array_equal(np.isnan(gpu_img), np.isnan(cpu["image"][0]))),
                "sample_num_bit_exact": bool(np.array_equal(gpu_num, cpu["sample_num"])),
            }
            line["parity_vs_oracle_on_sample"].update(relative_distance(gpu_img, cpu["image"][0]))
        print(json.dumps(line), flush=True)

    ctx.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
