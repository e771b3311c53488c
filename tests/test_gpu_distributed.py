"""Multi-GPU product path on the one GPU of the test box: two (three) ranks share cuda:0, collectives over gloo, the
rendering through the real library. blacklight_amd.distributed.render_adaptive must give rank 0 exactly what one GPU
gives: the adaptive cases' .npz files equal the reference's record for record (tier B, bit-exact), and the warning
carries the totals of the level, in the reference's words."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import golden_util as gu
from test_gpu_adaptive_cli import _assert_npz_equals_golden

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("case,world", [("sim_adaptive", 2), ("sim_polarized_adaptive", 2), ("formula_adaptive_multifreq", 3)])
def test_adaptive_cases_over_emulated_ranks(case, world, built_library, tmp_path):
    import dist_worker
    fx, params, mock_args = gu.load_case(case)
    out_path = str(tmp_path / "out.npz")
    mp.spawn(dist_worker.worker, args=(world, _free_port(), "gpu", params, mock_args, True, out_path), nprocs=world, join=True)
    _assert_npz_equals_golden(np.load(out_path), fx)
    assert open(out_path + ".warnings").read() == str(fx["B_warnings"])


def test_warning_carries_the_totals_of_the_level(built_library, tmp_path):
    """sim_few_steps: 252 of 256 rays run into ray_max_steps. Each rank sees its own share; the text must be the
    single-GPU one (geodesics.cpp:389-394), max_sample_num the maximum over the ranks."""
    import dist_worker
    fx, params, mock_args = gu.load_case("sim_few_steps")
    out_path = str(tmp_path / "out.npz")
    mp.spawn(dist_worker.worker, args=(2, _free_port(), "gpu", params, mock_args, False, out_path), nprocs=2, join=True)
    _assert_npz_equals_golden(np.load(out_path), fx)
    assert open(out_path + ".warnings").read() == str(fx["B_warnings"]) == "Warning: 252 out of 256 geodesics terminate unexpectedly.\n"
    counts = np.load(out_path + ".counts.npy")
    assert counts[0] == int(fx["B_geodesic_num_steps"]) and counts[1] == 252


@pytest.mark.parametrize("case,devices", [("sim_adaptive", 2), ("sim_polarized_adaptive", 3), ("sim_few_steps", 2), ("sim_render_light", 2)])
def test_command_line_driver_over_several_devices(case, devices, built_library, tmp_path):
    """bin/blacklight_amd with BLACKLIGHT_AMD_DEVICES = N: one process, one host thread and one context per device (here N
    contexts on the one GPU of the box), every level cut over them and put back together. Output files and warnings are
    the single-device ones, i.e. the reference's."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "blacklight_amd", "bin", "blacklight_amd")
    fx, params, mock_args = gu.load_case(case)
    params = dict(params)
    params["output_file"] = str(tmp_path / "image.npz")
    grid_path = tmp_path / "grid.blgrid"
    gu.golden_grid(mock_args).save_raw(grid_path)
    params["simulation_file"] = str(grid_path)
    input_path = tmp_path / "case.input"
    with open(input_path, "w") as f:
        for key, value in params.items():
            f.write(f"{key} = {value}\n")
    run = subprocess.run([exe, str(input_path)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, BLACKLIGHT_AMD_DEVICES=str(devices)))
    assert run.returncode == 0, run.stdout + run.stderr
    _assert_npz_equals_golden(np.load(params["output_file"]), fx)
    assert run.stderr == str(fx["B_warnings"])


def _device_path_worker(rank, port, params, mock_args, out_path):
    """One rank over nccl (RCCL): the product path of a multi-GPU run - bl_render into torch tensors in HBM, gather on the
    device, one download on rank 0 - must give what the plain single-GPU loop gives."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import blacklight_amd as bl
    from blacklight_amd import distributed as bd
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    comm = bd.Comm()
    assert comm.on_gpu
    with bl.Context(bl.Params.from_dict(params), device=0) as ctx:
        if mock_args is not None:
            ctx.set_grid(gu.golden_grid(mock_args))
        levels = ctx.render_adaptive(want_camera=True, distributed=True, comm=comm)
        plain = ctx.render_adaptive(want_camera=True)
        assert len(levels) == len(plain)
        for got, want in zip(levels, plain):
            for key in ("image", "sample_num", "sample_flags", "camera_pos", "camera_dir", "rendering"):
                if want.get(key) is None:
                    assert got.get(key) is None
                else:
                    assert gu.same_bits(np.asarray(got[key], dtype=np.float64), np.asarray(want[key], dtype=np.float64)).all(), key
        ctx.write_output(levels, path=out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["sim_adaptive", "sim_polarized_adaptive", "sim_render_light"])
def test_device_resident_gather_over_rccl(case, built_library, tmp_path):
    fx, params, mock_args = gu.load_case(case)
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_device_path_worker, args=(_free_port(), params, mock_args, out_path), nprocs=1, join=True)
    _assert_npz_equals_golden(np.load(out_path), fx)


@pytest.mark.parametrize("tier", ["exact", "tolerant", "tolerant-reproducible"])
def test_bench_tiled_frame_rehearsal(tier, built_library):
    """bench.py's own multi-GPU code path - tile dealing, padded shares, gather, de-tiling - with four ranks that share the one
    GPU and gather over gloo (`--rehearse`; a box allows six processes on its card, hence four ranks and not eight): the frame
    the ranks assemble equals the frame rank 0 renders alone - bit for bit in the exact tier; in the tolerant tier to rounding
    level (which of a ray's affine maps are composed with which follows the order in which the persistent geodesic kernel emitted
    the records, and that differs between a share and the whole frame), NaN masks and counts equal; and bit for bit again in the
    tolerant tier under bl_set_reproducible (`--reproducible`: one transfer record per sample). No 8-GPU node was available
    to this builder: the RCCL path itself runs with one rank (above) and in the driver's scaling run."""
    import json
    import subprocess
    import sys
    world = 4
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reproducible = tier.endswith("-reproducible")
    tier = tier.split("-")[0]
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", str(world), "--rehearse", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--resolution", "256", "--grid", "64", "--arithmetic", tier] + (["--reproducible"] if reproducible else []),
                         capture_output=True, text=True, timeout=900, cwd=repo)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    line = json.loads([text for text in run.stdout.splitlines() if text.startswith("{")][-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["config"]["arithmetic"] == tier
    rehearsal = line["rehearsal"]
    assert rehearsal["nan_mask_equal"] and rehearsal["sample_num_equal"], rehearsal
    assert line["config"]["bit_reproducible"] == (tier == "exact" or reproducible), line["config"]
    if tier == "exact" or reproducible:
        assert rehearsal["assembled_frame_equals_single_rank_frame_bit_for_bit"], rehearsal
    else:
        assert rehearsal["image_linf_over_max"] < 1.0e-13, rehearsal
