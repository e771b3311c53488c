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
