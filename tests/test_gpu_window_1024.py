"""Windows of the 1024^2 benchmark frame computed by the unmodified reference (tools/make_goldens.py
window_1024: a 256^2 base camera with forced adaptive refinement to level 2 evaluates exactly the pixels of
the 1024^2 lattice inside the refined blocks) against the HIP path rendering those pixels of
a plain 1024^2 camera over the same 256^3 grid - the workload bench.py times."""
import json
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(gu.GOLDEN_DIR, "window_1024.npz")


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="window_1024 fixture not generated")
def test_reference_windows_of_the_benchmark_frame(built_library):
    import blacklight_amd as bl
    from blacklight_amd import mock
    fx = np.load(FIXTURE, allow_pickle=False)
    params = json.loads(str(fx["params"]))
    mock_args = json.loads(str(fx["mock_args"]))
    # the build's generator reproduces the reference script's arrays bit for bit (tests/test_mock.py)
    grid = mock.generate(**mock_args)
    p = bl.Params.from_dict(params)
    res, bs = int(p.get("camera_resolution")), 16
    assert res == 1024
    locs = fx["B_block_locs"]
    assert np.array_equal(locs, fx["A_block_locs"])
    iv, iu = np.mgrid[0:bs, 0:bs]
    pixels = np.concatenate([((bv * bs + iv) * res + (bu * bs + iu)).reshape(-1) for bv, bu in locs]).astype(np.int32)
    with bl.Context(p) as ctx:
        ctx.set_grid(grid)
        out = ctx.render(pixel_map=pixels)
    got = out["image"][0]
    want_b = fx["B_I_nu"].reshape(-1)
    assert got.shape == want_b.shape
    same = gu.same_bits(got, want_b)
    assert same.all(), f"{(~same).sum()} of {same.size} window pixels differ from the reference (pinned math)"
    # stock glibc reference: a = 0, stated tolerance
    want_a = fx["A_I_nu"].reshape(-1)
    assert np.nanmax(np.abs(got - want_a)) / np.nanmax(np.abs(want_a)) < 1.0e-6
