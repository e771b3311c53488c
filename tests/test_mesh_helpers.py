"""The test-side mesh builders (tests/golden_util.py: refined_grid, subdivide_blocks) and the sweep tool's generator: pure host code,
checked without a GPU - the GPU tests and tools/gpu_fuzz_*.py rest on them."""
import importlib.util
import os

import numpy as np

import golden_util as gu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_subdividing_blocks_keeps_every_cell_and_every_coordinate():
    from blacklight_amd import mock
    base = mock.generate(n_r=16, n_th=12, n_ph=16)
    mesh = gu.refined_grid(base, block=(4, 3, 4))            # 4 coarse + 32 fine blocks
    small = gu.subdivide_blocks(mesh, (2, 1, 2))
    n_var, n_b, n_k, n_j, n_i = mesh.prim.shape
    assert small.prim.shape == (n_var, 4 * n_b, n_k // 2, n_j, n_i // 2) and small.levels.shape == (4 * n_b,)
    # block b's sub-block (a, 0, c) is entry 4 b + 2 c + a: its cells, its faces, its centres, its place on its level
    for b in (0, 7, n_b - 1):
        for c in range(2):
            for a in range(2):
                n = 4 * b + 2 * c + a
                assert np.array_equal(small.prim[:, n], mesh.prim[:, b, c * 2:(c + 1) * 2, :, a * 2:(a + 1) * 2])
                assert np.array_equal(small.x1f[n], mesh.x1f[b, a * 2:a * 2 + 3]) and np.array_equal(small.x3v[n], mesh.x3v[b, c * 2:c * 2 + 2])
                assert np.array_equal(small.x2f[n], mesh.x2f[b]) and small.levels[n] == mesh.levels[b]
                assert tuple(small.locations[n]) == (2 * mesh.locations[b][0] + a, mesh.locations[b][1], 2 * mesh.locations[b][2] + c)
    # the blocks still tile the domain: every face of the original mesh is a face of a sub-block, total cell volume unchanged
    assert small.prim[0].size == mesh.prim[0].size and small.n_3_root == mesh.n_3_root
    assert set(np.unique(small.x1f)) == set(np.unique(mesh.x1f))


def test_the_sweep_generator_builds_every_layout():
    spec = importlib.util.spec_from_file_location("gpu_fuzz_tiers", os.path.join(REPO, "tools", "gpu_fuzz_tiers.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    seen = {}
    for seed in range(400):
        params, grid, what = tool.draw(seed)
        if what["layout"] in seen:
            continue
        seen[what["layout"]] = grid
        n_var, n_b, n_k, n_j, n_i = grid.prim.shape
        assert grid.x1f.shape == (n_b, n_i + 1) and grid.x2v.shape == (n_b, n_j) and grid.x3f.shape == (n_b, n_k + 1)
        assert np.all(np.diff(grid.x1f, axis=1) > 0) and np.all(np.diff(grid.x2f, axis=1) > 0) and np.all(np.diff(grid.x3f, axis=1) > 0)
        if what["layout"] in (5, 6, 7):
            assert set(np.unique(grid.levels)) == {0, 1} and grid.locations.shape == (n_b, 3) and min(n_i, n_j, n_k) >= 2
        if len(seen) == 8:
            break
    assert sorted(seen) == list(range(8))
