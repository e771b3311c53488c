"""GPU: INTEGRATION.md section 2 end to end. oracle/_ref/blacklight_bound is the reference's own main() - its InputReader, its
constructors, its OutputWriter - with the four blocks of that section spliced in where it calls its integrators
(tools/check_integration_binding.py builds it in the build container, where /root/reference is; the binary travels like the other
files of oracle/_ref). BASELINE.json's configuration 1 - input/example_formula.input with a 64 x 64 camera - through it: the .npz the
reference's writer produces from the library's image equals the reference's own file (tests/golden/formula_64.npz, the pinned-libm
run) record for record, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUND = os.path.join(REPO, "oracle", "_ref", "blacklight_bound")


def test_reference_main_with_the_binding_renders_configuration_1(tmp_path):
    if not os.path.exists(BOUND):
        pytest.skip("oracle/_ref/blacklight_bound was not built (python tools/check_integration_binding.py in the build container)")
    fx, params, _ = gu.load_case("formula_64")
    params = dict(params, output_file=str(tmp_path / "bound.npz"), num_threads=1)
    # (the reference's constructor reads an uninitialised member in formula mode and may ask for this key: tools/check_integration_binding.py)
    params["image_rotation_split"] = "false"
    input_path = tmp_path / "formula_64.input"
    with open(input_path, "w") as f:
        for key, value in params.items():
            f.write(f"{key} = {str(value).lower() if isinstance(value, bool) else value}\n")
    run = subprocess.run([BOUND, str(input_path)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stderr == str(fx["B_warnings"]), (run.stderr, str(fx["B_warnings"]))   # "Warning: 1 out of 4096 geodesics terminate unexpectedly."
    got = np.load(tmp_path / "bound.npz")
    names = [k[len("B_npz_"):] for k in fx.files if k.startswith("B_npz_")]
    assert sorted(got.files) == sorted(names)
    for name in names:
        want = fx["B_npz_" + name]
        assert got[name].shape == want.shape and got[name].dtype == want.dtype, name
        if want.dtype.kind == "f":
            assert gu.same_bits(got[name], want).all(), name
        else:
            assert np.array_equal(got[name], want), name


def _write_input(path, params):
    with open(path, "w") as f:
        for key, value in params.items():
            f.write(f"{key} = {str(value).lower() if isinstance(value, bool) else value}\n")


def _assert_records_equal(npz, expected, prefix):
    names = [k[len(prefix):] for k in expected.files if k.startswith(prefix)]
    assert names and sorted(npz.files) == sorted(names), (sorted(npz.files), sorted(names))
    for name in names:
        want, got = expected[prefix + name], npz[name]
        assert got.shape == want.shape and got.dtype == want.dtype, name
        assert gu.same_bits(got, want).all() if want.dtype.kind == "f" else np.array_equal(got, want), name


def test_reference_main_with_the_binding_renders_a_series_through_its_own_reader(tmp_path):
    """Simulation mode, where the metric lives: the reference's SimulationReader reads the two .athdf files of the
    reader fixtures, the `binding:grid` block hands its arrays to bl_set_grid, bl_render produces image[0] of each snapshot - the second
    over the sample records the first left in HBM, as the reference integrates its geodesics once - and the reference's OutputWriter
    writes both files: equal to the reference's own (tests/golden/reader/expected.npz) record for record, stderr included."""
    if not os.path.exists(BOUND):
        pytest.skip("oracle/_ref/blacklight_bound was not built (python tools/check_integration_binding.py in the build container)")
    import json
    reader_dir = os.path.join(gu.GOLDEN_DIR, "reader")
    expected = np.load(os.path.join(reader_dir, "expected.npz"), allow_pickle=False)
    params = json.loads(str(expected["series_params"]))
    params.update(simulation_file=os.path.join(reader_dir, "series_{04d}.athdf"), output_file=str(tmp_path / "bound_{02d}.npz"), num_threads=1)
    _write_input(tmp_path / "series.input", params)
    run = subprocess.run([BOUND, str(tmp_path / "series.input")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stderr == str(expected["series_B_warnings"]), (run.stderr, str(expected["series_B_warnings"]))
    for number in (3, 4):
        _assert_records_equal(np.load(tmp_path / f"bound_{number:02d}.npz"), expected, f"series_B_{number}_")
    assert not gu.same_bits(expected["series_B_3_I_nu"], expected["series_B_4_I_nu"]).all()


def test_reference_main_with_the_binding_runs_the_adaptive_loop(tmp_path):
    """The `binding:render` / `binding:adaptive` pair: the reference's while (not adaptive_complete) loop around
    bl_render (d.level = L) and bl_adaptive_refine, its OutputWriter on the arrays the blocks fill (image[L], camera_loc[L],
    camera_pos[L], block_counts, adaptive_num_levels): two refined levels over an .athdf file read by the reference's own reader, every
    record of the .npz - block lists, per-level positions and intensities - equal to the reference's (tools/make_goldens.py binding_adaptive)."""
    if not os.path.exists(BOUND):
        pytest.skip("oracle/_ref/blacklight_bound was not built (python tools/check_integration_binding.py in the build container)")
    import json
    reader_dir = os.path.join(gu.GOLDEN_DIR, "reader")
    expected = np.load(os.path.join(reader_dir, "expected_binding_adaptive.npz"), allow_pickle=False)
    params = json.loads(str(expected["params"]))
    params.update(simulation_file=os.path.join(reader_dir, "series_0003.athdf"), output_file=str(tmp_path / "bound_adaptive.npz"), num_threads=1)
    _write_input(tmp_path / "adaptive.input", params)
    run = subprocess.run([BOUND, str(tmp_path / "adaptive.input")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stderr == str(expected["B_warnings"]), (run.stderr, str(expected["B_warnings"]))
    got = np.load(tmp_path / "bound_adaptive.npz")
    assert int(got["adaptive_num_levels"][0]) == 2 and got["adaptive_num_blocks"].tolist() == [16, 64, 56]
    _assert_records_equal(got, expected, "B_3_")


def test_command_line_driver_on_the_same_adaptive_file(tmp_path):
    """... and the library's own command-line driver (bl_main.cpp: its reader, its loop, its writer) on that input: the same file."""
    import json
    exe = os.path.join(REPO, "blacklight_amd", "bin", "blacklight_amd")
    reader_dir = os.path.join(gu.GOLDEN_DIR, "reader")
    expected = np.load(os.path.join(reader_dir, "expected_binding_adaptive.npz"), allow_pickle=False)
    params = json.loads(str(expected["params"]))
    params.update(simulation_file=os.path.join(reader_dir, "series_0003.athdf"), output_file=str(tmp_path / "cli_adaptive.npz"), num_threads=1)
    _write_input(tmp_path / "adaptive.input", params)
    run = subprocess.run([exe, str(tmp_path / "adaptive.input")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stderr == str(expected["B_warnings"])
    assert "exact arithmetic tier" in run.stdout   # (tests/conftest.py pins the tier; the driver says which one wrote the file)
    _assert_records_equal(np.load(tmp_path / "cli_adaptive.npz"), expected, "B_3_")
