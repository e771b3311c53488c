"""simulation_coord = fmks on the GPU: an iharm3d FMKS dump (tests/golden/reader/iharm3d_fmks.h5, written by
tools/make_goldens.py fmks) through the reader, the locate kernel's FMKS branch (simulation_sampling.cpp:190-198,
:396-456) and the rest of the path, against the reference's own images of that file - library, tolerant tier and the
command-line driver."""
import json
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "blacklight_amd", "bin", "blacklight_amd")
READER_DIR = os.path.join(gu.GOLDEN_DIR, "reader")
FMKS_FILE = os.path.join(READER_DIR, "iharm3d_fmks.h5")


@pytest.fixture(scope="module")
def fmks():
    return np.load(os.path.join(READER_DIR, "expected_fmks.npz"), allow_pickle=False)


def _params(fmks, case, **overrides):
    params = json.loads(str(fmks[f"{case}_params"]))
    params["simulation_file"] = FMKS_FILE
    params.update(overrides)
    return params


@pytest.mark.parametrize("case", ["interp", "nearest"])
def test_fmks_against_the_reference(built_library, fmks, case):
    from blacklight_amd import Context, Params, Snapshot
    p = Params.from_dict(_params(fmks, case))
    rows = ["I_nu"] + (["tau"] if f"{case}_B_tau" in fmks.files else [])
    want = np.stack([fmks[f"{case}_B_{name}"].reshape(-1) for name in rows])
    want_a = np.stack([fmks[f"{case}_A_{name}"].reshape(-1) for name in rows])
    with Context(p) as ctx:
        with Snapshot(p) as snap:
            ctx.set_grid(snap)
        out = ctx.render()
        assert gu.same_bits(out["image"], want).all()
        assert np.max(np.abs(out["image"] - want_a) / np.max(np.abs(want_a), axis=1, keepdims=True)) < 1.0e-6
        ctx.set_arithmetic("tolerant")
        tol = ctx.render()
        assert tol["stats"].arithmetic == 1   # (an optical-depth image beside the intensities stays on the fast path)
        assert np.array_equal(tol["sample_num"], out["sample_num"])
        # (the bound comes from the measured spread, not from a round number: over 300 drawn cameras on this fixture -
        # tools/gpu_fuzz_fmks.py - the tolerant tier's distance reaches 1.0e-11 ... 1.4e-11 where rays graze the steep gradients
        # next to the fixture's polar cut; 5e-11 leaves a factor of four, and is five orders inside north_star's 1e-6)
        assert np.max(np.abs(tol["image"] - want) / np.max(np.abs(want), axis=1, keepdims=True)) < 5.0e-11


def test_fmks_undefined_reads_are_refused(built_library, fmks):
    """Without the goldens' polar cut the camera reaches the last polar zone of the last azimuthal plane, where the
    reference's unbounded arrays hand back another variable's data: refused, never approximated."""
    from blacklight_amd import BlacklightError, Context, Params, Snapshot
    p = Params.from_dict(_params(fmks, "interp", cut_midplane_theta=0.0))
    with Context(p) as ctx:
        with Snapshot(p) as snap:
            ctx.set_grid(snap)
        with pytest.raises(BlacklightError, match="no defined result"):
            ctx.render()
        # the caller's other choice: the edge of the data that exists, and a warning with the count
        ctx.set_undefined_policy("edge")
        out = ctx.render()
        assert np.isfinite(out["image"]).all() and "where the reference reads past its arrays" in ctx.warnings
        # ... which changes nothing where the reference is defined: the goldens' pixels that no cut sample reaches
        ctx.set_undefined_policy("refuse")
    p = Params.from_dict(_params(fmks, "interp"))
    with Context(p) as ctx:
        with Snapshot(p) as snap:
            ctx.set_grid(snap)
        ctx.set_undefined_policy("edge")
        out = ctx.render()
        assert "reads past its arrays" not in ctx.warnings
    assert gu.same_bits(out["image"][0], fmks["interp_B_I_nu"].reshape(-1)).all()


@pytest.mark.parametrize("case", ["interp", "nearest"])
def test_fmks_command_line(built_library, fmks, case, tmp_path):
    params = _params(fmks, case, output_file=str(tmp_path / "image.npz"))
    input_path = tmp_path / "fmks.input"
    with open(input_path, "w") as f:
        for key, value in params.items():
            f.write(f"{key} = {value}\n")
    run = subprocess.run([EXE, str(input_path)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stderr == str(fmks[f"{case}_B_warnings"])
    npz = np.load(params["output_file"])
    prefix = f"{case}_B_"
    names = [k[len(prefix):] for k in fmks.files if k.startswith(prefix) and k != prefix + "warnings"]
    assert sorted(npz.files) == sorted(names)
    for name in names:
        want, got = fmks[prefix + name], npz[name]
        assert got.shape == want.shape and got.dtype == want.dtype, name
        assert gu.same_bits(got, want).all() if want.dtype.kind == "f" else np.array_equal(got, want), name


@pytest.fixture(scope="module")
def fmks_slow():
    return np.load(os.path.join(READER_DIR, "expected_fmks_slow.npz"), allow_pickle=False)


@pytest.mark.parametrize("case", ["interp", "nearest"])
def test_fmks_slow_light_against_the_reference(built_library, fmks_slow, case, tmp_path):
    """slow_light_on over a series of FMKS dumps (tests/golden/reader/fmksslow_*.h5): the time-slice lookup
    (simulation_sampling.cpp:296-349) in front of the locate kernel's FMKS branch (:396-456), values blended from the slices
    (:710-786, :809-912). Through the library's own window of files (bl_slow_light_read) and through the command-line
    driver, against the reference's images of every camera time: bit-exact."""
    from blacklight_amd import Context, Params
    fx = fmks_slow
    params = json.loads(str(fx[f"{case}_params"]))
    params["simulation_file"] = os.path.join(READER_DIR, "fmksslow_{02d}.h5")
    rows = ["I_nu"] + (["tau"] if f"{case}_B_0_tau" in fx.files else [])
    p = Params.from_dict({k: v for k, v in params.items() if k != "output_file"})
    with Context(p) as ctx:
        for image in range(int(params["slow_num_images"])):
            ctx.slow_light_read(image)
            out = ctx.render()
            want = np.stack([fx[f"{case}_B_{image}_{name}"].reshape(-1) for name in rows])
            assert gu.same_bits(out["image"], want).all(), image
            want_a = np.stack([fx[f"{case}_A_{image}_{name}"].reshape(-1) for name in rows])
            assert np.max(np.abs(out["image"] - want_a) / np.max(np.abs(want_a), axis=1, keepdims=True)) < 1.0e-6
            ctx.set_arithmetic("tolerant")   # (slow light has the tolerant tier's coefficient kernel since round 6: within its tolerance)
            again = ctx.render()
            assert again["stats"].arithmetic == 1 and np.array_equal(again["sample_num"], out["sample_num"])
            assert np.array_equal(np.isnan(again["image"]), np.isnan(want))
            with np.errstate(invalid="ignore"):
                assert np.nanmax(np.abs(again["image"] - want) / np.nanmax(np.abs(want), axis=1, keepdims=True)) < 1.0e-10
            ctx.set_arithmetic("exact")
    params["output_file"] = str(tmp_path / "image_{02d}.npz")
    input_path = tmp_path / "fmks_slow.input"
    with open(input_path, "w") as f:
        for key, value in params.items():
            f.write(f"{key} = {value}\n")
    run = subprocess.run([EXE, str(input_path)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stderr == str(fx[f"{case}_B_warnings"])
    for image in range(int(params["slow_num_images"])):
        npz = np.load(tmp_path / f"image_{image + int(params['slow_offset']):02d}.npz")
        prefix = f"{case}_B_{image}_"
        names = [k[len(prefix):] for k in fx.files if k.startswith(prefix)]
        assert sorted(npz.files) == sorted(names)
        for name in names:
            want, got = fx[prefix + name], npz[name]
            assert got.shape == want.shape and got.dtype == want.dtype, name
            assert gu.same_bits(got, want).all() if want.dtype.kind == "f" else np.array_equal(got, want), name
