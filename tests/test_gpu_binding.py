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
