"""The .input grammar of the drop-in boundary (blacklight_amd/csrc/bl_params.cpp) behaves like the
reference's InputReader (src/input_reader/input_reader.cpp:72-500, enum_readers.cpp,
adaptive_reader.cpp): whitespace stripping, comments, degrees -> radians, pole detection, triples,
enum spellings, region keys, and the reference's error texts."""
import json
import math

import pytest

import golden_util as gu


@pytest.fixture()
def bl(built_library):
    import blacklight_amd
    return blacklight_amd


def test_whitespace_comments_and_units(bl):
    p = bl.Params.from_text("""
        # a comment line
        camera_th   =   60.0   # degrees
        camera_ph=90
        camera _ rotation = 1 8 0
        cut_midplane_theta = 45.0
        ray_max_steps = 2000 steps
        fallback_rho = 1.5e-3
        simulation_coord = mks
        output_file = out/put.npz   # spaces are stripped even inside values
    """)
    assert p.get("camera_th") == 60.0 * 3.141592653589793 / 180.0
    assert p.get("camera_ph") == 90.0 * 3.141592653589793 / 180.0
    assert p.get("camera_rotation") == 180.0 * 3.141592653589793 / 180.0
    assert p.get("cut_midplane_theta") == 45.0 * 3.141592653589793 / 180.0
    assert p.get("camera_pole") == 0.0
    assert p.get("ray_max_steps") == 2000          # std::stoi reads the leading number
    assert p.get("fallback_rho") == float.fromhex("0x1.89374cp-10")  # std::stof
    assert p.get("simulation_coord") == 1          # "mks" is an alias of sks (enum_readers.cpp:100)
    assert p.get("output_file") == "out/put.npz"
    assert not p.has("camera_r")


@pytest.mark.parametrize("text,pole", [("0.0", 1), ("180.0", 1), ("180", 1), ("1e-30", 0), ("179.9999", 0)])
def test_camera_pole_detection(bl, text, pole):
    p = bl.Params.from_text(f"camera_th = {text}")
    assert p.get("camera_pole") == pole


def test_triples_and_regions(bl):
    p = bl.Params.from_text("""
        cut_plane_origin = 1.0,-2.5,3e1
        cut_plane_normal = 0,0,1
        adaptive_num_regions = 2
        adaptive_region_1_level = 3
        adaptive_region_1_x_min = -0.25
        adaptive_region_2_y_max = 0.5
        adaptive_region_3_level = 9
    """)
    assert [p.get(f"cut_plane_origin_{c}") for c in "xyz"] == [1.0, -2.5, 30.0]
    assert [p.get(f"cut_plane_normal_{c}") for c in "xyz"] == [0.0, 0.0, 1.0]
    assert p.get("adaptive_num_regions") == 2     # region 3 is beyond num_regions: silently ignored


@pytest.mark.parametrize("line,message", [
    ("no_such_key = 1", "Error: Unknown key (no_such_key) in input file.\n"),
    ("camera_pole = true", "Error: Unknown key (camera_pole) in input file.\n"),
    ("camera_r 50", "Error: Invalid assignment in input file.\n"),
    ("ray_flat = yes", "Error: Unknown string used for boolean value.\n"),
    ("model_type = grmhd", "Error: Unknown string used for ModelType value.\n"),
    ("ray_integrator = rk45", "Error: Unknown string used for RayIntegrator value.\n"),
    ("camera_type = fisheye", "Error: Unknown string used for Camera value.\n"),
    ("image_normalization = here", "Error: Unknown string used for FrequencyNormalization value.\n"),
    ("cut_plane_origin = 1.0;2.0;3.0", "Error: Invalid triple (1.0;2.0;3.0) in input file.\n"),
    ("adaptive_region_1_colour = 3", "Error: Unknown key (adaptive_region_1_colour) in input file.\n"),
    ("camera_r = fifty", "Error: Could not read input file.\n"),
])
def test_error_texts(bl, line, message):
    p = bl.Params()
    with pytest.raises(bl.BlacklightError) as err:
        p.set_line(line)
    assert str(err.value) + "\n" == message


def test_read_file_and_run_count(bl, tmp_path):
    path = tmp_path / "case.input"
    path.write_text("model_type = simulation\nsimulation_multiple = true\nslow_light_on = false\n"
                    "simulation_start = 3\nsimulation_end = 7\n")
    p = bl.Params.from_file(path)
    assert p.num_runs == 5
    with pytest.raises(bl.BlacklightError) as err:
        bl.Params.from_file(tmp_path / "missing.input")
    assert str(err.value) == "Error: Could not open input file."


def test_context_validation_messages(bl):
    """bl_init reproduces the constructors' checks (geodesic_integrator.cpp:50-104,
    radiation_integrator.cpp:198-357) before it ever touches a device."""
    fx, params, _ = gu.load_case("sim_dp_interp")

    def failing(**changes):
        q = dict(params)
        for key, value in changes.items():
            if value is None:
                q.pop(key, None)
            else:
                q[key] = value
        with pytest.raises(bl.BlacklightError) as err:
            bl.Context(bl.Params.from_dict(q))
        return str(err.value)

    assert failing(camera_resolution=0) == "Error: Must have positive camera_resolution."
    assert failing(ray_max_steps=-5) == "Error: Must have positive ray_max_steps."
    assert failing(ray_max_retries=0) == "Error: Must have nonnegative ray_max_retries."
    assert failing(image_frequency=-1.0) == "Error: Must choose positive image_frequency."
    assert failing(image_num_frequencies=0) == "Error: Must have positive image_num_frequencies."
    assert failing(checkpoint_geodesic_save="true", checkpoint_geodesic_load="true") == \
        "Error: Cannot both save and load a geodesic checkpoint."
    assert failing(checkpoint_geodesic_save="true", checkpoint_geodesic_file=None) == \
        "Error: GeodesicIntegrator unable to find all needed values in input file."
    assert failing(camera_r=None) == "Error: GeodesicIntegrator unable to find all needed values in input file."
    assert failing(plasma_mu=None) == "Error: RadiationIntegrator unable to find all needed values in input file."
    assert failing(checkpoint_sample_save="true", checkpoint_sample_load="true") == "Error: Cannot both save and load a sample checkpoint."
    # sample checkpoints: saving writes the reference's file; loading is refused - the reference segfaults reading its own
    # file (sample_cut is never restored), so there is nothing to match
    assert "cannot load its own sample checkpoints" in failing(checkpoint_sample_load="true", checkpoint_sample_file="s.dat")
    assert failing(checkpoint_sample_save="true", checkpoint_sample_file=None) == \
        "Error: RadiationIntegrator unable to find all needed values in input file."
    bl.Context(bl.Params.from_dict(dict(params, checkpoint_sample_save="true", checkpoint_sample_file="s.dat")), device=-2).close()
    assert failing(image_light="false") == "Error: No image or rendering selected."
    assert failing(adaptive_max_level=1, adaptive_block_size=5) == \
        "Error: Must have adaptive_block_size divide camera_resolution."
    # reference configurations outside the hot-path scope are refused loudly, never approximated
    # kappa-distribution electrons: the constructor's checks (radiation_integrator.cpp:296-308). An unpolarized run passes them,
    # as in the reference; bl_render refuses it without BL_UNDEFINED_KAPPA (the reference reads an uninitialised constant
    # there: tests/test_gpu_parity.py::test_unpolarized_kappa_electrons_under_the_opt_in_policy)
    assert failing(plasma_kappa_frac=0.2, plasma_w=1.0) == "Error: RadiationIntegrator unable to find all needed values in input file."
    assert failing(plasma_kappa_frac=0.2, plasma_kappa=4.0) == "Error: RadiationIntegrator unable to find all needed values in input file."
    assert failing(plasma_kappa_frac=0.2, plasma_kappa=5.5, plasma_w=1.0, image_polarization="true",
                   image_rotation_split="false") == "Error: Polarized transport only supports kappa in [3.5, 5]."
    assert failing(image_polarization="true", image_rotation_split=None) == "Error: RadiationIntegrator unable to find all needed values in input file."
    # a rendering without its features: bad_optional_access in the reference constructor (radiation_integrator.cpp:161)
    assert failing(render_num_images=1) == "Error: RadiationIntegrator unable to find all needed values in input file."
