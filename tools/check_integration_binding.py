#!/usr/bin/env python3
"""Compile proof of INTEGRATION.md section 2: the five fenced blocks marked `<!-- binding:NAME -->` there are spliced into a copy
of the reference's main() (/root/reference/src/blacklight.cpp, read where it lies; the patched copy lives in a temporary directory
and is never committed), compiled with the reference's other translation units (the objects oracle/Makefile builds into oracle/_ref/obj) and
linked against blacklight_amd/libblacklight_amd.so -> oracle/_ref/blacklight_bound (git-ignored, travels to the GPU box like the
other binaries there). Then the program is run:

  * here (no GPU): BLACKLIGHT_AMD_BINDING_DEVICE=-2 - a host-only context: the reference's InputReader, constructors and the
    library's parameter validation / camera frame / frequency list run, bl_render refuses with its BL_E_DEVICE text;
  * on a GPU box (tests/test_gpu_binding.py): input/example_formula.input at 64 x 64, a two-file .athdf series through the reference's
    SimulationReader, and that series with adaptive refinement, end to end - reference InputReader, SimulationReader and OutputWriter
    around bl_init / bl_set_grid / bl_render / bl_adaptive_refine - every .npz against the reference's own.

    python tools/check_integration_binding.py [--no-run] [--no-build]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "oracle", "_ref", "blacklight_bound")
REFFLAGS = ["-std=c++17", "-fopenmp", "-O3", "-flto", "-fno-math-errno", "-fno-signed-zeros", "-fno-trapping-math"]   # oracle/Makefile


def snippets():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    found = dict(re.findall(r"<!-- binding:(\w+) -->\n```cpp\n(.*?)```", text, flags=re.S))
    missing = {"includes", "init", "grid", "render", "adaptive"} - set(found)
    if missing:
        raise SystemExit(f"INTEGRATION.md: binding blocks missing: {sorted(missing)}")
    return found


def splice(source, blocks):
    def replace_once(text, old, new):
        if text.count(old) != 1:
            raise SystemExit(f"reference main(): expected exactly one occurrence of {old!r}")
        return text.replace(old, new)

    source = replace_once(source, '#include "utils/exceptions.hpp"', '#include "utils/exceptions.hpp"\n' + blocks["includes"].rstrip() + "\n#define BL_EXCEPTIONS_INCLUDED")
    # the first occurrence carries a trailing comment; restore it harmlessly
    source = source.replace("#define BL_EXCEPTIONS_INCLUDED", "", 1)
    source = replace_once(source, "    p_geodesic_integrator = new GeodesicIntegrator(p_input_reader);\n    time_geodesic += p_geodesic_integrator->Integrate();\n",
                          blocks["init"])
    source = replace_once(source, "      time_read += p_simulation_reader->Read(n);\n", "      time_read += p_simulation_reader->Read(n);\n" + blocks["grid"])
    source = replace_once(source, "        adaptive_complete =\n            p_radiation_integrator->Integrate(n, &time_sample, &time_image, &time_render);\n", blocks["render"])
    source = replace_once(source, "          time_geodesic += p_geodesic_integrator->AddGeodesics(p_radiation_integrator);\n", blocks["adaptive"])
    source = replace_once(source, "  delete p_input_reader;\n", "  delete p_input_reader;\n  bl_free(bl_context);\n")
    return source


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        raise SystemExit(f"{REF}/src is not here: the binding is compiled in the build container only")
    if "--no-build" not in sys.argv:
        sys.path.insert(0, REPO)
        import __graft_entry__
        __graft_entry__.build()   # the library, and oracle/_ref/obj with the reference's objects
    WORK = tempfile.mkdtemp(prefix="bl_binding_")   # (a directory of this run's own: builds side by side do not meet)
    main_cpp = open(os.path.join(REF, "src", "blacklight.cpp")).read()
    main_cpp = main_cpp.replace('#include "utils/exceptions.hpp"                           // BlacklightException', '#include "utils/exceptions.hpp"')
    patched = os.path.join(WORK, "blacklight_bound.cpp")
    with open(patched, "w") as f:
        f.write(splice(main_cpp, snippets()))
    objects = []
    for root, _, files in os.walk(os.path.join(REPO, "oracle", "_ref", "obj")):
        objects += [os.path.join(root, name) for name in files if name.endswith(".o") and not (name == "blacklight.o" and root.endswith("obj"))]
    lib_dir = os.path.join(REPO, "blacklight_amd")
    cmd = ["g++"] + REFFLAGS + [f"-I{REF}/src", f"-I{REPO}/include", patched] + sorted(objects) + [f"-L{lib_dir}", "-lblacklight_amd", "-Wl,-rpath,$ORIGIN/../../blacklight_amd",
                                                                                                  "-o", OUT]
    subprocess.run(cmd, check=True)
    print("built", OUT)
    if "--no-run" in sys.argv:
        return
    # host-only run of the reference's example (64 x 64): everything up to the render
    work_input = os.path.join(WORK, "example_formula_64.input")
    text = open(os.path.join(REF, "input", "example_formula.input")).read()
    text = re.sub(r"camera_resolution\s*=\s*\d+", "camera_resolution = 64", text)
    text = re.sub(r"output_file\s*=\s*\S+", f"output_file = {WORK}/example_formula_64.npz", text)
    # (In formula mode the reference never sets RadiationIntegrator::image_polarization, then tests it - radiation_integrator.cpp:
    # 99-105 - and asks for image_rotation_split when the uninitialised byte reads true: whether it does depends on what the heap held,
    # and without the geodesic integrator's allocations in front of it, it does. Giving the key keeps the reference's constructor quiet.)
    text += "\nimage_rotation_split = false\n"
    with open(work_input, "w") as f:
        f.write(text)
    run = subprocess.run([OUT, work_input], env=dict(os.environ, BLACKLIGHT_AMD_BINDING_DEVICE="-2"), capture_output=True, text=True)
    print("host-only run: exit", run.returncode, "| stdout:", run.stdout.strip(), "| stderr:", run.stderr.strip())
    if run.returncode != 1 or "Host-only context" not in run.stdout:
        raise SystemExit("the bound program did not reach bl_render on the host-only context")
    print("binding compiles, links and runs up to bl_render (which needs a GPU)")


if __name__ == "__main__":
    main()
