"""Build libblacklight_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m blacklight_amd.build [--force] [--verbose]

Objects are cached under blacklight_amd/csrc/_obj and rebuilt when a source or header is newer.
-ffp-contract=off is mandatory: bit-exact ray-step counts depend on no implicit FMA contraction
(blmath.h); the only fused operations are the explicit fma calls of the math library.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libblacklight_amd.so")
EXE = os.path.join(HERE, "bin", "blacklight_amd")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

SOURCES = ["bl_locate.hip", "bl_shade.hip", "bl_shade_fast.hip", "bl_shade_fused.hip", "bl_geodesic.hip", "bl_geodesic_quad.hip", "bl_coefficients_freq.hip", "bl_transfer.hip", "bl_polarized.hip", "bl_api.hip",
           "bl_render.hip", "bl_params.cpp", "bl_host.cpp", "bl_snapshot.cpp"]   # (slowest first: they are compiled side by side)
ARCH = "gfx950"
DEVICE_FLAGS = ["-mllvm", "-disable-machine-licm"]
COMMON = ["-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", f"-I{INCLUDE}", f"-I{CSRC}"]


def hipcc():
    path = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(path):
        raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")
    return path


def _newest_header():
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(INCLUDE, "blacklight_amd.h"))
    return max(os.path.getmtime(h) for h in headers)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    header_time = _newest_header()
    objects = []
    stale_sources = []
    for src in SOURCES:
        src_path = os.path.join(CSRC, src)
        obj_path = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objects.append(obj_path)
        if (force or not os.path.exists(obj_path)
                or os.path.getmtime(obj_path) < max(os.path.getmtime(src_path), header_time)):
            stale_sources.append(src)

    def compile_one(src):
        src_path = os.path.join(CSRC, src)
        obj_path = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        cmd = [cc, "-c", src_path, "-o", obj_path] + COMMON + os.environ.get("BLACKLIGHT_AMD_EXTRA_FLAGS", "").split()
        if src.endswith(".hip"):
            # -disable-machine-licm: left on, the back end hoists the dozens of 64-bit literals of the math library (each a
            # register pair) out of every sample loop and then spills some of them; rematerialised where they are used
            # they cost two moves each, no kernel needs scratch memory, and the locate kernel drops from 128 to 51 registers
            # (measured: locate 13.7 -> 12.8 ms, exact coefficient kernel 62.9 -> 60.3 ms per frame; instruction
            # placement only, results are bit-identical)
            cmd += [f"--offload-arch={ARCH}", "-Rpass-analysis=kernel-resource-usage"] + DEVICE_FLAGS
        if verbose:
            print(" ".join(cmd), flush=True)
        result = subprocess.run(cmd, capture_output=True, text=True)
        if result.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{result.stdout}\n{result.stderr}")
        if src.endswith(".hip"):   # registers, scratch and occupancy of every kernel (tests/test_kernel_resources.py)
            with open(obj_path[:-2] + ".resources.txt", "w") as f:
                f.write(result.stderr)
        if verbose and result.stderr:
            print(result.stderr)

    # the translation units side by side (one per stage of the pipeline: the slowest takes ~20 s)
    workers = max(1, min(len(stale_sources), os.cpu_count() or 1, 8))
    if stale_sources:
        with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
            for future in [pool.submit(compile_one, src) for src in stale_sources]:
                future.result()
    rebuilt = bool(stale_sources)
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objects
        result = subprocess.run(cmd, capture_output=True, text=True)
        if result.returncode != 0:
            raise RuntimeError(f"link failed:\n{result.stdout}\n{result.stderr}")
    # command-line driver with the reference's outer contract (blacklight_amd/bin/blacklight_amd)
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    main_src = os.path.join(CSRC, "bl_main.cpp")
    if force or rebuilt or not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(main_src):
        cmd = ["g++", "-std=c++17", "-O2", "-pthread", f"-I{INCLUDE}", main_src, "-o", EXE, f"-L{HERE}", "-lblacklight_amd",
               "-Wl,-rpath,$ORIGIN/.."]
        result = subprocess.run(cmd, capture_output=True, text=True)
        if result.returncode != 0:
            raise RuntimeError(f"g++ failed on bl_main.cpp:\n{result.stdout}\n{result.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
