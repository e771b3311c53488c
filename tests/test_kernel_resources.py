"""The register budget the measured numbers rest on: the kernels of the benchmark path
must compile without scratch at their intended occupancy. Read from the compiler's own resource remarks,
which blacklight_amd.build keeps next to the objects."""
import os
import re

import pytest

from blacklight_amd import build as bl_build

RESOURCES = [os.path.join(bl_build.OBJ, name + ".resources.txt")
             for name in ("bl_geodesic", "bl_geodesic_quad", "bl_locate", "bl_shade", "bl_shade_fast", "bl_shade_fused", "bl_transfer")]   # one translation unit per stage of the pipeline

# mangled name -> (waves per SIMD, largest scratch in bytes per lane)
BENCHMARK_KERNELS = {
    "_Z18bl_geodesic_kernelILi0ELb0ELb1ELb0EEv11BlTraceArgs": (2, 0),      # Dormand-Prince, no sample times, zero spin
    "_Z18bl_geodesic_kernelILi0ELb0ELb0ELb0EEv11BlTraceArgs": (1, 0),      # ... any spin: one wave, accumulation registers behind it
    "_Z18bl_geodesic_kernelILi0ELb0ELb1ELb1EEv11BlTraceArgs": (2, 0),      # ... leaving no records of the empty shell around the grid
    "_Z18bl_geodesic_kernelILi0ELb0ELb0ELb1EEv11BlTraceArgs": (1, 0),
    "_Z23bl_geodesic_quad_kernelILb1EEv11BlTraceArgs": (2, 0),             # a ray per quad of lanes (BL_TAIL_QUAD, BL_TAIL_SPLIT)
    "_Z23bl_geodesic_quad_kernelILb0EEv11BlTraceArgs": (2, 0),
    "_Z16bl_locate_kernelILb0ELb0ELb0ELb0EEv11BlShadeArgs": (4, 0),          # merged grid, no slow light (at least 4)
    "_Z16bl_locate_kernelILb1ELb0ELb0ELb0EEv11BlShadeArgs": (4, 0),          # mesh with refinement
    "_Z15bl_shade_kernelILi0ELb0ELb0ELb0ELb0ELb0ELb0EEv11BlShadeArgs": (2, 0),   # simulation, thermal electrons, any coordinates
    "_Z20bl_shade_fast_kernelILb1ELi0EEv11BlShadeArgs": (2, 0),              # tolerant tier, zero spin
    "_Z20bl_shade_fast_kernelILb0ELi0EEv11BlShadeArgs": (2, 0),
    "_Z20bl_shade_fast_kernelILb0ELi1EEv11BlShadeArgs": (2, 0),              # ... power laws / Cartesian grids
    "_Z20bl_shade_fast_kernelILb0ELi2EEv11BlShadeArgs": (2, 0),              # ... behind inter-block interpolation (anchor cells)
    "_Z20bl_shade_fast_kernelILb0ELi3EEv11BlShadeArgs": (2, 0),              # ... behind slow light (time slices)
    "_Z22bl_shade_fused2_kernelILb1ELb1ELb0ELb0EEv11BlShadeArgs": (2, 0),     # ... the benchmark's kernel: locate step inside, composed maps
    "_Z22bl_shade_fused2_kernelILb1ELb0ELb0ELb0EEv11BlShadeArgs": (2, 0),        # ... one record per sample
    "_Z22bl_shade_fused2_kernelILb0ELb1ELb0ELb0EEv11BlShadeArgs": (2, 0),        # ... any spin
    "_Z22bl_shade_fused2_kernelILb0ELb0ELb0ELb0EEv11BlShadeArgs": (2, 0),
    "_Z22bl_shade_fused2_kernelILb1ELb0ELb1ELb0EEv11BlShadeArgs": (2, 0),        # ... several frequencies: a sample leaves as its factors
    "_Z22bl_shade_fused2_kernelILb0ELb0ELb1ELb0EEv11BlShadeArgs": (2, 0),
    "_Z22bl_shade_fused2_kernelILb1ELb1ELb0ELb1EEv11BlShadeArgs": (2, 0),     # ... over a mesh with refinement (box descriptors and row chunks in LDS)
    "_Z22bl_shade_fused2_kernelILb0ELb1ELb0ELb1EEv11BlShadeArgs": (2, 0),
    "_Z22bl_shade_exact2_kernelILb1EEv11BlShadeArgs": (2, 0),                # exact tier, locate step inside (the benchmark's exact kernel)
    "_Z22bl_shade_exact2_kernelILb0EEv11BlShadeArgs": (2, 0),
    "_Z26bl_shade_polarized2_kernelILb1ELb0ELb0EEv11BlShadeArgs": (2, 0),    # polarized runs, locate step inside, no auxiliary records
    "_Z26bl_shade_polarized2_kernelILb0ELb0ELb0EEv11BlShadeArgs": (2, 0),
    "_Z26bl_shade_polarized2_kernelILb1ELb1ELb0EEv11BlShadeArgs": (2, 0),    # ... with auxiliary records
    "_Z26bl_shade_polarized2_kernelILb0ELb1ELb0EEv11BlShadeArgs": (2, 0),
    "_Z26bl_shade_polarized2_kernelILb1ELb0ELb1EEv11BlShadeArgs": (2, 0),    # ... one frequency, thermal electrons: the polarized coefficients evaluated inside
    "_Z21bl_shade_exact_kernelILb1EEv11BlShadeArgs": (2, 0),                 # exact tier behind a locate kernel, software-pipelined
    "_Z21bl_shade_exact_kernelILb0EEv11BlShadeArgs": (2, 0),
    "_Z22bl_locate_plain_kernelILb1EEv11BlShadeArgs": (4, 0),                # exact tier's locate step, common grid case
    "_Z22bl_locate_plain_kernelILb0EEv11BlShadeArgs": (4, 0),
    "_Z28bl_shade_formula_fast_kernel11BlShadeArgs": (None, 0),              # tolerant tier, formula mode
    "_Z15bl_shade_kernelILi0ELb0ELb0ELb1ELb0ELb1ELb1EEv11BlShadeArgs": (2, 0),   # ... its exact second pass
    "_Z18bl_transfer_kernelILb0EEv14BlTransferArgs": (None, 0),
    "_Z18bl_transfer_kernelILb1EEv14BlTransferArgs": (None, 0),
    "_Z23bl_transfer_quad_kernel14BlTransferArgs": (None, 0),               # tolerant tier, one frequency: four lanes per ray
    "_Z27bl_transfer_composed_kernel14BlTransferArgs": (None, 0),           # ... composed maps: one lane per ray
}


def _parse():
    text = "".join(open(path).read() for path in RESOURCES)
    kernels = {}
    current = None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            current = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\])?: (\d+)", line)
        if m and current is not None:
            current[m.group(1).strip()] = int(m.group(2))
    return kernels


def test_benchmark_kernels_fit_their_registers(built_library):
    if not all(os.path.exists(path) for path in RESOURCES):
        bl_build.build(force=True)
    kernels = _parse()
    for name, (occupancy, scratch) in BENCHMARK_KERNELS.items():
        assert name in kernels, (name, sorted(kernels))
        usage = kernels[name]
        assert usage["ScratchSize"] <= scratch, (name, usage)
        if occupancy is not None:
            assert usage["Occupancy"] >= occupancy, (name, usage)


# bl_debug_math_kernel: the diagnostics kernel behind bl_debug_math (a switch over three dozen math functions, one of them with a
# branch into bl_pow): on no render path, timed by nobody
SCRATCH_ALLOWED = {"_Z20bl_debug_math_kernelixPKdS0_Pd": 64}


def test_no_kernel_needs_scratch_memory(built_library):
    """Every kernel of every device translation unit - all instantiations, polarized and slow-light ones included - keeps its
    state in registers (and LDS): no private-segment memory, so no hidden memory traffic behind the measured numbers."""
    everything = RESOURCES + [os.path.join(bl_build.OBJ, name + ".resources.txt") for name in ("bl_coefficients_freq", "bl_polarized")]
    if not all(os.path.exists(path) for path in everything):
        bl_build.build(force=True)
    for path in everything:
        text = open(path).read()
        names = re.findall(r"remark: Function Name: (\S+)", text)
        scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
        assert len(names) == len(scratch) and len(names) >= 2
        over = {n: s for n, s in zip(names, scratch) if s > SCRATCH_ALLOWED.get(n, 0)}
        assert not over, over


def test_the_surface_stays_what_the_design_says(built_library):
    """DESIGN.md section 0 counts 75 kernel instantiations (98 at the end of round 5): a new one is a decision, not an accident."""
    import glob
    reports = sorted(glob.glob(os.path.join(bl_build.OBJ, "*.resources.txt")))
    if not reports:
        bl_build.build(force=True)
        reports = sorted(glob.glob(os.path.join(bl_build.OBJ, "*.resources.txt")))
    names = [n for path in reports for n in re.findall(r"remark: Function Name: (\S+)", open(path).read())]
    assert len(names) == len(set(names)) and len(names) <= 75, len(names)
