#!/usr/bin/env python3
"""Static instruction counts per kernel from the gfx950 assembly of the kernel files of blacklight_amd/csrc
(tools only; a proxy for the dynamic counts rocprofv3 reports).  python tools/isa_count.py [filter ...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/bl_isa"
os.makedirs(OUT, exist_ok=True)
asm = os.path.join(OUT, "kernels.s")
sources = [os.path.join(REPO, "blacklight_amd", "csrc", f + ".hip") for f in ("bl_geodesic", "bl_shade", "bl_shade_fast", "bl_transfer")]
if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(os.path.join(REPO, "blacklight_amd", "csrc", f))
                                                          for f in os.listdir(os.path.join(REPO, "blacklight_amd", "csrc")) if f.endswith((".hip", ".h"))):
    with open(asm, "w") as out:
        for src in sources:
            part = os.path.join(OUT, os.path.basename(src)[:-4] + ".s")
            subprocess.run(["hipcc", "-S", "--offload-device-only", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-mllvm",
                            "-disable-machine-licm", f"-I{REPO}/include", f"-I{REPO}/blacklight_amd/csrc", src, "-o", part], check=True, capture_output=True)
            out.write(open(part).read())
lines = open(asm).read().split("\n")
filters = sys.argv[1:] or ["shade", "locate_kernelILb0ELb0", "geodesic_kernelILi0ELb0", "transfer_kernel"]
name = None
counts = {}
for line in lines:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        counts[name] = dict(total=0, valu=0, f64=0, fma=0, vmem=0, salu=0, lds=0, trans=0)
        continue
    if line.startswith(".Lfunc_end"):
        name = None
        continue
    if name is None or not line.startswith("\t"):
        continue
    op = line.strip().split()[0] if line.strip() else ""
    if not op or op.startswith((".", ";")):
        continue
    c = counts[name]
    c["total"] += 1
    if op.startswith("v_"):
        c["valu"] += 1
        if "f64" in op:
            c["f64"] += 1
        if op.startswith("v_fma_f64"):
            c["fma"] += 1
        if op.startswith(("v_rcp", "v_rsq", "v_exp", "v_log", "v_sqrt")):
            c["trans"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c["vmem"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
for name, c in counts.items():
    if any(f in name for f in filters):
        print(f"{name[:80]:80s} " + " ".join(f"{k}={v}" for k, v in c.items()))
