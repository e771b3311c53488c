"""Test-side restatement (numpy, loops) of the reference's adaptive refinement decision and of the next level's block list -
TEST INFRASTRUCTURE, never shipped: RadiationIntegrator::CheckAdaptiveRefinement and ::EvaluateBlock
(radiation_adaptive.cpp:19-139, :163-321) and the block order of Camera::AugmentCamera (camera.cpp:445-458). The library's own
version is host C++ behind bl_adaptive_refine; tests/test_refinement_sweep.py runs both over random images."""
import math

import numpy as np


def evaluate_block(q, p):
    """EvaluateBlock (radiation_adaptive.cpp:163-321) on one block's first image row, q[i, j] = intensity(i, j)."""
    n = q.shape[0]

    def decide(values, cut, frac_limit):
        examined = exceeded = 0
        for v in values:
            if not math.isfinite(v):
                continue
            examined += 1
            if v > cut:
                exceeded += 1
        frac = exceeded / examined if examined > 0 else float("nan")   # 0 / 0 in the reference: NaN, which compares false
        return frac > frac_limit

    if p["adaptive_val_frac"] >= 0.0:   # :169-187
        if decide([abs(q[i, j]) for i in range(n) for j in range(n)], p["adaptive_val_cut"], p["adaptive_val_frac"]):
            return True
    if p["adaptive_abs_grad_frac"] >= 0.0:   # :190-223
        vals = []
        for i in range(n):
            for j in range(n):
                q_x = q[i, j + 1] - q[i, j] if j == 0 else (q[i, j] - q[i, j - 1] if j == n - 1 else 0.5 * (q[i, j + 1] - q[i, j - 1]))
                q_y = q[i + 1, j] - q[i, j] if i == 0 else (q[i, j] - q[i - 1, j] if i == n - 1 else 0.5 * (q[i + 1, j] - q[i - 1, j]))
                vals.append(math.hypot(q_x, q_y))
        if decide(vals, p["adaptive_abs_grad_cut"], p["adaptive_abs_grad_frac"]):
            return True
    if p["adaptive_rel_grad_frac"] >= 0.0:   # :226-263
        vals = []
        with np.errstate(all="ignore"):
            for i in range(n):
                for j in range(n):
                    if j == 0:
                        q_x = 2.0 * (q[i, j + 1] - q[i, j]) / (q[i, j] + q[i, j + 1])
                    elif j == n - 1:
                        q_x = 2.0 * (q[i, j] - q[i, j - 1]) / (q[i, j - 1] + q[i, j])
                    else:
                        q_x = 2.0 * (q[i, j + 1] - q[i, j - 1]) / (q[i, j - 1] + 2.0 * q[i, j] + q[i, j + 1])
                    if i == 0:
                        q_y = 2.0 * (q[i + 1, j] - q[i, j]) / (q[i, j] + q[i + 1, j])
                    elif i == n - 1:
                        q_y = 2.0 * (q[i, j] - q[i - 1, j]) / (q[i - 1, j] + q[i, j])
                    else:
                        q_y = 2.0 * (q[i + 1, j] - q[i - 1, j]) / (q[i - 1, j] + 2.0 * q[i, j] + q[i + 1, j])
                    vals.append(float(np.hypot(q_x, q_y)))
        if decide(vals, p["adaptive_rel_grad_cut"], p["adaptive_rel_grad_frac"]):
            return True
    if p["adaptive_abs_lapl_frac"] >= 0.0:   # :266-289
        vals = []
        for i in range(1, n - 1):
            for j in range(1, n - 1):
                q_x = q[i, j - 1] - 2.0 * q[i, j] + q[i, j + 1]
                q_y = q[i - 1, j] - 2.0 * q[i, j] + q[i + 1, j]
                vals.append(abs(q_x + q_y))
        if decide(vals, p["adaptive_abs_lapl_cut"], p["adaptive_abs_lapl_frac"]):
            return True
    if p["adaptive_rel_lapl_frac"] >= 0.0:   # :292-318
        vals = []
        with np.errstate(all="ignore"):
            for i in range(1, n - 1):
                for j in range(1, n - 1):
                    q_x = 4.0 * (q[i, j - 1] - 2.0 * q[i, j] + q[i, j + 1]) / (q[i, j - 1] + 2.0 * q[i, j] + q[i, j + 1])
                    q_y = 4.0 * (q[i - 1, j] - 2.0 * q[i, j] + q[i + 1, j]) / (q[i - 1, j] + 2.0 * q[i, j] + q[i + 1, j])
                    vals.append(abs(float(q_x + q_y)))
        if decide(vals, p["adaptive_rel_lapl_cut"], p["adaptive_rel_lapl_frac"]):
            return True
    return False


def check_refinement(p, level, image, block_locs, polarized):
    """CheckAdaptiveRefinement (radiation_adaptive.cpp:19-139): (flags, next level's block locations).
    image: (rows, pixels) of the level; block_locs: (blocks, 2) for level > 0 (camera_loc: v, u), None at the root."""
    res, bs = int(p["camera_resolution"]), int(p["adaptive_block_size"])
    root = res // bs
    if level >= int(p["adaptive_max_level"]):   # :21-22: nothing is refined
        n_blocks = root * root if level == 0 else len(block_locs)
        return np.zeros(n_blocks, dtype=bool), np.zeros((0, 2), dtype=np.int32)
    if level == 0:
        locs = np.array([[b // root, b % root] for b in range(root * root)], dtype=np.int32)
    else:
        locs = np.asarray(block_locs, dtype=np.int32).reshape(-1, 2)
    linear = root * 2 ** level
    width = float(p["camera_width"])
    # radiation_integrator.cpp:229-237: the input counts frequencies from 1, and only when there are several
    frequency = int(p["adaptive_frequency_num"]) - 1 if int(p.get("image_num_frequencies", 1)) > 1 else 0
    row = frequency * (4 if polarized else 1)   # radiation_adaptive.cpp:72-75; EvaluateBlock reads the block's first row (:165-166)
    flags = np.zeros(len(locs), dtype=bool)
    for b, (v, u) in enumerate(locs):
        regions = int(p.get("adaptive_num_regions", 0))
        if regions > 0:   # :45-62 / :91-108
            y = ((v + 0.5) / linear - 0.5) * width
            x = ((u + 0.5) / linear - 0.5) * width
            inside = False
            for r in range(1, regions + 1):
                if (level < int(p[f"adaptive_region_{r}_level"]) and p[f"adaptive_region_{r}_x_min"] < x < p[f"adaptive_region_{r}_x_max"]
                        and p[f"adaptive_region_{r}_y_min"] < y < p[f"adaptive_region_{r}_y_max"]):
                    inside = True
                    break
            if inside:
                flags[b] = True
                continue
        if level == 0:   # :68-76: the block's pixels out of the full frame
            block = image[row].reshape(res, res)[v * bs:(v + 1) * bs, u * bs:(u + 1) * bs]
        else:            # :114-119: the level's pixels are stored block by block
            block = image[row, b * bs * bs:(b + 1) * bs * bs].reshape(bs, bs)
        flags[b] = evaluate_block(np.asarray(block, dtype=np.float64), p)
    nxt = [[bv, bu] for (v, u), f in zip(locs, flags) if f for bv in (2 * v, 2 * v + 1) for bu in (2 * u, 2 * u + 1)]   # camera.cpp:445-458
    return flags, np.array(nxt, dtype=np.int32).reshape(-1, 2)
