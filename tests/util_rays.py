"""Deterministic probe rays for the per-ray parity tests (PCG32, seed 0x5EED -- SURVEY.md 8d)."""
import numpy as np


def _pcg32(n, seed=0x5EED):
    state = np.uint64(seed)
    inc = np.uint64(1442695040888963407)
    mul = np.uint64(6364136223846793005)
    out = np.empty(n, np.uint32)
    with np.errstate(over="ignore"):
        for i in range(n):
            old = state
            state = old * mul + inc
            xs = np.uint32(((old >> np.uint64(18)) ^ old) >> np.uint64(27))
            rot = np.uint32(old >> np.uint64(59))
            out[i] = (xs >> rot) | (xs << ((np.uint32(32) - rot) & np.uint32(31)))
    return out


def probe_rays(n):
    """n x 6 float32 rays: 3/4 aimed from around the camera into the scene volume, 1/4 grazing / axis-aligned
    (zero direction components exercise the +-inf slab-test paths, objects.cpp:543-567)."""
    u = (_pcg32(n * 6).astype(np.float64) / 2**32).reshape(n, 6)
    rays = np.zeros((n, 6), np.float32)
    rays[:, 0:3] = ((u[:, 0:3] - 0.5) * np.array([1.0, 1.0, 1.0])).astype(np.float32)
    tgt = (u[:, 3:6] - 0.5) * np.array([6.0, 4.0, 4.0]) + np.array([0.0, 0.0, -4.0])
    d = tgt - rays[:, 0:3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3:6] = d.astype(np.float32)
    k = n // 4
    # grazing: nearly tangent directions and exact zeros in the direction
    rays[:k:4, 3] = 0.0
    rays[1:k:4, 4] = 0.0
    rays[2:k:4, 3:6] = np.array([0.0, 0.0, -1.0], np.float32)
    rays[3:k:4, 4] = np.float32(1e-7)
    return rays
