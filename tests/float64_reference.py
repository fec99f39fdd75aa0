"""A float64 evaluation of the path's FORMULAS (not of its rounding): M = T·R·S, the eight-corner
world AABB, the six-plane test with its margin. Used to show that (a) matrices agree to the north
star's 1e-5 relative tolerance independent of any operation order, and (b) visibility agrees
wherever an instance is not within rounding distance of a plane."""
import numpy as np


def run(s):
    pos = s["pos"].astype(np.float64)
    i, j, k, w = (s["rot"][:, c].astype(np.float64) for c in range(4))
    sc = s["scale"].astype(np.float64)
    n = len(sc)
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = w * w + i * i - j * j - k * k; R[:, 0, 1] = 2 * (i * j - w * k); R[:, 0, 2] = 2 * (w * j + i * k)
    R[:, 1, 0] = 2 * (w * k + i * j); R[:, 1, 1] = w * w - i * i + j * j - k * k; R[:, 1, 2] = 2 * (j * k - w * i)
    R[:, 2, 0] = 2 * (i * k - w * j); R[:, 2, 1] = 2 * (w * i + j * k); R[:, 2, 2] = w * w - i * i - j * j + k * k
    M = np.zeros((n, 4, 4))
    M[:, :3, :3] = R * sc[:, None, None]
    M[:, :3, 3] = pos
    M[:, 3, 3] = 1.0
    mn = s["meshes"]["aabb_min"][s["mesh_id"]].astype(np.float64)
    mx = s["meshes"]["aabb_max"][s["mesh_id"]].astype(np.float64)
    corners = np.stack([np.where(np.array(sel, bool)[None, :], mx, mn)
                        for sel in ((0, 0, 0), (1, 0, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (1, 1, 0), (0, 1, 1), (1, 1, 1))], axis=1)
    world = np.einsum("nrc,nkc->nkr", M[:, :3, :3], corners) + M[:, None, :3, 3]
    lo, hi = world.min(axis=1), world.max(axis=1)
    centre, half = (lo + hi) / 2, (hi - lo) / 2
    planes = s["planes"].astype(np.float64).reshape(6, 4)
    margin = (centre @ planes[:, :3].T + planes[:, 3]) - half @ np.abs(planes[:, :3]).T  # s - e per plane, (n, 6)
    scale = np.abs(centre) @ np.abs(planes[:, :3]).T + np.abs(planes[:, 3]) + half @ np.abs(planes[:, :3]).T
    culled = (margin > 0).any(axis=1)
    # an instance is "decided" when no plane's margin is within rounding distance of zero
    decided = (np.abs(margin) > 1e-4 * scale).all(axis=1)
    model_colmajor = M.transpose(0, 2, 1).reshape(n, 16)
    return dict(model=model_colmajor, mins=lo, maxs=hi, culled=culled, decided=decided)
