"""A second, independent restatement of the reference path in vectorised numpy float32 (one
rounding per elementwise op, no FMA), used only to cross-check the C oracle: same written
operation order, different language, compiler and code shape."""
import numpy as np

F = np.float32


def _gemm(a, b):
    """nalgebra static gemm: out[:,c] = a[:,0]*b[0,c]; then += a[:,k]*b[k,c] (k = 1..3).
    a, b: dict[(r,c)] -> array."""
    out = {}
    for c in range(4):
        for r in range(4):
            y = a[(r, 0)] * b[(0, c)]
            for k in range(1, 4):
                y = a[(r, k)] * b[(k, c)] + y
            out[(r, c)] = y
    return out


def model_matrices(pos, rot, scale):
    n = len(scale)
    i, j, k, w = (rot[:, c].astype(F) for c in range(4))
    two = F(2.0)
    with np.errstate(all="ignore"):
        ww, ii, jj, kk = w * w, i * i, j * j, k * k
        ij, wk, wj = i * j * two, w * k * two, w * j * two
        ik, jk, wi = i * k * two, j * k * two, w * i * two
    zero, one = np.zeros(n, F), np.ones(n, F)
    with np.errstate(all="ignore"):
      R = {(0, 0): ww + ii - jj - kk, (0, 1): ij - wk, (0, 2): wj + ik,
         (1, 0): wk + ij, (1, 1): ww - ii + jj - kk, (1, 2): jk - wi,
         (2, 0): ik - wj, (2, 1): wi + jk, (2, 2): ww - ii - jj + kk}
    Rh = {(r, c): (R[(r, c)] if r < 3 and c < 3 else (one if r == c else zero)) for r in range(4) for c in range(4)}
    T = {(r, c): (one if r == c else zero) for r in range(4) for c in range(4)}
    for r in range(3):
        T[(r, 3)] = pos[:, r].astype(F)
    S = {(r, c): zero for r in range(4) for c in range(4)}
    for d in range(3):
        S[(d, d)] = scale.astype(F)
    S[(3, 3)] = one
    with np.errstate(all="ignore"):
        M = _gemm(_gemm(T, Rh), S)
    out = np.empty((n, 16), F)
    for c in range(4):
        for r in range(4):
            out[:, c * 4 + r] = M[(r, c)]
    return out


def world_aabbs(model, mesh_min, mesh_max):
    """model (n,16) column-major; mesh_min/max (n,3) per instance."""
    n = len(model)
    big = F(3.40282347e38)
    lo = np.full((n, 3), big, F)
    hi = np.full((n, 3), -big, F)
    sel = [(0, 0, 0), (1, 0, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (1, 1, 0), (0, 1, 1), (1, 1, 1)]  # (x,y,z) picks
    with np.errstate(all="ignore"):
        for sx, sy, sz in sel:
            x = np.where(sx, mesh_max[:, 0], mesh_min[:, 0]).astype(F)
            y = np.where(sy, mesh_max[:, 1], mesh_min[:, 1]).astype(F)
            z = np.where(sz, mesh_max[:, 2], mesh_min[:, 2]).astype(F)
            t = []
            for r in range(4):
                v = model[:, 0 * 4 + r] * x
                v = model[:, 1 * 4 + r] * y + v
                v = model[:, 2 * 4 + r] * z + v
                v = model[:, 3 * 4 + r] * F(1.0) + v
                t.append(v)
            for a in range(3):
                c = t[a] / t[3]
                lo[:, a] = np.fmin(lo[:, a], c)
                hi[:, a] = np.fmax(hi[:, a], c)
        centre = (hi + lo) / F(2.0)
        half = (hi - lo) / F(2.0)
        return centre - half, centre + half


def coarse_culled(mins, maxs, planes):
    with np.errstate(all="ignore"):
        h = (maxs - mins) * F(0.5)
        c = (mins + maxs) * F(0.5)
        outside = np.zeros(len(mins), bool)
        for p in range(6):
            nx, ny, nz, d = (F(v) for v in planes[p * 4 : p * 4 + 4])
            e = h[:, 0] * np.abs(nx) + h[:, 1] * np.abs(ny) + h[:, 2] * np.abs(nz)
            a0, a1, a2, a3 = nx * c[:, 0], ny * c[:, 1], nz * c[:, 2], d * F(1.0)
            s = (a0 + a2) + (a1 + a3)
            outside |= (s - e) > 0
    return outside


def draw_commands(pos, mesh_id, culled, meshes, cam_pos, first_instance_base=0, first_index_base=0):
    with np.errstate(all="ignore"):
        d = cam_pos.astype(F)[None, :] - pos.astype(F)
        sq = F(0.0) + ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        far = np.sqrt(sq) > F(10.0)
    n_lods = meshes["n_lods"][mesh_id]
    lod = (far & (n_lods > 1)).astype(np.int64)
    length = meshes["index_len"][mesh_id, lod].astype(np.uint64)
    vis = ~culled
    len_vis = np.where(vis, length, 0)
    first_index = (np.cumsum(len_vis) - len_vis + np.uint64(first_index_base)) & np.uint64(0xFFFFFFFF)
    keep = vis & (length > 0)
    idx = np.nonzero(keep)[0]
    return dict(indexCount=length[idx].astype(np.uint32), firstIndex=first_index[idx].astype(np.uint32),
                vertexOffset=meshes["vertex_offset"][mesh_id[idx]], firstInstance=(idx + first_instance_base).astype(np.uint32),
                total=int(len_vis.sum()) & 0xFFFFFFFF)


def light_draw_lists(pos, mesh_id, meshes, lights, first_instance_base=0):
    """shadow_mapping.rs:405-478: every light x every instance, LOD picked against the light."""
    out = []
    for lp in np.asarray(lights, F).reshape(-1, 3):
        with np.errstate(all="ignore"):
            d = lp[None, :] - pos.astype(F)
            sq = F(0.0) + ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
            far = np.sqrt(sq) > F(10.0)
        lod = (far & (meshes["n_lods"][mesh_id] > 1)).astype(np.int64)
        out.append(np.stack([meshes["index_len"][mesh_id, lod].astype(np.uint32), np.ones(len(pos), np.uint32),
                             meshes["index_offset"][mesh_id, lod].astype(np.uint32),
                             meshes["vertex_offset"][mesh_id].astype(np.int32).view(np.uint32),
                             (np.arange(len(pos)) + first_instance_base).astype(np.uint32)], axis=1))
    return np.stack(out)


def run(s, first_instance_base=0, first_index_base=0):
    model = model_matrices(s["pos"], s["rot"], s["scale"])
    mn = s["meshes"]["aabb_min"][s["mesh_id"]]
    mx = s["meshes"]["aabb_max"][s["mesh_id"]]
    mins, maxs = world_aabbs(model, mn, mx)
    culled = coarse_culled(mins, maxs, s["planes"])
    cmds = draw_commands(s["pos"], s["mesh_id"], culled, s["meshes"], s["cam_pos"], first_instance_base, first_index_base)
    return dict(model=model, world_aabb=np.concatenate([mins, maxs], axis=1), coarse_culled=culled, cmds=cmds)
