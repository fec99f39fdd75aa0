"""TEST-ONLY numpy restatement of meshoptimizer's meshopt_simplifySloppy (v0.13, what the reference's loader calls
through the `meshopt 0.1.9` crate, scene_loader.rs:739-753), written independently of renderer_amd/host/
simplify_sloppy.cpp so that the two can be held against each other: vectorised float32 arithmetic in the library's
operation order, cells by first appearance, representatives by least quadric error (first on ties), duplicate
triangles dropped. Neither restatement could be diffed against the library itself (not in this image): unpinned."""
import numpy as np

F = np.float32


def _to_int(v):
    """float -> int as the compiled library does it (cvttss2si): truncation; out of range / NaN -> INT_MIN."""
    v = np.atleast_1d(np.asarray(v, np.float32))
    bad = ~((v > F(-2147483904.0)) & (v < F(2147483648.0)))
    out = np.trunc(np.where(bad, 0, v)).astype(np.int64)
    out[bad] = -(2 ** 31)
    return out


def _vertex_ids(p, grid):
    cs = F(grid - 1)
    xi = _to_int(p[:, 0] * cs + F(0.5)) & 0xFFFFFFFF
    yi = _to_int(p[:, 1] * cs + F(0.5)) & 0xFFFFFFFF
    zi = _to_int(p[:, 2] * cs + F(0.5)) & 0xFFFFFFFF
    return (((xi << 20) & 0xFFFFFFFF) | ((yi << 10) & 0xFFFFFFFF) | zi).astype(np.uint32)


def _count(ids, tri):
    a, b, c = ids[tri[:, 0]], ids[tri[:, 1]], ids[tri[:, 2]]
    return int(((a != b) & (a != c) & (b != c)).sum())


def _interpolate(y, x0, y0, x1, y1, x2, y2):
    y, x0, y0, x1, y1, x2, y2 = map(F, (y, x0, y0, x1, y1, x2, y2))
    with np.errstate(all="ignore"):
        num = (y1 - y) * (x1 - x2) * (x1 - x0) * (y2 - y0)
        den = (y2 - y) * (x1 - x2) * (y0 - y1) + (y0 - y) * (x1 - x0) * (y1 - y2)
        return F(x1 + num / den)


def simplify_sloppy(indices, positions, target_index_count):
    indices = np.asarray(indices, np.int64)
    index_count = len(indices) - len(indices) % 3
    tri = indices[:index_count].reshape(-1, 3)
    pos = np.asarray(positions, np.float32).reshape(-1, 3)
    target_index_count = min(int(target_index_count), index_count)
    target_cells = target_index_count // 6
    if target_cells == 0:
        return np.zeros(0, np.uint32)
    mn, mx = pos.min(0), pos.max(0)
    extent = F(max(F(0), *(mx - mn)))
    scale = F(0) if extent == 0 else F(1) / extent
    p = ((pos - mn) * scale).astype(np.float32)

    min_grid, max_grid, min_tris, max_tris = 0, 1025, 0, index_count // 3
    nxt = int(_to_int(np.sqrt(F(target_cells)) + F(0.5))[0])
    for ps in range(15):
        g = min_grid + 1 if nxt <= min_grid else (max_grid - 1 if nxt >= max_grid else nxt)
        t = _count(_vertex_ids(p, g), tri)
        tip = _interpolate(target_index_count // 3, min_grid, min_tris, g, t, max_grid, max_tris)
        if t <= target_index_count // 3:
            min_grid, min_tris = g, t
        else:
            max_grid, max_tris = g, t
        if t == target_index_count // 3 or max_grid - min_grid <= 1:
            break
        nxt = int(_to_int(tip + F(0.5))[0]) if ps < 5 else (min_grid + max_grid) // 2
    if min_tris == 0:
        return np.zeros(0, np.uint32)

    ids = _vertex_ids(p, min_grid)
    _, first, inverse = np.unique(ids, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))  # cells numbered by first appearance
    cells = order[inverse]
    n_cells = len(first)

    # quadrics, accumulated triangle by triangle in float32 (np.add.at keeps the order of the source's loop per cell)
    p0, p1, p2 = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    c0, c1, c2 = cells[tri[:, 0]], cells[tri[:, 1]], cells[tri[:, 2]]
    single = (c0 == c1) & (c0 == c2)
    e1, e2 = p1 - p0, p2 - p0
    nrm = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                    e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], 1).astype(np.float32)
    length = np.sqrt((nrm[:, 0] * nrm[:, 0] + nrm[:, 1] * nrm[:, 1]).astype(np.float32) + nrm[:, 2] * nrm[:, 2]).astype(np.float32)
    with np.errstate(all="ignore"):
        nrm = np.where(length[:, None] > 0, nrm / length[:, None], nrm).astype(np.float32)
    dist = ((nrm[:, 0] * p0[:, 0] + nrm[:, 1] * p0[:, 1]).astype(np.float32) + nrm[:, 2] * p0[:, 2]).astype(np.float32)
    w = (np.sqrt(length).astype(np.float32) * np.where(single, F(3), F(1))).astype(np.float32)
    a, b, c, d = nrm[:, 0], nrm[:, 1], nrm[:, 2], -dist
    aw, bw, cw, dw = a * w, b * w, c * w, d * w
    q = np.stack([a * aw, b * bw, c * cw, a * bw, a * cw, b * cw, a * dw, b * dw, c * dw, d * dw, w], 1).astype(np.float32)
    Q = np.zeros((n_cells, 11), np.float32)
    # sequential accumulation in the source's order: triangle i adds to c0, then c1, then c2 (once if single)
    tri_rep = np.stack([c0, np.where(single, -1, c1), np.where(single, -1, c2)], 1).reshape(-1)
    q_rep = np.repeat(q, 3, axis=0)
    keep = tri_rep >= 0
    for cell, row in zip(tri_rep[keep], q_rep[keep]):  # float32 adds, one at a time
        Q[cell] += row

    qa = Q[cells]
    rx, ry, rz = qa[:, 6].copy(), qa[:, 7].copy(), qa[:, 8].copy()
    rx = rx + qa[:, 3] * p[:, 1]; ry = ry + qa[:, 5] * p[:, 2]; rz = rz + qa[:, 4] * p[:, 0]
    rx = rx * F(2); ry = ry * F(2); rz = rz * F(2)
    rx = rx + qa[:, 0] * p[:, 0]; ry = ry + qa[:, 1] * p[:, 1]; rz = rz + qa[:, 2] * p[:, 2]
    r = qa[:, 9] + rx * p[:, 0]
    r = r + ry * p[:, 1]
    r = r + rz * p[:, 2]
    with np.errstate(all="ignore"):
        s = np.where(qa[:, 10] == 0, F(0), F(1) / qa[:, 10]).astype(np.float32)
    err = (np.abs(r) * s).astype(np.float32)
    remap = np.full(n_cells, -1, np.int64)
    best = np.zeros(n_cells, np.float32)
    for i in range(len(p)):  # first vertex of least error (strict `>` replaces)
        cell = cells[i]
        if remap[cell] < 0 or best[cell] > err[i]:
            remap[cell], best[cell] = i, err[i]

    live = (c0 != c1) & (c0 != c2) & (c1 != c2)
    A, B, C = remap[c0[live]], remap[c1[live]], remap[c2[live]]
    out, seen = [], set()
    for a_, b_, c_ in zip(A.tolist(), B.tolist(), C.tolist()):
        if b_ < a_ and b_ < c_:
            a_, b_, c_ = b_, c_, a_
        elif c_ < a_ and c_ < b_:
            a_, b_, c_ = c_, a_, b_
        if (a_, b_, c_) not in seen:
            seen.add((a_, b_, c_))
            out += [a_, b_, c_]
    return np.asarray(out, np.uint32)
