"""Synthetic N-instance scenes: portable splitmix64 PRNG so the same
instances can be regenerated anywhere, reference-default camera, per-config mesh tables."""
import numpy as np

from .pipeline import MESH_DTYPE

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
SEED_BASE = 0x5EED0000

# reference defaults: camera at (0,1,2), identity rotation (src/ecs/camera_controller.rs:21),
# fov 70 deg, near 0.1, far 100 (src/ecs.rs:69-72), window 2000x1000 (src/renderer/instance.rs:45)
DEFAULT_CAMERA = dict(cam_pos=(0.0, 1.0, 2.0), cam_rot_ijkw=(0.0, 0.0, 0.0, 1.0), aspect=2.0,
                      fovy_degrees=70.0, near=0.1, far=100.0)

CONFIGS = {
    1: dict(name="box_1k", n=1024, workload="glTF Box, 1 024 static instances"),
    2: dict(name="damaged_helmet_100k", n=100_000, workload="DamagedHelmet, 100 k static instances"),
    3: dict(name="mixed_1m", n=1_000_000, workload="mixed 64-mesh scene, 1 M instances"),
    4: dict(name="mixed_10m", n=10_000_000, workload="mixed 64-mesh scene, 10 M instances (8 shards)"),
    5: dict(name="rigged_figure_256k", n=256_000, workload="RiggedFigure-like skinned figure (19 joints), 256 k animated instances"),
}


def splitmix64(seed, start, count):
    """Outputs start .. start+count-1 (0-based) of the splitmix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def u01(z):
    return (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def default_planes():
    """Frustum planes of the reference's default camera, as float32[24]. Computed in float64
    from the formulas of src/ecs.rs:66-91 and rounded once (the planes are an input of the
    path; SURVEY.md §8c KAT 2)."""
    c = DEFAULT_CAMERA
    t = np.tan(np.radians(c["fovy_degrees"]) / 2.0)
    n, f = c["near"], c["far"]
    proj = np.zeros((4, 4))
    proj[0, 0] = 1.0 / (c["aspect"] * t)
    proj[1, 1] = 1.0 / t
    proj[2, 2] = f / (f - n)
    proj[2, 3] = -(f * n) / (f - n)
    proj[3, 2] = 1.0
    eye = np.array(c["cam_pos"], dtype=np.float64)
    view = np.eye(4)
    view[:3, 3] = -eye  # identity rotation: look down +z, up +y (left-handed)
    m = proj @ view
    rows = []
    for k in range(3):
        rows.append(-(m[3] + m[k]))
        rows.append(-(m[3] - m[k]))
    return np.asarray(rows, dtype=np.float64).astype(np.float32).reshape(24)


def box_mesh_table():
    t = np.zeros(1, MESH_DTYPE)
    t["aabb_min"] = (-0.5, -0.5, -0.5)
    t["aabb_max"] = (0.5, 0.5, 0.5)
    t["n_lods"] = 1
    t["index_len"][0, 0] = 36
    t["vertex_offset"] = 0
    return t


def _lod_chain(len0):
    """LOD k targets len0 * 0.5^k indices (scene_loader.rs:741-751), kept only while it shrinks."""
    lens = [int(len0)]
    for k in range(1, 6):
        target = int(np.float32(len0) * np.float32(0.5) ** np.float32(k))
        target -= target % 3
        if 0 < target < lens[-1]:
            lens.append(target)
    return lens


def damaged_helmet_mesh_table():
    t = np.zeros(1, MESH_DTYPE)
    t["aabb_min"] = (-0.9475, -1.1872, -0.9010)
    t["aabb_max"] = (0.9425, 0.8129, 0.9010)
    lens = _lod_chain(46356)
    t["n_lods"] = len(lens)
    off = 0
    for k, l in enumerate(lens):
        t["index_len"][0, k] = l
        t["index_offset"][0, k] = off
        off += l
    t["vertex_offset"] = 0
    return t


def mixed_mesh_table(m=64, seed=SEED_BASE + 0x100):
    z = u01(splitmix64(seed, 0, m * 9)).reshape(m, 9)
    t = np.zeros(m, MESH_DTYPE)
    centre = z[:, 0:3] - 0.5
    half = 0.1 + 1.4 * z[:, 3:6]
    t["aabb_min"] = (centre - half).astype(np.float32)
    t["aabb_max"] = (centre + half).astype(np.float32)
    len0 = np.exp(np.log(36.0) + z[:, 6] * (np.log(70074.0) - np.log(36.0)))
    len0 = (len0.astype(np.int64) // 3) * 3
    max_lods = 1 + (z[:, 7] * 6).astype(np.int64).clip(0, 5)
    index_off = 0
    vertex_off = 0
    for k in range(m):
        lens = _lod_chain(int(len0[k]))[: int(max_lods[k])]
        t["n_lods"][k] = len(lens)
        for j, l in enumerate(lens):
            t["index_len"][k, j] = l
            t["index_offset"][k, j] = index_off & 0xFFFFFFFF
            index_off += l
        t["vertex_offset"][k] = vertex_off
        vertex_off += max(int(len0[k]) // 3, 3)
    return t


def make_instances(n, n_meshes, seed, box=((-64.0, 64.0), (-32.0, 32.0), (-64.0, 64.0)), first=0):
    """Instances first .. first+n-1 of the stream: 8 draws per instance
    (pos x,y,z; quaternion u1,u2,u3; scale; mesh)."""
    pos = np.empty((n, 3), np.float32)
    rot = np.empty((n, 4), np.float32)
    scale = np.empty(n, np.float32)
    mesh = np.empty(n, np.uint32)
    step = 1 << 20
    for b in range(0, n, step):
        e = min(n, b + step)
        u = u01(splitmix64(seed, (first + b) * 8, (e - b) * 8)).reshape(e - b, 8)
        for a in range(3):
            lo, hi = box[a]
            pos[b:e, a] = (lo + u[:, a] * (hi - lo)).astype(np.float32)
        # Shoemake: uniform unit quaternion, stored [i, j, k, w]
        s1, s2 = np.sqrt(1.0 - u[:, 3]), np.sqrt(u[:, 3])
        a1, a2 = 2.0 * np.pi * u[:, 4], 2.0 * np.pi * u[:, 5]
        rot[b:e, 0] = (s1 * np.sin(a1)).astype(np.float32)
        rot[b:e, 1] = (s1 * np.cos(a1)).astype(np.float32)
        rot[b:e, 2] = (s2 * np.sin(a2)).astype(np.float32)
        rot[b:e, 3] = (s2 * np.cos(a2)).astype(np.float32)
        scale[b:e] = (0.5 + 1.5 * u[:, 6]).astype(np.float32)
        mesh[b:e] = np.minimum((u[:, 7] * n_meshes).astype(np.uint32), n_meshes - 1)
    return pos, rot, scale, mesh


def make_scene(config, n=None, first=0, all_visible=False):
    """config 1..4 (BASELINE.json configs[0..3]); `n` overrides the instance count, `first`
    selects a shard of the stream, all_visible puts every instance inside the frustum cone
    (worst-case write traffic)."""
    cfg = CONFIGS[config]
    n = cfg["n"] if n is None else int(n)
    if config == 1:
        meshes = box_mesh_table()
    elif config == 2:
        meshes = damaged_helmet_mesh_table()
    else:
        meshes = mixed_mesh_table()
    seed = SEED_BASE + (3 if config == 4 else config)  # config 4 = config 3's generator, larger N
    box = ((-64.0, 64.0), (-32.0, 32.0), (-64.0, 64.0))
    if all_visible:
        box = ((-8.0, 8.0), (-3.0, 5.0), (20.0, 90.0))
    pos, rot, scale, mesh = make_instances(n, len(meshes), seed, box=box, first=first)
    return dict(config=config, name=cfg["name"], workload=cfg["workload"], n=n, meshes=meshes,
                pos=pos, rot=rot, scale=scale, mesh_id=mesh, planes=default_planes(),
                cam_pos=np.asarray(DEFAULT_CAMERA["cam_pos"], np.float32))


# ---- skinned extension (BASELINE config 5; the reference has no skinning) ----------------------

def rigged_figure_skeleton():
    """A 19-joint humanoid in the shape of the Khronos RiggedFigure sample (the asset is not in the
    container: joint count 19 as BASELINE names it; positions are a stand-in). Bind pose: every joint
    at position p_k with identity rotation, so inverse_bind_k = T(-p_k) and the bind-pose local
    translation is p_k - p_parent. joint_box_k: the bone towards the first child, padded."""
    names = ["hips", "spine", "chest", "neck", "head", "l_shoulder", "l_upper_arm", "l_forearm", "r_shoulder", "r_upper_arm",
             "r_forearm", "l_thigh", "l_shin", "l_foot", "r_thigh", "r_shin", "r_foot", "l_toe", "r_toe"]
    parent = np.array([-1, 0, 1, 2, 3, 2, 5, 6, 2, 8, 9, 0, 11, 12, 0, 14, 15, 13, 16], np.int32)
    p = np.array([[0, 0.95, 0], [0, 1.10, 0], [0, 1.30, 0], [0, 1.50, 0], [0, 1.60, 0],
                  [0.10, 1.45, 0], [0.22, 1.42, 0], [0.48, 1.42, 0], [-0.10, 1.45, 0], [-0.22, 1.42, 0], [-0.48, 1.42, 0],
                  [0.10, 0.90, 0], [0.10, 0.50, 0], [0.10, 0.08, 0], [-0.10, 0.90, 0], [-0.10, 0.50, 0], [-0.10, 0.08, 0],
                  [0.10, 0.02, 0.15], [-0.10, 0.02, 0.15]], np.float64)
    end = p.copy()  # far end of each bone: the first child, or a short stub
    for k in range(len(parent)):
        kids = np.nonzero(parent == k)[0]
        end[k] = p[kids[0]] if len(kids) else p[k] + (p[k] - p[parent[k]]) * 0.6
    end[4] = p[4] + [0, 0.22, 0]      # head
    end[7] = p[7] + [0.25, 0, 0]      # forearms reach the hands
    end[10] = p[10] - [0.25, 0, 0]
    pad = 0.07
    box = np.concatenate([np.minimum(p, end) - pad, np.maximum(p, end) + pad], axis=1).astype(np.float32)
    ibm = np.tile(np.eye(4, dtype=np.float32).T.reshape(16), (len(parent), 1))
    ibm[:, 12:15] = (-p).astype(np.float32)  # column-major: translation in elements 12..14
    local_t = p.copy()
    local_t[1:] = p[1:] - p[parent[1:]]
    return dict(names=names, parent=parent, inverse_bind=ibm, joint_box=box, bind_translation=local_t.astype(np.float32))


def rigged_figure_mesh_table():
    """One skinned mesh. Bind-pose box of the figure above; 2 LODs; index counts of the order of the
    Khronos sample (unverified, only used as table values)."""
    t = np.zeros(1, dtype=MESH_DTYPE)
    t["aabb_min"] = (-0.80, -0.05, -0.15)
    t["aabb_max"] = (0.80, 1.90, 0.30)
    t["n_lods"] = 2
    t["index_len"][0, :2] = (2304, 1152)
    t["index_offset"][0, :2] = (0, 2304)
    t["vertex_offset"] = 0
    return t


def make_poses(n, skeleton, seed, first=0, max_angle_deg=50.0):
    """Per instance and joint: bind translation, a random rotation of up to max_angle_deg about a random
    axis (4 draws), unit scale except every 7th joint slot (0.9 .. 1.1 per axis from the same draws)."""
    j = len(skeleton["parent"])
    poses = np.empty((n, j, 10), np.float32)
    poses[:, :, 0:3] = skeleton["bind_translation"][None]
    step = 1 << 16
    for b in range(0, n, step):
        e = min(n, b + step)
        u = u01(splitmix64(seed, (first + b) * j * 4, (e - b) * j * 4)).reshape(e - b, j, 4)
        z = 2.0 * u[..., 0] - 1.0
        phi = 2.0 * np.pi * u[..., 1]
        rxy = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        half = np.radians(max_angle_deg) * u[..., 2] * 0.5
        sn = np.sin(half)
        poses[b:e, :, 3] = (rxy * np.cos(phi) * sn).astype(np.float32)
        poses[b:e, :, 4] = (rxy * np.sin(phi) * sn).astype(np.float32)
        poses[b:e, :, 5] = (z * sn).astype(np.float32)
        poses[b:e, :, 6] = np.cos(half).astype(np.float32)
        sc = np.ones((e - b, j, 3))
        odd = (np.arange(j) % 7) == 3
        sc[:, odd, 0] = 0.9 + 0.2 * u[:, odd, 3]
        sc[:, odd, 1] = 1.1 - 0.2 * u[:, odd, 3]
        sc[:, odd, 2] = 0.9 + 0.2 * u[:, odd, 0]
        poses[b:e, :, 7:10] = sc.astype(np.float32)
    return poses


def make_skinned_scene(n=None, first=0):
    """BASELINE config 5: n instances of one skinned figure, each with its own pose."""
    cfg = CONFIGS[5]
    n = cfg["n"] if n is None else int(n)
    meshes = rigged_figure_mesh_table()
    pos, rot, scale, mesh = make_instances(n, 1, SEED_BASE + 5, first=first)
    sk = rigged_figure_skeleton()
    poses = make_poses(n, sk, SEED_BASE + 0x500, first=first)
    return dict(config=5, name=cfg["name"], workload=cfg["workload"], n=n, meshes=meshes, pos=pos, rot=rot, scale=scale,
                mesh_id=mesh, planes=default_planes(), cam_pos=np.asarray(DEFAULT_CAMERA["cam_pos"], np.float32),
                skeleton=sk, poses=poses)


# ---- synthetic geometry for the per-triangle stage (row f-1) --------------------------------

def default_pv():
    """CameraMatrices.pv = projection * view of the reference's default camera, float32[16]
    column-major (float64 arithmetic, rounded once), consistent with default_planes()."""
    c = DEFAULT_CAMERA
    t = np.tan(np.radians(c["fovy_degrees"]) / 2.0)
    n, f = c["near"], c["far"]
    proj = np.zeros((4, 4))
    proj[0, 0] = 1.0 / (c["aspect"] * t)
    proj[1, 1] = 1.0 / t
    proj[2, 2] = f / (f - n)
    proj[2, 3] = -(f * n) / (f - n)
    proj[3, 2] = 1.0
    view = np.eye(4)
    view[:3, 3] = -np.array(c["cam_pos"], dtype=np.float64)
    return (proj @ view).T.astype(np.float32).reshape(16)  # .T: row-major array -> column-major storage


def _torus_mesh(n_vertices, n_triangles, aabb_min, aabb_max):
    """A closed, consistently wound surface (torus grid) with exactly n_vertices positions and
    n_triangles triangles, inside the given box."""
    w = max(3, int(np.sqrt(n_vertices)))
    h = max(3, n_vertices // w)
    while w * h > n_vertices and h > 3:
        h -= 1
    if w * h > n_vertices:  # tiny meshes: fall back to a 3x3 patch, extra triangles reuse it
        w = h = 3
    u = (np.arange(w) / w) * 2 * np.pi
    v = (np.arange(h) / h) * 2 * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    big, small = 0.7, 0.3
    x = (big + small * np.cos(vv)) * np.cos(uu)
    y = small * np.sin(vv)
    z = (big + small * np.cos(vv)) * np.sin(uu)
    pts = np.stack([x, y / small * 1.0 * small, z], axis=-1).reshape(-1, 3)  # in [-1,1] x [-0.3,0.3] x [-1,1]
    pts[:, 1] /= small  # stretch y to [-1, 1]
    centre = (np.asarray(aabb_max, np.float64) + np.asarray(aabb_min, np.float64)) / 2
    half = (np.asarray(aabb_max, np.float64) - np.asarray(aabb_min, np.float64)) / 2
    pos = np.zeros((max(n_vertices, w * h), 3))
    pos[: w * h] = centre + pts * half
    pos[w * h :] = centre
    pos = pos[: max(n_vertices, w * h)]
    i, j = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    a = (i * h + j).reshape(-1)
    b = (((i + 1) % w) * h + j).reshape(-1)
    c = (((i + 1) % w) * h + (j + 1) % h).reshape(-1)
    d = (i * h + (j + 1) % h).reshape(-1)
    tris = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], axis=0)
    order = (np.arange(n_triangles, dtype=np.int64) * len(tris)) // max(n_triangles, 1) if n_triangles <= len(tris) \
        else np.arange(n_triangles, dtype=np.int64) % len(tris)
    return pos.astype(np.float32), tris[order].astype(np.uint32)


def _first_use_order(pos, tris):
    """Renumbers the vertices in the order the triangle list first uses them (what meshoptimizer's
    optimizeVertexFetch produces, and roughly what exporters write); unused vertices keep their relative order
    behind the used ones. Same surface, same triangle order."""
    flat = tris.reshape(-1).astype(np.int64)
    _, first = np.unique(flat, return_index=True)
    used = flat[np.sort(first)]
    rest = np.setdiff1d(np.arange(len(pos), dtype=np.int64), used, assume_unique=False)
    order = np.concatenate([used, rest])
    new_id = np.empty(len(pos), np.int64)
    new_id[order] = np.arange(len(pos), dtype=np.int64)
    return pos[order], new_id[tris].astype(np.uint32)


def make_geometry(meshes, ordering="rows"):
    """Consolidated position and index buffers matching a mesh table's vertex_offset /
    index_offset / index_len (what consolidate_mesh_buffers builds). Returns (vertices (V,3) f32,
    indices (I,) u32). LOD k keeps an evenly spaced subset of LOD 0's triangles.

    ordering: "rows" — the torus grid as generated: all "lower" triangles of the grid, then all "upper" ones,
    vertices row-major: a triangle's corners are a grid row (~sqrt(V) vertices) apart. "strips" — the same
    surface listed quad by quad along the strips with the vertices renumbered in first-use order: the layout
    of a vertex-cache/vertex-fetch optimised mesh. "shuffled" — the triangles of "rows" in random order: no
    locality at all (every 64-triangle step spans the whole mesh)."""
    if ordering not in ("rows", "strips", "shuffled"):
        raise ValueError(ordering)
    m = len(meshes)
    voff = meshes["vertex_offset"].astype(np.int64)
    order = np.argsort(voff, kind="stable")
    vcount = np.zeros(m, np.int64)
    for rank, k in enumerate(order):
        nxt = voff[order[rank + 1]] if rank + 1 < m else None
        len0 = int(meshes["index_len"][k, 0])
        vcount[k] = (nxt - voff[k]) if nxt is not None and nxt > voff[k] else max(len0 // 3, 9)
    total_v = int((voff + vcount).max()) if m else 0
    total_i = 0
    for k in range(m):
        for l in range(int(meshes["n_lods"][k])):
            total_i = max(total_i, int(meshes["index_offset"][k, l]) + int(meshes["index_len"][k, l]))
    vertices = np.zeros((total_v, 3), np.float32)
    indices = np.zeros(total_i, np.uint32)
    for k in range(m):
        t0 = int(meshes["index_len"][k, 0]) // 3
        pos, tris = _torus_mesh(int(vcount[k]), max(t0, 1), meshes["aabb_min"][k], meshes["aabb_max"][k])
        tris = np.minimum(tris, vcount[k] - 1)
        pos = pos[: vcount[k]]
        if ordering == "strips":
            half = len(tris) // 2  # _torus_mesh lists the lower triangle of every quad, then the upper ones
            if len(tris) == 2 * half and half > 0 and t0 >= len(tris):
                tris = np.stack([tris[:half], tris[half:]], axis=1).reshape(-1, 3)  # quad by quad
            pos, tris = _first_use_order(pos, tris)
        elif ordering == "shuffled":
            tris = tris[np.random.default_rng(1234 + k).permutation(len(tris))]
        vertices[voff[k] : voff[k] + vcount[k]] = pos
        for l in range(int(meshes["n_lods"][k])):
            tl = int(meshes["index_len"][k, l]) // 3
            if tl == 0:
                continue
            pick = (np.arange(tl, dtype=np.int64) * max(t0, 1)) // tl if t0 else np.zeros(tl, np.int64)
            off = int(meshes["index_offset"][k, l])
            indices[off : off + tl * 3] = tris[pick % len(tris)].reshape(-1)
    return vertices, indices
