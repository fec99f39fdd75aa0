"""Random scene generator for the fuzz tests: arbitrary cameras (planes from the oracle's
project_camera), mesh tables with empty LODs, special floating-point values sprinkled over every
input column, random shard bases."""
import numpy as np

from renderer_amd.pipeline import MESH_DTYPE

SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, -3.4e38, 1e19, -1e19,
                    1e-20, 0.5, 2.0, 10.0, 100.0, 16777216.0], dtype=np.float32)


def random_scene(rng, oracle, n_max=5000, special_rate=0.02):
    n = int(rng.integers(0, n_max + 1))
    m = int(rng.integers(1, 40))
    meshes = np.zeros(m, MESH_DTYPE)
    centre = rng.uniform(-1, 1, (m, 3))
    half = rng.uniform(0.01, 3, (m, 3))
    meshes["aabb_min"] = (centre - half).astype(np.float32)
    meshes["aabb_max"] = (centre + half).astype(np.float32)
    meshes["n_lods"] = rng.integers(1, 7, m)
    meshes["index_len"] = rng.integers(0, 70000, (m, 6)) // 3 * 3
    meshes["index_len"][rng.random((m, 6)) < 0.1] = 0          # empty LODs: dropped by compaction
    meshes["index_offset"] = rng.integers(0, 2 ** 31, (m, 6))
    meshes["vertex_offset"] = rng.integers(-1000, 2 ** 30, m)
    spread = float(rng.choice([5.0, 40.0, 200.0]))
    pos = rng.normal(0, spread, (n, 3)).astype(np.float32)
    rot = rng.normal(0, 1, (n, 4)).astype(np.float32)
    if rng.random() < 0.7:
        rot /= np.maximum(np.linalg.norm(rot, axis=1, keepdims=True), 1e-6).astype(np.float32)   # mostly unit quaternions
    scale = rng.uniform(-1, 3, n).astype(np.float32)
    mesh_id = rng.integers(0, m, n).astype(np.uint32)
    for arr in (pos, rot):
        hit = rng.random(arr.shape) < special_rate
        arr[hit] = rng.choice(SPECIAL, int(hit.sum()))
    hit = rng.random(n) < special_rate
    scale[hit] = rng.choice(SPECIAL, int(hit.sum()))
    cam_pos = rng.normal(0, 10, 3).astype(np.float32)
    q = rng.normal(0, 1, 4)
    q /= np.linalg.norm(q)
    planes = oracle.project_camera(cam_pos, q.astype(np.float32), aspect=float(rng.uniform(0.5, 3)),
                                   fovy_degrees=float(rng.uniform(20, 120)), near=float(rng.uniform(0.01, 1)),
                                   far=float(rng.uniform(10, 1000)))
    if rng.random() < 0.2:   # degenerate planes too: zeros, NaN, huge
        planes[rng.integers(0, 24, 3)] = rng.choice(SPECIAL, 3)
    return dict(n=n, pos=pos, rot=rot, scale=scale, mesh_id=mesh_id, meshes=meshes, planes=planes, cam_pos=cam_pos,
                first_instance_base=int(rng.integers(0, 2 ** 32)), first_index_base=int(rng.integers(0, 2 ** 32)))
