"""ctypes binding of oracle/mip_oracle.c (TEST INFRASTRUCTURE ONLY, parity unpinned)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmip_oracle.so")
if os.environ.get("MIP_ORACLE_LIBRARY"):  # the sanitizer build (make -C oracle asan), loaded by tests/test_sanitizers.py
    _SO = os.path.abspath(os.environ["MIP_ORACLE_LIBRARY"])

MAX_LODS = 6

# numpy mirrors of OrcMesh / OrcDrawCmd (= MipMesh / VkDrawIndexedIndirectCommand)
ORC_MESH_DTYPE = np.dtype(
    [
        ("aabb_min", "<f4", (3,)),
        ("aabb_max", "<f4", (3,)),
        ("n_lods", "<u4"),
        ("index_len", "<u4", (MAX_LODS,)),
        ("index_offset", "<u4", (MAX_LODS,)),
        ("vertex_offset", "<i4"),
    ]
)
assert ORC_MESH_DTYPE.itemsize == 80
DRAW_CMD_DTYPE = np.dtype(
    [
        ("indexCount", "<u4"),
        ("instanceCount", "<u4"),
        ("firstIndex", "<u4"),
        ("vertexOffset", "<i4"),
        ("firstInstance", "<u4"),
    ]
)
assert DRAW_CMD_DTYPE.itemsize == 20


class _Outputs(C.Structure):
    _fields_ = [
        ("model", C.c_void_p),
        ("world_aabb", C.c_void_p),
        ("visible_bitmap", C.c_void_p),
        ("coarse_culled", C.c_void_p),
        ("draw_cmds", C.c_void_p),
        ("draw_count", C.c_uint32),
        ("draw_index_total", C.c_uint32),
    ]


_lib = None


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    src_newer = (not os.path.exists(_SO)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("mip_oracle.c", "mip_oracle.h", "Makefile")
    )
    if os.environ.get("MIP_ORACLE_LIBRARY"):
        return _SO
    if force or src_newer:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_run.restype = C.c_int
        _lib.orc_run_mt.restype = C.c_int
        _lib.orc_coarse_culled.restype = C.c_int
        _lib.orc_pick_lod.restype = C.c_uint32
        _lib.orc_compact_draw_stream.restype = C.c_uint32
        _lib.orc_merge_draw_lists.restype = C.c_uint32
        _lib.orc_cull_all_triangles.restype = C.c_uint32
        _lib.orc_run_skinned.restype = C.c_int
    return _lib


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def model_matrix(pos, rot_ijkw, scale):
    m = np.empty(16, np.float32)
    lib().orc_model_matrix(_p(_f32(pos)), _p(_f32(rot_ijkw)), C.c_float(float(scale)), _p(m))
    return m


def world_aabb(m, mesh_min, mesh_max):
    mins = np.empty(3, np.float32)
    maxs = np.empty(3, np.float32)
    lib().orc_world_aabb(_p(_f32(m)), _p(_f32(mesh_min)), _p(_f32(mesh_max)), _p(mins), _p(maxs))
    return mins, maxs


def coarse_culled(mins, maxs, planes):
    return bool(lib().orc_coarse_culled(_p(_f32(mins)), _p(_f32(maxs)), _p(_f32(planes, (24,)))))


def pick_lod(n_lods, cam_pos, mesh_pos):
    return int(lib().orc_pick_lod(C.c_uint32(n_lods), _p(_f32(cam_pos)), _p(_f32(mesh_pos))))


def project_camera(cam_pos=(0.0, 1.0, 2.0), cam_rot_ijkw=(0.0, 0.0, 0.0, 1.0), aspect=2.0,
                   fovy_degrees=70.0, near=0.1, far=100.0):
    """Reference defaults: camera_controller.rs:21, ecs.rs:69-72, instance.rs:45 (2000x1000)."""
    planes = np.empty(24, np.float32)
    lib().orc_project_camera(_p(_f32(cam_pos)), _p(_f32(cam_rot_ijkw)), C.c_float(aspect),
                             C.c_float(fovy_degrees), C.c_float(near), C.c_float(far), _p(planes))
    return planes


def compact_draw_stream(cmds):
    cmds = np.ascontiguousarray(cmds, dtype=DRAW_CMD_DTYPE)
    out = np.zeros_like(cmds)
    n = lib().orc_compact_draw_stream(_p(cmds), C.c_uint32(len(cmds)), _p(out))
    return out[:n]


def merge_draw_lists(lists, index_totals):
    lists = [np.ascontiguousarray(l, dtype=DRAW_CMD_DTYPE) for l in lists]
    counts = np.array([len(l) for l in lists], np.uint32)
    totals = np.ascontiguousarray(index_totals, dtype=np.uint32)
    ptrs = (C.c_void_p * len(lists))(*[l.ctypes.data for l in lists])
    out = np.zeros(int(counts.sum()), DRAW_CMD_DTYPE)
    tot = C.c_uint32(0)
    n = lib().orc_merge_draw_lists(C.c_uint32(len(lists)), ptrs, _p(counts), _p(totals), _p(out),
                                   C.byref(tot))
    return out[:n], int(tot.value)


def tlas_instances(model, mesh_id, blas_address=None, first_instance_base=0):
    model = _f32(model).reshape(-1, 16)
    n = len(model)
    mesh_id = np.ascontiguousarray(mesh_id, dtype=np.uint32)
    out = np.zeros((n, 16), np.uint32)
    blas = None if blas_address is None else np.ascontiguousarray(blas_address, dtype=np.uint64)
    lib().orc_tlas_instances(C.c_uint32(n), _p(model), _p(mesh_id), _p(blas) if blas is not None else None,
                             C.c_uint32(first_instance_base), _p(out))
    return out


def light_draw_lists(pos_xyz, mesh_id, meshes, light_pos_xyz, first_instance_base=0):
    """shadow_mapping.rs:405-478 as indirect lists: (n_lights, n) commands."""
    pos = _f32(pos_xyz).reshape(-1, 3)
    n = len(pos)
    mesh_id = np.ascontiguousarray(mesh_id, dtype=np.uint32)
    meshes = np.ascontiguousarray(meshes, dtype=ORC_MESH_DTYPE)
    lights = _f32(light_pos_xyz).reshape(-1, 3)
    out = np.zeros((len(lights), n), DRAW_CMD_DTYPE)
    lib().orc_light_draw_lists(C.c_uint32(n), _p(pos), _p(mesh_id), _p(meshes), _p(lights), C.c_uint32(len(lights)),
                               C.c_uint32(first_instance_base), _p(out))
    return out


def camera_pv(cam_pos=(0.0, 1.0, 2.0), cam_rot_ijkw=(0.0, 0.0, 0.0, 1.0), aspect=2.0, fovy_degrees=70.0,
              near=0.1, far=100.0):
    pv = np.empty(16, np.float32)
    lib().orc_camera_pv(_p(_f32(cam_pos)), _p(_f32(cam_rot_ijkw)), C.c_float(aspect), C.c_float(fovy_degrees),
                        C.c_float(near), C.c_float(far), _p(pv))
    return pv


def cull_all_triangles(res, pos_xyz, mesh_id, meshes, cam_pos, pv, vertices, indices, first_instance_base=0,
                       out_capacity=None, threads=8):
    """Row f-1 on top of a `run` result (needs model, coarse_culled, draw_cmds). Returns
    (final commands, culled index stream as u32 array of length out_capacity)."""
    pos_xyz = _f32(pos_xyz).reshape(-1, 3)
    n = pos_xyz.shape[0]
    mesh_id = np.ascontiguousarray(mesh_id, dtype=np.uint32)
    meshes = np.ascontiguousarray(meshes, dtype=ORC_MESH_DTYPE)
    cmds = np.ascontiguousarray(res["draw_cmds"], dtype=DRAW_CMD_DTYPE).copy()
    count = len(cmds)
    src = np.zeros(max(count, 1), np.uint32)
    culled = np.ascontiguousarray(res["coarse_culled"], dtype=np.uint8)
    lib().orc_src_index_offsets(C.c_uint32(n), _p(pos_xyz), _p(mesh_id), _p(culled), _p(meshes), _p(_f32(cam_pos, (3,))), _p(src))
    if out_capacity is None:
        out_capacity = int(res["draw_index_total"]) + 3
    out = np.full(int(out_capacity), 0xFFFFFFFF, np.uint32)
    vertices = _f32(vertices).reshape(-1, 3)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    model = _f32(res["model"]).reshape(-1, 16)
    new_count = lib().orc_cull_all_triangles(_p(cmds), C.c_uint32(count), _p(src), _p(model), C.c_uint32(first_instance_base),
                                             _p(_f32(pv, (16,))), _p(vertices), _p(indices), _p(out), C.c_uint32(threads))
    return cmds[:new_count].copy(), out, src[:count].copy()


def run(pos_xyz, rot_ijkw, scale, mesh_id, meshes, planes, cam_pos, first_instance_base=0,
        first_index_base=0, threads=None, want=("model", "world_aabb", "visible_bitmap",
                                                "coarse_culled", "draw_cmds")):
    """Whole path. Returns a dict of numpy arrays (+ draw_count, draw_index_total)."""
    pos_xyz = _f32(pos_xyz).reshape(-1, 3)
    n = pos_xyz.shape[0]
    rot_ijkw = _f32(rot_ijkw).reshape(-1, 4)
    scale = _f32(scale).reshape(-1)
    mesh_id = np.ascontiguousarray(mesh_id, dtype=np.uint32).reshape(-1)
    meshes = np.ascontiguousarray(meshes, dtype=ORC_MESH_DTYPE).reshape(-1)
    assert rot_ijkw.shape[0] == n and scale.shape[0] == n and mesh_id.shape[0] == n
    planes = _f32(planes, (24,))
    cam_pos = _f32(cam_pos, (3,))
    res = {}
    o = _Outputs()
    if "model" in want:
        res["model"] = np.empty((n, 16), np.float32)
        o.model = res["model"].ctypes.data
    if "world_aabb" in want:
        res["world_aabb"] = np.empty((n, 6), np.float32)
        o.world_aabb = res["world_aabb"].ctypes.data
    if "visible_bitmap" in want:
        res["visible_bitmap"] = np.zeros((n + 31) // 32, np.uint32)
        o.visible_bitmap = res["visible_bitmap"].ctypes.data
    if "coarse_culled" in want:
        res["coarse_culled"] = np.empty(n, np.uint8)
        o.coarse_culled = res["coarse_culled"].ctypes.data
    cmds = None
    if "draw_cmds" in want:
        cmds = np.zeros(max(n, 1), DRAW_CMD_DTYPE)
        o.draw_cmds = cmds.ctypes.data
    args = [C.c_uint32(n), _p(pos_xyz), _p(rot_ijkw), _p(scale), _p(mesh_id), _p(meshes),
            C.c_uint32(len(meshes)), _p(planes), _p(cam_pos), C.c_uint32(first_instance_base),
            C.c_uint32(first_index_base), C.byref(o)]
    if threads is None:
        rc = lib().orc_run(*args)
    else:
        rc = lib().orc_run_mt(*args, C.c_uint32(int(threads)))
    if rc != 0:
        raise ValueError("oracle: mesh id out of range or allocation failure")
    res["draw_count"] = int(o.draw_count)
    res["draw_index_total"] = int(o.draw_index_total)
    if cmds is not None:
        res["draw_cmds"] = cmds[: o.draw_count].copy()
    return res


class Runner:
    """The whole path with the outputs allocated and touched ONCE: `runner()` then only runs orc_run(_mt) into them, as
    the reference's systems write into components that already exist. bench.py's cpu_baseline times this (round 2 timed
    `run`, whose ~85 MB of fresh, untouched output arrays per pass flattered the GPU)."""

    def __init__(self, pos_xyz, rot_ijkw, scale, mesh_id, meshes, planes, cam_pos, threads=None,
                 want=("model", "visible_bitmap", "draw_cmds"), first_instance_base=0, first_index_base=0):
        self.pos = _f32(pos_xyz).reshape(-1, 3)
        n = self.n = self.pos.shape[0]
        self.rot = _f32(rot_ijkw).reshape(-1, 4)
        self.scale = _f32(scale).reshape(-1)
        self.mesh_id = np.ascontiguousarray(mesh_id, dtype=np.uint32).reshape(-1)
        self.meshes = np.ascontiguousarray(meshes, dtype=ORC_MESH_DTYPE).reshape(-1)
        self.planes, self.cam = _f32(planes, (24,)), _f32(cam_pos, (3,))
        self.threads = threads
        self.out = {}
        o = self.o = _Outputs()
        if "model" in want:
            self.out["model"] = np.zeros((n, 16), np.float32)
            o.model = self.out["model"].ctypes.data
        if "world_aabb" in want:
            self.out["world_aabb"] = np.zeros((n, 6), np.float32)
            o.world_aabb = self.out["world_aabb"].ctypes.data
        if "visible_bitmap" in want:
            self.out["visible_bitmap"] = np.zeros((n + 31) // 32, np.uint32)
            o.visible_bitmap = self.out["visible_bitmap"].ctypes.data
        if "draw_cmds" in want:
            self.out["draw_cmds"] = np.zeros(max(n, 1), DRAW_CMD_DTYPE)
            o.draw_cmds = self.out["draw_cmds"].ctypes.data
        self.args = [C.c_uint32(n), _p(self.pos), _p(self.rot), _p(self.scale), _p(self.mesh_id), _p(self.meshes),
                     C.c_uint32(len(self.meshes)), _p(self.planes), _p(self.cam), C.c_uint32(first_instance_base),
                     C.c_uint32(first_index_base), C.byref(o)]

    def __call__(self):
        rc = lib().orc_run(*self.args) if self.threads is None else lib().orc_run_mt(*self.args, C.c_uint32(int(self.threads)))
        if rc != 0:
            raise ValueError("oracle: mesh id out of range or allocation failure")
        return int(self.o.draw_count)


def run_skinned(pos_xyz, rot_ijkw, scale, mesh_id, meshes, skeleton, poses, planes, cam_pos, first_instance_base=0,
                first_index_base=0, threads=8, want=("model", "world_aabb", "visible_bitmap", "coarse_culled",
                                                     "draw_cmds", "palette")):
    """Extension (BASELINE config 5, no reference semantics): the frame of skinned instances.
    skeleton = dict(parent int32[J], inverse_bind f32[J,16], joint_box f32[J,6]); poses f32[n,J,10]."""
    pos_xyz = _f32(pos_xyz).reshape(-1, 3)
    n = pos_xyz.shape[0]
    rot_ijkw = _f32(rot_ijkw).reshape(-1, 4)
    scale = _f32(scale).reshape(-1)
    mesh_id = np.ascontiguousarray(mesh_id, dtype=np.uint32).reshape(-1)
    meshes = np.ascontiguousarray(meshes, dtype=ORC_MESH_DTYPE).reshape(-1)
    parent = np.ascontiguousarray(skeleton["parent"], dtype=np.int32)
    n_joints = len(parent)
    ibm = _f32(skeleton["inverse_bind"]).reshape(n_joints, 16)
    box = _f32(skeleton["joint_box"]).reshape(n_joints, 6)
    poses = _f32(poses).reshape(n, n_joints, 10)
    planes = _f32(planes, (24,))
    cam_pos = _f32(cam_pos, (3,))
    res = {}
    o = _Outputs()
    if "model" in want:
        res["model"] = np.empty((n, 16), np.float32)
        o.model = res["model"].ctypes.data
    if "world_aabb" in want:
        res["world_aabb"] = np.empty((n, 6), np.float32)
        o.world_aabb = res["world_aabb"].ctypes.data
    if "visible_bitmap" in want:
        res["visible_bitmap"] = np.zeros((n + 31) // 32, np.uint32)
        o.visible_bitmap = res["visible_bitmap"].ctypes.data
    if "coarse_culled" in want:
        res["coarse_culled"] = np.empty(n, np.uint8)
        o.coarse_culled = res["coarse_culled"].ctypes.data
    cmds = None
    if "draw_cmds" in want:
        cmds = np.zeros(max(n, 1), DRAW_CMD_DTYPE)
        o.draw_cmds = cmds.ctypes.data
    palette = None
    if "palette" in want:
        palette = res["palette"] = np.empty((n, n_joints, 16), np.float32)
    local_box = res["local_box"] = np.empty((n, 6), np.float32)
    rc = lib().orc_run_skinned(C.c_uint32(n), _p(pos_xyz), _p(rot_ijkw), _p(scale), _p(mesh_id), _p(meshes),
                               C.c_uint32(len(meshes)), C.c_uint32(n_joints), _p(parent), _p(ibm), _p(box), _p(poses),
                               _p(planes), _p(cam_pos), C.c_uint32(first_instance_base), C.c_uint32(first_index_base),
                               C.byref(o), _p(palette) if palette is not None else None, _p(local_box),
                               C.c_uint32(int(threads)))
    if rc != 0:
        raise ValueError("oracle: bad skeleton, mesh id out of range or allocation failure")
    res["draw_count"] = int(o.draw_count)
    res["draw_index_total"] = int(o.draw_index_total)
    if cmds is not None:
        res["draw_cmds"] = cmds[: o.draw_count].copy()
    return res
