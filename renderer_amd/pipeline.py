"""ctypes host wrapper over the C ABI. Mirrors the order the reference's schedule runs the
path in (src/main.rs:780-839 RenderSetup, then cull_pass): upload the ECS columns once,
then one `run` per frame."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MipConfig, MipError, MipFrame, MipOutputs, MipShardedOutputs, MipTimings

MESH_DTYPE = np.dtype(
    [
        ("aabb_min", "<f4", (3,)),
        ("aabb_max", "<f4", (3,)),
        ("n_lods", "<u4"),
        ("index_len", "<u4", (_lib.MIP_MAX_LODS,)),
        ("index_offset", "<u4", (_lib.MIP_MAX_LODS,)),
        ("vertex_offset", "<i4"),
    ]
)
DRAW_CMD_DTYPE = np.dtype(
    [
        ("indexCount", "<u4"),
        ("instanceCount", "<u4"),
        ("firstIndex", "<u4"),
        ("vertexOffset", "<i4"),
        ("firstInstance", "<u4"),
    ]
)
SHARD_HEADER_BYTES = 32


def wire_form(wire):
    """0 = 20-byte commands, 1 = 8-byte wire records (MIP_OUT_WIRE), 2 = packed 4-byte records (MIP_OUT_WIRE_PACKED).
    Accepts False / True / 1 / 2 / "packed"."""
    if wire in (2, "packed"):
        return 2
    return 1 if wire else 0


def wire_index_bits(n_meshes):
    """mip_wire_index_bits: what a table of n_meshes entries leaves of a packed record for the instance index."""
    mesh_bits = 0
    while mesh_bits < 31 and (1 << mesh_bits) < int(n_meshes):
        mesh_bits += 1
    return 31 - mesh_bits


def wire_body_bytes(capacity, packed=False):
    """MIP_WIRE_BODY_BYTES / MIP_WIRE_PACKED_BODY_BYTES: whole blocks of 256 8-byte records, or of 64 packed 4-byte records,
    each behind a 16-byte block header."""
    per = _lib.MIP_WIRE_PACKED_BLOCK_COMMANDS if packed else _lib.MIP_WIRE_BLOCK_COMMANDS
    blocks = (int(capacity) + per - 1) // per
    return blocks * (_lib.MIP_WIRE_PACKED_BLOCK_BYTES if packed else _lib.MIP_WIRE_BLOCK_BYTES)


def make_frame(planes, cam_pos, first_instance_base=0, first_index_base=0, pv=None):
    f = MipFrame()
    if pv is not None:
        f.pv[:] = np.ascontiguousarray(pv, dtype=np.float32).reshape(16).tolist()
    planes = np.ascontiguousarray(planes, dtype=np.float32).reshape(24)
    cam_pos = np.ascontiguousarray(cam_pos, dtype=np.float32).reshape(3)
    f.planes[:] = planes.tolist()
    f.cam_pos[:] = cam_pos.tolist()
    f.first_instance_base = int(first_instance_base) & 0xFFFFFFFF
    f.first_index_base = int(first_index_base) & 0xFFFFFFFF
    return f


class InstancePipeline:
    """One context on one GPU (one per rank)."""

    def __init__(self, max_instances, max_meshes, device=0, timing=False, stream=None, frames_in_flight=1,
                 ordered_tiles=False):
        self._lib = _lib.load_library()
        self._ctx = C.c_void_p()
        cfg = MipConfig()
        cfg.struct_size = C.sizeof(MipConfig)
        cfg.device_ordinal = int(device)
        cfg.max_instances = int(max_instances)
        cfg.max_meshes = int(max_meshes)
        cfg.flags = (_lib.MIP_CFG_TIMING if timing else 0) | (_lib.MIP_CFG_ORDERED_TILES if ordered_tiles else 0)
        cfg.frames_in_flight = int(frames_in_flight)
        cfg.stream = stream
        rc = self._lib.mip_create(C.byref(cfg), C.byref(self._ctx))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise MipError(rc, "mip_create failed (is there a gfx950 GPU?)")
        self.max_instances = int(max_instances)
        self.n = 0
        self.stream = stream  # the caller's HIP stream handle, or None when the context created its own

    # -- lifetime --
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.mip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.mip_last_error(self._ctx)
            raise MipError(rc, msg.decode() if msg else "")

    # -- uploads --
    def set_mesh_table(self, meshes):
        meshes = np.ascontiguousarray(meshes, dtype=MESH_DTYPE).reshape(-1)
        self._check(self._lib.mip_set_mesh_table(self._ctx, meshes.ctypes.data, len(meshes)))
        self.n_meshes = len(meshes)

    def set_instances(self, pos_xyz, rot_ijkw, scale, mesh_id):
        pos = np.ascontiguousarray(pos_xyz, dtype=np.float32).reshape(-1, 3)
        rot = np.ascontiguousarray(rot_ijkw, dtype=np.float32).reshape(-1, 4)
        scl = np.ascontiguousarray(scale, dtype=np.float32).reshape(-1)
        mid = np.ascontiguousarray(mesh_id, dtype=np.uint32).reshape(-1)
        n = pos.shape[0]
        if not (rot.shape[0] == n and scl.shape[0] == n and mid.shape[0] == n):
            raise ValueError("instance columns differ in length")
        self._check(self._lib.mip_set_instances(self._ctx, pos.ctypes.data, rot.ctypes.data,
                                                scl.ctypes.data, mid.ctypes.data, n))
        self.n = n

    def set_geometry(self, vertex_xyz, indices):
        """Consolidated position / index buffers for the per-triangle stage."""
        v = np.ascontiguousarray(vertex_xyz, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        self._check(self._lib.mip_set_geometry(self._ctx, v.ctypes.data, len(v), i.ctypes.data, len(i)))

    def update_instances(self, first, pos_xyz=None, rot_ijkw=None, scale=None, mesh_id=None):
        """Overwrite a range of the resident columns (None = keep)."""
        cols, count = [], None
        for arr, dtype, width in ((pos_xyz, np.float32, 3), (rot_ijkw, np.float32, 4), (scale, np.float32, 1), (mesh_id, np.uint32, 1)):
            if arr is None:
                cols.append(None)
                continue
            a = np.ascontiguousarray(arr, dtype=dtype).reshape(-1, width)
            if count is not None and len(a) != count:
                raise ValueError("columns differ in length")
            count = len(a)
            cols.append(a)
        ptrs = [c.ctypes.data if c is not None else None for c in cols]
        self._check(self._lib.mip_update_instances(self._ctx, int(first), int(count or 0), *ptrs))

    def set_blas_addresses(self, addresses):
        a = np.ascontiguousarray(addresses, dtype=np.uint64).reshape(-1)
        self._check(self._lib.mip_set_blas_addresses(self._ctx, a.ctypes.data, len(a)))

    def set_instances_device(self, pos_ptr, rot_ptr, scale_ptr, mesh_id_ptr, n):
        self._check(self._lib.mip_set_instances_device(self._ctx, pos_ptr, rot_ptr, scale_ptr,
                                                       mesh_id_ptr, int(n)))
        self.n = int(n)

    # -- per frame --
    def run_host(self, planes, cam_pos, first_instance_base=0, first_index_base=0,
                 want=("model", "visible_bitmap", "draw_cmds", "world_aabb")):
        """Synchronous; results copied back into fresh numpy arrays (PCIe-inclusive)."""
        n = self.n
        frame = make_frame(planes, cam_pos, first_instance_base, first_index_base)
        out = MipOutputs()
        out.flags = _lib.MIP_OUT_HOST
        res = {}
        if "model" in want:
            res["model"] = np.zeros((n, 16), np.float32)
            out.model = res["model"].ctypes.data
        if "visible_bitmap" in want:
            res["visible_bitmap"] = np.zeros((n + 31) // 32, np.uint32)
            out.visible_bitmap = res["visible_bitmap"].ctypes.data
        if "world_aabb" in want:
            res["world_aabb"] = np.zeros((n, 6), np.float32)
            out.world_aabb = res["world_aabb"].ctypes.data
        count = C.c_uint32(0)
        total = C.c_uint32(0)
        cmds = None
        if "draw_cmds" in want:
            cmds = np.zeros(max(n, 1), DRAW_CMD_DTYPE)
            out.draw_cmds = cmds.ctypes.data
            out.draw_count = C.addressof(count)
            out.draw_index_total = C.addressof(total)
        self._check(self._lib.mip_run(self._ctx, C.byref(frame), C.byref(out)))
        if cmds is not None:
            res["draw_cmds"] = cmds[: count.value].copy()
            res["draw_count"] = int(count.value)
            res["draw_index_total"] = int(total.value)
        return res

    def run_device(self, frame, model=0, visible_bitmap=0, draw_cmds=0, draw_count=0,
                   draw_index_total=0, world_aabb=0, async_=False, culled_index_buffer=0, culled_index_capacity=0,
                   tlas_instances=0, wire=False):
        """Device pointers in, nothing copied. `frame` from make_frame(). wire=True: draw_cmds receives the
        wire form of the list (MIP_OUT_WIRE); wire="packed" (or 2): its packed form (MIP_OUT_WIRE_PACKED)."""
        out = MipOutputs()
        form = wire_form(wire)
        out.flags = (_lib.MIP_OUT_DEVICE | (_lib.MIP_OUT_ASYNC if async_ else 0) | (_lib.MIP_OUT_WIRE if form else 0)
                     | (_lib.MIP_OUT_WIRE_PACKED if form == 2 else 0))
        out.model = model or None
        out.visible_bitmap = visible_bitmap or None
        out.draw_cmds = draw_cmds or None
        out.draw_count = draw_count or None
        out.draw_index_total = draw_index_total or None
        out.world_aabb = world_aabb or None
        out.culled_index_buffer = culled_index_buffer or None
        out.culled_index_capacity = int(culled_index_capacity)
        out.tlas_instances = tlas_instances or None
        self._check(self._lib.mip_run(self._ctx, C.byref(frame), C.byref(out)))

    def prepare_outputs(self, model=0, visible_bitmap=0, draw_cmds=0, draw_count=0, draw_index_total=0,
                        world_aabb=0, async_=True, culled_index_buffer=0, culled_index_capacity=0):
        """A reusable MipOutputs (device pointers) for run_prepared: keeps the per-frame host cost
        to one foreign call."""
        out = MipOutputs()
        out.flags = _lib.MIP_OUT_DEVICE | (_lib.MIP_OUT_ASYNC if async_ else 0)
        out.model = model or None
        out.visible_bitmap = visible_bitmap or None
        out.draw_cmds = draw_cmds or None
        out.draw_count = draw_count or None
        out.draw_index_total = draw_index_total or None
        out.world_aabb = world_aabb or None
        out.culled_index_buffer = culled_index_buffer or None
        out.culled_index_capacity = int(culled_index_capacity)
        out._as_parameter_ = C.c_void_p(C.addressof(out))  # lets ctypes pass the struct by address
        return out

    @staticmethod
    def frame_ref(frame):
        """A MipFrame prepared for run_prepared (passed by address)."""
        frame._as_parameter_ = C.c_void_p(C.addressof(frame))
        return frame

    def run_prepared(self, frame, outputs):
        rc = self._lib.mip_run(self._ctx, frame, outputs)
        if rc != 0:
            self._check(rc)

    def run_many(self, frames, prepared_outputs, steps):
        """`steps` frames issued from compiled code: step k runs frames[k % len(frames)] into
        prepared_outputs[k % len(prepared_outputs)]. `frames` is one MipFrame or a sequence of them (a moving
        camera); recorded launch graphs are reused whatever the frames are."""
        if isinstance(frames, MipFrame):
            frames = [frames]
        fr = (MipFrame * len(frames))()
        for k, f in enumerate(frames):
            C.memmove(C.addressof(fr[k]), C.addressof(f), C.sizeof(MipFrame))
        arr = (MipOutputs * len(prepared_outputs))()
        for k, o in enumerate(prepared_outputs):
            C.memmove(C.addressof(arr[k]), C.addressof(o), C.sizeof(MipOutputs))
        rc = self._lib.mip_run_many(self._ctx, C.addressof(fr), len(frames), C.addressof(arr), len(prepared_outputs), int(steps))
        if rc != 0:
            self._check(rc)

    # -- zero-copy interop (row f-2) --
    def import_external_fd(self, fd, size_bytes):
        """Maps memory another API exported as an fd (VK_KHR_external_memory_fd / a dma-buf) into this
        context's device; returns the device pointer. The fd belongs to the driver afterwards."""
        ptr = C.c_void_p()
        self._check(self._lib.mip_import_external_fd(self._ctx, int(fd), int(size_bytes), C.byref(ptr)))
        return ptr.value

    def release_external(self, ptr):
        self._check(self._lib.mip_release_external(self._ctx, ptr))

    def import_external_semaphore_fd(self, fd, timeline=True):
        """Imports a semaphore another API exported as an fd (vkGetSemaphoreFdKHR); returns the opaque handle."""
        h = C.c_void_p()
        kind = _lib.MIP_SEMAPHORE_TIMELINE if timeline is True else (_lib.MIP_SEMAPHORE_BINARY if timeline is False else int(timeline))
        self._check(self._lib.mip_import_external_semaphore_fd(self._ctx, int(fd), kind, C.byref(h)))
        return h.value

    def external_semaphore_on_device(self, semaphore):
        """True: the HIP runtime imported the semaphore (device-side waits/signals); False: the DRM sync object path."""
        rc = self._lib.mip_external_semaphore_on_device(self._ctx, semaphore)
        if rc < 0:
            self._check(rc)
        return bool(rc)

    def wait_external(self, semaphore, value=0):
        """The next frame's stream waits on the device until the semaphore reaches `value`."""
        self._check(self._lib.mip_wait_external(self._ctx, semaphore, int(value)))

    def signal_external(self, semaphore, value=0):
        """Signals the semaphore to `value` behind the frame issued last."""
        self._check(self._lib.mip_signal_external(self._ctx, semaphore, int(value)))

    def release_external_semaphore(self, semaphore):
        self._check(self._lib.mip_release_external_semaphore(self._ctx, semaphore))

    # -- native sharded exchange (RCCL opened by the library itself) --
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from ncclGetUniqueId; create on one rank, share with the others."""
        buf = (C.c_uint8 * 128)()
        rc = _lib.load_library().mip_comm_unique_id(buf)
        if rc != 0:
            raise MipError(rc, "mip_comm_unique_id failed (is librccl.so.1 loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.mip_comm_init(self._ctx, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._lib.mip_comm_destroy(self._ctx))

    def run_sharded(self, frame, draw_cmds, draw_count, model=0, visible_bitmap=0, world_aabb=0, chunk_capacity=0,
                    async_=False):
        """Shard kernel -> one ncclAllGather -> merge, all inside the library."""
        out = MipShardedOutputs()
        out.model = model or None
        out.visible_bitmap = visible_bitmap or None
        out.world_aabb = world_aabb or None
        out.draw_cmds = draw_cmds
        out.draw_count = draw_count
        out.chunk_capacity = int(chunk_capacity)
        out.flags = _lib.MIP_OUT_DEVICE | (_lib.MIP_OUT_ASYNC if async_ else 0)
        self._check(self._lib.mip_run_sharded(self._ctx, C.addressof(frame), C.addressof(out)))

    def wait(self):
        self._check(self._lib.mip_wait(self._ctx))

    def merge_draw_lists(self, chunks_ptr, n_chunks, chunk_stride_bytes, out_cmds_ptr, out_count_ptr,
                         async_=False, chunk_capacity=0):
        """chunk_capacity = commands one chunk may carry (out_cmds has room for n_chunks x that); 0 = what
        the stride holds. A chunk whose header count exceeds it is cut and reported (MIP_ERR_CAPACITY)."""
        self._check(self._lib.mip_merge_draw_lists(self._ctx, chunks_ptr, int(n_chunks),
                                                   int(chunk_stride_bytes), int(chunk_capacity), out_cmds_ptr,
                                                   out_count_ptr, 1 if async_ else 0))

    def merge_wire_lists(self, chunks_ptr, n_chunks, chunk_stride_bytes, out_cmds_ptr, out_count_ptr,
                         async_=False, chunk_capacity=0, packed=False):
        """The same merge over chunks in the wire form (MIP_OUT_WIRE; packed=True: MIP_OUT_WIRE_PACKED), expanded against
        this context's mesh table."""
        fn = self._lib.mip_merge_wire_lists_packed if packed else self._lib.mip_merge_wire_lists
        self._check(fn(self._ctx, chunks_ptr, int(n_chunks), int(chunk_stride_bytes), int(chunk_capacity), out_cmds_ptr,
                       out_count_ptr, 1 if async_ else 0))

    # -- extension: skinned instances (BASELINE config 5; not a reference behaviour) --
    def set_skeleton(self, parent, inverse_bind, joint_box):
        parent = np.ascontiguousarray(parent, dtype=np.int32)
        j = len(parent)
        ibm = np.ascontiguousarray(inverse_bind, dtype=np.float32).reshape(j, 16)
        box = np.ascontiguousarray(joint_box, dtype=np.float32).reshape(j, 6)
        self._check(self._lib.mip_set_skeleton(self._ctx, parent.ctypes.data, ibm.ctypes.data, box.ctypes.data, j))
        self._n_joints = j

    def set_poses(self, joint_trs):
        """Host array n x J x 10 (t xyz, q ijkw, s xyz per joint); copied."""
        poses = np.ascontiguousarray(joint_trs, dtype=np.float32)
        n = poses.size // (max(getattr(self, "_n_joints", 0), 1) * 10)
        self._check(self._lib.mip_set_poses(self._ctx, poses.ctypes.data, n, 0))

    def set_poses_device(self, ptr, n):
        """Borrow a device buffer of n x J x 10 floats (not copied)."""
        self._check(self._lib.mip_set_poses(self._ctx, ptr, int(n), 1))

    def run_skinned(self, frame, palette=0, async_=False, **outputs):
        """One frame of skinned instances; outputs as prepare_outputs (device pointers)."""
        out = self.prepare_outputs(async_=async_, **outputs)
        rc = self._lib.mip_run_skinned(self._ctx, C.addressof(frame), C.addressof(out), palette or None)
        if rc != 0:
            self._check(rc)

    def run_views(self, frames, prepared_outputs):
        """Up to 16 views (per-light culled lists, cascades ...) of the resident instances, four per launch:
        frames[v] (make_frame) with prepared_outputs[v] (prepare_outputs: bitmap / draw_cmds / draw_count / index total)."""
        k = len(frames)
        fr = (MipFrame * k)()
        ou = (MipOutputs * k)()
        for v in range(k):
            C.memmove(C.addressof(fr[v]), C.addressof(frames[v]), C.sizeof(MipFrame))
            C.memmove(C.addressof(ou[v]), C.addressof(prepared_outputs[v]), C.sizeof(MipOutputs))
        self._check(self._lib.mip_run_views(self._ctx, C.addressof(fr), C.addressof(ou), k))

    def light_draw_lists(self, light_pos_xyz, out_cmds_ptr, first_instance_base=0, async_=False):
        """Per-light shadow-pass draw lists (shadow_mapping.rs:405-478): n_lights x n commands, light-major,
        into device memory at out_cmds_ptr."""
        lights = np.ascontiguousarray(light_pos_xyz, dtype=np.float32).reshape(-1, 3)
        self._check(self._lib.mip_light_draw_lists(self._ctx, lights.ctypes.data, len(lights), int(first_instance_base),
                                                   out_cmds_ptr, 1 if async_ else 0))

    # -- diagnostics --
    def timings(self):
        t = MipTimings()
        self._check(self._lib.mip_get_timings(self._ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in MipTimings._fields_}

    def reset_timings(self):
        self._check(self._lib.mip_reset_timings(self._ctx))
