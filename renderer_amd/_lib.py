"""Loads libmi_instance_pipeline.so and declares the C ABI (include/mi_instance_pipeline.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libmi_instance_pipeline.so")
if os.environ.get("MIP_LIBRARY"):  # tuning/diagnostic builds only (tools/); never set by the product path
    _SO = os.path.abspath(os.environ["MIP_LIBRARY"])

MIP_OK = 0
ERR_NAMES = {
    -1: "MIP_ERR_INVALID_ARGUMENT",
    -2: "MIP_ERR_NO_DEVICE",
    -3: "MIP_ERR_OUT_OF_MEMORY",
    -4: "MIP_ERR_CAPACITY",
    -5: "MIP_ERR_DEVICE",
    -6: "MIP_ERR_NOT_READY",
    -7: "MIP_ERR_TIMEOUT",
}
MIP_CFG_TIMING = 0x1
MIP_CFG_ORDERED_TILES = 0x2
MIP_OUT_HOST = 0x0
MIP_OUT_DEVICE = 0x1
MIP_OUT_ASYNC = 0x2
MIP_OUT_WIRE = 0x4
MIP_OUT_WIRE_PACKED = 0x8
MIP_WIRE_BLOCK_COMMANDS = 256
MIP_WIRE_SUB_BLOCK_COMMANDS = 64
MIP_WIRE_BLOCK_HEADER_BYTES = 16
MIP_WIRE_RECORD_BYTES = 8
MIP_WIRE_BLOCK_BYTES = MIP_WIRE_BLOCK_HEADER_BYTES + MIP_WIRE_BLOCK_COMMANDS * MIP_WIRE_RECORD_BYTES
MIP_WIRE_PACKED_RECORD_BYTES = 4
MIP_WIRE_PACKED_BLOCK_COMMANDS = 64
MIP_WIRE_PACKED_BLOCK_BYTES = MIP_WIRE_BLOCK_HEADER_BYTES + MIP_WIRE_PACKED_BLOCK_COMMANDS * MIP_WIRE_PACKED_RECORD_BYTES
MIP_MAX_LODS = 6
MIP_SEMAPHORE_BINARY = 0
MIP_SEMAPHORE_TIMELINE = 1

# Every symbol include/mi_instance_pipeline.h declares.
EXPORTS = (
    "mip_abi_version", "mip_create", "mip_destroy", "mip_set_mesh_table", "mip_set_instances",
    "mip_set_instances_device", "mip_update_instances", "mip_set_geometry", "mip_set_blas_addresses", "mip_run", "mip_run_many", "mip_wait", "mip_merge_draw_lists", "mip_merge_wire_lists", "mip_merge_wire_lists_packed", "mip_wire_index_bits", "mip_light_draw_lists", "mip_set_skeleton", "mip_set_poses", "mip_run_skinned", "mip_run_views", "mip_comm_unique_id", "mip_comm_init", "mip_comm_destroy", "mip_run_sharded", "mip_import_external_fd", "mip_release_external", "mip_import_external_semaphore_fd", "mip_external_semaphore_on_device", "mip_wait_external", "mip_signal_external", "mip_release_external_semaphore", "mip_last_error",
    "mip_get_timings", "mip_reset_timings", "mip_instance_count",
)


class MipError(RuntimeError):
    def __init__(self, code, message=""):
        self.code = code
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")


class MipConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device_ordinal", C.c_int32),
        ("max_instances", C.c_uint32),
        ("max_meshes", C.c_uint32),
        ("flags", C.c_uint32),
        ("frames_in_flight", C.c_uint32),
        ("stream", C.c_void_p),
    ]


class MipFrame(C.Structure):
    _fields_ = [
        ("planes", C.c_float * 24),
        ("cam_pos", C.c_float * 3),
        ("first_instance_base", C.c_uint32),
        ("first_index_base", C.c_uint32),
        ("pv", C.c_float * 16),
    ]


class MipOutputs(C.Structure):
    _fields_ = [
        ("model", C.c_void_p),
        ("visible_bitmap", C.c_void_p),
        ("draw_cmds", C.c_void_p),
        ("draw_count", C.c_void_p),
        ("draw_index_total", C.c_void_p),
        ("world_aabb", C.c_void_p),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
        ("culled_index_buffer", C.c_void_p),
        ("culled_index_capacity", C.c_uint64),
        ("tlas_instances", C.c_void_p),
    ]


class MipShardedOutputs(C.Structure):
    _fields_ = [
        ("model", C.c_void_p),
        ("visible_bitmap", C.c_void_p),
        ("world_aabb", C.c_void_p),
        ("draw_cmds", C.c_void_p),
        ("draw_count", C.c_void_p),
        ("chunk_capacity", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class MipTimings(C.Structure):
    _fields_ = [
        ("runs", C.c_uint64),
        ("last_kernel_ms", C.c_double),
        ("total_kernel_ms", C.c_double),
        ("last_merge_ms", C.c_double),
        ("total_merge_ms", C.c_double),
        ("merges", C.c_uint64),
        ("graph_frames", C.c_uint64),
        ("graph_records", C.c_uint64),
        ("sharded_retries", C.c_uint64),
        ("sharded_bytes_sent", C.c_uint64),
        ("prefix_helps", C.c_uint64),
        ("general_launches", C.c_uint64),
        ("reserved0", C.c_uint64),
    ]


_lib = None


def library_path():
    return _SO


def _declare_newer(lib, name, argtypes, restype=C.c_int32):
    """Entry points added after ABI 2. The product library has them all (tests/test_abi.py); an older build loaded through
    MIP_LIBRARY for a same-box A/B (tools/kbench.py against a previous round's kernel) may not — only then is one skipped."""
    try:
        fn = getattr(lib, name)
    except AttributeError:
        if os.environ.get("MIP_LIBRARY"):
            return
        raise
    fn.argtypes = argtypes
    fn.restype = restype


def load_library():
    """Returns the ctypes handle. Raises ImportError if the HIP library has not been built —
    there is deliberately nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C renderer_amd/csrc`. The instance pipeline has no CPU fallback."
        )
    # torch wheels bundle their own libamdhip64.so.7. Whichever copy of that SONAME is loaded first
    # serves the whole process, and torch cannot initialise on top of /opt/rocm's copy ("No HIP GPUs
    # are available"), so in a process that has torch, torch's runtime must come first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(_SO)
    vp = C.c_void_p
    lib.mip_abi_version.restype = C.c_uint32
    lib.mip_create.argtypes = [C.POINTER(MipConfig), C.POINTER(vp)]
    lib.mip_create.restype = C.c_int32
    lib.mip_destroy.argtypes = [vp]
    lib.mip_destroy.restype = None
    lib.mip_set_mesh_table.argtypes = [vp, vp, C.c_uint32]
    lib.mip_set_mesh_table.restype = C.c_int32
    lib.mip_set_instances.argtypes = [vp, vp, vp, vp, vp, C.c_uint32]
    lib.mip_set_instances.restype = C.c_int32
    lib.mip_set_instances_device.argtypes = [vp, vp, vp, vp, vp, C.c_uint32]
    lib.mip_set_instances_device.restype = C.c_int32
    lib.mip_update_instances.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    lib.mip_update_instances.restype = C.c_int32
    lib.mip_set_blas_addresses.argtypes = [vp, vp, C.c_uint32]
    lib.mip_set_blas_addresses.restype = C.c_int32
    lib.mip_set_geometry.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.mip_set_geometry.restype = C.c_int32
    lib.mip_run.argtypes = [vp, C.c_void_p, C.c_void_p]
    lib.mip_run.restype = C.c_int32
    lib.mip_run_many.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32]
    lib.mip_run_many.restype = C.c_int32
    lib.mip_wait.argtypes = [vp]
    lib.mip_wait.restype = C.c_int32
    lib.mip_merge_draw_lists.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, C.c_int32]
    lib.mip_merge_draw_lists.restype = C.c_int32
    _declare_newer(lib, "mip_merge_wire_lists", [vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, C.c_int32])
    _declare_newer(lib, "mip_merge_wire_lists_packed", [vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, C.c_int32])
    _declare_newer(lib, "mip_wire_index_bits", [C.c_uint32], restype=C.c_uint32)
    lib.mip_light_draw_lists.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_int32]
    lib.mip_light_draw_lists.restype = C.c_int32
    lib.mip_set_skeleton.argtypes = [vp, vp, vp, vp, C.c_uint32]
    lib.mip_set_skeleton.restype = C.c_int32
    lib.mip_set_poses.argtypes = [vp, vp, C.c_uint32, C.c_int32]
    lib.mip_set_poses.restype = C.c_int32
    lib.mip_run_skinned.argtypes = [vp, vp, vp, vp]
    lib.mip_run_skinned.restype = C.c_int32
    lib.mip_run_views.argtypes = [vp, vp, vp, C.c_uint32]
    lib.mip_run_views.restype = C.c_int32
    lib.mip_comm_unique_id.argtypes = [vp]
    lib.mip_comm_unique_id.restype = C.c_int32
    lib.mip_comm_init.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    lib.mip_comm_init.restype = C.c_int32
    lib.mip_comm_destroy.argtypes = [vp]
    lib.mip_comm_destroy.restype = C.c_int32
    lib.mip_run_sharded.argtypes = [vp, vp, vp]
    lib.mip_run_sharded.restype = C.c_int32
    lib.mip_import_external_fd.argtypes = [vp, C.c_int32, C.c_uint64, C.POINTER(vp)]
    lib.mip_import_external_fd.restype = C.c_int32
    lib.mip_release_external.argtypes = [vp, vp]
    lib.mip_release_external.restype = C.c_int32
    _declare_newer(lib, "mip_import_external_semaphore_fd", [vp, C.c_int32, C.c_uint32, C.POINTER(vp)])
    _declare_newer(lib, "mip_external_semaphore_on_device", [vp, vp])
    _declare_newer(lib, "mip_wait_external", [vp, vp, C.c_uint64])
    _declare_newer(lib, "mip_signal_external", [vp, vp, C.c_uint64])
    _declare_newer(lib, "mip_release_external_semaphore", [vp, vp])
    lib.mip_last_error.argtypes = [vp]
    lib.mip_last_error.restype = C.c_char_p
    lib.mip_get_timings.argtypes = [vp, C.POINTER(MipTimings)]
    lib.mip_get_timings.restype = C.c_int32
    lib.mip_reset_timings.argtypes = [vp]
    lib.mip_reset_timings.restype = C.c_int32
    lib.mip_instance_count.argtypes = [vp]
    lib.mip_instance_count.restype = C.c_uint32
    _lib = lib
    return lib
