//! Raw bindings of include/mi_instance_pipeline.h (what bindgen would generate) plus the
//! newtype that makes the context usable as a bevy resource, like `VmaAllocator` in
//! vma/src/lib.rs:31-40.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const MIP_OK: i32 = 0;
pub const MIP_OUT_HOST: u32 = 0;
pub const MIP_OUT_DEVICE: u32 = 1;
pub const MIP_OUT_ASYNC: u32 = 2;
pub const MIP_OUT_WIRE: u32 = 4;
pub const MIP_OUT_WIRE_PACKED: u32 = 8;
pub const MIP_SEMAPHORE_BINARY: u32 = 0;
pub const MIP_SEMAPHORE_TIMELINE: u32 = 1;
pub const MIP_MAX_LODS: usize = 6;

#[repr(C)]
pub struct MipContext {
    _private: [u8; 0],
}

/// Opaque handle of an imported external semaphore (mip_import_external_semaphore_fd).
#[repr(C)]
pub struct MipExternalSemaphore {
    _private: [u8; 0],
}

#[repr(transparent)]
pub struct Pipeline(pub *mut MipContext);
unsafe impl Send for Pipeline {}
unsafe impl Sync for Pipeline {}

#[repr(C)]
pub struct MipConfig {
    pub struct_size: u32,
    pub device_ordinal: i32,
    pub max_instances: u32,
    pub max_meshes: u32,
    pub flags: u32,
    pub frames_in_flight: u32,
    pub stream: *mut c_void,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct MipMesh {
    pub aabb_min: [f32; 3],
    pub aabb_max: [f32; 3],
    pub n_lods: u32,
    pub index_len: [u32; MIP_MAX_LODS],
    pub index_offset: [u32; MIP_MAX_LODS],
    pub vertex_offset: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct MipDrawIndexedIndirectCommand {
    pub index_count: u32,
    pub instance_count: u32,
    pub first_index: u32,
    pub vertex_offset: i32,
    pub first_instance: u32,
}

#[repr(C)]
pub struct MipFrame {
    pub planes: [f32; 24],
    pub cam_pos: [f32; 3],
    pub first_instance_base: u32,
    pub first_index_base: u32,
    pub pv: [f32; 16],
}

#[repr(C)]
pub struct MipOutputs {
    pub model: *mut c_void,
    pub visible_bitmap: *mut u32,
    pub draw_cmds: *mut c_void,
    pub draw_count: *mut u32,
    pub draw_index_total: *mut u32,
    pub world_aabb: *mut c_void,
    pub flags: u32,
    pub reserved: u32,
    pub culled_index_buffer: *mut c_void,
    pub culled_index_capacity: u64,
    pub tlas_instances: *mut c_void,
}

#[repr(C)]
pub struct MipShardedOutputs {
    pub model: *mut c_void,
    pub visible_bitmap: *mut u32,
    pub world_aabb: *mut c_void,
    pub draw_cmds: *mut c_void,
    pub draw_count: *mut u32,
    pub chunk_capacity: u32,
    pub flags: u32,
}

extern "C" {
    pub fn mip_abi_version() -> u32;
    pub fn mip_create(cfg: *const MipConfig, out: *mut *mut MipContext) -> i32;
    pub fn mip_destroy(ctx: *mut MipContext);
    pub fn mip_set_mesh_table(ctx: *mut MipContext, meshes: *const MipMesh, m: u32) -> i32;
    pub fn mip_set_instances(ctx: *mut MipContext, pos_xyz: *const f32, rot_ijkw: *const f32, scale: *const f32,
                             mesh_id: *const u32, n: u32) -> i32;
    pub fn mip_update_instances(ctx: *mut MipContext, first: u32, count: u32, pos_xyz: *const f32, rot_ijkw: *const f32,
                                scale: *const f32, mesh_id: *const u32) -> i32;
    pub fn mip_set_geometry(ctx: *mut MipContext, vertex_xyz: *const f32, n_vertices: u32, indices: *const u32,
                            n_indices: u32) -> i32;
    pub fn mip_set_blas_addresses(ctx: *mut MipContext, addresses: *const u64, m: u32) -> i32;
    pub fn mip_run(ctx: *mut MipContext, frame: *const MipFrame, out: *const MipOutputs) -> i32;
    /// Step k runs frames[k % n_frames] into outputs[k % n_outputs]; recorded launch graphs do not bake the frame.
    pub fn mip_run_many(ctx: *mut MipContext, frames: *const MipFrame, n_frames: u32, outputs: *const MipOutputs,
                        n_outputs: u32, steps: u32) -> i32;
    pub fn mip_wait(ctx: *mut MipContext) -> i32;
    pub fn mip_merge_draw_lists(ctx: *mut MipContext, chunks: *const c_void, n_chunks: u32, chunk_stride_bytes: u64,
                                chunk_capacity: u32, out_cmds: *mut c_void, out_count: *mut u32, async_: i32) -> i32;
    /// The same merge over chunks in the wire form (MIP_OUT_WIRE: 8-byte records, expanded against the mesh table).
    pub fn mip_merge_wire_lists(ctx: *mut MipContext, chunks: *const c_void, n_chunks: u32, chunk_stride_bytes: u64,
                                chunk_capacity: u32, out_cmds: *mut c_void, out_count: *mut u32, async_: i32) -> i32;
    /// ... and in the packed wire form (MIP_OUT_WIRE_PACKED: one 32-bit record per command).
    pub fn mip_merge_wire_lists_packed(ctx: *mut MipContext, chunks: *const c_void, n_chunks: u32, chunk_stride_bytes: u64,
                                       chunk_capacity: u32, out_cmds: *mut c_void, out_count: *mut u32, async_: i32) -> i32;
    /// Bits of a packed record left for the instance index by a table of `n_meshes` entries.
    pub fn mip_wire_index_bits(n_meshes: u32) -> u32;
    /// Shadow pass (shadow_mapping.rs:405-478): n_lights x n commands, light-major, into device memory.
    pub fn mip_light_draw_lists(ctx: *mut MipContext, light_pos_xyz: *const f32, n_lights: u32, first_instance_base: u32,
                                out_cmds: *mut c_void, async_: i32) -> i32;
    /// Extension (not a reference behaviour): skinned instances, see the header.
    pub fn mip_set_skeleton(ctx: *mut MipContext, parent: *const i32, inverse_bind: *const f32, joint_box: *const f32,
                            n_joints: u32) -> i32;
    pub fn mip_set_poses(ctx: *mut MipContext, joint_trs: *const c_void, n: u32, device: i32) -> i32;
    pub fn mip_run_skinned(ctx: *mut MipContext, frame: *const MipFrame, out: *const MipOutputs, palette: *mut c_void) -> i32;
    /// Up to 4 culled views (per-light lists, cascades) of the resident instances in one launch.
    pub fn mip_run_views(ctx: *mut MipContext, frames: *const MipFrame, outs: *const MipOutputs, n_views: u32) -> i32;
    pub fn mip_comm_unique_id(out_id: *mut u8) -> i32;
    pub fn mip_comm_init(ctx: *mut MipContext, id: *const u8, rank: u32, world: u32) -> i32;
    pub fn mip_comm_destroy(ctx: *mut MipContext) -> i32;
    pub fn mip_run_sharded(ctx: *mut MipContext, frame: *const MipFrame, out: *const MipShardedOutputs) -> i32;
    /// Row f-2: map an fd exported with vkGetMemoryFdKHR (OPAQUE_FD; a dma-buf on amdgpu) into the HIP device.
    pub fn mip_import_external_fd(ctx: *mut MipContext, fd: i32, size_bytes: u64, out_device_ptr: *mut *mut c_void) -> i32;
    pub fn mip_release_external(ctx: *mut MipContext, device_ptr: *mut c_void) -> i32;
    /// Row f-2, the semaphore half: a timeline semaphore exported with vkGetSemaphoreFdKHR (renderer.rs:3757-3861).
    pub fn mip_import_external_semaphore_fd(ctx: *mut MipContext, fd: i32, kind: u32, out_semaphore: *mut *mut MipExternalSemaphore) -> i32;
    /// 1: the HIP runtime imported it (device-side waits); 0: the DRM sync object path (host functions on the stream).
    pub fn mip_external_semaphore_on_device(ctx: *mut MipContext, semaphore: *mut MipExternalSemaphore) -> i32;
    /// The next frame's stream waits on the device until the semaphore reaches `value`.
    pub fn mip_wait_external(ctx: *mut MipContext, semaphore: *mut MipExternalSemaphore, value: u64) -> i32;
    /// Signals `value` behind the frame issued last.
    pub fn mip_signal_external(ctx: *mut MipContext, semaphore: *mut MipExternalSemaphore, value: u64) -> i32;
    pub fn mip_release_external_semaphore(ctx: *mut MipContext, semaphore: *mut MipExternalSemaphore) -> i32;
    pub fn mip_last_error(ctx: *const MipContext) -> *const c_char;
    pub fn mip_instance_count(ctx: *const MipContext) -> u32;
}

// Layout guards in the style of src/renderer.rs:178-185 (the C side is checked by tests/test_abi.py)
const _: () = assert!(std::mem::size_of::<MipConfig>() == 32);
const _: () = assert!(std::mem::size_of::<MipMesh>() == 80);
const _: () = assert!(std::mem::size_of::<MipFrame>() == 180);
const _: () = assert!(std::mem::size_of::<MipOutputs>() == 80);
const _: () = assert!(std::mem::size_of::<MipDrawIndexedIndirectCommand>() == 20);

impl Pipeline {
    /// `panic = "abort"` (Cargo.toml:133,138) makes a panic here as final as in the rest of the renderer.
    pub fn new(max_instances: u32, max_meshes: u32, frames_in_flight: u32) -> Pipeline {
        let cfg = MipConfig { struct_size: std::mem::size_of::<MipConfig>() as u32, device_ordinal: 0, max_instances,
                              max_meshes, flags: 0, frames_in_flight, stream: std::ptr::null_mut() };
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { mip_create(&cfg, &mut ctx) };
        assert_eq!(rc, MIP_OK, "mip_create failed: {}", rc);
        Pipeline(ctx)
    }
}

impl Drop for Pipeline {
    fn drop(&mut self) {
        unsafe { mip_destroy(self.0) }
    }
}
