//! src/renderer/systems/instance_pipeline.rs — the system that replaces the bodies of
//! model_matrix_calculation (src/ecs.rs:52-64), aabb_calculation (src/ecs.rs:138-181),
//! coarse_culling (cull_pipeline.rs:99-120) and the per-instance dispatch loop of cull_pass
//! (cull_pipeline.rs:534-577) + compact_draw_stream.comp. Register it where
//! model_matrix_calculation is registered today (RenderSetup, src/main.rs:785-807), after
//! assign_draw_index; drop the three CPU systems. The tested C++ twin is renderer_amd/host/ecs.cpp.
use bevy_ecs::prelude::*;

use crate::{
    ecs::{
        components::{ModelMatrix, Position, Rotation, Scale, AABB},
        resources::Camera,
    },
    renderer::{CoarseCulled, DrawIndex, GltfMesh},
};

pub(crate) struct InstancePipeline {
    pub(crate) ctx: mip_sys::Pipeline,
    /// GltfMesh.vertex_buffer handle -> row of the mesh table (rebuilt when consolidate_mesh_buffers changes)
    pub(crate) mesh_rows: hashbrown::HashMap<u64, u32>,
    pub(crate) scratch: Scratch,
}

#[derive(Default)]
pub(crate) struct Scratch {
    pos: Vec<f32>,
    rot: Vec<f32>,
    scale: Vec<f32>,
    mesh: Vec<u32>,
    model: Vec<f32>,
    aabb: Vec<f32>,
    bitmap: Vec<u32>,
    pub(crate) cmds: Vec<mip_sys::MipDrawIndexedIndirectCommand>,
    pub(crate) count: u32,
    uploaded: bool,
}

pub(crate) fn instance_pipeline(
    mut pipeline: ResMut<InstancePipeline>,
    camera: Res<Camera>,
    changed: Query<(), Or<(Changed<Position>, Changed<Rotation>, Changed<Scale>, Changed<GltfMesh>)>>,
    mut query: Query<(&DrawIndex, &Position, &Rotation, &Scale, &GltfMesh, &mut ModelMatrix, &mut AABB, &mut CoarseCulled)>,
) {
    let InstancePipeline { ctx, mesh_rows, scratch } = &mut *pipeline;
    let n = query.iter().count();
    // 1. gather the columns in draw_index order (assign_draw_index has just numbered the entities)
    if !scratch.uploaded || changed.iter().next().is_some() {
        scratch.pos.resize(n * 3, 0.0);
        scratch.rot.resize(n * 4, 0.0);
        scratch.scale.resize(n, 0.0);
        scratch.mesh.resize(n, 0);
        for (draw_index, pos, rot, scale, mesh, ..) in query.iter() {
            let i = draw_index.0 as usize;
            scratch.pos[i * 3..i * 3 + 3].copy_from_slice(pos.0.coords.as_slice());
            scratch.rot[i * 4..i * 4 + 4].copy_from_slice(rot.0.coords.as_slice()); // [i, j, k, w]
            scratch.scale[i] = scale.0;
            scratch.mesh[i] = mesh_rows[&mesh.vertex_buffer.handle.as_raw()];
        }
        let rc = unsafe {
            mip_sys::mip_set_instances(ctx.0, scratch.pos.as_ptr(), scratch.rot.as_ptr(), scratch.scale.as_ptr(),
                                       scratch.mesh.as_ptr(), n as u32)
        };
        assert_eq!(rc, mip_sys::MIP_OK);
        scratch.uploaded = true;
    }
    // 2. one fused launch per frame
    let mut frame: mip_sys::MipFrame = unsafe { std::mem::zeroed() };
    for (p, plane) in camera.frustum_planes.iter().enumerate() {
        frame.planes[p * 4..p * 4 + 4].copy_from_slice(plane.as_slice());
    }
    frame.cam_pos.copy_from_slice(camera.position.coords.as_slice());
    scratch.model.resize(n * 16, 0.0);
    scratch.aabb.resize(n * 6, 0.0);
    scratch.bitmap.resize((n + 31) / 32, 0);
    scratch.cmds.resize(n.max(1), unsafe { std::mem::zeroed() });
    let out = mip_sys::MipOutputs {
        model: scratch.model.as_mut_ptr().cast(),
        visible_bitmap: scratch.bitmap.as_mut_ptr(),
        draw_cmds: scratch.cmds.as_mut_ptr().cast(),
        draw_count: &mut scratch.count,
        draw_index_total: std::ptr::null_mut(),
        world_aabb: scratch.aabb.as_mut_ptr().cast(),
        flags: mip_sys::MIP_OUT_HOST,
        reserved: 0,
        culled_index_buffer: std::ptr::null_mut(),
        culled_index_capacity: 0,
        tlas_instances: std::ptr::null_mut(),
    };
    let rc = unsafe { mip_sys::mip_run(ctx.0, &frame, &out) };
    assert_eq!(rc, mip_sys::MIP_OK);
    // 3. scatter back what other systems still read (TLAS build, debug AABB pass, cull_pass)
    for (draw_index, _, _, _, _, mut model_matrix, mut aabb, mut coarse_culled) in query.iter_mut() {
        let i = draw_index.0 as usize;
        model_matrix.0.as_mut_slice().copy_from_slice(&scratch.model[i * 16..i * 16 + 16]);
        let b = &scratch.aabb[i * 6..i * 6 + 6];
        aabb.0 = ncollide3d::bounding_volume::AABB::new(na::Point3::new(b[0], b[1], b[2]), na::Point3::new(b[3], b[4], b[5]));
        coarse_culled.0 = (scratch.bitmap[i >> 5] >> (i & 31)) & 1 == 0;
    }
    // cull_pass then copies scratch.cmds[..scratch.count] and scratch.count into
    // IndirectCommandsBuffer / IndirectCommandsCount instead of recording one dispatch per instance.
}

/// The zero-copy variant: the renderer's own allocations — ModelData.model_buffer
/// (src/renderer.rs:1225-1265), IndirectCommandsBuffer / IndirectCommandsCount (cull_pipeline.rs:70-72) — exported once
/// with vkGetMemoryFdKHR and mapped into the HIP device, and the frame ordered against the Vulkan queues by the pass's own
/// timeline semaphores (src/renderer.rs:3757-3861) instead of a host wait. Built once, when the buffers exist:
pub(crate) struct ZeroCopyTargets {
    pub(crate) model: *mut std::ffi::c_void,        // mip_import_external_fd(ctx, fd_of(model_buffer), alloc_size, &mut ptr)
    pub(crate) draw_cmds: *mut std::ffi::c_void,    // ... IndirectCommandsBuffer
    pub(crate) draw_count: *mut u32,                // ... IndirectCommandsCount
    pub(crate) bitmap: *mut u32,                    // a HIP-side allocation is fine: only this system reads it back
    /// ComputeCull's timeline semaphore, exported with vkGetSemaphoreFdKHR and imported with
    /// mip_import_external_semaphore_fd(ctx, fd, MIP_SEMAPHORE_TIMELINE, &mut sem)
    pub(crate) cull_done: *mut mip_sys::MipExternalSemaphore,
    /// the semaphore of the passes that READ these buffers (depth pre-pass / main pass), imported the same way
    pub(crate) consumers_done: *mut mip_sys::MipExternalSemaphore,
}
unsafe impl Send for ZeroCopyTargets {}
unsafe impl Sync for ZeroCopyTargets {}

/// MipConfig.frames_in_flight of the context = the renderer's swapchain image count (DoubleBuffered<..>, src/renderer.rs:1225-1249)
const FRAMES_IN_FLIGHT: u32 = 2;

pub(crate) fn instance_pipeline_zero_copy(pipeline: Res<InstancePipeline>, targets: Res<ZeroCopyTargets>, camera: Res<Camera>,
                                          frame_number: Res<crate::renderer::FrameNumber>) {
    let ctx = pipeline.ctx.0;
    let mut frame: mip_sys::MipFrame = unsafe { std::mem::zeroed() };
    for (p, plane) in camera.frustum_planes.iter().enumerate() {
        frame.planes[p * 4..p * 4 + 4].copy_from_slice(plane.as_slice());
    }
    frame.cam_pos.copy_from_slice(camera.position.coords.as_slice());
    let out = mip_sys::MipOutputs {
        model: targets.model,
        visible_bitmap: targets.bitmap,
        draw_cmds: targets.draw_cmds,
        draw_count: targets.draw_count,
        draw_index_total: std::ptr::null_mut(),
        world_aabb: std::ptr::null_mut(),
        flags: mip_sys::MIP_OUT_DEVICE | mip_sys::MIP_OUT_ASYNC,
        reserved: 0,
        culled_index_buffer: std::ptr::null_mut(),
        culled_index_capacity: 0,
        tlas_instances: std::ptr::null_mut(),
    };
    let n = frame_number.0 as u64;
    unsafe {
        // the draws of the previous frame that read these buffers have finished ...
        assert_eq!(mip_sys::mip_wait_external(ctx, targets.consumers_done, n.saturating_sub(1)), mip_sys::MIP_OK);
        // ... the frame runs on the library's stream ...
        assert_eq!(mip_sys::mip_run(ctx, &frame, &out), mip_sys::MIP_OK);
        // ... and ComputeCull's semaphore reaches this frame's value when its kernels have: the graphics submit waits on it
        assert_eq!(mip_sys::mip_signal_external(ctx, targets.cull_done, n), mip_sys::MIP_OK);
    }
    // Nothing is copied and nothing waits on the host PER FRAME; cull_pass keeps its frame-graph node only to carry the semaphore.
    // The library's deferred errors (an external semaphore that expired, MIP_ERR_CAPACITY of the per-triangle stage) surface in
    // mip_wait, so a semaphore-ordered loop still calls it at a bounded cadence: every FRAMES_IN_FLIGHT-th frame, i.e. behind a
    // frame whose consumer semaphore (`consumers_done`, value n - 1 above) this thread has already seen pass — the wait then
    // returns at once. (Frames cannot stall or time out on the device any more: MipTimings.prefix_helps only counts how often a
    // tile had to compute a predecessor's aggregate itself.)
    if n % FRAMES_IN_FLIGHT as u64 == 0 {
        let rc = unsafe { mip_sys::mip_wait(ctx) };
        if rc != mip_sys::MIP_OK {
            let msg = unsafe { std::ffi::CStr::from_ptr(mip_sys::mip_last_error(ctx)) };
            panic!("instance pipeline: {} ({})", msg.to_string_lossy(), rc);  // panic = "abort" in both profiles (Cargo.toml:133,138)
        }
    }
}
