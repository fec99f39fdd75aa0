/*
 * mi_instance_pipeline.h — C ABI of the MI355X (gfx950) instance pipeline.
 *
 * One call per frame replaces, for every instance of a scene, this part of
 * farnoy/renderer (paths relative to the reference checkout):
 *
 *   src/ecs.rs:52-64       systems::model_matrix_calculation   M = T(p)·R(q)·S(s)
 *   src/ecs.rs:138-181     systems::aabb_calculation           mesh AABB -> world AABB
 *   src/renderer/systems/cull_pipeline.rs:99-120  coarse_culling   AABB vs 6 planes
 *   src/ecs.rs:117-136     systems::assign_draw_index          draw_index = array index
 *   src/renderer.rs:2266-2288  model_matrices_upload           model[draw_index] = M
 *   src/renderer/systems/cull_pipeline.rs:534-577 cull_pass    per-instance draw command
 *   src/renderer/helpers.rs:3-11  pick_lod                     LOD 1 beyond 10 units
 *   src/shaders/generate_work.comp:61-67          command header fields
 *   src/shaders/compact_draw_stream.comp:34-63    stream compaction + count
 *
 * The reference has no plugin API for this path; the boundary follows the one
 * FFI precedent in the repository, the `vma` crate (vma/src/lib.rs:31-64,
 * src/renderer/device/alloc.rs:192-226): opaque handle, plain #[repr(C)] POD
 * parameter structs, integer status returns (0 = success), out-pointers,
 * explicit create/destroy, no callbacks, nothing unwinds across the boundary.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, and is
 * implemented only by the HIP path: there is no CPU backend behind this ABI.
 * Without a usable gfx950 device mip_create fails with MIP_ERR_NO_DEVICE.
 */
#ifndef MI_INSTANCE_PIPELINE_H
#define MI_INSTANCE_PIPELINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIP_ABI_VERSION 4u

/* ---- status codes (0 = success, negative = error; like VkResult in vma) ---- */
#define MIP_OK 0
#define MIP_ERR_INVALID_ARGUMENT (-1) /* NULL pointer, bad size, mesh id out of range ... */
#define MIP_ERR_NO_DEVICE (-2)        /* no HIP device / not gfx950 / ordinal out of range */
#define MIP_ERR_OUT_OF_MEMORY (-3)    /* hipMalloc failed */
#define MIP_ERR_CAPACITY (-4)         /* more instances / meshes than the context was created for */
#define MIP_ERR_DEVICE (-5)           /* a HIP runtime call failed; see mip_last_error */
#define MIP_ERR_NOT_READY (-6)        /* run before instances / mesh table were set */
#define MIP_ERR_TIMEOUT (-7)          /* a stream-ordered wait for an EXTERNAL semaphore expired (10 s; mip_wait_external). No kernel
                                       * of this library waits for another workgroup to run — a tile whose predecessor has not
                                       * published computes that predecessor's aggregate itself (MipTimings.prefix_helps) — so
                                       * a frame cannot time out, whatever order the hardware starts workgroups in and
                                       * whatever else shares the GPU. (ABI <= 3 reported expired in-kernel waits here.) */

/* ---- MipConfig.flags ---- */
#define MIP_CFG_TIMING 0x1u /* bracket every kernel with hipEvents (mip_get_timings) */
#define MIP_CFG_ORDERED_TILES 0x2u /* accepted and ignored since ABI 4. Up to ABI 3 the default frame kernel relied on the
                                     * hardware starting a launch's workgroups in index order (not guaranteed by HIP, and seen
                                     * to fail when several processes shared one GPU) and this flag selected slower modes that
                                     * did not (tickets: 56 us at 1 M instances; three wait-free launches: 26 us, against 18.5).
                                     * Now EVERY launch is independent of the order workgroups start in: the one-hop look-up of
                                     * a tile's prefix polls a bounded number of times and then computes what is missing itself
                                     * (decoupled look-back with a fallback; renderer_amd/csrc/instance_kernel.hpp). In-order
                                     * dispatch is a performance property only. */

/* ---- MipOutputs.flags ---- */
#define MIP_OUT_HOST 0x0u   /* output pointers are host memory (copied back, synchronous) */
#define MIP_OUT_DEVICE 0x1u /* output pointers are device memory of the context's GPU. The frame runs on the
                              * context's own stream(s) (MipConfig.stream if given): work the caller queued on
                              * OTHER streams for those buffers (a clear, a previous reader) is not waited for —
                              * order it with an event / synchronize, or hand the library that stream */
#define MIP_OUT_ASYNC 0x2u  /* with MIP_OUT_DEVICE: return after enqueue; pair with mip_wait */
#define MIP_OUT_WIRE 0x4u   /* with MIP_OUT_DEVICE and a 16-BYTE ALIGNED draw_cmds: it receives the list in the WIRE form below (8.06 B per
                              * command instead of 20) — what a rank sends through the all-gather; draw_count and
                              * draw_index_total as usual. Not with culled_index_buffer (the wire form carries no
                              * indexCount: it is the mesh table's). mip_merge_wire_lists expands it again. */
#define MIP_OUT_WIRE_PACKED 0x8u /* with MIP_OUT_WIRE: the PACKED wire form below, 4.25 B per command — one 32-bit record
                              * {instance index in the frame | mesh_id << index_bits | lod << 31}. Only while the
                              * context's instance count fits: n <= 1 << mip_wire_index_bits(n_meshes), else
                              * MIP_ERR_INVALID_ARGUMENT. mip_merge_wire_lists_packed expands it. */

/* Largest LOD chain the scene loader can produce: LOD0 + 5 simplified levels
 * (src/renderer/systems/scene_loader.rs:740-753). */
#define MIP_MAX_LODS 6u
#define MIP_MAX_FRAMES_IN_FLIGHT 8u

typedef struct MipContext MipContext; /* opaque; Send + Sync like VmaAllocator */

typedef struct MipConfig {
  uint32_t struct_size;   /* = sizeof(MipConfig); guards ABI drift */
  int32_t device_ordinal; /* HIP device index of this process' GPU */
  uint32_t max_instances; /* capacity; the reference's is 4096 (generate_work.comp:25-27) */
  uint32_t max_meshes;    /* capacity of the mesh table */
  uint32_t flags;         /* MIP_CFG_* */
  /* Frames the caller keeps in flight (0 or 1 = one): each gets its own stream and its own
   * cross-tile prefix state inside the context, and consecutive mip_run calls rotate over
   * them, so frame k+1 may start on the device while frame k drains — what the reference's
   * per-swapchain-image buffers (DoubleBuffered<..>, src/renderer.rs:1225-1249) allow. The
   * caller must give `frames_in_flight` consecutive async runs distinct output buffers.
   * Needs stream == NULL when > 1. */
  uint32_t frames_in_flight;
  void* stream; /* hipStream_t to enqueue on, or NULL for streams owned by the context */
} MipConfig;

/* One entry per distinct mesh. Stands in for GltfMesh.aabb (src/renderer.rs:125),
 * GltfMesh.index_buffers[lod].1 (index_len) and the ConsolidatedMeshBuffers
 * vertex_offsets / index_offsets lookups (cull_pipeline.rs:540-548). */
typedef struct MipMesh {
  float aabb_min[3]; /* mesh-local box, finite */
  float aabb_max[3];
  uint32_t n_lods;                     /* 1..MIP_MAX_LODS */
  uint32_t index_len[MIP_MAX_LODS];    /* indices in LOD k */
  uint32_t index_offset[MIP_MAX_LODS]; /* offset of LOD k in the consolidated index buffer */
  int32_t vertex_offset;               /* offset in the consolidated vertex buffer */
} MipMesh;

/* Byte-identical to VkDrawIndexedIndirectCommand (generate_work.comp:9-15,
 * asserted against ash's struct at src/renderer.rs:178-185). */
typedef struct MipDrawIndexedIndirectCommand {
  uint32_t indexCount;
  uint32_t instanceCount;
  uint32_t firstIndex;
  int32_t vertexOffset;
  uint32_t firstInstance;
} MipDrawIndexedIndirectCommand;

/* Per-frame inputs. Everything else is resident on the device. */
typedef struct MipFrame {
  /* Camera.frustum_planes (src/ecs/camera_controller.rs:15-16): 6 x (nx,ny,nz,d),
   * order left,right,bottom,top,near,far, outward-facing, not normalised
   * (src/ecs.rs:83-90). Produced on the host by project_camera. */
  float planes[24];
  float cam_pos[3];           /* Camera.position, for pick_lod */
  uint32_t first_instance_base; /* added to firstInstance: draw_index of instance 0 of this shard */
  uint32_t first_index_base;    /* added to firstIndex (wrapping u32) */
  /* CameraMatrices.pv = projection * view, column-major (generate_work.comp:29-34). Only read
   * when MipOutputs.culled_index_buffer is set (per-triangle culling). */
  float pv[16];
} MipFrame;

typedef struct MipOutputs {
  /* N x mat4, column-major, 64 B each: the `mat4 model[]` storage buffer
   * (ModelData.model_buffer, src/renderer.rs:1225-1249). May be NULL. */
  void* model;
  /* ceil(N/32) words; bit (i & 31) of word (i >> 5) = !CoarseCulled[i]. May be NULL. */
  uint32_t* visible_bitmap;
  /* Up to N MipDrawIndexedIndirectCommand, compacted, ascending draw_index
   * (IndirectCommandsBuffer). Entries past *draw_count are left untouched. May be NULL
   * only together with draw_count. */
  void* draw_cmds;
  uint32_t* draw_count; /* IndirectCommandsCount.count */
  /* Optional: Σ indexCount over the emitted commands (wrapping u32) = the firstIndex the
   * next appended command would get, relative to first_index_base. Needed when shards
   * of one scene are merged (mip_merge_draw_lists). May be NULL. */
  uint32_t* draw_index_total;
  /* Optional: N x {mins[3], maxs[3]} world AABB as the ECS `AABB` component holds it
   * (src/ecs/components.rs:18-20). May be NULL. Identical to the reference's values AS NUMBERS; the sign
   * of a coordinate that is exactly zero is not specified (the kernel folds the box without enumerating
   * the corners where that is exact — a zero may come out as +0 where the corner loop gives -0 — and which
   * arithmetic tier runs depends on the other instances of the scene). Visibility does not depend on it. */
  void* world_aabb;
  uint32_t flags; /* MIP_OUT_* */
  uint32_t reserved;
  /* Optional (needs MIP_OUT_DEVICE, mip_set_geometry, model and draw_cmds): the culled index
   * stream `uvec3 out_index_buffer[]` (CulledIndexBuffer, generate_work.comp:40-42). When set,
   * every emitted command's triangles go through the per-triangle back-face + x/y frustum test
   * of generate_work.comp:68-200; survivors are appended, in mesh order, at
   * culled_index_buffer[firstIndex ...]; indexCount becomes 3 x survivors and commands without
   * survivors are dropped by the compaction that follows (compact_draw_stream.comp runs after
   * generate_work). firstIndex keeps the reference's layout: the running sum of the full
   * index_len of the earlier commands. */
  void* culled_index_buffer;
  uint64_t culled_index_capacity; /* in indices (u32); a command that would not fit raises MIP_ERR_CAPACITY */
  /* Optional (needs MIP_OUT_DEVICE): N x VkAccelerationStructureInstanceKHR (64 B), one per
   * instance in draw_index order, as build_acceleration_structures fills them
   * (src/renderer/systems/acceleration_strucures.rs:419-451): transform = rows 0..2 of the model
   * matrix (row-major 3x4), instanceCustomIndex = draw_index, mask = 0xFF, sbt offset 0,
   * flags = TRIANGLE_FACING_CULL_DISABLE, accelerationStructureReference = the mesh's BLAS
   * address (mip_set_blas_addresses; 0 if never set). */
  void* tlas_instances;
} MipOutputs;

typedef struct MipTimings {
  uint64_t runs;              /* mip_run calls timed since create / last reset */
  double last_kernel_ms;      /* hipEvent time of the pipeline kernel of the last run */
  double total_kernel_ms;     /* sum over `runs` */
  double last_merge_ms;       /* same for mip_merge_draw_lists */
  double total_merge_ms;
  uint64_t merges;
  uint64_t graph_frames;      /* frames mip_run_many replayed from recorded launch graphs (counted even without MIP_CFG_TIMING) */
  uint64_t graph_records;     /* times it had to record a new set of graphs */
  uint64_t sharded_retries;   /* sharded frames whose tightened chunk overflowed and were re-gathered at full capacity */
  uint64_t sharded_bytes_sent; /* bytes this rank contributed to the last sharded frame's all-gather */
  uint64_t prefix_helps;      /* tile aggregates a WAITING tile computed itself because their owner had not published within the
                                 patient polls (see MIP_ERR_TIMEOUT): 0 on a GPU this context has to itself; non-zero means the
                                 hardware started workgroups out of order or another tenant held compute units — results are
                                 the same either way. Cumulative since create / mip_reset_timings */
  uint64_t general_launches;  /* frames launched with the kernel that carries the literal path for non-finite
                                 inputs (some resident instance failed the upload-time finite test, or a skinned frame) */
  uint64_t reserved0;         /* (ABI 3: timeout_recoveries) */
} MipTimings;

/* Chunk header used by mip_merge_draw_lists: what each rank contributes to the
 * all-gather in front of its commands. 32 B so the commands stay 16-B aligned. */
typedef struct MipShardHeader {
  uint32_t draw_count;
  uint32_t draw_index_total;
  uint32_t reserved[6];
} MipShardHeader;

/* ---- wire form of a shard's draw list (MIP_OUT_WIRE, mip_merge_wire_lists) ---------------------
 * An emitted command of cull_pass (cull_pipeline.rs:534-577) is determined by its draw_index, its
 * mesh, the LOD picked and its position in the running index sum: indexCount = index_len[lod] and
 * vertexOffset come from the mesh table every rank holds, instanceCount is 1 (generate_work.comp:63).
 * The wire form therefore carries per command the 8-byte record
 *     { firstInstance, mesh_id | lod << 31 }            lod = 0 or 1 (pick_lod, helpers.rs:3-11)
 * in blocks of MIP_WIRE_BLOCK_COMMANDS records, each block behind a 16-byte block header of FOUR
 * words: word q is the firstIndex of the block's record MIP_WIRE_SUB_BLOCK_COMMANDS * q (relative to
 * the shard, plus the frame's first_index_base, as the 20-byte form has it; a word whose record does
 * not exist is unspecified). The firstIndex of the other records is their anchor plus the index_len of
 * the records between the anchor and them. (ABI <= 3 anchored only record 0 of a block; 64-record
 * sub-blocks let one wave of the merge kernel expand its share without talking to the others.)
 * Block b of a list lives at byte b * MIP_WIRE_BLOCK_BYTES of the body; a list cut after any whole
 * number of blocks is a valid shorter list, which is what lets a rank send a tightened slice of it. */
#define MIP_WIRE_BLOCK_COMMANDS 256u
#define MIP_WIRE_SUB_BLOCK_COMMANDS 64u
#define MIP_WIRE_BLOCK_HEADER_BYTES 16u
#define MIP_WIRE_RECORD_BYTES 8u
#define MIP_WIRE_BLOCK_BYTES (MIP_WIRE_BLOCK_HEADER_BYTES + MIP_WIRE_BLOCK_COMMANDS * MIP_WIRE_RECORD_BYTES) /* 2064 */
/* bytes of the body of a wire list with room for `capacity` commands (whole blocks) */
#define MIP_WIRE_BODY_BYTES(capacity) \
  ((((uint64_t)(capacity) + MIP_WIRE_BLOCK_COMMANDS - 1u) / MIP_WIRE_BLOCK_COMMANDS) * MIP_WIRE_BLOCK_BYTES)
/* The PACKED wire form (MIP_OUT_WIRE | MIP_OUT_WIRE_PACKED): ONE 32-bit record per command,
 *     instance_index | mesh_id << index_bits | lod << 31,      instance_index = firstInstance - first_instance_base,
 * in blocks of MIP_WIRE_PACKED_BLOCK_COMMANDS = 64 records, each behind a self-describing 16-byte header
 * {firstIndex of the block's first command, the frame's first_instance_base, index_bits, 0}: 4.25 B per command.
 * index_bits = mip_wire_index_bits(n_meshes) = 31 - ceil(log2(n_meshes)) is what the mesh ids leave of the word, so the
 * form exists for frames of at most 1 << index_bits instances (64 meshes: 33 M; 1 024 meshes: 2 M) — every rank of an
 * exchange derives the same answer from the replicated mesh table and the largest shard. */
#define MIP_WIRE_PACKED_RECORD_BYTES 4u
#define MIP_WIRE_PACKED_BLOCK_COMMANDS 64u
#define MIP_WIRE_PACKED_BLOCK_BYTES (MIP_WIRE_BLOCK_HEADER_BYTES + MIP_WIRE_PACKED_BLOCK_COMMANDS * MIP_WIRE_PACKED_RECORD_BYTES) /* 272 */
#define MIP_WIRE_PACKED_BODY_BYTES(capacity) \
  ((((uint64_t)(capacity) + MIP_WIRE_PACKED_BLOCK_COMMANDS - 1u) / MIP_WIRE_PACKED_BLOCK_COMMANDS) * MIP_WIRE_PACKED_BLOCK_BYTES)
/* bits of a packed record left for the instance index by a mesh table of n_meshes entries (pure function) */
uint32_t mip_wire_index_bits(uint32_t n_meshes);

uint32_t mip_abi_version(void);

/* Create a context on cfg->device_ordinal. Allocates device storage for
 * max_instances / max_meshes. Returns MIP_OK and writes *out, or a negative code
 * (then *out = NULL). Never aborts. */
int32_t mip_create(const MipConfig* cfg, MipContext** out);

/* Frees everything the context owns. NULL is a no-op. */
void mip_destroy(MipContext* ctx);

/* Copy the mesh table to the device (caller keeps its memory). m <= max_meshes.
 * Bounds must be finite and n_lods in 1..MIP_MAX_LODS. If the table is smaller than the one it replaces and a
 * resident instance names a mesh outside it, the instances stop being resident (upload the new scene's
 * next; a frame before that fails with MIP_ERR_NOT_READY): the kernels never gather outside the table. */
int32_t mip_set_mesh_table(MipContext* ctx, const MipMesh* meshes, uint32_t m);

/* Upload the instance columns: SoA, tightly packed, draw_index = array index
 * (assign_draw_index for a single-archetype static scene, src/ecs.rs:117-136).
 *   pos_xyz  n x 3 floats   Position(Point3<f32>)
 *   rot_ijkw n x 4 floats   Rotation(UnitQuaternion<f32>), stored [i,j,k,w]; NOT renormalised
 *   scale    n floats       Scale(f32)
 *   mesh_id  n u32          index into the mesh table; every id must be < m
 * Host pointers; copied. Static scenes call this once. n may be 0. */
int32_t mip_set_instances(MipContext* ctx, const float* pos_xyz, const float* rot_ijkw,
                          const float* scale, const uint32_t* mesh_id, uint32_t n);

/* Overwrite a range [first, first + count) of the resident columns (moving entities: the
 * Changed<Position|Rotation|Scale> filter of a bevy query). A NULL column is left as it is.
 * The instance count does not change; first + count must not exceed it. Host pointers; copied
 * after everything in flight has drained. */
int32_t mip_update_instances(MipContext* ctx, uint32_t first, uint32_t count, const float* pos_xyz,
                             const float* rot_ijkw, const float* scale, const uint32_t* mesh_id);

/* Same, from DEVICE pointers of the context's GPU (device-to-device copies). Mesh ids are checked on the
 * device after the copy (the upload-time census reads every instance anyway): an id >= m fails the call with
 * MIP_ERR_INVALID_ARGUMENT and leaves NO instances resident. */
int32_t mip_set_instances_device(MipContext* ctx, const void* pos_xyz, const void* rot_ijkw,
                                 const void* scale, const void* mesh_id, uint32_t n);

/* Per-mesh bottom-level acceleration structure device addresses for MipOutputs.tlas_instances
 * (vkGetAccelerationStructureDeviceAddressKHR per GltfMesh, acceleration_strucures.rs:430-437).
 * m must equal the mesh table's size. Host pointer; copied. */
int32_t mip_set_blas_addresses(MipContext* ctx, const uint64_t* addresses, uint32_t m);

/* Upload the consolidated geometry the per-triangle stage reads (ConsolidatedMeshBuffers'
 * position_buffer and index_buffer, consolidate_mesh_buffers.rs): packed vec3 positions and
 * u32 indices; MipMesh.vertex_offset / index_offset[] index into them. Host pointers; copied. */
int32_t mip_set_geometry(MipContext* ctx, const float* vertex_xyz, uint32_t n_vertices,
                         const uint32_t* indices, uint32_t n_indices);


/* One frame: model matrices, world AABBs, visibility, compacted draw commands.
 * Call from one thread at a time per context. Synchronous on return unless
 * MIP_OUT_ASYNC. */
int32_t mip_run(MipContext* ctx, const MipFrame* frame, const MipOutputs* out);

/* Enqueue `steps` frames back to back from compiled code — the renderer's 'frame: loop
 * (src/main.rs:907-926), in which project_camera (src/ecs.rs:66-91) produces new planes every frame:
 * step k runs frames[k % n_frames] into outputs[k % n_outputs] (give at least frames_in_flight
 * output sets). Every output set must carry MIP_OUT_DEVICE | MIP_OUT_ASYNC. Equivalent to calling
 * mip_run `steps` times; exists so that a host in a scripting language does not pay its per-call
 * overhead per frame, and so that the launches can be recorded once and replayed: when every
 * output set asks for draw commands and none for the per-triangle stage, whole rounds of ~64 frames
 * go out as one hipGraph per frame slot. The graphs are recorded on first use and cached by the
 * OUTPUT sets only — a frame's planes, camera position and bases are not baked into them (each
 * recorded launch reads its frame from a small device-side ring that one copy refreshes per
 * replay), so a moving camera replays the same graphs (MipTimings.graph_records stays put,
 * graph_frames counts the replayed frames); the remainder and every other case are plain mip_run
 * calls. */
int32_t mip_run_many(MipContext* ctx, const MipFrame* frames, uint32_t n_frames, const MipOutputs* outputs,
                     uint32_t n_outputs, uint32_t steps);

/* Row f-4, second consumer — the shadow pass's per-light draw lists
 * (src/renderer/systems/shadow_mapping.rs:405-478: for every light, for every mesh entity,
 * pick_lod(index_buffers, light_position, mesh_position) and cmd_draw_indexed(index_count, 1,
 * 0, 0, draw_index); nothing is culled). For light l and resident instance i
 *   out_cmds[l*n + i] = { index_len[lod], 1, index_offset[lod], vertex_offset,
 *                         first_instance_base + i }
 * over the consolidated buffers (as cull_pass addresses them, cull_pipeline.rs:540-553), so the
 * shadow pass becomes one vkCmdDrawIndexedIndirect(buffer, l*n*20, n, 20) per light.
 * light_pos_xyz: n_lights x 3 host floats, 1 <= n_lights <= MIP_MAX_LIGHTS (the 4x4 shadow atlas,
 * shadow_mapping.rs:24). out_cmds: DEVICE pointer, n_lights * n * 20 bytes. async != 0: returns
 * after enqueueing on the context's stream (mip_wait to finish). */
#define MIP_MAX_LIGHTS 16
int32_t mip_light_draw_lists(MipContext* ctx, const float* light_pos_xyz, uint32_t n_lights,
                             uint32_t first_instance_base, void* out_cmds, int32_t async);

/* ---- Extension: skinned instances (BASELINE config 5) ----------------------------------------
 * NOT a reference behaviour: farnoy/renderer has no skins, joints or animation (SURVEY.md
 * section 8, top table). Specified from glTF 2.0 section 3.7.3 and checked against this repository's
 * oracle only. One skeleton per context, shared by every instance:
 *   parent[k]        index of the parent joint, < k, or -1 (parents precede children)
 *   inverse_bind     n_joints x 16 floats, column-major mat4 (skin.inverseBindMatrices; rows 0..2 used)
 *   joint_box        n_joints x 6 floats: min xyz, max xyz of the bind-pose vertices weighted to
 *                    joint k, in mesh space (min > max: the joint binds no vertex)
 * 1 <= n_joints <= MIP_MAX_JOINTS. Host pointers; copied. */
#define MIP_MAX_JOINTS 32
#define MIP_POSE_FLOATS 10 /* per joint: translation xyz, rotation quaternion ijkw, scale xyz (the LOCAL TRS) */
int32_t mip_set_skeleton(MipContext* ctx, const int32_t* parent, const float* inverse_bind,
                         const float* joint_box, uint32_t n_joints);

/* The animated pose of every instance: n x n_joints x MIP_POSE_FLOATS floats, instance-major.
 * n must equal the resident instance count. device == 0: host pointer, copied (waits for the frames in
 * flight). device != 0: a DEVICE pointer (8-byte aligned) that is borrowed, not copied; the call does not
 * wait for anything — frames already queued keep the pointer they were launched with, so an animation
 * system can alternate two buffers with frames_in_flight = 2. It keeps each buffer alive and unmodified
 * while a frame that reads it is in flight. */
int32_t mip_set_poses(MipContext* ctx, const void* joint_trs, uint32_t n, int32_t device);

/* One frame of skinned instances. Per instance and joint
 *   L_k = T*R*S of the pose, G_k = G_parent * L_k, J_k = G_k * inverse_bind_k   (affine, fp32)
 * the palette (n x n_joints mat4, column-major, DEVICE pointer, may be NULL) receives J_k. The
 * union over joints of J_k * joint_box_k (8 corners each) is the instance's posed box in mesh
 * space; it takes the place of the mesh table's aabb for that instance, and everything else —
 * model[], world box, frustum test, bitmap, draw commands, count, TLAS rows — is produced from
 * it exactly as mip_run does from GltfMesh.aabb. out must carry MIP_OUT_DEVICE;
 * culled_index_buffer is not supported (the per-triangle stage does not skin vertices). */
int32_t mip_run_skinned(MipContext* ctx, const MipFrame* frame, const MipOutputs* out, void* palette);

/* Several views of the resident instances in ONE launch — per-light culled draw lists (the shadow pass
 * with a frustum per light), cascades, cube faces, stereo. View v is a complete cull_pass with
 * frames[v]'s planes, LOD reference point (cam_pos) and bases, and gives exactly what mip_run would give
 * for that frame: outs[v].visible_bitmap (optional), outs[v].draw_cmds + draw_count (required),
 * outs[v].draw_index_total (optional). The instance data is read once and the model matrix / world box
 * built once for all views; nothing view-independent is written, so model, world_aabb, tlas_instances
 * and culled_index_buffer must be NULL (run the frame's mip_run for those). 1 <= n_views <=
 * MIP_MAX_VIEWS (the 4 x 4 shadow atlas, shadow_mapping.rs:24); four views share a launch, more views
 * are further launches on the same stream. Every output set carries MIP_OUT_DEVICE; the call is
 * asynchronous if outs[0] carries MIP_OUT_ASYNC. Runs on the context's first stream. */
#define MIP_MAX_VIEWS 16
int32_t mip_run_views(MipContext* ctx, const MipFrame* frames, const MipOutputs* outs, uint32_t n_views);

/* Block until everything enqueued by this context has finished; reports a
 * deferred error of an async run (MIP_ERR_CAPACITY, MIP_ERR_DEVICE, MIP_ERR_TIMEOUT of an external semaphore).
 * Frames ordered by external semaphores still need this call at a bounded cadence (e.g. every
 * frames_in_flight frames): it is where their errors surface. */
int32_t mip_wait(MipContext* ctx);

/* Merge `n_chunks` shard draw lists (each: MipShardHeader followed by its commands,
 * chunks `chunk_stride_bytes` apart, as an all-gather lays them out; DEVICE memory) into
 * one contiguous list in shard order, adding to each shard's firstIndex the
 * draw_index_total of all earlier shards. out_count[0] = total commands,
 * out_count[1] = total indices (so out_count needs room for 2 words). DEVICE pointers. Enqueued on the
 * context's first stream (MipConfig.stream, or frame slot 0's): it is ordered after a frame of the same
 * context only when frames_in_flight == 1 — use one context per frame in flight for sharded frames, as
 * mip_run_sharded and renderer_amd/sharded.py do. Synchronous unless `async` is non-zero.
 * `chunk_capacity` = commands one chunk may carry (0 = what the stride holds): `out_cmds` needs room for
 * n_chunks x chunk_capacity commands, and a chunk whose header count exceeds it is cut there and
 * reported (MIP_ERR_CAPACITY from this call, or from mip_wait for an async one) — the stride is
 * usually rounded up and may physically hold a few commands more than the capacity. */
int32_t mip_merge_draw_lists(MipContext* ctx, const void* chunks, uint32_t n_chunks,
                             uint64_t chunk_stride_bytes, uint32_t chunk_capacity, void* out_cmds,
                             uint32_t* out_count, int32_t async);

/* The same merge for chunks in the WIRE form (each: MipShardHeader followed by a wire body, see
 * MIP_OUT_WIRE): every record is expanded against THIS context's mesh table — which must be the table
 * the emitting ranks ran with; it is replicated by construction — into the 20-byte
 * command, and the result is byte-identical to mip_merge_draw_lists over the 20-byte chunks of the
 * same frames. chunk_stride_bytes >= sizeof(MipShardHeader) + MIP_WIRE_BODY_BYTES(chunk_capacity);
 * chunk_capacity = 0 means what the stride holds in whole blocks. A record whose mesh id is outside the table (a corrupt chunk)
 * is expanded as mesh 0 and reported as MIP_ERR_DEVICE. */
int32_t mip_merge_wire_lists(MipContext* ctx, const void* chunks, uint32_t n_chunks,
                             uint64_t chunk_stride_bytes, uint32_t chunk_capacity, void* out_cmds,
                             uint32_t* out_count, int32_t async);
/* ... and for chunks in the PACKED wire form (MIP_OUT_WIRE_PACKED; chunk_stride_bytes >= sizeof(MipShardHeader) +
 * MIP_WIRE_PACKED_BODY_BYTES(chunk_capacity)). Each block header carries its own index_bits; one that cannot be (> 31)
 * is a corrupt chunk and reported like a bad mesh id. */
int32_t mip_merge_wire_lists_packed(MipContext* ctx, const void* chunks, uint32_t n_chunks,
                                    uint64_t chunk_stride_bytes, uint32_t chunk_capacity, void* out_cmds,
                                    uint32_t* out_count, int32_t async);

/* ---- sharded scenes without a Python host: RCCL straight from the library ------------------
 * librccl.so.1 is opened with dlopen on first use, so single-GPU hosts do not need it. The
 * exchange is the one of SURVEY.md §8e: every rank runs its shard, ONE ncclAllGather moves the
 * fixed-size chunks [MipShardHeader | wire body for chunk_capacity commands: 4.25 B each (packed form; 8.06 B when the
 * largest shard does not fit a packed record) instead of 20, see MIP_OUT_WIRE], the merge kernel expands and concatenates them. Needs a context with one frame in flight. */
#define MIP_COMM_ID_BYTES 128u

/* ncclGetUniqueId: call on one rank, hand the 128 bytes to the others by any means. */
int32_t mip_comm_unique_id(uint8_t out_id[MIP_COMM_ID_BYTES]);
/* ncclCommInitRank on the context's device: collective over all `world` ranks. */
int32_t mip_comm_init(MipContext* ctx, const uint8_t id[MIP_COMM_ID_BYTES], uint32_t rank, uint32_t world);
int32_t mip_comm_destroy(MipContext* ctx);

typedef struct MipShardedOutputs {
  void* model;              /* this rank's shard: n_local x mat4, or NULL */
  uint32_t* visible_bitmap; /* this rank's shard, or NULL */
  void* world_aabb;         /* this rank's shard, or NULL */
  void* draw_cmds;          /* the MERGED global list; room for world x chunk_capacity commands */
  uint32_t* draw_count;     /* [0] merged command count, [1] merged index total */
  uint32_t chunk_capacity;  /* commands each rank contributes at most; 0 = the largest max_instances over the ranks
                               (mip_comm_init settles on it with a 4-byte all-gather, so ranks created for
                               shards of different sizes still exchange chunks of one size).
                               If ANY rank emits more, every rank sees it in the gathered headers and the
                               library repeats the all-gather + merge of that frame once at full capacity
                               (at once for a synchronous call, inside mip_wait for an asynchronous one;
                               MipTimings.sharded_retries counts them): draw_cmds therefore needs room for
                               world x (largest max_instances) commands whenever chunk_capacity is tightened */
  uint32_t flags;           /* MIP_OUT_DEVICE, optionally | MIP_OUT_ASYNC */
} MipShardedOutputs;

/* One frame of a sharded scene on this rank (collective: every rank calls it with the same
 * chunk_capacity). frame->first_instance_base must be the shard's first draw_index. With
 * MIP_OUT_ASYNC and a tightened chunk_capacity call mip_wait before the next sharded frame: the
 * repair of an overflowing frame re-sends this rank's list, which the next frame overwrites
 * (an overflow that can no longer be repaired is reported as MIP_ERR_CAPACITY). */
int32_t mip_run_sharded(MipContext* ctx, const MipFrame* frame, const MipShardedOutputs* out);

/* ---- zero-copy interop with the renderer's own allocations ----------------
 * The reference keeps the buffers this path fills in VMA allocations of its Vulkan device:
 * ModelData.model_buffer (src/renderer.rs:1225-1265), IndirectCommandsBuffer / IndirectCommandsCount
 * (src/renderer/systems/cull_pipeline.rs:70-72,183-220), declared GPU_ONLY by the buffer macro
 * (src/renderer/macros/macros.rs:67-79) on an allocator created without exportable handle types
 * (src/renderer/device/alloc.rs:154-171). Once such a buffer is allocated from a memory block created
 * with VkExportMemoryAllocateInfo{handleTypes = VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT} (on amdgpu
 * the fd is a dma-buf; INTEGRATION.md §3 lists the VMA / Vulkan flags) and exported with
 * vkGetMemoryFdKHR, this call maps the same bytes into the context's HIP device:
 * hipImportExternalMemory(OpaqueFd) + hipExternalMemoryGetMappedBuffer. `*out_device_ptr` is then a
 * valid MIP_OUT_DEVICE output pointer (or mip_set_instances_device input) for `size_bytes` bytes.
 * As with cudaImportExternalMemory, a successfully imported fd belongs to the driver: do not use or
 * close it afterwards. Ordering against the Vulkan queue: the semaphore entry points below, or
 * vkQueueWaitIdle / mip_wait at the hand-over points. */
int32_t mip_import_external_fd(MipContext* ctx, int32_t fd, uint64_t size_bytes, void** out_device_ptr);
/* Unmaps a pointer returned by mip_import_external_fd (after mip_wait); mip_destroy releases the rest. */
int32_t mip_release_external(MipContext* ctx, void* device_ptr);

/* ---- the semaphore half of the same interop -------------------------------------------------------
 * The reference orders its passes with TIMELINE semaphores, one per frame-graph pass, signalled and
 * waited with a value derived from the frame number (src/renderer.rs:3757-3861, AutoSemaphores;
 * `ComputeCull` is the pass this library replaces and the graphics submit waits for it). A semaphore
 * created with VkExportSemaphoreCreateInfo{handleTypes = OPAQUE_FD_BIT} (+ VkSemaphoreTypeCreateInfo
 * {TIMELINE}) and exported with vkGetSemaphoreFdKHR is imported here with hipImportExternalSemaphore;
 * the library then takes the place of the ComputeCull submit:
 *
 *   mip_wait_external(ctx, sem_prev, value)   the stream the NEXT frame will run on waits, on the device,
 *                                             until the semaphore reaches `value` (the reader of the
 *                                             buffers this frame overwrites has finished)
 *   mip_run(ctx, frame, outs | MIP_OUT_ASYNC)
 *   mip_signal_external(ctx, sem_cull, value) enqueued behind the frame that was issued LAST: the
 *                                             semaphore reaches `value` when its kernels have finished
 *
 * and the graphics submit lists sem_cull/value as a wait semaphore: no host wait per frame.
 * kind: MIP_SEMAPHORE_TIMELINE (what the reference uses) or MIP_SEMAPHORE_BINARY (`value` ignored).
 * As with memory, a successfully imported fd belongs to the library. Two implementations sit behind the
 * handle: the HIP runtime's (waits and signals execute on the device) when it accepts the handle type, and
 * otherwise — ROCm 7.2 on Linux refuses both: TimelineSemaphoreFd "invalid argument", OpaqueFd "operation
 * not supported" — the kernel object itself: on amdgpu the exported fd IS a DRM sync object, which the
 * library imports on a render node (DRM_IOCTL_SYNCOBJ_FD_TO_HANDLE) and waits for / signals from host
 * functions enqueued on the frame's stream (hipLaunchHostFunc: stream-ordered, no wait on the caller's
 * thread; a wait is bounded at 10 s and then reported by mip_wait as MIP_ERR_TIMEOUT).
 * mip_external_semaphore_on_device tells which one a handle got (1 = HIP runtime, 0 = host functions).
 * Errors: MIP_ERR_INVALID_ARGUMENT for a bad fd / kind / a handle this context did not import;
 * MIP_ERR_DEVICE with the runtime's message when neither path accepts the fd. */
#define MIP_SEMAPHORE_BINARY 0u
#define MIP_SEMAPHORE_TIMELINE 1u
typedef struct MipExternalSemaphore MipExternalSemaphore; /* opaque */
int32_t mip_import_external_semaphore_fd(MipContext* ctx, int32_t fd, uint32_t kind, MipExternalSemaphore** out_semaphore);
int32_t mip_external_semaphore_on_device(MipContext* ctx, MipExternalSemaphore* semaphore);
int32_t mip_wait_external(MipContext* ctx, MipExternalSemaphore* semaphore, uint64_t value);
int32_t mip_signal_external(MipContext* ctx, MipExternalSemaphore* semaphore, uint64_t value);
/* After mip_wait; mip_destroy releases the rest. */
int32_t mip_release_external_semaphore(MipContext* ctx, MipExternalSemaphore* semaphore);

const char* mip_last_error(const MipContext* ctx);
int32_t mip_get_timings(MipContext* ctx, MipTimings* out);  /* touches the device: a blocking 4-byte read of the help counter */
int32_t mip_reset_timings(MipContext* ctx);                /* (the same read: prefix_helps counts from here on, also with frames in flight) */

/* Number of instances currently resident (set by mip_set_instances*). */
uint32_t mip_instance_count(const MipContext* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MI_INSTANCE_PIPELINE_H */
