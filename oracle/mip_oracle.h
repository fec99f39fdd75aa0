/*
 * mip_oracle.h — CPU oracle of the instance pipeline. TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, plain-C restatement of the reference's per-frame instance path
 * (farnoy/renderer; citations are paths in the reference checkout). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product library (renderer_amd/csrc) never links, includes or calls it.
 *
 * PARITY UNPINNED: the reference has no tests, fixtures or golden vectors for this
 * path, it is nightly Rust and cannot be built here, and the
 * arithmetic it calls lives in crates that are not vendored (nalgebra 0.29.0,
 * nalgebra-glm 0.15.0, ncollide3d 0.32.0 — Cargo.lock). The operation order below
 * restates those crates' published algorithms (column-axpy gemm/gemv, the 3- and
 * 4-wide dot special cases, UnitQuaternion::to_rotation_matrix,
 * AABB::from_half_extents/center/half_extents) anchored on the reference's call
 * sites; nothing here could be checked against an execution of the reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile). All
 * arithmetic is IEEE-754 binary32, round-to-nearest-even, no fused multiply-add.
 */
#ifndef MIP_ORACLE_H
#define MIP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LODS 6

/* Same layout as MipMesh in include/mi_instance_pipeline.h. */
typedef struct OrcMesh {
  float aabb_min[3];
  float aabb_max[3];
  uint32_t n_lods;
  uint32_t index_len[ORC_MAX_LODS];
  uint32_t index_offset[ORC_MAX_LODS];
  int32_t vertex_offset;
} OrcMesh;

/* VkDrawIndexedIndirectCommand, src/shaders/generate_work.comp:9-15 */
typedef struct OrcDrawCmd {
  uint32_t indexCount;
  uint32_t instanceCount;
  uint32_t firstIndex;
  int32_t vertexOffset;
  uint32_t firstInstance;
} OrcDrawCmd;

/* 4x4 matrices are column-major float[16]: m[c*4 + r] (glm::Mat4 / nalgebra storage). */

/* nalgebra UnitQuaternion::to_homogeneous (via to_rotation_matrix); q = [i,j,k,w]. */
void orc_quat_to_homogeneous(const float q[4], float m[16]);
/* glm::translation / glm::scaling (uniform). */
void orc_translation(const float p[3], float m[16]);
void orc_scaling(float s, float m[16]);
/* nalgebra Matrix4 * Matrix4 for statically sized matrices: per output column a gemv built
 * from axpy steps over the columns of `a`, left to right. */
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);
/* nalgebra Matrix4 * Vector4 (same gemv). */
void orc_mat4_mul_vec4(const float a[16], const float v[4], float out[4]);

/* src/ecs.rs:52-64 model_matrix_calculation: translation(p) * rot.to_homogeneous() * scaling(s). */
void orc_model_matrix(const float pos[3], const float rot_ijkw[4], float scale, float m[16]);

/* src/ecs.rs:138-181 aabb_calculation. */
void orc_world_aabb(const float m[16], const float mesh_min[3], const float mesh_max[3],
                    float mins[3], float maxs[3]);

/* src/renderer/systems/cull_pipeline.rs:99-120 coarse_culling; returns CoarseCulled (1 = culled). */
int orc_coarse_culled(const float mins[3], const float maxs[3], const float planes[24]);

/* src/renderer/helpers.rs:3-11 pick_lod; returns the LOD index (0 or 1). */
uint32_t orc_pick_lod(uint32_t n_lods, const float cam_pos[3], const float mesh_pos[3]);

/* src/ecs.rs:66-91 project_camera. The planes are an INPUT of the path (computed once per
 * frame on the host); this restatement exists to make default-camera test inputs and is
 * not claimed to match nalgebra-glm's look_at_lh bit for bit (it builds the view matrix
 * from the basis vectors directly instead of through a quaternion). cam_rot = [i,j,k,w]. */
void orc_project_camera(const float cam_pos[3], const float cam_rot_ijkw[4], float aspect,
                        float fovy_degrees, float near_z, float far_z, float planes[24]);

/* src/renderer/systems/cull_pipeline.rs:498-577 + src/shaders/generate_work.comp:61-67:
 * zero-fill cmds[0..n), then for every instance in draw_index order that is not
 * coarse-culled write cmds[i]; instance-level contract: indexCount = index_len of the
 * picked LOD. Returns Σ index_len (the final index_offset_in_output, wrapping u32). */
uint32_t orc_emit_draw_commands(uint32_t n, const float* pos_xyz, const uint32_t* mesh_id,
                                const uint8_t* coarse_culled, const OrcMesh* meshes,
                                const float cam_pos[3], uint32_t first_instance_base,
                                uint32_t first_index_base, OrcDrawCmd* cmds);

/* src/shaders/compact_draw_stream.comp:34-63 with its intended semantics (all n entries)
 * and the stable member of its outcome set (ascending index): keep indexCount > 0.
 * In-place capable (out may equal cmds). Returns count. */
uint32_t orc_compact_draw_stream(const OrcDrawCmd* cmds, uint32_t n, OrcDrawCmd* out);

typedef struct OrcOutputs {
  float* model;             /* n*16, may be NULL */
  float* world_aabb;        /* n*6 (mins, maxs), may be NULL */
  uint32_t* visible_bitmap; /* ceil(n/32) words, may be NULL */
  uint8_t* coarse_culled;   /* n bytes, may be NULL */
  OrcDrawCmd* draw_cmds;    /* n entries, may be NULL */
  uint32_t draw_count;
  uint32_t draw_index_total;
} OrcOutputs;

/* The whole path on one thread, literally: matrices -> AABBs -> cull -> emit -> compact.
 * Returns 0, or -1 on a mesh id >= m / allocation failure. */
int orc_run(uint32_t n, const float* pos_xyz, const float* rot_ijkw, const float* scale,
            const uint32_t* mesh_id, const OrcMesh* meshes, uint32_t m, const float planes[24],
            const float cam_pos[3], uint32_t first_instance_base, uint32_t first_index_base,
            OrcOutputs* out);

/* Same result on `threads` pthreads (static contiguous chunks, per-chunk emission, serial
 * prefix over chunks). Used for the CPU baseline. */
int orc_run_mt(uint32_t n, const float* pos_xyz, const float* rot_ijkw, const float* scale,
               const uint32_t* mesh_id, const OrcMesh* meshes, uint32_t m, const float planes[24],
               const float cam_pos[3], uint32_t first_instance_base, uint32_t first_index_base,
               OrcOutputs* out, uint32_t threads);

/* Concatenate shard draw lists in shard order, rebasing firstIndex by the index totals of
 * the earlier shards (what mip_merge_draw_lists does on the device). Returns total count. */
uint32_t orc_merge_draw_lists(uint32_t n_shards, const OrcDrawCmd* const* lists,
                              const uint32_t* counts, const uint32_t* index_totals,
                              OrcDrawCmd* out, uint32_t* out_index_total);

/* ---- row f-1: per-triangle cull + index-stream append ------------------------------------
 * src/shaders/generate_work.comp:68-200 for ONE draw command (one visible instance):
 * for every triangle of the picked LOD, in triangle order: fetch the index triple and the
 * three positions, clip = pv * (model * vec4(v, 1)), cull when the triangle is back-facing
 * (determinant of the xyw columns > 0) or all three vertices are beyond the same x or y
 * NDC bound, append the surviving index triples at out_index_buffer[firstIndex/3 + k].
 * Returns 3 x survivors (the command's final indexCount).
 *
 * PARITY UNPINNED, doubly: the reference evaluates this in GLSL on a Vulkan driver, which
 * fixes neither the summation order of mat*vec nor FMA contraction nor the determinant's
 * expansion. This restatement fixes them (column combination left to right, no FMA,
 * cofactor expansion along the x components of the three vertices, true division for the perspective divide);
 * where workgroups race in the reference (the order of surviving triangles across
 * workgroups, generate_work.comp:176-186) it emits the stable order.
 *   index_buffer / vertex_buffer : the consolidated buffers (u32 indices; packed vec3)
 *   src_index_offset             : push constant indexOffset (start of the LOD's indices)
 */
uint32_t orc_cull_triangles(const OrcDrawCmd* cmd, uint32_t src_index_offset, const float model[16],
                            const float pv[16], const float* vertex_buffer, const uint32_t* index_buffer,
                            uint32_t* out_index_buffer);

/* The per-frame sequence with per-triangle culling: instance path -> per command triangle
 * cull (rewrites indexCount) -> compaction on indexCount > 0. `cmds`/`count` are the output of
 * orc_run (indexCount = index_len); src_index_offset[k] belongs to cmds[k]. Returns the new count. */
uint32_t orc_cull_all_triangles(OrcDrawCmd* cmds, uint32_t count, const uint32_t* src_index_offset,
                                const float* model, uint32_t first_instance_base, const float pv[16],
                                const float* vertex_buffer, const uint32_t* index_buffer,
                                uint32_t* out_index_buffer, uint32_t threads);

/* index_offset of the LOD each emitted command draws from (what cull_pass puts in the push
 * constants, cull_pipeline.rs:545-553), in command order. */
void orc_src_index_offsets(uint32_t n, const float* pos_xyz, const uint32_t* mesh_id, const uint8_t* coarse_culled,
                           const OrcMesh* meshes, const float cam_pos[3], uint32_t* out);

/* ---- row f-4: TLAS instance rows (src/renderer/systems/acceleration_strucures.rs:419-451) ----
 * out: n x 64 bytes = VkAccelerationStructureInstanceKHR { float transform[12] (rows 0..2 of the
 * model matrix, row-major), u32 custom_index:24|mask:8, u32 sbt_offset:24|flags:8, u64 blas }. */
void orc_tlas_instances(uint32_t n, const float* model, const uint32_t* mesh_id, const uint64_t* blas_address,
                        uint32_t first_instance_base, void* out);

/* ---- row f-4, second consumer: the shadow pass's per-light draw lists
 * (src/renderer/systems/shadow_mapping.rs:405-478). out: n_lights x n commands, light-major. */
void orc_light_draw_lists(uint32_t n, const float* pos_xyz, const uint32_t* mesh_id, const OrcMesh* meshes,
                          const float* light_pos_xyz, uint32_t n_lights, uint32_t first_instance_base,
                          OrcDrawCmd* out);

/* ---- Extension: skinned instances (BASELINE config 5) — NOT a reference behaviour (the reference has
 * no skins). glTF 2.0 section 3.7.3: L_k = T*R*S of the pose (10 floats per joint:
 * t xyz, q ijkw, s xyz), G_k = G_parent*L_k (parent[k] < k or -1), J_k = G_k*inverse_bind_k, all as
 * affine 3x4 products in the order of orc_affine_mul. The posed box (mesh space) = fold, over the
 * joints with a non-empty box in ascending order, of the 8 corners of joint_box_k under J_k (corner
 * order and NaN-ignoring min/max of src/ecs.rs:146-173; seeds +-f32::MAX). It replaces GltfMesh.aabb
 * for the instance and the reference path runs on it unchanged. palette: n x n_joints x 16 (mat4
 * column-major) or NULL; local_box: n x 6 (min xyz, max xyz). */
void orc_affine_mul(const float a[12], const float b[12], float out[12]);
void orc_joint_local(const float trs[10], float l[12]);
int orc_skinned_bounds(uint32_t n, uint32_t n_joints, const int32_t* parent, const float* inverse_bind,
                       const float* joint_box, const float* poses, float* palette, float* local_box, uint32_t threads);
int orc_run_skinned(uint32_t n, const float* pos_xyz, const float* rot_ijkw, const float* scale, const uint32_t* mesh_id,
                    const OrcMesh* meshes, uint32_t m, uint32_t n_joints, const int32_t* parent, const float* inverse_bind,
                    const float* joint_box, const float* poses, const float planes[24], const float cam_pos[3],
                    uint32_t first_instance_base, uint32_t first_index_base, OrcOutputs* out, float* palette,
                    float* local_box_out, uint32_t threads);

/* CameraMatrices.pv = projection * view for orc_project_camera's camera (column-major). */
void orc_camera_pv(const float cam_pos[3], const float cam_rot_ijkw[4], float aspect, float fovy_degrees,
                   float near_z, float far_z, float pv[16]);

#ifdef __cplusplus
}
#endif
#endif
