/*
 * mip_oracle.c — CPU oracle of the instance pipeline. TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED — see mip_oracle.h for what that means and why.
 *
 * Every function restates one reference function (cited) with the arithmetic of the
 * un-vendored crates it calls written out operation by operation. Nothing is
 * simplified algebraically: the two 4x4 products of model_matrix_calculation are
 * performed in full (so non-finite inputs poison exactly the entries they would
 * poison in the reference), the eight corners go through a full mat4*vec4 and a
 * divide by w, and the AABB makes its lossy centre/half-extent round trip.
 *
 * Must be compiled without FMA contraction and without fast-math.
 */
#include "mip_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__FAST_MATH__)
#error "the oracle must not be built with -ffast-math"
#endif

/* ------------------------------------------------------------------------------------ */
/* nalgebra building blocks                                                              */
/* ------------------------------------------------------------------------------------ */

/* nalgebra 0.29 UnitQuaternion::to_rotation_matrix, then Rotation3::to_homogeneous
 * (identity with the 3x3 block copied in). Call site: src/ecs.rs:62 rot.0.to_homogeneous().
 * The quaternion is used as stored; nothing renormalises it. */
void orc_quat_to_homogeneous(const float q[4], float m[16]) {
  const float i = q[0], j = q[1], k = q[2], w = q[3];
  const float ww = w * w;
  const float ii = i * i;
  const float jj = j * j;
  const float kk = k * k;
  const float ij = i * j * 2.0f;
  const float wk = w * k * 2.0f;
  const float wj = w * j * 2.0f;
  const float ik = i * k * 2.0f;
  const float jk = j * k * 2.0f;
  const float wi = w * i * 2.0f;
  /* Matrix3::new takes its arguments row by row. */
  const float m11 = ww + ii - jj - kk, m12 = ij - wk, m13 = wj + ik;
  const float m21 = wk + ij, m22 = ww - ii + jj - kk, m23 = jk - wi;
  const float m31 = ik - wj, m32 = wi + jk, m33 = ww - ii - jj + kk;
  m[0] = m11; m[1] = m21; m[2] = m31; m[3] = 0.0f;
  m[4] = m12; m[5] = m22; m[6] = m32; m[7] = 0.0f;
  m[8] = m13; m[9] = m23; m[10] = m33; m[11] = 0.0f;
  m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;
}

/* glm::translation(&v) = Matrix4::new_translation: identity, column 3 = (v, 1). */
void orc_translation(const float p[3], float m[16]) {
  memset(m, 0, 16 * sizeof(float));
  m[0] = 1.0f; m[5] = 1.0f; m[10] = 1.0f; m[15] = 1.0f;
  m[12] = p[0]; m[13] = p[1]; m[14] = p[2];
}

/* glm::scaling(&Vec3::repeat(s)) = Matrix4::new_nonuniform_scaling: diag(s, s, s, 1). */
void orc_scaling(float s, float m[16]) {
  memset(m, 0, 16 * sizeof(float));
  m[0] = s; m[5] = s; m[10] = s; m[15] = 1.0f;
}

/* nalgebra gemv with alpha = 1, beta = 0 on a statically sized matrix:
 *   y  = (alpha * a[:,0]) * x[0]                       (axcpy with beta == 0)
 *   y  = (alpha * a[:,k]) * x[k] + 1 * y   for k = 1.. (axcpy with beta == 1)
 * so each output row is ((a_r0 x0 + a_r1 x1) + a_r2 x2) + a_r3 x3. */
static void gemv4(const float a[16], const float x[4], float y[4]) {
  for (int r = 0; r < 4; ++r) y[r] = 1.0f * a[0 * 4 + r] * x[0];
  for (int k = 1; k < 4; ++k)
    for (int r = 0; r < 4; ++r) y[r] = 1.0f * a[k * 4 + r] * x[k] + 1.0f * y[r];
}

/* nalgebra gemm for static dimensions: one gemv per output column. */
void orc_mat4_mul(const float a[16], const float b[16], float out[16]) {
  float tmp[16];
  for (int c = 0; c < 4; ++c) gemv4(a, &b[c * 4], &tmp[c * 4]);
  memcpy(out, tmp, sizeof tmp);
}

void orc_mat4_mul_vec4(const float a[16], const float v[4], float out[4]) {
  float tmp[4];
  gemv4(a, v, tmp);
  memcpy(out, tmp, sizeof tmp);
}

/* f32::min / f32::max (Rust): IEEE minNum/maxNum — a NaN operand is ignored. */
static inline float rust_min(float a, float b) { return fminf(a, b); }
static inline float rust_max(float a, float b) { return fmaxf(a, b); }

/* nalgebra dot, 3-wide special case: (a + b) + c. */
static inline float dot3(const float x[3], const float y[3]) {
  const float a = x[0] * y[0];
  const float b = x[1] * y[1];
  const float c = x[2] * y[2];
  return a + b + c;
}

/* nalgebra dot, 4-wide special case: a += c; b += d; a + b. */
static inline float dot4(const float x[4], const float y[4]) {
  float a = x[0] * y[0];
  float b = x[1] * y[1];
  const float c = x[2] * y[2];
  const float d = x[3] * y[3];
  a += c;
  b += d;
  return a + b;
}

/* ------------------------------------------------------------------------------------ */
/* the reference systems                                                                 */
/* ------------------------------------------------------------------------------------ */

/* src/ecs.rs:61-62
 *   model_matrix.0 = glm::translation(&pos.0.coords) * rot.0.to_homogeneous()
 *                    * glm::scaling(&glm::Vec3::repeat(scale.0));
 * `*` associates to the left: (T * R) * S. */
void orc_model_matrix(const float pos[3], const float rot_ijkw[4], float scale, float m[16]) {
  float t[16], r[16], s[16], tr[16];
  orc_translation(pos, t);
  orc_quat_to_homogeneous(rot_ijkw, r);
  orc_scaling(scale, s);
  orc_mat4_mul(t, r, tr);
  orc_mat4_mul(tr, s, m);
}

/* src/ecs.rs:146-179 */
void orc_world_aabb(const float m[16], const float mesh_min[3], const float mesh_max[3],
                    float mins[3], float maxs[3]) {
  const float* mn = mesh_min;
  const float* mx = mesh_max;
  /* corner order of src/ecs.rs:149-160 */
  const float corners[8][3] = {
      {mn[0], mn[1], mn[2]}, {mx[0], mn[1], mn[2]}, {mn[0], mn[1], mx[2]}, {mx[0], mn[1], mx[2]},
      {mn[0], mx[1], mn[2]}, {mx[0], mx[1], mn[2]}, {mn[0], mx[1], mx[2]}, {mx[0], mx[1], mx[2]},
  };
  /* fold seed: ((f32::MAX,..),(f32::MIN,..)); f32::MIN is -f32::MAX */
  float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
  float hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
  for (int c = 0; c < 8; ++c) {
    const float vh[4] = {corners[c][0], corners[c][1], corners[c][2], 1.0f}; /* to_homogeneous */
    float t[4];
    orc_mat4_mul_vec4(m, vh, t);           /* model_matrix.0 * vertex.to_homogeneous() */
    const float v[3] = {t[0] / t[3], t[1] / t[3], t[2] / t[3]}; /* vertex.xyz() / vertex.w */
    for (int a = 0; a < 3; ++a) {
      lo[a] = rust_min(lo[a], v[a]);
      hi[a] = rust_max(hi[a], v[a]);
    }
  }
  /* AABB::from_half_extents((max + min) / 2, (max - min) / 2) = new(c - h, c + h) */
  for (int a = 0; a < 3; ++a) {
    const float centre = (hi[a] + lo[a]) / 2.0f;
    const float half = (hi[a] - lo[a]) / 2.0f;
    mins[a] = centre - half;
    maxs[a] = centre + half;
  }
}

/* src/renderer/systems/cull_pipeline.rs:108-119 */
int orc_coarse_culled(const float mins[3], const float maxs[3], const float planes[24]) {
  /* ncollide AABB::half_extents = (maxs - mins) * 0.5; AABB::center = (mins + maxs) * 0.5 */
  float half[3], centre_h[4];
  for (int a = 0; a < 3; ++a) {
    half[a] = (maxs[a] - mins[a]) * 0.5f;
    centre_h[a] = (mins[a] + maxs[a]) * 0.5f;
  }
  centre_h[3] = 1.0f; /* center().to_homogeneous() */
  int outside = 0;
  for (int p = 0; p < 6; ++p) {
    const float* plane = &planes[p * 4];
    const float n_abs[3] = {fabsf(plane[0]), fabsf(plane[1]), fabsf(plane[2])};
    const float e = dot3(half, n_abs);
    const float s = dot4(plane, centre_h);
    if (s - e > 0.0f) {
      outside = 1;
      break;
    }
  }
  return outside;
}

/* src/renderer/helpers.rs:3-11 */
uint32_t orc_pick_lod(uint32_t n_lods, const float cam_pos[3], const float mesh_pos[3]) {
  const float d[3] = {cam_pos[0] - mesh_pos[0], cam_pos[1] - mesh_pos[1], cam_pos[2] - mesh_pos[2]};
  /* magnitude() = norm_squared().sqrt(); norm_squared accumulates the column's dot into 0 */
  float sq = 0.0f;
  sq += dot3(d, d);
  const float distance_from_camera = sqrtf(sq);
  return (distance_from_camera > 10.0f && n_lods > 1) ? 1u : 0u;
}

/* src/ecs.rs:66-82 (see header: input producer, not bit-pinned) */
void orc_camera_pv(const float cam_pos[3], const float cam_rot_ijkw[4], float aspect, float fovy_degrees,
                   float near_z, float far_z, float pv[16]) {
  float proj[16], view[16], rot[16];
  memset(proj, 0, sizeof proj);
  const float fovy = fovy_degrees * (3.14159265358979323846f / 180.0f);
  const float tan_half_fovy = tanf(fovy / 2.0f);
  /* glm::perspective_lh_zo */
  proj[0 * 4 + 0] = 1.0f / (aspect * tan_half_fovy);
  proj[1 * 4 + 1] = 1.0f / tan_half_fovy;
  proj[2 * 4 + 2] = far_z / (far_z - near_z);
  proj[3 * 4 + 2] = -(far_z * near_z) / (far_z - near_z);
  proj[2 * 4 + 3] = 1.0f;
  /* dir = rotation * forward(+z), up = rotation * up(+y): columns 2 and 1 of the rotation */
  orc_quat_to_homogeneous(cam_rot_ijkw, rot);
  float z[3] = {rot[8], rot[9], rot[10]};
  float up[3] = {rot[4], rot[5], rot[6]};
  float zl = sqrtf(dot3(z, z));
  for (int a = 0; a < 3; ++a) z[a] /= zl;
  float x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
  float xl = sqrtf(dot3(x, x));
  for (int a = 0; a < 3; ++a) x[a] /= xl;
  float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
  /* glm::look_at_lh: rows are the basis, translation = -basis . eye */
  memset(view, 0, sizeof view);
  for (int a = 0; a < 3; ++a) {
    view[a * 4 + 0] = x[a];
    view[a * 4 + 1] = y[a];
    view[a * 4 + 2] = z[a];
  }
  view[12] = -dot3(x, cam_pos);
  view[13] = -dot3(y, cam_pos);
  view[14] = -dot3(z, cam_pos);
  view[15] = 1.0f;
  orc_mat4_mul(proj, view, pv); /* m = camera.projection * camera.view */
}

/* src/ecs.rs:82-90 */
void orc_project_camera(const float cam_pos[3], const float cam_rot_ijkw[4], float aspect,
                        float fovy_degrees, float near_z, float far_z, float planes[24]) {
  float pv[16];
  orc_camera_pv(cam_pos, cam_rot_ijkw, aspect, fovy_degrees, near_z, far_z, pv);
  /* -(row3 +/- row k), k = 0,1,2: left, right, bottom, top, near, far (src/ecs.rs:83-90) */
  for (int k = 0; k < 3; ++k)
    for (int c = 0; c < 4; ++c) {
      const float r3 = pv[c * 4 + 3], rk = pv[c * 4 + k];
      planes[(2 * k + 0) * 4 + c] = -(r3 + rk);
      planes[(2 * k + 1) * 4 + c] = -(r3 - rk);
    }
}

/* src/renderer/systems/cull_pipeline.rs:498-577 (host loop) with the header writes of
 * src/shaders/generate_work.comp:61-67. */
uint32_t orc_emit_draw_commands(uint32_t n, const float* pos_xyz, const uint32_t* mesh_id,
                                const uint8_t* coarse_culled, const OrcMesh* meshes,
                                const float cam_pos[3], uint32_t first_instance_base,
                                uint32_t first_index_base, OrcDrawCmd* cmds) {
  /* cmd_fill_buffer(commands_buffer, 0) — cull_pipeline.rs:498-504 */
  memset(cmds, 0, (size_t)n * sizeof(OrcDrawCmd));
  uint32_t index_offset_in_output = 0; /* cull_pipeline.rs:534 (u32 push-constant field) */
  for (uint32_t i = 0; i < n; ++i) {
    if (coarse_culled[i]) continue; /* :537-539 */
    const OrcMesh* mesh = &meshes[mesh_id[i]];
    const uint32_t lod = orc_pick_lod(mesh->n_lods, cam_pos, &pos_xyz[(size_t)i * 3]); /* :544 */
    const uint32_t index_len = mesh->index_len[lod];
    OrcDrawCmd* c = &cmds[i]; /* indirect_commands[gltfIndex], gltfIndex = draw_index */
    /* Instance-level contract of this tier: every triangle survives, so
     * the indexCount the per-triangle kernel would accumulate equals index_len. */
    c->indexCount = index_len;
    c->instanceCount = 1;                                    /* generate_work.comp:63 */
    c->firstInstance = first_instance_base + i;              /* :64 */
    c->firstIndex = first_index_base + index_offset_in_output; /* :65 */
    c->vertexOffset = mesh->vertex_offset;                   /* :66 */
    index_offset_in_output += index_len;                     /* cull_pipeline.rs:558 */
  }
  return index_offset_in_output;
}

/* src/shaders/compact_draw_stream.comp:39-62 — keep `indexCount > 0`, pack to the front. */
uint32_t orc_compact_draw_stream(const OrcDrawCmd* cmds, uint32_t n, OrcDrawCmd* out) {
  uint32_t count = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const OrcDrawCmd copied = cmds[i];
    const int use = copied.indexCount > 0;
    if (use) out[count++] = copied;
  }
  return count;
}

/* ------------------------------------------------------------------------------------ */
/* whole path                                                                            */
/* ------------------------------------------------------------------------------------ */

static void instance_range(uint32_t begin, uint32_t end, const float* pos_xyz, const float* rot_ijkw,
                           const float* scale, const uint32_t* mesh_id, const OrcMesh* meshes,
                           const float planes[24], float* model, float* world_aabb,
                           uint8_t* coarse_culled) {
  for (uint32_t i = begin; i < end; ++i) {
    float m[16], mins[3], maxs[3];
    const OrcMesh* mesh = &meshes[mesh_id[i]];
    orc_model_matrix(&pos_xyz[(size_t)i * 3], &rot_ijkw[(size_t)i * 4], scale[i], m);
    orc_world_aabb(m, mesh->aabb_min, mesh->aabb_max, mins, maxs);
    coarse_culled[i] = (uint8_t)orc_coarse_culled(mins, maxs, planes);
    if (model) memcpy(&model[(size_t)i * 16], m, sizeof m); /* model[draw_index] = M, renderer.rs:2281 */
    if (world_aabb) {
      memcpy(&world_aabb[(size_t)i * 6], mins, sizeof mins);
      memcpy(&world_aabb[(size_t)i * 6 + 3], maxs, sizeof maxs);
    }
  }
}

static void pack_bitmap(uint32_t n, const uint8_t* coarse_culled, uint32_t* bitmap) {
  const uint32_t words = (n + 31u) / 32u;
  memset(bitmap, 0, (size_t)words * sizeof(uint32_t));
  for (uint32_t i = 0; i < n; ++i)
    if (!coarse_culled[i]) bitmap[i >> 5] |= 1u << (i & 31u);
}

static int check_mesh_ids(uint32_t n, const uint32_t* mesh_id, uint32_t m) {
  for (uint32_t i = 0; i < n; ++i)
    if (mesh_id[i] >= m) return -1;
  return 0;
}

int orc_run(uint32_t n, const float* pos_xyz, const float* rot_ijkw, const float* scale,
            const uint32_t* mesh_id, const OrcMesh* meshes, uint32_t m, const float planes[24],
            const float cam_pos[3], uint32_t first_instance_base, uint32_t first_index_base,
            OrcOutputs* out) {
  if (check_mesh_ids(n, mesh_id, m)) return -1;
  uint8_t* culled = out->coarse_culled;
  uint8_t* culled_owned = NULL;
  if (!culled) {
    culled = culled_owned = (uint8_t*)malloc(n ? n : 1);
    if (!culled) return -1;
  }
  instance_range(0, n, pos_xyz, rot_ijkw, scale, mesh_id, meshes, planes, out->model,
                 out->world_aabb, culled);
  if (out->visible_bitmap) pack_bitmap(n, culled, out->visible_bitmap);
  out->draw_count = 0;
  out->draw_index_total = 0;
  if (out->draw_cmds) {
    OrcDrawCmd* sparse = (OrcDrawCmd*)malloc((size_t)(n ? n : 1) * sizeof(OrcDrawCmd));
    if (!sparse) {
      free(culled_owned);
      return -1;
    }
    out->draw_index_total = orc_emit_draw_commands(n, pos_xyz, mesh_id, culled, meshes, cam_pos,
                                                   first_instance_base, first_index_base, sparse);
    out->draw_count = orc_compact_draw_stream(sparse, n, out->draw_cmds);
    free(sparse);
  }
  free(culled_owned);
  return 0;
}

/* ---- threaded variant (CPU baseline) ---- */

typedef struct Job {
  uint32_t begin, end;
  const float *pos_xyz, *rot_ijkw, *scale;
  const uint32_t* mesh_id;
  const OrcMesh* meshes;
  const float* planes;
  const float* cam_pos;
  float *model, *world_aabb;
  uint8_t* culled;
  OrcDrawCmd* cmds;
  uint32_t first_instance_base;
  /* phase A results */
  uint32_t count, index_sum;
  /* phase B inputs */
  uint32_t count_base, index_base;
} Job;

static void* phase_a(void* arg) {
  Job* j = (Job*)arg;
  instance_range(j->begin, j->end, j->pos_xyz, j->rot_ijkw, j->scale, j->mesh_id, j->meshes,
                 j->planes, j->model, j->world_aabb, j->culled);
  uint32_t count = 0, sum = 0;
  for (uint32_t i = j->begin; i < j->end; ++i) {
    if (j->culled[i]) continue;
    const OrcMesh* mesh = &j->meshes[j->mesh_id[i]];
    const uint32_t len = mesh->index_len[orc_pick_lod(mesh->n_lods, j->cam_pos, &j->pos_xyz[(size_t)i * 3])];
    sum += len;
    count += len > 0;
  }
  j->count = count;
  j->index_sum = sum;
  return NULL;
}

static void* phase_b(void* arg) {
  Job* j = (Job*)arg;
  uint32_t at = j->count_base, first_index = j->index_base;
  for (uint32_t i = j->begin; i < j->end; ++i) {
    if (j->culled[i]) continue;
    const OrcMesh* mesh = &j->meshes[j->mesh_id[i]];
    const uint32_t len = mesh->index_len[orc_pick_lod(mesh->n_lods, j->cam_pos, &j->pos_xyz[(size_t)i * 3])];
    if (len > 0) {
      OrcDrawCmd* c = &j->cmds[at++];
      c->indexCount = len;
      c->instanceCount = 1;
      c->firstIndex = first_index;
      c->vertexOffset = mesh->vertex_offset;
      c->firstInstance = j->first_instance_base + i;
    }
    first_index += len;
  }
  return NULL;
}

int orc_run_mt(uint32_t n, const float* pos_xyz, const float* rot_ijkw, const float* scale,
               const uint32_t* mesh_id, const OrcMesh* meshes, uint32_t m, const float planes[24],
               const float cam_pos[3], uint32_t first_instance_base, uint32_t first_index_base,
               OrcOutputs* out, uint32_t threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if (check_mesh_ids(n, mesh_id, m)) return -1;
  uint8_t* culled = out->coarse_culled;
  uint8_t* culled_owned = NULL;
  if (!culled) {
    culled = culled_owned = (uint8_t*)malloc(n ? n : 1);
    if (!culled) return -1;
  }
  Job* jobs = (Job*)calloc(threads, sizeof(Job));
  pthread_t* tids = (pthread_t*)calloc(threads, sizeof(pthread_t));
  if (!jobs || !tids) {
    free(jobs); free(tids); free(culled_owned);
    return -1;
  }
  const uint32_t chunk = (n + threads - 1) / threads;
  for (uint32_t t = 0; t < threads; ++t) {
    Job* j = &jobs[t];
    const uint64_t b = (uint64_t)t * chunk, e = b + chunk;
    j->begin = (uint32_t)(b < n ? b : n);
    j->end = (uint32_t)(e < n ? e : n);
    j->pos_xyz = pos_xyz; j->rot_ijkw = rot_ijkw; j->scale = scale; j->mesh_id = mesh_id;
    j->meshes = meshes; j->planes = planes; j->cam_pos = cam_pos;
    j->model = out->model; j->world_aabb = out->world_aabb; j->culled = culled;
    j->cmds = out->draw_cmds; j->first_instance_base = first_instance_base;
  }
  for (uint32_t t = 1; t < threads; ++t) pthread_create(&tids[t], NULL, phase_a, &jobs[t]);
  phase_a(&jobs[0]);
  for (uint32_t t = 1; t < threads; ++t) pthread_join(tids[t], NULL);
  uint32_t count = 0, sum = 0;
  for (uint32_t t = 0; t < threads; ++t) {
    jobs[t].count_base = count;
    jobs[t].index_base = first_index_base + sum;
    count += jobs[t].count;
    sum += jobs[t].index_sum;
  }
  out->draw_count = 0;
  out->draw_index_total = 0;
  if (out->draw_cmds) {
    for (uint32_t t = 1; t < threads; ++t) pthread_create(&tids[t], NULL, phase_b, &jobs[t]);
    phase_b(&jobs[0]);
    for (uint32_t t = 1; t < threads; ++t) pthread_join(tids[t], NULL);
    out->draw_count = count;
    out->draw_index_total = sum;
  }
  if (out->visible_bitmap) pack_bitmap(n, culled, out->visible_bitmap);
  free(jobs); free(tids); free(culled_owned);
  return 0;
}

uint32_t orc_merge_draw_lists(uint32_t n_shards, const OrcDrawCmd* const* lists,
                              const uint32_t* counts, const uint32_t* index_totals,
                              OrcDrawCmd* out, uint32_t* out_index_total) {
  uint32_t at = 0, index_base = 0;
  for (uint32_t s = 0; s < n_shards; ++s) {
    for (uint32_t i = 0; i < counts[s]; ++i) {
      OrcDrawCmd c = lists[s][i];
      c.firstIndex += index_base;
      out[at++] = c;
    }
    index_base += index_totals[s];
  }
  if (out_index_total) *out_index_total = index_base;
  return at;
}

/* ------------------------------------------------------------------------------------ */
/* row f-1: per-triangle cull + index-stream append (generate_work.comp:68-200)          */
/* ------------------------------------------------------------------------------------ */

/* GLSL mat4 * vec4, fixed here as the column combination left to right, no FMA. */
static void glsl_mat4_mul_vec4(const float m[16], const float v[4], float out[4]) {
  for (int r = 0; r < 4; ++r) out[r] = ((m[0 * 4 + r] * v[0] + m[1 * 4 + r] * v[1]) + m[2 * 4 + r] * v[2]) + m[3 * 4 + r] * v[3];
}

uint32_t orc_cull_triangles(const OrcDrawCmd* cmd, uint32_t src_index_offset, const float model[16],
                            const float pv[16], const float* vertex_buffer, const uint32_t* index_buffer,
                            uint32_t* out_index_buffer) {
  const uint32_t n_tris = cmd->indexCount / 3;             /* `indexCount / 3`, :77 */
  uint32_t survivors = 0;
  for (uint32_t t = 0; t < n_tris; ++t) {
    const uint32_t* ix = &index_buffer[(size_t)(src_index_offset / 3 + t) * 3]; /* index_buffer[indexOffset / 3 + id], :78 */
    float clip[3][4];
    for (int k = 0; k < 3; ++k) {
      const float* p = &vertex_buffer[(size_t)((int64_t)cmd->vertexOffset + (int64_t)ix[k]) * 3]; /* :121,:135 */
      const float vh[4] = {p[0], p[1], p[2], 1.0f};
      float world[4];
      glsl_mat4_mul_vec4(model, vh, world);                /* model_mat * vec4(v, 1.0), :138-140 */
      glsl_mat4_mul_vec4(pv, world, clip[k]);              /* pv * (...) */
    }
    /* determinant(mat3(v0.xyw, v1.xyw, v2.xyw)) > 0: columns are the xyw of the three vertices, :143 */
    const float a00 = clip[0][0], a01 = clip[0][1], a02 = clip[0][3];
    const float a10 = clip[1][0], a11 = clip[1][1], a12 = clip[1][3];
    const float a20 = clip[2][0], a21 = clip[2][1], a22 = clip[2][3];
    const float det = (a00 * (a11 * a22 - a21 * a12) - a10 * (a01 * a22 - a21 * a02)) + a20 * (a01 * a12 - a11 * a02);
    int cull = det > 0.0f;
    float ndc[3][2];
    for (int k = 0; k < 3; ++k) {
      ndc[k][0] = clip[k][0] / clip[k][3];                 /* vertex.xyz / vertex.w, :145-147 (z unused below) */
      ndc[k][1] = clip[k][1] / clip[k][3];
    }
    if (!cull)
      cull = (ndc[0][0] < -1.0f && ndc[1][0] < -1.0f && ndc[2][0] < -1.0f) ||
             (ndc[0][0] > 1.0f && ndc[1][0] > 1.0f && ndc[2][0] > 1.0f) ||
             (ndc[0][1] < -1.0f && ndc[1][1] < -1.0f && ndc[2][1] < -1.0f) ||
             (ndc[0][1] > 1.0f && ndc[1][1] > 1.0f && ndc[2][1] > 1.0f); /* :149-155 */
    /* the degenerate-triangle test is disabled in the reference (`cull = false`, :165) */
    if (!cull) {
      uint32_t* dst = &out_index_buffer[((size_t)cmd->firstIndex / 3 + survivors) * 3]; /* :181-190 */
      dst[0] = ix[0]; dst[1] = ix[1]; dst[2] = ix[2];
      ++survivors;
    }
  }
  return survivors * 3;
}

void orc_src_index_offsets(uint32_t n, const float* pos_xyz, const uint32_t* mesh_id, const uint8_t* coarse_culled,
                           const OrcMesh* meshes, const float cam_pos[3], uint32_t* out) {
  uint32_t at = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (coarse_culled[i]) continue;
    const OrcMesh* mesh = &meshes[mesh_id[i]];
    const uint32_t lod = orc_pick_lod(mesh->n_lods, cam_pos, &pos_xyz[(size_t)i * 3]);
    if (mesh->index_len[lod] > 0) out[at++] = mesh->index_offset[lod];
  }
}

typedef struct TriJob {
  OrcDrawCmd* cmds;
  uint32_t begin, end;
  const uint32_t* src;
  const float* model;
  uint32_t base;
  const float* pv;
  const float* vb;
  const uint32_t* ib;
  uint32_t* out;
} TriJob;

static void* tri_job(void* arg) {
  TriJob* j = (TriJob*)arg;
  for (uint32_t k = j->begin; k < j->end; ++k) {
    OrcDrawCmd* c = &j->cmds[k];
    const float* m = &j->model[(size_t)(c->firstInstance - j->base) * 16]; /* model[gltfIndex] */
    c->indexCount = orc_cull_triangles(c, j->src[k], m, j->pv, j->vb, j->ib, j->out);
  }
  return NULL;
}

uint32_t orc_cull_all_triangles(OrcDrawCmd* cmds, uint32_t count, const uint32_t* src_index_offset,
                                const float* model, uint32_t first_instance_base, const float pv[16],
                                const float* vertex_buffer, const uint32_t* index_buffer,
                                uint32_t* out_index_buffer, uint32_t threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  TriJob jobs[256];
  pthread_t tids[256];
  const uint32_t chunk = (count + threads - 1) / threads;
  for (uint32_t t = 0; t < threads; ++t) {
    const uint64_t b = (uint64_t)t * chunk, e = b + chunk;
    jobs[t] = (TriJob){cmds, (uint32_t)(b < count ? b : count), (uint32_t)(e < count ? e : count), src_index_offset,
                       model, first_instance_base, pv, vertex_buffer, index_buffer, out_index_buffer};
  }
  for (uint32_t t = 1; t < threads; ++t) pthread_create(&tids[t], NULL, tri_job, &jobs[t]);
  tri_job(&jobs[0]);
  for (uint32_t t = 1; t < threads; ++t) pthread_join(tids[t], NULL);
  /* compact_draw_stream.comp runs after generate_work: commands whose triangles all died vanish */
  return orc_compact_draw_stream(cmds, count, cmds);
}

/* src/renderer/systems/acceleration_strucures.rs:419-451 */
void orc_tlas_instances(uint32_t n, const float* model, const uint32_t* mesh_id, const uint64_t* blas_address,
                        uint32_t first_instance_base, void* out) {
  unsigned char* o = (unsigned char*)out;
  for (uint32_t i = 0; i < n; ++i, o += 64) {
    const float* m = &model[(size_t)i * 16];
    float transform[12];
    for (int r = 0; r < 3; ++r)          /* model_matrix.rows(0, 3).transpose().as_slice(): row-major 3x4 */
      for (int c = 0; c < 4; ++c) transform[r * 4 + c] = m[c * 4 + r];
    const uint32_t custom_and_mask = ((first_instance_base + i) & 0xffffffu) | (0xFFu << 24); /* Packed24_8::new(draw_index, 0xFF) */
    const uint32_t sbt_and_flags = 0u | (0x1u << 24); /* TRIANGLE_FACING_CULL_DISABLE = 1 */
    const uint64_t blas = blas_address ? blas_address[mesh_id[i]] : 0;
    memcpy(o, transform, 48);
    memcpy(o + 48, &custom_and_mask, 4);
    memcpy(o + 52, &sbt_and_flags, 4);
    memcpy(o + 56, &blas, 8);
  }
}

/* src/renderer/systems/shadow_mapping.rs:405-478: every light draws every mesh entity with the LOD
 * picked against the LIGHT's position; firstIndex/vertexOffset address the consolidated buffers
 * as cull_pass does (cull_pipeline.rs:540-553). */
void orc_light_draw_lists(uint32_t n, const float* pos_xyz, const uint32_t* mesh_id, const OrcMesh* meshes,
                          const float* light_pos_xyz, uint32_t n_lights, uint32_t first_instance_base,
                          OrcDrawCmd* out) {
  for (uint32_t l = 0; l < n_lights; ++l)
    for (uint32_t i = 0; i < n; ++i) {
      const OrcMesh* mesh = &meshes[mesh_id[i]];
      const uint32_t lod = orc_pick_lod(mesh->n_lods, &light_pos_xyz[(size_t)l * 3], &pos_xyz[(size_t)i * 3]);
      OrcDrawCmd* c = &out[(size_t)l * n + i];
      c->indexCount = mesh->index_len[lod];
      c->instanceCount = 1;
      c->firstIndex = mesh->index_offset[lod];
      c->vertexOffset = mesh->vertex_offset;
      c->firstInstance = first_instance_base + i;
    }
}

/* ---- Extension: skinned instances (BASELINE config 5). NOT a reference behaviour: the reference
 * has no skins (top table). Specified from glTF 2.0 section 3.7.3; this
 * restatement is the only authority for it, so parity here is against the build's own oracle. ---- */

/* Affine 3x4, column-major a[c*3 + r]. Per column: (a0*b0c + a1*b1c) + a2*b2c, translation column "+ a3". */
void orc_affine_mul(const float a[12], const float b[12], float out[12]) {
  float o[12];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r) {
      float v = a[0 * 3 + r] * b[c * 3 + 0] + a[1 * 3 + r] * b[c * 3 + 1] + a[2 * 3 + r] * b[c * 3 + 2];
      if (c == 3) v = v + a[9 + r];
      o[c * 3 + r] = v;
    }
  memcpy(out, o, sizeof o);
}

/* L = T(t) * R(q) * S(s) for one joint; trs = t xyz, q ijkw, s xyz. R as UnitQuaternion::to_rotation_matrix. */
void orc_joint_local(const float trs[10], float l[12]) {
  float rh[16];
  orc_quat_to_homogeneous(&trs[3], rh);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) l[c * 3 + r] = rh[c * 4 + r] * trs[7 + c];
  l[9] = trs[0];
  l[10] = trs[1];
  l[11] = trs[2];
}

typedef struct SkinJob {
  uint32_t begin, end, n_joints;
  const float *inverse_bind, *joint_box, *poses;
  const int32_t* parent;
  float *palette, *local_box;
} SkinJob;

static void* skin_range(void* arg) {
  const SkinJob* j = (const SkinJob*)arg;
  const uint32_t J = j->n_joints;
  float g[32][12];
  for (uint32_t i = j->begin; i < j->end; ++i) {
    float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
    float hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
    for (uint32_t k = 0; k < J; ++k) {
      float l[12], ibm[12], jm[12];
      orc_joint_local(&j->poses[((size_t)i * J + k) * 10], l);
      if (j->parent[k] < 0) memcpy(g[k], l, sizeof l);
      else orc_affine_mul(g[j->parent[k]], l, g[k]);
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r) ibm[c * 3 + r] = j->inverse_bind[(size_t)k * 16 + c * 4 + r];
      orc_affine_mul(g[k], ibm, jm);
      if (j->palette) {
        float* p = &j->palette[((size_t)i * J + k) * 16];
        for (int c = 0; c < 4; ++c) {
          for (int r = 0; r < 3; ++r) p[c * 4 + r] = jm[c * 3 + r];
          p[c * 4 + 3] = c == 3 ? 1.0f : 0.0f;
        }
      }
      const float* b = &j->joint_box[(size_t)k * 6];
      if (b[0] > b[3] || b[1] > b[4] || b[2] > b[5]) continue; /* the joint binds no vertex */
      for (int c = 0; c < 8; ++c) { /* corner order of src/ecs.rs:149-160 */
        const float x = b[(c & 1) ? 3 : 0], z = b[(c & 2) ? 5 : 2], y = b[(c & 4) ? 4 : 1];
        for (int r = 0; r < 3; ++r) {
          const float v = jm[0 * 3 + r] * x + jm[1 * 3 + r] * y + jm[2 * 3 + r] * z + jm[9 + r];
          lo[r] = rust_min(lo[r], v);
          hi[r] = rust_max(hi[r], v);
        }
      }
    }
    memcpy(&j->local_box[(size_t)i * 6], lo, sizeof lo);
    memcpy(&j->local_box[(size_t)i * 6 + 3], hi, sizeof hi);
  }
  return NULL;
}

int orc_skinned_bounds(uint32_t n, uint32_t n_joints, const int32_t* parent, const float* inverse_bind,
                       const float* joint_box, const float* poses, float* palette, float* local_box, uint32_t threads) {
  if (n_joints == 0 || n_joints > 32) return -1;
  for (uint32_t k = 0; k < n_joints; ++k)
    if (parent[k] >= (int32_t)k || parent[k] < -1) return -1;
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  SkinJob jobs[64];
  pthread_t tids[64];
  const uint32_t per = (n + threads - 1) / threads;
  for (uint32_t t = 0; t < threads; ++t) {
    SkinJob* j = &jobs[t];
    j->begin = t * per < n ? t * per : n;
    j->end = (t + 1) * per < n ? (t + 1) * per : n;
    j->n_joints = n_joints;
    j->inverse_bind = inverse_bind; j->joint_box = joint_box; j->poses = poses;
    j->parent = parent; j->palette = palette; j->local_box = local_box;
  }
  for (uint32_t t = 1; t < threads; ++t)
    if (pthread_create(&tids[t], NULL, skin_range, &jobs[t]) != 0) return -1;
  skin_range(&jobs[0]);
  for (uint32_t t = 1; t < threads; ++t) pthread_join(tids[t], NULL);
  return 0;
}

/* The frame of skinned instances: the posed mesh-space box takes the place of GltfMesh.aabb, and the
 * reference path runs on it unchanged (model matrix, aabb_calculation, coarse_culling, emission,
 * compaction). */
int orc_run_skinned(uint32_t n, const float* pos_xyz, const float* rot_ijkw, const float* scale, const uint32_t* mesh_id,
                    const OrcMesh* meshes, uint32_t m, uint32_t n_joints, const int32_t* parent, const float* inverse_bind,
                    const float* joint_box, const float* poses, const float planes[24], const float cam_pos[3],
                    uint32_t first_instance_base, uint32_t first_index_base, OrcOutputs* out, float* palette,
                    float* local_box_out, uint32_t threads) {
  if (check_mesh_ids(n, mesh_id, m)) return -1;
  uint8_t* culled = out->coarse_culled;
  uint8_t* culled_owned = NULL;
  float* box = local_box_out;
  float* box_owned = NULL;
  if (!culled) culled = culled_owned = (uint8_t*)malloc(n ? n : 1);
  if (!box) box = box_owned = (float*)malloc((size_t)(n ? n : 1) * 6 * sizeof(float));
  int rc = (culled && box) ? 0 : -1;
  if (rc == 0) rc = orc_skinned_bounds(n, n_joints, parent, inverse_bind, joint_box, poses, palette, box, threads);
  if (rc == 0) {
    for (uint32_t i = 0; i < n; ++i) {
      float m16[16], mins[3], maxs[3];
      orc_model_matrix(&pos_xyz[(size_t)i * 3], &rot_ijkw[(size_t)i * 4], scale[i], m16);
      orc_world_aabb(m16, &box[(size_t)i * 6], &box[(size_t)i * 6 + 3], mins, maxs);
      culled[i] = (uint8_t)orc_coarse_culled(mins, maxs, planes);
      if (out->model) memcpy(&out->model[(size_t)i * 16], m16, sizeof m16);
      if (out->world_aabb) {
        memcpy(&out->world_aabb[(size_t)i * 6], mins, sizeof mins);
        memcpy(&out->world_aabb[(size_t)i * 6 + 3], maxs, sizeof maxs);
      }
    }
    if (out->visible_bitmap) pack_bitmap(n, culled, out->visible_bitmap);
    out->draw_count = 0;
    out->draw_index_total = 0;
    if (out->draw_cmds) {
      OrcDrawCmd* sparse = (OrcDrawCmd*)malloc((size_t)(n ? n : 1) * sizeof(OrcDrawCmd));
      if (!sparse) rc = -1;
      else {
        out->draw_index_total = orc_emit_draw_commands(n, pos_xyz, mesh_id, culled, meshes, cam_pos, first_instance_base,
                                                       first_index_base, sparse);
        out->draw_count = orc_compact_draw_stream(sparse, n, out->draw_cmds);
        free(sparse);
      }
    }
  }
  free(culled_owned);
  free(box_owned);
  return rc;
}
