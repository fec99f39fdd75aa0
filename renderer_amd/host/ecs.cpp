// ecs.cpp — see ecs.hpp. Everything numeric on the per-entity path happens on the device
// through the C ABI; the only host arithmetic is project_camera (once per frame).
#include "ecs.hpp"

#include <cmath>
#include <cstring>

namespace renderer {
namespace ecs {

size_t World::spawn(components::Position p, components::Rotation r, components::Scale s, components::GltfMesh m) {
  position.push_back(p);
  rotation.push_back(r);
  scale.push_back(s);
  mesh.push_back(m);
  components::ModelMatrix id{};  // ModelMatrix::default() = identity (components.rs:49-53)
  id.m[0] = id.m[5] = id.m[10] = id.m[15] = 1.0f;
  model_matrix.push_back(id);
  aabb.push_back(components::AABB{{0, 0, 0}, {0, 0, 0}});  // AABB::default() (components.rs:40-47)
  draw_index.push_back(DrawIndex{0});
  coarse_culled.push_back(CoarseCulled{false});
  changed = true;
  return len() - 1;
}

InstancePipeline::InstancePipeline(uint32_t max_instances, const std::vector<MipMesh>& mesh_library, int device) {
  MipConfig cfg{};
  cfg.struct_size = sizeof(MipConfig);
  cfg.device_ordinal = device;
  cfg.max_instances = max_instances;
  cfg.max_meshes = (uint32_t)mesh_library.size();
  int32_t rc = mip_create(&cfg, &ctx_);
  if (rc != MIP_OK) throw Error(rc, "mip_create failed (no gfx950 device? there is no CPU fallback)");
  n_meshes_ = (uint32_t)mesh_library.size();
  check(mip_set_mesh_table(ctx_, mesh_library.data(), n_meshes_), "mip_set_mesh_table");
}

InstancePipeline::~InstancePipeline() { mip_destroy(ctx_); }

void InstancePipeline::check(int32_t rc, const char* what) const {
  if (rc != MIP_OK) throw Error(rc, std::string(what) + ": " + mip_last_error(ctx_));
}

void InstancePipeline::ensure_frame(World& world, const resources::Camera& camera) {
  if (frame_valid_) return;
  const uint32_t n = (uint32_t)world.len();
  if (world.changed) {
    // The component structs are exactly the packed SoA columns the ABI takes.
    static_assert(sizeof(components::Position) == 12 && sizeof(components::Rotation) == 16 &&
                  sizeof(components::Scale) == 4 && sizeof(components::GltfMesh) == 4, "column layout");
    check(mip_set_instances(ctx_, n ? &world.position[0].x : nullptr, n ? &world.rotation[0].i : nullptr,
                            n ? &world.scale[0].s : nullptr, n ? &world.mesh[0].mesh : nullptr, n),
          "mip_set_instances");
    world.changed = false;
  }
  MipFrame frame{};
  std::memcpy(frame.planes, camera.frustum_planes, sizeof frame.planes);
  std::memcpy(frame.cam_pos, camera.position, sizeof frame.cam_pos);
  model_.resize((size_t)n * 16);
  aabb_.resize((size_t)n * 6);
  bitmap_.assign((n + 31) / 32, 0u);
  indirect_.commands.resize(n ? n : 1);
  MipOutputs out{};
  out.model = model_.data();
  out.world_aabb = aabb_.data();
  out.visible_bitmap = bitmap_.data();
  out.draw_cmds = indirect_.commands.data();
  out.draw_count = &indirect_.count;
  out.draw_index_total = &index_total_;
  out.flags = MIP_OUT_HOST;
  check(mip_run(ctx_, &frame, &out), "mip_run");
  frame_valid_ = true;
}

namespace systems {

namespace {
// column-major 4x4 product in nalgebra's gemm order (per output column: axpy over a's columns)
void mat4_mul(const float a[16], const float b[16], float out[16]) {
  float t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float y = a[0 * 4 + r] * b[c * 4 + 0];
      for (int k = 1; k < 4; ++k) y = a[k * 4 + r] * b[c * 4 + k] + y;
      t[c * 4 + r] = y;
    }
  std::memcpy(out, t, sizeof t);
}
void rotate(const float q[4], const float v[3], float out[3]) {  // UnitQuaternion::transform_vector
  const float i = q[0], j = q[1], k = q[2], w = q[3];
  const float tx = 2.0f * (j * v[2] - k * v[1]), ty = 2.0f * (k * v[0] - i * v[2]), tz = 2.0f * (i * v[1] - j * v[0]);
  out[0] = v[0] + w * tx + (j * tz - k * ty);
  out[1] = v[1] + w * ty + (k * tx - i * tz);
  out[2] = v[2] + w * tz + (i * ty - j * tx);
}
float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
void cross(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
void normalize(float v[3]) {
  const float l = std::sqrt(dot3(v, v));
  v[0] /= l; v[1] /= l; v[2] /= l;
}
}  // namespace

void project_camera(const resources::Swapchain& swapchain, resources::Camera& camera) {
  const float near_z = 0.1f, far_z = 100.0f;                                  // ecs.rs:69-70
  const float aspect = (float)swapchain.width / (float)swapchain.height;      // :71
  const float fov_y = 70.0f * (3.14159265358979323846f / 180.0f);             // :72
  float* p = camera.projection;                                                // glm::perspective_lh_zo, :74
  std::memset(p, 0, 16 * sizeof(float));
  const float t = std::tan(fov_y / 2.0f);
  p[0] = 1.0f / (aspect * t);
  p[5] = 1.0f / t;
  p[10] = far_z / (far_z - near_z);
  p[14] = -(far_z * near_z) / (far_z - near_z);
  p[11] = 1.0f;
  const float fwd[3] = {0, 0, 1}, upv[3] = {0, 1, 0};                          // forward_vector / up_vector
  float dir[3], up[3];
  rotate(camera.rotation, fwd, dir);                                           // :76
  rotate(camera.rotation, upv, up);                                            // :78
  float z[3] = {dir[0], dir[1], dir[2]}, x[3], y[3];                           // glm::look_at_lh(eye, eye + dir, up), :80
  normalize(z);
  cross(up, z, x);
  normalize(x);
  cross(z, x, y);
  float* v = camera.view;
  std::memset(v, 0, 16 * sizeof(float));
  for (int a = 0; a < 3; ++a) {
    v[a * 4 + 0] = x[a];
    v[a * 4 + 1] = y[a];
    v[a * 4 + 2] = z[a];
  }
  v[12] = -dot3(x, camera.position);
  v[13] = -dot3(y, camera.position);
  v[14] = -dot3(z, camera.position);
  v[15] = 1.0f;
  float m[16];
  mat4_mul(camera.projection, camera.view, m);                                 // :82
  for (int k = 0; k < 3; ++k)                                                  // :83-90
    for (int c = 0; c < 4; ++c) {
      const float r3 = m[c * 4 + 3], rk = m[c * 4 + k];
      camera.frustum_planes[2 * k + 0][c] = -(r3 + rk);
      camera.frustum_planes[2 * k + 1][c] = -(r3 - rk);
    }
}

void assign_draw_index(World& world) {
  uint32_t counter = 0;
  for (auto& d : world.draw_index) d.v = counter++;
}

void model_matrix_calculation(World& world, const resources::Camera& camera, InstancePipeline& pipeline) {
  pipeline.ensure_frame(world, camera);
  if (world.len()) std::memcpy(world.model_matrix.data(), pipeline.model().data(), world.len() * sizeof(components::ModelMatrix));
}

void aabb_calculation(World& world, const resources::Camera& camera, InstancePipeline& pipeline) {
  pipeline.ensure_frame(world, camera);
  if (world.len()) std::memcpy(world.aabb.data(), pipeline.world_aabb().data(), world.len() * sizeof(components::AABB));
}

void coarse_culling(World& world, const resources::Camera& camera, InstancePipeline& pipeline) {
  pipeline.ensure_frame(world, camera);
  const auto& bits = pipeline.visible_bitmap();
  for (size_t e = 0; e < world.len(); ++e) world.coarse_culled[e].culled = !((bits[e >> 5] >> (e & 31)) & 1u);
}

void model_matrices_upload(const World& world, float* model_buffer_mapped) {
  for (size_t e = 0; e < world.len(); ++e)
    std::memcpy(model_buffer_mapped + (size_t)world.draw_index[e].v * 16, world.model_matrix[e].m, sizeof(float) * 16);
}

void cull_pass(World& world, const resources::Camera& camera, InstancePipeline& pipeline, IndirectCommands& out) {
  pipeline.ensure_frame(world, camera);
  const IndirectCommands& src = pipeline.indirect();
  out.count = src.count;
  out.commands.assign(src.commands.begin(), src.commands.begin() + src.count);
}

}  // namespace systems
}  // namespace ecs
}  // namespace renderer
