// ecs.hpp — C++ host-side mirror of the reference's ECS interface for the instance path,
// above the C ABI (include/mi_instance_pipeline.h). The reference is Rust (bevy_ecs); this
// image has no rustc, so the host layer that a maintainer would write in Rust is written in
// C++ with the same names and argument meaning:
//
//   components   Position / Rotation / Scale / ModelMatrix / AABB      src/ecs/components.rs:5-23
//                DrawIndex (src/renderer.rs:148-149), CoarseCulled (cull_pipeline.rs:67-68),
//                GltfMesh (src/renderer.rs:117-126; here: an id into the mesh library)
//   resources    Camera (src/ecs/camera_controller.rs:9-17), Swapchain (width/height only)
//   systems      project_camera (ecs.rs:66-91), assign_draw_index (ecs.rs:117-136),
//                model_matrix_calculation (ecs.rs:52-64), aabb_calculation (ecs.rs:138-181),
//                coarse_culling (cull_pipeline.rs:99-120), model_matrices_upload
//                (renderer.rs:2266-2288), cull_pass (cull_pipeline.rs:423-616)
//
// The three per-entity systems, the upload and the draw-command emission are ONE kernel
// launch on the device. The mirror keeps the reference's schedule (src/main.rs:780-839): the
// first of those systems to run in a frame launches the fused pipeline, each system then
// scatters its own outputs into the components it owns in the reference. Errors are status
// codes surfaced as `Error` (the reference panics, with panic = "abort").
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi_instance_pipeline.h"

namespace renderer {

struct Error : std::runtime_error {
  int32_t code;
  Error(int32_t c, const std::string& what) : std::runtime_error(what), code(c) {}
};

namespace ecs {
namespace components {
struct Position { float x, y, z; };               // Position(na::Point3<f32>)
struct Rotation { float i, j, k, w; };            // Rotation(na::UnitQuaternion<f32>), coords [i,j,k,w]
struct Scale { float s; };                        // Scale(f32)
struct ModelMatrix { float m[16]; };              // ModelMatrix(glm::Mat4), column-major
struct AABB { float mins[3], maxs[3]; };          // AABB(ncollide3d AABB<f32>)
struct GltfMesh { uint32_t mesh; };               // index into MeshLibrary (stands in for the buffer handles)
}  // namespace components

struct DrawIndex { uint32_t v; };                 // renderer::DrawIndex(u32)
struct CoarseCulled { bool culled; };             // renderer::CoarseCulled(bool)

namespace resources {
struct Swapchain { uint32_t width = 2000, height = 1000; };  // src/renderer/instance.rs:45
struct Camera {                                                // camera_controller.rs:9-35
  float position[3] = {0.0f, 1.0f, 2.0f};
  float rotation[4] = {0.0f, 0.0f, 0.0f, 1.0f};  // [i,j,k,w]
  float projection[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float frustum_planes[6][4] = {};               // left, right, bottom, top, near, far
};
}  // namespace resources

// One archetype's table: every entity has all of these components (the query of cull_pass,
// cull_pipeline.rs:435). Columns are contiguous, as bevy stores them.
struct World {
  std::vector<components::Position> position;
  std::vector<components::Rotation> rotation;
  std::vector<components::Scale> scale;
  std::vector<components::GltfMesh> mesh;
  std::vector<components::ModelMatrix> model_matrix;
  std::vector<components::AABB> aabb;
  std::vector<DrawIndex> draw_index;
  std::vector<CoarseCulled> coarse_culled;
  bool changed = true;  // Changed<Position|Rotation|Scale|GltfMesh>: re-upload the columns

  size_t len() const { return position.size(); }
  size_t spawn(components::Position p, components::Rotation r, components::Scale s, components::GltfMesh m);
};

// IndirectCommandsBuffer / IndirectCommandsCount (cull_pipeline.rs:70-72), host copies.
struct IndirectCommands {
  std::vector<MipDrawIndexedIndirectCommand> commands;  // [0, count) valid
  uint32_t count = 0;
};

// Owns the MipContext: the counterpart of CullPassData + ModelData for this path.
class InstancePipeline {
 public:
  InstancePipeline(uint32_t max_instances, const std::vector<MipMesh>& mesh_library, int device = 0);
  ~InstancePipeline();
  InstancePipeline(const InstancePipeline&) = delete;
  InstancePipeline& operator=(const InstancePipeline&) = delete;

  // Runs the fused device pipeline for this frame if it has not run yet.
  void ensure_frame(World& world, const resources::Camera& camera);
  void end_frame() { frame_valid_ = false; }

  const std::vector<float>& model() const { return model_; }
  const std::vector<float>& world_aabb() const { return aabb_; }
  const std::vector<uint32_t>& visible_bitmap() const { return bitmap_; }
  const IndirectCommands& indirect() const { return indirect_; }
  uint32_t draw_index_total() const { return index_total_; }

 private:
  void check(int32_t rc, const char* what) const;
  MipContext* ctx_ = nullptr;
  uint32_t n_meshes_ = 0;
  bool frame_valid_ = false;
  std::vector<float> model_, aabb_;
  std::vector<uint32_t> bitmap_;
  IndirectCommands indirect_;
  uint32_t index_total_ = 0;
};

namespace systems {
// src/ecs.rs:66-91. Host arithmetic (once per frame); produces Camera.frustum_planes.
void project_camera(const resources::Swapchain& swapchain, resources::Camera& camera);
// src/ecs.rs:117-136: sequential counter in query order.
void assign_draw_index(World& world);
// src/ecs.rs:52-64 / :138-181 / cull_pipeline.rs:99-120: fill ModelMatrix / AABB / CoarseCulled.
void model_matrix_calculation(World& world, const resources::Camera& camera, InstancePipeline& pipeline);
void aabb_calculation(World& world, const resources::Camera& camera, InstancePipeline& pipeline);
void coarse_culling(World& world, const resources::Camera& camera, InstancePipeline& pipeline);
// src/renderer.rs:2266-2288: model[draw_index] = ModelMatrix into the caller's `mat4 model[]` storage.
void model_matrices_upload(const World& world, float* model_buffer_mapped);
// cull_pipeline.rs:423-616: the compacted draw stream + count for vkCmdDrawIndexedIndirectCount.
void cull_pass(World& world, const resources::Camera& camera, InstancePipeline& pipeline, IndirectCommands& out);
}  // namespace systems
}  // namespace ecs
}  // namespace renderer
