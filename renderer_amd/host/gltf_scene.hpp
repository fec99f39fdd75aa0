// gltf_scene.hpp — glTF 2.0 -> instance columns + mesh table + consolidated geometry, the
// upstream side of the instance path. Mirrors what the reference's
// scene loader extracts (src/renderer/systems/scene_loader.rs:100-145 traversal, :642-789
// visit_node): every scene's root nodes are walked depth-first; each mesh primitive that has a
// base-colour texture (:659-667) and at least 100 positions (:677-679) becomes ONE entity with
//   Position/Rotation/Scale = the node's LOCAL transform decomposed, scale[0] only (:760,:765);
//                             parents are NOT accumulated (:786-788 visits children with no transform)
//   GltfMesh.aabb            = the POSITION accessor's min/max (primitive.bounding_box(), :694-698)
//   index LODs               = LOD 0 = the primitive's indices; LOD x = 1..5 targets
//                              len * 0.5^x indices (:741-751). The reference runs meshopt's
//                              simplify_sloppy, which is not available here: LOD x keeps the
//                              target count rounded down to whole triangles, taken as an
//                              evenly spaced subset of LOD 0 (documented stand-in).
// No mesh de-duplication (the reference's cache is commented out, :644), so mesh id = entity id.
// Self-contained: own JSON reader, base64 data URIs, external .bin files and .glb containers.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi_instance_pipeline.h"

namespace renderer {
namespace gltf {

struct Scene {
  // instance columns, entity order = traversal order
  std::vector<float> pos_xyz, rot_ijkw, scale;
  std::vector<uint32_t> mesh_id;
  std::vector<MipMesh> meshes;
  // consolidated geometry (ConsolidatedMeshBuffers): packed vec3 positions, u32 indices
  std::vector<float> vertices;
  std::vector<uint32_t> indices;
  // bookkeeping
  uint32_t primitives_seen = 0, skipped_no_base_color = 0, skipped_small = 0;
  std::vector<std::string> entity_names;
};

// Throws std::runtime_error with a message on malformed input.
Scene load(const std::string& path);

}  // namespace gltf
}  // namespace renderer
