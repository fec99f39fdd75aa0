// simplify_sloppy.hpp — LOD generation of the reference's scene loader (scene_loader.rs:739-753): a restatement of
// meshoptimizer's meshopt_simplifySloppy, which the loader calls through the `meshopt 0.1.9` crate. See the .cpp.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace renderer {
namespace gltf {

// indices: whole triangles of one primitive (a trailing partial triangle is ignored); positions_xyz: vertex_count packed
// vec3; target_index_count: what the loader passes, `(indices.len() as f32 * 0.5^x) as usize` (not necessarily a
// multiple of 3). Returns the simplified index list: at most target_index_count indices, possibly far fewer, possibly
// none. Every index must be < vertex_count.
std::vector<uint32_t> simplify_sloppy(const std::vector<uint32_t>& indices, const float* positions_xyz, size_t vertex_count,
                                      size_t target_index_count);

}  // namespace gltf
}  // namespace renderer
