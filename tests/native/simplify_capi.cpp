// simplify_capi.cpp — TEST-ONLY C entry point over renderer_amd/host/simplify_sloppy.cpp so that tests/test_simplify_properties.py
// can call the loader's LOD simplifier directly (the product reaches it only through the glTF extractor).
#include "../../renderer_amd/host/simplify_sloppy.hpp"

#include <cstring>

extern "C" size_t mip_test_simplify_sloppy(const uint32_t* indices, size_t index_count, const float* positions_xyz, size_t vertex_count,
                                           size_t target_index_count, uint32_t* out, size_t out_capacity) {
  const std::vector<uint32_t> in(indices, indices + index_count);
  const std::vector<uint32_t> res = renderer::gltf::simplify_sloppy(in, positions_xyz, vertex_count, target_index_count);
  if (res.size() <= out_capacity && !res.empty()) std::memcpy(out, res.data(), res.size() * sizeof(uint32_t));
  return res.size();
}
