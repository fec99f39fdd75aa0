// gltf_extract.cpp — CLI: glTF/GLB -> scene.bin for mip_frame_driver / the Python tests.
//   mip_gltf_extract <scene.gltf|.glb> <scene.bin> [replicate]
// scene.bin: u32 n, u32 m, m x MipMesh, n x pos, n x rot, n x scale, n x mesh id,
//            then u32 n_vertices, u32 n_indices, vertices, indices (ignored by older readers).
// `replicate` > 1 lays copies of the scene's entities out on a grid (an N-instance scene from one asset).
#include <cstdio>
#include <cstdlib>

#include "gltf_scene.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s scene.gltf scene.bin [replicate]\n", argv[0]);
    return 2;
  }
  try {
    renderer::gltf::Scene s = renderer::gltf::load(argv[1]);
    const uint32_t base_n = (uint32_t)s.scale.size();
    const uint32_t copies = argc > 3 ? (uint32_t)atoi(argv[3]) : 1;
    for (uint32_t c = 1; c < copies; ++c)
      for (uint32_t e = 0; e < base_n; ++e) {
        const float dx = 4.0f * (float)(c % 32), dz = 4.0f * (float)(c / 32);
        s.pos_xyz.push_back(s.pos_xyz[e * 3 + 0] + dx);
        s.pos_xyz.push_back(s.pos_xyz[e * 3 + 1]);
        s.pos_xyz.push_back(s.pos_xyz[e * 3 + 2] + dz);
        for (int k = 0; k < 4; ++k) s.rot_ijkw.push_back(s.rot_ijkw[e * 4 + k]);
        s.scale.push_back(s.scale[e]);
        s.mesh_id.push_back(s.mesh_id[e]);
      }
    const uint32_t n = (uint32_t)s.scale.size(), m = (uint32_t)s.meshes.size();
    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror("out"); return 2; }
    // an empty vector's data() may be null, which fwrite must not be handed (found by the sanitizer fuzz)
    auto put = [f](const void* p, size_t size, size_t count) {
      if (count) fwrite(p, size, count, f);
    };
    put(&n, 4, 1);
    put(&m, 4, 1);
    put(s.meshes.data(), sizeof(MipMesh), m);
    put(s.pos_xyz.data(), 4, s.pos_xyz.size());
    put(s.rot_ijkw.data(), 4, s.rot_ijkw.size());
    put(s.scale.data(), 4, s.scale.size());
    put(s.mesh_id.data(), 4, s.mesh_id.size());
    const uint32_t nv = (uint32_t)(s.vertices.size() / 3), ni = (uint32_t)s.indices.size();
    put(&nv, 4, 1);
    put(&ni, 4, 1);
    put(s.vertices.data(), 4, s.vertices.size());
    put(s.indices.data(), 4, s.indices.size());
    fclose(f);
    printf("entities=%u meshes=%u primitives=%u skipped_no_base_color=%u skipped_small=%u vertices=%u indices=%u\n", n, m,
           s.primitives_seen, s.skipped_no_base_color, s.skipped_small, nv, ni);
  } catch (const std::exception& e) {
    fprintf(stderr, "gltf error: %s\n", e.what());
    return 3;
  }
  return 0;
}
