// frame_driver.cpp — runs the reference's per-frame schedule (src/main.rs:780-839 RenderSetup,
// then cull_pass) over a scene file through the C++ mirror, and dumps every component the
// systems write. Used by tests/test_gpu_host_mirror.py.
//
//   mip_frame_driver <scene.bin> <out.bin> [frames]
//
// scene.bin : u32 n, u32 m, then m x MipMesh (80 B), n x Position, n x Rotation, n x Scale, n x u32 mesh
// out.bin   : u32 n, u32 count, u32 index_total, 24 planes, n x mat4 (mapped model buffer, by draw_index),
//             n x AABB, n x u8 coarse_culled, count x VkDrawIndexedIndirectCommand
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ecs.hpp"

using namespace renderer::ecs;

static bool read_all(FILE* f, void* p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s scene.bin out.bin [frames]\n", argv[0]);
    return 2;
  }
  const int frames = argc > 3 ? atoi(argv[3]) : 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("scene"); return 2; }
  uint32_t n = 0, m = 0;
  if (!read_all(f, &n, 4) || !read_all(f, &m, 4)) return 2;
  std::vector<MipMesh> library(m);
  std::vector<components::Position> pos(n);
  std::vector<components::Rotation> rot(n);
  std::vector<components::Scale> scl(n);
  std::vector<uint32_t> mesh(n);
  if (!read_all(f, library.data(), (size_t)m * sizeof(MipMesh)) || !read_all(f, pos.data(), (size_t)n * 12) ||
      !read_all(f, rot.data(), (size_t)n * 16) || !read_all(f, scl.data(), (size_t)n * 4) || !read_all(f, mesh.data(), (size_t)n * 4)) {
    fprintf(stderr, "short scene file\n");
    return 2;
  }
  fclose(f);
  try {
    World world;
    for (uint32_t e = 0; e < n; ++e) world.spawn(pos[e], rot[e], scl[e], components::GltfMesh{mesh[e]});
    InstancePipeline pipeline(n ? n : 1, library);
    resources::Swapchain swapchain;
    resources::Camera camera;
    IndirectCommands indirect;
    std::vector<float> model_buffer((size_t)(n ? n : 1) * 16, 0.0f);  // the mapped `mat4 model[]`
    for (int frame = 0; frame < frames; ++frame) {
      systems::project_camera(swapchain, camera);                        // Gameplay set
      systems::assign_draw_index(world);                                 // RenderSetup
      systems::model_matrix_calculation(world, camera, pipeline);
      systems::aabb_calculation(world, camera, pipeline);                // after ModelMatrixCalculation
      systems::coarse_culling(world, camera, pipeline);                  // after AABBCalculation
      systems::model_matrices_upload(world, model_buffer.data());        // after AssignDrawIndex + MMC
      systems::cull_pass(world, camera, pipeline, indirect);             // stage "graphics work"
      pipeline.end_frame();
    }
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror("out"); return 2; }
    const uint32_t total = pipeline.draw_index_total();
    fwrite(&n, 4, 1, o);
    fwrite(&indirect.count, 4, 1, o);
    fwrite(&total, 4, 1, o);
    fwrite(camera.frustum_planes, sizeof(float), 24, o);
    fwrite(model_buffer.data(), sizeof(float), (size_t)n * 16, o);
    fwrite(world.aabb.data(), sizeof(components::AABB), n, o);
    for (uint32_t e = 0; e < n; ++e) fputc(world.coarse_culled[e].culled ? 1 : 0, o);
    fwrite(indirect.commands.data(), sizeof(MipDrawIndexedIndirectCommand), indirect.count, o);
    fclose(o);
    printf("frames=%d n=%u draw_count=%u\n", frames, n, indirect.count);
  } catch (const renderer::Error& e) {
    fprintf(stderr, "renderer error %d: %s\n", e.code, e.what());
    return 10 + (e.code < 0 ? -e.code : e.code);
  }
  return 0;
}
