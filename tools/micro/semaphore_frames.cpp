// semaphore_frames.cpp — what the external-semaphore hand-over of row f-2 costs per frame.
//
// The reference orders ComputeCull against the graphics queue with timeline semaphores (src/renderer.rs:3757-3861). ROCm 7.2
// refuses to import them on the device, so the library waits for / signals the DRM sync object behind the exported fd from host
// functions on the frame's stream (api_interop.hip). This program drives frames from compiled code (no Python between the calls)
// against two kernel timeline sync objects — standing in for the renderer's exported semaphores — and reports microseconds per frame:
//
//   bare          mip_run(ASYNC) back to back, one mip_wait at the end                       (the 18.5 us of the headline)
//   free_running  mip_wait_external(consumers, k) -> mip_run(ASYNC) -> mip_signal_external(cull, k + 1) with the consumer
//                 timeline already far ahead: the waits never block — the cost of the two host functions per frame
//   ping_pong     the same, but a "renderer" thread advances the consumer timeline to k + 1 only when it has SEEN cull reach
//                 k + 1 (DRM_IOCTL_SYNCOBJ_TIMELINE_WAIT): every frame waits for the consumer of the previous one — the full
//                 round trip GPU -> host function -> ioctl -> renderer thread -> ioctl -> host function -> GPU
//
// Built by renderer_amd/host/Makefile into renderer_amd/lib/mip_semaphore_bench (g++, the C ABI header, libamdhip64 for hipMalloc).
// Output: one JSON object on stdout.
#include "mi_instance_pipeline.h"

#include <hip/hip_runtime_api.h>

#include <drm/drm.h>
#include <fcntl.h>
#include <sys/ioctl.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    const int rc_ = (x);                                                                           \
    if (rc_ != 0) {                                                                                \
      std::fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, ctx ? mip_last_error(ctx) : "");       \
      return 1;                                                                                    \
    }                                                                                              \
  } while (0)

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct SyncObj {
  int fd = -1;
  uint32_t handle = 0;
  bool create() {
    char node[64];
    for (int k = 128; k < 192 && fd < 0; ++k) {
      std::snprintf(node, sizeof node, "/dev/dri/renderD%d", k);
      const int f = open(node, O_RDWR | O_CLOEXEC);
      if (f < 0) continue;
      drm_syncobj_create c{};
      if (ioctl(f, DRM_IOCTL_SYNCOBJ_CREATE, &c) == 0) {
        fd = f;
        handle = c.handle;
      } else {
        close(f);
      }
    }
    return fd >= 0;
  }
  int export_fd() const {
    drm_syncobj_handle h{};
    h.handle = handle;
    h.fd = -1;
    return ioctl(fd, DRM_IOCTL_SYNCOBJ_HANDLE_TO_FD, &h) == 0 ? h.fd : -1;
  }
  void signal(uint64_t value) const {
    uint32_t hd = handle;
    drm_syncobj_timeline_array a{};
    a.handles = (uintptr_t)&hd;
    a.points = (uintptr_t)&value;
    a.count_handles = 1;
    ioctl(fd, DRM_IOCTL_SYNCOBJ_TIMELINE_SIGNAL, &a);
  }
  bool wait(uint64_t value, int64_t timeout_ns) const {
    timespec now;
    clock_gettime(CLOCK_MONOTONIC, &now);
    uint32_t hd = handle;
    drm_syncobj_timeline_wait w{};
    w.handles = (uintptr_t)&hd;
    w.points = (uintptr_t)&value;
    w.timeout_nsec = (int64_t)now.tv_sec * 1000000000ll + now.tv_nsec + timeout_ns;
    w.count_handles = 1;
    w.flags = DRM_SYNCOBJ_WAIT_FLAGS_WAIT_FOR_SUBMIT;
    return ioctl(fd, DRM_IOCTL_SYNCOBJ_TIMELINE_WAIT, &w) == 0;
  }
};

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 10) : 1000000u;
  const uint32_t frames = argc > 2 ? (uint32_t)std::strtoul(argv[2], nullptr, 10) : 2000u;
  MipContext* ctx = nullptr;
  MipConfig cfg{};
  cfg.struct_size = sizeof cfg;
  cfg.max_instances = n;
  cfg.max_meshes = 4;
  cfg.frames_in_flight = 1;
  CHECK(mip_create(&cfg, &ctx));
  // a synthetic scene (the hand-over's cost does not depend on what the frame computes): a jittered grid of unit-scale instances of
  // four meshes, a box-shaped "frustum" that keeps about a quarter of them
  MipMesh meshes[4]{};
  for (int m = 0; m < 4; ++m) {
    for (int a = 0; a < 3; ++a) { meshes[m].aabb_min[a] = -0.5f - 0.1f * m; meshes[m].aabb_max[a] = 0.5f + 0.1f * m; }
    meshes[m].n_lods = 2;
    meshes[m].index_len[0] = 36u * (m + 1);
    meshes[m].index_len[1] = 12u * (m + 1);
    meshes[m].index_offset[1] = 36u * (m + 1);
    meshes[m].vertex_offset = 24 * m;
  }
  CHECK(mip_set_mesh_table(ctx, meshes, 4));
  std::vector<float> pos(3 * (size_t)n), rot(4 * (size_t)n), scale(n);
  std::vector<uint32_t> mesh(n);
  uint64_t x = 0x5EED0000ull;
  auto u01 = [&]() { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; return (float)(z >> 40) * (1.0f / 16777216.0f); };
  for (uint32_t i = 0; i < n; ++i) {
    pos[3 * i] = 128.f * u01() - 64.f; pos[3 * i + 1] = 64.f * u01() - 32.f; pos[3 * i + 2] = 128.f * u01() - 64.f;
    rot[4 * i] = 0.f; rot[4 * i + 1] = 0.f; rot[4 * i + 2] = 0.f; rot[4 * i + 3] = 1.f;
    scale[i] = 0.5f + 1.5f * u01();
    mesh[i] = i & 3u;
  }
  CHECK(mip_set_instances(ctx, pos.data(), rot.data(), scale.data(), mesh.data(), n));
  MipFrame frame{};
  const float box[6][4] = {{-1, 0, 0, -32}, {1, 0, 0, -32}, {0, -1, 0, -16}, {0, 1, 0, -16}, {0, 0, -1, -32}, {0, 0, 1, -32}};
  std::memcpy(frame.planes, box, sizeof box);
  frame.cam_pos[1] = 1.f; frame.cam_pos[2] = 2.f;
  void *d_model = nullptr, *d_cmds = nullptr;
  uint32_t *d_bitmap = nullptr, *d_scal = nullptr;
  if (hipMalloc(&d_model, (size_t)n * 64) || hipMalloc(&d_cmds, (size_t)n * 20) || hipMalloc((void**)&d_bitmap, ((size_t)n + 31) / 32 * 4 + 64) ||
      hipMalloc((void**)&d_scal, 64)) {
    std::fprintf(stderr, "hipMalloc failed\n");
    return 1;
  }
  MipOutputs out{};
  out.model = d_model; out.visible_bitmap = d_bitmap; out.draw_cmds = d_cmds; out.draw_count = d_scal; out.draw_index_total = d_scal + 1;
  out.flags = MIP_OUT_DEVICE | MIP_OUT_ASYNC;

  SyncObj cull, consumers;
  if (!cull.create() || !consumers.create()) {
    std::printf("{\"available\": false, \"why\": \"no DRM render node allows DRM_IOCTL_SYNCOBJ_CREATE\"}\n");
    return 0;
  }
  MipExternalSemaphore *s_cull = nullptr, *s_cons = nullptr;
  CHECK(mip_import_external_semaphore_fd(ctx, cull.export_fd(), MIP_SEMAPHORE_TIMELINE, &s_cull));
  CHECK(mip_import_external_semaphore_fd(ctx, consumers.export_fd(), MIP_SEMAPHORE_TIMELINE, &s_cons));
  const int on_device = mip_external_semaphore_on_device(ctx, s_cull);

  // ---- bare ----
  for (uint32_t k = 0; k < 200; ++k) CHECK(mip_run(ctx, &frame, &out));
  CHECK(mip_wait(ctx));
  double t0 = now_us();
  for (uint32_t k = 0; k < frames; ++k) CHECK(mip_run(ctx, &frame, &out));
  CHECK(mip_wait(ctx));
  const double bare = (now_us() - t0) / frames;
  uint32_t count = 0;
  (void)hipMemcpy(&count, d_scal, 4, hipMemcpyDeviceToHost);

  // ---- free running: the consumer timeline is far ahead, no wait ever blocks ----
  uint64_t v = 0;  // value the cull timeline has been asked to reach
  consumers.signal(1ull << 40);
  t0 = now_us();
  for (uint32_t k = 0; k < frames; ++k) {
    CHECK(mip_wait_external(ctx, s_cons, v));
    CHECK(mip_run(ctx, &frame, &out));
    CHECK(mip_signal_external(ctx, s_cull, ++v));
  }
  CHECK(mip_wait(ctx));
  const double free_running = (now_us() - t0) / frames;

  // ---- ping-pong with a renderer thread ----
  SyncObj cull2, cons2;  // fresh timelines starting at 0
  if (!cull2.create() || !cons2.create()) return 1;
  MipExternalSemaphore *s_cull2 = nullptr, *s_cons2 = nullptr;
  CHECK(mip_import_external_semaphore_fd(ctx, cull2.export_fd(), MIP_SEMAPHORE_TIMELINE, &s_cull2));
  CHECK(mip_import_external_semaphore_fd(ctx, cons2.export_fd(), MIP_SEMAPHORE_TIMELINE, &s_cons2));
  const uint32_t pp_frames = frames < 500u ? frames : 500u;
  std::atomic<bool> failed{false};
  std::thread renderer([&]() {  // consumes frame k (sees cull reach k) and only then lets frame k + 1 overwrite the buffers
    for (uint64_t k = 1; k <= pp_frames; ++k) {
      if (!cull2.wait(k, 10ll * 1000000000ll)) { failed = true; return; }
      cons2.signal(k);
    }
  });
  t0 = now_us();
  for (uint64_t k = 0; k < pp_frames; ++k) {
    if (k) CHECK(mip_wait_external(ctx, s_cons2, k));  // (nothing has read the buffers before the first frame)
    CHECK(mip_run(ctx, &frame, &out));
    CHECK(mip_signal_external(ctx, s_cull2, k + 1));
  }
  const int rc_pp = mip_wait(ctx);
  const double ping_pong = (now_us() - t0) / pp_frames;
  renderer.join();
  if (rc_pp != MIP_OK) { std::fprintf(stderr, "mip_wait after the ping-pong: %d (%s)\n", rc_pp, mip_last_error(ctx)); return 1; }

  std::printf("{\"available\": true, \"instances\": %u, \"draw_count\": %u, \"frames\": %u, \"semaphores_on_device\": %s, "
              "\"hand_over\": \"%s\", \"bare_us_per_frame\": %.2f, \"free_running_us_per_frame\": %.2f, "
              "\"ping_pong_us_per_frame\": %.2f, \"ping_pong_frames\": %u, \"ping_pong_ok\": %s}\n",
              n, count, frames, on_device == 1 ? "true" : "false",
              on_device == 1 ? "HIP runtime external semaphores" : (std::getenv("MIP_TUNE_SEMAPHORE_HOST_FUNCTIONS") ? "two hipLaunchHostFunc per frame (round 3)" : "hipStreamWaitValue64 / hipStreamWriteValue64 + two helper threads"),
              bare, free_running, ping_pong, pp_frames,
              failed ? "false" : "true");
  mip_destroy(ctx);
  return failed ? 1 : 0;
}
