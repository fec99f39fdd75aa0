// Micro-benchmark: does the cache-policy encoding of the 16-byte matrix stores matter? The mover of floor_1m.hip at
// 1 M instances (loads 36 B, matrix stores 64 B per instance as lane-contiguous 1 KiB per instruction), the stores
// written in inline assembly with every combination of the gfx950 bits sc0 / sc1 / nt. Timed like floor_1m.hip.
// build: hipcc -O3 --offload-arch=gfx950 -o store_bits store_bits.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int kBits>
__device__ __forceinline__ void store16(float4* p, v4f v) {
  if constexpr (kBits == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if constexpr (kBits == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (kBits == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if constexpr (kBits == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (kBits == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if constexpr (kBits == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (kBits == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  if constexpr (kBits == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

struct Args { const float* pos; const float4* rot; const float* scale; const uint32_t* mesh; float4* model; uint32_t n; };

template <int kBits, bool kLoads>
__global__ __launch_bounds__(256) void mover(const Args a) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, tile = blockIdx.x;
  const uint32_t i = tile * 256u + tid;
  const uint32_t il = i < a.n ? i : a.n - 1u;
  float px = 1, py = 2, pz = 3, sc = 4; float4 q = make_float4(0, 0, 0, 1); uint32_t mesh = 0;
  if constexpr (kLoads) {
    px = a.pos[3 * (size_t)il]; py = a.pos[3 * (size_t)il + 1]; pz = a.pos[3 * (size_t)il + 2];
    q = a.rot[il]; sc = a.scale[il]; mesh = a.mesh[il];
  }
  float4* out = a.model + ((size_t)tile * 256u + wave * 64u) * 4;
  if (tile * 256u + wave * 64u + 64u <= a.n) {
    store16<kBits>(&out[lane], (v4f){px, py, pz, sc});
    store16<kBits>(&out[64 + lane], (v4f){q.x, q.y, q.z, q.w});
    store16<kBits>(&out[128 + lane], (v4f){q.w, q.z, q.y, q.x});
    store16<kBits>(&out[192 + lane], (v4f){sc, px, __uint_as_float(mesh), 1.f});
  }
}

template <int kBits>
static void run(const Args& a, uint32_t tiles, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const char* name) {
  for (int loads = 0; loads < 2; ++loads) {
    auto launch = [&]() {
      if (loads) hipLaunchKernelGGL((mover<kBits, true>), dim3(tiles), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((mover<kBits, false>), dim3(tiles), dim3(256), 0, st, a);
    };
    for (int k = 0; k < 20; ++k) launch();
    CHECK(hipStreamSynchronize(st));
    std::vector<float> samples;
    for (int r = 0; r < 7; ++r) {
      CHECK(hipEventRecord(e0, st));
      for (int k = 0; k < 200; ++k) launch();
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      samples.push_back(ms / 200 * 1e3f);
    }
    std::sort(samples.begin(), samples.end());
    const double bytes = a.n * (loads ? 100.0 : 64.0);
    printf("n=%u stores [%-10s] %-12s median %6.2f us  min %6.2f us  %.2f TB/s\n", a.n, name, loads ? "loads+stores" : "stores only", samples[3], samples[0],
           bytes / (samples[3] * 1e-6) / 1e12);
  }
}

int main() {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (uint32_t n : {1000000u, 10000000u}) {
    const uint32_t tiles = (n + 255) / 256;
    Args a{};
    float* pos; float4* rot; float* scale; uint32_t* mesh;
    CHECK(hipMalloc(&pos, (size_t)n * 12)); CHECK(hipMalloc(&rot, (size_t)n * 16));
    CHECK(hipMalloc(&scale, (size_t)n * 4)); CHECK(hipMalloc(&mesh, (size_t)n * 4));
    CHECK(hipMemset(pos, 0x3f, (size_t)n * 12)); CHECK(hipMemset(rot, 0x3f, (size_t)n * 16));
    CHECK(hipMemset(scale, 0x3f, (size_t)n * 4)); CHECK(hipMemset(mesh, 0, (size_t)n * 4));
    CHECK(hipMalloc(&a.model, (size_t)(tiles + 4) * 256 * 64));
    a.pos = pos; a.rot = rot; a.scale = scale; a.mesh = mesh; a.n = n;
    CHECK(hipDeviceSynchronize());
    run<0>(a, tiles, st, e0, e1, "");
    run<1>(a, tiles, st, e0, e1, "nt");
    run<2>(a, tiles, st, e0, e1, "sc0");
    run<3>(a, tiles, st, e0, e1, "sc0 nt");
    run<4>(a, tiles, st, e0, e1, "sc1");
    run<5>(a, tiles, st, e0, e1, "sc1 nt");
    run<6>(a, tiles, st, e0, e1, "sc0 sc1");
    run<7>(a, tiles, st, e0, e1, "sc0 sc1 nt");
    CHECK(hipFree(pos)); CHECK(hipFree(rot)); CHECK(hipFree(scale)); CHECK(hipFree(mesh)); CHECK(hipFree(a.model));
  }
  return 0;
}
