// Micro-benchmark: what the 5 % of bytes that are draw commands cost the byte mover, by store shape. Loads 36 B and
// 'sc1 nt' matrix stores as in the pipeline; then per wave 85 dwords of commands (17 x 20 B, v ~ 0.27) written
//   0: not at all                    1: as the pipeline does (dword stores, 64 + 21 lanes, 4-byte aligned start)
//   2: as 16-byte stores at a 16-byte aligned (padded) position       3: shape 2 with 'sc1 nt'
//   4: shape 1 with 'nt'
// and the visibility words on or off. Timed like floor_1m.hip.
// build: hipcc -O3 --offload-arch=gfx950 -o cmd_shapes cmd_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

struct Args { const float* pos; const float4* rot; const float* scale; const uint32_t* mesh; float4* model; uint32_t* bitmap; uint32_t* cmds; uint32_t n; };

template <int kCmd, bool kBitmap, bool kSmallFirst = false, bool kBitmapThrough = false>
__global__ __launch_bounds__(256) void mover(const Args a) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, tile = blockIdx.x;
  const uint32_t i = tile * 256u + tid;
  const uint32_t il = i < a.n ? i : a.n - 1u;
  const float px = a.pos[3 * (size_t)il], py = a.pos[3 * (size_t)il + 1], pz = a.pos[3 * (size_t)il + 2];
  const float4 q = a.rot[il]; const float sc = a.scale[il]; const uint32_t mesh = a.mesh[il];
  float4* out = a.model + ((size_t)tile * 256u + wave * 64u) * 4;
  const bool full = tile * 256u + wave * 64u + 64u <= a.n;
  if constexpr (kSmallFirst) {
    const unsigned long long vis = __ballot(px > 0.f);
    if (lane < 2u && i < a.n) a.bitmap[(tile * 256u >> 5) + wave * 2u + lane] = (uint32_t)(vis >> (32u * lane));
    if (full) {
      uint32_t* c = a.cmds + ((size_t)tile * 4u + wave) * 85u;
      c[lane] = __float_as_uint(px) + mesh;
      if (lane < 21u) c[64u + lane] = __float_as_uint(sc);
    }
  }
  if (full) {
    const v4f v0 = {px, py, pz, sc}, v1 = {q.x, q.y, q.z, q.w}, v2 = {q.w, q.z, q.y, q.x}, v3 = {sc, px, __uint_as_float(mesh), 1.f};
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n global_store_dwordx4 %0, %2, off offset:1024 sc1 nt\n"
                 "global_store_dwordx4 %0, %3, off offset:2048 sc1 nt\n global_store_dwordx4 %0, %4, off offset:3072 sc1 nt\n s_nop 1"
                 :: "v"(out + lane), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
  }
  if constexpr (kSmallFirst) return;
  if constexpr (kBitmap) {
    const unsigned long long vis = __ballot(px > 0.f);
    if (lane < 2u && i < a.n) {
      uint32_t* b = &a.bitmap[(tile * 256u >> 5) + wave * 2u + lane];
      const uint32_t w = (uint32_t)(vis >> (32u * lane));
      if constexpr (kBitmapThrough) asm volatile("global_store_dword %0, %1, off sc1\n s_nop 1" :: "v"(b), "v"(w) : "memory");
      else *b = w;
    }
  }
  if (!full) return;
  const uint32_t w0 = __float_as_uint(px) + mesh, w1 = __float_as_uint(sc);
  if constexpr (kCmd == 1) {
    uint32_t* c = a.cmds + ((size_t)tile * 4u + wave) * 85u;
    c[lane] = w0;
    if (lane < 21u) c[64u + lane] = w1;
  }
  if constexpr (kCmd == 4) {
    uint32_t* c = a.cmds + ((size_t)tile * 4u + wave) * 85u;
    __builtin_nontemporal_store(w0, &c[lane]);
    if (lane < 21u) __builtin_nontemporal_store(w1, &c[64u + lane]);
  }
  if constexpr (kCmd == 2 || kCmd == 3) {
    u4* c = reinterpret_cast<u4*>(a.cmds + ((size_t)tile * 4u + wave) * 96u);  // padded to a 16-byte multiple
    const u4 v = {w0, w1, w0 ^ w1, lane};
    if (lane < 22u) {
      if constexpr (kCmd == 2) c[lane] = v;
      else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n s_nop 1" :: "v"(c + lane), "v"(v) : "memory");
    }
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
    float4* model_all;
    CHECK(hipMalloc(&model_all, (size_t)(tiles + 4) * 256 * 64 + (size_t)(tiles + 4) * 32 + 256 + (size_t)(tiles + 4) * 4 * 96 * 4));
    a.model = model_all;
    CHECK(hipMalloc(&a.bitmap, (size_t)(tiles + 4) * 32));
    CHECK(hipMalloc(&a.cmds, (size_t)(tiles + 4) * 4 * 96 * 4));
    a.pos = pos; a.rot = rot; a.scale = scale; a.mesh = mesh; a.n = n;
    CHECK(hipDeviceSynchronize());
    struct V { const char* name; int id; };
    const V vs[] = {{"no commands, no bitmap", 0}, {"bitmap only", 1}, {"commands as dword stores (the pipeline's shape) + bitmap", 2},
                    {"commands as aligned 16-byte stores (padded) + bitmap", 3}, {"commands as aligned 16-byte 'sc1 nt' stores + bitmap", 4},
                    {"commands as 'nt' dword stores + bitmap", 5}, {"commands as dword stores, no bitmap", 6},
                    {"bitmap + commands (dword) issued BEFORE the matrix stores", 7},
                    {"bitmap only, written through (sc1): nothing left dirty in the L2", 8},
                    {"bitmap sc1 + commands aligned 16-byte 'sc1 nt': nothing left dirty", 9}};
    for (int alias = 0; alias < 2; ++alias) {
    // alias = 1: the visibility words and commands live at the end of the matrix allocation (same pages, same region)
    Args b = a;
    if (alias) { b.bitmap = reinterpret_cast<uint32_t*>(model_all + (size_t)(tiles + 4) * 256 * 4); b.cmds = b.bitmap + (size_t)(tiles + 4) * 8 + 64; }
    printf("--- %s\n", alias ? "visibility words and commands inside the matrix allocation" : "separate allocations");
    for (const V& v : vs) {
      const Args& a = b;
      auto launch = [&]() {
        switch (v.id) {
          case 0: hipLaunchKernelGGL((mover<0, false>), dim3(tiles), dim3(256), 0, st, a); break;
          case 1: hipLaunchKernelGGL((mover<0, true>), dim3(tiles), dim3(256), 0, st, a); break;
          case 2: hipLaunchKernelGGL((mover<1, true>), dim3(tiles), dim3(256), 0, st, a); break;
          case 3: hipLaunchKernelGGL((mover<2, true>), dim3(tiles), dim3(256), 0, st, a); break;
          case 4: hipLaunchKernelGGL((mover<3, true>), dim3(tiles), dim3(256), 0, st, a); break;
          case 5: hipLaunchKernelGGL((mover<4, true>), dim3(tiles), dim3(256), 0, st, a); break;
          case 6: hipLaunchKernelGGL((mover<1, false>), dim3(tiles), dim3(256), 0, st, a); break;
          case 7: hipLaunchKernelGGL((mover<1, true, true>), dim3(tiles), dim3(256), 0, st, a); break;
          case 8: hipLaunchKernelGGL((mover<0, true, false, true>), dim3(tiles), dim3(256), 0, st, a); break;
          case 9: hipLaunchKernelGGL((mover<3, true, false, true>), dim3(tiles), dim3(256), 0, st, a); break;
        }
      };
      const int K = n >= 10000000u ? 40 : 200;
      for (int k = 0; k < 20; ++k) launch();
      CHECK(hipStreamSynchronize(st));
      std::vector<float> samples;
      for (int r = 0; r < 7; ++r) {
        CHECK(hipEventRecord(e0, st));
        for (int k = 0; k < K; ++k) launch();
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        samples.push_back(ms / K * 1e3f);
      }
      std::sort(samples.begin(), samples.end());
      printf("n=%-9u %-60s median %7.2f us  min %7.2f us\n", n, v.name, samples[3], samples[0]);
    }
    }
    CHECK(hipFree(pos)); CHECK(hipFree(rot)); CHECK(hipFree(scale)); CHECK(hipFree(mesh));
    CHECK(hipFree(a.model)); CHECK(hipFree(a.bitmap)); CHECK(hipFree(a.cmds));
  }
  return 0;
}
