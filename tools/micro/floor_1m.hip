// Micro-benchmark: the floor for ONE launch that moves the instance pipeline's bytes at 1 M instances (and 100 k / 10 M),
// with the pipeline's own access shapes but no arithmetic and no dependency between workgroups:
//   loads  : pos dwordx3, quat dwordx4, scale dword, mesh id dword (SoA, 36 B per instance)
//   stores : mat4 as 4 store instructions per wave, each 1 KiB contiguous (what the LDS transpose buys the real kernel),
//            one visibility word per 32 instances, 20 B per emitted command at a position known up front (v ~ 0.27)
// Variants: one instance per thread (the real grid), two / four instances per thread, a persistent grid-stride loop,
// non-temporal matrix stores, loads only, stores only. Timed with hipEvents over back-to-back launches on one stream
// (the way bench.py times the serialized kernel), so every figure includes the dependent-launch gap.
// build: hipcc -O3 --offload-arch=gfx950 -o floor_1m floor_1m.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const float* pos; const float4* rot; const float* scale; const uint32_t* mesh;
  float4* model; uint32_t* bitmap; uint32_t* cmds; uint32_t n;
};

// kNT: 0 = plain, 1 = non-temporal, 2 = 'sc1 nt' (streamed and written through at device scope; what the pipeline uses)
template <int kNT>
__device__ __forceinline__ void store16(float4* p, float4 v) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  if constexpr (kNT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n s_nop 1" :: "v"(p), "v"((v4f){v.x, v.y, v.z, v.w}) : "memory");
  else if constexpr (kNT == 1) __builtin_nontemporal_store((v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
  else *p = v;
}

// one tile of 256 instances handled by the 256 threads of a workgroup
template <bool kLoads, bool kStores, int kNT>
__device__ __forceinline__ void move_tile(const Args& a, uint32_t tile) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t i = tile * 256u + tid;
  const uint32_t il = i < a.n ? i : a.n - 1u;
  float px = 0, py = 0, pz = 0, sc = 0; float4 q = make_float4(0, 0, 0, 1); uint32_t mesh = 0;
  if constexpr (kLoads) {
    px = a.pos[3 * (size_t)il]; py = a.pos[3 * (size_t)il + 1]; pz = a.pos[3 * (size_t)il + 2];
    q = a.rot[il]; sc = a.scale[il]; mesh = a.mesh[il];
  }
  if constexpr (kStores) {
    // lane-contiguous 16-B stores: instruction s4 of a wave writes 1 KiB; the VALUES are whatever this lane has
    float4* out = a.model + ((size_t)tile * 256u + wave * 64u) * 4;
    const float4 v0 = make_float4(px, py, pz, sc), v1 = q, v2 = make_float4(q.w, q.z, q.y, q.x), v3 = make_float4(sc, px, __uint_as_float(mesh), 1.f);
    if (tile * 256u + wave * 64u + 64u <= a.n) {
      store16<kNT>(&out[lane], v0); store16<kNT>(&out[64 + lane], v1); store16<kNT>(&out[128 + lane], v2); store16<kNT>(&out[192 + lane], v3);
    } else {
      const uint32_t first = tile * 256u + wave * 64u;
#pragma unroll
      for (uint32_t s4 = 0; s4 < 4; ++s4)
        if (first + 16u * s4 + (lane >> 2) < a.n) out[64u * s4 + lane] = s4 == 0 ? v0 : (s4 == 1 ? v1 : (s4 == 2 ? v2 : v3));
    }
    const unsigned long long vis = __ballot(px > 0.f);
    if (lane < 2u && i < a.n) a.bitmap[(tile * 256u >> 5) + wave * 2u + lane] = (uint32_t)(vis >> (32u * lane));
    // ~0.27 commands per instance at a static position: 17 commands = 85 dwords per wave, coalesced dword stores
    uint32_t* c = a.cmds + ((size_t)tile * 4u + wave) * 85u;
    if (tile * 256u + wave * 64u < a.n) {
      c[lane] = __float_as_uint(px) + mesh;
      if (lane < 21u) c[64u + lane] = __float_as_uint(sc);
    }
  } else {
    // keep the loads alive
    if (px + py + pz + sc + q.x + q.y + q.z + q.w + (float)mesh == 123.456f) a.bitmap[i] = 1;
  }
}

template <bool kLoads, bool kStores, int kNT, int kPerWg>
__global__ __launch_bounds__(256) void mover(const Args a) {
#pragma unroll
  for (int k = 0; k < kPerWg; ++k) move_tile<kLoads, kStores, kNT>(a, blockIdx.x * kPerWg + k);
}

template <int kNT>
__global__ __launch_bounds__(256) void mover_persistent(const Args a, uint32_t n_tiles) {
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) move_tile<true, true, kNT>(a, t);
}

int main(int argc, char** argv) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (uint32_t n : {100000u, 1000000u, 10000000u}) {
    const uint32_t tiles = (n + 255) / 256;
    Args a{};
    float* pos; float4* rot; float* scale; uint32_t* mesh;
    CHECK(hipMalloc(&pos, (size_t)n * 12)); CHECK(hipMalloc(&rot, (size_t)n * 16));
    CHECK(hipMalloc(&scale, (size_t)n * 4)); CHECK(hipMalloc(&mesh, (size_t)n * 4));
    CHECK(hipMemset(pos, 0x3f, (size_t)n * 12)); CHECK(hipMemset(rot, 0x3f, (size_t)n * 16));
    CHECK(hipMemset(scale, 0x3f, (size_t)n * 4)); CHECK(hipMemset(mesh, 0, (size_t)n * 4));
    CHECK(hipMalloc(&a.model, (size_t)(tiles + 4) * 256 * 64));
    CHECK(hipMalloc(&a.bitmap, (size_t)(tiles + 4) * 256 * 4));  // also the keep-alive target of the loads-only variant
    CHECK(hipMalloc(&a.cmds, (size_t)(tiles + 4) * 4 * 85 * 4));
    a.pos = pos; a.rot = rot; a.scale = scale; a.mesh = mesh; a.n = n;
    CHECK(hipDeviceSynchronize());
    const double bytes_full = n * (36.0 + 64.0 + 0.125) + (double)tiles * 4 * 85 * 4;
    struct V { const char* name; double bytes; int id; };
    const V variants[] = {
        {"1 tile / WG (the real grid)", bytes_full, 0},
        {"2 tiles / WG", bytes_full, 1},
        {"4 tiles / WG", bytes_full, 2},
        {"persistent, 256 x 8 WGs", bytes_full, 3},
        {"persistent, 256 x 4 WGs", bytes_full, 4},
        {"1 tile / WG, non-temporal matrix stores", bytes_full, 5},
        {"persistent 256 x 8, non-temporal", bytes_full, 6},
        {"loads only (36 B)", n * 36.0, 7},
        {"stores only (64 B + bitmap + commands)", bytes_full - n * 36.0, 8},
        {"1 tile / WG, 'sc1 nt' matrix stores", bytes_full, 9},
        {"stores only, 'sc1 nt' matrix stores", bytes_full - n * 36.0, 10},
    };
    for (const V& v : variants) {
      auto launch = [&]() {
        switch (v.id) {
          case 0: hipLaunchKernelGGL((mover<true, true, 0, 1>), dim3(tiles), dim3(256), 0, st, a); break;
          case 1: hipLaunchKernelGGL((mover<true, true, 0, 2>), dim3((tiles + 1) / 2), dim3(256), 0, st, a); break;
          case 2: hipLaunchKernelGGL((mover<true, true, 0, 4>), dim3((tiles + 3) / 4), dim3(256), 0, st, a); break;
          case 3: hipLaunchKernelGGL((mover_persistent<0>), dim3(std::min(tiles, 2048u)), dim3(256), 0, st, a, tiles); break;
          case 4: hipLaunchKernelGGL((mover_persistent<0>), dim3(std::min(tiles, 1024u)), dim3(256), 0, st, a, tiles); break;
          case 5: hipLaunchKernelGGL((mover<true, true, 1, 1>), dim3(tiles), dim3(256), 0, st, a); break;
          case 6: hipLaunchKernelGGL((mover_persistent<1>), dim3(std::min(tiles, 2048u)), dim3(256), 0, st, a, tiles); break;
          case 7: hipLaunchKernelGGL((mover<true, false, 0, 1>), dim3(tiles), dim3(256), 0, st, a); break;
          case 8: hipLaunchKernelGGL((mover<false, true, 0, 1>), dim3(tiles), dim3(256), 0, st, a); break;
          case 9: hipLaunchKernelGGL((mover<true, true, 2, 1>), dim3(tiles), dim3(256), 0, st, a); break;
          case 10: hipLaunchKernelGGL((mover<false, true, 2, 1>), dim3(tiles), dim3(256), 0, st, a); break;
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
      printf("n=%-9u %-44s median %7.2f us  min %7.2f us  %.2f TB/s (median)\n", n, v.name, samples[3], samples[0],
             v.bytes / (samples[3] * 1e-6) / 1e12);
    }
    CHECK(hipFree(pos)); CHECK(hipFree(rot)); CHECK(hipFree(scale)); CHECK(hipFree(mesh));
    CHECK(hipFree(a.model)); CHECK(hipFree(a.bitmap)); CHECK(hipFree(a.cmds));
  }
  return 0;
}
