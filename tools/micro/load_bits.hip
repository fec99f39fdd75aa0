// Micro-benchmark: the byte mover of store_bits.hip with the matrix stores fixed at 'sc1 nt' and the four input loads
// (pos dwordx3, quat dwordx4, scale dword, mesh id dword) issued with every combination of the sc0 / sc1 / nt bits.
// build: hipcc -O3 --offload-arch=gfx950 -o load_bits load_bits.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));

#define LOADS(BITS)                                                                                   \
  asm volatile("global_load_dwordx3 %0, %4, off " BITS "\n global_load_dwordx4 %1, %5, off " BITS "\n"  \
               "global_load_dword %2, %6, off " BITS "\n global_load_dword %3, %7, off " BITS "\n s_waitcnt vmcnt(0)" \
               : "=&v"(p), "=&v"(q), "=&v"(sc), "=&v"(mesh)                                          \
               : "v"(pp), "v"(qp), "v"(sp), "v"(mp) : "memory")

struct Args { const float* pos; const float4* rot; const float* scale; const uint32_t* mesh; float4* model; uint32_t n; };

template <int kBits>
__global__ __launch_bounds__(256) void mover(const Args a) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, tile = blockIdx.x;
  const uint32_t i = tile * 256u + tid;
  const uint32_t il = i < a.n ? i : a.n - 1u;
  v3f p; v4f q; float sc; uint32_t mesh;
  const float* pp = a.pos + 3 * (size_t)il; const float4* qp = a.rot + il; const float* sp = a.scale + il; const uint32_t* mp = a.mesh + il;
  if constexpr (kBits == 0) LOADS("");
  if constexpr (kBits == 1) LOADS("nt");
  if constexpr (kBits == 2) LOADS("sc0");
  if constexpr (kBits == 3) LOADS("sc0 nt");
  if constexpr (kBits == 4) LOADS("sc1");
  if constexpr (kBits == 5) LOADS("sc1 nt");
  if constexpr (kBits == 6) LOADS("sc0 sc1");
  if constexpr (kBits == 7) LOADS("sc0 sc1 nt");
  float4* out = a.model + ((size_t)tile * 256u + wave * 64u) * 4;
  if (tile * 256u + wave * 64u + 64u <= a.n) {
    const v4f v0 = {p.x, p.y, p.z, sc}, v1 = q, v2 = {q.w, q.z, q.y, q.x}, v3 = {sc, p.x, __uint_as_float(mesh), 1.f};
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n global_store_dwordx4 %0, %2, off offset:1024 sc1 nt\n"
                 "global_store_dwordx4 %0, %3, off offset:2048 sc1 nt\n global_store_dwordx4 %0, %4, off offset:3072 sc1 nt\n s_nop 1"
                 :: "v"(out + lane), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
  }
}

template <int kBits>
static void run(const Args& a, uint32_t tiles, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const char* name) {
  auto launch = [&]() { hipLaunchKernelGGL((mover<kBits>), dim3(tiles), dim3(256), 0, st, a); };
  for (int k = 0; k < 20; ++k) launch();
  CHECK(hipStreamSynchronize(st));
  std::vector<float> samples;
  const int K = a.n >= 10000000u ? 40 : 200;
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
  printf("n=%u loads [%-10s] + 'sc1 nt' matrix stores: median %6.2f us  min %6.2f us  %.2f TB/s\n", a.n, name, samples[3], samples[0],
         a.n * 100.0 / (samples[3] * 1e-6) / 1e12);
}

int main() {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (uint32_t n : {1000000u, 4000000u, 10000000u}) {
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
