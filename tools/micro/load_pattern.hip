// Micro-benchmark: 40 B per lane read as five 8-byte loads at a 40-byte lane stride (the skinned
// kernel's pose pattern) against lane-contiguous 8-byte loads of the same bytes.
// build: hipcc -O3 --offload-arch=gfx950 -o load_pattern load_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void strided40(const float2* in, float* out, size_t n_threads) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_threads) return;
  const float2* p = in + i * 5;
  const float2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
  const float s = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y + e.x + e.y;
  if (s == 12345.678f) out[i] = s;  // keeps the loads alive, never true for the zero-filled input
}

__global__ void contiguous8(const float2* in, float* out, size_t n_threads) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_threads) return;
  const size_t wave_base = (i & ~(size_t)63) * 5;
  const size_t lane = i & 63;
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const float2 v = in[wave_base + 64 * q + lane];
    s += v.x + v.y;
  }
  if (s == 12345.678f) out[i] = s;
}

int main() {
  const size_t n_threads = 256000ull * 19;
  const size_t bytes = n_threads * 40;
  float2* in;
  float* out;
  CHECK(hipMalloc(&in, bytes));
  CHECK(hipMemset(in, 0, bytes));
  CHECK(hipMalloc(&out, n_threads * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int block = 256;
  const int grid = (int)((n_threads + block - 1) / block);
  for (int pattern = 0; pattern < 2; ++pattern) {
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
      CHECK(hipEventRecord(e0));
      if (pattern == 0) strided40<<<grid, block>>>(in, out, n_threads);
      else contiguous8<<<grid, block>>>(in, out, n_threads);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-48s mean %.1f us  min %.1f us  %.2f TB/s (min)\n",
           pattern == 0 ? "5 x 8 B per lane, 40-B lane stride" : "lane-contiguous 8 B (512 B per instruction)", sum / reps * 1e3,
           best * 1e3, bytes / best / 1e9);
  }
  return 0;
}
