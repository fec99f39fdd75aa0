// Micro-benchmark: how fast does the chip take 64 B per lane written as four 16-byte stores at a
// 64-byte lane stride (the skinned kernel's palette pattern) against lane-contiguous 16-byte stores?
// build: hipcc -O3 --offload-arch=gfx950 -o store_pattern store_pattern.hip ; run: ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// pattern 0: lane l of the grid writes float4 #(4*i + q) for q = 0..3 with i = global thread: 64-B lane stride
__global__ void strided64(float4* out, size_t n_threads, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_threads) return;
  float4* o = out + i * 4;
  o[0] = make_float4(v, v + 1, v + 2, 0.f);
  o[1] = make_float4(v + 3, v + 4, v + 5, 0.f);
  o[2] = make_float4(v + 6, v + 7, v + 8, 0.f);
  o[3] = make_float4(v + 9, v + 10, v + 11, 1.f);
}

// pattern 1: the same bytes, but every store instruction of a wave is 1 KiB contiguous
__global__ void contiguous16(float4* out, size_t n_threads, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_threads) return;
  const size_t wave_base = (i & ~(size_t)63) * 4;
  const size_t lane = i & 63;
#pragma unroll
  for (int q = 0; q < 4; ++q) out[wave_base + 64 * q + lane] = make_float4(v + q, v + 1, v + 2, 0.f);
}

// pattern 2: strided64 after ~N dependent FP operations (does a compute phase in front of the stores matter?)
__global__ void strided64_after_alu(float4* out, size_t n_threads, float v, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_threads) return;
  float a = v + (float)(i & 7);
  for (int k = 0; k < iters; ++k) a = a * 1.0001f + 0.5f;
  float4* o = out + i * 4;
  o[0] = make_float4(a, v + 1, v + 2, 0.f);
  o[1] = make_float4(v + 3, a, v + 5, 0.f);
  o[2] = make_float4(v + 6, v + 7, a, 0.f);
  o[3] = make_float4(v + 9, v + 10, v + 11, 1.f);
}

int main() {
  const size_t n_threads = 256000ull * 19;  // one per (instance, joint)
  const size_t bytes = n_threads * 64;
  float4* out;
  CHECK(hipMalloc(&out, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int block = 256;
  const int grid = (int)((n_threads + block - 1) / block);
  for (int pattern = 0; pattern < 5; ++pattern) {
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
      CHECK(hipEventRecord(e0));
      if (pattern == 0) strided64<<<grid, block>>>(out, n_threads, (float)r);
      else if (pattern == 1) contiguous16<<<grid, block>>>(out, n_threads, (float)r);
      else strided64_after_alu<<<grid, block>>>(out, n_threads, (float)r, pattern == 2 ? 100 : (pattern == 3 ? 400 : 1600));
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    const char* names[] = {"4 x 16 B per lane, 64-B lane stride", "lane-contiguous 16 B (1 KiB per instruction)",
                           "strided after 100 dependent FMAs", "strided after 400", "strided after 1600"};
    printf("%-48s mean %.1f us  min %.1f us  %.2f TB/s (min)\n", names[pattern], sum / reps * 1e3, best * 1e3, bytes / best / 1e9);
  }
  return 0;
}
