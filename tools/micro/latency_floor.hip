// Micro-benchmark: what is the floor for a kernel of the instance pipeline's SHAPE at 100 k and 1 M instances?
//   empty        : 391 / 3907 workgroups of 256 threads that do nothing (dispatch + drain only)
//   stream       : each thread reads 36 B and writes 64 B + 20 B (the pipeline's traffic, no arithmetic, no dependency)
//   stream + hop : the same, plus what a single-pass compaction cannot avoid — every workgroup publishes a word and
//                  waits for the word of the workgroup before it (one cross-workgroup round trip through memory)
// Timed with hipEvents over back-to-back launches on one stream (the way bench.py times the kernel alone).
// build: hipcc -O3 --offload-arch=gfx950 -o latency_floor latency_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void empty_kernel(int) {}

__global__ void stream_kernel(const float* in, float4* out_m, uint32_t* out_c, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * 9;
  float a[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) a[k] = p[k];
  float4* o = out_m + (size_t)i * 4;
  o[0] = make_float4(a[0], a[1], a[2], 0.f);
  o[1] = make_float4(a[3], a[4], a[5], 0.f);
  o[2] = make_float4(a[6], a[7], a[8], 0.f);
  o[3] = make_float4(a[0], a[4], a[8], 1.f);
  if ((i & 3u) == 0u) {  // ~ one 20-byte command per four instances (v = 0.27)
    uint32_t* c = out_c + (size_t)(i >> 2) * 5;
    c[0] = i; c[1] = 1; c[2] = i; c[3] = 0; c[4] = i;
  }
}

__global__ void stream_hop_kernel(const float* in, float4* out_m, uint32_t* out_c, uint32_t n, unsigned long long* flags,
                                  uint32_t epoch) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (i < n) {
    const float* p = in + (size_t)i * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = p[k];
  }
  // publish this workgroup's word, then wait for the predecessor's (ONE hop: nobody waits before publishing)
  __shared__ uint32_t s_prev;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&flags[blockIdx.x], ((unsigned long long)epoch << 32) | (uint32_t)(a[0] != 0.f), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prev = 0;
    if (blockIdx.x > 0) {
      unsigned long long v;
      do {
        v = __hip_atomic_load(&flags[blockIdx.x - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while ((uint32_t)(v >> 32) != epoch);
      prev = (uint32_t)v;
    }
    s_prev = prev;
  }
  __syncthreads();
  if (i >= n) return;
  float4* o = out_m + (size_t)i * 4;
  o[0] = make_float4(a[0], a[1], a[2], 0.f);
  o[1] = make_float4(a[3], a[4], a[5], 0.f);
  o[2] = make_float4(a[6], a[7], a[8], 0.f);
  o[3] = make_float4(a[0], a[4], a[8], 1.f);
  if ((i & 3u) == 0u) {
    uint32_t* c = out_c + (size_t)(i >> 2) * 5;
    c[0] = i + s_prev; c[1] = 1; c[2] = i; c[3] = 0; c[4] = i;
  }
}

int main() {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (uint32_t n : {100000u, 1000000u}) {
    const uint32_t blocks = (n + 255) / 256;
    float* in;
    float4* out_m;
    uint32_t* out_c;
    unsigned long long* flags;
    CHECK(hipMalloc(&in, (size_t)n * 36));
    CHECK(hipMemset(in, 0, (size_t)n * 36));
    CHECK(hipMalloc(&out_m, (size_t)n * 64));
    CHECK(hipMalloc(&out_c, (size_t)n * 5 + 64));
    CHECK(hipMalloc(&flags, (size_t)blocks * 8));
    CHECK(hipMemset(flags, 0, (size_t)blocks * 8));
    CHECK(hipDeviceSynchronize());
    const double bytes = n * (36.0 + 64.0 + 5.0);
    uint32_t epoch = 0;
    for (int variant = 0; variant < 3; ++variant) {
      const int K = 300;
      auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, st, 0);
        else if (variant == 1) hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, st, in, out_m, out_c, n);
        else hipLaunchKernelGGL(stream_hop_kernel, dim3(blocks), dim3(256), 0, st, in, out_m, out_c, n, flags, ++epoch);
      };
      for (int k = 0; k < 30; ++k) launch();
      CHECK(hipStreamSynchronize(st));
      CHECK(hipEventRecord(e0, st));
      for (int k = 0; k < K; ++k) launch();
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const char* names[] = {"empty", "stream (36 B in, 69 B out per instance)", "stream + one cross-workgroup hop"};
      printf("n=%-8u %-44s %6.2f us per launch", n, names[variant], ms / K * 1e3);
      if (variant) printf("  (%.2f TB/s)", bytes / (ms / K * 1e-3) / 1e12);
      printf("\n");
    }
    CHECK(hipFree(in)); CHECK(hipFree(out_m)); CHECK(hipFree(out_c)); CHECK(hipFree(flags));
  }
  return 0;
}
