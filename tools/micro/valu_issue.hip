// valu_issue.hip — what does one SIMD of gfx950 sustain in wave64 f32 VALU instructions per cycle, and at what clock?
//
// bench.py priced its VALU-bound legs (per-triangle stage, four culled views, skinning) against
// "one wave64 instruction per 4 cycles per SIMD" and a 2.4 GHz clock, and one leg came out at 1.002 of that "peak".
// The guide has both figures (a wave64 VALU instruction issues over 2 cycles on a SIMD-32; ONE wave alone sustains one
// per 4). This program measures it: streams of INDEPENDENT instructions (16 accumulators in rotation, far beyond the
// dependent latency) of one kind — v_mul_f32, v_add_f32, v_fma_f32, v_pk_mul_f32, v_pk_fma_f32, and the mul/add mix of the
// library's no-FMA arithmetic — at 1, 2, 4 and 8 waves per SIMD, on ONE CU and on EVERY CU at once (the clock under load
// is part of the answer), timed per wave with s_memtime (shader cycles) and s_memrealtime (100 MHz), which also gives the
// clock actually running.
//
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_issue.hip -o tools/micro/valu_issue && tools/micro/valu_issue
//
// Output: one line per (instruction, waves per SIMD, CUs): cycles per wave-instruction per SIMD, the same in wave-
// instructions per second for the whole chip, and the measured clock. profiles/r04_valu_issue.txt is this output.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                     \
      std::exit(1);                                                                           \
    }                                                                                         \
  } while (0)

enum Op { kMul = 0, kAdd, kFma, kPkMul, kPkFma, kMulAddMix, kOps };
static const char* kOpName[kOps] = {"v_mul_f32", "v_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_mul_f32+v_add_f32"};
constexpr int kGroup = 16;  // wave-instructions per unrolled group

typedef float v2f __attribute__((ext_vector_type(2)));

// One group = 16 instructions over kChains accumulators in rotation: kChains = 16 is a stream of independent instructions,
// kChains = 1 one dependent chain (what a wave's instruction-to-instruction latency is), 2 and 4 in between — real
// arithmetic (a matrix-vector product without FMA) has 3-4 independent chains at a time.
template <int kOp, int kChains, class T>
__device__ __forceinline__ void group(T (&a)[16], T s, T t) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    T& x = a[k % kChains];
    if constexpr (kOp == kMul) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(s));
    else if constexpr (kOp == kAdd) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(t));
    else if constexpr (kOp == kFma) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(s), "v"(t));
    else if constexpr (kOp == kPkMul) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(s));
    else if constexpr (kOp == kPkFma) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(s), "v"(t));
    else if constexpr (kOp == kMulAddMix) {
      if (k & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(t));
      else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(s));
    }
  }
}

struct Stamp {
  unsigned long long cycles, r0, r1;
};

template <int kOp, int kChains>
__global__ __launch_bounds__(1024, 2) void issue_kernel(Stamp* out, float* sink, int iters, float sf, float tf) {
  constexpr bool kPacked = kOp == kPkMul || kOp == kPkFma;
  using T = typename std::conditional<kPacked, v2f, float>::type;
  T a[16];
  T s, t;
  if constexpr (kPacked) { s = v2f{sf, sf}; t = v2f{tf, tf}; } else { s = sf; t = tf; }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float v = 1.0f + 0.001f * (float)(threadIdx.x + k);
    if constexpr (kPacked) a[k] = v2f{v, v + 0.5f}; else a[k] = v;
  }
  __syncthreads();  // every wave of the workgroup starts together
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    group<kOp, kChains>(a, s, t);
    group<kOp, kChains>(a, s, t);
    group<kOp, kChains>(a, s, t);
    group<kOp, kChains>(a, s, t);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if constexpr (kPacked) acc += a[k].x + a[k].y; else acc += a[k];
  }
  if (acc == 123.456f) sink[0] = acc;  // keeps the arithmetic alive
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[w].cycles = c1 - c0;
    out[w].r0 = r0;
    out[w].r1 = r1;
  }
}

template <int kOp, int kChains>
static void run(int cus_used, int cus_total, int waves_per_simd, Stamp* d_out, float* d_sink, hipStream_t st) {
  // waves_per_simd W: one workgroup of 256*W threads per CU (W <= 4: the four SIMDs of a CU get W waves each), two
  // workgroups of 1024 threads for W = 8 (the kernel is built for two of them per CU: __launch_bounds__(1024, 2)).
  const int threads = waves_per_simd <= 4 ? 256 * waves_per_simd : 1024;
  const int blocks_per_cu = waves_per_simd <= 4 ? 1 : waves_per_simd / 4;
  const int blocks = cus_used * blocks_per_cu;
  const int iters = 4096;
  const long long instr_per_wave = (long long)iters * 4 * kGroup;
  const int waves = blocks * (threads / 64);
  std::vector<Stamp> h(waves);
  double best = 1e30, clock_mhz = 0, best_span_rate = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((issue_kernel<kOp, kChains>), dim3(blocks), dim3(threads), 0, st, d_out, d_sink, iters, 1.0000001f, 1e-30f);
    CHECK(hipGetLastError());
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
    // (a) per SIMD: its W waves share it for the whole run; the MEDIAN wave's cycle count over the instructions of W waves
    std::vector<unsigned long long> c(waves), t(waves);
    unsigned long long first = ~0ull, last = 0;
    for (int k = 0; k < waves; ++k) {
      c[k] = h[k].cycles;
      t[k] = h[k].r1 - h[k].r0;
      first = std::min(first, h[k].r0);
      last = std::max(last, h[k].r1);
    }
    std::nth_element(c.begin(), c.begin() + waves / 2, c.end());
    std::nth_element(t.begin(), t.begin() + waves / 2, t.end());
    const double cyc_per_instr = (double)c[waves / 2] / ((double)instr_per_wave * waves_per_simd);
    // (b) the whole launch: every wave-instruction over the span from the first wave's start to the last wave's end (100 MHz
    // ticks) — does not assume how many waves were resident together
    const double span_rate = (double)instr_per_wave * waves / ((double)(last - first) / 1e8);
    if (cyc_per_instr < best) {
      best = cyc_per_instr;
      clock_mhz = (double)c[waves / 2] / ((double)t[waves / 2] / 100.0);
      best_span_rate = span_rate;
    }
  }
  std::printf("%-20s chains %2d  waves/SIMD %d  CUs busy %3d of %d  %6.3f cycles per wave-instruction per SIMD  clock %4.0f MHz  launch as a whole: %.3e wave-instr/s = %.3e per busy SIMD\n",
              kOpName[kOp], kChains, waves_per_simd, cus_used, cus_total, best, clock_mhz, best_span_rate, best_span_rate / (4.0 * cus_used));
  std::fflush(stdout);
}

template <int kOp, int kChains>
static void sweep(int cus, Stamp* d_out, float* d_sink, hipStream_t st) {
  for (int w : {1, 2, 4}) run<kOp, kChains>(1, cus, w, d_out, d_sink, st);  // (8 waves per SIMD need two workgroups, which an idle chip would put on two CUs)
  for (int w : {1, 2, 4, 8}) run<kOp, kChains>(cus, cus, w, d_out, d_sink, st);
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("# %s, %d CUs, clock reported by the runtime %d MHz; every stream is 262 144 wave-instructions per wave\n",
              prop.gcnArchName, cus, prop.clockRate / 1000);
  Stamp* d_out = nullptr;
  float* d_sink = nullptr;
  CHECK(hipMalloc(&d_out, sizeof(Stamp) * (size_t)cus * 2 * 16));
  CHECK(hipMalloc(&d_sink, 64));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  sweep<kMul, 16>(cus, d_out, d_sink, st);
  sweep<kAdd, 16>(cus, d_out, d_sink, st);
  sweep<kFma, 16>(cus, d_out, d_sink, st);
  sweep<kMulAddMix, 16>(cus, d_out, d_sink, st);
  sweep<kPkMul, 16>(cus, d_out, d_sink, st);
  sweep<kPkFma, 16>(cus, d_out, d_sink, st);
  std::printf("# dependent chains: how much instruction-level parallelism a wave needs\n");
  sweep<kMulAddMix, 1>(cus, d_out, d_sink, st);
  sweep<kMulAddMix, 2>(cus, d_out, d_sink, st);
  sweep<kMulAddMix, 4>(cus, d_out, d_sink, st);
  sweep<kPkMul, 1>(cus, d_out, d_sink, st);
  sweep<kPkMul, 4>(cus, d_out, d_sink, st);
  return 0;
}
