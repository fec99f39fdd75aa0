// l1_line_rate.hip — how many cache lines per cycle does one CU's vector L1 look up for a gathered load?
//
// The per-triangle kernel (row f-1) gathers three packed vec3 positions per lane and step; its PMC pass shows
// TCP_TOTAL_CACHE_ACCESSES = 0.89 per cycle per CU (profiles/r04_triangle_cull_100k_rows_pmc_summary.json) and its time doubles
// when the triangle order is shuffled (every lane its own line) although the arithmetic is the same. This program measures the
// ceiling that figure is a fraction of: waves issue `global_load_dwordx3` (12 B per lane, as the kernel's gathers) from an
// L2-resident buffer with a chosen LANE STRIDE — 12 B (consecutive packed vec3: ~6 lines of 128 B per wave-load), 48 B, 128 B
// and 256 B (every lane its own line) and a pseudo-random line per lane — eight waves per SIMD on every CU, nothing else in
// the loop. Reported: wave-loads per second, lanes' distinct 128-byte and 64-byte lines per wave-load, and lines per cycle per
// CU at the clock measured by s_memtime against s_memrealtime.
//
//   hipcc --offload-arch=gfx950 -O2 tools/micro/l1_line_rate.hip -o /tmp/l1_line_rate && /tmp/l1_line_rate
// Output = profiles/r04_l1_line_rate.txt.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                 \
      std::exit(1);                                                                       \
    }                                                                                     \
  } while (0)

struct Stamp {
  unsigned long long cycles, ticks;
};

constexpr int kLoadsPerIter = 8;
constexpr int kIters = 512;
constexpr unsigned kBufferBytes = 2u << 20;  // 2 MiB: inside one XCD's 4 MiB L2, far beyond the 32 KiB L1

// offsets[lane] = byte offset of the lane's 12 bytes inside a 16 KiB window; the window moves through the buffer per load
// kScalarBase: the window's base is made wave-uniform (readfirstlane), so the load is the SGPR-base + 32-bit-VGPR-offset form
// (global_load_dwordx3 v, v_off, s[base:base+1]) instead of a 64-bit address per lane (global_load_dwordx3 v, v[lo:hi], off).
// kValu: independent f32 multiply / add instructions per iteration beside the 8 loads (0 = loads only; kLoads = false: the
// arithmetic alone) — does the vector-load path overlap with VALU issue, or do the two add up?
template <bool kScalarBase, int kValu = 0, bool kLoads = true>
__global__ __launch_bounds__(256) void gather_kernel(const char* buf, const unsigned* offsets, Stamp* out, float* sink, unsigned window_mask) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave_global = blockIdx.x * 4u + (threadIdx.x >> 6);
  const unsigned my = offsets[lane];
  unsigned window = (wave_global * 40503u) & window_mask;
  float acc = 0.f;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  typedef float v3f __attribute__((ext_vector_type(3)));
  float m[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) m[q] = 1.0f + 0.001f * (float)(lane + q);
  for (int it = 0; it < kIters; ++it) {
    v3f x[kLoadsPerIter];
#pragma unroll
    for (int k = 0; k < kLoadsPerIter; ++k) x[k] = v3f{0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < (kLoads ? kLoadsPerIter : 0); ++k) {
      // (inline assembly: the compiler folds a wave-uniform base back into a 64-bit address per lane)
      if constexpr (kScalarBase) {
        const char* base = buf + (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)window) * 16384u;
        // s_nop: the hazard recogniser does not see that this statement READS the SGPR pair a scalar add has just written
        // (an SALU write needs wait states before a vector-memory instruction uses it as an address: without them the load
        // goes to a stale address and faults)
        asm volatile("s_nop 7\n\tglobal_load_dwordx3 %0, %1, %2" : "=v"(x[k]) : "v"(my), "s"(base) : "memory");
      } else {
        const char* p = buf + (size_t)window * 16384u + my;
        asm volatile("s_nop 7\n\tglobal_load_dwordx3 %0, %1, off" : "=v"(x[k]) : "v"(p) : "memory");
      }
      window = (window + 1u) & window_mask;
    }
#pragma unroll
    for (int v = 0; v < kValu; ++v) {  // eight independent chains, mul and add alternating: issued while the loads are in flight
      if (v & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(m[v & 7]) : "v"(1e-30f));
      else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(m[v & 7]) : "v"(1.0000001f));
    }
    // the loaded registers are operands of the wait, so nothing reads (or reuses) them before it
    static_assert(kLoadsPerIter == 8, "operand list below");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])::"memory");
#pragma unroll
    for (int k = 0; k < kLoadsPerIter; ++k) acc += x[k].x + x[k].y + x[k].z;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int q = 0; q < 8; ++q) acc += m[q];
  if (acc == 123.456f) sink[0] = acc;
  if (lane == 0) {
    out[wave_global].cycles = c1 - c0;
    out[wave_global].ticks = r1 - r0;
  }
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  char* buf;
  unsigned* d_off;
  Stamp* d_out;
  float* d_sink;
  CHECK(hipMalloc(&buf, kBufferBytes + 65536));
  CHECK(hipMemset(buf, 0, kBufferBytes + 65536));
  CHECK(hipMalloc(&d_off, 64 * 4));
  CHECK(hipMalloc(&d_sink, 64));
  std::printf("# %s, %d CUs; %d wave-loads of 12 B per lane per wave, 8 waves per SIMD, buffer %u KiB (L2-resident)\n", prop.gcnArchName, cus,
              kIters * kLoadsPerIter, kBufferBytes >> 10);
  struct Pattern { const char* name; int stride; } patterns[] = {
      {"lane stride 12 B (consecutive packed vec3)", 12}, {"lane stride 24 B", 24},   {"lane stride 48 B (every 4th vertex)", 48},
      {"lane stride 64 B", 64},                           {"lane stride 128 B", 128}, {"lane stride 256 B (a line per lane)", 256},
      {"random 12-byte slot of the 16 KiB window per lane", -1}};
  for (int scalar_base : {0, 1})
  for (unsigned windows : {1u, kBufferBytes / 16384u})
  for (int waves_per_simd : {8, 4}) {
    if (windows != kBufferBytes / 16384u && waves_per_simd != 8) continue;
    std::printf("# working set %u KiB (%s), address = %s\n", windows * 16u, windows <= 2u ? "fits the CU's 32 KiB L1: every load hits" : "served by the L2",
                scalar_base ? "SGPR base + 32-bit lane offset" : "64-bit per lane");
    const int blocks = cus * waves_per_simd;  // 256 threads = one wave per SIMD of a CU per block
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * (size_t)blocks * 4));
    for (const Pattern& p : patterns) {
      std::vector<unsigned> off(64);
      unsigned seed = 12345u;
      for (int l = 0; l < 64; ++l) {
        if (p.stride > 0) off[l] = (unsigned)(l * p.stride);
        else { seed = seed * 1664525u + 1013904223u; off[l] = ((seed >> 8) % (16384u / 12u - 1u)) * 12u; }
      }
      std::set<unsigned> l128, l64;
      for (int l = 0; l < 64; ++l)
        for (unsigned b : {off[l], off[l] + 11u}) { l128.insert(b / 128u); l64.insert(b / 64u); }
      CHECK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
      double best = 0, clock_mhz = 0;
      for (int rep = 0; rep < 3; ++rep) {
        if (scalar_base) hipLaunchKernelGGL(gather_kernel<true>, dim3(blocks), dim3(256), 0, 0, buf, d_off, d_out, d_sink, windows - 1u);
        else hipLaunchKernelGGL(gather_kernel<false>, dim3(blocks), dim3(256), 0, 0, buf, d_off, d_out, d_sink, windows - 1u);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        std::vector<Stamp> h((size_t)blocks * 4);
        CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        std::vector<unsigned long long> c(h.size()), t(h.size());
        for (size_t k = 0; k < h.size(); ++k) { c[k] = h[k].cycles; t[k] = h[k].ticks; }
        std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end());
        std::nth_element(t.begin(), t.begin() + t.size() / 2, t.end());
        // per CU: 4 SIMDs x waves_per_simd waves, each kIters * kLoadsPerIter wave-loads in the median wave's cycles
        const double loads_per_cycle_per_cu = 4.0 * waves_per_simd * kIters * kLoadsPerIter / (double)c[c.size() / 2];
        if (loads_per_cycle_per_cu > best) {
          best = loads_per_cycle_per_cu;
          clock_mhz = (double)c[c.size() / 2] / ((double)t[t.size() / 2] / 100.0);
        }
      }
      std::printf("%-52s waves/SIMD %d  lines per wave-load: %2zu of 128 B, %2zu of 64 B   %.4f wave-loads per cycle per CU = %6.3f (128 B) / %6.3f (64 B) "
                  "lines per cycle per CU   clock %4.0f MHz\n",
                  p.name, waves_per_simd, l128.size(), l64.size(), best, best * l128.size(), best * l64.size(), clock_mhz);
      std::fflush(stdout);
    }
    CHECK(hipFree(d_out));
  }
  // ---- overlap: 8 loads (every 4th vertex, L2-served) + 346 VALU instructions per iteration = two steps of the per-triangle kernel
  {
    const int blocks = cus * 8;
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * (size_t)blocks * 4));
    std::vector<unsigned> off(64);
    for (int l = 0; l < 64; ++l) off[l] = (unsigned)(l * 48);
    CHECK(hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice));
    auto run = [&](const char* what, auto kernel, unsigned windows) {
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, buf, d_off, d_out, d_sink, windows - 1u);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        std::vector<Stamp> h((size_t)blocks * 4);
        CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
        std::vector<unsigned long long> c(h.size());
        for (size_t k = 0; k < h.size(); ++k) c[k] = h[k].cycles;
        std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end());
        best = std::min(best, (double)c[c.size() / 2] / kIters);
      }
      std::printf("%-78s %8.1f cycles per iteration per wave (8 waves per SIMD) = %6.1f cycles of the CU per wave-iteration\n", what, best, best / 32.0);
    };
    std::printf("# overlap of the load path with VALU issue: one iteration = 8 gathers at the every-4th-vertex stride and/or 346 independent f32 instructions\n");
    for (unsigned windows : {1u, kBufferBytes / 16384u}) {
      std::printf("# working set %u KiB\n", windows * 16u);
      run("loads only", gather_kernel<false, 0, true>, windows);
      run("346 VALU instructions only", gather_kernel<false, 346, false>, windows);
      run("loads + 346 VALU instructions (in flight together)", gather_kernel<false, 346, true>, windows);
    }
    CHECK(hipFree(d_out));
  }
  return 0;
}
