// triangle_kernels.hpp — row f-1: per-triangle cull + index-stream append, and the re-compaction after it (gfx950).
#pragma once

#include "instance_kernel.hpp"
#include "stage_args.hpp"
#include "frame_plan.hpp"

#pragma clang fp contract(off)

namespace mip {

// ---------------------------------------------------------------------------------------
// row f-1: per-triangle cull + index-stream append (src/shaders/generate_work.comp:68-200)
// ---------------------------------------------------------------------------------------
// The reference records one dispatch per visible instance (cull_pipeline.rs:536-577). Here
// one launch walks the compacted command list: ONE WAVE PER COMMAND, 64 triangles per
// step, the running survivor count in a register — no inter-wave communication, and the
// surviving triangles keep their mesh order (the stable member of the reference's
// outcome set; its workgroups append in atomicAdd arrival order, :176-186).
// Arithmetic: clip = pv * (model * vec4(v,1)) as column combinations left to right, no
// FMA; back-face = determinant of the xyw columns > 0; x/y NDC rejection after a true
// divide — exactly what the oracle (orc_cull_triangles) fixes where GLSL leaves it open.



__device__ __forceinline__ void glsl_mat4_mul_vec4(const float (&m)[16], float x, float y, float z, float w, float (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = m[0 * 4 + r] * x + m[1 * 4 + r] * y + m[2 * 4 + r] * z + m[3 * 4 + r] * w;
}

// The three positions of a triangle (packed vec3 each).
__device__ __forceinline__ void triangle_fetch(const float* vertices, long long vertex_offset, uint32_t i0, uint32_t i1,
                                               uint32_t i2, float (&v)[9]) {
  const uint32_t ix[3] = {i0, i1, i2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float* p = vertices + (vertex_offset + (long long)ix[k]) * 3;
    v[k * 3 + 0] = p[0]; v[k * 3 + 1] = p[1]; v[k * 3 + 2] = p[2];
  }
}

// clip.xyw of one vertex: pv * (model * vec4(v, 1)) (generate_work.comp:132-136) as column combinations left to
// right, no FMA — the x, y and w rows only: clip.z never enters the tests below, so it is never computed.
// kAffine: the caller has checked that row 3 of `model` is (0,0,0,1) and that the geometry holds
// only finite positions. Then world.w = ((0*x + 0*y) + 0*z) + 1 is exactly 1 and pv[:,3] * world.w
// is exactly pv[:,3], so that row and those products are skipped: same bits, fewer flops.
//
// A/B build -DMIP_TRI_PACKED: rows 0 and 1 of a column-major matrix are adjacent (model[c*4 + 0], model[c*4 + 1]), so
// the two rows of every column combination can be written as ONE two-wide operation (v_pk_mul_f32 / v_pk_add_f32:
// element-wise IEEE multiply and add, each rounded on its own — the bits of two scalar instructions; the coordinate is
// broadcast by op_sel, no register moves). 24 instead of 36 vector instructions per vertex, 120 instead of 165 per
// 64-triangle step — and SLOWER: 1.07 against 1.03 ms for the 100 k frame (profiles/r03_triangle_packed_ab.txt). A
// packed f32 instruction occupies the SIMD for two passes; the kernel is bound by VALU issue TIME (95 % busy), which
// counts flops, not instructions. So the product build keeps one instruction per flop.
typedef float tri_v2f __attribute__((ext_vector_type(2)));
template <bool kAffine>
__device__ __forceinline__ void vertex_clip_xyw(const float (&model)[16], const float (&pv)[16], float x, float y, float z, float (&c)[3]) {
#ifdef MIP_TRI_PACKED
  auto col01 = [](const float (&m)[16], int col) { return tri_v2f{m[col * 4 + 0], m[col * 4 + 1]}; };
  if constexpr (kAffine) {
    const tri_v2f w01 = col01(model, 0) * x + col01(model, 1) * y + col01(model, 2) * z + col01(model, 3);
    const float w2 = model[0 * 4 + 2] * x + model[1 * 4 + 2] * y + model[2 * 4 + 2] * z + model[3 * 4 + 2];
    const tri_v2f cxy = col01(pv, 0) * w01.x + col01(pv, 1) * w01.y + col01(pv, 2) * w2 + col01(pv, 3);
    c[0] = cxy.x; c[1] = cxy.y;
    c[2] = pv[0 * 4 + 3] * w01.x + pv[1 * 4 + 3] * w01.y + pv[2 * 4 + 3] * w2 + pv[3 * 4 + 3];
  } else {
    auto col23 = [](const float (&m)[16], int col) { return tri_v2f{m[col * 4 + 2], m[col * 4 + 3]}; };
    const tri_v2f w01 = col01(model, 0) * x + col01(model, 1) * y + col01(model, 2) * z + col01(model, 3) * 1.0f;
    const tri_v2f w23 = col23(model, 0) * x + col23(model, 1) * y + col23(model, 2) * z + col23(model, 3) * 1.0f;
    const tri_v2f cxy = col01(pv, 0) * w01.x + col01(pv, 1) * w01.y + col01(pv, 2) * w23.x + col01(pv, 3) * w23.y;
    c[0] = cxy.x; c[1] = cxy.y;
    c[2] = pv[0 * 4 + 3] * w01.x + pv[1 * 4 + 3] * w01.y + pv[2 * 4 + 3] * w23.x + pv[3 * 4 + 3] * w23.y;
  }
#else
  constexpr int kRows[3] = {0, 1, 3};
  if constexpr (kAffine) {
    float world[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) world[r] = model[0 * 4 + r] * x + model[1 * 4 + r] * y + model[2 * 4 + r] * z + model[3 * 4 + r];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int r = kRows[q];
      c[q] = pv[0 * 4 + r] * world[0] + pv[1 * 4 + r] * world[1] + pv[2 * 4 + r] * world[2] + pv[3 * 4 + r];
    }
  } else {
    float world[4];
    glsl_mat4_mul_vec4(model, x, y, z, 1.0f, world);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int r = kRows[q];
      c[q] = pv[0 * 4 + r] * world[0] + pv[1 * 4 + r] * world[1] + pv[2 * 4 + r] * world[2] + pv[3 * 4 + r] * world[3];
    }
  }
#endif
}

// One triangle of generate_work.comp:137-155 from the clip.xyw of its corners: true = culled (back-facing or
// beyond one x/y bound).
__device__ __forceinline__ bool triangle_cull_clip(const float (&clip)[3][3]) {
  const float a00 = clip[0][0], a01 = clip[0][1], a02 = clip[0][2];
  const float a10 = clip[1][0], a11 = clip[1][1], a12 = clip[1][2];
  const float a20 = clip[2][0], a21 = clip[2][1], a22 = clip[2][2];
  const float det = (a00 * (a11 * a22 - a21 * a12) - a10 * (a01 * a22 - a21 * a02)) + a20 * (a01 * a12 - a11 * a02);
  bool cull = det > 0.0f;
  // ndc = clip.xy / clip.w compared with -1 and 1 (generate_work.comp:143-155), without dividing:
  // for floats x, w the correctly rounded quotient q = RN(x / w) satisfies
  //     q > 1  <=>  x*sgn(w) > |w|        q < -1  <=>  x*sgn(w) < -|w|
  // because x*sgn(w) > |w| puts x/w at least one ulp(w)/|w| >= 2^-23 above 1, past the rounding
  // boundary 1 + 2^-24, and x*sgn(w) <= |w| gives x/w <= 1. It also holds at w = +-0 (q = +-inf by
  // the signs, NaN for 0/0), for infinities and NaNs (every comparison false), and for subnormals
  // (tests/test_oracle.py::test_ndc_comparison_without_division checks it against real divisions).
  // The six correctly rounded divides were 60 of the 197 VALU instructions of a step.
  bool xl = true, xg = true, yl = true, yg = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t sw = __float_as_uint(clip[k][2]) & 0x80000000u;
    const float w = fabsf(clip[k][2]);
    const float x = __uint_as_float(__float_as_uint(clip[k][0]) ^ sw);
    const float y = __uint_as_float(__float_as_uint(clip[k][1]) ^ sw);
    xl = xl && (x < -w);
    xg = xg && (x > w);
    yl = yl && (y < -w);
    yg = yg && (y > w);
  }
  return cull || xl || xg || yl || yg;
}

template <bool kAffine>
__device__ __forceinline__ bool triangle_test(const float (&model)[16], const float (&pv)[16], const float (&v)[9]) {
  float clip[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) vertex_clip_xyw<kAffine>(model, pv, v[k * 3 + 0], v[k * 3 + 1], v[k * 3 + 2], clip[k]);
  return triangle_cull_clip(clip);
}

__device__ __forceinline__ bool triangle_culled(bool affine, const float (&model)[16], const float (&pv)[16], const float* vertices,
                                                long long vertex_offset, uint32_t i0, uint32_t i1, uint32_t i2) {
  float v[9];
  triangle_fetch(vertices, vertex_offset, i0, i1, i2, v);
  return affine ? triangle_test<true>(model, pv, v) : triangle_test<false>(model, pv, v);  // wave-uniform
}

// Wave-uniform: may this command's triangles take the affine path?
__device__ __forceinline__ bool model_is_affine(const float (&model)[16], uint32_t geometry_finite) {
  return geometry_finite != 0u && model[3] == 0.0f && model[7] == 0.0f && model[11] == 0.0f && model[15] == 1.0f;
}

// Large frames launch the wave-per-command grid AND a workgroup-per-command grid; every workgroup of both asks this
// (wave-uniform, three scalar loads) and one grid returns at once. The rule and its measurements: frame_plan.hpp.
__device__ __forceinline__ bool tri_choice_is_block(const TriangleArgs& a) {
  return plan_tri_choice_is_block(a.max_lod_tris, *a.index_total, *a.count);
}

#ifndef MIP_TRI_MIN_WAVES_PER_SIMD
#define MIP_TRI_MIN_WAVES_PER_SIMD 4
#endif

__global__ __launch_bounds__(256, MIP_TRI_MIN_WAVES_PER_SIMD) void mip_triangle_cull_kernel(const TriangleArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  if (a.index_total && tri_choice_is_block(a)) return;  // this frame is the workgroup-per-command grid's
  const uint32_t count = *a.count;
  float pv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) pv[k] = a.pv[k];

  // Commands differ 1000x in triangle count (LODs, mixed meshes): waves pull the next command
  // from a ticket counter instead of striding over the list (measured: static striding left a
  // third of the waves idle at 5 k commands). The counter is zeroed by the host per launch.
  // Every lane takes part in the add (lane 0 adds 1, the others 0: the compiler folds the wave's
  // adds into one atomic), so there is no divergent branch around it, and the loop is bounded
  // by the command count whatever the counter holds.
  for (uint32_t pulled = 0; pulled <= count; ++pulled) {
    const uint32_t old = atomicAdd(a.ticket, lane == 0u ? 1u : 0u);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);  // wave-uniform: scalar loads below
    if (c >= count) break;

    const uint32_t index_count = a.cmds[c * kCmdWords + 0];
    const uint32_t first_index = a.cmds[c * kCmdWords + 2];
    const int32_t vertex_offset = (int32_t)a.cmds[c * kCmdWords + 3];
    const uint32_t instance = a.cmds[c * kCmdWords + 4] - a.first_instance_base;
    const uint32_t src_tri = a.src_index_offset[c] / 3u;  // index_buffer[indexOffset / 3 + id]
    const uint32_t n_tris = index_count / 3u;
    float model[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 col = a.model[(size_t)instance * 4 + q];
      model[q * 4 + 0] = col.x; model[q * 4 + 1] = col.y; model[q * 4 + 2] = col.z; model[q * 4 + 3] = col.w;
    }
    const bool affine = model_is_affine(model, a.geometry_finite);
    const bool fits = (unsigned long long)first_index + index_count <= a.capacity;
    if (!fits && lane == 0) raise_error(a.error_flag, kErrIndexOverflow);
    const size_t dst_tri = (size_t)first_index / 3u;
    const uint32_t* tri_indices = a.indices + (size_t)src_tri * 3;
    uint32_t survivors = 0;

    // one loop per path: the choice is per command, not per step
    auto walk = [&](auto affine_tag) {
      constexpr bool kAffine = decltype(affine_tag)::value;
      // software pipeline: the index triple of step k+1 is in flight while step k gathers and tests
      uint32_t n0 = 0, n1 = 0, n2 = 0;
      if (lane < n_tris) {
        const uint32_t* ip = tri_indices + (size_t)lane * 3;
        n0 = ip[0]; n1 = ip[1]; n2 = ip[2];
      }
      for (uint32_t t0 = 0; t0 < n_tris; t0 += 64u) {
        const uint32_t t = t0 + lane;
        const bool valid = t < n_tris;
        const uint32_t i0 = n0, i1 = n1, i2 = n2;
        if (t + 64u < n_tris) {
          const uint32_t* ip = tri_indices + (size_t)(t + 64u) * 3;
          n0 = ip[0]; n1 = ip[1]; n2 = ip[2];
        }
        float v[9];
        triangle_fetch(a.vertices, (long long)vertex_offset, i0, i1, i2, v);
        const bool keep = valid && !triangle_test<kAffine>(model, pv, v);
        const unsigned long long mask = __ballot(keep);
        if (keep && fits) {
          uint32_t* dst = a.out_indices + (dst_tri + survivors + lanes_below(mask)) * 3;
          dst[0] = i0; dst[1] = i1; dst[2] = i2;
        }
        survivors += (uint32_t)__popcll(mask);
      }
    };
    if (affine) walk(std::true_type{});
    else walk(std::false_type{});
    if (lane == 0) a.cmds[c * kCmdWords + 0] = survivors * 3u;  // the command's final indexCount
  }
}

// compact_draw_stream.comp runs after generate_work: commands whose triangles all died are
// dropped, order kept. One workgroup of kThreads threads walks the (already dense) list.
template <uint32_t kThreads>
__device__ __forceinline__ void recompact_commands(const uint32_t* in_cmds, const uint32_t* index_count, uint32_t count, uint32_t* out_cmds,
                                                   uint32_t* out_count, uint32_t (&s_totals)[kThreads / 64], uint32_t& s_running) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) s_running = 0;
  __syncthreads();
  for (uint32_t base = 0; base < count; base += kThreads) {
    const uint32_t k = base + tid;
    const bool valid = k < count;
    uint32_t w[kCmdWords];
#pragma unroll
    for (uint32_t f = 0; f < kCmdWords; ++f) w[f] = valid ? in_cmds[(size_t)k * kCmdWords + f] : 0u;
    if (index_count && valid) w[0] = index_count[k];
    const bool keep = valid && w[0] > 0u;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_totals[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = s_running, total = 0;
#pragma unroll
    for (uint32_t q = 0; q < kThreads / 64; ++q) {
      if (q < wave) before += s_totals[q];
      total += s_totals[q];
    }
    if (keep) {
      uint32_t* dst = out_cmds + (size_t)(before + lanes_below(mask)) * kCmdWords;
#pragma unroll
      for (uint32_t f = 0; f < kCmdWords; ++f) dst[f] = w[f];
    }
    __syncthreads();
    if (tid == 0) s_running += total;
    __syncthreads();
  }
  if (tid == 0) *out_count = s_running;
}


// Its own launch after the wave-per-command kernel (large frames).
__global__ __launch_bounds__(1024) void mip_recompact_kernel(const RecompactArgs a) {
  __shared__ uint32_t s_totals[16];
  __shared__ uint32_t s_running;
  recompact_commands<1024>(a.in_cmds, a.index_count, *a.in_count, a.out_cmds, a.out_count, s_totals, s_running);
}

// Large frames: the same re-compaction over many workgroups, as three small launches — per-block
// survivor counts, one block scanning them, the scatter (a single workgroup takes 44 us at 27 k
// commands and 0.4 ms at 258 k).

__global__ __launch_bounds__(1024) void mip_recompact_count_kernel(const RecompactWideArgs a) {
  __shared__ uint32_t s_totals[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t count = *a.in_count;
  const uint32_t k = blockIdx.x * 1024u + tid;
  const bool keep = k < count && (a.index_count ? a.index_count[k] : a.in_cmds[(size_t)k * kCmdWords]) > 0u;
  const unsigned long long mask = __ballot(keep);
  if (lane == 0) s_totals[wave] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) total += s_totals[q];
    a.block_base[blockIdx.x] = total;
  }
}

__global__ __launch_bounds__(1024) void mip_recompact_scan_kernel(const RecompactWideArgs a) {
  __shared__ uint32_t s_totals[16];
  __shared__ uint32_t s_running;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) s_running = 0;
  __syncthreads();
  for (uint32_t base = 0; base < a.n_blocks; base += 1024u) {
    const uint32_t b = base + tid;
    const uint32_t v = b < a.n_blocks ? a.block_base[b] : 0u;
    const uint32_t incl = wave_inclusive_scan(v);
    if (lane == 63u) s_totals[wave] = incl;
    __syncthreads();
    uint32_t before = s_running, total = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
      if (q < wave) before += s_totals[q];
      total += s_totals[q];
    }
    if (b < a.n_blocks) a.block_base[b] = before + incl - v;
    __syncthreads();
    if (tid == 0) s_running += total;
    __syncthreads();
  }
  if (tid == 0) *a.out_count = s_running;
}

__global__ __launch_bounds__(1024) void mip_recompact_scatter_kernel(const RecompactWideArgs a) {
  __shared__ uint32_t s_totals[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t count = *a.in_count;
  const uint32_t k = blockIdx.x * 1024u + tid;
  const bool valid = k < count;
  uint32_t w[kCmdWords];
#pragma unroll
  for (uint32_t f = 0; f < kCmdWords; ++f) w[f] = valid ? a.in_cmds[(size_t)k * kCmdWords + f] : 0u;
  if (a.index_count && valid) w[0] = a.index_count[k];
  const bool keep = valid && w[0] > 0u;
  const unsigned long long mask = __ballot(keep);
  if (lane == 0) s_totals[wave] = (uint32_t)__popcll(mask);
  __syncthreads();
  uint32_t before = a.block_base[blockIdx.x];
#pragma unroll
  for (uint32_t q = 0; q < 16; ++q)
    if (q < wave) before += s_totals[q];
  if (keep) {
    uint32_t* dst = a.out_cmds + (size_t)(before + lanes_below(mask)) * kCmdWords;
#pragma unroll
    for (uint32_t f = 0; f < kCmdWords; ++f) dst[f] = w[f];
  }
}

// Small frames (the reference's own regime: tens to a few thousand commands) leave a
// wave-per-command launch mostly idle and make one wave walk a 15 k-triangle mesh alone
// (measured 0.1 ms for 20 commands). There ONE WORKGROUP of 1024 threads takes a command:
// 1024 triangles per step, survivors ordered by a ballot per wave + the 16 wave totals in LDS.
template <uint32_t kTriBlock>
__global__ __launch_bounds__(kTriBlock) void mip_triangle_cull_block_kernel(const TriangleArgs a) {
  __shared__ uint32_t s_wave[2][kTriBlock / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (a.index_total && !tri_choice_is_block(a)) return;  // this frame is the wave-per-command grid's
  const uint32_t count = *a.count;
  float pv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) pv[k] = a.pv[k];

  // Many more commands than workgroups (a.pull_tickets, set by the plan): commands are pulled from the counter instead of
  // dealt by a static stride — a workgroup's 13 commands of a 100 k-instance frame differ by +-18 % in triangles between
  // workgroups, and the launch waits for the unluckiest (round 4, profiles/r04_triangle_block_tickets.txt: one-mesh scene
  // 100 k instances 1.08 -> 1.03 ms, mixed 200 k 0.89 -> 0.75, mixed 1 M 3.70 -> 3.31; below ~30 k instances the stride wins).
  __shared__ uint32_t s_ticket;
  const bool ticketed = a.pull_tickets != 0u;
  // Consecutive commands per ticket: same-address returning atomics are served at ~11 ns each, so a quarter of a million
  // tickets ARE the launch (mixed scene, 1 M instances, 258 k commands: 3.32 -> 2.81 ms with four commands per ticket;
  // 400 k: 1.38 -> 1.27; below ~60 k commands single tickets balance better: profiles/r04_triangle_block_tickets.txt)
  const uint32_t batch = ticketed && count >= a.pull_tickets ? 4u : 1u;
  for (uint32_t c = blockIdx.x, pulled = 0; pulled <= count; c += gridDim.x, ++pulled) {
    if (ticketed) {
      if (pulled % batch == 0u) {
        __syncthreads();  // everybody has read the previous ticket
        if (tid == 0) s_ticket = atomicAdd(a.ticket, batch);
        __syncthreads();
        c = s_ticket;
      } else {
        c = c - gridDim.x + 1u;  // the next command of the batch
      }
    }
    if (c >= count) { if (ticketed && pulled % batch != batch - 1u) continue; break; }
    const uint32_t index_count = a.cmds[c * kCmdWords + 0];
    const uint32_t first_index = a.cmds[c * kCmdWords + 2];
    const int32_t vertex_offset = (int32_t)a.cmds[c * kCmdWords + 3];
    const uint32_t instance = a.cmds[c * kCmdWords + 4] - a.first_instance_base;
    const uint32_t src_tri = a.src_index_offset[c] / 3u;
    const uint32_t n_tris = index_count / 3u;
    float model[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 col = a.model[(size_t)instance * 4 + q];
      model[q * 4 + 0] = col.x; model[q * 4 + 1] = col.y; model[q * 4 + 2] = col.z; model[q * 4 + 3] = col.w;
    }
    const bool affine = model_is_affine(model, a.geometry_finite);
    const bool fits = (unsigned long long)first_index + index_count <= a.capacity;
    if (!fits && tid == 0) raise_error(a.error_flag, kErrIndexOverflow);
    const size_t dst_tri = (size_t)first_index / 3u;
    const uint32_t* tri_indices = a.indices + (size_t)src_tri * 3;
    uint32_t survivors = 0, buf = 0;
    __syncthreads();  // the previous command's last totals have been read
    // 256/512-thread variants: two-stage software pipeline — the positions of step k+1 and the index
    // triple of step k+2 are in flight while step k is transformed and tested (5 k instances 0.108 ->
    // 0.097 ms, mixed scene at 20 k 0.188 -> 0.165 ms). The 1024-thread variant has 128 VGPRs per lane
    // and would spill: it fetches inside the step.
    constexpr bool kPipelined = kTriBlock <= 512u;
    auto fetch_indices = [&](uint32_t t, uint32_t& j0, uint32_t& j1, uint32_t& j2) {
      j0 = 0; j1 = 0; j2 = 0;  // idle lanes read vertex 0 of the mesh: in bounds
      if (t < n_tris) {
        const uint32_t* ip = tri_indices + (size_t)t * 3;
        j0 = ip[0]; j1 = ip[1]; j2 = ip[2];
      }
    };
    uint32_t c0 = 0, c1 = 0, c2 = 0, n0 = 0, n1 = 0, n2 = 0;
    float cv[9], nv[9];
    if constexpr (kPipelined) {
      fetch_indices(tid, c0, c1, c2);
      fetch_indices(tid + kTriBlock, n0, n1, n2);
      triangle_fetch(a.vertices, (long long)vertex_offset, c0, c1, c2, cv);
    }
    for (uint32_t t0 = 0; t0 < n_tris; t0 += kTriBlock, buf ^= 1u) {
      const uint32_t t = t0 + tid;
      const bool valid = t < n_tris;
      uint32_t i0, i1, i2;
      if constexpr (kPipelined) {
        i0 = c0; i1 = c1; i2 = c2;
        c0 = n0; c1 = n1; c2 = n2;
        triangle_fetch(a.vertices, (long long)vertex_offset, c0, c1, c2, nv);  // step k+1
        fetch_indices(t + 2u * kTriBlock, n0, n1, n2);                          // step k+2
      } else {
        fetch_indices(t, i0, i1, i2);
        triangle_fetch(a.vertices, (long long)vertex_offset, i0, i1, i2, cv);
      }
      const bool keep = valid && !(affine ? triangle_test<true>(model, pv, cv) : triangle_test<false>(model, pv, cv));
      if constexpr (kPipelined) {
#pragma unroll
        for (int q = 0; q < 9; ++q) cv[q] = nv[q];
      }
      const unsigned long long mask = __ballot(keep);
      if (lane == 0) s_wave[buf][wave] = (uint32_t)__popcll(mask);
      __syncthreads();  // one barrier per step: the totals alternate between two buffers
      uint32_t before = 0, total = 0;
#pragma unroll
      for (uint32_t w = 0; w < kTriBlock / 64; ++w) {
        const uint32_t v = s_wave[buf][w];
        if (w < wave) before += v;
        total += v;
      }
      if (keep && fits) {
        uint32_t* dst = a.out_indices + (dst_tri + survivors + before + lanes_below(mask)) * 3;
        dst[0] = i0; dst[1] = i1; dst[2] = i2;
      }
      survivors += total;
    }
    if (tid == 0) a.cmds[c * kCmdWords + 0] = survivors * 3u;
  }
}

// ---------------------------------------------------------------------------------------
// Small frames — the reference's own regime (its demo scene has 30 entities, its buffers hold 2 400 commands):
// fewer commands than the chip has workgroup slots, so one workgroup per command leaves most CUs idle and walks a
// 15 k-triangle mesh in 15 dependent steps. Here every command is cut into kTriParts = 16 equal PARTS of its
// triangle range; a part is one work item (item w = command w / 16, part w % 16) of a 256-thread workgroup, up to
// kTriPartMaxT triangles per thread, all of them in registers. A part tests its triangles,
// publishes its survivor count as one tagged granule, reads the <= 15 earlier parts of its command (one round trip
// in the common case), and writes its survivors behind theirs: the stream keeps mesh order, exactly as the
// one-wave and one-workgroup kernels produce it. The last part writes the command's final indexCount.


__global__ __launch_bounds__(256, 4) void mip_triangle_cull_parts_kernel(const TrianglePartsArgs pa) {
  const TriangleArgs& a = pa.t;
  __shared__ uint32_t s_prefix;
  __shared__ uint32_t s_counts[kTriPartMaxT][4];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t count = *a.count;
  float pv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) pv[k] = a.pv[k];
  const uint32_t items = count * kTriParts;

  // Items are dealt by a static stride over a grid the host sizes to be resident as a whole, so a part's predecessors
  // normally belong to workgroups that are running and have published by the time they are looked at; nothing depends on
  // that: a part that has not published is counted by the wave that needs it (below). (A ticket counter was measured
  // first: one returning atomic per item on one address serialises at ~11 ns each, 82 against 48 us at 1 000 instances.)
  for (uint32_t dealt = blockIdx.x; dealt < items; dealt += gridDim.x) {
    __syncthreads();  // s_counts / s_prefix of the previous item have been read
    uint32_t item = dealt;
#ifdef MIP_DEBUG_STAMPS
    if (pa.debug_reverse) item = items - 1u - dealt;
#endif
    const uint32_t c = item / kTriParts, part = item % kTriParts;

    const uint32_t index_count = a.cmds[c * kCmdWords + 0];
    const uint32_t first_index = a.cmds[c * kCmdWords + 2];
    const int32_t vertex_offset = (int32_t)a.cmds[c * kCmdWords + 3];
    const uint32_t instance = a.cmds[c * kCmdWords + 4] - a.first_instance_base;
    const uint32_t src_tri = a.src_index_offset[c] / 3u;
    const uint32_t n_tris = index_count / 3u;
    const uint32_t per_part = (n_tris + kTriParts - 1u) / kTriParts;
    const uint32_t t_begin = part * per_part < n_tris ? part * per_part : n_tris;
    const uint32_t t_end = t_begin + per_part < n_tris ? t_begin + per_part : n_tris;
    float model[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 col = a.model[(size_t)instance * 4 + q];
      model[q * 4 + 0] = col.x; model[q * 4 + 1] = col.y; model[q * 4 + 2] = col.z; model[q * 4 + 3] = col.w;
    }
    const bool affine = model_is_affine(model, a.geometry_finite);
    const bool fits = (unsigned long long)first_index + index_count <= a.capacity;
    if (!fits && tid == 0 && part == 0) raise_error(a.error_flag, kErrIndexOverflow);
    const uint32_t* tri_indices = a.indices + (size_t)src_tri * 3;

    // ---- test this part's triangles: triangle t_begin + k*256 + tid in step k, kept in registers ----
    uint32_t i0[kTriPartMaxT], i1[kTriPartMaxT], i2[kTriPartMaxT];
    uint32_t keep_bits = 0;
    unsigned long long masks[kTriPartMaxT];
#pragma unroll
    for (uint32_t k = 0; k < kTriPartMaxT; ++k) {
      masks[k] = 0ull;
      i0[k] = i1[k] = i2[k] = 0u;
      if (t_begin + k * 256u < t_end) {  // uniform: this step exists for the part
        const uint32_t t = t_begin + k * 256u + tid;
        const bool valid = t < t_end;
        if (valid) {
          const uint32_t* ip = tri_indices + (size_t)t * 3;
          i0[k] = ip[0]; i1[k] = ip[1]; i2[k] = ip[2];
        }
        float v[9];
        triangle_fetch(a.vertices, (long long)vertex_offset, i0[k], i1[k], i2[k], v);  // idle lanes read vertex 0 of the mesh: in bounds
        const bool keep = valid && !(affine ? triangle_test<true>(model, pv, v) : triangle_test<false>(model, pv, v));
        masks[k] = __ballot(keep);
        keep_bits |= keep ? (1u << k) : 0u;
        if (lane == 0) s_counts[k][wave] = (uint32_t)__popcll(masks[k]);
      } else if (lane == 0) {
        s_counts[k][wave] = 0u;
      }
    }
    __syncthreads();
    // survivors before this thread's triangle of step k: all earlier steps, earlier waves of the step, earlier lanes
    uint32_t before[kTriPartMaxT], total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kTriPartMaxT; ++k) {
      uint32_t in_step = 0, mine = total;
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t cnt = s_counts[k][w];
        if (w < wave) mine += cnt;
        in_step += cnt;
      }
      before[k] = mine + lanes_below(masks[k]);
      total += in_step;
    }

    // ---- publish, then the survivors of the earlier parts of this command ----
    unsigned long long* status = pa.part_status + (size_t)c * kTriParts;
#ifdef MIP_DEBUG_STAMPS
    const bool skip_publish = pa.debug_skip_part == part + 1u;  // fault injection: this part never publishes
#else
    const bool skip_publish = false;
#endif
    if (tid == 0 && !skip_publish)
      __hip_atomic_store(&status[part], ((unsigned long long)pa.epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prefix = 0;
    if (wave == 0) {
      const bool need = lane < part;
      bool ready = !need;
      uint32_t got = 0, polls = 0;
      for (;;) {
        if (!ready) {
          const unsigned long long g = status_load(&status[lane]);
          if ((uint32_t)(g >> 32) == pa.epoch) { ready = true; got = (uint32_t)g; }
        }
        if (__all(ready)) break;
        if (__builtin_expect(++polls > kPatientPolls, 0)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      // An earlier part that has not published within the patient polls is not waited for (its workgroup may not be
      // running: instance_kernel.hpp, "no wait depends on another workgroup ever running"): this wave counts that part's
      // survivors itself — the same triangles through the same test, 64 per step — and publishes the count for it.
      unsigned long long missing = __ballot(!ready);
      while (__builtin_expect(missing != 0ull, 0)) {  // wave-uniform
        const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(missing));
        unsigned long long g = status_load(&status[p]);
        if ((uint32_t)(g >> 32) != pa.epoch) {
          const uint32_t b = p * per_part < n_tris ? p * per_part : n_tris;
          const uint32_t e = b + per_part < n_tris ? b + per_part : n_tris;
          uint32_t survivors = 0;
          for (uint32_t t0 = b; t0 < e; t0 += 64u) {
            const uint32_t t = t0 + lane;
            const bool valid = t < e;
            uint32_t j0 = 0, j1 = 0, j2 = 0;
            if (valid) {
              const uint32_t* ip = tri_indices + (size_t)t * 3;
              j0 = ip[0]; j1 = ip[1]; j2 = ip[2];
            }
            const bool keep = valid && !triangle_culled(affine, model, pv, a.vertices, (long long)vertex_offset, j0, j1, j2);
            survivors += (uint32_t)__popcll(__ballot(keep));
          }
          g = ((unsigned long long)pa.epoch << 32) | survivors;
          if (lane == 0u) {
            __hip_atomic_store(&status[p], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(a.help_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (lane == p) got = (uint32_t)g;
        missing &= ~(1ull << p);
      }
      prefix = wave_sum(got);
      if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    prefix = s_prefix;

    // ---- write the survivors behind those of the earlier parts ----
    const size_t dst_tri = (size_t)first_index / 3u + prefix;
    if (fits) {
#pragma unroll
      for (uint32_t k = 0; k < kTriPartMaxT; ++k)
        if ((keep_bits >> k) & 1u) {
          uint32_t* dst = a.out_indices + (dst_tri + before[k]) * 3;
          dst[0] = i0[k]; dst[1] = i1[k]; dst[2] = i2[k];
        }
    }
    // the command's final indexCount — beside the command, never into it: a part of this command that has not started yet (its
    // successors have helped themselves past it) still needs the ORIGINAL indexCount to find its triangles
    if (part == kTriParts - 1u && tid == 0) a.final_index_count[c] = (prefix + total) * 3u;
  }
}

}  // namespace mip
