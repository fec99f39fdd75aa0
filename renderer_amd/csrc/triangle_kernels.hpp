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

// ---- one command as the walking kernels see it, and the walk itself ----
struct ChunkCmd {  // one command as a range sees it (wave-uniform)
  uint32_t index_count, first_index, instance, src_tri, n_tris, slot0;
  int32_t vertex_offset;
};

// (The kernel stores through other pointers, so the compiler cannot prove these wave-uniform loads unclobbered and issues
// them as vector loads; readfirstlane at least returns the words to scalar registers — seven VGPRs per command in flight otherwise.)
__device__ __forceinline__ uint32_t uniform_word(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uniform_word(float v) { return __uint_as_float(uniform_word(__float_as_uint(v))); }

__device__ __forceinline__ ChunkCmd chunk_load_cmd(const TriangleArgs& a, uint32_t base, uint32_t c) {
  ChunkCmd m;
  m.index_count = uniform_word(a.cmds[c * kCmdWords + 0]);
  m.first_index = uniform_word(a.cmds[c * kCmdWords + 2]);
  m.vertex_offset = (int32_t)uniform_word(a.cmds[c * kCmdWords + 3]);
  m.instance = uniform_word(a.cmds[c * kCmdWords + 4]) - a.first_instance_base;
  m.src_tri = uniform_word(a.src_index_offset[c]) / 3u;
  m.n_tris = m.index_count / 3u;
  m.slot0 = (m.first_index - base) / 3u;
  return m;
}

__device__ __forceinline__ void chunk_load_model(const TriangleArgs& a, uint32_t instance, float (&model)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 col = a.model[(size_t)instance * 4 + q];
    model[q * 4 + 0] = uniform_word(col.x); model[q * 4 + 1] = uniform_word(col.y); model[q * 4 + 2] = uniform_word(col.z); model[q * 4 + 3] = uniform_word(col.w);
  }
}

// Triangles [t_begin, t_end) of command m, 64 per step.
//   kDirect   survivors go straight to their place (dst_tri + running count), as in the wave-per-command kernel;
//   !kDirect  nothing is written yet: the step's 64-bit keep mask goes to masks[step].
// Returns the survivors.
// The step is a software pipeline with NO conditional memory operation in it — the wait counter is in order, and the
// compiler has to assume that a load or store inside a branch may not have been issued, so it waits for everything that
// is older (round 4's loop waited for the store it had issued a moment ago, and for the next step's indices, before it
// touched the first vertex):
//   gathers of step k  ->  store of step k-1  ->  index triple of step k+1  ->  wait for the gathers only  ->  test k.
// The store goes through a buffer descriptor over the command's own region of the stream: a lane without a survivor
// gets an offset beyond it and is dropped by the hardware (no branch); a command that does not fit the index buffer gets
// an empty region. The index prefetch of the last step re-reads that step's own triple (in bounds, never used).
typedef unsigned int tri_u32x3 __attribute__((ext_vector_type(3)));
constexpr uint32_t kTriRegionMax = 0x7ffffff0u;   // bytes a region descriptor spans at most (a command of 178 M triangles)
constexpr uint32_t kTriDropOffset = 0x80000000u;  // beyond every region: the store is dropped
// Cache policy of the culled stream's stores (the builtin's last operand: 2 = nt, 16 = sc1): written once, gigabytes per frame, read by
// nobody in this launch — non-temporal keeps it from pushing the meshes' indices and positions out of the L2 (mixed 1 M instances 2.41 ->
// 2.24 ms, mixed 100 k 0.369 -> 0.362, one-mesh 100 k unchanged; `sc1 nt`, the frame kernel's policy for its matrices, LOSES here: 3.2 ms —
// profiles/r05_triangle_store_policy.txt).
#ifndef MIP_TRI_STORE_AUX
#define MIP_TRI_STORE_AUX 2
#endif

// Addresses: the index triples and the positions are read through buffer descriptors too — over the command's own index range and
// over the vertex buffer from the mesh's vertex_offset on — with 32-bit byte offsets: `t * 12` advances by one add per step, a
// corner's `i * 12` is a shift and a shift-add. (Round 4's loop built a 64-bit address per load: four v_mad_u64_u32, three
// v_lshl_add_u64 and a v_mul_lo_u32 per step — quarter-rate instructions, ~27 of the step's ~190 issue slots.) A triangle number
// beyond the command reads zeros (vertex 0 of the mesh), so idle lanes need no clamp.
// (Written as the two full-rate instructions it is: the compiler folds `(i << 3) + (i << 2)` back into a quarter-rate v_mul_lo_u32.)
__device__ __forceinline__ uint32_t times12(uint32_t i) {
  uint32_t r;
  asm("v_lshlrev_b32 %0, 2, %1\n\tv_lshl_add_u32 %0, %1, 3, %0" : "=&v"(r) : "v"(i));
  return r;
}

template <bool kAffine, bool kDirect>
__device__ __forceinline__ uint32_t chunk_walk(const TriangleArgs& a, const ChunkCmd& m, const float (&model)[16], const float (&pv)[16],
                                               uint32_t t_begin, uint32_t t_end, bool fits, size_t dst_tri, unsigned long long* masks, uint32_t lane) {
  const unsigned long long index_bytes = (unsigned long long)m.n_tris * 12ull;
  const __amdgpu_buffer_rsrc_t idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.indices) + (size_t)m.src_tri * 3, 0,
      (int)(index_bytes < kTriRegionMax ? (uint32_t)index_bytes : kTriRegionMax), 0x00020000);
  const long long vertex_first = (long long)m.vertex_offset * 12ll;  // bytes; the positions of this mesh start here
  const unsigned long long vertex_left = vertex_first < (long long)a.vertex_bytes ? a.vertex_bytes - (unsigned long long)vertex_first : 0ull;
  const __amdgpu_buffer_rsrc_t vtx = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(a.vertices)) + vertex_first, 0,
      (int)(vertex_left < kTriRegionMax ? (uint32_t)vertex_left : kTriRegionMax), 0x00020000);
  const unsigned long long region = (unsigned long long)(t_end - t_begin) * 12ull;
  const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(a.out_indices + dst_tri * 3, 0,
      (int)(kDirect && fits ? (region < kTriRegionMax ? (uint32_t)region : kTriRegionMax) : 0u), 0x00020000);
  uint32_t survivors = 0, step = 0;
  uint32_t t_offset = times12(t_begin + lane);  // byte offset of this lane's triangle of the step in the command's index range
  tri_u32x3 next = __builtin_amdgcn_raw_buffer_load_b96(idx, (int)t_offset, 0, 0);
  tri_u32x3 q = {0u, 0u, 0u};          // the previous step's survivors, stored behind this step's gathers
  uint32_t q_offset = kTriDropOffset;
  for (uint32_t t0 = t_begin; t0 < t_end; t0 += 64u, ++step) {
    const bool valid = t0 + lane < t_end;
    const tri_u32x3 i = next;
    const tri_u32x3 v0 = __builtin_amdgcn_raw_buffer_load_b96(vtx, (int)times12(i.x), 0, 0);
    const tri_u32x3 v1 = __builtin_amdgcn_raw_buffer_load_b96(vtx, (int)times12(i.y), 0, 0);
    const tri_u32x3 v2 = __builtin_amdgcn_raw_buffer_load_b96(vtx, (int)times12(i.z), 0, 0);
    if constexpr (kDirect) __builtin_amdgcn_raw_buffer_store_b96(q, out, (int)q_offset, 0, MIP_TRI_STORE_AUX);
    t_offset += 768u;
    next = __builtin_amdgcn_raw_buffer_load_b96(idx, (int)t_offset, 0, 0);  // (beyond the command: zeros)
    const float v[9] = {__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v1.x), __uint_as_float(v1.y),
                        __uint_as_float(v1.z), __uint_as_float(v2.x), __uint_as_float(v2.y), __uint_as_float(v2.z)};
    const bool keep = !triangle_test<kAffine>(model, pv, v) && valid;
    const unsigned long long mask = __ballot(keep);
    if constexpr (kDirect) {
      q = i;
      q_offset = keep ? times12(survivors + lanes_below(mask)) : kTriDropOffset;
    } else {
      if (lane == 0u) masks[step] = mask;  // (the range kernel sizes the buffer for its longest range)
    }
    survivors += (uint32_t)__popcll(mask);
  }
  if constexpr (kDirect) __builtin_amdgcn_raw_buffer_store_b96(q, out, (int)q_offset, 0, MIP_TRI_STORE_AUX);
  return survivors;
}

// Round 5, frames above tri_block_max instances: the range kernel's grid (choice_mode 1) and the wave-per-command grid over the
// sorted commands (choice_mode 2) are both launched; every workgroup of both — and of the sort and map kernels in front of them
// — asks this (wave-uniform, from the slot's own command list), and the grid the frame is not for returns at once.
__device__ __forceinline__ uint32_t stream_slots(const TriangleArgs& a, uint32_t base, uint32_t count) {
  return (a.cmds[(count - 1u) * kCmdWords + 2] - base) / 3u + a.cmds[(count - 1u) * kCmdWords + 0] / 3u;
}
__device__ __forceinline__ bool tri_not_this_grid(const TriangleArgs& a, uint32_t count) {
  if (a.choice_mode == 0u || count == 0u) return false;
  const bool ranges = plan_tri_choice_is_ranges(a.max_lod_tris, stream_slots(a, a.first_index_base, count), count, a.choice_waves);
  return ranges != (a.choice_mode == 1u);
}

// Large frames launch the wave-per-command grid AND a workgroup-per-command grid; every workgroup of both asks this
// (wave-uniform, three scalar loads) and one grid returns at once. The rule and its measurements: frame_plan.hpp.
__device__ __forceinline__ bool tri_choice_is_block(const TriangleArgs& a) {
  return plan_tri_choice_is_block(a.max_lod_tris, *a.index_total, *a.count);
}

#ifndef MIP_TRI_MIN_WAVES_PER_SIMD
#define MIP_TRI_MIN_WAVES_PER_SIMD 4
#endif

#ifndef MIP_TRI_WAVE_KERNEL_WAVES_PER_SIMD
#define MIP_TRI_WAVE_KERNEL_WAVES_PER_SIMD 8
#endif
__device__ __forceinline__ uint32_t tri_size_class(uint32_t index_count) {
  const uint32_t n_tris = index_count / 3u;
  return n_tris ? 31u - (uint32_t)__builtin_clz(n_tris) : 0u;
}

// One wave per command (the body of mip_triangle_cull_kernel, and of mip_triangle_stage_kernel when the frame is not the range kernel's).
__device__ __forceinline__ void triangle_waves_body(const TriangleArgs& a, uint32_t count) {
  const uint32_t lane = threadIdx.x & 63u;
  float pv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) pv[k] = a.pv[k];

  // Commands differ 1000x in triangle count (LODs, mixed meshes): waves pull their work from a ticket counter instead of
  // striding over the list (measured: static striding left a third of the waves idle at 5 k commands). The counter is
  // zeroed by the host per launch. Every lane takes part in the add (lane 0 adds 1, the others 0: the compiler folds the
  // wave's adds into one atomic), so there is no divergent branch around it, and the loop is bounded by the command count
  // whatever the counter holds.
  // Round 5 (a.order): a ticket is a run of commands of ONE size class in the list the sort kernels wrote, largest class
  // first — the launch no longer ends with waves walking their last long command alone (19 % of the 100 k frame in round
  // 4) — and a ticket of a small class is several commands (a same-address atomic takes ~11 ns: one per command was the
  // whole launch at 258 k commands). Which wave walks which command changes nothing in the stream.
  const bool sorted = a.order != nullptr;
  const uint32_t n_tickets = sorted ? a.sort_info[kSortTickets + 31u] : count;
#ifdef MIP_EXP_RANGE_TIMES  // experiment build (tools/r05_range_times.py): per wave {start, end, commands, triangles, first ticket end, -, -, -} behind the index buffer's capacity
  uint32_t* exp_at = a.out_indices + a.capacity + 8u * (blockIdx.x * 4u + (threadIdx.x >> 6));
  uint32_t exp_cmds = 0, exp_tris = 0;
  if (lane == 0u) exp_at[0] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
  for (uint32_t pulled = 0; pulled <= count; ++pulled) {
    const uint32_t old = atomicAdd(a.ticket, lane == 0u ? 1u : 0u);
    const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);  // wave-uniform: scalar loads below
    if (ticket >= n_tickets) break;
    uint32_t first = ticket, n_cmds = 1;
    if (sorted) {
      uint32_t d = 0, before = 0;
      for (; d < 31u; ++d) {  // (n_tickets > ticket: the loop ends at the class that holds it)
        const uint32_t upto = a.sort_info[kSortTickets + d];
        if (ticket < upto) break;
        before = upto;
      }
      const uint32_t batch = a.sort_info[kSortBatch + d];
      const uint32_t class_end = d < 31u ? a.sort_info[kSortStart + d + 1u] : count;
      first = a.sort_info[kSortStart + d] + (ticket - before) * batch;
      n_cmds = class_end - first < batch ? class_end - first : batch;
    }
    // The SIMD issues its OLDEST wave first: left alone, the youngest wave of a SIMD walks its first command for most of the
    // launch (first command done after 207 .. 943 us of a 990 us launch, 1 .. 8 commands per wave) and the launch ends with
    // those waves alone on their SIMDs. A wave that has pulled fewer tickets asks for a higher issue priority: 196 .. 462 us,
    // 3 .. 4 commands per wave (profiles/r05_wave_kernel_lifetimes.txt).
    if (pulled == 0u) __builtin_amdgcn_s_setprio(3);
    else if (pulled == 1u) __builtin_amdgcn_s_setprio(2);
    else if (pulled == 2u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    for (uint32_t k = 0; k < n_cmds; ++k) {
      const uint32_t c = sorted ? a.order[first + k] : first + k;
      const ChunkCmd m = chunk_load_cmd(a, 0u, c);
      float model[16];
      chunk_load_model(a, m.instance, model);
      const bool affine = model_is_affine(model, a.geometry_finite);
      const bool fits = (unsigned long long)m.first_index + m.index_count <= a.capacity;
      if (!fits && lane == 0) raise_error(a.error_flag, kErrIndexOverflow);
      // one loop per path: the choice is per command, not per step
      const uint32_t survivors = m.n_tris == 0u ? 0u
                                 : affine ? chunk_walk<true, true>(a, m, model, pv, 0u, m.n_tris, fits, (size_t)m.first_index / 3u, nullptr, lane)
                                          : chunk_walk<false, true>(a, m, model, pv, 0u, m.n_tris, fits, (size_t)m.first_index / 3u, nullptr, lane);
      // the command's final indexCount: beside the command when the re-compaction is told to look there (it must, when another
      // grid may have taken the frame instead), else into it
      if (lane == 0) (a.final_index_count ? a.final_index_count[c] : a.cmds[c * kCmdWords + 0]) = survivors * 3u;
#ifdef MIP_EXP_RANGE_TIMES
      if (lane == 0u && exp_cmds == 0u) exp_at[4] = (uint32_t)__builtin_amdgcn_s_memrealtime();
      exp_cmds += 1u; exp_tris += m.n_tris;
#endif
    }
  }
#ifdef MIP_EXP_RANGE_TIMES
  if (lane == 0u) { exp_at[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); exp_at[2] = exp_cmds; exp_at[3] = exp_tris; }
#endif
}

__global__ __launch_bounds__(256, MIP_TRI_WAVE_KERNEL_WAVES_PER_SIMD) void mip_triangle_cull_kernel(const TriangleArgs a) {
  if (a.index_total && tri_choice_is_block(a)) return;  // (round 4's pairing) this frame is the workgroup-per-command grid's
  const uint32_t count = *a.count;
  if (tri_not_this_grid(a, count)) return;
  triangle_waves_body(a, count);
}

// ---- commands by descending size class (round 5) ----
// count (a part of mip_triangle_prepare_kernel): a histogram of the classes — LDS per workgroup, one global add per class and
// workgroup into the workgroup's COPY of the histogram. scatter: every workgroup turns the histogram into positions (32 classes:
// a scan across half a wave, redundantly in every workgroup — no "last workgroup" counter), reserves its commands' places with one
// add per class on its copy's cursor and writes them; workgroup 0 also leaves positions, tickets and batch sizes for the stage. The
// order inside a class is whatever the adds make it: it decides which wave walks which command, nothing else.
__device__ __forceinline__ void triangle_sort_count_part(const TriangleArgs& a, uint32_t count) {
  __shared__ uint32_t s_hist[32];
  const uint32_t tid = threadIdx.x;
  if (tid < 32u) s_hist[tid] = 0u;
  __syncthreads();
  const uint32_t c = blockIdx.x * 256u + tid;
  if (c < count) atomicAdd(&s_hist[tri_size_class(a.cmds[c * kCmdWords + 0])], 1u);
  __syncthreads();
  if (tid < 32u && s_hist[tid])
    (void)__hip_atomic_fetch_add(&a.sort_info[kSortHist + (blockIdx.x % kSortCopies) * 32u + tid], s_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void mip_triangle_sort_scatter_kernel(const TriangleArgs a, uint32_t* order) {
  __shared__ uint32_t s_hist[32];
  __shared__ uint32_t s_base[32];
  const uint32_t tid = threadIdx.x;
  const uint32_t count = *a.count;
  if (tri_not_this_grid(a, count)) return;
  const uint32_t copy = blockIdx.x % kSortCopies;
  if (tid < 64u) {  // wave 0: lane d < 32 takes the class of rank d (d = 0: the largest)
    const uint32_t k = 31u - (tid & 31u);
    uint32_t total = 0, below = 0;
    if (tid < 32u) {
#pragma unroll
      for (uint32_t q = 0; q < kSortCopies; ++q) {
        const uint32_t h = a.sort_info[kSortHist + q * 32u + k];
        total += h;
        if (q < copy) below += h;
      }
    }
    // a ticket is worth ~4 096 triangles, and at most 16 commands (a small command costs its start-up latency, not its triangles)
    uint32_t batch = k >= 12u ? 1u : 4096u >> k;
    if (batch > 16u) batch = 16u;
    const uint32_t start = wave_inclusive_scan(total) - total;
    const uint32_t tickets = tid < 32u ? (total + batch - 1u) / batch : 0u;
    const uint32_t tickets_upto = wave_inclusive_scan(tickets);
    if (tid < 32u) {
      s_hist[k] = 0u;
      s_base[k] = start + below;  // where this copy's commands of the class begin
      if (blockIdx.x == 0u) {
        a.sort_info[kSortStart + tid] = start;
        a.sort_info[kSortBatch + tid] = batch;
        a.sort_info[kSortTickets + tid] = tickets_upto;
      }
    }
  }
  __syncthreads();
  const uint32_t c = blockIdx.x * 256u + tid;
  uint32_t k = 0, rank = 0;
  if (c < count) {
    k = tri_size_class(a.cmds[c * kCmdWords + 0]);
    rank = atomicAdd(&s_hist[k], 1u);
  }
  __syncthreads();
  if (tid < 32u && s_hist[tid])
    s_base[tid] += __hip_atomic_fetch_add(&a.sort_info[kSortCursor + copy * 32u + tid], s_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (c < count) order[s_base[k] + rank] = c;
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
  // the stage's counters for the NEXT frame of this slot (the stage itself is over: this kernel runs behind it on the stream)
  if (threadIdx.x < a.n_zero) a.zero_words[threadIdx.x] = 0u;
}

// Large frames (more than tri_block_max instances): the same re-compaction in ONE launch of many workgroups (round 4: three —
// counts per 1 024 commands, one workgroup scanning them, the scatter: 3 x 4.8 us of a 0.37 ms frame). A workgroup counts
// its 1 024 commands, publishes the count as a tagged granule {tag : 32 | inclusive : 1 | value : 31}, looks back — 64
// predecessors per round trip, down to the nearest one that has published its INCLUSIVE sum —, publishes its own inclusive
// sum and scatters. A predecessor that has not published after the patient polls is not waited for: its count is 1 024
// index counts away, the waiting wave computes it (no wait depends on another workgroup ever running).
constexpr unsigned long long kRecompactInclusive = 1ull << 31;
__global__ __launch_bounds__(1024) void mip_recompact_onepass_kernel(const RecompactWideArgs a) {
  __shared__ uint32_t s_totals[16];
  __shared__ uint32_t s_exclusive;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t count = *a.in_count;
  const uint32_t b = blockIdx.x;
  const uint32_t last_block = count ? (count - 1u) / 1024u : 0u;
  if (b > last_block) return;
  const uint32_t k = b * 1024u + tid;
  const bool valid = k < count;
  uint32_t w[kCmdWords];
#pragma unroll
  for (uint32_t f = 0; f < kCmdWords; ++f) w[f] = valid ? a.in_cmds[(size_t)k * kCmdWords + f] : 0u;
  if (a.index_count && valid) w[0] = a.index_count[k];
  const bool keep = valid && w[0] > 0u;
  const unsigned long long mask = __ballot(keep);
  if (lane == 0) s_totals[wave] = (uint32_t)__popcll(mask);
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (uint32_t q = 0; q < 16; ++q) {
    const uint32_t v = s_totals[q];
    if (q < wave) before += v;
    total += v;
  }
  if (wave == 0) {
    const unsigned long long tag = (unsigned long long)a.epoch << 32;
#ifdef MIP_DEBUG_STAMPS
    const bool skip_publish = a.debug_skip && (b & 3u) == 1u;  // fault injection: every fourth workgroup publishes nothing; its successors count for it
#else
    const bool skip_publish = false;
#endif
    if (lane == 0u && b > 0u && !skip_publish) __hip_atomic_store(&a.block_status[b], tag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t exclusive = 0;
    for (uint32_t pos = b; pos > 0u;) {
      const uint32_t lo = pos > 64u ? pos - 64u : 0u;
      const uint32_t j = lo + lane;
      const bool need = j < pos;
      bool ready = !need;
      unsigned long long g = 0;
      uint32_t polls = 0;
      for (;;) {
        if (!ready) {
          g = status_load(&a.block_status[j]);
          ready = (uint32_t)(g >> 32) == a.epoch;
        }
        if (__all(ready)) break;
        if (__builtin_expect(++polls > kPatientPolls, 0)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      unsigned long long missing = __ballot(!ready);
      while (__builtin_expect(missing != 0ull, 0)) {  // wave-uniform: that block's count, computed here on its owner's behalf
        const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(missing));
        const uint32_t jj = lo + q;
        uint32_t kept = 0;
        for (uint32_t i = 0; i < 16u; ++i) {
          const uint32_t kk = jj * 1024u + i * 64u + lane;
          const bool kv = kk < count && (a.index_count ? a.index_count[kk] : a.in_cmds[(size_t)kk * kCmdWords]) > 0u;
          kept += (uint32_t)__popcll(__ballot(kv));
        }
        if (lane == q) g = tag | kept;
        if (lane == 0u) {
          __hip_atomic_store(&a.block_status[jj], tag | kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (an aggregate: the owner's later inclusive store wins)
          if (a.help_counter) (void)__hip_atomic_fetch_add(a.help_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        missing &= ~(1ull << q);
      }
      // from the nearest predecessor that knows its inclusive sum upwards
      const unsigned long long inclusive = __ballot(need && (g & kRecompactInclusive) != 0ull);
      const uint32_t from = inclusive ? 63u - (uint32_t)__builtin_clzll(inclusive) : 0u;
      exclusive += wave_sum(need && lane >= from ? (uint32_t)g & 0x7fffffffu : 0u);
      if (inclusive) break;
      pos = lo;
    }
    if (lane == 0u) {
      if (!skip_publish) __hip_atomic_store(&a.block_status[b], tag | kRecompactInclusive | (exclusive + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_exclusive = exclusive;
      if (b == last_block) *a.out_count = exclusive + total;
    }
  }
  __syncthreads();
  if (keep) {
    uint32_t* dst = a.out_cmds + (size_t)(s_exclusive + before + lanes_below(mask)) * kCmdWords;
#pragma unroll
    for (uint32_t f = 0; f < kCmdWords; ++f) dst[f] = w[f];
  }
  if (b == last_block && tid < a.n_zero) a.zero_words[tid] = 0u;  // the stage's counters for the next frame of this slot
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

// ---------------------------------------------------------------------------------------
// Round 5 — every frame size: the stage as EQUAL RANGES of the frame's triangle stream
// ---------------------------------------------------------------------------------------
// The reference cuts every instance's triangles into workgroups of 384 (generate_work.comp:56,74-75;
// cull_pipeline.rs:560-576 dispatches ceil(tris / wg) of them): every unit of work is the same size. Rounds 1-4 handed a
// WHOLE command (up to tens of thousands of triangles) to one wave or one workgroup from a ticket counter, and a launch
// ended with waves walking their last command alone (up to 19 % of it: profiles/r04_triangle_bound_experiments.txt).
// Here the frame's triangle stream is cut into equal RANGES, one per wave of a grid that is resident as a whole (a few
// per wave when MIP_TUNE_TRI_RANGES_PER_WAVE says so); no ticket (11 ns per same-address atomic), no tail. Command c owns
// the SLOTS
//     [slot0(c), slot0(c) + indexCount(c) / 3),   slot0(c) = (firstIndex(c) - first_index_base) / 3
// of that stream (firstIndex is already the running sum of indexCount over the emitted commands — the instance kernel's
// prefix — so the slots of different commands are disjoint and ascending; index counts that are no multiple of 3 leave
// unused slots). Range b = slots [b * S, (b + 1) * S), S = the stream's length over the number of ranges, in whole steps
// of 64. The part of a command inside a range is a SEGMENT, walked 64 triangles per step with the command's matrix in
// scalar registers, exactly as the wave-per-command kernel walks a whole command.
//   * where a range starts: `range_first_cmd`, written by mip_triangle_prepare_kernel (one thread per command);
//   * order: a command's survivors follow those of its earlier triangles. A segment whose command STARTS in the range
//     knows its position (0) and writes as it goes. The one segment of a range that CONTINUES a command from earlier
//     ranges — the first — needs that command's survivors there: it is TESTED at once, its 64-bit keep masks kept in LDS
//     (8 bytes per step), every range publishes the survivors of its LAST segment as one tagged granule when it is
//     through, and the continuing segment is WRITTEN one range later — after the wave's next range has been tested: by
//     then the ranges it looks back at (claimed before it, by waves that are running) have published, and the look-back
//     is one round trip instead of a wait for the slowest neighbour. The kept triangles' index triples are read again
//     from the mesh's own index range (shared by every instance of the mesh: the L2 has it), eight steps in flight.
//     A granule that is not there after the patient polls is not waited for: the wave counts that range's share itself
//     (same triangles, same test) and publishes it — no wait depends on another wave ever running.
// Slots per range, in whole steps of 64. A stream that gives every wave of the (resident) grid at most `ticket_slots`
// slots is cut into one range per wave, dealt statically: nothing to balance, no ticket. A longer stream is cut into
// ranges of `ticket_slots`: a wave's first range is its own, the others are pulled from a counter — waves do NOT advance
// at the same rate (the SIMD issues its oldest wave first: equal static shares ended 1.1 .. 3.1 ms apart on the 300 k
// frame, profiles/r05_wave_kernel_lifetimes.txt), and a ticket per 4 096 triangles is one per ~20 ns of the launch, above
// the ~11 ns a same-address atomic takes.
constexpr uint32_t kRangeMinSlots = 256;
constexpr uint32_t kRangeMaxSlots = 8192;   // = kRangeMaskSteps * 64: a continuing segment's masks always fit
__host__ __device__ __forceinline__ uint32_t range_slots(uint32_t total, uint32_t n_waves, uint32_t ticket_slots) {
  const uint32_t one = (uint32_t)((((unsigned long long)total + n_waves - 1u) / n_waves + 63u) / 64u) * 64u;
  if (one <= ticket_slots) return one < kRangeMinSlots ? kRangeMinSlots : one;
  return ticket_slots;
}

// (range b starts with the first command whose slots END behind b * S)
// The kernel in front of the stage, one thread per command: for the range decomposition the map of range -> first command; for
// the wave-per-command decomposition (frames above tri_block_max that are not the range decomposition's: choice_mode 3) the
// histogram of the size classes and, by the last workgroup to arrive, positions, tickets and batch sizes.
__global__ __launch_bounds__(256) void mip_triangle_prepare_kernel(const TriangleChunkArgs ca) {
  const TriangleArgs& a = ca.t;
  const uint32_t count = *a.count;
  if (count == 0u) return;
  if (a.choice_mode == 3u && !plan_tri_choice_is_ranges(a.max_lod_tris, stream_slots(a, a.first_index_base, count), count, a.choice_waves)) {
    triangle_sort_count_part(a, count);
    return;
  }
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= count) return;
  const uint32_t base = ca.first_index_base;
  const uint32_t S = range_slots(stream_slots(a, base, count), ca.n_waves, ca.ticket_slots);
  const uint32_t index_count = a.cmds[c * kCmdWords + 0], first_index = a.cmds[c * kCmdWords + 2];
  const uint32_t n_tris = index_count / 3u;
  const uint32_t end = (first_index - base) / 3u + n_tris;
  const uint32_t prev_end = c ? (a.cmds[(c - 1u) * kCmdWords + 2] - base) / 3u + a.cmds[(c - 1u) * kCmdWords + 0] / 3u : 0u;
  if ((unsigned long long)first_index + index_count > a.capacity) raise_error(a.error_flag, kErrIndexOverflow);
  if (n_tris == 0u) a.final_index_count[c] = 0u;  // (an indexCount of 1 or 2: no range ever visits it)
  uint32_t b = (prev_end + S - 1u) / S;
  uint32_t b_end = (end + S - 1u) / S;
  if (b_end > ca.ranges_cap) b_end = ca.ranges_cap;  // (only commands that do not fit the index buffer reach past it: reported above)
  for (; b < b_end; ++b) ca.range_first_cmd[b] = c;
}

#ifndef MIP_TRI_CHUNK_WAVES_PER_SIMD
#define MIP_TRI_CHUNK_WAVES_PER_SIMD 8
#endif
constexpr uint32_t kChunkPolls = 128;        // x (s_sleep 2 + one load): some tens of microseconds
constexpr uint32_t kRangeMaskSteps = kRangeMaxSlots / 64u;  // keep masks of a continuing segment: 1 KB of LDS per wave and buffer

// The same walk without the software pipeline and with the path chosen at run time: the rare paths (a range's share
// counted for a wave that has not published; the tail of a continuing segment beyond the mask buffer).
template <bool kWrite>
__device__ __forceinline__ uint32_t chunk_rewalk(const TriangleArgs& a, uint32_t base, const float (&pv)[16], uint32_t c, uint32_t b, uint32_t e, size_t dst_tri) {
  const uint32_t lane = threadIdx.x & 63u;
  const ChunkCmd m = chunk_load_cmd(a, base, c);
  float model[16];
  chunk_load_model(a, m.instance, model);
  const bool affine = model_is_affine(model, a.geometry_finite);
  const uint32_t* tri_indices = a.indices + (size_t)m.src_tri * 3;
  uint32_t survivors = 0;
  for (uint32_t t0 = b; t0 < e; t0 += 64u) {
    const uint32_t t = t0 + lane;
    const bool valid = t < e;
    uint32_t j0 = 0, j1 = 0, j2 = 0;
    if (valid) {
      const uint32_t* ip = tri_indices + (size_t)t * 3;
      j0 = ip[0]; j1 = ip[1]; j2 = ip[2];
    }
    const bool keep = valid && !triangle_culled(affine, model, pv, a.vertices, (long long)m.vertex_offset, j0, j1, j2);
    const unsigned long long mask = __ballot(keep);
    if constexpr (kWrite) {
      if (keep) {
        uint32_t* dst = a.out_indices + (dst_tri + survivors + lanes_below(mask)) * 3;
        dst[0] = j0; dst[1] = j1; dst[2] = j2;
      }
    }
    survivors += (uint32_t)__popcll(mask);
  }
  return survivors;
}

// A range's continuing segment between its test and its write.
struct ChunkPending {
  bool valid, ends, fits;
  uint32_t b, c, survivors, slot0, first_index, src_tri, t_begin, t_end;
};

// The write of a tested continuing segment: the kept triangles' index triples are read again through a descriptor over
// the segment's part of the mesh's index range (a lane without a survivor gets an offset beyond it: no load, no branch)
// and stored through one over the segment's place in the stream — twelve steps' loads in flight, no address arithmetic
// in vector registers (the step advances the scalar offset).
__device__ __forceinline__ void chunk_write_deferred(const TriangleArgs& a, uint32_t base, const float (&pv)[16], const ChunkPending& p,
                                                     const unsigned long long* masks, uint32_t prefix, uint32_t lane) {
  if (!p.fits) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // lane 0's masks are every lane's from here on (the wave's own LDS: no barrier)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t n_steps = (p.t_end - p.t_begin + 63u) / 64u;
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.indices) + ((size_t)p.src_tri + p.t_begin) * 3, 0,
                                                                       (int)((p.t_end - p.t_begin) * 12u), 0x00020000);
  const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(a.out_indices + ((size_t)p.first_index / 3u + prefix) * 3, 0,
                                                                       (int)(p.survivors * 12u), 0x00020000);
  const uint32_t lane_offset = lane * 12u;
  uint32_t at = 0;  // survivors written so far
  constexpr uint32_t kBatch = 12;
  for (uint32_t s0 = 0; s0 < n_steps; s0 += kBatch) {
    unsigned long long mk[kBatch];
    tri_u32x3 idx[kBatch];
#pragma unroll
    for (uint32_t q = 0; q < kBatch; ++q) {
      const unsigned long long raw = s0 + q < n_steps ? masks[s0 + q] : 0ull;  // the same for every lane: kept in scalar registers
      mk[q] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(raw >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)raw);
      const bool keep = ((mk[q] >> lane) & 1ull) != 0ull;
      idx[q] = __builtin_amdgcn_raw_buffer_load_b96(src, (int)(keep ? lane_offset : kTriDropOffset), (int)((s0 + q) * 768u), 0);
    }
#pragma unroll
    for (uint32_t q = 0; q < kBatch; ++q) {
      const bool keep = ((mk[q] >> lane) & 1ull) != 0ull;
      __builtin_amdgcn_raw_buffer_store_b96(idx[q], out, (int)(keep ? (at + lanes_below(mk[q])) * 12u : kTriDropOffset), 0, MIP_TRI_STORE_AUX);
      at += (uint32_t)__popcll(mk[q]);
    }
  }
}

// Σ survivors of command p.c in the ranges [b0, p.b) — every one of them ends with a segment of that command.
__device__ __forceinline__ uint32_t chunk_lookback(const TriangleChunkArgs& ca, uint32_t S, const float (&pv)[16], const ChunkPending& p, uint32_t lane) {
  const TriangleArgs& a = ca.t;
  const uint32_t b0 = p.slot0 / S;
  uint32_t prefix = 0;
  for (uint32_t hi = p.b; hi > b0;) {
    const uint32_t lo = hi - b0 > 64u ? hi - 64u : b0;
    const uint32_t j = lo + lane;
    const bool need = j < hi;
    bool ready = !need;
    uint32_t got = 0, polls = 0;
    for (;;) {
      if (!ready) {
        const unsigned long long g = status_load(&ca.range_status[j]);
        if ((uint32_t)(g >> 32) == ca.epoch) { ready = true; got = (uint32_t)g; }
      }
      if (__all(ready)) break;
      if (__builtin_expect(++polls > kChunkPolls, 0)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    unsigned long long missing = __ballot(!ready);
    while (__builtin_expect(missing != 0ull, 0)) {  // wave-uniform: count that range's share of the command ourselves
      const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_ctzll(missing));
      const uint32_t jj = lo + q;
      unsigned long long g = status_load(&ca.range_status[jj]);
      if ((uint32_t)(g >> 32) != ca.epoch) {
        const uint32_t b_slot = jj * S > p.slot0 ? jj * S : p.slot0;
        const uint32_t b = b_slot - p.slot0, e = (jj + 1u) * S - p.slot0;  // (the command goes on past range jj: e < n_tris)
        const uint32_t survivors = chunk_rewalk<false>(a, ca.first_index_base, pv, p.c, b, e, 0);
        g = ((unsigned long long)ca.epoch << 32) | survivors;
        if (lane == 0u) {
          __hip_atomic_store(&ca.range_status[jj], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          (void)__hip_atomic_fetch_add(a.help_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (lane == q) got = (uint32_t)g;
      missing &= ~(1ull << q);
    }
    prefix += wave_sum(got);
    hi = lo;
  }
  return prefix;
}

typedef unsigned long long RangeMasks[4][2][kRangeMaskSteps];  // per wave: two buffers of keep masks (LDS of the kernel that runs the body)

__device__ __forceinline__ void triangle_ranges_body(const TriangleChunkArgs& ca, uint32_t count, RangeMasks& s_masks) {
  const TriangleArgs& a = ca.t;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t base = ca.first_index_base;
  const uint32_t total = stream_slots(a, base, count);
  const uint32_t S = range_slots(total, ca.n_waves, ca.ticket_slots);
  uint32_t n_ranges = (total + S - 1u) / S;
  if (n_ranges > ca.ranges_cap) n_ranges = ca.ranges_cap;
  float pv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) pv[k] = a.pv[k];

  const uint32_t n_waves = gridDim.x * 4u;
#ifdef MIP_EXP_RANGE_TIMES  // experiment build (tools/r05_range_times.py): when every wave started and ended, behind the index buffer's capacity
  struct RangeTimes {
    uint32_t* at; uint32_t lane;
    __device__ RangeTimes(uint32_t* p, uint32_t l) : at(p), lane(l) { if (lane == 0u) at[0] = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> 0); }
    __device__ ~RangeTimes() { if (lane == 0u) at[1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> 0); }
  } range_times(a.out_indices + a.capacity + 2u * (blockIdx.x * 4u + wave), lane);
#endif
  // the previous range's continuing segment: tested and published, written after this range's test
  ChunkPending pend{};
  uint32_t buf = 0;
  // a wave's first range is its own; the others come from the counter (zeroed by the host), which starts behind the static ones
  for (uint32_t dealt = blockIdx.x * 4u + wave;;
       dealt = n_ranges <= n_waves ? n_ranges : n_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)atomicAdd(a.ticket, lane == 0u ? 1u : 0u))) {
    ChunkPending cur{};
    const bool walk = dealt < n_ranges;
    if (walk) {
      uint32_t b = dealt;
#ifdef MIP_DEBUG_STAMPS
      if (ca.debug_reverse) b = n_ranges - 1u - dealt;
      const bool skip_publish = ca.debug_skip_part && (b & 15u) == ca.debug_skip_part - 1u;  // fault injection
#else
      const bool skip_publish = false;
#endif
      const uint32_t slot_lo = b * S, slot_hi = slot_lo + S;
      uint32_t last_survivors = 0;
      uint32_t c = ca.range_first_cmd[b];
      ChunkCmd m{};
      if (c < count) m = chunk_load_cmd(a, base, c);
      while (c < count) {
        if (m.slot0 >= slot_hi) break;
        const uint32_t seg_lo = m.slot0 > slot_lo ? m.slot0 : slot_lo;
        const uint32_t seg_hi = m.slot0 + m.n_tris < slot_hi ? m.slot0 + m.n_tris : slot_hi;
        const bool ends_here = m.slot0 + m.n_tris <= slot_hi;
        ChunkCmd m_after{};
        if (ends_here && c + 1u < count) m_after = chunk_load_cmd(a, base, c + 1u);  // in flight while this segment is walked
        if (seg_lo < seg_hi) {
          const uint32_t t_begin = seg_lo - m.slot0, t_end = seg_hi - m.slot0;
          float model[16];
          chunk_load_model(a, m.instance, model);
          const bool affine = model_is_affine(model, a.geometry_finite);
          const bool fits = (unsigned long long)m.first_index + m.index_count <= a.capacity;  // (reported by the map kernel)
          const size_t dst_tri = (size_t)m.first_index / 3u;
          uint32_t survivors;
          if (t_begin > 0u) {  // continues a command that started in an earlier range (only a range's first segment can)
            survivors = affine ? chunk_walk<true, false>(a, m, model, pv, t_begin, t_end, fits, dst_tri, s_masks[wave][buf], lane)
                               : chunk_walk<false, false>(a, m, model, pv, t_begin, t_end, fits, dst_tri, s_masks[wave][buf], lane);
            cur.valid = true; cur.ends = ends_here; cur.fits = fits; cur.b = b; cur.c = c; cur.survivors = survivors;
            cur.slot0 = m.slot0; cur.first_index = m.first_index; cur.src_tri = m.src_tri; cur.t_begin = t_begin; cur.t_end = t_end;
          } else {             // the command starts here: its survivors' position is known
            survivors = affine ? chunk_walk<true, true>(a, m, model, pv, t_begin, t_end, fits, dst_tri, nullptr, lane)
                               : chunk_walk<false, true>(a, m, model, pv, t_begin, t_end, fits, dst_tri, nullptr, lane);
            if (ends_here && lane == 0u) a.final_index_count[c] = survivors * 3u;
          }
          last_survivors = survivors;
        }
        if (!ends_here) break;
        m = m_after;
        ++c;
      }
      // ---- publish the last segment's survivors (the next range needs them iff that segment's command goes on there) ----
      if (lane == 0u && !skip_publish)
        __hip_atomic_store(&ca.range_status[b], ((unsigned long long)ca.epoch << 32) | last_survivors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the PREVIOUS range's continuing segment: behind the survivors of its command in the earlier ranges ----
    if (pend.valid) {
      const uint32_t prefix = chunk_lookback(ca, S, pv, pend, lane);
      chunk_write_deferred(a, base, pv, pend, s_masks[wave][buf ^ 1u], prefix, lane);
      if (pend.ends && lane == 0u) a.final_index_count[pend.c] = (prefix + pend.survivors) * 3u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // those masks have been read before a later range writes its own there
      __builtin_amdgcn_wave_barrier();
    }
    if (!walk) break;
    pend = cur;
    buf ^= 1u;
  }
}

__global__ __launch_bounds__(256, MIP_TRI_CHUNK_WAVES_PER_SIMD) void mip_triangle_cull_ranges_kernel(const TriangleChunkArgs ca) {
  __shared__ RangeMasks s_masks;
  const uint32_t count = *ca.t.count;
  if (count == 0u || tri_not_this_grid(ca.t, count)) return;
  triangle_ranges_body(ca, count, s_masks);
}

// Frames above tri_block_max instances: ONE grid, and every workgroup takes the decomposition the frame is for — equal ranges of
// the triangle stream, or one wave per command over the size-sorted list (plan_tri_choice_is_ranges, from the slot's own command
// list). (First build of the pairing: two grids of 2 048 workgroups, one of which returned at once — 4.75 us of every frame.)
__global__ __launch_bounds__(256, MIP_TRI_CHUNK_WAVES_PER_SIMD) void mip_triangle_stage_kernel(const TriangleChunkArgs ca) {
  __shared__ RangeMasks s_masks;
  const uint32_t count = *ca.t.count;
  if (count == 0u) return;
  if (plan_tri_choice_is_ranges(ca.t.max_lod_tris, stream_slots(ca.t, ca.t.first_index_base, count), count, ca.t.choice_waves)) triangle_ranges_body(ca, count, s_masks);
  else triangle_waves_body(ca.t, count);
}

}  // namespace mip
