// instance_pipeline_kernels.hpp — gfx950 (CDNA4) kernels of the instance pipeline.
//
// One fused, single-pass kernel per frame:
//
//   tile = 256 instances = one 256-thread workgroup (4 wave64), one instance per lane
//   loads   : pos (12 B) + quat (16 B) + scale (4 B) + mesh id (4 B)            = 36 B
//   compute : M = T·R·S, 8-corner world AABB, 6-plane test, LOD pick             (VALU, no FMA)
//   stores  : mat4 through an LDS transpose so every store instruction writes
//             1 KiB contiguous (64 B), 1 visibility bit, and — after a one-hop look-up of
//             the tile's exclusive prefix over per-tile granules and per-group atomic
//             accumulators of {count, Σ index_len} — the tile's surviving
//             VkDrawIndexedIndirectCommands, coalesced, in draw_index order.
//   also here: the shard-merge kernel (multi-GPU), the per-triangle cull kernel (row f-1)
//             and the command re-compaction that follows it.
//
// Reference semantics (paths in farnoy/renderer):
//   src/ecs.rs:52-64 model_matrix_calculation, :138-181 aabb_calculation,
//   src/renderer/systems/cull_pipeline.rs:99-120 coarse_culling, :534-577 cull_pass,
//   src/renderer/helpers.rs:3-11 pick_lod, src/shaders/generate_work.comp:61-67,
//   src/shaders/compact_draw_stream.comp:34-63.
//
// Arithmetic contract: IEEE binary32, every multiply and add rounded separately, in
// the operation order nalgebra 0.29 / ncollide3d 0.32 use. This TU
// must be compiled with -ffp-contract=off and without fast-math.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mip {

// Register budget: minimum waves per SIMD the kernel is compiled for (k workgroups of 256
// threads per CU <=> k waves per SIMD). Overridable to build tuning variants.
#ifndef MIP_MIN_WAVES_PER_SIMD
#define MIP_MIN_WAVES_PER_SIMD 6
#endif

#ifndef MIP_TILE
#define MIP_TILE 256
#endif
constexpr uint32_t kTile = MIP_TILE;      // instances per tile == threads per workgroup
constexpr uint32_t kWaves = kTile / 64;   // wave64
constexpr uint32_t kCmdWords = 5;         // VkDrawIndexedIndirectCommand = 5 dwords
constexpr uint32_t kCmdLdsWords = 6;      // in LDS each command also carries its source index offset

// Device-side mesh entry: what the kernel needs of MipMesh, 32 B, two 16-B gathers.
// len0 = index_len[0]; len1 = index_len[1] if n_lods > 1 else index_len[0]
// (pick_lod falls back to LOD 0 when there is only one, helpers.rs:6).
struct alignas(16) MeshEntry {
  float min_x, min_y, min_z;
  uint32_t len0;
  float max_x, max_y, max_z;
  uint32_t len1;
};

// Per-mesh draw data, one 16-B gather for the lanes that emit a command.
struct alignas(16) MeshDraw {
  int32_t vertex_offset;  // ConsolidatedMeshBuffers.vertex_offsets[mesh]
  uint32_t src_offset0;   // index_offsets[LOD 0] in the consolidated index buffer
  uint32_t src_offset1;   // index_offsets[LOD 1] (= LOD 0's when there is only one)
  uint32_t pad;
};

struct KernelArgs {
  const float* pos;             // n*3
  const float4* rot;            // n  [i,j,k,w]
  const float* scale;           // n
  const uint32_t* mesh_id;      // n
  const MeshEntry* meshes;      // m
  const MeshDraw* mesh_draw;    // m
  float4* model;                // n*4 or null
  uint32_t* bitmap;             // ceil(n/32) or null
  uint32_t* cmds;               // n*5 or null
  uint32_t* draw_count;         // with cmds
  uint32_t* index_total;        // optional
  uint32_t* src_index_offset;   // optional: per emitted command, where its LOD's indices start (row f-1)
  float* world_aabb;            // n*6 or null
  uint4* tlas_instances;        // n x VkAccelerationStructureInstanceKHR (64 B) or null (row f-4)
  const unsigned long long* blas_address;  // m, BLAS device address per mesh (with tlas_instances)
  const float* box_override;    // n*6 or null: per-instance mesh-space box (min xyz, max xyz) that replaces the mesh table's (skinned instances)
  unsigned long long* status0;  // level 0: one tagged granule per tile
  unsigned long long* acc1;     // level 1: [2 parities][groups_cap] 64-bit accumulators
  unsigned long long* start1;   // level 1: exclusive prefix at the start of each group, 2 tagged granules
  uint32_t groups_cap;
  uint32_t group_shift;         // log2(tiles per group), <= 6
  uint32_t* error_flag;         // host-mapped
  uint32_t n;
  uint32_t n_tiles;
  uint32_t epoch;               // 1 .. 2^31-1, unique per launch
  uint32_t bitmap_words;
  uint32_t first_instance_base;
  uint32_t first_index_base;
  float planes[24];
  float cam[3];
#ifdef MIP_DEBUG_STAMPS
  unsigned long long* stamps;  // diagnostic build only: 8 realtime stamps per tile
  uint32_t debug_skip_publish_tile;  // diagnostic build only: tile index + 1 that never publishes (0 = off)
#endif
};

#ifdef MIP_DEBUG_STAMPS
#define MIP_STAMP(k)                                                                       \
  do {                                                                                     \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define MIP_STAMP(k) do { } while (0)
#endif

// Largest q with sqrt_rn(q) <= 10: pick_lod tests `magnitude() > 10.0`
// (helpers.rs:4-6) and magnitude = sqrt(norm_squared) correctly rounded, so
// sqrt_rn(q) > 10  <=>  q > nextafter(100) (tests/test_oracle.py checks this
// equivalence exhaustively around 100).
constexpr float kLodDistSqThreshold = 100.00000762939453125f;  // 100 + 2^-17

constexpr uint32_t kErrTimeout = 1u;

// ---------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------

// Inclusive prefix sum over the 64 lanes with DPP row shifts + row broadcasts (gfx9).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------------------------------
// per-instance arithmetic
// ---------------------------------------------------------------------------------------

struct Instance {
  float m[12];      // rows 0..2 of the model matrix, column-major: m[c*3 + r]
  uint32_t row3;    // bit c set <=> M[3][c] is NaN (otherwise it is 0,0,0,1)
  float mins[3], maxs[3];
};

// nalgebra UnitQuaternion::to_rotation_matrix; r[row][col].
__device__ __forceinline__ void quat_to_rotation(float i, float j, float k, float w, float (&r)[3][3]) {
  const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
  const float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
  r[0][0] = ww + ii - jj - kk; r[0][1] = ij - wk;           r[0][2] = wj + ik;
  r[1][0] = wk + ij;           r[1][1] = ww - ii + jj - kk; r[1][2] = jk - wi;
  r[2][0] = ik - wj;           r[2][1] = wi + jk;           r[2][2] = ww - ii - jj + kk;
}

__device__ __forceinline__ void fold_corner(const float (&v)[3], float (&lo)[3], float (&hi)[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = fminf(lo[a], v[a]);  // f32::min: a NaN operand is ignored
    hi[a] = fmaxf(hi[a], v[a]);
  }
}

__device__ __forceinline__ void finish_aabb(const float (&lo)[3], const float (&hi)[3], Instance& o) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float centre = (hi[a] + lo[a]) / 2.0f;
    const float half = (hi[a] - lo[a]) / 2.0f;
    o.mins[a] = centre - half;  // AABB::from_half_extents
    o.maxs[a] = centre + half;
  }
}

// Fast path, exact whenever the 9 rotation entries, the position and the scale are all
// finite (and the mesh box is, which mip_set_mesh_table enforces): then every product
// with a 0 or 1 entry of T, S and the homogeneous row/column is exact, (T·R)·S collapses
// to M[r][c] = fl(R[r][c]·s), M[:,3] = (p,1), M[3,:] = (0,0,0,1), and w = 1 for every
// corner, so `/ w` is the identity.
__device__ __forceinline__ void model_fast(const float (&r)[3][3], float px, float py, float pz, float s, Instance& o) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) o.m[c * 3 + rr] = r[rr][c] * s;
  o.m[9] = px; o.m[10] = py; o.m[11] = pz;
  o.row3 = 0;
}

__device__ __forceinline__ void instance_fast(const float (&r)[3][3], float px, float py, float pz,
                                              float s, const MeshEntry& mb, Instance& o) {
  model_fast(r, px, py, pz, s, o);
  float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
  float hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
  const float bx[2] = {mb.min_x, mb.max_x}, by[2] = {mb.min_y, mb.max_y}, bz[2] = {mb.min_z, mb.max_z};
  // corner order of src/ecs.rs:149-160: x toggles fastest, then z, then y
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float x = bx[c & 1], z = bz[(c >> 1) & 1], y = by[(c >> 2) & 1];
    float v[3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)  // gemv as column axpys: ((m0 x + m1 y) + m2 z) + m3·1
      v[rr] = o.m[0 * 3 + rr] * x + o.m[1 * 3 + rr] * y + o.m[2 * 3 + rr] * z + o.m[9 + rr];
    fold_corner(v, lo, hi);
  }
  finish_aabb(lo, hi, o);
}

// nalgebra gemv (alpha = 1, beta = 0): column axpys left to right.
__device__ __forceinline__ void gemv4(const float (&a)[16], const float (&x)[4], float (&y)[4]) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) y[rr] = a[rr] * x[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) y[rr] = a[k * 4 + rr] * x[k] + y[rr];
}

__device__ __forceinline__ void gemm4(const float (&a)[16], const float (&b)[16], float (&out)[16]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float x[4] = {b[c * 4 + 0], b[c * 4 + 1], b[c * 4 + 2], b[c * 4 + 3]};
    float y[4];
    gemv4(a, x, y);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) out[c * 4 + rr] = y[rr];
  }
}

// General path: the reference chain performed literally — translation(p) *
// rot.to_homogeneous() * scaling(s) as two full 4x4 products, full mat4*vec4 per corner
// and the divide by w — so non-finite inputs poison exactly the entries they poison in
// the reference. Taken by a whole wave when any of its lanes fails the finite test.
__device__ __forceinline__ void model_general(const float (&r)[3][3], float px, float py, float pz, float s, Instance& o,
                                              float (&m)[16]) {
  float t[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, px, py, pz, 1};
  float rh[16] = {r[0][0], r[1][0], r[2][0], 0, r[0][1], r[1][1], r[2][1], 0,
                  r[0][2], r[1][2], r[2][2], 0, 0, 0, 0, 1};
  float sc[16] = {s, 0, 0, 0, 0, s, 0, 0, 0, 0, s, 0, 0, 0, 0, 1};
  float tr[16];
  gemm4(t, rh, tr);
  gemm4(tr, sc, m);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) o.m[c * 3 + rr] = m[c * 4 + rr];
  o.row3 = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) o.row3 |= (m[c * 4 + 3] != m[c * 4 + 3]) ? (1u << c) : 0u;
}

__device__ __forceinline__ void instance_general(const float (&r)[3][3], float px, float py, float pz,
                                              float s, const MeshEntry& mb, Instance& o) {
  float m[16];
  model_general(r, px, py, pz, s, o, m);
  float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
  float hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
#pragma unroll
  for (int c = 0; c < 8; ++c) {  // unrolled so the corner selects stay in registers
    const float vh[4] = {(c & 1) ? mb.max_x : mb.min_x, (c & 4) ? mb.max_y : mb.min_y,
                         (c & 2) ? mb.max_z : mb.min_z, 1.0f};
    float tv[4];
    gemv4(m, vh, tv);
    const float v[3] = {tv[0] / tv[3], tv[1] / tv[3], tv[2] / tv[3]};
    fold_corner(v, lo, hi);
  }
  finish_aabb(lo, hi, o);
}

// src/renderer/systems/cull_pipeline.rs:108-119. Planes are wave-uniform (SGPRs).
__device__ __forceinline__ bool coarse_culled(const Instance& o, const float (&planes)[24]) {
  float h[3], c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    h[a] = (o.maxs[a] - o.mins[a]) * 0.5f;  // AABB::half_extents
    c[a] = (o.mins[a] + o.maxs[a]) * 0.5f;  // AABB::center
  }
  bool outside = false;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const float nx = planes[p * 4 + 0], ny = planes[p * 4 + 1], nz = planes[p * 4 + 2], d = planes[p * 4 + 3];
    const float e = h[0] * fabsf(nx) + h[1] * fabsf(ny) + h[2] * fabsf(nz);  // 3-wide dot: (a+b)+c
    float a0 = nx * c[0];
    float a1 = ny * c[1];
    const float a2 = nz * c[2];
    const float a3 = d;  // d * 1
    a0 += a2;            // 4-wide dot: (a0+a2) + (a1+a3)
    a1 += a3;
    const float sd = a0 + a1;
    outside = outside || (sd - e > 0.0f);  // the reference's early break changes nothing
  }
  return outside;
}

// ---------------------------------------------------------------------------------------
// cross-tile prefix: one hop, no chains
// ---------------------------------------------------------------------------------------
// A tile's exclusive prefix (count, Σ index_len) over all earlier tiles is assembled from
// words that every tile publishes as soon as it knows its own aggregate — it never depends
// on another tile having finished its own look-up, so the wait is one memory round trip
// after the slowest predecessor has published (measured: a hop costs ~1 µs on an idle
// chip and ~3 µs behind streaming traffic, so chains of hops are what must be avoided).
//
//   level 0  status0[tile]   ONE 8-byte granule {Σ index_len : 32 | tag : 23 | count : 9},
//                            written by one relaxed agent-scope atomic store
//                            (global_store_dwordx2 sc1); tag = launch epoch (never 0), so
//                            the array is never cleared between launches.
//   level 1  acc1[parity][g] one 64-bit accumulator per group of 2^group_shift consecutive
//                            tiles, 256 B apart (packed words put every tile's reads and
//                            the atomics on one or two memory channels: measured 5x
//                            slower); every tile of the group adds
//                            {Σ index_len : 32 | arrivals : 12 | count : 20} with one
//                            no-return agent-scope atomic add (executes at the memory side).
//                            A group is complete when arrivals == tiles per group. The
//                            buffer of the other parity is zeroed for the next launch by
//                            the first tile of each group; the host clears everything
//                            whenever the instance count changes or the tag wraps.
//
//            start1[g]       {epoch : 32 | count : 32} {epoch : 32 | Σ index_len : 32}: the exclusive
//                            prefix at the start of group g, published by the group's first
//                            tile when it has resolved its own prefix (a by-product).
//
//   prefix(tile) = start1[g_lo] + Σ acc1[g_lo .. g-1] + Σ status0[first tile of g .. tile-1],
//   g_lo = max(0, g - 64)
//
// No payload is handed off behind these words (every tile writes its own commands), so no
// release/acquire fence is involved; readers use relaxed agent-scope atomic loads (sc1).

constexpr uint32_t kAccCountBits = 20;  // the 12 bits above it count the tiles that have added
// Accumulators live 256 B apart: every tile reads every earlier group's word, and packed
// words would put all of that traffic (and the atomics) on one or two memory channels.
constexpr uint32_t kAccStrideWords = 32;
constexpr uint32_t kTileCountBits = kTile <= 256 ? 9 : (kTile <= 512 ? 10 : 11);
constexpr uint32_t kTagBits = 32 - kTileCountBits;
constexpr uint32_t kMaxEpoch = (1u << kTagBits) - 1u;
static_assert(kTile < (1u << kTileCountBits), "tile count must fit its field");

__device__ __forceinline__ unsigned long long status_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ONE lane of the tile once its aggregate is known.
__device__ __forceinline__ void publish_aggregate(const KernelArgs& a, uint32_t tile, uint32_t count, uint32_t sum) {
  const unsigned long long granule = ((unsigned long long)sum << 32) | ((unsigned long long)a.epoch << kTileCountBits) | count;
  __hip_atomic_store(&a.status0[tile], granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t group = tile >> a.group_shift;
  const uint32_t parity = a.epoch & 1u;
  const unsigned long long add = ((unsigned long long)sum << 32) | (1ull << kAccCountBits) | count;
  (void)__hip_atomic_fetch_add(&a.acc1[((size_t)parity * a.groups_cap + group) * kAccStrideWords], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((tile & ((1u << a.group_shift) - 1u)) == 0u)  // first tile of the group: reset the next launch's word
    __hip_atomic_store(&a.acc1[((size_t)(parity ^ 1u) * a.groups_cap + group) * kAccStrideWords], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned long long kSpinTimeoutTicks = 50000000ull;  // 0.5 s of the 100 MHz realtime counter
constexpr uint32_t kLevel1Window = 64;  // most recent groups whose accumulators a tile sums itself

// Run by one whole wave after publish_aggregate(tile). Returns the exclusive prefix of `tile`:
//   prefix = start1[g_lo]  +  Σ acc1[g_lo .. g-1]  +  Σ status0[first tile of g .. tile-1]
// with g_lo = max(0, g - 64). start1[g] (the exclusive prefix at the start of group g) is
// published for free by the first tile of group g once it has resolved its own prefix; the
// entry read here is 64 groups = thousands of tiles back, i.e. long resolved, so the look-up
// stays ONE round of <= 63 + 64 + 1 words for any N (without it every tile would read every
// earlier group: quadratic, measured +100 us at 10 M instances).
__device__ __forceinline__ void resolve_prefix(const KernelArgs& a, uint32_t tile, uint32_t lane,
                                               uint32_t& base_count, uint32_t& base_sum) {
  const uint32_t group = tile >> a.group_shift;
  const uint32_t group_first = group << a.group_shift;
  const uint32_t r = tile - group_first;  // earlier tiles of the own group (< 64)
  const uint32_t per_group = 1u << a.group_shift;
  const uint32_t g_lo = group > kLevel1Window ? group - kLevel1Window : 0u;
  const unsigned long long* acc = &a.acc1[(size_t)(a.epoch & 1u) * a.groups_cap * kAccStrideWords];
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  bool ok = true;

  // level 0: lane l < r reads the aggregate of tile group_first + l
  const bool v0 = lane < r;
  const unsigned long long* e0 = &a.status0[group_first + (v0 ? lane : 0u)];
  // level 1: lane l reads the accumulator of group g_lo + l
  const bool v1 = g_lo + lane < group;
  const unsigned long long* e1 = &acc[(size_t)(g_lo + (v1 ? lane : 0u)) * kAccStrideWords];
  // far prefix: lane 0 reads the two granules of start1[g_lo]
  const bool v2 = g_lo > 0u && lane == 0u;
  const unsigned long long* e2 = &a.start1[2 * (size_t)g_lo];

  bool ready0 = !v0, ready1 = !v1, ready2 = !v2;
  uint32_t c = 0, s = 0;
  for (;;) {
    if (!ready0) {
      const unsigned long long g = status_load(e0);
      if ((((uint32_t)g >> kTileCountBits) & kMaxEpoch) == a.epoch) {
        ready0 = true;
        c += (uint32_t)g & ((1u << kTileCountBits) - 1u);
        s += (uint32_t)(g >> 32);
      }
    }
    if (!ready1) {
      const unsigned long long w = status_load(e1);
      if (((uint32_t)w >> kAccCountBits) == per_group) {  // every tile of that group has added
        ready1 = true;
        c += (uint32_t)w & ((1u << kAccCountBits) - 1u);
        s += (uint32_t)(w >> 32);
      }
    }
    if (!ready2) {
      const unsigned long long pc = status_load(e2), ps = status_load(e2 + 1);
      if ((uint32_t)(pc >> 32) == a.epoch && (uint32_t)(ps >> 32) == a.epoch) {
        ready2 = true;
        c += (uint32_t)pc;
        s += (uint32_t)ps;
      }
    }
    const bool all = ready0 && ready1 && ready2;
#ifdef MIP_DEBUG_STAMPS
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] += 1;
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] += (unsigned long long)__popcll(__ballot(!all));
#endif
    if (__all(all)) break;
    if (__builtin_amdgcn_s_memrealtime() - t_start > kSpinTimeoutTicks) {  // scalar: wave-uniform
      ok = false;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (ok) {
    base_count = wave_sum(c);
    base_sum = wave_sum(s);
    if (r == 0u && group > 0u && lane == 0u) {  // first tile of a group: publish the group's start
      unsigned long long* p = &a.start1[2 * (size_t)group];
      __hip_atomic_store(p, ((unsigned long long)a.epoch << 32) | base_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 1, ((unsigned long long)a.epoch << 32) | base_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    if (lane == 0) __hip_atomic_store(a.error_flag, kErrTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    base_count = 0;
    base_sum = 0;
  }
}

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(kTile, MIP_MIN_WAVES_PER_SIMD) void mip_instance_pipeline_kernel(const KernelArgs a) {
  __shared__ __attribute__((aligned(16))) float s_mat[kTile * 12];     // rows 0..2 of every matrix
  __shared__ uint32_t s_row3[kTile];                                     // NaN bits of row 3
  // The tile's commands (5 KB) reuse the staging area of waves 1-3 (9 KB) once those waves have
  // stored their matrices: 13.6 KB of LDS per workgroup instead of 18.5 KB, so that more
  // workgroups whose wave 0 is still waiting for its prefix fit beside the running ones.
  uint32_t* const s_cmd = reinterpret_cast<uint32_t*>(&s_mat[64 * 12]);
  static_assert((kTile - 64) * 12 >= kTile * kCmdLdsWords, "commands must fit the staging area of waves 1-3");
  __shared__ uint32_t s_wave_count[kWaves], s_wave_sum[kWaves];

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t tile = blockIdx.x;
  const uint32_t tile_first = tile * kTile;
  const uint32_t i = tile_first + tid;
  const bool active = i < a.n;
  const uint32_t il = active ? i : a.n - 1u;  // keep the loads of idle lanes in bounds
  MIP_STAMP(0);

  // ---- loads: 36 B per instance ----
  const float px = a.pos[3 * (size_t)il + 0], py = a.pos[3 * (size_t)il + 1], pz = a.pos[3 * (size_t)il + 2];
  const float4 q = a.rot[il];
  const float sc = a.scale[il];
  const uint32_t mesh = a.mesh_id[il];
  const float4 mb0 = *reinterpret_cast<const float4*>(&a.meshes[mesh].min_x);
  const float4 mb1 = *reinterpret_cast<const float4*>(&a.meshes[mesh].max_x);
  MeshEntry mb;
  mb.min_x = mb0.x; mb.min_y = mb0.y; mb.min_z = mb0.z; mb.len0 = __float_as_uint(mb0.w);
  mb.max_x = mb1.x; mb.max_y = mb1.y; mb.max_z = mb1.z; mb.len1 = __float_as_uint(mb1.w);

  // ---- model matrix + world AABB ----
  float r[3][3];
  quat_to_rotation(q.x, q.y, q.z, q.w, r);
  // Finite test for the fast path: a sum of magnitudes is NaN/inf as soon as one term is
  // (or the sum overflows — then the general path, which is exact for everything, runs).
  float mag = fabsf(px) + fabsf(py) + fabsf(pz) + fabsf(sc);
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
#pragma unroll
    for (int c = 0; c < 3; ++c) mag += fabsf(r[rr][c]);
  if (a.box_override) {  // skinned instances: the posed mesh-space box computed by mip_skinned_bounds_kernel
    const float2* b2 = reinterpret_cast<const float2*>(a.box_override + (size_t)il * 6);
    const float2 b01 = b2[0], b23 = b2[1], b45 = b2[2];
    mb.min_x = b01.x; mb.min_y = b01.y; mb.min_z = b23.x;
    mb.max_x = b23.y; mb.max_y = b45.x; mb.max_z = b45.y;
    // unlike a mesh-table box it may be non-finite: then the literal path is the exact one
    mag += fabsf(b01.x) + fabsf(b01.y) + fabsf(b23.x) + fabsf(b23.y) + fabsf(b45.x) + fabsf(b45.y);
  }
  const bool all_finite = mag < 3.0e38f;
  Instance inst;
  if (__builtin_expect(__any(!all_finite), 0)) {
    instance_general(r, px, py, pz, sc, mb, inst);
  } else {
    instance_fast(r, px, py, pz, sc, mb, inst);
  }

  MIP_STAMP(1);
  // ---- frustum test, LOD, command length ----
  const bool culled = coarse_culled(inst, a.planes);
  const bool visible = active && !culled;
  const float dx = a.cam[0] - px, dy = a.cam[1] - py, dz = a.cam[2] - pz;
  const float dist_sq = dx * dx + dy * dy + dz * dz;
  const uint32_t len = (dist_sq > kLodDistSqThreshold) ? mb.len1 : mb.len0;  // len1/offset1 already fall back to LOD 0
  const bool keep = visible && len > 0u;  // compact_draw_stream.comp:41 `indexCount > 0`
  const uint32_t len_vis = visible ? len : 0u;

  // ---- wave-level compaction offsets ----
  const unsigned long long keep_mask = __ballot(keep);
  const unsigned long long vis_mask = __ballot(visible);
  const uint32_t rank_in_wave = lanes_below(keep_mask);
  const uint32_t incl_sum = wave_inclusive_scan(len_vis);
  if (lane == 63u) {
    s_wave_count[wave] = (uint32_t)__popcll(keep_mask);
    s_wave_sum[wave] = incl_sum;
  }

  // ---- stage the matrix rows for the transposed store ----
  if (a.model || a.tlas_instances) {
    float4* dst = reinterpret_cast<float4*>(&s_mat[tid * 12]);
    dst[0] = make_float4(inst.m[0], inst.m[1], inst.m[2], inst.m[3]);
    dst[1] = make_float4(inst.m[4], inst.m[5], inst.m[6], inst.m[7]);
    dst[2] = make_float4(inst.m[8], inst.m[9], inst.m[10], inst.m[11]);
    s_row3[tid] = inst.row3 | (mesh << 4);  // NaN bits of row 3 + the mesh id (for the TLAS rows)
  }
  __syncthreads();

  uint32_t wave_off_count = 0, wave_off_sum = 0, tile_count = 0, tile_sum = 0;
#pragma unroll
  for (uint32_t w = 0; w < kWaves; ++w) {
    const uint32_t wc = s_wave_count[w], ws = s_wave_sum[w];
    if (w < wave) { wave_off_count += wc; wave_off_sum += ws; }
    tile_count += wc;
    tile_sum += ws;
  }

  const bool want_cmds = a.cmds != nullptr;
#ifdef MIP_DEBUG_STAMPS
  // fault injection (diagnostic build only): one tile never publishes, so every later tile's
  // bounded wait must expire and the launch must end with MIP_ERR_TIMEOUT instead of hanging
  const bool skip_publish = a.debug_skip_publish_tile == tile + 1u;
#else
  const bool skip_publish = false;
#endif
  if (want_cmds && tid == 0 && !skip_publish) publish_aggregate(a, tile, tile_count, tile_sum);
  MIP_STAMP(2);

  // Bulk stores of this wave: matrices, visibility words, optional AABBs. Wave 0 issues
  // them only AFTER it has resolved the tile prefix: loads return in order with stores
  // (vmcnt counts both), so a poll behind 4 KiB of stores would wait for their acks.
  auto bulk_stores = [&]() {
    // ---- model matrices: 4 store instructions per wave, each 1 KiB contiguous ----
    if (a.model) {
      const uint32_t wave_first = wave * 64u;
      const float* src = &s_mat[wave_first * 12];
      float4* out = a.model + ((size_t)tile_first + wave_first) * 4;
      const uint32_t col = lane & 3u;
  #pragma unroll
      for (uint32_t s4 = 0; s4 < 4; ++s4) {
        const uint32_t local = 16u * s4 + (lane >> 2);  // matrix within the wave
        const uint32_t flat = 192u * s4 + 3u * lane;    // = local*12 + col*3
        const uint32_t bits = s_row3[wave_first + local] & 15u;
        float w = (col == 3u) ? 1.0f : 0.0f;
        if ((bits >> col) & 1u) w = __uint_as_float(0x7fc00000u);
        if (tile_first + wave_first + local < a.n)
          out[64u * s4 + lane] = make_float4(src[flat], src[flat + 1], src[flat + 2], w);
      }
    }

    // ---- optional TLAS instance rows (acceleration_strucures.rs:419-451), same transposed store ----
    // VkAccelerationStructureInstanceKHR = { 3x4 row-major transform = rows 0..2 of M,
    //   instanceCustomIndex:24 = draw_index | mask:8 = 0xFF, sbtOffset:24 = 0 | flags:8 =
    //   TRIANGLE_FACING_CULL_DISABLE, BLAS device address }, for EVERY instance (visible or not).
    if (a.tlas_instances) {
      const uint32_t wave_first = wave * 64u;
      const float* src = &s_mat[wave_first * 12];
      uint4* out = a.tlas_instances + ((size_t)tile_first + wave_first) * 4;
      const uint32_t q = lane & 3u;
  #pragma unroll
      for (uint32_t s4 = 0; s4 < 4; ++s4) {
        const uint32_t local = 16u * s4 + (lane >> 2);
        const uint32_t draw = tile_first + wave_first + local;
        uint4 v;
        if (q < 3u) {  // row q: one element of each staged column
          const float* col = src + local * 12u + q;
          v = make_uint4(__float_as_uint(col[0]), __float_as_uint(col[3]), __float_as_uint(col[6]), __float_as_uint(col[9]));
        } else {
          const uint32_t mesh_of = s_row3[wave_first + local] >> 4;
          const unsigned long long blas = (draw < a.n && a.blas_address) ? a.blas_address[mesh_of] : 0ull;
          v = make_uint4(((a.first_instance_base + draw) & 0xffffffu) | 0xff000000u, 0x01000000u,
                         (uint32_t)blas, (uint32_t)(blas >> 32));
        }
        if (draw < a.n) out[64u * s4 + lane] = v;
      }
    }

    // ---- visibility bitmap: one 64-bit ballot per wave, written as two words ----
    if (a.bitmap && lane < 2u) {
      const uint32_t word = (tile_first >> 5) + wave * 2u + lane;
      if (word < a.bitmap_words) a.bitmap[word] = (uint32_t)(vis_mask >> (32u * lane));
    }

    // ---- optional world AABB (mins, maxs) as the ECS component holds it ----
    if (a.world_aabb && active) {
      float2* o2 = reinterpret_cast<float2*>(a.world_aabb + (size_t)i * 6);
      o2[0] = make_float2(inst.mins[0], inst.mins[1]);
      o2[1] = make_float2(inst.mins[2], inst.maxs[0]);
      o2[2] = make_float2(inst.maxs[1], inst.maxs[2]);
    }

  };

  if (!want_cmds) {
    bulk_stores();
    return;
  }

  if (wave != 0) bulk_stores();
  __syncthreads();  // waves 1-3 have read their staged matrices: their area is free for the commands

  // ---- tile-local command assembly in LDS (firstIndex still relative to the tile) ----
  if (keep) {
    const bool far_lod = dist_sq > kLodDistSqThreshold;
    const uint4 md = *reinterpret_cast<const uint4*>(&a.mesh_draw[mesh]);
    uint32_t* c = &s_cmd[(wave_off_count + rank_in_wave) * kCmdLdsWords];
    c[0] = len;                                               // indexCount
    c[1] = 1u;                                                // instanceCount, generate_work.comp:63
    c[2] = wave_off_sum + (incl_sum - len_vis);               // firstIndex (tile-relative)
    c[3] = md.x;                                              // vertexOffset, :66
    c[4] = a.first_instance_base + i;                         // firstInstance = draw_index, :64
    c[5] = far_lod ? md.z : md.y;                             // push constant indexOffset, cull_pipeline.rs:552
  }
  __syncthreads();  // s_cmd complete
  MIP_STAMP(3);

  // Waves 1-3 are finished: they exit and free their registers and wave slots for the next
  // workgroup while wave 0 alone waits for the tile's prefix and copies the commands out.
  if (wave != 0) return;

  // ---- exclusive prefix over the earlier tiles, before any bulk store of this wave ----
  uint32_t base_count = 0, base_sum = 0;
  if (tile > 0) resolve_prefix(a, tile, lane, base_count, base_sum);
  if (lane == 0 && tile == a.n_tiles - 1u) {
    *a.draw_count = base_count + tile_count;
    if (a.index_total) *a.index_total = base_sum + tile_sum;
  }
  MIP_STAMP(4);

  // ---- coalesced copy-out of the tile's commands ----
  const uint32_t first_index_add = base_sum + a.first_index_base;
  uint32_t* out = a.cmds + (size_t)base_count * kCmdWords;
  const uint32_t words = tile_count * kCmdWords;
  for (uint32_t j = lane; j < words; j += 64u) {
    const uint32_t k = j / kCmdWords, f = j - k * kCmdWords;
    uint32_t v = s_cmd[k * kCmdLdsWords + f];
    if (f == 2u) v += first_index_add;
    out[j] = v;
  }
  if (a.src_index_offset)
    for (uint32_t k = lane; k < tile_count; k += 64u) a.src_index_offset[base_count + k] = s_cmd[k * kCmdLdsWords + 5u];
  bulk_stores();
  MIP_STAMP(5);
}

// ---------------------------------------------------------------------------------------
// shard merge (multi-GPU): concatenate all-gathered chunks, rebasing firstIndex
// ---------------------------------------------------------------------------------------

struct MergeArgs {
  const unsigned char* chunks;
  unsigned long long stride;
  uint32_t n_chunks;
  uint32_t* out_cmds;
  uint32_t* out_count;  // [0] = commands, [1] = indices
  uint32_t* error_flag; // host-mapped
};

constexpr uint32_t kErrChunkOverflow = 2u;

constexpr uint32_t kMaxMergeChunks = 64;

__global__ __launch_bounds__(256) void mip_merge_draw_lists_kernel(const MergeArgs a) {
  __shared__ uint32_t s_count_base[kMaxMergeChunks + 1], s_index_base[kMaxMergeChunks + 1];
  if (threadIdx.x == 0) {
    uint32_t c = 0, s = 0;
    const uint32_t capacity = (uint32_t)((a.stride - 32u) / (kCmdWords * 4u));
    for (uint32_t k = 0; k < a.n_chunks; ++k) {
      const uint32_t* h = reinterpret_cast<const uint32_t*>(a.chunks + k * a.stride);
      uint32_t count = h[0];
      if (count > capacity) {  // the shard emitted more than the exchanged chunk holds
        count = capacity;
        if (blockIdx.x == 0) __hip_atomic_store(a.error_flag, kErrChunkOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      s_count_base[k] = c;
      s_index_base[k] = s;
      c += count;
      s += h[1];
    }
    s_count_base[a.n_chunks] = c;
    s_index_base[a.n_chunks] = s;
    if (blockIdx.x == 0) {
      a.out_count[0] = c;
      a.out_count[1] = s;
    }
  }
  __syncthreads();
  const uint32_t total_words = s_count_base[a.n_chunks] * kCmdWords;
  const uint32_t stride_threads = gridDim.x * blockDim.x;
  uint32_t chunk = 0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < total_words; j += stride_threads) {
    const uint32_t cmd = j / kCmdWords, field = j - cmd * kCmdWords;
    while (cmd >= s_count_base[chunk + 1]) ++chunk;  // j only grows
    const uint32_t* body = reinterpret_cast<const uint32_t*>(a.chunks + chunk * a.stride + 32);
    uint32_t v = body[(cmd - s_count_base[chunk]) * kCmdWords + field];
    if (field == 2u) v += s_index_base[chunk];
    a.out_cmds[j] = v;
  }
}

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

struct TriangleArgs {
  uint32_t* cmds;                 // compacted commands of the instance kernel; indexCount is rewritten
  const uint32_t* count;          // number of commands (device)
  const uint32_t* src_index_offset;
  const float4* model;            // n x mat4 of the same frame
  const float* vertices;          // consolidated positions, packed vec3
  const uint32_t* indices;        // consolidated indices
  uint32_t* out_indices;          // culled index stream (uvec3 out_index_buffer[])
  unsigned long long capacity;    // in indices
  uint32_t first_instance_base;
  uint32_t* error_flag;
  uint32_t* ticket;               // next command to hand out; zeroed by the host before the launch
  uint32_t geometry_finite;       // every position passed to mip_set_geometry was finite
  float pv[16];
};

constexpr uint32_t kErrIndexOverflow = 4u;

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

// One triangle of generate_work.comp:132-155: true = culled (back-facing or beyond one x/y bound).
// kAffine: the caller has checked that row 3 of `model` is (0,0,0,1) and that the geometry holds
// only finite positions. Then world.w = ((0*x + 0*y) + 0*z) + 1 is exactly 1 and pv[:,3] * world.w
// is exactly pv[:,3], so that row and those four products are skipped: same bits, 126 instead
// of 156 flops per triangle.
template <bool kAffine>
__device__ __forceinline__ bool triangle_test(const float (&model)[16], const float (&pv)[16], const float (&v)[9]) {
  float clip[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if constexpr (kAffine) {
      const float x = v[k * 3 + 0], y = v[k * 3 + 1], z = v[k * 3 + 2];
      float world[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) world[r] = model[0 * 4 + r] * x + model[1 * 4 + r] * y + model[2 * 4 + r] * z + model[3 * 4 + r];
#pragma unroll
      for (int r = 0; r < 4; ++r) clip[k][r] = pv[0 * 4 + r] * world[0] + pv[1 * 4 + r] * world[1] + pv[2 * 4 + r] * world[2] + pv[3 * 4 + r];
    } else {
      float world[4];
      glsl_mat4_mul_vec4(model, v[k * 3 + 0], v[k * 3 + 1], v[k * 3 + 2], 1.0f, world);
      glsl_mat4_mul_vec4(pv, world[0], world[1], world[2], world[3], clip[k]);
    }
  }
  const float a00 = clip[0][0], a01 = clip[0][1], a02 = clip[0][3];
  const float a10 = clip[1][0], a11 = clip[1][1], a12 = clip[1][3];
  const float a20 = clip[2][0], a21 = clip[2][1], a22 = clip[2][3];
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
    const uint32_t sw = __float_as_uint(clip[k][3]) & 0x80000000u;
    const float w = fabsf(clip[k][3]);
    const float x = __uint_as_float(__float_as_uint(clip[k][0]) ^ sw);
    const float y = __uint_as_float(__float_as_uint(clip[k][1]) ^ sw);
    xl = xl && (x < -w);
    xg = xg && (x > w);
    yl = yl && (y < -w);
    yg = yg && (y > w);
  }
  return cull || xl || xg || yl || yg;
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

#ifndef MIP_TRI_MIN_WAVES_PER_SIMD
#define MIP_TRI_MIN_WAVES_PER_SIMD 4
#endif

__global__ __launch_bounds__(256, MIP_TRI_MIN_WAVES_PER_SIMD) void mip_triangle_cull_kernel(const TriangleArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
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
    if (!fits && lane == 0) __hip_atomic_store(a.error_flag, kErrIndexOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

// Small frames (the reference's own regime: tens to a few thousand commands) leave a
// wave-per-command launch mostly idle and make one wave walk a 15 k-triangle mesh alone
// (measured 0.1 ms for 20 commands). There ONE WORKGROUP of 1024 threads takes a command:
// 1024 triangles per step, survivors ordered by a ballot per wave + the 16 wave totals in LDS.
constexpr uint32_t kTriBlock = 1024;

__global__ __launch_bounds__(kTriBlock) void mip_triangle_cull_block_kernel(const TriangleArgs a) {
  __shared__ uint32_t s_wave[2][kTriBlock / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t count = *a.count;
  float pv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) pv[k] = a.pv[k];

  for (uint32_t c = blockIdx.x; c < count; c += gridDim.x) {
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
    if (!fits && tid == 0) __hip_atomic_store(a.error_flag, kErrIndexOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const size_t dst_tri = (size_t)first_index / 3u;
    const uint32_t* tri_indices = a.indices + (size_t)src_tri * 3;
    uint32_t survivors = 0, buf = 0;
    __syncthreads();  // the previous command's last totals have been read
    for (uint32_t t0 = 0; t0 < n_tris; t0 += kTriBlock, buf ^= 1u) {
      const uint32_t t = t0 + tid;
      const bool valid = t < n_tris;
      const uint32_t* ip = tri_indices + (size_t)(valid ? t : 0u) * 3;
      const uint32_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
      const bool keep = valid && !triangle_culled(affine, model, pv, a.vertices, (long long)vertex_offset, i0, i1, i2);
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

// compact_draw_stream.comp runs after generate_work: commands whose triangles all died are
// dropped, order kept. One workgroup of 1024 threads walks the (already dense) list.
struct RecompactArgs {
  const uint32_t* in_cmds;
  const uint32_t* in_count;
  uint32_t* out_cmds;
  uint32_t* out_count;
};

__global__ __launch_bounds__(1024) void mip_recompact_kernel(const RecompactArgs a) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_running;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t count = *a.in_count;
  if (tid == 0) s_running = 0;
  __syncthreads();
  for (uint32_t base = 0; base < count; base += 1024u) {
    const uint32_t k = base + tid;
    const bool valid = k < count;
    uint32_t w[kCmdWords];
#pragma unroll
    for (uint32_t f = 0; f < kCmdWords; ++f) w[f] = valid ? a.in_cmds[(size_t)k * kCmdWords + f] : 0u;
    const bool keep = valid && w[0] > 0u;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = s_running, total = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
      if (q < wave) before += s_wave[q];
      total += s_wave[q];
    }
    if (keep) {
      uint32_t* dst = a.out_cmds + (size_t)(before + lanes_below(mask)) * kCmdWords;
#pragma unroll
      for (uint32_t f = 0; f < kCmdWords; ++f) dst[f] = w[f];
    }
    __syncthreads();
    if (tid == 0) s_running += total;
    __syncthreads();
  }
  if (tid == 0) *a.out_count = s_running;
}

// ---------------------------------------------------------------------------------------
// Row f-4, second consumer: the shadow pass's per-light draw lists
// ---------------------------------------------------------------------------------------
// src/renderer/systems/shadow_mapping.rs:405-478: for every light, for EVERY mesh entity (no
// culling) `pick_lod(index_buffers, light_position, mesh_position)` and
// `cmd_draw_indexed(index_count, 1, 0, 0, draw_index)`. As indirect lists over the consolidated
// buffers (the addressing cull_pass uses, cull_pipeline.rs:540-553): for light l and instance i
//   out[l*n + i] = { index_len[lod], 1, index_offset[lod], vertex_offset, first_instance_base + i }.
// One workgroup per 256 instances: positions and mesh data are read once, each light's 256
// commands go through LDS so the stores are whole 1-KiB rows per wave (5 120 contiguous bytes
// per tile and light). HBM-bound: 16 B read + n_lights * 20 B written per instance.
constexpr uint32_t kMaxLights = 16;  // the shadow atlas is DIM x DIM = 4 x 4 maps, shadow_mapping.rs:24

struct LightListArgs {
  const float* pos;          // n*3
  const uint32_t* mesh_id;   // n
  const MeshEntry* meshes;   // m
  const MeshDraw* mesh_draw; // m
  uint32_t* out;             // n_lights * n * 5 words
  uint32_t n;
  uint32_t n_lights;
  uint32_t first_instance_base;
  float light[kMaxLights][3];
};

template <bool kAligned16>
__global__ __launch_bounds__(kTile) void mip_light_draw_lists_kernel(const LightListArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_row[2][kTile * kCmdWords];
  const uint32_t tid = threadIdx.x;
  const uint32_t first = blockIdx.x * kTile;
  const uint32_t i = first + tid;
  const bool active = i < a.n;
  const uint32_t in_tile = a.n - first < kTile ? a.n - first : kTile;
  const uint32_t words = in_tile * kCmdWords;

  float px = 0.f, py = 0.f, pz = 0.f;
  uint32_t len0 = 0, len1 = 0;
  uint4 md = make_uint4(0, 0, 0, 0);
  if (active) {
    px = a.pos[(size_t)i * 3 + 0];
    py = a.pos[(size_t)i * 3 + 1];
    pz = a.pos[(size_t)i * 3 + 2];
    const uint32_t mesh = a.mesh_id[i];
    len0 = a.meshes[mesh].len0;
    len1 = a.meshes[mesh].len1;  // falls back to LOD 0 when the mesh has one LOD (helpers.rs:6)
    md = *reinterpret_cast<const uint4*>(&a.mesh_draw[mesh]);
  }
  for (uint32_t l = 0; l < a.n_lights; ++l) {
    uint32_t* row = s_row[l & 1u];
    if (active) {
      // (light - mesh).magnitude() > 10, helpers.rs:4-6, as in the instance kernel
      const float dx = a.light[l][0] - px, dy = a.light[l][1] - py, dz = a.light[l][2] - pz;
      const float dist_sq = dx * dx + dy * dy + dz * dz;
      const bool far_lod = dist_sq > kLodDistSqThreshold;
      uint32_t* c = &row[tid * kCmdWords];
      c[0] = far_lod ? len1 : len0;   // indexCount
      c[1] = 1u;                      // instanceCount
      c[2] = far_lod ? md.z : md.y;   // firstIndex: the LOD's range in the consolidated index buffer
      c[3] = md.x;                    // vertexOffset
      c[4] = a.first_instance_base + i;  // firstInstance = draw_index, shadow_mapping.rs:475
    }
    __syncthreads();  // the other buffer is free again: its readers passed the previous barrier
    uint32_t* dst = a.out + ((size_t)l * a.n + first) * kCmdWords;
    if constexpr (kAligned16) {
      // n % 4 == 0: every tile row starts on a 16-B boundary and in_tile % 4 == 0
      for (uint32_t q = tid; q * 4u < words; q += kTile)
        reinterpret_cast<uint4*>(dst)[q] = reinterpret_cast<const uint4*>(row)[q];
    } else {
      for (uint32_t w = tid; w < words; w += kTile) dst[w] = row[w];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Extension (BASELINE config 5): skinned instances — joint palette + posed mesh-space box
// ---------------------------------------------------------------------------------------
// The reference has no skinning: this is specified from glTF 2.0
// (section 3.7.3, skins) and checked against this repository's oracle only (orc_skinned_bounds).
//   L_k = T(t_k) * R(q_k) * S(s_k)           the animated LOCAL transform of joint k
//   G_k = G_parent(k) * L_k                  (roots: G_k = L_k; parents precede children)
//   J_k = G_k * inverseBind_k                the palette entry the vertex shader blends
//   posed box = union over k of J_k * joint_box_k          (mesh space, 8 corners per joint)
// A skinned vertex is a convex combination of J_k * v over the joints that influence it, so the
// union of the transformed per-joint bind-pose boxes bounds the posed mesh. That box takes the
// place of GltfMesh.aabb for the instance: the instance kernel reads it (KernelArgs.box_override)
// and runs rows a-2 / a-3 / a-7 on it unchanged — 8 corners under M, fold, centre/half round
// trip, planes, command. All matrices here are affine 3x4 (column-major, a[c*3 + r]); a product
// is, per column c, the column axpys (a0*b0c + a1*b1c) + a2*b2c, plus "+ a3" for the
// translation column — no FMA, this order.
//
// Mapping: one lane per (instance, joint); a wave holds floor(64 / J) instances, a workgroup four
// waves. The two steps that would run mostly idle lanes are re-packed through LDS:
//   hierarchy  level by level over the whole workgroup: the (instance, joint) pairs of one depth
//              are dense in the thread index, so a level costs one or two wave-wide 3x4 products
//              per workgroup instead of one per wave and level (a lane-per-joint loop leaves
//              4 of 5 lanes idle on a humanoid);
//   box fold   one thread per (instance, component) runs over the joints in ascending order —
//              the oracle's order — instead of a log-step exchange of six values per lane.
// Poses are read as five 8-byte loads per lane (40 B, lane-contiguous); palette entries leave
// through LDS so that every store instruction is 1 KiB contiguous: 40 B read + 64 B written per
// joint. What bounds the kernel is workgroup lifetime x resident workgroups (its phases are
// separated by barriers), so registers and LDS are kept small: 66 VGPRs, 18 KB, 7 waves per SIMD.
constexpr uint32_t kMaxJoints = 32;
constexpr uint32_t kPoseWords = 10;  // t xyz, q ijkw, s xyz
constexpr uint32_t kSkinBlock = 256;

struct alignas(16) JointEntry {
  float ibm[12];     // rows 0..2 of inverseBindMatrices[k], column-major 3x4
  float box[6];      // min xyz, max xyz of the bind-pose vertices weighted to this joint; min > max: none
  int32_t parent;    // < k, or -1
  uint32_t sorted;   // entry i: the i-th joint in depth order and its parent, joint | parent << 8
};
static_assert(sizeof(JointEntry) == 80, "JointEntry layout");

struct SkinArgs {
  const float* poses;          // n * J * 10, 8-byte aligned
  const JointEntry* joints;    // J
  float4* palette;             // n * J * 4 (mat4 column-major) or null
  float* local_box;            // n*6: min xyz, max xyz of the posed mesh (the fold's raw result)
  uint32_t n;
  uint32_t n_joints;
  uint32_t max_depth;
  uint32_t inv_joints;                      // ceil(2^16 / J): x / J == (x * inv) >> 16 for x < 256
  uint32_t level_inv[kMaxJoints + 1];       // ceil(2^16 / joints at depth d)
  uint8_t level_start[kMaxJoints + 2];      // depth d owns sorted entries [level_start[d], level_start[d+1])
};

__device__ __forceinline__ void affine_mul(const float (&a)[12], const float (&b)[12], float (&o)[12]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float v = a[0 * 3 + r] * b[c * 3 + 0] + a[1 * 3 + r] * b[c * 3 + 1] + a[2 * 3 + r] * b[c * 3 + 2];
      if (c == 3) v = v + a[9 + r];
      o[c * 3 + r] = v;
    }
}

__device__ __forceinline__ void lds_read12(const float* p, float (&m)[12]) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float4 a = q[0], b = q[1], c = q[2];
  m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
  m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
}

__device__ __forceinline__ void lds_write12(float* p, const float (&m)[12]) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(m[0], m[1], m[2], m[3]);
  q[1] = make_float4(m[4], m[5], m[6], m[7]);
  q[2] = make_float4(m[8], m[9], m[10], m[11]);
}

__global__ __launch_bounds__(kSkinBlock) void mip_skinned_bounds_kernel(const SkinArgs a) {
  __shared__ __attribute__((aligned(16))) float s_g[kSkinBlock * 12];   // L, then G, per (instance, joint) pair
  __shared__ float s_box[kSkinBlock * 6];                                // per pair: lo xyz, hi xyz
  __shared__ uint32_t s_sorted[kMaxJoints];                              // joints in depth order (joint | parent << 8)
  __shared__ uint32_t s_level[kMaxJoints + 2];                           // per depth: first sorted entry | ceil(2^16/count) << 8
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t J = a.n_joints;
  // The hierarchy loop below is a chain of short dependent steps; its per-level look-ups come from
  // LDS (a global or kernarg load per level would put ~1 us of cache latency on that chain).
  if (tid < J) s_sorted[tid] = a.joints[tid].sorted;
  if (tid < kMaxJoints + 2u) s_level[tid] = (uint32_t)a.level_start[tid] | ((tid <= kMaxJoints ? a.level_inv[tid] : 0u) << 8);
  const uint32_t ipw = 64u / J, ipb = 4u * ipw;           // instances per wave / workgroup
  const uint32_t block_first = blockIdx.x * ipb;          // < n by the grid size
  const uint32_t in_block = a.n - block_first < ipb ? a.n - block_first : ipb;
  const uint32_t g = (lane * a.inv_joints) >> 16, joint = lane - g * J;
  const uint32_t li = wave * ipw + g;                     // instance within the workgroup
  const bool valid = g < ipw && li < in_block;
  const uint32_t pair = valid ? li * J + joint : 0u;      // == wave*ipw*J + lane for valid lanes

  // ---- pose and joint constants ----
  const float2* pp = reinterpret_cast<const float2*>(a.poses + ((size_t)block_first * J + pair) * kPoseWords);
  const float2 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3], p4 = pp[4];
  const float t[kPoseWords] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y, p4.x, p4.y};
  const float4* jp = reinterpret_cast<const float4*>(&a.joints[valid ? joint : 0u]);
  const float4 j0 = jp[0], j1 = jp[1], j2 = jp[2], j3 = jp[3], j4 = jp[4];
  const float ibm[12] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w, j2.x, j2.y, j2.z, j2.w};
  const float box[6] = {j3.x, j3.y, j3.z, j3.w, j4.x, j4.y};

  // ---- local transform L = T * R * S ----
  float lr[3][3];
  quat_to_rotation(t[3], t[4], t[5], t[6], lr);
  float G[12];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) G[c * 3 + rr] = lr[rr][c] * t[7 + c];
  G[9] = t[0]; G[10] = t[1]; G[11] = t[2];
  if (valid) lds_write12(&s_g[pair * 12u], G);
  __syncthreads();

  // ---- hierarchy, one level at a time over the whole workgroup ----
  for (uint32_t d = 1; d <= a.max_depth; ++d) {
    const uint32_t lv = s_level[d];
    const uint32_t start = lv & 0xffu, cnt = (s_level[d + 1] & 0xffu) - start, inv = lv >> 8;
    if (tid < in_block * cnt) {
      const uint32_t inst_l = (tid * inv) >> 16;  // tid / cnt
      const uint32_t packed = s_sorted[start + (tid - inst_l * cnt)];
      const uint32_t k = packed & 0xffu, pk = packed >> 8;
      float P[12], Lk[12], Gk[12];
      lds_read12(&s_g[(inst_l * J + pk) * 12u], P);
      lds_read12(&s_g[(inst_l * J + k) * 12u], Lk);
      affine_mul(P, Lk, Gk);
      lds_write12(&s_g[(inst_l * J + k) * 12u], Gk);
    }
    __syncthreads();
  }
  if (a.max_depth) lds_read12(&s_g[pair * 12u], G);

  // ---- palette entry (mat4 per joint: 64 B per lane, lane-contiguous) ----
  float Jm[12];
  affine_mul(G, ibm, Jm);
  if (a.palette) {
    // Through LDS, so that every store instruction of the wave is 1 KiB contiguous (four 16-byte
    // stores per lane at a 64-byte lane stride reach 3.4 TB/s on this chip, lane-contiguous ones
    // 6.4: tools/micro/store_pattern.hip). Staged as 3x4 in the pair's own slot — nobody else
    // reads it after the last level — and written out as mat4: float4 q of the wave's range is
    // column q%4 of pair q/4, with w = 0,0,0,1.
    const uint32_t wave_inst0 = wave * ipw;
    const uint32_t wave_insts = wave_inst0 < in_block ? (in_block - wave_inst0 < ipw ? in_block - wave_inst0 : ipw) : 0u;
    const uint32_t wave_pair0 = wave_inst0 * J;
    const float* wave_lds = &s_g[wave_pair0 * 12u];
    if (valid) lds_write12(&s_g[pair * 12u], Jm);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4* out = a.palette + ((size_t)block_first * J + wave_pair0) * 4u;
    const uint32_t quads = wave_insts * J * 4u;
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {
      const uint32_t qd = lane + 64u * i;
      if (qd < quads) {
        const float* c = wave_lds + (qd >> 2) * 12u + (qd & 3u) * 3u;
        out[qd] = make_float4(c[0], c[1], c[2], (qd & 3u) == 3u ? 1.0f : 0.0f);
      }
    }
  }

  // ---- this joint's share of the posed box ----
  float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
  float hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
  if (!(box[0] > box[3] || box[1] > box[4] || box[2] > box[5])) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {  // corner order of src/ecs.rs:149-160
      const float x = box[(c & 1) ? 3 : 0], z = box[(c & 2) ? 5 : 2], y = box[(c & 4) ? 4 : 1];
      float v[3];
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) v[rr] = Jm[0 * 3 + rr] * x + Jm[1 * 3 + rr] * y + Jm[2 * 3 + rr] * z + Jm[9 + rr];
      fold_corner(v, lo, hi);
    }
  }
  if (valid) {
    float2* b2 = reinterpret_cast<float2*>(&s_box[pair * 6u]);
    b2[0] = make_float2(lo[0], lo[1]);
    b2[1] = make_float2(lo[2], hi[0]);
    b2[2] = make_float2(hi[1], hi[2]);
  }
  __syncthreads();

  // ---- fold over the joints, in the oracle's order: one thread per (instance, component) ----
  for (uint32_t e = tid; e < in_block * 6u; e += kSkinBlock) {
    const uint32_t inst_l = e / 6u, comp = e - inst_l * 6u;
    const bool is_min = comp < 3u;
    float v = is_min ? 3.40282347e+38f : -3.40282347e+38f;
    const float* src = &s_box[inst_l * J * 6u + comp];
    uint32_t k = 0;
    for (; k + 4u <= J; k += 4u) {  // four reads in flight, folded in ascending order
      const float x0 = src[k * 6u], x1 = src[k * 6u + 6u], x2 = src[k * 6u + 12u], x3 = src[k * 6u + 18u];
      // f32::min / f32::max: a NaN operand is ignored
      v = is_min ? fminf(fminf(fminf(fminf(v, x0), x1), x2), x3) : fmaxf(fmaxf(fmaxf(fmaxf(v, x0), x1), x2), x3);
    }
    for (; k < J; ++k) {
      const float x = src[k * 6u];
      v = is_min ? fminf(v, x) : fmaxf(v, x);
    }
    a.local_box[(size_t)block_first * 6u + e] = v;
  }
}

}  // namespace mip
