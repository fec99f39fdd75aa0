// instance_kernel.hpp — the gfx950 (CDNA4) kernel of the instance pipeline and the device helpers it shares.
//
// One fused, single-pass kernel per frame:
//
//   tile = 256 instances = one 256-thread workgroup (4 wave64), one instance per lane
//   loads   : pos (12 B) + quat (16 B) + scale (4 B) + mesh id (4 B)            = 36 B
//   compute : M = T·R·S, 8-corner world AABB, 6-plane test, LOD pick             (VALU, no FMA)
//   stores  : mat4 through an LDS transpose so every store instruction writes
//             1 KiB contiguous (64 B per instance; buffer_store_dwordx4 ... sc1 nt), 1 visibility bit, and — after a one-hop look-up of
//             the tile's exclusive prefix over per-tile granules and per-group atomic
//             accumulators of {count, Σ index_len} — the tile's surviving
//             VkDrawIndexedIndirectCommands, coalesced, in draw_index order.
//   Other kernels of the library live beside this file: merge_kernel.hpp (multi-GPU shard merge),
//   triangle_kernels.hpp (row f-1), light_lists_kernel.hpp (row f-4), skinning_kernel.hpp (config 5).
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
#define MIP_MIN_WAVES_PER_SIMD 5
#endif

#ifndef MIP_TILE
#define MIP_TILE 256
#endif

constexpr uint32_t kTile = MIP_TILE;      // instances per tile == threads per workgroup
constexpr uint32_t kWaves = kTile / 64;   // wave64
constexpr uint32_t kPieces = kTile / 16;  // 1-KiB pieces of a tile's matrices: one store instruction of a wave each
static_assert(kWaves >= 2, "wave 0 resolves the prefix while the other waves store");
// Waves 1 .. kWaves-1 store ALL pieces of the tile (wave 0's too), in contiguous runs: the first piece of wave w's run
// (256 instances: 0, 6, 11, 16 — runs of 6 + 5 + 5 KiB).
__host__ __device__ constexpr uint32_t store_run_first(uint32_t w) { return ((w - 1u) * kPieces + kWaves - 2u) / (kWaves - 1u); }
static_assert(store_run_first(1) == 0 && store_run_first(kWaves) == kPieces, "the runs cover the tile");
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
  // The first-mover rule (mark_tile_started): a tile's workgroup marks its granule STARTED when it starts, and whoever moves a
  // granule from an earlier launch's value to this launch's — the owner, or a wave that computes the tile for it — adds the tile
  // to its group's accumulator, so that groups complete without their owners. The mark is a returning device-scope atomic at the
  // head of every tile: +0.3 us per launch whether anybody helps or not (profiles/r05_first_mover.txt), so a launch follows
  // the rule only when recent launches had to help (`first_mover_rule`: note_helps_for_the_host), or when told to.
#ifndef MIP_FIRST_MOVER_ADDS  // 0: A/B builds only (tools/r05_first_mover.sh) — the rule of rounds 3-4 whatever happens
#define MIP_FIRST_MOVER_ADDS 1
#endif
  static constexpr bool kFirstMoverAdds = MIP_FIRST_MOVER_ADDS != 0;
  // ---- what every wave of every tile reads before its first instruction that depends on memory: 184 bytes = three 64-byte lines
  //      of the argument block (with the fields in the order they were added, these were spread over five lines; one line more in
  //      front of the planes cost a 100 k launch 0.15 us: profiles/r05_first_mover.txt) ----
  const float* pos;             // n*3
  const float4* rot;            // n  [i,j,k,w]
  const float* scale;           // n
  const uint32_t* mesh_id;      // n
  const MeshEntry* meshes;      // m
  const MeshDraw* mesh_draw;    // m (the commands-first order gathers from it at the tile's head)
  uint32_t* cmds;               // n*5 or null
  // the frame: in the argument block for a direct launch; `frame_ring` (device memory, 128-B entries)
  // instead when the launch is a node of a recorded graph, so that a replay can carry a new camera
  // without re-recording: the host refreshes the ring with one copy per replay
  const uint32_t* frame_ring;   // this launch's FrameWords, or null
  uint32_t n;
  uint32_t first_instance_base;
  uint32_t first_index_base;
  float planes[24];
  float cam[3];
  // ---- the tile's aggregate and its stores ----
  uint32_t n_tiles;
  uint32_t epoch;               // 1 .. 2^31-1, unique per launch
  uint32_t group_shift;         // log2(tiles per group), <= 6
  uint32_t groups_cap;
  unsigned long long* status0;  // level 0: one tagged granule per tile
  unsigned long long* acc1;     // level 1: [2 parities][groups_cap] 64-bit accumulators
  unsigned long long* start1;   // level 1: exclusive prefix at the start of each group, 2 tagged granules
  float4* model;                // n*4 or null
  uint32_t* bitmap;             // ceil(n/32) or null
  uint32_t bitmap_words;
  uint32_t n_meshes;            // mesh-table entries
  uint32_t* draw_count;         // with cmds
  uint32_t* index_total;        // optional
  // ---- optional outputs, the cold path ----
  uint32_t* src_index_offset;   // optional: per emitted command, where its LOD's indices start (row f-1)
  float* world_aabb;            // n*6 or null
  uint4* tlas_instances;        // n x VkAccelerationStructureInstanceKHR (64 B) or null (row f-4)
  const unsigned long long* blas_address;  // m, BLAS device address per mesh (with tlas_instances)
  const float* box_override;    // n*8 or null: per-instance mesh-space box {min xyz, -, max xyz, -} that replaces the mesh table's (skinned instances)
  uint32_t wire_index_bits;     // MIP_OUT_WIRE_PACKED: bits of a record that hold the instance index (mesh id above them, LOD in bit 31)
  uint32_t one_mesh;            // 1: the mesh table has one entry (the kernel's leading argument h_one_mesh)
  uint32_t first_mover_rule;    // 1: this launch follows the first-mover rule (the host's choice: api_frame.hip, MIP_TUNE_FIRST_MOVER)
  uint32_t* error_flag;         // host-mapped error words
  uint32_t* help_counter;       // DEVICE memory: tile aggregates that waiting tiles computed themselves (MipTimings.prefix_helps), kHelpShards words
  uint32_t* helps_seen;         // DEVICE memory, one word beside the prefix state: the help count the last launch that looked saw
  uint32_t* help_hint;          // host-mapped word (or null: nobody is told): note_helps_for_the_host
#ifdef MIP_EXP_FAKE_DELAY
  uint32_t delay_first, delay_last;  // tuning builds: tiles in [first, last) idle in place of the look-up
#endif
#ifdef MIP_DEBUG_STAMPS
  unsigned long long* stamps;  // diagnostic build only: 8 realtime stamps per tile
  uint32_t debug_skip_publish_tile;  // diagnostic build only: tile index + 1 that never publishes (0 = off)
  uint32_t debug_tile_mult, debug_tile_add;  // diagnostic build only: tile = (blockIdx.x * mult + add) % n_tiles, a permutation of the
                                             // tiles (mult coprime to n_tiles; mult = add = n_tiles - 1 reverses them): the launch must
                                             // give the same bytes in ANY order workgroups start in (0 = off)
#endif
};

// The frame kernel's leading scalar arguments (MIP_FRAME_HEAD_PARAMS, at the kernel): 6 pointers, n, one_mesh — to the block's
// alignment — where KernelArgs starts in the kernarg segment.
constexpr uint32_t kFrameHeadBytes = 56;
static_assert(alignof(KernelArgs) == 8 && 6 * sizeof(void*) + 2 * sizeof(uint32_t) <= kFrameHeadBytes && kFrameHeadBytes % alignof(KernelArgs) == 0, "argument layout");

// Device-side image of a frame for recorded launches: 32 words (128 B).
//   [0..23] planes, [24..26] cam_pos, [27] first_instance_base, [28] first_index_base, [29..31] pad
constexpr uint32_t kFrameWords = 32;

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

// Device -> host error word(s), host-mapped memory. Every kind of error has a word of its own (plain stores from
// different kernels of one frame must not overwrite each other: a timeout in the shard kernel followed by an
// overflow seen by the merge kernel are BOTH needed by the host); the host ORs the words together.
// (words 0 and 4 were the time-outs of the in-kernel waits of rounds 1-3; no kernel waits on another workgroup any more:
//  resolve_prefix below)
constexpr uint32_t kErrChunkOverflow = 2u;   // word 1: a gathered shard list is longer than the exchanged chunk
constexpr uint32_t kErrIndexOverflow = 4u;   // word 2: culled_index_buffer too small
constexpr uint32_t kErrWireRecord = 8u;      // word 3: a wire record names a mesh outside the table
constexpr uint32_t kErrSemaphore = 32u;      // word 5: written by the HOST (a stream-ordered wait on an external semaphore expired)
constexpr uint32_t kErrWords = 6;
__device__ __forceinline__ void raise_error(uint32_t* error_flag, uint32_t bit) {
  __hip_atomic_store(error_flag + (31 - __builtin_clz(bit)), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

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

// 16-byte streaming store of an output that this launch never reads again: non-temporal
// (global_store_dwordx4 ... nt). Measured on whole 1-KiB-per-instruction streams: the byte mover of
// tools/micro/floor_1m.hip 17.1 -> 15.9 us at 1 M instances, this kernel 21.5 -> 20.0 us.
// (Pointer form: the skinning palette, where the descriptor form below measured 3 % slower.)
__device__ __forceinline__ void store_stream16(float4* p, float4 v) {
#ifndef MIP_EXP_NO_NT
  typedef float v4f __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
#else
  *p = v;
#endif
}

// The same through a buffer descriptor, with the cache-policy bits of the instruction chosen explicitly
// (gfx940+ encoding of the builtin's last operand: 1 = sc0, 2 = nt, 16 = sc1). The descriptor's byte count
// bounds the store: lanes past it are dropped by the hardware, so the caller needs no range test.
// `sc1 nt` (streaming AND written through at device scope: the L2 keeps no dirty line to evict later) is the
// fastest encoding for the matrix stream at every size — tools/micro/store_bits.hip (byte mover, 1 M instances,
// loads + matrix stores): plain 16.1 us, nt 14.8, sc1 nt 13.9; 10 M: 175 / 167 / 164 — and in this kernel:
// 1 M 19.6 -> 18.5-19.1 us, 10 M 182.6 -> 178.5, 200 k 7.2 -> 7.0 (profiles/r02_store_policy_ab.txt).
// `sc1` without `nt` is as good at 1 M and collapses at 4 M (83 us against 64).
#ifndef MIP_STORE_AUX
#define MIP_STORE_AUX 18  // sc1 nt
#endif
typedef unsigned int stream_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_descriptor(void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);  // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
}
__device__ __forceinline__ void store_stream16(__amdgpu_buffer_rsrc_t d, uint32_t byte_offset, float4 v) {
  const stream_u32x4 w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(w, d, (int)byte_offset, 0, MIP_STORE_AUX);
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

// The same eight-corner fold WITHOUT enumerating the corners, bit-exact whenever no intermediate value
// can overflow or be NaN (separable_safe below). A corner's coordinate is
//     v = fl(fl(fl(P_x + P_y) + P_z) + m3),   P_x in {fl(m0*min.x), fl(m0*max.x)}, P_y, P_z likewise,
// and fl(a + b) is monotone non-decreasing in each operand, so the minimum over the eight independent
// choices is attained at the three smallest products and the maximum at the three largest:
//     min_corners v = fl(fl(fl(min P_x + min P_y) + min P_z) + m3)      (and the same with max).
// 18 multiplies, 18 min/max and 18 adds instead of 18 + 60 + 48 — the result the reference's
// loop over the corners (src/ecs.rs:146-173) produces, up to the sign of a zero.
__device__ __forceinline__ void instance_separable(const float (&r)[3][3], float px, float py, float pz,
                                                   float s, const MeshEntry& mb, Instance& o) {
  model_fast(r, px, py, pz, s, o);
  float lo[3], hi[3];
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    const float x0 = o.m[0 * 3 + rr] * mb.min_x, x1 = o.m[0 * 3 + rr] * mb.max_x;
    const float y0 = o.m[1 * 3 + rr] * mb.min_y, y1 = o.m[1 * 3 + rr] * mb.max_y;
    const float z0 = o.m[2 * 3 + rr] * mb.min_z, z1 = o.m[2 * 3 + rr] * mb.max_z;
    lo[rr] = fminf(x0, x1) + fminf(y0, y1) + fminf(z0, z1) + o.m[9 + rr];
    hi[rr] = fmaxf(x0, x1) + fmaxf(y0, y1) + fmaxf(z0, z1) + o.m[9 + rr];
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

// Called by ONE lane of the tile once its aggregate is known. `A` is anything that carries a prefix
// state: status0, acc1, start1, groups_cap, group_shift, epoch, error_flag (KernelArgs, or one view
// of the multi-view kernel).
// `adds_to_group`: false when a helping wave has added (or is going to add) this tile to the accumulator on the owner's behalf
// (A::kFirstMoverAdds; mark_tile_started tells the owner).
template <class A>
__device__ __forceinline__ void add_to_group(const A& a, uint32_t tile, uint32_t count, uint32_t sum) {
  const uint32_t group = tile >> a.group_shift;
  const unsigned long long add = ((unsigned long long)sum << 32) | (1ull << kAccCountBits) | count;
  (void)__hip_atomic_fetch_add(&a.acc1[((size_t)(a.epoch & 1u) * a.groups_cap + group) * kAccStrideWords], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class A>
__device__ __forceinline__ void publish_aggregate(const A& a, uint32_t tile, uint32_t count, uint32_t sum, bool adds_to_group = true) {
  const unsigned long long granule = ((unsigned long long)sum << 32) | ((unsigned long long)a.epoch << kTileCountBits) | count;
  __hip_atomic_store(&a.status0[tile], granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t group = tile >> a.group_shift;
  const uint32_t parity = a.epoch & 1u;
  if (adds_to_group) add_to_group(a, tile, count, sum);
  if ((tile & ((1u << a.group_shift) - 1u)) == 0u)  // first tile of the group: reset the next launch's word
    __hip_atomic_store(&a.acc1[((size_t)(parity ^ 1u) * a.groups_cap + group) * kAccStrideWords], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NO WAIT IN THIS LIBRARY DEPENDS ON ANOTHER WORKGROUP EVER RUNNING. A tile polls its predecessors' words a bounded
// number of times (kPatientPolls: tens of microseconds, many times what a predecessor that IS running needs to
// publish); a predecessor that still has not published — because the hardware has not started its workgroup yet,
// because another tenant of the GPU holds the compute units it would run on, because a debugger holds it — is then not
// waited for: the waiting wave computes that tile's aggregate ITSELF from the tile's inputs (`help`, supplied by the
// kernel: the same loads and the same arithmetic the owner runs, so the same {count, sum}) and publishes the level-0
// granule on the owner's behalf (idempotent: the owner stores the same 8 bytes later; a tile is added to its level-1
// accumulator exactly once — by its owner, or, in the frame kernel since round 5, by whoever was first to replace the granule
// of an earlier launch: mark_tile_started — so those stay exact). This is decoupled look-back with a fallback: the launch makes progress in ANY
// order the hardware starts workgroups in, with no ticket at the head of every workgroup (11 ns each, serialised) and
// no second launch. In-order dispatch — what an idle chip does — is now a performance property (no helps), not a
// correctness assumption. Helps are counted in a device word (MipTimings.prefix_helps).
// Round 3 and before: the same polls bounded at 0.5 s, then MIP_ERR_TIMEOUT, a ticketed re-issue and a recovery path.
constexpr uint32_t kPatientPolls = 64;
constexpr uint32_t kNotStartedPolls = 2;   // when a tile of the own group has not started (kernels whose tiles mark themselves STARTED)
constexpr uint32_t kImpatientPolls = 12;  // when more than eight of the words a tile needs are missing (resolve_prefix)
constexpr uint32_t kLevel1Window = 64;  // most recent groups whose accumulators a tile sums itself

// A granule of this launch is either PUBLISHED ({sum, tag, count <= kTile}) or, round 5, CLAIMED by a wave that is computing the
// tile's aggregate on the owner's behalf at this moment (count field all ones, which no tile can count): the other waiting waves
// then compute other tiles, or look again in a moment, instead of computing the same tile by the dozen.
constexpr uint32_t kClaimedCount = (1u << kTileCountBits) - 1u;
// ... or STARTED: the owner's workgroup is running (its first instruction swaps this in, kernels with A::kFirstMoverAdds) and has
// not published yet. A tile whose granule still carries an EARLIER launch's tag after the looking tile has done all of its own
// arithmetic is a tile whose workgroup the hardware has not started: waiting for it buys nothing, it is computed at once; a
// STARTED one publishes within microseconds and is waited for like a claimed one.
constexpr uint32_t kStartedCount = kClaimedCount - 1u;
static_assert(kTile < kStartedCount, "the markers must not be counts");
template <class A>
__device__ __forceinline__ bool granule_of_this_launch(const A& a, unsigned long long g) {
  return (((uint32_t)g >> kTileCountBits) & kMaxEpoch) == a.epoch;
}
template <class A>
__device__ __forceinline__ bool granule_ready(const A& a, unsigned long long g) {
  return granule_of_this_launch(a, g) && ((uint32_t)g & kClaimedCount) < kStartedCount;
}
// First instruction of a tile's workgroup (one lane). Returns whether a helping wave got to the tile first — it then adds the tile
// to the group's accumulator, and the owner must not (publish_aggregate's adds_to_group). The swap overwrites what the helper
// published with STARTED: the owner publishes the same 8 bytes again in a few microseconds.
// Tile 0's resolving wave (it has no prefix to resolve): has anybody helped since the last launch of this prefix state that looked?
// Then it tells the HOST (one system-scope store to a pinned word, `help_hint`), which makes its next few launches follow the
// first-mover rule (KernelArgs.first_mover_rule, api_frame.hip). One wave per launch asks, off every other tile's path, and the
// rule reaches the kernel as an argument. Measured and not kept: the helping waves saying so themselves (thousands of stores to
// one word are served one at a time, ~8 ns each: 2.5 M instances in scrambled order, 20 000 helps, 0.19 -> 0.36 ms), and every
// tile reading a device word for the rule (+0.2 us per launch at 100 k as a vector load behind the instance loads, +0.7 us as a
// scalar load, which the compiler waits for before it issues them). A degraded environment lasts many frames: the rule sets in
// a launch or two after the first help and ends a few launches after the last.
constexpr uint32_t kHelpShards = 16;  // words of help_counter the frame kernel's helpers spread their counts over
template <class A>
__device__ __forceinline__ void note_helps_for_the_host(const A& a, uint32_t lane) {
  if constexpr (A::kFirstMoverAdds) {
    if (!a.help_hint) return;
    uint32_t h = lane < kHelpShards ? __hip_atomic_load(&a.help_counter[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const uint32_t seen = __hip_atomic_load(a.helps_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h = wave_sum(h);
    if (lane == 0u && h != seen) {
      __hip_atomic_store(a.helps_seen, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.help_hint, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// Two halves, because the compiler waits for a returning atomic where it is issued when the issue sits in a branch of its own —
// here before the wave has issued its instance loads, a round trip at the head of every tile (+0.4-0.7 us per launch measured):
// the swap is issued as it stands, under the lanes of `who`, and its answer is waited for where it is needed, long after
// (loads and returning atomics come back in order, so every wait the compiler places for a later load covers it too).
template <class A>
__device__ __forceinline__ unsigned long long mark_tile_started_issue(const A& a, uint32_t tile, bool who) {
  unsigned long long g = ((unsigned long long)a.epoch << kTileCountBits) | kStartedCount;  // in: the STARTED granule, out: what was there
  const unsigned long long* p = &a.status0[tile];
  const unsigned long long lanes = __ballot(who);
  const uint32_t zero = 0;
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %[saved], exec\n\t"
      "s_mov_b64 exec, %[lanes]\n\t"
      "global_atomic_swap_x2 %[g], %[zero], %[g], %[p] sc0\n\t"
      "s_mov_b64 exec, %[saved]"
      : [g] "+v"(g), [saved] "=&s"(saved)
      : [lanes] "s"(lanes), [zero] "v"(zero), [p] "s"(p)
      : "memory");
  return g;
}
// whether a helping wave got to the tile first (meaningful in the lanes of `who` only)
template <class A>
__device__ __forceinline__ bool mark_tile_started_answer(const A& a, unsigned long long g) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(g));
  return granule_of_this_launch(a, g);
}
constexpr uint32_t kClaimPolls = 48;  // looks at a claimed tile before the claim is overridden (a help takes one or two of them)

// Σ aggregates of tiles [first, first + count), count <= 64, run by one whole wave: granules where they are published,
// `help` where they are not. `spread` rotates the order in which missing tiles are taken so that several waiting waves
// help different tiles first (each publishes what it computed; the others then find it).
template <class A, class Help>
__device__ __forceinline__ void sum_tiles_helping(const A& a, uint32_t first, uint32_t count, uint32_t lane, uint32_t spread,
                                                  Help& help, uint32_t& c, uint32_t& s, bool first_mover_rule, uint32_t self) {
  // (c and s are wave-uniform, and so is the set of tiles still to be summed; nothing per lane stays alive across a help: the
  //  cold path must fit the hot path's registers)
  unsigned long long pending = count >= 64u ? ~0ull : ((1ull << count) - 1ull);
  // where this wave starts among the tiles: spread over the `count` tiles there are (a rotation taken modulo 64 sent every wave
  // whose number fell beyond a short run back to its first tile)
  const uint32_t rot = count >= 64u ? (spread & 63u) : (count ? spread % count : 0u);
  uint32_t idle_looks = 0;
#pragma nounroll
  while (pending) {  // wave-uniform
    // every pass looks at ALL the tiles still missing with one coalesced load: what other waiting tiles have published since the
    // last pass is taken from them, and at most ONE tile per pass is computed here
    const bool mine = (pending >> lane) & 1ull;
    unsigned long long g = 0;
    if (mine) g = status_load(&a.status0[first + lane]);
    const bool ok = mine && granule_ready(a, g);
    c += wave_sum(ok ? ((uint32_t)g & ((1u << kTileCountBits) - 1u)) : 0u);
    s += wave_sum(ok ? (uint32_t)(g >> 32) : 0u);
    pending &= ~__ballot(ok);
    if (!pending) break;
    // Round 5: a tile that another wave is computing right now (claimed) is left to it; this wave takes one nobody has taken, and
    // when every missing tile is taken it looks again shortly — a claim is held by a wave that is RUNNING (it is inside `help`),
    // but nothing may depend on that: after kClaimPolls looks the claim is ignored. (Round 4: every waiting wave computed one
    // missing tile per pass whoever else was computing it — 45-70 k helps for the 1 950 tiles a scrambled 1 M launch is missing.)
    const bool claimed = mine && !ok && granule_of_this_launch(a, g);
    unsigned long long open = pending & ~__ballot(claimed);
    if (!open) {
      if (++idle_looks <= kClaimPolls) {
#if defined(MIP_DEBUG_STAMPS) && defined(MIP_EXP_HELP_STAMPS)  // experiment (tools/r05_help_stamps.py): looks at tiles somebody else is computing
        if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] += 1ull;
#endif
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      open = pending;  // the claimants are not getting there: compute it after all
    }
    const unsigned long long turned = rot ? ((open >> rot) | (open << (64u - rot))) : open;
    const uint32_t pick = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)__builtin_ctzll(turned) + rot) & 63u));
    const uint32_t u = first + pick;
    // take it: the stale (or overridden) granule this wave saw is replaced by the claim, unless somebody got there first
    const unsigned long long seen = (unsigned long long)__builtin_amdgcn_readlane((int)(uint32_t)g, pick) |
                                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(g >> 32), pick) << 32);
    bool took = true;
    if (lane == 0u) {
      unsigned long long expected = seen;
      // (the claim names its claimant — `self`, the waiting wave's own tile — so that an expired claim is taken over by
      //  ONE of the waves that gave up on it, with this same swap, and not computed by all of them at once: scrambled order,
      //  2.5 M instances: 20 400 helps for 7 700 missing tiles before)
      const unsigned long long claim = ((unsigned long long)(self + 1u) << 32) | ((unsigned long long)a.epoch << kTileCountBits) | kClaimedCount;
      took = __hip_atomic_compare_exchange_strong(&a.status0[u], &expected, claim, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    took = __builtin_amdgcn_readfirstlane((int)took) != 0;
    if (!took) continue;  // published or claimed since the look: the next pass sees which
    // the ONE agent that replaces an earlier launch's granule — this wave's swap just did, or the owner's first instruction would
    // have — adds the tile to its group (an overridden claim or a STARTED tile is somebody else's to add)
    const bool first_mover = first_mover_rule && !granule_of_this_launch(a, seen);
    idle_looks = 0;
#if defined(MIP_DEBUG_STAMPS) && defined(MIP_EXP_HELP_STAMPS)
    const unsigned long long help_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned long long agg = help(u);  // {sum : 32 | count : 32}, wave-uniform
    const uint32_t uc = (uint32_t)agg, us = (uint32_t)(agg >> 32);
#if defined(MIP_DEBUG_STAMPS) && defined(MIP_EXP_HELP_STAMPS)  // ticks inside help() in the low half, helps in the high half
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] += (__builtin_amdgcn_s_memrealtime() - help_t0) | (1ull << 32);
#endif
    if (lane == 0u) {
      __hip_atomic_store(&a.status0[u], ((unsigned long long)us << 32) | ((unsigned long long)a.epoch << kTileCountBits) | uc,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (a DEVICE word: as a system-scope atomic on the host-mapped error block every help cost ~1 us of PCIe round trip, serialised
      //  on one address — 10 000 helps per frame were 10 ms, and that, not the helping, was the degraded mode's cost)
      (void)__hip_atomic_fetch_add(a.help_counter + (A::kFirstMoverAdds ? u % kHelpShards : 0u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (first_mover) add_to_group(a, u, uc, us);
    }
    c += uc;
    s += us;
    pending &= ~(1ull << pick);
  }
}

// The prefix of `tile` without waiting for anybody (cold path of resolve_prefix).
template <class A, class Help>
__device__ __forceinline__ void resolve_prefix_unaided(const A& a, uint32_t tile, uint32_t lane, Help& help,
                                                                 uint32_t& base_count, uint32_t& base_sum, bool first_mover_rule) {
  const uint32_t group = tile >> a.group_shift;
  const uint32_t group_first = group << a.group_shift;
  const uint32_t per_group = 1u << a.group_shift;
  const unsigned long long* acc = &a.acc1[(size_t)(a.epoch & 1u) * a.groups_cap * kAccStrideWords];
  uint32_t c = 0, s = 0;
  sum_tiles_helping(a, group_first, tile - group_first, lane, tile, help, c, s, first_mover_rule, tile);  // earlier tiles of the own group
  // earlier groups: from the published start of group g_lo if there is one, else all of them
  uint32_t lo = group > kLevel1Window ? group - kLevel1Window : 0u;
  if (lo > 0u) {
    const unsigned long long pc = status_load(&a.start1[2 * (size_t)lo]), ps = status_load(&a.start1[2 * (size_t)lo + 1]);
    if ((uint32_t)(pc >> 32) == a.epoch && (uint32_t)(ps >> 32) == a.epoch) { c += (uint32_t)pc; s += (uint32_t)ps; }
    else lo = 0u;
  }
#pragma nounroll
  for (uint32_t gb = lo; gb < group; gb += 64u) {
    const uint32_t gg = gb + lane;
    const bool valid = gg < group;
    unsigned long long open;
    {
      uint32_t myc = 0, mys = 0;
      bool complete = false;
      if (valid) {
        const unsigned long long w = status_load(&acc[(size_t)gg * kAccStrideWords]);
        complete = ((uint32_t)w >> kAccCountBits) == per_group;
        if (complete) { myc = (uint32_t)w & ((1u << kAccCountBits) - 1u); mys = (uint32_t)(w >> 32); }
      }
      c += wave_sum(myc);
      s += wave_sum(mys);
      open = __ballot(valid && !complete);
    }
    // wave-uniform: groups some of whose tiles have not added yet — tile by tile. Hundreds of waiting tiles need the same missing
    // aggregates: every one of them takes the open groups in an order of its own (a rotation hashed from its tile number, and
    // within a group another one), so that together they cover the missing set once instead of all walking it from the bottom —
    // what one publishes the others find (tiles in scrambled order, 1 M instances: 57 k helps and 19 ms per frame before,
    // profiles/r04_selfhelp_any_order.txt for after).
    // (Round 5 measured a second level — a wave that has walked a whole group publishes the group's total for the others to read —
    //  and four groups' granules loaded per round trip, both on top of the claims: no faster in scrambled order; not kept:
    //  profiles/r05_selfhelp_claims.txt.)
    // ... uniformly over the groups of this window (round 4 took 6 hash bits as they came: with 30 earlier groups more than half
    // of the waves started at group 0 and walked upwards together — a chain of 30 visits instead of one round of helps)
    const uint32_t span = group - gb < 64u ? group - gb : 64u;
    const uint32_t rot = ((tile * 2654435761u) >> 12) % span;
    bool walked = false;
#pragma nounroll
    while (open) {
      if (first_mover_rule && walked) {
        // helping waves complete groups on their owners' behalf: what has become complete while this wave walked the last group is
        // taken from its accumulator (one look for all of them) instead of tile by tile
        uint32_t myc = 0, mys = 0;
        bool complete = false;
        if ((open >> lane) & 1ull) {
          const unsigned long long w = status_load(&acc[(size_t)gg * kAccStrideWords]);
          complete = ((uint32_t)w >> kAccCountBits) == per_group;
          if (complete) { myc = (uint32_t)w & ((1u << kAccCountBits) - 1u); mys = (uint32_t)(w >> 32); }
        }
        c += wave_sum(myc);
        s += wave_sum(mys);
        open &= ~__ballot(complete);
        if (!open) break;
      }
      const unsigned long long turned = rot ? ((open >> rot) | (open << (64u - rot))) : open;
      const uint32_t pick = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)__builtin_ctzll(turned) + rot) & 63u));
      sum_tiles_helping(a, (gb + pick) << a.group_shift, per_group, lane, tile * 7u + pick, help, c, s, first_mover_rule, tile);
      open &= ~(1ull << pick);
      walked = true;
    }
  }
  base_count = c;
  base_sum = s;
}

// Run by one whole wave after publish_aggregate(tile). Returns the exclusive prefix of `tile`:
//   prefix = start1[g_lo]  +  Σ acc1[g_lo .. g-1]  +  Σ status0[first tile of g .. tile-1]
// with g_lo = max(0, g - 64). start1[g] (the exclusive prefix at the start of group g) is
// published for free by the first tile of group g once it has resolved its own prefix; the
// entry read here is 64 groups = thousands of tiles back, i.e. long resolved, so the look-up
// stays ONE round of <= 63 + 64 + 1 words for any N (without it every tile would read every
// earlier group: quadratic, measured +100 us at 10 M instances).
// `help(u)`: the aggregate {sum : 32 | count : 32} of tile u computed from its inputs by the calling wave.
template <class A, class Help>
__device__ __forceinline__ void resolve_prefix(const A& a, uint32_t tile, uint32_t lane,
                                               uint32_t& base_count, uint32_t& base_sum, Help help, bool first_mover_rule = false) {
  const uint32_t group = tile >> a.group_shift;
  const uint32_t group_first = group << a.group_shift;
  const uint32_t r = tile - group_first;  // earlier tiles of the own group (< 64)
  const uint32_t per_group = 1u << a.group_shift;
  const uint32_t g_lo = group > kLevel1Window ? group - kLevel1Window : 0u;
  const unsigned long long* acc = &a.acc1[(size_t)(a.epoch & 1u) * a.groups_cap * kAccStrideWords];
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
  uint32_t c = 0, s = 0, polls = 0;
  for (;;) {
    bool not_started = false;
    if (!ready0) {
      const unsigned long long g = status_load(e0);
      if (granule_ready(a, g)) {
        ready0 = true;
        c += (uint32_t)g & ((1u << kTileCountBits) - 1u);
        s += (uint32_t)(g >> 32);
      } else {
        not_started = first_mover_rule && !granule_of_this_launch(a, g);
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
#if defined(MIP_DEBUG_STAMPS) && !defined(MIP_EXP_HELP_STAMPS)
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 6] += 1;
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] += (unsigned long long)__popcll(__ballot(!all));
#endif
    if (__all(all)) break;
    // Patience in proportion: a handful of words missing are contemporaries that publish in a moment (the full budget); dozens
    // missing are workgroups that have not started — waiting for those buys nothing, and every waiting wave waits the same.
    // ... and a predecessor whose workgroup has not even started (its granule is not marked STARTED: mark_tile_started) after this
    // tile has done all of its own arithmetic and looked twice is not going to publish in a moment either
    const uint32_t budget = (uint32_t)__popcll(__ballot(!all)) > 8u ? kImpatientPolls : (__any(not_started) ? kNotStartedPolls : kPatientPolls);
    if (__builtin_expect(++polls > budget, 0)) {  // scalar: wave-uniform
      ok = false;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (__builtin_expect(ok, 1)) {
    base_count = wave_sum(c);
    base_sum = wave_sum(s);
  } else {
    resolve_prefix_unaided(a, tile, lane, help, base_count, base_sum, first_mover_rule);
  }
  if (r == 0u && group > 0u && lane == 0u) {  // first tile of a group: publish the group's start
    unsigned long long* p = &a.start1[2 * (size_t)group];
    __hip_atomic_store(p, ((unsigned long long)a.epoch << 32) | base_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, ((unsigned long long)a.epoch << 32) | base_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------

// The finite test that selects the collapsed arithmetic (instance_fast): a sum of magnitudes is NaN/inf
// as soon as one term is (or the sum overflows — then the literal path, exact for everything, runs).
// ONE definition: the frame kernel's per-wave test and the upload-time census
// (mip_count_nonfinite_kernel) that lets the host pick the kernel without a cold path must agree.
__device__ __forceinline__ float finite_magnitude(const float (&r)[3][3], float px, float py, float pz, float sc) {
  float mag = fabsf(px) + fabsf(py) + fabsf(pz) + fabsf(sc);
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
#pragma unroll
    for (int c = 0; c < 3; ++c) mag += fabsf(r[rr][c]);
  return mag;
}
constexpr float kFiniteLimit = 3.0e38f;

// separable_safe: every product, partial sum and corner coordinate of the world-box computation is finite,
// so instance_separable is exact. `box_abs` bounds the sum of the magnitudes of the six mesh-box coordinates
// (the instance's own box, or the largest such sum over the mesh table). |corner| <= sum|M| * box + sum|p| with
// sum|M| <= sum|R| * |s| (1 + 2^-23); anything non-finite in the inputs makes the bound NaN or inf.
constexpr float kSeparableLimit = 1.0e37f;
__device__ __forceinline__ float separable_bound(const float (&r)[3][3], float px, float py, float pz, float sc, float box_abs) {
  float sum_r = 0.0f;
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
#pragma unroll
    for (int c = 0; c < 3; ++c) sum_r += fabsf(r[rr][c]);
  return sum_r * fabsf(sc) * box_abs + (fabsf(px) + fabsf(py) + fabsf(pz));
}

// Upload-time census: how many instances of [first, first + count) fail the finite test. The
// host keeps the total (mip_set_instances*, mip_update_instances); while it is zero the frame is
// launched as mip_instance_pipeline_kernel<.., .., kGeneral = false>.
struct CensusArgs {
  const float* pos; const float4* rot; const float* scale;
  const uint32_t* mesh_id;  // or null: ids not checked
  uint32_t n_meshes;
  uint32_t first, count;
  float box_abs;  // largest sum of |box coordinates| over the mesh table
  uint32_t* out;  // [0] += number of instances that need the kernel with the fall-back paths
                  // [1] += number of instances whose mesh id is outside the table (the frame kernel gathers
                  //         meshes[mesh_id] unchecked: such an upload is refused, as mip_set_instances refuses it on the host)
};
// (static: this header is included by every translation unit of the library; launched from api_context.hip)
static __global__ __launch_bounds__(256) __attribute__((unused)) void mip_count_nonfinite_kernel(const CensusArgs a) {
  uint32_t bad = 0, bad_id = 0;
  for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < a.count; k += gridDim.x * 256u) {
    const size_t i = (size_t)a.first + k;
    const float4 q = a.rot[i];
    float r[3][3];
    quat_to_rotation(q.x, q.y, q.z, q.w, r);
    const float px = a.pos[3 * i], py = a.pos[3 * i + 1], pz = a.pos[3 * i + 2], sc = a.scale[i];
    const bool ok = finite_magnitude(r, px, py, pz, sc) < kFiniteLimit && separable_bound(r, px, py, pz, sc, a.box_abs) < kSeparableLimit;
    bad += ok ? 0u : 1u;
    if (a.mesh_id) bad_id += a.mesh_id[i] >= a.n_meshes ? 1u : 0u;
  }
  bad = wave_sum(bad);
  bad_id = wave_sum(bad_id);
  if ((threadIdx.x & 63u) == 0u && bad) atomicAdd(a.out, bad);
  if ((threadIdx.x & 63u) == 0u && bad_id) atomicAdd(a.out + 1, bad_id);
}

// The wire form of a shard's draw list (include/mi_instance_pipeline.h, MIP_OUT_WIRE): blocks of 256 8-byte records
// {firstInstance, mesh | lod << 31}, each block behind a 16-byte header of FOUR firstIndex words — word q is the firstIndex of
// the block's record 64 q, so every 64 records ("sub-block": what one wave of the merge kernel expands with one DPP scan, on
// its own, without a barrier) carry their own anchor. 8.06 B per command instead of 20 through the all-gather;
// mip_merge_wire_lists_kernel (merge_kernel.hpp) expands it against the replicated mesh table.
// (ABI 3 anchored only record 0 of a block: the merge then needed the whole workgroup, two barriers and an LDS prefix per block.)
constexpr uint32_t kWireSubBlock = 64;
constexpr uint32_t kWireBlockCmds = 256, kWireBlockHeaderWords = 4, kWireBlockWords = kWireBlockHeaderWords + 2 * kWireBlockCmds;
// MIP_OUT_WIRE_PACKED: one word per record in blocks of 64 = one sub-block behind its own self-describing 16-byte header
// {firstIndex, first_instance_base, index_bits, 0}: 4.25 B per command
constexpr uint32_t kWirePackedBlockCmds = kWireSubBlock, kWirePackedBlockWords = kWireBlockHeaderWords + kWirePackedBlockCmds;

// Copy-out of a tile's `tile_count` staged commands (LDS, kCmdLdsWords each: [2] tile-relative firstIndex,
// [4] firstInstance, [5] mesh | lod << 31) as wire records at list positions base_count .. ; run by one wave.
__device__ __forceinline__ void wire_copy_out(uint32_t* body, const uint32_t* s_cmd, uint32_t lane, uint32_t base_count,
                                              uint32_t first_index_add, uint32_t tile_count) {
  for (uint32_t k = lane; k < tile_count; k += 64u) {
    const uint32_t g = base_count + k, block = g / kWireBlockCmds, slot = g % kWireBlockCmds;
    uint32_t* b = body + (size_t)block * kWireBlockWords;
    const uint32_t* c = &s_cmd[k * kCmdLdsWords];
    *reinterpret_cast<uint2*>(b + kWireBlockHeaderWords + 2u * slot) = make_uint2(c[4], c[5]);
    if (slot % kWireSubBlock == 0u) b[slot / kWireSubBlock] = c[2] + first_index_add;  // this record anchors its sub-block
  }
}

// The same for the PACKED wire form (MIP_OUT_WIRE_PACKED): one word per command,
// instance index in the frame | mesh << index_bits | lod << 31; block header {firstIndex, first_instance_base, index_bits, 0}.
__device__ __forceinline__ void wire_packed_copy_out(uint32_t* body, const uint32_t* s_cmd, uint32_t lane, uint32_t base_count,
                                                     uint32_t first_index_add, uint32_t tile_count, uint32_t first_instance_base,
                                                     uint32_t index_bits) {
  for (uint32_t k = lane; k < tile_count; k += 64u) {
    const uint32_t g = base_count + k, block = g / kWirePackedBlockCmds, slot = g % kWirePackedBlockCmds;
    uint32_t* b = body + (size_t)block * kWirePackedBlockWords;
    const uint32_t* c = &s_cmd[k * kCmdLdsWords];
    b[kWireBlockHeaderWords + slot] = (c[4] - first_instance_base) | ((c[5] & 0x7fffffffu) << index_bits) | (c[5] & 0x80000000u);
    if (slot == 0u) *reinterpret_cast<uint4*>(b) = make_uint4(c[2] + first_index_add, first_instance_base, index_bits, 0u);
  }
}

// Model matrix + world box of one instance in the arithmetic tier its WAVE takes (kGeneral) or the separable fold
// (census-selected launches). Shared by the owner of a tile and by a wave that computes the tile's aggregate in its
// place (help_tile_aggregate): same loads, same functions, same tier decision per 64 consecutive instances.
template <bool kBoxOverride, bool kGeneral, class Args>
__device__ __forceinline__ void instance_tiered(const Args& a, uint32_t il, const float (&r)[3][3], float px, float py, float pz,
                                                float sc, MeshEntry& mb, Instance& inst) {
  if constexpr (kGeneral) {
    float mag = finite_magnitude(r, px, py, pz, sc);
    if constexpr (kBoxOverride) {  // skinned instances: the posed mesh-space box computed by mip_skinned_bounds_kernel
      const float4* b4 = reinterpret_cast<const float4*>(a.box_override) + 2 * (size_t)il;  // {min xyz, -}, {max xyz, -}
      const float4 lo = b4[0], hi = b4[1];
      mb.min_x = lo.x; mb.min_y = lo.y; mb.min_z = lo.z;
      mb.max_x = hi.x; mb.max_y = hi.y; mb.max_z = hi.z;
    }
    // the instance's own box: a mesh-table box is finite, a box override may be anything
    const float box_abs = fabsf(mb.min_x) + fabsf(mb.min_y) + fabsf(mb.min_z) + fabsf(mb.max_x) + fabsf(mb.max_y) + fabsf(mb.max_z);
    mag += box_abs;
    const bool all_finite = mag < kFiniteLimit;
    const bool separable = all_finite && separable_bound(r, px, py, pz, sc, box_abs) < kSeparableLimit;
    // three tiers, chosen per wave: the separable fold; the corner enumeration (finite inputs whose
    // intermediate values may overflow); the literal chain (non-finite inputs)
    if (__builtin_expect(__any(!separable), 0)) {
      if (__any(!all_finite)) instance_general(r, px, py, pz, sc, mb, inst);
      else instance_fast(r, px, py, pz, sc, mb, inst);
    } else {
      instance_separable(r, px, py, pz, sc, mb, inst);
    }
  } else {
    instance_separable(r, px, py, pz, sc, mb, inst);  // the upload-time census found every instance separable_safe
  }
}

// pick_lod (helpers.rs:3-11) against the frame's reference point, as the frame kernel evaluates it.
__device__ __forceinline__ bool lod_is_far(const float (&cam)[3], float px, float py, float pz) {
  const float dx = cam[0] - px, dy = cam[1] - py, dz = cam[2] - pz;
  const float dist_sq = dx * dx + dy * dy + dz * dz;
  return dist_sq > kLodDistSqThreshold;
}

// The kernel's argument block as the COLD path sees it: re-read from the kernarg segment through a pointer the compiler
// cannot see through. A cold path that used the hot path's copies (the six planes, the camera, eight base pointers: ~45
// scalar registers) would keep them alive across the whole kernel: measured, the allocator then parks them in vector-
// register lanes — ~50 v_writelane / s_mov at the head of every workgroup, 1 M instances 18.65 -> 19.3 us. Reloading
// costs the hot path nothing.
template <class Args, uint32_t kSkipBytes = 0>
__device__ __forceinline__ const __attribute__((address_space(4))) Args* cold_kernel_args() {
  typedef const __attribute__((address_space(4))) Args* Ptr;
  // (the views kernel takes ONE argument, by value: it starts the segment; the frame kernel's block follows its leading scalars)
  Ptr p = (Ptr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + kSkipBytes);
  asm volatile("" : "+s"(p));
  return p;
}

// The frame of a launch (planes, LOD reference point) for the cold path: from the frame ring of a recorded launch, or
// from the argument block.
__device__ __forceinline__ void cold_frame(const __attribute__((address_space(4))) KernelArgs* ka, float (&planes)[24], float (&cam)[3]) {
  const uint32_t* ring = ka->frame_ring;
  if (ring) {
#pragma unroll
    for (int k = 0; k < 24; ++k) planes[k] = __uint_as_float(__builtin_amdgcn_readfirstlane((int)ring[k]));
#pragma unroll
    for (int k = 0; k < 3; ++k) cam[k] = __uint_as_float(__builtin_amdgcn_readfirstlane((int)ring[24 + k]));
  } else {
#pragma unroll
    for (int k = 0; k < 24; ++k) planes[k] = ka->planes[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) cam[k] = ka->cam[k];
  }
}

// The aggregate {Σ index_len of the visible : 32 | emitted commands : 32} of tile u, computed by ONE wave from the
// tile's inputs — what the tile's own four waves publish. Cold: only a wave whose predecessor has not published
// within kPatientPolls gets here (resolve_prefix).
template <bool kBoxOverride, bool kGeneral>
__device__ __forceinline__ unsigned long long help_tile_aggregate(uint32_t u, uint32_t lane) {
  const auto* ka = cold_kernel_args<KernelArgs, kFrameHeadBytes>();
  float planes[24], cam[3];
  cold_frame(ka, planes, cam);
  const float* pos = ka->pos;
  const float4* rot = ka->rot;
  const float* scale = ka->scale;
  const uint32_t* mesh_id = ka->mesh_id;
  const MeshEntry* meshes = ka->meshes;
  const uint32_t n = ka->n;
  struct { const float* box_override; } box_args = {ka->box_override};
  uint32_t cnt = 0, sum = 0;
// (unrolled 2x / 4x: a help takes 13.7 / 12.4 us instead of 14.1 at 2.5 M in scrambled order, the launch the same: not kept)
#pragma nounroll
  for (uint32_t w = 0; w < kWaves; ++w) {
    const uint32_t j = u * kTile + w * 64u + lane;
    const bool active = j < n;
    const uint32_t jl = active ? j : n - 1u;
    const float px = pos[3 * (size_t)jl + 0], py = pos[3 * (size_t)jl + 1], pz = pos[3 * (size_t)jl + 2];
    const float4 q = rot[jl];
    const float sc = scale[jl];
    const uint32_t mesh = mesh_id[jl];
    const float4 mb0 = *reinterpret_cast<const float4*>(&meshes[mesh].min_x);
    const float4 mb1 = *reinterpret_cast<const float4*>(&meshes[mesh].max_x);
    MeshEntry mb;
    mb.min_x = mb0.x; mb.min_y = mb0.y; mb.min_z = mb0.z; mb.len0 = __float_as_uint(mb0.w);
    mb.max_x = mb1.x; mb.max_y = mb1.y; mb.max_z = mb1.z; mb.len1 = __float_as_uint(mb1.w);
    float r[3][3];
    quat_to_rotation(q.x, q.y, q.z, q.w, r);
    Instance inst;
    instance_tiered<kBoxOverride, kGeneral>(box_args, jl, r, px, py, pz, sc, mb, inst);
    const bool visible = active && !coarse_culled(inst, planes);
    const uint32_t len = lod_is_far(cam, px, py, pz) ? mb.len1 : mb.len0;
    cnt += (uint32_t)__popcll(__ballot(visible && len > 0u));
    sum += wave_sum(visible ? len : 0u);
  }
  return ((unsigned long long)sum << 32) | cnt;
}

// Tile aggregate assembled in LDS by the four waves: {Σ index_len : 32 | arrivals : 8 | - : 7 | helped first : 1 | count : 16}.
constexpr uint32_t kAggArrivalShift = 24;
constexpr uint32_t kAggHelpedFirst = 1u << 16;  // a helping wave adds this tile to its group's accumulator (mark_tile_started)

// kBoxOverride: every instance brings its own mesh-space box (KernelArgs.box_override; the skinned
// extension).
// kGeneral: the kernel carries the literal arithmetic chain for non-finite inputs (taken per wave).
// That path costs 16-20 VGPRs (80-84 against 64: compiled for MIP_MIN_WAVES_PER_SIMD = 5 waves per SIMD against 8)
// whether it runs or not, so
// the host launches <.., .., false> whenever its upload-time census found every instance finite
// (mesh-table boxes always are); a per-instance box override may be non-finite, so it implies kGeneral.
// kOrder: what a workgroup does between its arithmetic and its exit (both orders produce identical bytes).
//   1  stores first: waves 1-3 store all sixteen matrix pieces, THEN the commands are assembled (in the
//      staging area those waves have just freed: 13.4 KB of LDS per workgroup) and wave 0 resolves the
//      prefix and copies the commands out (it has no bulk store of its own). The store issue of waves
//      1-3 paces the look-up: a tile that looks at its predecessors' words early, while they are still
//      being written, slows the whole launch (profiles/r02_lookup_probes.txt); best once the launch has
//      a steady state (>= 1 M instances: 181 vs 232 us at 10 M).
//   3  commands first: assembly in an area of its own (19.5 KB), then waves 1-3 store all sixteen
//      matrix pieces while wave 0 spends that time on the prefix round trip and the copy-out.
//      Shortest dependency chain per tile; best while every tile is in the launch's first and last
//      generation of workgroups (100 k instances: 6.1 vs 6.9 us).
// kWire: 1 = the tile's commands leave in the wire form of a shard's draw list (MIP_OUT_WIRE, wire_copy_out above), 2 = in its
// packed form (MIP_OUT_WIRE_PACKED, wire_packed_copy_out; KernelArgs.wire_index_bits); 0 = 20-byte commands. A template
// parameter, not a run-time flag: as a flag it cost the plain frame 0.2 us at 100 k and at 1 M (profiles/r03_vs_r02_kbench.txt).
// The kernel's arguments: what a tile needs before it can issue its first load — the input arrays, the tables, the command
// pointer (is there a prefix at all), n and whether the mesh table has ONE entry — as LEADING SCALAR arguments, 14 dwords (a wave has 16 user SGPRs, two of them the
// kernarg segment's address: 14 can be preloaded), which the dispatcher preloads into SGPRs
// (-mllvm -amdgpu-kernarg-preload-count=16: Makefile) before the wave's first instruction; then the argument block. Inside the
// block the same values were a scalar load from the kernarg segment and a wait — a round trip in front of every tile's instance
// loads (profiles/r05_tile_head.txt). The block keeps its copies: the cold path (help_tile_aggregate) and the late uses read those.
#define MIP_FRAME_HEAD_PARAMS                                                                                              \
  const float* __restrict__ h_pos, const float4* __restrict__ h_rot, const float* __restrict__ h_scale,                    \
      const uint32_t* __restrict__ h_mesh_id, const MeshEntry* __restrict__ h_meshes, uint32_t* h_cmds, uint32_t h_n, uint32_t h_one_mesh

// kFirstMover: the launch follows the first-mover rule (KernelArgs::kFirstMoverAdds above; the host launches this instantiation
// when KernelArgs.first_mover_rule == 1). An instantiation of its own: as a run-time flag the rule's branches and the registers
// they keep alive cost the ordinary launch 0.05-0.3 us (profiles/r05_first_mover.txt).
template <bool kBoxOverride, bool kGeneral, int kOrder, int kWire = 0, bool kFirstMover = false>
__global__ __launch_bounds__(kTile, kGeneral ? MIP_MIN_WAVES_PER_SIMD : 8) void mip_instance_pipeline_kernel(MIP_FRAME_HEAD_PARAMS, const KernelArgs a) {
  static_assert(!kFirstMover || KernelArgs::kFirstMoverAdds, "this build has no first-mover rule");
  static_assert(kOrder == 1 || kOrder == 3, "unknown order");
  static_assert(!kWire || !kBoxOverride, "skinned frames do not emit the wire form");
  static_assert(kGeneral || !kBoxOverride, "a box override may be non-finite");
  __shared__ __attribute__((aligned(16))) float s_mat[kTile * 12];    // rows 0..2 of every matrix
  __shared__ uint32_t s_row3[kTile];                                     // NaN bits of row 3 + mesh id
  // the tile's commands, in order: order 1 reuses the staging area of waves 1-3 once they have stored;
  // order 3 has an area of its own
  __shared__ __attribute__((aligned(16))) uint32_t s_cmd_own[kOrder == 3 ? kTile * kCmdLdsWords : 4];
  static_assert((kTile - 64) * 12 >= kTile * kCmdLdsWords, "commands must fit the staging area of waves 1-3");
  uint32_t* const s_cmd = kOrder == 1 ? reinterpret_cast<uint32_t*>(&s_mat[64 * 12]) : s_cmd_own;
  __shared__ uint32_t s_wave_count[kWaves], s_wave_sum[kWaves];
  __shared__ unsigned long long s_vis[kWaves];
  __shared__ unsigned long long s_tile_agg;

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool want_cmds = h_cmds != nullptr;
  uint32_t tile = blockIdx.x;
#ifdef MIP_DEBUG_STAMPS
  if (a.debug_tile_mult) tile = (uint32_t)(((unsigned long long)blockIdx.x * a.debug_tile_mult + a.debug_tile_add) % a.n_tiles);
#endif
  const uint32_t tile_first = tile * kTile;
  const uint32_t i = tile_first + tid;
  const bool active = i < h_n;
  const uint32_t il = active ? i : h_n - 1u;  // keep the loads of idle lanes in bounds
  MIP_STAMP(0);
#ifdef MIP_DEBUG_STAMPS
  // fault injection (diagnostic build only): one tile never marks itself started and never publishes, so the later tiles have
  // to compute that tile's aggregate themselves (resolve_prefix) — and the launch must end with the right bytes
  const bool skip_publish = a.debug_skip_publish_tile == tile + 1u;
#else
  const bool skip_publish = false;
#endif
  unsigned long long granule_before = 0;

  // ---- loads: 36 B per instance. FIRST, from preloaded arguments: nothing is waited for in front of them. (Behind the frame
  //      below, with the pointers in the argument block, the compiler fetched `frame_ring`, waited, branched on it, and only then
  //      fetched the pointers: two scalar round trips in front of the first instance load. The planes now arrive under the loads.) ----
  const float px = h_pos[3 * (size_t)il + 0], py = h_pos[3 * (size_t)il + 1], pz = h_pos[3 * (size_t)il + 2];
  const float4 q = h_rot[il];
  const float sc = h_scale[il];
  // A mesh table of ONE entry (BASELINE configs[1]: 100 k instances of one mesh): every id is 0 (uploads with an id outside the
  // table are refused) and the entry is the same for every lane — a scalar load issued here, beside the instance loads, instead of
  // an id load and a gather BEHIND it: one dependent round trip and 4 of the 36 input bytes per instance less.
  uint32_t mesh = 0;
  float4 mb0, mb1;
  if (!h_one_mesh) {
    mesh = h_mesh_id[il];
  } else {  // (through the constant address space: a uniform address there is a scalar load; the table is not written during a launch)
    typedef const __attribute__((address_space(4))) float* ConstWords;
    ConstWords e = (ConstWords)(unsigned long long)h_meshes;
    mb0 = make_float4(e[0], e[1], e[2], e[3]);
    mb1 = make_float4(e[4], e[5], e[6], e[7]);
  }
  // ---- the frame: kernel arguments, or (recorded launches) 128 B of device memory read by the
  //      first 32 lanes of every wave and broadcast, in flight together with the instance loads ----
  float planes[24], cam[3];
  uint32_t first_instance_base, first_index_base;
  if (a.frame_ring) {
    const uint32_t word = a.frame_ring[lane & (kFrameWords - 1u)];
#pragma unroll
    for (int k = 0; k < 24; ++k) planes[k] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)word, k));
#pragma unroll
    for (int k = 0; k < 3; ++k) cam[k] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)word, 24 + k));
    first_instance_base = (uint32_t)__builtin_amdgcn_readlane((int)word, 27);
    first_index_base = (uint32_t)__builtin_amdgcn_readlane((int)word, 28);
  } else {
#pragma unroll
    for (int k = 0; k < 24; ++k) planes[k] = a.planes[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) cam[k] = a.cam[k];
    first_instance_base = a.first_instance_base;
    first_index_base = a.first_index_base;
  }

  // a predecessor that does not publish is not waited for: this wave computes its aggregate itself (resolve_prefix)
  auto help = [lane](uint32_t u) { return help_tile_aggregate<kBoxOverride, kGeneral>(u, lane); };

  // the LDS word the waves add their aggregates to; the barrier does not wait for the instance
  // loads above, and every wave of the workgroup has only just started
  if (want_cmds) {
    if (tid == 0) s_tile_agg = 0ull;
    __syncthreads();
  }
  // (a copy of small mesh tables in LDS was measured: no gain — profiles/r02_lds_pad_occupancy_and_mesh_cache_ab.txt)
  if (!h_one_mesh) {
    mb0 = *reinterpret_cast<const float4*>(&h_meshes[mesh].min_x);
    mb1 = *reinterpret_cast<const float4*>(&h_meshes[mesh].max_x);
  }
  // this tile is running (launches that follow the first-mover rule). Wherever in the tile's head the swap is issued — first of
  // all, behind the instance loads, behind the mesh-table gather — it costs the launch 0.3 us: loads and returning atomics come
  // back in the order they were issued, and this one takes longer than a load (profiles/r05_first_mover.txt)
  constexpr bool first_mover_rule = kFirstMover;
  const bool marks = first_mover_rule && want_cmds && tid == 63u && !skip_publish;  // (a lane that publishes for wave 0)
  if constexpr (first_mover_rule)
    if (want_cmds) granule_before = mark_tile_started_issue(a, tile, marks);
  MeshEntry mb;
  mb.min_x = mb0.x; mb.min_y = mb0.y; mb.min_z = mb0.z; mb.len0 = __float_as_uint(mb0.w);
  mb.max_x = mb1.x; mb.max_y = mb1.y; mb.max_z = mb1.z; mb.len1 = __float_as_uint(mb1.w);

  // ---- model matrix + world AABB ----
  float r[3][3];
  quat_to_rotation(q.x, q.y, q.z, q.w, r);
  Instance inst;
  instance_tiered<kBoxOverride, kGeneral>(a, il, r, px, py, pz, sc, mb, inst);

  MIP_STAMP(1);
  // The emitting lanes need their mesh's vertex_offset. Its gather used to sit behind the keep decision, i.e. one L2
  // round trip inside the tile's dependency chain (barrier -> gather -> assembly -> barrier -> copy-out); issued here
  // it returns under the plane tests. (The two source index offsets of the per-triangle stage stay a late, conditional
  // 16-byte gather: only those frames read them.)
  // Only in the commands-first order (small launches, latency-bound: 100 k 5.83 -> 5.52 us, 200 k 7.04 -> 6.82): a gather
  // by every lane instead of the emitting ones costs the stores-first order 0.5 us at 1 M (profiles/r03_vertex_offset_early_ab.txt).
  int32_t vertex_offset_of_mesh = 0;
  if constexpr (kOrder == 3)
    if (want_cmds) vertex_offset_of_mesh = a.mesh_draw[mesh].vertex_offset;
  // ---- frustum test, LOD, command length ----
  const bool culled = coarse_culled(inst, planes);
  const bool visible = active && !culled;
  const bool far_lod = lod_is_far(cam, px, py, pz);
  const uint32_t len = far_lod ? mb.len1 : mb.len0;  // len1/offset1 already fall back to LOD 0
  const bool keep = visible && len > 0u;  // compact_draw_stream.comp:41 `indexCount > 0`
  const uint32_t len_vis = visible ? len : 0u;

  // ---- wave-level compaction offsets; the last wave to get here publishes the tile's aggregate ----
  const unsigned long long keep_mask = __ballot(keep);
  const unsigned long long vis_mask = __ballot(visible);
  const uint32_t rank_in_wave = lanes_below(keep_mask);
  const uint32_t incl_sum = wave_inclusive_scan(len_vis);
  if (want_cmds && lane == 63u) {
    const uint32_t wc = (uint32_t)__popcll(keep_mask);
    s_wave_count[wave] = wc;
    s_wave_sum[wave] = incl_sum;
    // (wave 0 brings what mark_tile_started answered: whichever wave arrives last publishes)
    const bool helped_first = marks && mark_tile_started_answer(a, granule_before);
    const unsigned long long mine = ((unsigned long long)incl_sum << 32) | (1ull << kAggArrivalShift) | (helped_first ? kAggHelpedFirst : 0u) | wc;
    const unsigned long long all = __hip_atomic_fetch_add(&s_tile_agg, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + mine;
    // Published before any bulk store of this wave and without waiting for a barrier: the
    // successors' look-ups depend on it, nothing else does.
    bool publish = ((uint32_t)all >> kAggArrivalShift) == kWaves && !skip_publish;
    if (publish) publish_aggregate(a, tile, (uint32_t)all & 0xffffu, (uint32_t)(all >> 32), !((uint32_t)all & kAggHelpedFirst));
  }
  // ---- stage the matrix rows for the transposed store (three conflict-free ds_write_b128, 48-B pitch) ----
  if (a.model || a.tlas_instances) {
    float4* dst = reinterpret_cast<float4*>(&s_mat[tid * 12]);
    dst[0] = make_float4(inst.m[0], inst.m[1], inst.m[2], inst.m[3]);
    dst[1] = make_float4(inst.m[4], inst.m[5], inst.m[6], inst.m[7]);
    dst[2] = make_float4(inst.m[8], inst.m[9], inst.m[10], inst.m[11]);
    s_row3[tid] = inst.row3 | (mesh << 4);  // NaN bits of row 3 + the mesh id (for the TLAS rows)
  }
  if (lane == 0u) s_vis[wave] = vis_mask;

  // One 1-KiB piece = 16 staged matrices -> one store instruction of a wave (64 B per matrix,
  // lane-contiguous 16-B stores). Piece p of the tile covers instances tile_first + 16 p ...
  // descriptors of this tile's 16 KiB of matrices (and TLAS rows): bounded at the last instance
  const uint32_t tile_bytes = (a.n - tile_first < kTile ? a.n - tile_first : kTile) * 64u;
  const __amdgpu_buffer_rsrc_t d_model = stream_descriptor(a.model ? a.model + (size_t)tile_first * 4 : nullptr, a.model ? tile_bytes : 0u);
  const __amdgpu_buffer_rsrc_t d_tlas = stream_descriptor(a.tlas_instances ? a.tlas_instances + (size_t)tile_first * 4 : nullptr, a.tlas_instances ? tile_bytes : 0u);
  auto store_piece = [&](uint32_t p) {
    const uint32_t local = 16u * p + (lane >> 2);  // matrix within the tile
    const uint32_t col = lane & 3u;
    const float* src = &s_mat[local * 12u];
    const bool in_range = tile_first + local < a.n;
    if (a.model) {
      float w = (col == 3u) ? 1.0f : 0.0f;
      if constexpr (kGeneral) {
        const uint32_t bits = s_row3[local] & 15u;
        if ((bits >> col) & 1u) w = __uint_as_float(0x7fc00000u);
      }
      store_stream16(d_model, (64u * p + lane) * 16u, make_float4(src[3u * col], src[3u * col + 1u], src[3u * col + 2u], w));
    }
    // optional TLAS instance rows (acceleration_strucures.rs:419-451), same transposed store:
    // VkAccelerationStructureInstanceKHR = { 3x4 row-major transform = rows 0..2 of M,
    //   instanceCustomIndex:24 = draw_index | mask:8 = 0xFF, sbtOffset:24 = 0 | flags:8 =
    //   TRIANGLE_FACING_CULL_DISABLE, BLAS device address }, for EVERY instance (visible or not).
    if (a.tlas_instances) {
      const uint32_t draw = tile_first + local;
      uint4 v;
      if (col < 3u) {  // row `col`: one element of each staged column
        v = make_uint4(__float_as_uint(src[col]), __float_as_uint(src[col + 3u]), __float_as_uint(src[col + 6u]), __float_as_uint(src[col + 9u]));
      } else {
        const uint32_t mesh_of = s_row3[local] >> 4;
        const unsigned long long blas = (in_range && a.blas_address) ? a.blas_address[mesh_of] : 0ull;
        v = make_uint4(((first_instance_base + draw) & 0xffffffu) | 0xff000000u, 0x01000000u,
                       (uint32_t)blas, (uint32_t)(blas >> 32));
      }
      store_stream16(d_tlas, (64u * p + lane) * 16u, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
    }
  };
  // optional world AABB (mins, maxs) as the ECS component holds it: 24 B per lane
  auto store_aabb = [&]() {
    if (a.world_aabb && active) {
      float2* o2 = reinterpret_cast<float2*>(a.world_aabb + (size_t)i * 6);
      o2[0] = make_float2(inst.mins[0], inst.mins[1]);
      o2[1] = make_float2(inst.mins[2], inst.maxs[0]);
      o2[2] = make_float2(inst.maxs[1], inst.maxs[2]);
    }
  };
  // visibility bitmap: the tile's eight words, one store instruction
  auto store_bitmap = [&]() {
    if (a.bitmap && lane < 2u * kWaves) {
      const uint32_t word = (tile_first >> 5) + lane;
      if (word < a.bitmap_words) a.bitmap[word] = (uint32_t)(s_vis[lane >> 1] >> (32u * (lane & 1u)));
    }
  };

  if constexpr (kOrder == 1) {
    auto own_stores = [&]() {
#pragma unroll
      for (uint32_t p = 0; p < 4; ++p) store_piece(wave * 4u + p);
      if (a.bitmap && lane < 2u) {
        const uint32_t word = (tile_first >> 5) + wave * 2u + lane;
        if (word < a.bitmap_words) a.bitmap[word] = (uint32_t)(vis_mask >> (32u * lane));
      }
      store_aabb();
    };
    __syncthreads();
    if (!want_cmds) { own_stores(); return; }
    uint32_t wave_off_count = 0, wave_off_sum = 0, tile_count = 0, tile_sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) {
      const uint32_t wc = s_wave_count[w], ws = s_wave_sum[w];
      if (w < wave) { wave_off_count += wc; wave_off_sum += ws; }
      tile_count += wc;
      tile_sum += ws;
    }
    MIP_STAMP(2);
    if (wave != 0) {  // waves 1-3 store all sixteen pieces (wave 0's too): wave 0 has nothing left to store after its copy-out
      const uint32_t p0 = store_run_first(wave), p1 = store_run_first(wave + 1u);  // 6 + 5 + 5 pieces
      for (uint32_t p = p0; p < p1; ++p) store_piece(p);
      if (wave == 1) store_bitmap();
      store_aabb();
    }
    __syncthreads();  // waves 1-3 have read their staged matrices: their area is free for the commands
    if (keep) {
      uint32_t* c = &s_cmd[(wave_off_count + rank_in_wave) * kCmdLdsWords];
      vertex_offset_of_mesh = a.mesh_draw[mesh].vertex_offset;  // stores-first order: gathered by the emitting lanes only
      c[0] = len; c[1] = 1u; c[2] = wave_off_sum + (incl_sum - len_vis); c[3] = (uint32_t)vertex_offset_of_mesh; c[4] = first_instance_base + i;
      if constexpr (kWire) c[5] = mesh | (far_lod ? 0x80000000u : 0u);
      else if (a.src_index_offset) c[5] = far_lod ? a.mesh_draw[mesh].src_offset1 : a.mesh_draw[mesh].src_offset0;
    }
    __syncthreads();
    MIP_STAMP(3);
    if (wave != 0) return;
    store_aabb();  // wave 0's own (optional) boxes go out before the look-up: nothing of the instance lives across it
    uint32_t base_count = 0, base_sum = 0;
    if (tile > 0) resolve_prefix(a, tile, lane, base_count, base_sum, help, first_mover_rule);
    else note_helps_for_the_host(a, lane);
    if (lane == 0 && tile == a.n_tiles - 1u) {
      *a.draw_count = base_count + tile_count;
      if (a.index_total) *a.index_total = base_sum + tile_sum;
    }
    MIP_STAMP(4);
    const uint32_t first_index_add = base_sum + first_index_base;
    if constexpr (kWire) {
      if constexpr (kWire == 2) wire_packed_copy_out(a.cmds, s_cmd, lane, base_count, first_index_add, tile_count, first_instance_base, a.wire_index_bits);
      else wire_copy_out(a.cmds, s_cmd, lane, base_count, first_index_add, tile_count);
      MIP_STAMP(5);
      return;
    }
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
    MIP_STAMP(5);
    return;
  }
  if (!want_cmds) {  // no compaction: every wave stores its own quarter of the tile
    __syncthreads();
#pragma unroll
    for (uint32_t p = 0; p < 4; ++p) store_piece(wave * 4u + p);
    if (wave == 0) store_bitmap();
    store_aabb();
    return;
  }

  __syncthreads();  // the four wave aggregates are in LDS
  MIP_STAMP(2);
  uint32_t wave_off_count = 0, wave_off_sum = 0, tile_count = 0, tile_sum = 0;
#pragma unroll
  for (uint32_t w = 0; w < kWaves; ++w) {
    const uint32_t wc = s_wave_count[w], ws = s_wave_sum[w];
    if (w < wave) { wave_off_count += wc; wave_off_sum += ws; }
    tile_count += wc;
    tile_sum += ws;
  }

  // ---- tile-local command assembly in LDS (firstIndex still relative to the tile) ----
  if (keep) {
    uint32_t* c = &s_cmd[(wave_off_count + rank_in_wave) * kCmdLdsWords];
    c[0] = len;                                               // indexCount
    c[1] = 1u;                                                // instanceCount, generate_work.comp:63
    c[2] = wave_off_sum + (incl_sum - len_vis);               // firstIndex (tile-relative)
    c[3] = (uint32_t)vertex_offset_of_mesh;                   // vertexOffset, :66
    c[4] = first_instance_base + i;                         // firstInstance = draw_index, :64
    if constexpr (kWire) c[5] = mesh | (far_lod ? 0x80000000u : 0u);  // wire form: the record's second word
    else if (a.src_index_offset)                              // push constant indexOffset, cull_pipeline.rs:552 (per-triangle stage only)
      c[5] = far_lod ? a.mesh_draw[mesh].src_offset1 : a.mesh_draw[mesh].src_offset0;
  }
  __syncthreads();  // commands, staged matrices and visibility words of every wave are in LDS
  MIP_STAMP(3);

  // Waves 1-3 put the tile's bulk bytes on their way — all sixteen matrix pieces, wave 0's
  // included — and exit; their store instructions stall for microseconds once the chip's write
  // path is saturated. Wave 0 issues no bulk store: it spends that time on the one memory round trip
  // of the tile's prefix and then copies the commands out.
  if (wave != 0) {
    // contiguous runs of 6 + 5 + 5 KiB
    const uint32_t p0 = store_run_first(wave), p1 = store_run_first(wave + 1u);
    for (uint32_t p = p0; p < p1; ++p) store_piece(p);
    if (wave == 1) store_bitmap();
    store_aabb();
    return;
  }

  // ---- exclusive prefix over the earlier tiles ----
  store_aabb();  // (optional output) before the look-up: nothing of the instance lives across it
  uint32_t base_count = 0, base_sum = 0;
#ifndef MIP_EXP_NO_HOP  // tuning builds only: what the kernel costs without the cross-tile look-up (results are wrong)
  if (tile > 0) resolve_prefix(a, tile, lane, base_count, base_sum, help, first_mover_rule);
  else note_helps_for_the_host(a, lane);
#else
  base_count = tile * 64u;
#ifdef MIP_EXP_FAKE_DELAY  // idle for the time a look-up takes, without its memory traffic
  if (tile >= a.delay_first && tile < a.delay_last) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < MIP_EXP_FAKE_DELAY) __builtin_amdgcn_s_sleep(8);
  }
#endif
#endif
  if (lane == 0 && tile == a.n_tiles - 1u) {
    *a.draw_count = base_count + tile_count;
    if (a.index_total) *a.index_total = base_sum + tile_sum;
  }
  MIP_STAMP(4);

  // ---- coalesced copy-out of the tile's commands ----
  const uint32_t first_index_add = base_sum + first_index_base;
  if constexpr (kWire) {
    if constexpr (kWire == 2) wire_packed_copy_out(a.cmds, s_cmd, lane, base_count, first_index_add, tile_count, first_instance_base, a.wire_index_bits);
    else wire_copy_out(a.cmds, s_cmd, lane, base_count, first_index_add, tile_count);
    MIP_STAMP(5);
    return;
  }
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
  MIP_STAMP(5);
}

}  // namespace mip
