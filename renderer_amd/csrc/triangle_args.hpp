// triangle_args.hpp — row f-1: argument blocks of the per-triangle stage and the host-side launchers of its
// kernels. The kernels themselves (triangle_kernels.hpp) are compiled in a translation unit of their own
// (triangle_tu.hip) with -fno-slp-vectorize: under plain -O3 the SLP vectoriser packs the two mat4*vec4 per
// vertex into v_pk_mul_f32 / v_pk_add_f32 and pays for it with one v_mov_b32 per operand pair (115 moves beside
// 136 packed operations per 64-triangle step in round 2's ISA) — on gfx950 a packed f32 instruction takes twice the
// issue time of a plain one, so packing buys nothing and the moves are pure cost. mip_api.hip sees only this header.
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

namespace mip {

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

struct RecompactArgs {
  const uint32_t* in_cmds;
  const uint32_t* in_count;
  uint32_t* out_cmds;
  uint32_t* out_count;
};

struct RecompactWideArgs {
  const uint32_t* in_cmds;
  const uint32_t* in_count;
  uint32_t* out_cmds;
  uint32_t* out_count;
  uint32_t* block_base;   // one word per 1024 commands: survivors in the block, then their exclusive prefix
  uint32_t n_blocks;
};

constexpr uint32_t kTriParts = 16;
constexpr uint32_t kTriPartMaxT = 8;  // triangles per thread and part: commands up to 16 * 256 * 8 = 32 768 triangles

struct TrianglePartsArgs {
  TriangleArgs t;
  unsigned long long* part_status;  // [commands][kTriParts] granules {epoch : 32 | survivors : 32}
  uint32_t epoch;                   // unique per launch on this frame slot, never 0
};

// Launchers (defined in triangle_tu.hip). Each enqueues one kernel on `stream`; errors surface through hipGetLastError.
void launch_triangle_cull_waves(uint32_t blocks, hipStream_t stream, const TriangleArgs& a);
void launch_triangle_cull_block(uint32_t threads /* 256 | 512 | 1024 */, uint32_t blocks, hipStream_t stream, const TriangleArgs& a);
void launch_triangle_cull_parts(uint32_t blocks, hipStream_t stream, const TrianglePartsArgs& a);
void launch_recompact(hipStream_t stream, const RecompactArgs& a);
void launch_recompact_wide(hipStream_t stream, const RecompactWideArgs& a);  // count, scan, scatter

}  // namespace mip
