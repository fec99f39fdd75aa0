// light_lists_kernel.hpp — row f-4: the shadow pass's per-light draw lists (gfx950).
#pragma once

#include "instance_kernel.hpp"
#include "stage_args.hpp"

#pragma clang fp contract(off)

namespace mip {

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
#if !defined(MIP_EXP_LIGHT_NO_NT) && !defined(MIP_EXP_LIGHT_PTR_NT)  // sc1 nt: 1 M x 16 lights 61.4 -> 56.7 us against the pointer form's nt
      const __amdgpu_buffer_rsrc_t d_dst = stream_descriptor(dst, words * 4u);
#endif
      for (uint32_t q = tid; q * 4u < words; q += kTile) {
        const uint4 v = reinterpret_cast<const uint4*>(row)[q];
#if defined(MIP_EXP_LIGHT_NO_NT)
        reinterpret_cast<uint4*>(dst)[q] = v;
#elif defined(MIP_EXP_LIGHT_PTR_NT)
        store_stream16(reinterpret_cast<float4*>(dst) + q, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
#else  // written once, lane-contiguous 16-B stores: streamed like the matrices of the instance kernel
        store_stream16(d_dst, q * 16u, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
#endif
      }
    } else {
      for (uint32_t w = tid; w < words; w += kTile) dst[w] = row[w];
    }
  }
}

}  // namespace mip
