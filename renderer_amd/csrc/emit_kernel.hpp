// emit_kernel.hpp — last launch of a large ordered-tiles frame: the compacted command list from the visibility bitmap (gfx950).
#pragma once

#include "instance_kernel.hpp"
#include "stage_args.hpp"

#pragma clang fp contract(off)

namespace mip {

// A frame in ordered-tiles mode (MIP_CFG_ORDERED_TILES, or after a stalled frame) must not depend on the order workgroups
// start in. Launches above 160 tiles do that with two or three launches none of which waits for another workgroup
// (instance_kernel.hpp, "the prefix without any wait"):
//   1  the frame kernel WITHOUT commands (KernelArgs.tile_agg_out): matrices, world boxes, TLAS rows, the visibility bitmap, and
//      per tile the pair {emitted commands, sum of index_len};
//   2  (above kEmitSelfPrefixTiles tiles only) mip_tile_scan_kernel: the pairs summed per group of kTileGroup tiles, exclusive
//      prefixes of the group sums, and the totals (draw_count, index total);
//   3  this kernel: per tile, the commands of compact_draw_stream.comp / cull_pass (cull_pipeline.rs:534-577) again from what
//      launch 1 left — visible(i) is bit i of the bitmap, the LOD is pick_lod of the same positions (helpers.rs:3-11, the
//      frame kernel's expression), index_len / vertex_offset come from the mesh table — at the positions the prefix gives.
// 16 B read per instance + the bitmap, 20 B (or a wire record) written per command; no cross-workgroup communication at all.
template <int kWire>
__global__ __launch_bounds__(kTile) void mip_emit_commands_kernel(const EmitArgs a) {
  __shared__ uint32_t s_cmd[kTile * kCmdLdsWords];
  __shared__ uint32_t s_wave_count[kWaves], s_wave_sum[kWaves];
  __shared__ uint32_t s_pre_count[kWaves], s_pre_sum[kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t tile = blockIdx.x;
  const uint32_t tile_first = tile * kTile;
  const uint32_t i = tile_first + tid;
  const bool active = i < a.n;
  const uint32_t il = active ? i : a.n - 1u;
  // The tile's exclusive prefix. Launches of up to kEmitSelfPrefixTiles tiles have no launch 2: the workgroup sums the pairs of
  // ALL earlier tiles itself (<= 16 coalesced 8-byte loads per thread, in flight together with the instance loads below; 61 MB
  // of L2 reads over the launch at 1 M instances — cheaper than a 4.8 us launch of one workgroup in between). Larger launches:
  // the group's prefix (launch 2) + the pairs of the earlier tiles of the own group, every wave for itself.
  uint2 pre = make_uint2(0u, 0u);
  const bool self_prefix = a.group_prefix == nullptr;
  if (self_prefix) {
    uint32_t c = 0, s = 0;
#pragma unroll 4
    for (uint32_t t = tid; t < tile; t += kTile) {
      const uint2 v = a.tile_agg[t];
      c += v.x;
      s += v.y;
    }
    c = wave_sum(c);
    s = wave_sum(s);
    if (lane == 0u) { s_pre_count[wave] = c; s_pre_sum[wave] = s; }  // read behind the barrier below
  } else {
    const uint32_t group = tile / kTileGroup, in_group = tile % kTileGroup;
    pre = a.group_prefix[group];  // wave-uniform: scalar loads
    const uint2 mine = lane < in_group ? a.tile_agg[group * kTileGroup + lane] : make_uint2(0u, 0u);
    pre.x += wave_sum(mine.x);
    pre.y += wave_sum(mine.y);
  }

  const uint32_t word = active ? a.bitmap[i >> 5] : 0u;
  const bool visible = ((word >> (i & 31u)) & 1u) != 0u;
  // culled lanes load nothing: in a scene whose instance order has any spatial coherence most 64-byte lines of the position and
  // mesh-id columns are never touched here (random order, v = 0.27: every line is, and the launch is 2 % faster at 1 M, equal at 10 M)
  float px = 0.f, py = 0.f, pz = 0.f;
  uint32_t mesh = 0, len0 = 0, len1 = 0;
  uint4 md = make_uint4(0u, 0u, 0u, 0u);  // vertex_offset, src_offset0, src_offset1, -
  if (visible) {
    px = a.pos[3 * (size_t)il + 0]; py = a.pos[3 * (size_t)il + 1]; pz = a.pos[3 * (size_t)il + 2];
    mesh = a.mesh_id[il];
    len0 = a.meshes[mesh].len0; len1 = a.meshes[mesh].len1;
    md = *reinterpret_cast<const uint4*>(&a.mesh_draw[mesh]);
  }
  const float dx = a.cam[0] - px, dy = a.cam[1] - py, dz = a.cam[2] - pz;
  const float dist_sq = dx * dx + dy * dy + dz * dz;
  const bool far_lod = dist_sq > kLodDistSqThreshold;
  const uint32_t len = far_lod ? len1 : len0;
  const bool keep = visible && len > 0u;  // compact_draw_stream.comp:41 `indexCount > 0`
  const uint32_t len_vis = visible ? len : 0u;

  const unsigned long long keep_mask = __ballot(keep);
  const uint32_t rank = lanes_below(keep_mask);
  const uint32_t incl = wave_inclusive_scan(len_vis);
  if (lane == 63u) {
    s_wave_count[wave] = (uint32_t)__popcll(keep_mask);
    s_wave_sum[wave] = incl;
  }
  __syncthreads();
  uint32_t off_count = 0, off_sum = 0, tile_count = 0, tile_sum = 0;
#pragma unroll
  for (uint32_t w = 0; w < kWaves; ++w) {
    if (w < wave) { off_count += s_wave_count[w]; off_sum += s_wave_sum[w]; }
    tile_count += s_wave_count[w];
    tile_sum += s_wave_sum[w];
  }
  if (self_prefix) {
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) { pre.x += s_pre_count[w]; pre.y += s_pre_sum[w]; }
    if (tid == 0u && tile == a.n_tiles - 1u) {  // the last tile knows the totals
      *a.draw_count = pre.x + tile_count;
      if (a.index_total) *a.index_total = pre.y + tile_sum;
    }
  }
  if (keep) {
    uint32_t* c = &s_cmd[(off_count + rank) * kCmdLdsWords];
    c[0] = len; c[1] = 1u; c[2] = off_sum + (incl - len_vis); c[3] = md.x; c[4] = a.first_instance_base + i;
    if constexpr (kWire != 0) c[5] = mesh | (far_lod ? 0x80000000u : 0u);
    else c[5] = far_lod ? md.z : md.y;
  }
  __syncthreads();

  // ---- copy-out by the whole workgroup (nobody has anything to wait for) ----
  const uint32_t base_count = pre.x;
  const uint32_t first_index_add = pre.y + a.first_index_base;
  if constexpr (kWire == 1) {
    for (uint32_t k = tid; k < tile_count; k += kTile) {
      const uint32_t g = base_count + k, block = g / kWireBlockCmds, slot = g % kWireBlockCmds;
      uint32_t* b = a.cmds + (size_t)block * kWireBlockWords;
      const uint32_t* c = &s_cmd[k * kCmdLdsWords];
      *reinterpret_cast<uint2*>(b + kWireBlockHeaderWords + 2u * slot) = make_uint2(c[4], c[5]);
      if (slot == 0u) *reinterpret_cast<uint4*>(b) = make_uint4(c[2] + first_index_add, 0u, 0u, 0u);
    }
  } else if constexpr (kWire == 2) {
    for (uint32_t k = tid; k < tile_count; k += kTile) {
      const uint32_t g = base_count + k, block = g / kWireBlockCmds, slot = g % kWireBlockCmds;
      uint32_t* b = a.cmds + (size_t)block * kWirePackedBlockWords;
      const uint32_t* c = &s_cmd[k * kCmdLdsWords];
      b[kWireBlockHeaderWords + slot] = (c[4] - a.first_instance_base) | ((c[5] & 0x7fffffffu) << a.wire_index_bits) | (c[5] & 0x80000000u);
      if (slot == 0u) *reinterpret_cast<uint4*>(b) = make_uint4(c[2] + first_index_add, a.first_instance_base, a.wire_index_bits, 0u);
    }
  } else {
    uint32_t* out = a.cmds + (size_t)base_count * kCmdWords;
    const uint32_t words = tile_count * kCmdWords;
    for (uint32_t j = tid; j < words; j += kTile) {
      const uint32_t k = j / kCmdWords, f = j - k * kCmdWords;
      uint32_t v = s_cmd[k * kCmdLdsWords + f];
      if (f == 2u) v += first_index_add;
      out[j] = v;
    }
    if (a.src_index_offset)
      for (uint32_t k = tid; k < tile_count; k += kTile) a.src_index_offset[base_count + k] = s_cmd[k * kCmdLdsWords + 5u];
  }
}

}  // namespace mip
