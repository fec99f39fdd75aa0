// merge_kernel.hpp — multi-GPU: merge of the all-gathered shard draw lists (gfx950).
#pragma once

#include "instance_kernel.hpp"

#pragma clang fp contract(off)

namespace mip {

// ---------------------------------------------------------------------------------------
// shard merge (multi-GPU): concatenate all-gathered chunks, rebasing firstIndex
// ---------------------------------------------------------------------------------------

struct MergeArgs {
  const unsigned char* chunks;
  unsigned long long stride;
  uint32_t n_chunks;
  uint32_t capacity;    // commands a chunk may carry (what the caller sized out_cmds for: n_chunks x capacity)
  uint32_t* out_cmds;
  uint32_t* out_count;  // [0] = commands, [1] = indices
  uint32_t* error_flag; // host-mapped
};

constexpr uint32_t kMaxMergeChunks = 64;

__global__ __launch_bounds__(256) void mip_merge_draw_lists_kernel(const MergeArgs a) {
  __shared__ uint32_t s_count_base[kMaxMergeChunks + 1], s_index_base[kMaxMergeChunks + 1];
  if (threadIdx.x == 0) {
    uint32_t c = 0, s = 0;
    // never more than the stride physically holds (the stride is rounded up to 256 B, so it may hold
    // a few commands more than the capacity the caller sized its output for)
    const uint32_t fits = (uint32_t)((a.stride - 32u) / (kCmdWords * 4u));
    const uint32_t capacity = a.capacity < fits ? a.capacity : fits;
    for (uint32_t k = 0; k < a.n_chunks; ++k) {
      const uint32_t* h = reinterpret_cast<const uint32_t*>(a.chunks + k * a.stride);
      uint32_t count = h[0];
      if (count > capacity) {  // the shard emitted more than the exchanged chunk holds
        count = capacity;
        if (blockIdx.x == 0) raise_error(a.error_flag, kErrChunkOverflow);
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
// the same merge over chunks in the WIRE form (MIP_OUT_WIRE; instance_kernel.hpp wire_copy_out)
// ---------------------------------------------------------------------------------------
// chunk = [32-B header {count, index total} | blocks of {16-B header: firstIndex of the block's first command |
// 256 x {firstInstance, mesh | lod << 31}}]. One workgroup expands one block at a time: each thread one record ->
// indexCount / vertexOffset from the mesh table, firstIndex = block header + the index_len of the records in front
// of it in the block (DPP scan per wave + the wave totals in LDS) + the index totals of the earlier chunks; the 256
// commands leave through LDS so that the 20-byte records are written as contiguous dwords.

struct MergeWireArgs {
  const unsigned char* chunks;
  unsigned long long stride;
  uint32_t n_chunks;
  uint32_t capacity;
  uint32_t* out_cmds;
  uint32_t* out_count;
  uint32_t* error_flag;
  const MeshEntry* meshes;
  const MeshDraw* mesh_draw;
  uint32_t n_meshes;
};

// kPacked: the chunks are in the packed wire form (MIP_OUT_WIRE_PACKED): one 32-bit record per command,
// instance index | mesh << index_bits | lod << 31, block header {firstIndex, first_instance_base, index_bits, 0}.
template <bool kPacked>
__global__ __launch_bounds__(256) void mip_merge_wire_lists_kernel(const MergeWireArgs a) {
  static_assert(kWireBlockCmds == 256, "one thread per record of a block");
  constexpr uint32_t kBlockWords = kPacked ? kWirePackedBlockWords : kWireBlockWords;
  __shared__ uint32_t s_count_base[kMaxMergeChunks + 1], s_index_base[kMaxMergeChunks + 1], s_block_base[kMaxMergeChunks + 1];
  __shared__ uint32_t s_wave_total[4];
  __shared__ uint32_t s_out[kWireBlockCmds * kCmdWords];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // chunk tables: lane k of wave 0 reads header k (one round trip for all <= 64 chunks, not one per chunk), three wave scans
  if (wave == 0) {
    const uint32_t fits = (uint32_t)((a.stride - 32u) / (kBlockWords * 4u)) * kWireBlockCmds;
    const uint32_t capacity = a.capacity < fits ? a.capacity : fits;
    uint32_t count = 0, total = 0;
    if (lane < a.n_chunks) {
      const uint2 h = *reinterpret_cast<const uint2*>(a.chunks + lane * a.stride);
      count = h.x;
      total = h.y;
    }
    if (__any(count > capacity) && blockIdx.x == 0 && lane == 0) raise_error(a.error_flag, kErrChunkOverflow);  // a shard emitted more than the exchanged chunk holds
    count = count > capacity ? capacity : count;
    const uint32_t blocks = (count + kWireBlockCmds - 1u) / kWireBlockCmds;
    const uint32_t c_incl = wave_inclusive_scan(count), s_incl = wave_inclusive_scan(total), b_incl = wave_inclusive_scan(blocks);
    if (lane < a.n_chunks) {
      s_count_base[lane] = c_incl - count;
      s_index_base[lane] = s_incl - total;
      s_block_base[lane] = b_incl - blocks;
    }
    if (lane == 63u) {  // lanes past the last chunk contributed zeros: lane 63 holds the totals
      s_count_base[a.n_chunks] = c_incl;
      s_index_base[a.n_chunks] = s_incl;
      s_block_base[a.n_chunks] = b_incl;
      if (blockIdx.x == 0) {
        a.out_count[0] = c_incl;
        a.out_count[1] = s_incl;
      }
    }
  }
  __syncthreads();
  const uint32_t total_blocks = s_block_base[a.n_chunks];
  // where block `blk` of the merged list lives: its chunk, its number inside the chunk, its live records, its words
  uint32_t chunk = 0;
  auto locate = [&](uint32_t blk, uint32_t& b, uint32_t& in_block, const uint32_t*& body) {
    while (blk >= s_block_base[chunk + 1]) ++chunk;  // blk only grows
    b = blk - s_block_base[chunk];
    const uint32_t chunk_count = s_count_base[chunk + 1] - s_count_base[chunk];
    in_block = chunk_count - b * kWireBlockCmds < kWireBlockCmds ? chunk_count - b * kWireBlockCmds : kWireBlockCmds;
    body = reinterpret_cast<const uint32_t*>(a.chunks + chunk * a.stride + 32) + (size_t)b * kBlockWords;
  };
  // Software pipeline, two deep: while block i is expanded, the table entries of block i + 1 (gathers that depend on its
  // records) and the records + header of block i + 2 are in flight.
  struct Located { uint32_t b, in_block, chunk, first_index; uint2 rec; };
  auto fetch_records = [&](uint32_t blk, Located& l) {
    const uint32_t* body = nullptr;
    l.rec = make_uint2(0u, 0u);
    l.in_block = 0u;
    l.first_index = 0u;
    if (blk < total_blocks) {
      locate(blk, l.b, l.in_block, body);
      l.chunk = chunk;
      if constexpr (kPacked) {
        // the record's words, unpacked: {firstInstance, mesh | lod << 31} as the 8-byte form carries them
        const uint4 h = *reinterpret_cast<const uint4*>(body);  // wave-uniform address
        uint32_t bits = h.z;
        if (bits > 31u) {  // a corrupt header must not become an undefined shift
          if (tid == 0) raise_error(a.error_flag, kErrWireRecord);
          bits = 31u;
        }
        if (tid < l.in_block) {
          const uint32_t r = body[kWireBlockHeaderWords + tid];
          const uint32_t low = r & 0x7fffffffu;
          l.rec = make_uint2(h.y + (low & ((1u << bits) - 1u)), (low >> bits) | (r & 0x80000000u));
        }
        l.first_index = h.x;
      } else {
        if (tid < l.in_block) l.rec = *reinterpret_cast<const uint2*>(body + kWireBlockHeaderWords + 2u * tid);
        l.first_index = body[0];
      }
    }
  };
  auto fetch_table = [&](const Located& l, uint32_t& len, int32_t& vertex_offset) {
    uint32_t mesh = l.rec.y & 0x7fffffffu;
    const bool valid = tid < l.in_block;
    if (mesh >= a.n_meshes) {  // never follow a corrupt record out of the table
      if (valid) raise_error(a.error_flag, kErrWireRecord);
      mesh = 0u;
    }
    len = 0u;
    vertex_offset = 0;
    if (valid && a.n_meshes) {
      len = (l.rec.y >> 31) ? a.meshes[mesh].len1 : a.meshes[mesh].len0;
      vertex_offset = a.mesh_draw[mesh].vertex_offset;
    }
  };
  Located cur{}, nxt{}, nxt2{};
  uint32_t cur_len = 0, nxt_len = 0;
  int32_t cur_vo = 0, nxt_vo = 0;
  fetch_records(blockIdx.x, cur);
  fetch_records(blockIdx.x + gridDim.x, nxt);
  fetch_table(cur, cur_len, cur_vo);
  for (uint32_t blk = blockIdx.x; blk < total_blocks; blk += gridDim.x) {
    fetch_records(blk + 2u * gridDim.x, nxt2);   // block i + 2: records
    fetch_table(nxt, nxt_len, nxt_vo);           // block i + 1: table entries
    const uint32_t b = cur.b, in_block = cur.in_block, this_chunk = cur.chunk, first_index = cur.first_index;
    const uint2 rec = cur.rec;
    const bool valid = tid < in_block;
    const uint32_t len = cur_len;
    const int32_t vertex_offset = cur_vo;
    cur = nxt; nxt = nxt2;
    cur_len = nxt_len; cur_vo = nxt_vo;
    const uint32_t incl = wave_inclusive_scan(len);
    if (lane == 63u) s_wave_total[wave] = incl;
    __syncthreads();  // wave totals in; also: the previous block's copy-out has read s_out
    uint32_t before = first_index + s_index_base[this_chunk] + (incl - len);
#pragma unroll
    for (uint32_t w = 0; w < 3; ++w)
      if (w < wave) before += s_wave_total[w];
    if (valid) {
      uint32_t* c = &s_out[tid * kCmdWords];
      c[0] = len; c[1] = 1u; c[2] = before; c[3] = (uint32_t)vertex_offset; c[4] = rec.x;
    }
    __syncthreads();
    uint32_t* out = a.out_cmds + ((size_t)s_count_base[this_chunk] + (size_t)b * kWireBlockCmds) * kCmdWords;
    for (uint32_t j = tid; j < in_block * kCmdWords; j += 256u) out[j] = s_out[j];
  }
}

}  // namespace mip
