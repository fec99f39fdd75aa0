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
// chunk = [32-B header {count, index total} | body]. Every 64 records of a body (a SUB-BLOCK) carry the firstIndex of their
// first record (8-byte form: word q of the 16-byte header of a 256-record block anchors records 64 q ..; packed form: every
// 64 records are a block of their own behind {firstIndex, first_instance_base, index_bits, 0}).
// The unit of work is a GROUP of four consecutive sub-blocks of one chunk — 256 commands, 5 KB of the merged list — and ONE
// WAVE expands it entirely on its own: lane l takes record l of each of the four sub-blocks (four coalesced loads in flight
// together, then the eight table gathers they name, together), every sub-block gets its own DPP scan from its own anchor
// (four independent scans, no carry between them), the 256 commands are staged in the wave's own 5 KB of LDS at the
// destination's offset modulo 16 and leave as five full-width 16-BYTE store instructions (1 KiB each) plus at most three
// dwords at either end. No barrier after the chunk tables, no cross-wave prefix, no shared staging.
// How it got here (profiles/r04_wire_merge.txt): round 3 expanded a 256-record block per WORKGROUP step — two
// __syncthreads and an LDS prefix across the waves per block, 4-byte copy-out: 16.8 us in the kernel trace for the 8-rank
// shape of BASELINE configs[3] (65 MB: 0.49 of 8 TB/s). One sub-block per wave and step, barrier-free: the same 16-17 us —
// the barriers were not the cost; the loads alone took 10 us: 8 192 waves x 5 dependent steps of (records -> table
// entries -> stores), 1.5 KB in flight per wave. Four sub-blocks per step with all their loads issued together: below.

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

#ifndef MIP_MERGE_STORE_NT
#define MIP_MERGE_STORE_NT 1  // 1 (default) = the 16-byte stores of the merged list are non-temporal, 2 = `sc1 nt` through a buffer descriptor
#endif
#ifndef MIP_MERGE_EXP
#define MIP_MERGE_EXP 0       // tuning builds only (results wrong): 1 = no stores, 2 = no table gather
#endif

#ifndef MIP_MERGE_GROUP_SUBS
#define MIP_MERGE_GROUP_SUBS 4
#endif
constexpr uint32_t kMergeGroupSubs = MIP_MERGE_GROUP_SUBS;               // sub-blocks whose loads a wave has in flight together (a multiple of 4)
constexpr uint32_t kMergeGroupCmds = kMergeGroupSubs * kWireSubBlock;    // commands a wave expands per step
constexpr uint32_t kMergeRoundSubs = 4;                                  // sub-blocks that go through the wave's LDS staging at a time
constexpr uint32_t kMergeRoundCmds = kMergeRoundSubs * kWireSubBlock;    // = 256 commands = 5 120 B of the merged list

// kPacked: the chunks are in the packed wire form (MIP_OUT_WIRE_PACKED): one 32-bit record per command,
// instance index | mesh << index_bits | lod << 31, block header {firstIndex, first_instance_base, index_bits, 0}.
template <bool kPacked>
__global__ __launch_bounds__(256) void mip_merge_wire_lists_kernel(const MergeWireArgs a) {
  constexpr uint32_t kSub = kWireSubBlock;
  static_assert(kSub == 64 && kWireBlockCmds == kMergeRoundCmds && kWireBlockCmds / kSub == kWireBlockHeaderWords && kWirePackedBlockCmds == kSub &&
                kMergeGroupSubs % kMergeRoundSubs == 0,
                "one lane per record of a sub-block; a staging round = one block of the 8-byte form = four blocks of the packed form");
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[4][kMergeRoundCmds * kCmdWords + 4];  // per wave: 256 commands + the alignment shift
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // Chunk tables, in EVERY wave's own registers: lane k reads header k (one round trip for all <= 64 chunks), three wave scans;
  // a group's chunk is a ballot + popcount and its bases are v_readlane — no LDS table, no barrier, no wave that the others
  // wait for (round 3 and the first versions of this kernel built the tables in wave 0 and paid a __syncthreads plus a chain of
  // ~10 dependent LDS reads per wave before the first record load could be issued).
  static_assert(kMaxMergeChunks <= 64, "one lane per chunk");
  const uint32_t fits = kPacked ? (uint32_t)((a.stride - 32u) / (kWirePackedBlockWords * 4u)) * kWirePackedBlockCmds
                                : (uint32_t)((a.stride - 32u) / (kWireBlockWords * 4u)) * kWireBlockCmds;
  const uint32_t capacity = a.capacity < fits ? a.capacity : fits;
  uint32_t t_count = 0, t_total = 0;
  if (lane < a.n_chunks) {
    const uint2 h = *reinterpret_cast<const uint2*>(a.chunks + lane * a.stride);
    t_count = h.x;
    t_total = h.y;
  }
  if (__any(t_count > capacity) && blockIdx.x == 0 && tid == 0) raise_error(a.error_flag, kErrChunkOverflow);  // a shard emitted more than the exchanged chunk holds
  t_count = t_count > capacity ? capacity : t_count;
  const uint32_t t_groups = (t_count + kMergeGroupCmds - 1u) / kMergeGroupCmds;
  const uint32_t c_incl = wave_inclusive_scan(t_count), s_incl = wave_inclusive_scan(t_total), g_incl = wave_inclusive_scan(t_groups);
  const uint32_t c_excl = c_incl - t_count, s_excl = s_incl - t_total, g_excl = g_incl - t_groups;
  if (blockIdx.x == 0 && tid == 63u) {  // lanes past the last chunk contributed zeros: lane 63 holds the totals
    a.out_count[0] = c_incl;
    a.out_count[1] = s_incl;
  }
  const uint32_t total_groups = (uint32_t)__builtin_amdgcn_readlane((int)g_incl, 63);
  uint32_t* const stage = s_stage[wave];
  // consecutive waves of the launch take consecutive groups: at any moment the launch writes one contiguous window of the list
  for (uint32_t group = blockIdx.x * 4u + wave; group < total_groups; group += gridDim.x * 4u) {
    // chunks whose groups all lie in front of this one (chunks without commands have none and are skipped)
    const uint32_t chunk = (uint32_t)__popcll(__ballot(lane < a.n_chunks && g_incl <= group));
    const uint32_t g = group - (uint32_t)__builtin_amdgcn_readlane((int)g_excl, (int)chunk);  // group within its chunk
    const uint32_t chunk_count = (uint32_t)__builtin_amdgcn_readlane((int)t_count, (int)chunk);
    const uint32_t count_base = (uint32_t)__builtin_amdgcn_readlane((int)c_excl, (int)chunk);
    const uint32_t index_base = (uint32_t)__builtin_amdgcn_readlane((int)s_excl, (int)chunk);
    const uint32_t in_group = chunk_count - g * kMergeGroupCmds < kMergeGroupCmds ? chunk_count - g * kMergeGroupCmds : kMergeGroupCmds;
    const uint32_t* body = reinterpret_cast<const uint32_t*>(a.chunks + chunk * a.stride + 32);

    // ---- the four records of this lane and the four anchors: every load issued before any is used ----
    uint32_t instance[kMergeGroupSubs], mesh_lod[kMergeGroupSubs], anchor[kMergeGroupSubs];
    bool valid[kMergeGroupSubs];
    if constexpr (kPacked) {
      uint32_t raw[kMergeGroupSubs];
      uint4 hdr[kMergeGroupSubs];
#pragma unroll
      for (uint32_t q = 0; q < kMergeGroupSubs; ++q) {
        valid[q] = q * kSub + lane < in_group;
        const uint32_t* blk = body + ((size_t)g * kMergeGroupSubs + q) * kWirePackedBlockWords;
        const bool exists = q * kSub < in_group;  // wave-uniform: the sub-block has at least one record (and a header)
        hdr[q] = exists ? *reinterpret_cast<const uint4*>(blk) : make_uint4(0u, 0u, 0u, 0u);
        raw[q] = valid[q] ? blk[kWireBlockHeaderWords + lane] : 0u;
      }
#pragma unroll
      for (uint32_t q = 0; q < kMergeGroupSubs; ++q) {
        uint32_t bits = hdr[q].z;
        if (bits > 31u) {  // a corrupt header must not become an undefined shift
          if (lane == 0) raise_error(a.error_flag, kErrWireRecord);
          bits = 31u;
        }
        const uint32_t low = raw[q] & 0x7fffffffu;
        instance[q] = hdr[q].y + (low & ((1u << bits) - 1u));
        mesh_lod[q] = (low >> bits) | (raw[q] & 0x80000000u);
        anchor[q] = hdr[q].x;
      }
    } else {
      const uint32_t* blk0 = body + (size_t)g * (kMergeGroupSubs / kMergeRoundSubs) * kWireBlockWords;
      uint4 hdr[kMergeGroupSubs / kMergeRoundSubs];
      uint2 rec[kMergeGroupSubs];
#pragma unroll
      for (uint32_t b = 0; b < kMergeGroupSubs / kMergeRoundSubs; ++b)  // wave-uniform addresses: four anchors per 256-record block
        hdr[b] = b * kWireBlockCmds < in_group ? *reinterpret_cast<const uint4*>(blk0 + (size_t)b * kWireBlockWords) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (uint32_t q = 0; q < kMergeGroupSubs; ++q) {
        valid[q] = q * kSub + lane < in_group;
        const uint32_t* blk = blk0 + (size_t)(q / kMergeRoundSubs) * kWireBlockWords;
        rec[q] = valid[q] ? *reinterpret_cast<const uint2*>(blk + kWireBlockHeaderWords + 2u * ((q % kMergeRoundSubs) * kSub + lane)) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (uint32_t q = 0; q < kMergeGroupSubs; ++q) {
        const uint4 h = hdr[q / kMergeRoundSubs];
        anchor[q] = (q % 4u) == 0u ? h.x : ((q % 4u) == 1u ? h.y : ((q % 4u) == 2u ? h.z : h.w));
        instance[q] = rec[q].x;
        mesh_lod[q] = rec[q].y;
      }
    }
    // ---- the table entries they name ----
    uint32_t len[kMergeGroupSubs];
    int32_t vertex_offset[kMergeGroupSubs];
#pragma unroll
    for (uint32_t q = 0; q < kMergeGroupSubs; ++q) {
      uint32_t mesh = mesh_lod[q] & 0x7fffffffu;
      if (mesh >= a.n_meshes) {  // never follow a corrupt record out of the table
        if (valid[q]) raise_error(a.error_flag, kErrWireRecord);
        mesh = 0u;
      }
      len[q] = 0u;
      vertex_offset[q] = 0;
#if MIP_MERGE_EXP == 2
      len[q] = mesh & 0xffu;
#else
      if (valid[q] && a.n_meshes) {
        len[q] = (mesh_lod[q] >> 31) ? a.meshes[mesh].len1 : a.meshes[mesh].len0;
        vertex_offset[q] = a.mesh_draw[mesh].vertex_offset;
      }
#endif
    }
    // ---- expand, 256 commands (four sub-blocks) at a time through the wave's staging area: round r of the group is words
    //      [1280 r, 1280 r + 5 in_round) of the merged list from gout0 on ----
    uint32_t* const gout0 = a.out_cmds + ((size_t)count_base + (size_t)g * kMergeGroupCmds) * kCmdWords;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(gout0) >> 2) & 3u;  // destination offset modulo 16 bytes, in words (the same for every round)
#pragma unroll
    for (uint32_t r = 0; r < kMergeGroupSubs / kMergeRoundSubs; ++r) {
      if (r * kMergeRoundCmds >= in_group) break;  // wave-uniform
      const uint32_t in_round = in_group - r * kMergeRoundCmds < kMergeRoundCmds ? in_group - r * kMergeRoundCmds : kMergeRoundCmds;
      uint32_t* const gout = gout0 + (size_t)r * kMergeRoundCmds * kCmdWords;
      __builtin_amdgcn_wave_barrier();  // (the previous round's reads of the staging area are done: LDS is in order per wave)
#pragma unroll
      for (uint32_t k = 0; k < kMergeRoundSubs; ++k) {
        const uint32_t q = r * kMergeRoundSubs + k;
        const uint32_t incl = wave_inclusive_scan(len[q]);
        if (valid[q]) {
          uint32_t* c = &stage[mis + (k * kSub + lane) * kCmdWords];  // 20-byte pitch: conflict-free
          c[0] = len[q]; c[1] = 1u; c[2] = anchor[q] + index_base + (incl - len[q]); c[3] = (uint32_t)vertex_offset[q]; c[4] = instance[q];
        }
      }
      __builtin_amdgcn_wave_barrier();
      const uint32_t words = in_round * kCmdWords;
      const uint32_t head = ((4u - mis) & 3u) < words ? ((4u - mis) & 3u) : words;  // words in front of the first 16-byte boundary
      const uint32_t quads = (words - head) >> 2;
      const uint32_t tail = words - head - 4u * quads;
      (void)tail;
#if MIP_MERGE_EXP == 1
      if (anchor[0] == 0x12345678u && len[0] == 0x7654321u) gout[lane] = stage[mis + lane];
#else
      if (lane < head) gout[lane] = stage[mis + lane];
#if MIP_MERGE_STORE_NT == 2
      const __amdgpu_buffer_rsrc_t d_out = stream_descriptor(gout + head, quads * 16u);
#endif
      for (uint32_t qd = lane; qd < quads; qd += 64u) {  // 320 for a whole round: five full-width stores of 1 KiB
        const uint4 v = *reinterpret_cast<const uint4*>(&stage[mis + head + 4u * qd]);  // 16-byte aligned in LDS too
        uint4* dst = reinterpret_cast<uint4*>(gout + head + 4u * qd);
#if MIP_MERGE_STORE_NT == 1
        typedef uint32_t merge_v4u __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store((merge_v4u){v.x, v.y, v.z, v.w}, reinterpret_cast<merge_v4u*>(dst));
#elif MIP_MERGE_STORE_NT == 2
        (void)dst;
        store_stream16(d_out, qd * 16u, make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)));
#else
        *dst = v;
#endif
      }
      if (lane < tail) gout[head + 4u * quads + lane] = stage[mis + head + 4u * quads + lane];
#endif
    }
  }
}

}  // namespace mip
