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

constexpr uint32_t kErrChunkOverflow = 2u;

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

}  // namespace mip
