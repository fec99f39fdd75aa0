// views_kernel.hpp — several frusta over the same instances in ONE launch (gfx950): per-light culled draw
// lists (SURVEY.md section 8 f-4 "per-light cull lists"), shadow cascades, cube faces, stereo.
#pragma once

#include "instance_kernel.hpp"
#include "stage_args.hpp"

#pragma clang fp contract(off)

namespace mip {

// Each view is a complete cull_pass over the resident instances — its own six planes
// (cull_pipeline.rs:99-120), its own LOD reference point (pick_lod, helpers.rs:3-11), its own compacted
// VkDrawIndexedIndirectCommand stream, count and visibility bitmap — with exactly the results of one
// mip_run per view. What the views share is everything that does not depend on the frustum: the 36
// bytes of instance data are read once, and the model matrix and world AABB (rows a-1, a-2) are built
// once per instance instead of once per view. Up to four views per launch: view v's cross-tile prefix
// (same one-hop scheme as the instance kernel, one state per view) is resolved and its commands copied
// out by wave v of the workgroup, so the four hops run side by side. The matrices are not written here
// (they do not depend on the view; the frame's mip_run writes them).
// The plane test of coarse_culled with the subtraction folded into the comparison: `sd - e > 0` and `sd > e` agree for
// every pair of floats while subnormals are kept (a difference of two floats that underflows is exact, so its sign
// is the comparison's; inf - inf and NaN give false on both sides) — one VALU instruction less per plane, and this
// kernel is bound by VALU issue (4 views: ~750 instructions per wave).
__device__ __forceinline__ bool view_culled(const float (&h)[3], const float (&c)[3], const float (&planes)[24]) {
  bool outside = false;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const float nx = planes[p * 4 + 0], ny = planes[p * 4 + 1], nz = planes[p * 4 + 2], d = planes[p * 4 + 3];
    const float e = h[0] * fabsf(nx) + h[1] * fabsf(ny) + h[2] * fabsf(nz);
    float a0 = nx * c[0];
    float a1 = ny * c[1];
    const float a2 = nz * c[2];
    a0 += a2;
    a1 += d;
    const float sd = a0 + a1;
    outside = outside || (sd > e);
  }
  return outside;
}

// The aggregate {Σ index_len of the visible : 32 | emitted commands : 32} of tile u FOR ONE VIEW, computed by one wave from
// the tile's inputs: what resolve_prefix falls back to when a predecessor has not published (instance_kernel.hpp).
template <bool kGeneral>
__device__ __forceinline__ unsigned long long help_view_aggregate(uint32_t v, uint32_t u, uint32_t lane) {
  const auto* ka = cold_kernel_args<ViewsArgs>();  // re-read from the kernarg segment: nothing of the hot path is kept alive for this
  float planes[24], cam[3];
#pragma unroll
  for (int k = 0; k < 24; ++k) planes[k] = ka->view[v].planes[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) cam[k] = ka->view[v].cam[k];
  const float* pos = ka->pos;
  const float4* rot = ka->rot;
  const float* scale = ka->scale;
  const uint32_t* mesh_id = ka->mesh_id;
  const MeshEntry* meshes = ka->meshes;
  const uint32_t n = ka->n;
  struct { const float* box_override; } no_box = {nullptr};
  uint32_t cnt = 0, sum = 0;
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
    instance_tiered<false, kGeneral>(no_box, jl, r, px, py, pz, sc, mb, inst);
    float box_h[3], box_c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      box_h[k] = (inst.maxs[k] - inst.mins[k]) * 0.5f;
      box_c[k] = (inst.mins[k] + inst.maxs[k]) * 0.5f;
    }
    const bool visible = active && !view_culled(box_h, box_c, planes);
    const uint32_t len = lod_is_far(cam, px, py, pz) ? mb.len1 : mb.len0;
    cnt += (uint32_t)__popcll(__ballot(visible && len > 0u));
    sum += wave_sum(visible ? len : 0u);
  }
  return ((unsigned long long)sum << 32) | cnt;
}

// kGeneral = false: the upload-time census found every instance finite and separable-safe (as for the frame kernel).
// Compiled for 8 waves per SIMD (round 5): with the bound of 4 the compiler took 96 scalar registers, and the hardware then
// keeps SIX workgroups on a CU, not the eight its vector registers and LDS allow (found on the per-triangle stage's
// wave-per-command kernel, which had the same bound and the same 96: profiles/r05_wave_kernel_lifetimes.txt).
#ifndef MIP_VIEWS_WAVES_PER_SIMD
#define MIP_VIEWS_WAVES_PER_SIMD 8
#endif
template <bool kGeneral>
__global__ __launch_bounds__(kTile, kGeneral ? 4 : MIP_VIEWS_WAVES_PER_SIMD) void mip_cull_views_kernel(const ViewsArgs a) {
  // four words per staged command (instanceCount is the constant 1: the copy-out writes it): 16 KB for the four views, which
  // with the VGPRs lets EIGHT workgroups share a CU (five words: 20.6 KB, seven; the launch is bound by tiles in flight x
  // a tile's latency, not by bytes)
  constexpr uint32_t kStaged = 4;
  __shared__ __attribute__((aligned(16))) uint32_t s_cmd[kMaxViews][kTile * kStaged];
  __shared__ uint32_t s_wave_count[kMaxViews][kWaves], s_wave_sum[kMaxViews][kWaves];
#if MIP_TILE >= 256  // (tile-size experiment builds, tools/r05_cfg2_tile128.sh, do not run this kernel)
  static_assert(kWaves >= kMaxViews, "one wave per view finishes that view");
#endif

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t tile = blockIdx.x;
#ifdef MIP_DEBUG_STAMPS
  if (a.debug_tile_mult) tile = (uint32_t)(((unsigned long long)blockIdx.x * a.debug_tile_mult + a.debug_tile_add) % a.n_tiles);
#endif
  const uint32_t tile_first = tile * kTile;
  const uint32_t i = tile_first + tid;
  const bool active = i < a.n;
  const uint32_t il = active ? i : a.n - 1u;

  // ---- shared by all views: loads, model matrix, world AABB (as the instance kernel) ----
  const float px = a.pos[3 * (size_t)il + 0], py = a.pos[3 * (size_t)il + 1], pz = a.pos[3 * (size_t)il + 2];
  const float4 q = a.rot[il];
  const float sc = a.scale[il];
  const uint32_t mesh = a.mesh_id[il];
  const float4 mb0 = *reinterpret_cast<const float4*>(&a.meshes[mesh].min_x);
  const float4 mb1 = *reinterpret_cast<const float4*>(&a.meshes[mesh].max_x);
  MeshEntry mb;
  mb.min_x = mb0.x; mb.min_y = mb0.y; mb.min_z = mb0.z; mb.len0 = __float_as_uint(mb0.w);
  mb.max_x = mb1.x; mb.max_y = mb1.y; mb.max_z = mb1.z; mb.len1 = __float_as_uint(mb1.w);
  const int32_t vertex_offset = a.mesh_draw[mesh].vertex_offset;
  float r[3][3];
  quat_to_rotation(q.x, q.y, q.z, q.w, r);
  Instance inst;
  instance_tiered<false, kGeneral>(a, il, r, px, py, pz, sc, mb, inst);  // the arithmetic tiers of the instance kernel
  float box_h[3], box_c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    box_h[k] = (inst.maxs[k] - inst.mins[k]) * 0.5f;  // AABB::half_extents
    box_c[k] = (inst.mins[k] + inst.maxs[k]) * 0.5f;  // AABB::center
  }

  // ---- per view: frustum test, LOD, wave-level compaction offsets ----
  bool keep[kMaxViews];
  uint32_t len[kMaxViews], rank[kMaxViews], excl_sum[kMaxViews];
  unsigned long long vis_mask[kMaxViews];
  // Everything of a view's arguments its block reads is fetched as ONE batch of scalar loads with one wait — and a view AHEAD:
  // the batch of view v + 1 is issued when view v's plane tests are done (their registers are free then) and arrives under view
  // v's ballots and scans. (Left to itself the compiler fetched 16 plane words, waited, 8 more, waited, the camera, waited, the
  // bitmap pointer, waited: four scalar round trips per view and wave in a kernel that is short of issue slots —
  // profiles/r05_tile_head.txt, 6.)
  float view_planes[24], cam_x = 0.0f, cam_y = 0.0f, cam_z = 0.0f;
  uint32_t* view_bitmap = nullptr;
  auto fetch_view = [&](uint32_t v) {
#pragma unroll
    for (int k = 0; k < 24; ++k) view_planes[k] = a.view[v].planes[k];
    cam_x = a.view[v].cam[0]; cam_y = a.view[v].cam[1]; cam_z = a.view[v].cam[2];
    view_bitmap = a.view[v].bitmap;
  };
  if (a.n_views) fetch_view(0);
#pragma unroll
  for (uint32_t v = 0; v < kMaxViews; ++v) {
    keep[v] = false; len[v] = 0; rank[v] = 0; excl_sum[v] = 0; vis_mask[v] = 0;
    if (v < a.n_views) {
      asm volatile("" ::"s"(view_planes[0]), "s"(view_planes[16]), "s"(cam_x), "s"(view_bitmap));  // the batch is here: one wait
      const bool visible = active && !view_culled(box_h, box_c, view_planes);
      const float dx = cam_x - px, dy = cam_y - py, dz = cam_z - pz;
      const float dist_sq = dx * dx + dy * dy + dz * dz;
      const bool far_lod = dist_sq > kLodDistSqThreshold;
      uint32_t* const this_bitmap = view_bitmap;
      if (v + 1u < kMaxViews)
        if (v + 1u < a.n_views) fetch_view(v + 1u);
      len[v] = far_lod ? mb.len1 : mb.len0;
      keep[v] = visible && len[v] > 0u;
      const uint32_t len_vis = visible ? len[v] : 0u;
      const unsigned long long keep_mask = __ballot(keep[v]);
      vis_mask[v] = __ballot(visible);
      rank[v] = lanes_below(keep_mask);
      const uint32_t incl = wave_inclusive_scan(len_vis);
      excl_sum[v] = incl - len_vis;
      if (lane == 63u) {
        s_wave_count[v][wave] = (uint32_t)__popcll(keep_mask);
        s_wave_sum[v][wave] = incl;
      }
      // visibility bitmap: one 64-bit ballot per wave, written as two words
      if (this_bitmap && lane < 2u) {
        const uint32_t word = (tile_first >> 5) + wave * 2u + lane;
        if (word < a.bitmap_words) this_bitmap[word] = (uint32_t)(vis_mask[v] >> (32u * lane));
      }
    }
  }
  __syncthreads();

  // ---- tile aggregates: WAVE v publishes view v ----
  // (round 2-4: THREAD v did, indexing the argument block per lane — the view's tag and pointers then come by vector loads from
  //  the kernarg segment: three dependent round trips between the barrier and the accumulator's add, on every successor's
  //  path. A wave's number is scalar: the same words are scalar loads, four waves publish their views side by side.)
  {
    const uint32_t pv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    if (pv < a.n_views && lane == 0u) {
      {  // the view's prefix state in ONE batch of scalar loads (publish_aggregate reads these fields one basic block at a time)
        const ViewArgs& vw = a.view[pv];
        asm volatile("" ::"s"(vw.status0), "s"(vw.acc1), "s"(vw.groups_cap), "s"(vw.group_shift), "s"(vw.epoch));
      }
      uint32_t c = 0, s = 0;
#pragma unroll
      for (uint32_t w = 0; w < kWaves; ++w) { c += s_wave_count[pv][w]; s += s_wave_sum[pv][w]; }
      publish_aggregate(a.view[pv], tile, c, s);
    }
  }

  // ---- tile-local command assembly, every view ----
  uint32_t view_first_instance[kMaxViews];  // (one batch of scalar loads, as above)
#pragma unroll
  for (uint32_t v = 0; v < kMaxViews; ++v) view_first_instance[v] = a.view[v].first_instance_base;
  asm volatile("" ::"s"(view_first_instance[0]), "s"(view_first_instance[1]), "s"(view_first_instance[2]), "s"(view_first_instance[3]));
#pragma unroll
  for (uint32_t v = 0; v < kMaxViews; ++v) {
    if (v < a.n_views && keep[v]) {
      uint32_t off_count = 0, off_sum = 0;
#pragma unroll
      for (uint32_t w = 0; w < kWaves; ++w)
        if (w < wave) { off_count += s_wave_count[v][w]; off_sum += s_wave_sum[v][w]; }
      // {indexCount, firstIndex (tile-relative), vertexOffset, firstInstance = draw_index}: one ds_write_b128
      *reinterpret_cast<uint4*>(&s_cmd[v][(off_count + rank[v]) * kStaged]) =
          make_uint4(len[v], off_sum + excl_sum[v], (uint32_t)vertex_offset, view_first_instance[v] + i);
    }
  }
  __syncthreads();

#ifdef MIP_VIEWS_COOP_COPY
  // ---- wave v resolves view v's prefix; then the whole workgroup copies every view out ----
  __shared__ uint32_t s_base_count[kMaxViews], s_base_sum[kMaxViews], s_tile_count[kMaxViews];
  {
    const uint32_t my_view = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    if (my_view < a.n_views) {
      const ViewArgs& view = a.view[my_view];
      uint32_t tile_count = 0, tile_sum = 0;
#pragma unroll
      for (uint32_t w = 0; w < kWaves; ++w) { tile_count += s_wave_count[my_view][w]; tile_sum += s_wave_sum[my_view][w]; }
      uint32_t base_count = 0, base_sum = 0;
      if (tile > 0) resolve_prefix(view, tile, lane, base_count, base_sum, [my_view, lane](uint32_t u) { return help_view_aggregate<kGeneral>(my_view, u, lane); });
      if (lane == 0) {
        if (tile == a.n_tiles - 1u) {
          *view.draw_count = base_count + tile_count;
          if (view.index_total) *view.index_total = base_sum + tile_sum;
        }
        s_base_count[my_view] = base_count;
        s_base_sum[my_view] = base_sum + view.first_index_base;
        s_tile_count[my_view] = tile_count;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t v = 0; v < kMaxViews; ++v) {
    if (v < a.n_views) {
      const uint32_t first_index_add = s_base_sum[v];
      uint32_t* out = a.view[v].cmds + (size_t)s_base_count[v] * kCmdWords;
      const uint32_t words = s_tile_count[v] * kCmdWords;
      for (uint32_t j = tid; j < words; j += kTile) {
        const uint32_t k = j / kCmdWords, f = j - k * kCmdWords;
        uint32_t val = f == 1u ? 1u : s_cmd[v][k * kStaged + (f ? f - 1u : 0u)];
        if (f == 2u) val += first_index_add;
        out[j] = val;
      }
    }
  }
#else
  // ---- wave v finishes view v: prefix over the earlier tiles, coalesced copy-out ----
  const uint32_t my_view = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);  // scalar: the view's words are scalar loads
  if (my_view >= a.n_views) return;
  const ViewArgs& view = a.view[my_view];
  uint32_t tile_count = 0, tile_sum = 0;
#pragma unroll
  for (uint32_t w = 0; w < kWaves; ++w) { tile_count += s_wave_count[my_view][w]; tile_sum += s_wave_sum[my_view][w]; }
  uint32_t base_count = 0, base_sum = 0;
  if (tile > 0) resolve_prefix(view, tile, lane, base_count, base_sum, [my_view, lane](uint32_t u) { return help_view_aggregate<kGeneral>(my_view, u, lane); });
  if (lane == 0 && tile == a.n_tiles - 1u) {
    *view.draw_count = base_count + tile_count;
    if (view.index_total) *view.index_total = base_sum + tile_sum;
  }
  const uint32_t first_index_add = base_sum + view.first_index_base;
  uint32_t* out = view.cmds + (size_t)base_count * kCmdWords;
  const uint32_t words = tile_count * kCmdWords;
  for (uint32_t j = lane; j < words; j += 64u) {
    const uint32_t k = j / kCmdWords, f = j - k * kCmdWords;
    uint32_t val = f == 1u ? 1u : s_cmd[my_view][k * kStaged + (f ? f - 1u : 0u)];  // instanceCount = 1 (generate_work.comp:63)
    if (f == 2u) val += first_index_add;
    out[j] = val;
  }
#endif
}

}  // namespace mip
