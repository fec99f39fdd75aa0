// stages_tu.hip — the library's second translation unit: the kernels that are faster without the SLP vectoriser
// (per-triangle stage, multi-view cull, shadow-pass lists, skinning, the commands-first frame kernel) and their
// launchers. Built with the library's floating-point flags PLUS -fno-slp-vectorize (stage_args.hpp has the numbers).
#include <cstdio>
#include <cstdlib>

#include "triangle_kernels.hpp"
#include "light_lists_kernel.hpp"
#include "skinning_kernel.hpp"
#include "views_kernel.hpp"

namespace mip {

void launch_triangle_cull_waves(uint32_t blocks, hipStream_t stream, const TriangleArgs& a) {
  static bool said = false;
  if (!said && std::getenv("MIP_TUNE_VERBOSE")) {
    said = true;
    int v = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, mip_triangle_cull_kernel, 256, 0);
    hipFuncAttributes fa{};
    (void)hipFuncGetAttributes(&fa, (const void*)mip_triangle_cull_kernel);
    fprintf(stderr, "mip: wave-per-command kernel: %d workgroups per CU by the runtime's count; %d registers, %zu B LDS, %zu B scratch; grid %u\n", v, fa.numRegs,
            fa.sharedSizeBytes, fa.localSizeBytes, blocks);
  }
  hipLaunchKernelGGL(mip_triangle_cull_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void launch_triangle_cull_block(uint32_t threads, uint32_t blocks, hipStream_t stream, const TriangleArgs& a) {
  if (threads == 256u) hipLaunchKernelGGL(mip_triangle_cull_block_kernel<256>, dim3(blocks), dim3(256), 0, stream, a);
  else if (threads == 512u) hipLaunchKernelGGL(mip_triangle_cull_block_kernel<512>, dim3(blocks), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL(mip_triangle_cull_block_kernel<1024>, dim3(blocks), dim3(1024), 0, stream, a);
}

void launch_triangle_cull_parts(uint32_t blocks, hipStream_t stream, const TrianglePartsArgs& a) {
  hipLaunchKernelGGL(mip_triangle_cull_parts_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

uint32_t triangle_chunks_blocks_per_cu() {
  static int per_cu = 0;  // (one device kind per process: gfx950)
  if (per_cu <= 0) {
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, mip_triangle_stage_kernel, 256, 0) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 4;
    }
    if (v > 8) v = 8;
    if (const char* env = std::getenv("MIP_TUNE_TRI_RANGE_BLOCKS_PER_CU")) {  // A/B runs
      const int forced = std::atoi(env);
      if (forced >= 1 && forced <= 8) v = forced;
    }
    if (std::getenv("MIP_TUNE_VERBOSE")) fprintf(stderr, "mip: range kernel: %d workgroups per CU\n", v);
    per_cu = v;
  }
  return (uint32_t)per_cu;
}

void launch_triangle_cull_chunks(uint32_t map_blocks, uint32_t blocks, hipStream_t stream, const TriangleChunkArgs& a) {
  hipLaunchKernelGGL(mip_triangle_prepare_kernel, dim3(map_blocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(mip_triangle_cull_ranges_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void launch_triangle_stage(uint32_t map_blocks, uint32_t blocks, hipStream_t stream, const TriangleChunkArgs& a) {
  TriangleChunkArgs p = a;
  p.t.choice_mode = 3u;  // range map OR size-class histogram, whichever decomposition the frame is for
  hipLaunchKernelGGL(mip_triangle_prepare_kernel, dim3(map_blocks), dim3(256), 0, stream, p);
  TriangleArgs w = a.t;
  w.choice_mode = 2u;    // (returns at once when the frame is the range decomposition's)
  hipLaunchKernelGGL(mip_triangle_sort_scatter_kernel, dim3(map_blocks), dim3(256), 0, stream, w, const_cast<uint32_t*>(a.t.order));
  hipLaunchKernelGGL(mip_triangle_stage_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

void launch_recompact(hipStream_t stream, const RecompactArgs& a) {
  hipLaunchKernelGGL(mip_recompact_kernel, dim3(1), dim3(1024), 0, stream, a);
}

void launch_recompact_wide(hipStream_t stream, const RecompactWideArgs& a) {
  if (a.block_status) {
    hipLaunchKernelGGL(mip_recompact_onepass_kernel, dim3(a.n_blocks), dim3(1024), 0, stream, a);
    return;
  }
  hipLaunchKernelGGL(mip_recompact_count_kernel, dim3(a.n_blocks), dim3(1024), 0, stream, a);
  hipLaunchKernelGGL(mip_recompact_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
  hipLaunchKernelGGL(mip_recompact_scatter_kernel, dim3(a.n_blocks), dim3(1024), 0, stream, a);
}

void launch_light_draw_lists(bool aligned16, uint32_t tiles, hipStream_t stream, const LightListArgs& a) {
  if (aligned16) hipLaunchKernelGGL(mip_light_draw_lists_kernel<true>, dim3(tiles), dim3(kTile), 0, stream, a);
  else hipLaunchKernelGGL(mip_light_draw_lists_kernel<false>, dim3(tiles), dim3(kTile), 0, stream, a);
}

void launch_skinned_bounds(uint32_t blocks, hipStream_t stream, const SkinArgs& a) {
  hipLaunchKernelGGL(mip_skinned_bounds_kernel, dim3(blocks), dim3(kSkinBlock), 0, stream, a);
}

void launch_cull_views(bool general, uint32_t tiles, hipStream_t stream, const ViewsArgs& a) {
  if (general) hipLaunchKernelGGL(mip_cull_views_kernel<true>, dim3(tiles), dim3(kTile), 0, stream, a);
  else hipLaunchKernelGGL(mip_cull_views_kernel<false>, dim3(tiles), dim3(kTile), 0, stream, a);
}

// Every kOrder == 3 instantiation of the frame kernel lives here and only here (api_frame.hip instantiates kOrder == 1):
// an instantiation referenced from both units would be registered twice under one host stub.
namespace {
template <bool kBox, bool kGeneral, int kWire>
FrameKernelFn commands_first(bool first_mover) {
  if constexpr (KernelArgs::kFirstMoverAdds)
    if (first_mover) return (FrameKernelFn)mip_instance_pipeline_kernel<kBox, kGeneral, 3, kWire, true>;
  return (FrameKernelFn)mip_instance_pipeline_kernel<kBox, kGeneral, 3, kWire>;
}
}  // namespace

FrameKernelFn frame_kernel_commands_first(bool box_override, bool general, int wire, bool first_mover) {
  if (box_override) return commands_first<true, true, 0>(first_mover);  // skinned frames: always general, never wire
  if (wire == 2) return general ? commands_first<false, true, 2>(first_mover) : commands_first<false, false, 2>(first_mover);
  if (wire == 1) return general ? commands_first<false, true, 1>(first_mover) : commands_first<false, false, 1>(first_mover);
  return general ? commands_first<false, true, 0>(first_mover) : commands_first<false, false, 0>(first_mover);
}

}  // namespace mip
