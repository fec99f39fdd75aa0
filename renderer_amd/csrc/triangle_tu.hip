// triangle_tu.hip — translation unit of the per-triangle stage (row f-1): the kernels of triangle_kernels.hpp and
// their launchers. Built with the library's floating-point flags PLUS -fno-slp-vectorize (see triangle_args.hpp).
#include "triangle_kernels.hpp"

namespace mip {

void launch_triangle_cull_waves(uint32_t blocks, hipStream_t stream, const TriangleArgs& a) {
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

void launch_recompact(hipStream_t stream, const RecompactArgs& a) {
  hipLaunchKernelGGL(mip_recompact_kernel, dim3(1), dim3(1024), 0, stream, a);
}

void launch_recompact_wide(hipStream_t stream, const RecompactWideArgs& a) {
  hipLaunchKernelGGL(mip_recompact_count_kernel, dim3(a.n_blocks), dim3(1024), 0, stream, a);
  hipLaunchKernelGGL(mip_recompact_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
  hipLaunchKernelGGL(mip_recompact_scatter_kernel, dim3(a.n_blocks), dim3(1024), 0, stream, a);
}

}  // namespace mip
