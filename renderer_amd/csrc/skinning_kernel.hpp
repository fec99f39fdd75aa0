// skinning_kernel.hpp — extension (BASELINE config 5): joint palette + posed mesh-space box (gfx950).
#pragma once

#include "instance_kernel.hpp"
#include "stage_args.hpp"

#pragma clang fp contract(off)

namespace mip {

// ---------------------------------------------------------------------------------------
// Extension (BASELINE config 5): skinned instances — joint palette + posed mesh-space box
// ---------------------------------------------------------------------------------------
// The reference has no skinning: this is specified from glTF 2.0
// (section 3.7.3, skins) and checked against this repository's oracle only (orc_skinned_bounds).
//   L_k = T(t_k) * R(q_k) * S(s_k)           the animated LOCAL transform of joint k
//   G_k = G_parent(k) * L_k                  (roots: G_k = L_k; parents precede children)
//   J_k = G_k * inverseBind_k                the palette entry the vertex shader blends
//   posed box = union over k of J_k * joint_box_k          (mesh space, 8 corners per joint)
// A skinned vertex is a convex combination of J_k * v over the joints that influence it, so the
// union of the transformed per-joint bind-pose boxes bounds the posed mesh. That box takes the
// place of GltfMesh.aabb for the instance: the instance kernel reads it (KernelArgs.box_override)
// and runs rows a-2 / a-3 / a-7 on it unchanged — 8 corners under M, fold, centre/half round
// trip, planes, command. All matrices here are affine 3x4 (column-major, a[c*3 + r]); a product
// is, per column c, the column axpys (a0*b0c + a1*b1c) + a2*b2c, plus "+ a3" for the
// translation column — no FMA, this order.
//
// Mapping: one lane per (instance, joint); a wave holds floor(64 / J) instances, a workgroup four
// waves. The two steps that would run mostly idle lanes are re-packed through LDS:
//   hierarchy  level by level over the whole workgroup: the (instance, joint) pairs of one depth
//              are dense in the thread index, so a level costs one or two wave-wide 3x4 products
//              per workgroup instead of one per wave and level (a lane-per-joint loop leaves
//              4 of 5 lanes idle on a humanoid);
//   box fold   one thread per (instance, component) runs over the joints in ascending order —
//              the oracle's order — instead of a log-step exchange of six values per lane.
// Poses are read as five 8-byte loads per lane (40 B, lane-contiguous); palette entries leave
// through LDS so that every store instruction is 1 KiB contiguous: 40 B read + 64 B written per
// joint. What bounds the kernel is workgroup lifetime x resident workgroups (its phases are
// separated by barriers), so registers and LDS are kept small: 66 VGPRs, 18 KB, 7 waves per SIMD.
__device__ __forceinline__ void affine_mul(const float (&a)[12], const float (&b)[12], float (&o)[12]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float v = a[0 * 3 + r] * b[c * 3 + 0] + a[1 * 3 + r] * b[c * 3 + 1] + a[2 * 3 + r] * b[c * 3 + 2];
      if (c == 3) v = v + a[9 + r];
      o[c * 3 + r] = v;
    }
}

__device__ __forceinline__ void lds_read12(const float* p, float (&m)[12]) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float4 a = q[0], b = q[1], c = q[2];
  m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
  m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
}

__device__ __forceinline__ void lds_write12(float* p, const float (&m)[12]) {
  float4* q = reinterpret_cast<float4*>(p);
  q[0] = make_float4(m[0], m[1], m[2], m[3]);
  q[1] = make_float4(m[4], m[5], m[6], m[7]);
  q[2] = make_float4(m[8], m[9], m[10], m[11]);
}

__global__ __launch_bounds__(kSkinBlock) void mip_skinned_bounds_kernel(const SkinArgs a) {
  __shared__ __attribute__((aligned(16))) float s_g[kSkinBlock * 12];   // L, then G, per (instance, joint) pair
  __shared__ float s_box[kSkinBlock * 6];                                // per pair: lo xyz, hi xyz
  __shared__ uint32_t s_sorted[kMaxJoints];                              // joints in depth order (joint | parent << 8)
  __shared__ uint32_t s_level[kMaxJoints + 2];                           // per depth: first sorted entry | ceil(2^16/count) << 8
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t J = a.n_joints;
  const uint32_t ipw = 64u / J, ipb = 4u * ipw;           // instances per wave / workgroup
  const uint32_t block_first = blockIdx.x * ipb;          // < n by the grid size
  const uint32_t in_block = a.n - block_first < ipb ? a.n - block_first : ipb;
  const uint32_t g = (lane * a.inv_joints) >> 16, joint = lane - g * J;
  const uint32_t li = wave * ipw + g;                     // instance within the workgroup
  const bool valid = g < ipw && li < in_block;
  const uint32_t pair = valid ? li * J + joint : 0u;      // == wave*ipw*J + lane for valid lanes

  // ---- pose and joint constants ----
  const float2* pp = reinterpret_cast<const float2*>(a.poses + ((size_t)block_first * J + pair) * kPoseWords);
  const float2 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3], p4 = pp[4];
  const float t[kPoseWords] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y, p4.x, p4.y};
  const float4* jp = reinterpret_cast<const float4*>(&a.joints[valid ? joint : 0u]);
  const float4 j0 = jp[0], j1 = jp[1], j2 = jp[2], j3 = jp[3], j4 = jp[4];
  const float ibm[12] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w, j2.x, j2.y, j2.z, j2.w};
  const float box[6] = {j3.x, j3.y, j3.z, j3.w, j4.x, j4.y};
  // The hierarchy loop below is a chain of short dependent steps; its per-level look-ups come from
  // LDS (a global or kernarg load per level would put ~1 us of cache latency on that chain). Loaded HERE, behind the pose and
  // joint loads: in front of them (round 2-4) the workgroup waited for these few words — two dependent round trips — before it
  // issued the loads its arithmetic waits for (profiles/r05_tile_head.txt, 5).
  // (unconditional loads at clamped indices, kept in registers: every load of the head is in flight at once, and the words go to
  //  LDS with the local transforms below)
  const uint32_t sorted_word = a.joints[tid < J ? tid : 0u].sorted;
  const uint32_t level_at = tid < kMaxJoints + 2u ? tid : 0u;
  const uint32_t inv_raw = a.level_inv[level_at <= kMaxJoints ? level_at : 0u];
  const uint32_t level_word = (uint32_t)a.level_start[level_at] | ((level_at <= kMaxJoints ? inv_raw : 0u) << 8);

  // ---- local transform L = T * R * S ----
  float lr[3][3];
  quat_to_rotation(t[3], t[4], t[5], t[6], lr);
  float G[12];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) G[c * 3 + rr] = lr[rr][c] * t[7 + c];
  G[9] = t[0]; G[10] = t[1]; G[11] = t[2];
  if (valid) lds_write12(&s_g[pair * 12u], G);
  if (tid < J) s_sorted[tid] = sorted_word;
  if (tid < kMaxJoints + 2u) s_level[tid] = level_word;
  __syncthreads();

  // ---- hierarchy, one level at a time over the whole workgroup ----
  for (uint32_t d = 1; d <= a.max_depth; ++d) {
    const uint32_t lv = s_level[d];
    const uint32_t start = lv & 0xffu, cnt = (s_level[d + 1] & 0xffu) - start, inv = lv >> 8;
    if (tid < in_block * cnt) {
      const uint32_t inst_l = (tid * inv) >> 16;  // tid / cnt
      const uint32_t packed = s_sorted[start + (tid - inst_l * cnt)];
      const uint32_t k = packed & 0xffu, pk = packed >> 8;
      float P[12], Lk[12], Gk[12];
      lds_read12(&s_g[(inst_l * J + pk) * 12u], P);
      lds_read12(&s_g[(inst_l * J + k) * 12u], Lk);
      affine_mul(P, Lk, Gk);
      lds_write12(&s_g[(inst_l * J + k) * 12u], Gk);
    }
    __syncthreads();
  }
  if (a.max_depth) lds_read12(&s_g[pair * 12u], G);

  // ---- palette entry (mat4 per joint: 64 B per lane, lane-contiguous) ----
  float Jm[12];
  affine_mul(G, ibm, Jm);
  if (a.palette) {
    // Through LDS, so that every store instruction of the wave is 1 KiB contiguous (four 16-byte
    // stores per lane at a 64-byte lane stride reach 3.4 TB/s on this chip, lane-contiguous ones
    // 6.4: tools/micro/store_pattern.hip). Staged as 3x4 in the pair's own slot — nobody else
    // reads it after the last level — and written out as mat4: float4 q of the wave's range is
    // column q%4 of pair q/4, with w = 0,0,0,1.
    const uint32_t wave_inst0 = wave * ipw;
    const uint32_t wave_insts = wave_inst0 < in_block ? (in_block - wave_inst0 < ipw ? in_block - wave_inst0 : ipw) : 0u;
    const uint32_t wave_pair0 = wave_inst0 * J;
    const float* wave_lds = &s_g[wave_pair0 * 12u];
    if (valid) lds_write12(&s_g[pair * 12u], Jm);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4* out = a.palette + ((size_t)block_first * J + wave_pair0) * 4u;
    const uint32_t quads = wave_insts * J * 4u;
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {
      const uint32_t qd = lane + 64u * i;
      if (qd < quads) {
        const float* c = wave_lds + (qd >> 2) * 12u + (qd & 3u) * 3u;
#ifdef MIP_EXP_SKIN_NO_NT
        out[qd] = make_float4(c[0], c[1], c[2], (qd & 3u) == 3u ? 1.0f : 0.0f);
#else
        store_stream16(&out[qd], make_float4(c[0], c[1], c[2], (qd & 3u) == 3u ? 1.0f : 0.0f));  // written once, 1 KiB per instruction
#endif
      }
    }
  }

  // ---- this joint's share of the posed box ----
  // The eight corners of the joint box under J_k, folded — or, whenever no value on the way can overflow or be NaN,
  // the same fold WITHOUT the corners (instance_kernel.hpp, instance_separable: fl(a + b) is monotone in each operand,
  // so the smallest corner coordinate is the sum of the three smallest products; equal up to the sign of a zero).
  // Guard, per wave: S = sum |J_k entries| (a NaN or an infinity anywhere makes it NaN or inf), and every product and
  // partial sum of the fold is bounded by S * (3 * max|box| + 1) = S * box_bound. 54 + 13 instead of 126 vector
  // instructions per lane; this kernel is bound by VALU issue.
  float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
  float hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
  float mag = fabsf(Jm[0]);
#pragma unroll
  for (int q = 1; q < 12; ++q) mag += fabsf(Jm[q]);
  const bool separable = mag * a.box_bound < 1.0e38f;
  if (!(box[0] > box[3] || box[1] > box[4] || box[2] > box[5])) {
    if (__builtin_expect(__any(!separable), 0)) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {  // corner order of src/ecs.rs:149-160
        const float x = box[(c & 1) ? 3 : 0], z = box[(c & 2) ? 5 : 2], y = box[(c & 4) ? 4 : 1];
        float v[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) v[rr] = Jm[0 * 3 + rr] * x + Jm[1 * 3 + rr] * y + Jm[2 * 3 + rr] * z + Jm[9 + rr];
        fold_corner(v, lo, hi);
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const float x0 = Jm[0 * 3 + rr] * box[0], x1 = Jm[0 * 3 + rr] * box[3];
        const float y0 = Jm[1 * 3 + rr] * box[1], y1 = Jm[1 * 3 + rr] * box[4];
        const float z0 = Jm[2 * 3 + rr] * box[2], z1 = Jm[2 * 3 + rr] * box[5];
        lo[rr] = fminf(x0, x1) + fminf(y0, y1) + fminf(z0, z1) + Jm[9 + rr];
        hi[rr] = fmaxf(x0, x1) + fmaxf(y0, y1) + fmaxf(z0, z1) + Jm[9 + rr];
      }
    }
  }
  if (valid) {
    float2* b2 = reinterpret_cast<float2*>(&s_box[pair * 6u]);
    b2[0] = make_float2(lo[0], lo[1]);
    b2[1] = make_float2(lo[2], hi[0]);
    b2[2] = make_float2(hi[1], hi[2]);
  }
  __syncthreads();

  // ---- fold over the joints, in the oracle's order: one thread per (instance, component) ----
  for (uint32_t e = tid; e < in_block * 6u; e += kSkinBlock) {
    const uint32_t inst_l = e / 6u, comp = e - inst_l * 6u;
    const bool is_min = comp < 3u;
    float v = is_min ? 3.40282347e+38f : -3.40282347e+38f;
    const float* src = &s_box[inst_l * J * 6u + comp];
    uint32_t k = 0;
    for (; k + 4u <= J; k += 4u) {  // four reads in flight, folded in ascending order
      const float x0 = src[k * 6u], x1 = src[k * 6u + 6u], x2 = src[k * 6u + 12u], x3 = src[k * 6u + 18u];
      // f32::min / f32::max: a NaN operand is ignored
      v = is_min ? fminf(fminf(fminf(fminf(v, x0), x1), x2), x3) : fmaxf(fmaxf(fmaxf(fmaxf(v, x0), x1), x2), x3);
    }
    for (; k < J; ++k) {
      const float x = src[k * 6u];
      v = is_min ? fminf(v, x) : fmaxf(v, x);
    }
    a.local_box[(size_t)(block_first + inst_l) * 8u + comp + (comp >= 3u ? 1u : 0u)] = v;
  }
}

}  // namespace mip
