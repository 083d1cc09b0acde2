// ghr_render_bwd2.h -- wave-level primitives of K8's "scan" arithmetic: the back-to-front gradient walk with lanes bound to
// (Gaussian, pixel) PAIRS.  Used by k_render_bwd_cells (ghr_render_bwd3.h).  The kernel these were developed in
// (k_render_bwd_scan: the same chunk arithmetic inside a tile pipeline that staged whole 256-entry batches; slower,
// DESIGN.md 10) was removed in round 3; it is archived as tools/experiments/r02_k8_scan_kernel_removed_r03.patch.
// Follows R:cuda_rasterizer/backward.cu:403-561 (renderCUDA); replaces k_render_bwd (ghr_render_bwd.h) for needle
// (strand) workloads.
//
// k_render_bwd binds a lane to a pixel and walks the cell's list one entry per pass: every pass pays the full
// per-pixel step (~60 VALU) plus a 16-lane butterfly over the 16 gradient components (33 VALU) for a handful of
// contributing pixels (measured on 500k strands: 31 % of the lanes useful, ~130 VALU per wave pass).  Here a wave owns
// ONE 4x4-pixel cell at a time and processes its list 16 entries per step:
//
//   lane = (k, m):  k = lane >> 4 = pixel column of the cell,  m = lane & 15 = entry of the 16-entry chunk;
//                   the lane evaluates its entry at the four pixels (k, q), q = 0..3, of its column.
//
//   * the per-pixel recurrences of the reference run ACROSS the 16 lanes of a DPP row as scans:
//       T_i  = T_in / prod_{j<=i} (1 - alpha_j)                       (backward.cu:507)   row product scan
//       dL/dalpha_i = cdot_i T_i - (PS_i + T_final bg.dL) / (1 - alpha_i)                  row sum scan (exclusive)
//     with PS_i = sum_{j behind i} alpha_j T_j cdot_j and cdot = colour . dL/dpixel.  That is the closed form of the
//     reference's accum_rec / last_alpha / last_color recurrence (backward.cu:519-538): by induction
//     accum_rec_i . dL = PS_i / (T_i (1 - alpha_i)).  Same real numbers, different fp32 rounding (1e-7 level).
//   * colour . dL/dpixel (10 channels x 16 pixels x 16 entries) is three v_mfma_f32_16x16x4_f32 (exact fp32 FMA
//     chains): A = dL/dpixel of the cell (constant per cell), B = the entries' colours.
//   * the reduction over the cell's 16 pixels -- what the butterfly did -- is a product with a matrix that does not
//     depend on the entry: the line components (ghr_device.h, LineAcc) are sums over pixels of
//       Q dx {1, u},  Q dx v,  Q dy,  Q dy v,  Q,  w dL/dpixel[ch]
//     so after pre-summing over the lane's own four pixels, five MFMAs (geometry) + four (colours) leave, in lane
//     (k', c), component c of the entries 4k' .. 4k'+3: a DPP row holds one whole 64-B line per register, and the
//     accumulation into the (tile, Gaussian) instance line is the same one-line-per-16-lanes workgroup-scope L2
//     atomic as before -- per 16 entries, not per entry.
//   MFMA here is not a GEMM reshaping of byte work: it is the exact-fp32 16-lane reduction tree, on the otherwise idle
//   matrix pipe.  ~170 VALU + 12 MFMA per 256 pair slots, against 4 x ~130 VALU for the same slots before.
//
// Which pairs contribute is decided exactly as in k_render_fwd / k_render_bwd (same unfused `power`, same exp), so the
// result differs from k_render_bwd only by fp32 summation order.
#pragma once
#include "ghr_device.h"

namespace ghr {

#define GHR_B2_PLANE 264  // stride (floats) of the staged colour planes: rows k and k+1 of a wave read planes 8 banks apart

#if defined(__HIP_DEVICE_COMPILE__)
// ---- wave primitives -------------------------------------------------------------------------------------------------
// Inclusive scans over the 16 lanes of every DPP row, four independent values at a time.  row_shr:n without bound_ctrl
// leaves the lanes that have no source untouched, which is the identity for a scan.  The four registers are
// interleaved, so the "VALU write -> DPP read" hazard (2 wait states) between consecutive steps on one register is
// covered by the three instructions in between; s_nop 1 covers the producer in front of the block (the hazard
// recogniser does not look inside inline asm).
#define GHR_SCAN4(op, n)                                                          \
    op " %0, %0, %0 row_shr:" n " row_mask:0xf bank_mask:0xf\n\t"                   \
    op " %1, %1, %1 row_shr:" n " row_mask:0xf bank_mask:0xf\n\t"                   \
    op " %2, %2, %2 row_shr:" n " row_mask:0xf bank_mask:0xf\n\t"                   \
    op " %3, %3, %3 row_shr:" n " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row_scan_mul4(float& a0, float& a1, float& a2, float& a3)
{
    asm volatile("s_nop 1\n\t" GHR_SCAN4("v_mul_f32_dpp", "1") GHR_SCAN4("v_mul_f32_dpp", "2")
                 GHR_SCAN4("v_mul_f32_dpp", "4") GHR_SCAN4("v_mul_f32_dpp", "8")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}
// inclusive sum scan of s0..s3 (in place), e = the same scan shifted down one lane with 0 entering (exclusive scan),
// t = lane 15 of the scanned row broadcast to the row (row total)
__device__ __forceinline__ void row_scan_add4(float& s0, float& s1, float& s2, float& s3, float& e0, float& e1, float& e2,
                                              float& e3, float& t0, float& t1, float& t2, float& t3)
{
    asm volatile("s_nop 1\n\t" GHR_SCAN4("v_add_f32_dpp", "1") GHR_SCAN4("v_add_f32_dpp", "2")
                 GHR_SCAN4("v_add_f32_dpp", "4") GHR_SCAN4("v_add_f32_dpp", "8")
                 "v_mov_b32_dpp %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %8, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %9, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %10, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %11, %3 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(t0),
                   "=&v"(t1), "=&v"(t2), "=&v"(t3));
}
#undef GHR_SCAN4
// lane 15 of every row broadcast to its row, four values
__device__ __forceinline__ void row_last4(float& a0, float& a1, float& a2, float& a3)
{
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %3 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}
// D = A B + C with A[i][k] = a of lane 16k + i, B[k][j] = b of lane 16k + j, D[4(l>>4) + r][l & 15] = d[r] of lane l
// (v_mfma_f32_16x16x4_f32: fp32 in, fp32 accumulate, bit-for-bit an fmaf chain over k)
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)  // set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
#endif

typedef float f2b __attribute__((ext_vector_type(2)));  // pairs of pixel rows: v_pk_mul / v_pk_add / v_pk_fma


// ---- self test of the wave primitives (tests/test_gpu_wave_primitives.py): in[8][64] -> out[12][64] ---------------------
// ghr_selftest_math (include/ghr.h): the device twins of the transcendentals of ghr_device.h on caller data
__global__ void __launch_bounds__(256) k_math_selftest(int n, const float* __restrict__ in, float* __restrict__ out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f4 v = reinterpret_cast<const f4*>(in)[i];
    reinterpret_cast<f4*>(out)[i] = f4{fast_exp(v.x), fast_rcp(v.y), fast_sqrt(v.z), fast_log(v.w)};
#endif
}

__global__ void k_wave_selftest(const float* __restrict__ in, float* __restrict__ out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int l = threadIdx.x & 63;
    float a0 = in[l], a1 = in[64 + l], a2 = in[128 + l], a3 = in[192 + l];
    float p0 = a0, p1 = a1, p2 = a2, p3 = a3;
    row_scan_mul4(p0, p1, p2, p3);
    out[l] = p0; out[64 + l] = p3;
    float s0 = a0, s1 = a1, s2 = a2, s3 = a3, e0, e1, e2, e3, t0, t1, t2, t3;
    row_scan_add4(s0, s1, s2, s3, e0, e1, e2, e3, t0, t1, t2, t3);
    out[128 + l] = s1; out[192 + l] = e1; out[256 + l] = t1; out[320 + l] = s2;
    float b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    row_last4(b0, b1, b2, b3);
    out[384 + l] = b2;
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = mfma16(in[256 + l], in[320 + l], d);
    d = mfma16(in[384 + l], in[448 + l], d);
    out[448 + l] = d.x; out[512 + l] = d.y; out[576 + l] = d.z; out[640 + l] = d.w;
    out[704 + l] = (float)lanes_below(0xF0F0F0F0F0F0F0F0ull);
#endif
}

}  // namespace ghr
