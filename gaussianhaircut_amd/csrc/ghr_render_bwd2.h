// ghr_render_bwd2.h -- K8, "scan" form: back-to-front gradient walk with lanes bound to (Gaussian, pixel) PAIRS.
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

typedef float f2b __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(GHR_BLOCK, 4) k_render_bwd_scan(int W, int H, int gx, uint32_t T_tiles,
                                                               const uint32_t* __restrict__ tile_start,
                                                               const uint32_t* __restrict__ point_list,
                                                               const f4* __restrict__ rec, const float* __restrict__ bg,
                                                               const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ dL_dpix,
                                                               const rect4* __restrict__ rects, float* ginst,
                                                               uint32_t cap)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ f4 s_r0[GHR_BLOCK], s_r1[GHR_BLOCK];          // batch entries: {x, y, a, b}, {c, opacity, -, -}
    __shared__ float s_col[GHR_C][GHR_B2_PLANE];             // batch entries: the ten colours, one plane per channel
    __shared__ uint32_t s_slot[GHR_BLOCK];                   // batch entries: gradient line of the instance
    __shared__ uint16_t s_cmask[GHR_BLOCK];                  // batch entries: which of the tile's 16 cells they can touch
    __shared__ uint8_t s_list[4][GHR_BLOCK];                 // per wave: the current cell's entries, in list order
    __shared__ float s_dL[GHR_C][GHR_BLOCK];                 // tile pixels: dL/dpixel planes (pixel = 16 y + x)
    __shared__ float s_T[GHR_BLOCK], s_PS[GHR_BLOCK];        // tile pixels: T and PS + T_final bg.dL carried over chunks
    __shared__ uint32_t s_last[GHR_BLOCK];                   // tile pixels: n_contrib
    __shared__ uint32_t s_gmax[16];                          // per cell: largest n_contrib

    const uint32_t tile = xcd_tile(blockIdx.x, T_tiles);
    if (tile >= T_tiles) return;  // grid padding
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane >> 4, m = lane & 15;
    const float wx0 = (float)(tx * GHR_TILE_X), wy0 = (float)(ty * GHR_TILE_Y);

    const uint32_t beg = min(tile_start[tile], cap);
    const uint32_t n = min(tile_start[tile + 1], cap) - beg;  // see k_render_bwd for `cap`

    if (tid < 16) s_gmax[tid] = 0u;
    __syncthreads();
    {   // this thread's pixel of the tile: tid = 16 y + x
        const int x = tid & 15, y = tid >> 4;
        const int px = tx * GHR_TILE_X + x, py = ty * GHR_TILE_Y + y;
        const bool inside = px < W && py < H;
        const size_t pix = (size_t)W * py + px, plane = (size_t)W * H;
        const float Tf = inside ? final_T[pix] : 0.f;
        const uint32_t last = inside ? n_contrib[pix] : 0u;
        float bgdot = 0.f;
#pragma unroll
        for (int c = 0; c < GHR_C; c++) {
            const float d = inside ? dL_dpix[c * plane + pix] : 0.f;
            s_dL[c][tid] = d;
            bgdot = fma_(bg[c], d, bgdot);
        }
        s_T[tid] = Tf;
        s_PS[tid] = Tf * bgdot;  // backward.cu:535-538: the background term enters like one more splat behind the list
        s_last[tid] = last;
        atomicMax(&s_gmax[(y >> 2) * 4 + (x >> 2)], last);
    }
    __syncthreads();
    uint32_t n_eff = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) n_eff = max(n_eff, s_gmax[c]);
    n_eff = min(n, n_eff);  // entries at list positions >= max n_contrib are skipped by every pixel (backward.cu:490-492)

    for (uint32_t base = 0; base < n_eff; base += GHR_BLOCK) {
        const uint32_t cnt = min((uint32_t)GHR_BLOCK, n_eff - base);
        __syncthreads();  // previous batch fully consumed
        uint32_t cm = 0;
        if ((uint32_t)tid < cnt) {
            // walk back to front: batch entry j is list position n_eff-1-(base+j)
            const uint32_t id = point_list[beg + (n_eff - 1 - (base + tid))];
            const f4* r = rec + 4 * (size_t)id;
            const f4 a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3];
            const uint32_t slot = min(rect4_slot(rects[id], tx, ty), cap - 1u);
            f4* dst = reinterpret_cast<f4*>(ginst) + 4 * (size_t)slot;  // zero the instance's gradient line
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
            s_slot[tid] = slot;
            s_r0[tid] = a0; s_r1[tid] = a1;
            s_col[0][tid] = a1.z; s_col[1][tid] = a1.w;
            s_col[2][tid] = a2.x; s_col[3][tid] = a2.y; s_col[4][tid] = a2.z; s_col[5][tid] = a2.w;
            s_col[6][tid] = a3.x; s_col[7][tid] = a3.y; s_col[8][tid] = a3.z; s_col[9][tid] = a3.w;
            // cells of the tile (c = 4 band + g, band = 4 pixel rows, g = 4 pixel columns) whose pixels the alpha >= 1/255
            // region can touch: box, then x-extent of the ellipse restricted to the band (ghr_device.h, cell_masks)
            const f4 bb = alpha_bbox(a0, a1), ep = ellipse_params(a0, a1);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const float cy0 = wy0 + 4.0f * b;
                const bool yhit = !(bb.w < cy0 || bb.z > cy0 + 3.0f);
                float lo, hi;
                ellipse_band_extent(a0.w, ep, a0.y - (cy0 + 3.0f), a0.y - cy0, lo, hi);
                const float xl = fmaxf(bb.x, a0.x - hi), xr = fminf(bb.y, a0.x - lo);
#pragma unroll
                for (int g = 0; g < 4; g++)
                    if (yhit && !(xr < wx0 + 4.0f * g || xl > wx0 + 4.0f * g + 3.0f)) cm |= 1u << (4 * b + g);
            }
        }
        s_cmask[tid] = (uint16_t)cm;
        __syncthreads();  // also orders the zero-fill (vmcnt(0) + workgroup fence) before the atomics below

        // wave `wave` owns the band of cells 4 wave .. 4 wave + 3, one cell at a time
        for (int g = 0; g < 4; g++) {
            const int cell = 4 * wave + g;
            // ---- this cell's entries of the batch, in list order, compacted into s_list[wave][0 .. n_c)
            uint32_t n_c = 0;
            // entry j sits at list position n_eff-1-(base+j); positions >= the cell's max n_contrib are dead for it
            const long long jmin = (long long)n_eff - (long long)s_gmax[cell] - (long long)base;
#pragma unroll
            for (int sub = 0; sub < 4; sub++) {
                if (64u * sub < cnt) {  // wave-uniform
                    const uint32_t e = 64u * sub + lane;
                    const bool bit = ((s_cmask[e] >> cell) & 1u) != 0u && (long long)e >= jmin;
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(bit);
                    if (bit) s_list[wave][n_c + lanes_below(mk)] = (uint8_t)e;
                    n_c += (uint32_t)__builtin_popcountll(mk);
                }
            }
            if (n_c == 0) continue;  // wave-uniform
            __builtin_amdgcn_wave_barrier();

            // ---- the cell's pixels: lane (k, m) evaluates the pixels (k, q), q = 0..3; p = 16 y + x inside the tile
            const int p0 = (16 * wave) * 4 + 4 * g + k;  // (x = 4g + k, y = 4 wave): + 16 q
            float Tin[4], PS[4];
            uint32_t last[4];
            float phiW[4];  // B operand of the colour MFMAs: dL/dpixel[m - 6] of pixel q (components 6..15 of the line)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                Tin[q] = s_T[p0 + 16 * q];
                PS[q] = s_PS[p0 + 16 * q];
                last[q] = s_last[p0 + 16 * q];
                phiW[q] = m >= 6 ? s_dL[m >= 6 ? m - 6 : 0][p0 + 16 * q] : 0.f;
            }
            // A operand of the colour-dot MFMAs: row i = m of the product is the cell pixel (x = m >> 2, y = m & 3), so
            // that lane (k, e) finds the dots of ITS pixels (k, 0..3) in its four result registers
            const int pa = 16 * (4 * wave + (m & 3)) + 4 * g + (m >> 2);
            float dLA[3];
#pragma unroll
            for (int s = 0; s < 3; s++) dLA[s] = (4 * s + k < GHR_C) ? s_dL[(4 * s + k < GHR_C) ? 4 * s + k : 0][pa] : 0.f;
            const float u = (float)(4 * g + k);                 // pixel - tile origin, x
            const float v0 = (float)(4 * wave);                 // ... y of q = 0
            // B operands of the geometry MFMAs: which line component a lane's column m receives
            const float phiSX = m == 0 ? 1.f : (m == 2 ? u : 0.f);   // a = sum_q Q dx      -> L0 (x1), L2 (x u)
            const float phi1 = m == 1 ? 1.f : 0.f;                   // a = sum_q Q dy      -> L1
            const float phi3 = m == 3 ? 1.f : 0.f;                   // a = sum_q Q dx v_q  -> L3
            const float phi4 = m == 4 ? 1.f : 0.f;                   // a = sum_q Q dy v_q  -> L4
            const float phi5 = m == 5 ? 1.f : 0.f;                   // a = sum_q Q         -> L5
            const float pxf = wx0 + u;
            const float pyf0 = wy0 + v0;

            for (uint32_t c0 = 0; c0 < n_c; c0 += 16) {
                const bool valid = c0 + m < n_c;
                const uint32_t j = valid ? (uint32_t)s_list[wave][c0 + m] : 0u;
                const f4 r0 = s_r0[j], r1 = s_r1[j];
                float col[3];
#pragma unroll
                for (int s = 0; s < 3; s++) col[s] = (4 * s + k < GHR_C) ? s_col[(4 * s + k < GHR_C) ? 4 * s + k : 0][j] : 0.f;
                // colour . dL/dpixel for the lane's four pixels
                f4 cd = {0.f, 0.f, 0.f, 0.f};
                cd = mfma16(dLA[0], col[0], cd);
                cd = mfma16(dLA[1], col[1], cd);
                cd = mfma16(dLA[2], col[2], cd);

                const uint32_t pos = n_eff - 1 - (base + j);  // 0-based list position == the reference's `contributor`
                const float o = r1.y;
                const float dx = r0.x - pxf;
                const float t1 = r0.z * dx * dx;   // unfused, source order: feeds the same discrete decisions as K7
                const float t3 = r0.w * dx;
                float alpha[4], G[4], dy[4], om[4];
                bool ct[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    dy[q] = r0.y - (pyf0 + (float)q);
                    const float power = -0.5f * (t1 + r1.x * dy[q] * dy[q]) - t3 * dy[q];
                    const float G_raw = fast_exp(power);
                    const float alpha_raw = fminf(0.99f, o * G_raw);
                    ct[q] = valid && pos < last[q] && !(power > 0.0f) && !(alpha_raw < 1.0f / 255.0f);
                    alpha[q] = ct[q] ? alpha_raw : 0.0f;
                    G[q] = ct[q] ? G_raw : 0.0f;
                    om[q] = 1.f - alpha[q];
                }
                float A0 = om[0], A1 = om[1], A2 = om[2], A3 = om[3];
                row_scan_mul4(A0, A1, A2, A3);
                const float Acum[4] = {A0, A1, A2, A3};
                float T[4], Wq[4], cdot[4], inv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    T[q] = Tin[q] * fast_rcp(Acum[q]);           // backward.cu:507, as a product over the row
                    inv[q] = fast_rcp(om[q]);
                    // a pair that does not contribute must not leak a non-finite colour of its Gaussian (0 * inf)
                    cdot[q] = ct[q] ? (q == 0 ? cd.x : (q == 1 ? cd.y : (q == 2 ? cd.z : cd.w))) : 0.0f;
                    Wq[q] = alpha[q] * T[q] * cdot[q];
                }
                float S0 = Wq[0], S1 = Wq[1], S2 = Wq[2], S3 = Wq[3], E0, E1, E2, E3, R0, R1, R2, R3;
                row_scan_add4(S0, S1, S2, S3, E0, E1, E2, E3, R0, R1, R2, R3);
                const float Ex[4] = {E0, E1, E2, E3}, Rt[4] = {R0, R1, R2, R3};
                float SQ = 0.f, SX = 0.f, SXv = 0.f, SY = 0.f, SYv = 0.f, w[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // (cdot - accum_rec . dL) T  -  T_final bg.dL / (1 - alpha)   (backward.cu:523-538)
                    const float dL_dalpha = fma_(-inv[q], PS[q] + Ex[q], cdot[q] * T[q]);
                    const float Q = G[q] * dL_dalpha;
                    const float qx = Q * dx, qy = Q * dy[q], vq = v0 + (float)q;
                    SQ += Q;
                    SX += qx;
                    SXv = fma_(qx, vq, SXv);
                    SY += qy;
                    SYv = fma_(qy, vq, SYv);
                    w[q] = alpha[q] * T[q];       // backward.cu:508,527
                    PS[q] += Rt[q];               // carried to the next chunk
                }
                float TL0 = T[0], TL1 = T[1], TL2 = T[2], TL3 = T[3];
                row_last4(TL0, TL1, TL2, TL3);
                Tin[0] = TL0; Tin[1] = TL1; Tin[2] = TL2; Tin[3] = TL3;

                // line components of the chunk's 16 entries: lane (k', c) gets component c of the entries 4k' + r
                f4 da = {0.f, 0.f, 0.f, 0.f}, db = da;
                da = mfma16(SX, phiSX, da);
                db = mfma16(SXv, phi3, db);
                da = mfma16(SY, phi1, da);
                db = mfma16(SYv, phi4, db);
                da = mfma16(SQ, phi5, da);
                db = mfma16(w[0], phiW[0], db);
                da = mfma16(w[1], phiW[1], da);
                db = mfma16(w[2], phiW[2], db);
                da = mfma16(w[3], phiW[3], da);
                const f4 d = da + db;
                // a DPP row adds one whole 64-B line per register: resolved in this XCD's L2 (only this workgroup ever
                // touches the instance's line)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t t = c0 + 4u * k + r;
                    if (t < n_c) {
                        const uint32_t slot = s_slot[s_list[wave][t]];
                        __hip_atomic_fetch_add(ginst + 16 * (size_t)slot + m, r == 0 ? d.x : (r == 1 ? d.y : (r == 2 ? d.z : d.w)),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (m == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) { s_T[p0 + 16 * q] = Tin[q]; s_PS[p0 + 16 * q] = PS[q]; }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // list entries no pixel of the tile ever reached (positions >= n_eff): their slots must read as zero
    for (uint32_t i = n_eff + tid; i < n; i += GHR_BLOCK) {
        const uint32_t id = point_list[beg + i];
        f4* dst = reinterpret_cast<f4*>(ginst) + 4 * (size_t)min(rect4_slot(rects[id], tx, ty), cap - 1u);
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
    }
#endif
}

// ---- self test of the wave primitives (tests/test_gpu_wave_primitives.py): in[8][64] -> out[12][64] ---------------------
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
