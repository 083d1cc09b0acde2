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

typedef float f2b __attribute__((ext_vector_type(2)));  // pairs of pixel rows: v_pk_mul / v_pk_add / v_pk_fma

#if defined(__HIP_DEVICE_COMPILE__)
// Which of the tile's 16 cells (c = 4 band + g: band = 4 pixel rows, g = 4 pixel columns) the alpha >= 1/255 region of a
// splat can touch: alpha_bbox + ellipse_params + ellipse_band_extent of ghr_device.h (same formulas, same margins) with
// the hardware log / sqrt / rcp instead of the correctly rounded library expansions -- the 1 % + 0.01 px (box) and
// 2 % + 0.05 / 0.02 px (ellipse) margins exceed their 1-ulp errors by four orders of magnitude, and a cull only has
// to be conservative: which pairs contribute is decided per pixel by the exact alpha test.
__device__ __forceinline__ uint32_t cell_mask16(const f4& a0, const f4& a1, float wx0, float wy0)
{
    const float BIG = 3.0e38f;
    const float o = a1.y, cx = a0.z, cy = a0.w, cz = a1.x;
    if (o < 0.999f * (1.0f / 255.0f)) return 0u;  // alpha <= o < 1/255 everywhere
    const float L = 0.6931471805599453f * __builtin_amdgcn_logf(255.0f * o);
    const float det = cx * cz - cy * cy;
    const bool pd = (L >= 1.0e-3f) && (cx > 0.0f) && (cz > 0.0f);
    float bx0 = -BIG, bx1 = BIG, by0 = -BIG, by1 = BIG;
    if (pd && det > 0.0f) {
        const float kk = 2.0f * L * fast_rcp(det);
        const float hx = 1.01f * fast_sqrt(kk * cz) + 0.01f, hy = 1.01f * fast_sqrt(kk * cx) + 0.01f;
        if (hx < BIG && hy < BIG) { bx0 = a0.x - hx; bx1 = a0.x + hx; by0 = a0.y - hy; by1 = a0.y + hy; }
    }
    float e_x = 0.f, e_det = -1.f, e_icx = 0.f, e_k = 0.f;  // ellipse_params; e_det <= 0: box only
    if (pd && det > 1.0e-4f * cx * cz && L < 100.0f) {
        const float thr = 2.04f * L + 0.05f;
        const float hx = fast_sqrt(thr * cz * fast_rcp(det));
        const float kq = cy * hx * fast_rcp(cz);
        if (hx < BIG && fabsf(kq) < BIG) { e_x = cx * thr; e_det = det; e_icx = fast_rcp(cx); e_k = kq; }
    }
    uint32_t cm = 0u;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const float cy0 = wy0 + 4.0f * b;
        const bool yhit = !(by1 < cy0 || by0 > cy0 + 3.0f);
        float lo = -BIG, hi = BIG;
        if (e_det > 0.0f) {
            const float ay = a0.y - (cy0 + 3.0f), by = a0.y - cy0;
            const float dyr = fminf(by, fmaxf(ay, -e_k)), dyl = fminf(by, fmaxf(ay, e_k));
            const float Dr = fmaxf(e_x - e_det * dyr * dyr, 0.0f), Dl = fmaxf(e_x - e_det * dyl * dyl, 0.0f);
            hi = (-cy * dyr + fast_sqrt(Dr)) * e_icx + 0.02f;
            lo = (-cy * dyl - fast_sqrt(Dl)) * e_icx - 0.02f;
        }
        const float xl = fmaxf(bx0, a0.x - hi), xr = fminf(bx1, a0.x - lo);
#pragma unroll
        for (int g = 0; g < 4; g++)
            if (yhit && !(xr < wx0 + 4.0f * g || xl > wx0 + 4.0f * g + 3.0f)) cm |= 1u << (4 * b + g);
    }
    return cm;
}
#endif

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
    __shared__ uint32_t s_next;                              // next cell of the batch nobody has taken yet

    const uint32_t tile = xcd_tile(blockIdx.x, T_tiles);
    if (tile >= T_tiles) return;  // grid padding
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane >> 4, m = lane & 15;
    const float wx0 = (float)(tx * GHR_TILE_X), wy0 = (float)(ty * GHR_TILE_Y);

    const uint32_t beg = min(tile_start[tile], cap);
    const uint32_t n = min(tile_start[tile + 1], cap) - beg;  // see k_render_bwd for `cap`

    // The list is walked back to front in batches of 256 from its END: batch entry j of the batch at `base` is list
    // position n-1-(base+j).  Which positions are dead (>= the largest n_contrib of the tile / of a cell,
    // backward.cu:490-492) is only known once the pixels are in, but the entries of the first batch do not depend on
    // it: they are requested together with the pixel data (one memory round trip for both).
    uint32_t e_id = 0u;
    f4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0, e2 = e0, e3 = e0;
    auto fetch_entry = [&](uint32_t base) {
        if (base + (uint32_t)tid < n) {
            e_id = point_list[beg + (n - 1 - (base + tid))];
            const f4* r = rec + 4 * (size_t)e_id;
            e0 = r[0]; e1 = r[1]; e2 = r[2]; e3 = r[3];
        }
    };
    fetch_entry(0u);

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
    n_eff = min(n, n_eff);  // positions >= the tile's largest n_contrib are skipped by every pixel (backward.cu:490-492)

    for (uint32_t base = 0; base < n; base += GHR_BLOCK) {
        const uint32_t cnt = min((uint32_t)GHR_BLOCK, n - base);
        if (base > 0) fetch_entry(base);
        const bool live_batch = n - base - cnt < n_eff;  // its lowest position lies below n_eff (workgroup-uniform)
        if (live_batch) __syncthreads();                 // previous batch fully consumed
        uint32_t cm = 0u;
        if ((uint32_t)tid < cnt) {
            const uint32_t slot = min(beg + (n - 1 - (base + tid)), cap - 1u);  // lines lie in list order
            f4* dst = reinterpret_cast<f4*>(ginst) + 4 * (size_t)slot;  // zero the instance's gradient line
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
            if (live_batch) {
                s_slot[tid] = slot;
                s_r0[tid] = e0; s_r1[tid] = e1;
                s_col[0][tid] = e1.z; s_col[1][tid] = e1.w;
                s_col[2][tid] = e2.x; s_col[3][tid] = e2.y; s_col[4][tid] = e2.z; s_col[5][tid] = e2.w;
                s_col[6][tid] = e3.x; s_col[7][tid] = e3.y; s_col[8][tid] = e3.z; s_col[9][tid] = e3.w;
                if (n - 1 - (base + tid) < n_eff) cm = cell_mask16(e0, e1, wx0, wy0);
            }
        }
        if (!live_batch) continue;  // every entry of the batch is dead: its lines read as zero, nothing else to do
        s_cmask[tid] = (uint16_t)cm;
        if (tid == 0) s_next = 0u;
        __syncthreads();  // also orders the zero-fill (vmcnt(0) + workgroup fence) before the atomics below

        // The 16 cells of the batch are dealt to the four waves as they become free (needle lists differ a lot between
        // the cells of a tile; a static band per wave left three waves waiting at the barrier for the longest one).
        for (;;) {
            uint32_t cell = 0u;
            if (lane == 0) cell = atomicAdd(&s_next, 1u);
            cell = (uint32_t)__builtin_amdgcn_readfirstlane((int)cell);
            if (cell >= 16u) break;
            const int band = (int)(cell >> 2), g = (int)(cell & 3u);
            // ---- this cell's entries of the batch, in list order, compacted into s_list[wave][0 .. n_c)
            uint32_t n_c = 0;
            // entry j sits at list position n-1-(base+j); positions >= the cell's max n_contrib are dead for it
            const long long jmin = (long long)n - (long long)s_gmax[cell] - (long long)base;
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
            const int p0 = 64 * band + 4 * g + k;  // (x = 4g + k, y = 4 band): + 16 q
            f2b TinA = {s_T[p0], s_T[p0 + 16]}, TinB = {s_T[p0 + 32], s_T[p0 + 48]};
            f2b PSA = {s_PS[p0], s_PS[p0 + 16]}, PSB = {s_PS[p0 + 32], s_PS[p0 + 48]};
            const uint32_t last0 = s_last[p0], last1 = s_last[p0 + 16], last2 = s_last[p0 + 32], last3 = s_last[p0 + 48];
            // B operand of the colour MFMAs: dL/dpixel[m - 6] of pixel q (components 6..15 of the line)
            const int mc = m >= 6 ? m - 6 : 0;
            const float phiW0 = m >= 6 ? s_dL[mc][p0] : 0.f, phiW1 = m >= 6 ? s_dL[mc][p0 + 16] : 0.f;
            const float phiW2 = m >= 6 ? s_dL[mc][p0 + 32] : 0.f, phiW3 = m >= 6 ? s_dL[mc][p0 + 48] : 0.f;
            // A operand of the colour-dot MFMAs: row i = m of the product is the cell pixel (x = m >> 2, y = m & 3), so
            // that lane (k, e) finds the dots of ITS pixels (k, 0..3) in its four result registers
            const int pa = 16 * (4 * band + (m & 3)) + 4 * g + (m >> 2);
            const float dLA0 = s_dL[k][pa], dLA1 = s_dL[4 + k][pa], dLA2 = k < 2 ? s_dL[k < 2 ? 8 + k : 0][pa] : 0.f;
            const float u = (float)(4 * g + k);                 // pixel - tile origin, x
            const float v0 = (float)(4 * band);                 // ... y of q = 0
            const f2b vA = {v0, v0 + 1.f}, vB = {v0 + 2.f, v0 + 3.f};
            // B operands of the geometry MFMAs: which line component a lane's column m receives
            const float phiSX = m == 0 ? 1.f : (m == 2 ? u : 0.f);   // a = sum_q Q dx      -> L0 (x1), L2 (x u)
            const float phi1 = m == 1 ? 1.f : 0.f;                   // a = sum_q Q dy      -> L1
            const float phi3 = m == 3 ? 1.f : 0.f;                   // a = sum_q Q dx v_q  -> L3
            const float phi4 = m == 4 ? 1.f : 0.f;                   // a = sum_q Q dy v_q  -> L4
            const float phi5 = m == 5 ? 1.f : 0.f;                   // a = sum_q Q         -> L5
            const float pxf = wx0 + u;
            const f2b pyA = {wy0 + v0, wy0 + v0 + 1.f}, pyB = {wy0 + v0 + 2.f, wy0 + v0 + 3.f};
            const int kc2 = k < 2 ? 8 + k : 0;

            // entry of the next chunk, requested one chunk ahead (its LDS round trips hide behind the arithmetic)
            bool nv = m < n_c;
            uint32_t nj = nv ? (uint32_t)s_list[wave][m] : 0u;
            f4 nr0 = s_r0[nj], nr1 = s_r1[nj];
            float nc0 = s_col[k][nj], nc1 = s_col[4 + k][nj], nc2 = k < 2 ? s_col[kc2][nj] : 0.f;
            for (uint32_t c0 = 0; c0 < n_c; c0 += 16) {
                const bool valid = nv;
                const uint32_t j = nj;
                const f4 r0 = nr0, r1 = nr1;
                const float col0 = nc0, col1 = nc1, col2 = nc2;
                if (c0 + 16 < n_c) {  // wave-uniform
                    nv = c0 + 16 + m < n_c;
                    nj = nv ? (uint32_t)s_list[wave][c0 + 16 + m] : 0u;
                    nr0 = s_r0[nj]; nr1 = s_r1[nj];
                    nc0 = s_col[k][nj]; nc1 = s_col[4 + k][nj]; nc2 = k < 2 ? s_col[kc2][nj] : 0.f;
                }
                // colour . dL/dpixel for the lane's four pixels
                f4 cd = {0.f, 0.f, 0.f, 0.f};
                cd = mfma16(dLA0, col0, cd);
                cd = mfma16(dLA1, col1, cd);
                cd = mfma16(dLA2, col2, cd);

                const uint32_t pos = n - 1 - (base + j);  // 0-based list position == the reference's `contributor`
                const float o = r1.y;
                const float dx = r0.x - pxf;
                const float t1 = r0.z * dx * dx;   // unfused, source order: feeds the same discrete decisions as K7
                const float t3 = r0.w * dx;
                const f2b dyA = r0.y - pyA, dyB = r0.y - pyB;
                const f2b pwA = -0.5f * (t1 + r1.x * dyA * dyA) - t3 * dyA, pwB = -0.5f * (t1 + r1.x * dyB * dyB) - t3 * dyB;
                const f2b eA = pwA * 1.4426950408889634f, eB = pwB * 1.4426950408889634f;
                const f2b GrA = {__builtin_amdgcn_exp2f(eA.x), __builtin_amdgcn_exp2f(eA.y)};
                const f2b GrB = {__builtin_amdgcn_exp2f(eB.x), __builtin_amdgcn_exp2f(eB.y)};
                const f2b oA = o * GrA, oB = o * GrB;
                const float ar0 = fminf(0.99f, oA.x), ar1 = fminf(0.99f, oA.y), ar2 = fminf(0.99f, oB.x), ar3 = fminf(0.99f, oB.y);
                const bool ct0 = valid && pos < last0 && !(pwA.x > 0.0f) && !(ar0 < 1.0f / 255.0f);
                const bool ct1 = valid && pos < last1 && !(pwA.y > 0.0f) && !(ar1 < 1.0f / 255.0f);
                const bool ct2 = valid && pos < last2 && !(pwB.x > 0.0f) && !(ar2 < 1.0f / 255.0f);
                const bool ct3 = valid && pos < last3 && !(pwB.y > 0.0f) && !(ar3 < 1.0f / 255.0f);
                const f2b alA = {ct0 ? ar0 : 0.f, ct1 ? ar1 : 0.f}, alB = {ct2 ? ar2 : 0.f, ct3 ? ar3 : 0.f};
                const f2b GA = {ct0 ? GrA.x : 0.f, ct1 ? GrA.y : 0.f}, GB = {ct2 ? GrB.x : 0.f, ct3 ? GrB.y : 0.f};
                const f2b omA = 1.f - alA, omB = 1.f - alB;
                // 1 / (1 - alpha), and its running product over the row: T_i = T_in prod_{j<=i} 1/(1 - alpha_j)  (:507)
                const f2b invA = {fast_rcp(omA.x), fast_rcp(omA.y)}, invB = {fast_rcp(omB.x), fast_rcp(omB.y)};
                float A0 = invA.x, A1 = invA.y, A2 = invB.x, A3 = invB.y;
                row_scan_mul4(A0, A1, A2, A3);
                const f2b TA = TinA * f2b{A0, A1}, TB = TinB * f2b{A2, A3};
                // a pair that does not contribute must not leak a non-finite colour of its Gaussian (0 * inf)
                const f2b cdA = {ct0 ? cd.x : 0.f, ct1 ? cd.y : 0.f}, cdB = {ct2 ? cd.z : 0.f, ct3 ? cd.w : 0.f};
                const f2b wA = alA * TA, wB = alB * TB;       // backward.cu:508,527
                const f2b WA = wA * cdA, WB = wB * cdB;
                float S0 = WA.x, S1 = WA.y, S2 = WB.x, S3 = WB.y, E0, E1, E2, E3, R0, R1, R2, R3;
                row_scan_add4(S0, S1, S2, S3, E0, E1, E2, E3, R0, R1, R2, R3);
                // (cdot - accum_rec . dL) T  -  T_final bg.dL / (1 - alpha)   (backward.cu:523-538)
                const f2b dLdaA = __builtin_elementwise_fma(-invA, PSA + f2b{E0, E1}, cdA * TA);
                const f2b dLdaB = __builtin_elementwise_fma(-invB, PSB + f2b{E2, E3}, cdB * TB);
                const f2b QA = GA * dLdaA, QB = GB * dLdaB;
                const f2b qxA = QA * dx, qxB = QB * dx, qyA = QA * dyA, qyB = QB * dyB;
                const f2b sq = QA + QB, sx = qxA + qxB, sy = qyA + qyB;
                const f2b sxv = __builtin_elementwise_fma(qxA, vA, qxB * vB), syv = __builtin_elementwise_fma(qyA, vA, qyB * vB);
                const float SQ = sq.x + sq.y, SX = sx.x + sx.y, SY = sy.x + sy.y, SXv = sxv.x + sxv.y, SYv = syv.x + syv.y;
                PSA += f2b{R0, R1};  // carried to the next chunk
                PSB += f2b{R2, R3};
                float TL0 = TA.x, TL1 = TA.y, TL2 = TB.x, TL3 = TB.y;
                row_last4(TL0, TL1, TL2, TL3);
                TinA = f2b{TL0, TL1};
                TinB = f2b{TL2, TL3};

                // line components of the chunk's 16 entries: lane (k', c) gets component c of the entries 4k' + r
                f4 da = {0.f, 0.f, 0.f, 0.f}, db = da;
                da = mfma16(SX, phiSX, da);
                db = mfma16(SXv, phi3, db);
                da = mfma16(SY, phi1, da);
                db = mfma16(SYv, phi4, db);
                da = mfma16(SQ, phi5, da);
                db = mfma16(wA.x, phiW0, db);
                da = mfma16(wA.y, phiW1, da);
                db = mfma16(wB.x, phiW2, db);
                da = mfma16(wB.y, phiW3, da);
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
                s_T[p0] = TinA.x; s_T[p0 + 16] = TinA.y; s_T[p0 + 32] = TinB.x; s_T[p0 + 48] = TinB.y;
                s_PS[p0] = PSA.x; s_PS[p0 + 16] = PSA.y; s_PS[p0 + 32] = PSB.x; s_PS[p0 + 48] = PSB.y;
            }
            __builtin_amdgcn_wave_barrier();
        }
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
