// ghr_render_bwd3.h -- K8, "cell list" form: the scan arithmetic of ghr_render_bwd2.h without its tile pipeline.
// Follows R:cuda_rasterizer/backward.cu:403-561 (renderCUDA).
//
// k_render_bwd_scan spends more than half of its time outside the 16-entry chunk loop (profiles/r02a_k8_scan_phase_
// profile.txt): staging whole 256-entry batches of the tile with the cull arithmetic, loading the tile's pixels through
// LDS, barriers between the phases.  Everything the cull decides is already known to the forward pass, so here
//   * k_render_fwd leaves, per 64 list positions and cell, the 64-bit mask of the entries that can touch the cell, and
//     per cell its largest n_contrib (2 B per instance + 64 B per tile of extra forward writes);
//   * a wave takes one 4x4 cell of the tile for its WHOLE list (cells are dealt to the four waves of the tile's
//     workgroup as they finish): it expands the cell's mask words into the list of hit positions, gathers those
//     entries -- and only those -- 64 at a time into wave-private LDS planes, and runs the chunk loop on them;
//   * the cell's pixels (T_final, n_contrib, dL/dpixel) go straight from global memory into the lanes that use them;
//     the background term of backward.cu:535-538 (bg . dL/dpixel) comes out of the same three MFMAs as the colour
//     dots, with the background as the "colour";
//   * the only workgroup barrier is the one between zero-filling the tile's gradient lines and the first atomic; the
//     same prologue leaves the tile's Gaussian ids and mask words in LDS, so that a cell costs ONE global round trip
//     (its records; its pixels travel at the same time).
// Loads are written so that the compiler can keep them in flight together: 32-bit offsets from uniform bases (one
// address register), no load under a lane-dependent branch (the wait for it would sit at the join and drain every
// outstanding atomic as well), selects deferred to the first use.
// The arithmetic of a chunk is that of k_render_bwd_scan (same decisions, same closed forms): see there.
#pragma once
#include "ghr_device.h"
#include "ghr_render_bwd2.h"

namespace ghr {

#define GHR_B3_SEG_WORDS 8                       // mask words (of 64 list positions) expanded at a time
#define GHR_B3_LIST (64 * GHR_B3_SEG_WORDS)      // ... hence at most this many hits per segment
#define GHR_B3_ROUND 32                          // entries gathered into the wave's planes at a time (two chunks)
#define GHR_B3_PLANE 48                          // floats per staged plane: rows k / k+1 read 16 banks apart
#define GHR_B3_NPLANE 17                         // x y a b c o | 10 colours | slot
#define GHR_B3_CACHE 1024                        // tiles with at most this many instances keep ids + masks in LDS
#define GHR_B3_CWORDS (GHR_B3_CACHE / 64)
#ifndef GHR_B3_WAVES
#define GHR_B3_WAVES 5                           // waves per SIMD the register budget is set for
#endif

// Largest sizes the 32-bit offsets cover (bytes < 4 GiB): checked on the host, which falls back to k_render_bwd
GHR_HD bool b3_fits(size_t rows, size_t R, size_t W, size_t H)
{
    const size_t lim = (size_t)1 << 32;
    return rows * 64 < lim && R * 64 < lim && W * H * GHR_C * 4 < lim;
}

#if defined(__HIP_DEVICE_COMPILE__)
// base + 32-bit BYTE offset: the uniform base stays in scalar registers, the lane's address is one VGPR
__device__ __forceinline__ float uniform_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

template <typename T>
__device__ __forceinline__ T ld32(const T* base, uint32_t byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

struct B3Shared {
    uint32_t id[GHR_B3_CACHE];                         // tile: Gaussian of each list position (tiles of <= GHR_B3_CACHE)
    unsigned long long mask[GHR_B3_CWORDS][16];        // tile: mask words, [word][cell]
    uint32_t clast[16];                                // tile: largest n_contrib of each cell
    uint16_t list[4][GHR_B3_LIST];                     // per wave: hit positions of the segment, ascending
    float e[4][GHR_B3_NPLANE][GHR_B3_PLANE];           // per wave: the round's entries, one plane per field
    uint32_t next;                                     // next cell of the tile nobody has taken yet
};

// The cells of one tile.  SMALL: the tile's ids and mask words are in LDS (n <= GHR_B3_CACHE); otherwise they are read
// from global memory where needed (one more round trip per round; dense tiles have long lists to amortise it).
template <bool SMALL>
__device__ __forceinline__ void b3_tile(B3Shared& sh, int W, int H, int tx, int ty, uint32_t tile, uint32_t beg, uint32_t n,
                                        const uint32_t* __restrict__ point_list, const f4* __restrict__ rec,
                                        const float* __restrict__ bg, const float* __restrict__ final_T,
                                        const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                                        const rect4* __restrict__ rects, float* ginst, uint32_t cap,
                                        const unsigned long long* __restrict__ cell_mask, size_t word0)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane >> 4, m = lane & 15;
    const float wx0 = (float)(tx * GHR_TILE_X), wy0 = (float)(ty * GHR_TILE_Y);
    GHR_PROF_DECL;
    const uint32_t plane = 4u * (uint32_t)(W * H);  // bytes
    float (*E)[GHR_B3_PLANE] = sh.e[wave];
    uint16_t* list = sh.list[wave];
    const int kc2 = k < 2 ? 8 + k : 0;
    // B operands of the geometry MFMAs that do not depend on the cell: which line component column m receives
    const float phi1 = m == 1 ? 1.f : 0.f;                   // a = sum_q Q dy      -> L1
    const float phi3 = m == 3 ? 1.f : 0.f;                   // a = sum_q Q dx v_q  -> L3
    const float phi4 = m == 4 ? 1.f : 0.f;                   // a = sum_q Q dy v_q  -> L4
    const float phi5 = m == 5 ? 1.f : 0.f;                   // a = sum_q Q         -> L5
#ifdef GHR_B3_NOATOM
    float abl = 0.f;
#endif

    for (;;) {
        uint32_t cell = 0u;
        if (lane == 0) cell = atomicAdd(&sh.next, 1u);
        cell = (uint32_t)__builtin_amdgcn_readfirstlane((int)cell);
        if (cell >= 16u) break;
        // positions at or beyond the cell's largest n_contrib are dead for all its pixels (backward.cu:490-492)
        const uint32_t gm = min(sh.clast[cell], n);
        if (gm == 0) continue;  // wave-uniform
        const int band = (int)(cell >> 2), g = (int)(cell & 3u);
        GHR_PROF(0);

        // ---- the cell's pixels: lane (k, m) evaluates the pixels (x = 4g + k, y = 4 band + q), q = 0..3.
        // Loads from clamped addresses, selects later (`fresh` below): these travel with the first round's gather.
        // (hk, hm: k and m made opaque per cell, so that the address arithmetic below is redone here instead of being
        // hoisted out of the cell loop into registers that live -- or spill -- across the chunk loop)
        int hk = k, hm = m;
        asm volatile("" : "+v"(hk), "+v"(hm));
        const int px = tx * GHR_TILE_X + 4 * g + hk, py0 = ty * GHR_TILE_Y + 4 * band;
        float Tf[4], phiW[4];
        uint32_t last[4];
        const uint32_t pxc = (uint32_t)min(px, W - 1);
        const uint32_t hmc = hm >= 6 ? (uint32_t)(hm - 6) : 0u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t pix = 4u * ((uint32_t)W * (uint32_t)min(py0 + q, H - 1) + pxc);  // byte offset
            Tf[q] = ld32(final_T, pix);
            last[q] = ld32(n_contrib, pix);
            phiW[q] = ld32(dL_dpix, hmc * plane + pix);  // B operand of the colour MFMAs: dL/dpixel[m - 6] of pixel q
        }
        // A operand of the colour-dot MFMAs: row i = m of the product is the cell pixel (x = m >> 2, y = m & 3), so
        // that lane (k, e) finds the dots of ITS pixels (k, 0..3) in its four result registers
        const int ax = tx * GHR_TILE_X + 4 * g + (hm >> 2), ay = py0 + (hm & 3);
        const uint32_t hkc2 = hk < 2 ? (uint32_t)(8 + hk) : 0u;
        float dLA0, dLA1, dLA2;
        {
            const uint32_t pix = 4u * ((uint32_t)W * (uint32_t)min(ay, H - 1) + (uint32_t)min(ax, W - 1));
            dLA0 = ld32(dL_dpix, (uint32_t)hk * plane + pix);
            dLA1 = ld32(dL_dpix, (uint32_t)(4 + hk) * plane + pix);
            dLA2 = ld32(dL_dpix, hkc2 * plane + pix);
        }
        // the background, as the "colour" of one more splat behind the list (re-read per cell: three registers less to
        // carry through the chunk loop)
        float bgA0 = ld32(bg, 4u * (uint32_t)hk), bgA1 = ld32(bg, 4u * (uint32_t)(4 + hk)), bgA2 = ld32(bg, 4u * hkc2);
        const float u = (float)(4 * g + k);                 // pixel - tile origin, x
        // wave-uniform values are pinned to scalar registers (results of float VALU ops are not, on their own)
        const float v0 = uniform_f((float)(4 * band));      // ... y of q = 0
        const float phiSX = m == 0 ? 1.f : (m == 2 ? u : 0.f);   // a = sum_q Q dx -> L0 (x1), L2 (x u)
        const float pxf = wx0 + u;
        const float py0f = uniform_f((float)py0), py1f = uniform_f((float)(py0 + 1)), py2f = uniform_f((float)(py0 + 2)),
                    py3f = uniform_f((float)(py0 + 3));  // exact integers, like K7's pyf

        f2b TinA = {0.f, 0.f}, TinB = TinA, PSA = TinA, PSB = TinA;
        bool fresh = true;  // the pixel loads above are first needed after the first round's gather has been issued

        // ---- segments of GHR_B3_SEG_WORDS mask words, from the back of the list
        for (int w_hi = (int)((gm - 1u) >> 6); w_hi >= 0; w_hi -= GHR_B3_SEG_WORDS) {
            const int w_lo = max(w_hi - (GHR_B3_SEG_WORDS - 1), 0);
            uint32_t n_c = 0;
            GHR_PROF(1);
            for (int w = w_lo; w <= w_hi; w++) {
                const unsigned long long mk = SMALL ? sh.mask[w][cell] : cell_mask[(word0 + w) * 16 + cell];  // wave-uniform
                const uint32_t pos = 64u * w + lane;
                const bool bit = ((mk >> lane) & 1ull) != 0ull && pos < gm;
                const unsigned long long bm = __builtin_amdgcn_ballot_w64(bit);
                if (bit) list[n_c + lanes_below(bm)] = (uint16_t)(pos - 64u * w_lo);
                n_c += (uint32_t)__builtin_popcountll(bm);
            }
            __builtin_amdgcn_wave_barrier();
            GHR_PROF(2);

            // ---- rounds of GHR_B3_ROUND entries from the top of the segment's list; staged index e = list index
            //      top-1-e, so that ascending staged index = back to front.  Two lanes gather one 64-B record.
            for (uint32_t top = n_c; top > 0; top -= min(top, (uint32_t)GHR_B3_ROUND)) {
                const uint32_t cnt = min(top, (uint32_t)GHR_B3_ROUND);
                {
                    const uint32_t e = min((uint32_t)lane >> 1, cnt - 1u), h = (uint32_t)lane & 1u;
                    const uint32_t pos = 64u * w_lo + list[top - 1u - e];
                    const uint32_t id = SMALL ? sh.id[pos] : ld32(point_list, 4u * (beg + pos));
                    const f4 ea = ld32(rec, 64u * id + 32u * h), eb = ld32(rec, 64u * id + 32u * h + 16u);
                    const rect4 rc = ld32(rects, 16u * id);
                    float* dst = &E[8 * h][e];  // lanes past the end rewrite the last entry with the same values
                    dst[0] = ea.x; dst[GHR_B3_PLANE] = ea.y; dst[2 * GHR_B3_PLANE] = ea.z; dst[3 * GHR_B3_PLANE] = ea.w;
                    dst[4 * GHR_B3_PLANE] = eb.x; dst[5 * GHR_B3_PLANE] = eb.y; dst[6 * GHR_B3_PLANE] = eb.z;
                    dst[7 * GHR_B3_PLANE] = eb.w;
                    E[16][e] = __uint_as_float(min(rect4_slot(rc, tx, ty), cap - 1u));
                }
                __builtin_amdgcn_wave_barrier();
                if (fresh) {  // wave-uniform
                    fresh = false;
#pragma unroll
                    for (int q = 0; q < 4; q++) {  // pixels outside the image take no part (n_contrib 0, no gradient)
                        const bool in = px < W && py0 + q < H;
                        Tf[q] = in ? Tf[q] : 0.f;
                        last[q] = in ? last[q] : 0u;
                        phiW[q] = (in && m >= 6) ? phiW[q] : 0.f;
                    }
                    {
                        const bool in = ax < W && ay < H;
                        dLA0 = in ? dLA0 : 0.f;
                        dLA1 = in ? dLA1 : 0.f;
                        dLA2 = (in && k < 2) ? dLA2 : 0.f;
                        bgA2 = k < 2 ? bgA2 : 0.f;
                    }
                    // backward.cu:535-538: the background term enters like one more splat behind the list; bg . dL/dpixel
                    // of the lane's four pixels from the same three MFMAs as the colour dots
                    f4 bd = {0.f, 0.f, 0.f, 0.f};
                    bd = mfma16(dLA0, bgA0, bd);
                    bd = mfma16(dLA1, bgA1, bd);
                    bd = mfma16(dLA2, bgA2, bd);
                    TinA = f2b{Tf[0], Tf[1]}; TinB = f2b{Tf[2], Tf[3]};
                    PSA = f2b{Tf[0] * bd.x, Tf[1] * bd.y}; PSB = f2b{Tf[2] * bd.z, Tf[3] * bd.w};
                }
                GHR_PROF(3);
                GHR_PROF_COUNT(7, (cnt + 15) / 16);

                for (uint32_t c0 = 0; c0 < cnt; c0 += 16) {
                    const bool valid = c0 + m < cnt;
                    const uint32_t j = min(c0 + m, cnt - 1u);  // lanes past the end recompute the last entry, masked
                    const float ex = E[0][j], ey = E[1][j], ca = E[2][j], cb = E[3][j], cc = E[4][j], o = E[5][j];
                    const float col0 = E[6 + k][j], col1 = E[10 + k][j], col2 = k < 2 ? E[6 + kc2][j] : 0.f;
                    const uint32_t pos = 64u * w_lo + list[top - 1u - j];
                    // colour . dL/dpixel for the lane's four pixels
                    f4 cd = {0.f, 0.f, 0.f, 0.f};
                    cd = mfma16(dLA0, col0, cd);
                    cd = mfma16(dLA1, col1, cd);
                    cd = mfma16(dLA2, col2, cd);

                    const float dx = ex - pxf;
                    const float t1 = ca * dx * dx;   // unfused, source order: feeds the same discrete decisions as K7
                    const float t3 = cb * dx;
                    const f2b dyA = {ey - py0f, ey - py1f}, dyB = {ey - py2f, ey - py3f};
                    const f2b pwA = -0.5f * (t1 + cc * dyA * dyA) - t3 * dyA, pwB = -0.5f * (t1 + cc * dyB * dyB) - t3 * dyB;
                    const f2b eA = pwA * 1.4426950408889634f, eB = pwB * 1.4426950408889634f;
                    const f2b GrA = {__builtin_amdgcn_exp2f(eA.x), __builtin_amdgcn_exp2f(eA.y)};
                    const f2b GrB = {__builtin_amdgcn_exp2f(eB.x), __builtin_amdgcn_exp2f(eB.y)};
                    const f2b oA = o * GrA, oB = o * GrB;
                    const float ar0 = fminf(0.99f, oA.x), ar1 = fminf(0.99f, oA.y), ar2 = fminf(0.99f, oB.x), ar3 = fminf(0.99f, oB.y);
                    const bool ct0 = valid && pos < last[0] && !(pwA.x > 0.0f) && !(ar0 < 1.0f / 255.0f);
                    const bool ct1 = valid && pos < last[1] && !(pwA.y > 0.0f) && !(ar1 < 1.0f / 255.0f);
                    const bool ct2 = valid && pos < last[2] && !(pwB.x > 0.0f) && !(ar2 < 1.0f / 255.0f);
                    const bool ct3 = valid && pos < last[3] && !(pwB.y > 0.0f) && !(ar3 < 1.0f / 255.0f);
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
                    const float SQ = sq.x + sq.y, SX = sx.x + sx.y, SY = sy.x + sy.y;
                    // sum_q (Q d)_q v_q with v_q = v0 + q
                    const float SXv = fma_(v0, SX, fma_(3.f, qxB.y, fma_(2.f, qxB.x, qxA.y)));
                    const float SYv = fma_(v0, SY, fma_(3.f, qyB.y, fma_(2.f, qyB.x, qyA.y)));
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
                    db = mfma16(wA.x, phiW[0], db);
                    da = mfma16(wA.y, phiW[1], da);
                    db = mfma16(wB.x, phiW[2], db);
                    da = mfma16(wB.y, phiW[3], da);
                    const f4 d = da + db;
                    // a DPP row adds one whole 64-B line per register: resolved in this XCD's L2 (only this workgroup
                    // ever touches the instance's line)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint32_t t = c0 + 4u * k + r;
                        if (t < cnt) {
                            const uint32_t slot = __float_as_uint(E[16][t]);
                            const float val = r == 0 ? d.x : (r == 1 ? d.y : (r == 2 ? d.z : d.w));
#ifdef GHR_B3_NOATOM  // ablation: the arithmetic stays alive, the memory operation goes
                            abl += val * (float)(slot & 1u);
#else
                            __hip_atomic_fetch_add(reinterpret_cast<float*>(reinterpret_cast<char*>(ginst) + (64u * slot + 4u * (uint32_t)m)),
                                                   val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();  // the round's planes are consumed before the next round overwrites them
                GHR_PROF(4);
            }
        }
        // No load may stay pending across the cell loop: the waits the compiler places for a value that is "maybe still in
        // flight" on some path are counted conservatively past the conditional atomics, i.e. they drain those too.
        if (fresh) {  // no entry of the cell was in reach: retire the pixel loads here
            asm volatile("" :: "v"(bgA0), "v"(bgA1), "v"(bgA2));
            asm volatile("" :: "v"(Tf[0]), "v"(Tf[1]), "v"(Tf[2]), "v"(Tf[3]), "v"(last[0]), "v"(last[1]), "v"(last[2]),
                         "v"(last[3]), "v"(phiW[0]), "v"(phiW[1]), "v"(phiW[2]), "v"(phiW[3]), "v"(dLA0), "v"(dLA1), "v"(dLA2));
        }
    }
#ifdef GHR_B3_NOATOM
    if (abl == 12345.678f) ginst[tid] = abl;
#endif
    GHR_PROF(5);
    GHR_PROF_END(6);
}
#endif

__global__ void __launch_bounds__(GHR_BLOCK, GHR_B3_WAVES) k_render_bwd_cells(int W, int H, int gx, uint32_t T_tiles,
                                                                const uint32_t* __restrict__ tile_start,
                                                                const uint32_t* __restrict__ point_list,
                                                                const f4* __restrict__ rec, const float* __restrict__ bg,
                                                                const float* __restrict__ final_T,
                                                                const uint32_t* __restrict__ n_contrib,
                                                                const float* __restrict__ dL_dpix,
                                                                const rect4* __restrict__ rects, float* ginst,
                                                                uint32_t cap,
                                                                const unsigned long long* __restrict__ cell_mask,
                                                                const uint32_t* __restrict__ cell_last)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ B3Shared sh;
    const uint32_t tile = xcd_tile(blockIdx.x, T_tiles);
    if (tile >= T_tiles) return;  // grid padding
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x;
    const uint32_t beg = min(tile_start[tile], cap);
    const uint32_t n = min(tile_start[tile + 1], cap) - beg;  // see k_render_bwd for `cap`
    if (n == 0) return;
    const size_t word0 = mask_word0(beg, tile);
    const bool small = n <= GHR_B3_CACHE;

    // ---- tile prologue: zero the gradient lines of all the tile's instances (k_geom_bwd / k_project_bwd read every
    //      line of a Gaussian) and bring what every cell needs from the lists into LDS
    for (uint32_t i = tid; i < n; i += GHR_BLOCK) {
        const uint32_t id = point_list[beg + i];
        if (small) sh.id[i] = id;
        const uint32_t slot = min(rect4_slot(rects[id], tx, ty), cap - 1u);
        f4* dst = reinterpret_cast<f4*>(ginst) + 4u * slot;
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
    }
    if (small && (uint32_t)tid < 16u * ((n + 63u) >> 6)) (&sh.mask[0][0])[tid] = cell_mask[word0 * 16 + tid];
    if (tid < 16) sh.clast[tid] = cell_last[16u * tile + tid];
    if (tid == 0) sh.next = 0u;
    __syncthreads();  // orders the zero-fill (vmcnt(0) + workgroup fence) before the atomics below

    if (small)
        b3_tile<true>(sh, W, H, tx, ty, tile, beg, n, point_list, rec, bg, final_T, n_contrib, dL_dpix, rects, ginst, cap,
                      cell_mask, word0);
    else
        b3_tile<false>(sh, W, H, tx, ty, tile, beg, n, point_list, rec, bg, final_T, n_contrib, dL_dpix, rects, ginst, cap,
                       cell_mask, word0);
#endif
}

}  // namespace ghr
