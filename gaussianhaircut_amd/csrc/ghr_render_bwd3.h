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
//   * the records of a chunk's 16 entries -- 16 x 64 B -- are ONE global_load_lds_dwordx4 per wave (gfx950: 16 B per
//     lane straight into LDS, four lanes per record): no registers, no LDS writes, no address arithmetic beyond the
//     record offset.  Three 1-KB buffers per wave: the gather of chunk t+2 is issued while chunk t is worked on.  The
//     compiler does not order LDS reads behind such loads; the waits are written out below, and for s_waitcnt vmcnt(N)
//     to mean "the load of chunk t has landed" the number of memory operations issued after it must be known exactly:
//     every chunk issues one gather (if there is one left) and FOUR atomics, always -- rows that have no entry add 0.0
//     to a line of the chunk -- and nothing else (no spills: the kernel must compile without scratch).
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
#define GHR_B3_CACHE 2048                        // tiles with at most this many instances keep ids and masks in LDS
#define GHR_B3_CWORDS (GHR_B3_CACHE / 64)
#define GHR_B3_NBUF 3                            // gather buffers per wave (chunk t, t+1, t+2)
#define GHR_B3_WAVES 5                           // waves per SIMD the register budget is set for
#define GHR_B3_NW 4                              // waves of a tile's workgroup (they share the tile's 16 cells)
#define GHR_B3_THREADS (64 * GHR_B3_NW)

// Largest sizes the 32-bit offsets cover (bytes < 4 GiB): checked on the host, which falls back to k_render_bwd
GHR_HD bool b3_fits(size_t rows, size_t R, size_t W, size_t H)
{
    const size_t lim = (size_t)1 << 32;
    return rows * 64 < lim && R * 64 < lim && W * H * GHR_C * 4 < lim;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float uniform_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// base + 32-bit BYTE offset: the uniform base stays in scalar registers, the lane's address is one VGPR
template <typename T>
__device__ __forceinline__ T ld32(const T* base, uint32_t byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
// ... for data a workgroup reads once (the cell's pixels)
template <typename T>
__device__ __forceinline__ T ld32_once(const T* base, uint32_t byte_off)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// cache policy bits of the record gathers (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#define GHR_B3_GATHER_AUX 0
// 16 B per lane from `base + byte_off` to `lds_dst + 16 * lane` (lds_dst wave-uniform), asynchronously: counted in vmcnt
__device__ __forceinline__ void gather16_to_lds(const void* base, uint32_t byte_off, void* lds_dst)
{
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + byte_off),
        (__attribute__((address_space(3))) void*)lds_dst, 16, 0, GHR_B3_GATHER_AUX);
}
#define GHR_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// LDS byte address of an object in __shared__ memory
template <typename T>
__device__ __forceinline__ uint32_t lds_addr(T* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p;
}
// atomicAdd(lds word, 1) by lane 0 only, result broadcast.  Written in asm: in front of the C++ atomic the compiler puts
// s_waitcnt vmcnt(0) (it cannot tell the LDS-DMA gathers in flight from the word), which drained the wave's outstanding
// line atomics and gathers at every cell draw.
// Only for WAVE-UNIFORM control flow with lane 0 active (the cell loop of b3_tile: whole waves): the atomic is issued by
// lane 0 under a hand-set EXEC and its result is read back from lane 0.
__device__ __forceinline__ uint32_t lds_draw(uint32_t addr)
{
    // (the address arrives in a scalar register and the result register doubles as the address operand: nothing of the
    // draw is live in a VGPR across a cell)
    uint32_t ret, one = 1u;
    unsigned long long saved;
    const uint32_t addr_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)addr);
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "v_mov_b32 %0, %2\n\t"
                 "ds_add_rtn_u32 %0, %0, %3\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(ret), "=&s"(saved)
                 : "s"(addr_s), "v"(one));   // (no "memory": the word is touched by nothing else between two barriers)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)ret);
}

struct B3Shared {
    uint32_t id[GHR_B3_CACHE];                         // tile: Gaussian of each list position (tiles of <= GHR_B3_CACHE)
    unsigned long long mask[GHR_B3_CWORDS][16];        // tile: mask words, [word][cell]
    uint32_t clast[16];                                // tile: largest n_contrib of each cell
    uint16_t list[GHR_B3_NW][GHR_B3_LIST];                     // per wave: hit positions of the segment, ascending
    f4 rec[GHR_B3_NW][GHR_B3_NBUF][64];                        // per wave: gathered records, 16 entries x 64 B per buffer
    uint32_t cslot[GHR_B3_NW][GHR_B3_NBUF][16];                // per wave: ... and their gradient lines
    uint32_t next;                                     // next cell of the tile nobody has taken yet
#ifdef GHR_B3_PAD_LDS
    char pad[GHR_B3_PAD_LDS];                          // experiment knob: workgroups per CU as a function of the LDS footprint
#endif
};

// The cells of one tile.  SMALL: the tile's ids and mask words are in LDS (n <= GHR_B3_CACHE) and the gather runs
// two chunks ahead; otherwise they are fetched from global memory chunk by chunk and nothing is overlapped (dense tiles
// of more than GHR_B3_CACHE instances: correct, not fast).
template <bool SMALL>
__device__ __forceinline__ void b3_tile(B3Shared& sh, int W, int H, int tx, int ty, uint32_t tile, uint32_t beg, uint32_t n,
                                        const uint32_t* __restrict__ point_list, const f4* __restrict__ rec,
                                        const float* __restrict__ bg, const float* __restrict__ final_T,
                                        const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                                        const rect4* __restrict__ rects, float* ginst, uint32_t cap,
                                        const unsigned long long* __restrict__ cell_mask, size_t word0)
{
    // (wave: pinned to a scalar register -- the per-wave LDS bases then are scalar too instead of lane-constant VGPRs)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), k = lane >> 4, m = lane & 15;
    const float wx0 = (float)(tx * GHR_TILE_X);
    GHR_PROF_DECL;
    const uint32_t plane = 4u * (uint32_t)(W * H);  // bytes
    uint16_t* list = sh.list[wave];
    const int kc2 = k < 2 ? 8 + k : 0;
    // B operands of the geometry MFMAs that do not depend on the cell: which line component column m receives
    const float phi1 = m == 1 ? 1.f : 0.f;                   // a = sum_q Q dy      -> L1
    const float phi3 = m == 3 ? 1.f : 0.f;                   // a = sum_q Q dx v_q  -> L3
    const float phi4 = m == 4 ? 1.f : 0.f;                   // a = sum_q Q dy v_q  -> L4
    const float phi5 = m == 5 ? 1.f : 0.f;                   // a = sum_q Q         -> L5

    for (;;) {
        const uint32_t cell = lds_draw(lds_addr(&sh.next));
        if (cell >= 16u) break;
        // positions at or beyond the cell's largest n_contrib are dead for all its pixels (backward.cu:490-492)
        const uint32_t gm = min(sh.clast[cell], n);
        if (gm == 0) continue;  // wave-uniform
        const int band = (int)(cell >> 2), g = (int)(cell & 3u);
        GHR_PROF(0);

        // ---- the first segment of GHR_B3_SEG_WORDS mask words (from the back of the list) that has hits in reach
        int w_hi = (int)((gm - 1u) >> 6), w_lo = 0;
        uint32_t n_c = 0;
        auto build = [&]() {  // expands the words w_lo..w_hi of the cell into list[0 .. n_c)
            w_lo = max(w_hi - (GHR_B3_SEG_WORDS - 1), 0);
            n_c = 0;
            for (int w = w_lo; w <= w_hi; w++) {
                const unsigned long long mk = SMALL ? sh.mask[w][cell] : cell_mask[(word0 + w) * 16 + cell];  // wave-uniform
                const uint32_t pos = 64u * w + lane;
                const bool bit = ((mk >> lane) & 1ull) != 0ull && pos < gm;
                const unsigned long long bm = __builtin_amdgcn_ballot_w64(bit);
                if (bit) list[n_c + lanes_below(bm)] = (uint16_t)(pos - 64u * w_lo);
                n_c += (uint32_t)__builtin_popcountll(bm);
            }
            __builtin_amdgcn_wave_barrier();
        };
        // chunks of 16 entries from the top of the segment's list: chunk t holds the list indices n_c - 16 t - 1 - e,
        // e = 0..15 (ascending e = back to front); the ones past the front of the list repeat the chunk's last entry and
        // are masked.  Lane L gathers quarter L >> 4 of entry L & 15, so that a buffer holds the 16 first quarters, then the 16
        // second ones, ...: the chunk's reads below (16 entries side by side) are free of bank conflicts.
        // (`buf` = t % GHR_B3_NBUF, carried by the callers: a running counter instead of a division per use)
        auto issue = [&](uint32_t t, uint32_t buf) {
            const uint32_t hi = n_c - 16u * t, cnt = min(hi, 16u);
            const uint32_t e = min((uint32_t)lane & 15u, cnt - 1u);
            const uint32_t pos = 64u * w_lo + list[hi - 1u - e];
            const uint32_t id = SMALL ? sh.id[pos] : ld32(point_list, 4u * (beg + pos));
            const uint32_t slot = min(beg + pos, cap - 1u);  // the gradient lines lie in list order
            gather16_to_lds(rec, 64u * id + 16u * ((uint32_t)lane >> 4), &sh.rec[wave][buf][0]);
            if (lane < 16) sh.cslot[wave][buf][lane] = slot;
        };
        build();
        while (n_c == 0 && w_hi >= GHR_B3_SEG_WORDS) { w_hi -= GHR_B3_SEG_WORDS; build(); }
        GHR_PROF(2);
        if (n_c == 0) continue;  // nothing of the cell's list reaches its pixels (wave-uniform)
        if (SMALL) {
            issue(0, 0);
            if (n_c > 16u) issue(1, 1);
        }

        // ---- the cell's pixels: lane (k, m) evaluates the pixels (x = 4g + k, y = 4 band + q), q = 0..3.  Their loads go
        // out behind the first gathers and all of them are in flight together.
        // (hk, hm: k and m made opaque per cell, so that the address arithmetic below is redone here instead of being
        // hoisted out of the cell loop into registers that live -- or spill -- across the chunk loop)
        int hk = k, hm = m;
        asm volatile("" : "+v"(hk), "+v"(hm));
        const int px = tx * GHR_TILE_X + 4 * g + hk, py0 = ty * GHR_TILE_Y + 4 * band;
        float phiW[4];
        uint32_t last[4];
        float dLA0, dLA1, dLA2;
        f2b TinA, TinB, PSA, PSB;
        {
            float Tf[4];
            const uint32_t pxc = (uint32_t)min(px, W - 1);
            const uint32_t hmc = hm >= 6 ? (uint32_t)(hm - 6) : 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) {  // clamped addresses, then a select: no load under a branch
                const uint32_t pix = 4u * ((uint32_t)W * (uint32_t)min(py0 + q, H - 1) + pxc);  // byte offset
                Tf[q] = ld32_once(final_T, pix);
                last[q] = ld32_once(n_contrib, pix);
                phiW[q] = ld32_once(dL_dpix, hmc * plane + pix);  // B operand of the colour MFMAs: dL/dpixel[m - 6] of pixel q
            }
            // A operand of the colour-dot MFMAs: row i = m of the product is the cell pixel (x = m >> 2, y = m & 3), so
            // that lane (k, e) finds the dots of ITS pixels (k, 0..3) in its four result registers
            const int ax = tx * GHR_TILE_X + 4 * g + (hm >> 2), ay = py0 + (hm & 3);
            const uint32_t hkc2 = hk < 2 ? (uint32_t)(8 + hk) : 0u;
            const uint32_t pixa = 4u * ((uint32_t)W * (uint32_t)min(ay, H - 1) + (uint32_t)min(ax, W - 1));
            dLA0 = ld32_once(dL_dpix, (uint32_t)hk * plane + pixa);
            dLA1 = ld32_once(dL_dpix, (uint32_t)(4 + hk) * plane + pixa);
            dLA2 = ld32_once(dL_dpix, hkc2 * plane + pixa);
            // the background, as the "colour" of one more splat behind the list (backward.cu:535-538): bg . dL/dpixel of
            // the lane's four pixels from the same three MFMAs as the colour dots
            float bgA0 = ld32(bg, 4u * (uint32_t)hk), bgA1 = ld32(bg, 4u * (uint32_t)(4 + hk)), bgA2 = ld32(bg, 4u * hkc2);
#pragma unroll
            for (int q = 0; q < 4; q++) {  // pixels outside the image take no part (n_contrib 0, no gradient)
                const bool in = px < W && py0 + q < H;
                Tf[q] = in ? Tf[q] : 0.f;
                last[q] = in ? last[q] : 0u;
                phiW[q] = (in && hm >= 6) ? phiW[q] : 0.f;
            }
            const bool ina = ax < W && ay < H;
            dLA0 = ina ? dLA0 : 0.f;
            dLA1 = ina ? dLA1 : 0.f;
            dLA2 = (ina && hk < 2) ? dLA2 : 0.f;
            bgA2 = hk < 2 ? bgA2 : 0.f;
            f4 bd = {0.f, 0.f, 0.f, 0.f};
            bd = mfma16(dLA0, bgA0, bd);
            bd = mfma16(dLA1, bgA1, bd);
            bd = mfma16(dLA2, bgA2, bd);
            TinA = f2b{Tf[0], Tf[1]}; TinB = f2b{Tf[2], Tf[3]};
            PSA = f2b{Tf[0] * bd.x, Tf[1] * bd.y}; PSB = f2b{Tf[2] * bd.z, Tf[3] * bd.w};
        }
        const float u = (float)(4 * g + k);                 // pixel - tile origin, x
        // wave-uniform values are pinned to scalar registers (results of float VALU ops are not, on their own)
        const float v0 = uniform_f((float)(4 * band));      // ... y of q = 0
        const float phiSX = m == 0 ? 1.f : (m == 2 ? u : 0.f);   // a = sum_q Q dx -> L0 (x1), L2 (x u)
        const float pxf = wx0 + u;
        const float py0f = uniform_f((float)py0), py1f = uniform_f((float)(py0 + 1)), py2f = uniform_f((float)(py0 + 2)),
                    py3f = uniform_f((float)(py0 + 3));  // exact integers, like K7's pyf
        GHR_PROF(1);

        for (;;) {  // segments
            const uint32_t nch = (n_c + 15u) >> 4;
            uint32_t bt = 0;  // t % GHR_B3_NBUF
            for (uint32_t t = 0; t < nch; t++, bt = bt == GHR_B3_NBUF - 1 ? 0u : bt + 1u) {
                const uint32_t hi = n_c - 16u * t, cnt = min(hi, 16u);
                if (SMALL) {
                    // memory operations issued after the gather of chunk t: [t >= 2: the four atomics of chunk t-2]
                    // [t >= 1: the gather of chunk t+1 if there is one, the four atomics of chunk t-1] [t == 0: the
                    // gather of chunk 1 if there is one; the pixel loads of a first segment have been waited for above]
                    const bool more = t + 1 < nch;
                    if (t >= 2) { if (more) GHR_VMCNT(9); else GHR_VMCNT(8); }
                    else if (t == 1) { if (more) GHR_VMCNT(5); else GHR_VMCNT(4); }
                    else { if (more) GHR_VMCNT(1); else GHR_VMCNT(0); }
                } else {
                    issue(t, bt);
                    GHR_VMCNT(0);
                }
                __builtin_amdgcn_wave_barrier();
                GHR_PROF(3);
                GHR_PROF_COUNT(7, 1);

                const bool valid = (uint32_t)m < cnt;
                const uint32_t j = min((uint32_t)m, cnt - 1u);  // lanes past the end recompute the last entry, masked
                const int f0 = 6 + k, f1 = 10 + k, f2 = 6 + kc2;            // colours k, 4 + k, 8 + k
                // field f of entry j sits at float (f >> 2) * 64 + 4 j + (f & 3): quarter-major, see the gather
                const float* R = reinterpret_cast<const float*>(&sh.rec[wave][bt][0]) + 4u * j;
                const f4 r0 = *reinterpret_cast<const f4*>(R);              // x y a b
                const f2b r1 = *reinterpret_cast<const f2b*>(R + 64);       // c o
                const float col0 = R[(f0 >> 2) * 64 + (f0 & 3)], col1 = R[(f1 >> 2) * 64 + (f1 & 3)],
                            col2 = k < 2 ? R[(f2 >> 2) * 64 + (f2 & 3)] : 0.f;
                const float ex = r0.x, ey = r0.y, ca = r0.z, cb = r0.w, cc = r1.x, o = r1.y;
                const uint32_t pos = 64u * w_lo + list[hi - 1u - j];
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
                // the gather of chunk t+2 goes out before this chunk's atomics (see GHR_VMCNT above)
                if (SMALL && t + 2 < nch) issue(t + 2, bt == 0 ? GHR_B3_NBUF - 1 : bt - 1u);  // (t + 2) % 3 == (t - 1) % 3
                // a DPP row adds one whole 64-B line per register: resolved in this XCD's L2 (only this workgroup ever
                // touches the instance's line).  Always four atomic INSTRUCTIONS (the vmcnt arithmetic above): the rows
                // without an entry are masked out of EXEC by hand -- under an `if` the compiler would branch around the
                // instruction -- and lane 0 always takes part (adding 0.0 to a line of the chunk when its row is empty),
                // so that the instruction never runs with an empty mask.
                // (lane (k, c) adds to the entries 4k + r: their lines from one 16-B read -- the gather fills all sixteen
                // slots of a chunk, the ones past the end with the last entry's; which rows have an entry is wave-uniform
                // arithmetic on cnt: the rows k < ceil((cnt - r) / 4))
                const uint32_t* cs4 = &sh.cslot[wave][bt][4 * k];
                const uint32_t sl[4] = {cs4[0], cs4[1], cs4[2], cs4[3]};
                if (cnt == 16u) {  // (wave-uniform) a full chunk -- three of five: every lane has an entry, nothing to mask
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float val = r == 0 ? d.x : (r == 1 ? d.y : (r == 2 ? d.z : d.w));
                        const uint32_t off = 64u * sl[r] + 4u * (uint32_t)m;
                        asm volatile("global_atomic_add_f32 %0, %1, %2" : : "v"(off), "v"(val), "s"(ginst) : "memory");
                    }
                } else {
                    const uint32_t k4 = 4u * (uint32_t)k;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        // lane (k, c) has an entry in register r when 4 k + r < cnt: one vector compare gives the lane mask
                        const unsigned long long has = __builtin_amdgcn_ballot_w64(k4 + (uint32_t)r < cnt);
                        float val = r == 0 ? d.x : (r == 1 ? d.y : (r == 2 ? d.z : d.w));
                        val = (k4 + (uint32_t)r < cnt) ? val : 0.f;  // (only lane 0 can be active without an entry)
                        const unsigned long long on = has | 1ull;
                        const uint32_t off = 64u * sl[r] + 4u * (uint32_t)m;
                        unsigned long long saved;
                        asm volatile("s_mov_b64 %0, exec\n\t"
                                     "s_and_b64 exec, exec, %1\n\t"
                                     "global_atomic_add_f32 %2, %3, %4\n\t"
                                     "s_mov_b64 exec, %0"
                                     : "=&s"(saved)
                                     : "s"(on), "v"(off), "v"(val), "s"(ginst)
                                     : "memory");
                    }
                }
                GHR_PROF(4);
            }
            // next segment towards the front of the list
            do {
                w_hi -= GHR_B3_SEG_WORDS;
                if (w_hi < 0) break;
                build();
            } while (n_c == 0);
            if (w_hi < 0) break;
            if (SMALL) {
                issue(0, 0);
                if (n_c > 16u) issue(1, 1);
            }
        }
    }
    GHR_PROF(5);
    GHR_PROF_END(6);
}
#endif

__global__ void __launch_bounds__(GHR_B3_THREADS, GHR_B3_WAVES) k_render_bwd_cells(int W, int H, int gx, uint32_t T_tiles,
                                                                const uint32_t* __restrict__ tile_start,
                                                                const uint32_t* __restrict__ point_list,
                                                                const f4* __restrict__ rec, const float* __restrict__ bg,
                                                                const float* __restrict__ final_T,
                                                                const uint32_t* __restrict__ n_contrib,
                                                                const float* __restrict__ dL_dpix,
                                                                const rect4* __restrict__ rects, float* ginst,
                                                                uint32_t cap,
                                                                const unsigned long long* __restrict__ cell_mask,
                                                                const uint32_t* __restrict__ cell_last, int ordered,
                                                                int prezeroed, const uint32_t* __restrict__ tile_order)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ B3Shared sh;
    const int tid = threadIdx.x;
    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : xcd_tile(blockIdx.x, T_tiles);  // heaviest first (k_tile_scan)
    if (tile >= T_tiles) return;  // grid padding
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t beg = min(tile_start[tile], cap);
    const uint32_t n = min(tile_start[tile + 1], cap) - beg;  // see k_render_bwd for `cap`
    if (n == 0) return;
    const size_t word0 = mask_word0(beg, tile);
    const bool small = n <= GHR_B3_CACHE;

    // ---- tile prologue: zero the gradient lines of all the tile's instances (k_geom_bwd / k_project_bwd read every
    //      line of a Gaussian) and bring what every cell needs from the lists into LDS
    for (uint32_t i = tid; i < n; i += GHR_B3_THREADS) {
        if (small) sh.id[i] = point_list[beg + i];
        // the lines lie in list order (the per-Gaussian gather finds them through the sort's inst_line): plain
        // consecutive stores, nothing to look up
        if (!prezeroed) {  // (wave-uniform) else the tile sort of this state's forward pass has zeroed them already
            f4* dst = reinterpret_cast<f4*>(ginst) + 4u * min(beg + i, cap - 1u);
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
        }
    }
    if (small)
        for (uint32_t i = tid; i < 16u * ((n + 63u) >> 6); i += GHR_B3_THREADS) (&sh.mask[0][0])[i] = cell_mask[word0 * 16 + i];
    if (tid < 16) sh.clast[tid] = cell_last[16u * tile + tid];
    // (volatile: the cell draws are asm the compiler cannot see)
    if (tid == 0) *(volatile uint32_t*)&sh.next = 0u;
    __syncthreads();  // orders the zero-fill (vmcnt(0) + workgroup fence) before the atomics below
    // ghr_set_deterministic: one wave draws all sixteen cells, in index order -- the additions into a line (issued by one
    // wave to one address) then happen in program order
    if (ordered && tid >= 64) return;

    if (small)
        b3_tile<true>(sh, W, H, tx, ty, tile, beg, n, point_list, rec, bg, final_T, n_contrib, dL_dpix, rects, ginst, cap,
                      cell_mask, word0);
    else
        b3_tile<false>(sh, W, H, tx, ty, tile, beg, n, point_list, rec, bg, final_T, n_contrib, dL_dpix, rects, ginst, cap,
                       cell_mask, word0);
#endif
}

}  // namespace ghr
