// ghr_render_fwd.h -- K7: front-to-back alpha compositing of the 10 feature channels, one 16x16 tile per workgroup.
// Follows R:cuda_rasterizer/forward.cu:287-400 (renderCUDA).
//
// CDNA4 mapping ("cell groups"): 256 threads = 4 wavefronts; wave w owns pixel rows 4w..4w+3 of the tile and each of
// its four 16-lane DPP rows ("groups") owns one 4x4-pixel CELL of that strip.  The reference visits every pixel of
// every tile in the 3-sigma SQUARE of a splat; strand-aligned (needle) Gaussians cover a few dozen pixels of it, so a
// whole-wave visit of one splat keeps ~15 % of the lanes busy.  Here every group walks its OWN ordered list of the
// batch entries whose alpha >= 1/255 box touches its cell (a per-lane 64-bit mask built from four ballots), so the
// four groups of a wave composite four different splats in the same instruction stream (measured on cfg3: 1.7x fewer
// wave passes than per-wave strips).  Skipped entries would have been rejected by every pixel of the cell, so the
// result is bit-identical to visiting everything.
// A batch of 256 list entries is staged as packed 64-B records (position, conic, opacity AND the 10 features -- the
// reference re-gathers features from global memory per contributing pixel, forward.cu:381) into four SoA float4 planes;
// the gather for batch b+1 is issued into registers before batch b is composited.
// The cell masks are also what the backward pass needs to know (which list entries can touch which cell): they are
// stored -- one 64-bit word per (64 list positions, cell), 2 B per instance -- together with each cell's largest
// n_contrib, so that k_render_bwd_cells (ghr_render_bwd3.h) neither recomputes the cull nor stages whole tiles.
#pragma once
#include "ghr_device.h"

namespace ghr {

struct PixFwd {
    float T;
    float C[GHR_C];
    uint32_t last;  // last_contributor
};

// One list entry applied to one pixel (forward.cu:351-388).  `pos1` is the 1-based position of the entry in the
// tile's list (the reference's `contributor`); `live` = the pixel is not finished yet.  Returns true when the pixel
// finishes on this entry (T would drop below 1e-4).  The three skip tests of the reference are evaluated as predicates
// and ONE branch guards the blend (a chain of four nested wave-level branches per visit costs more than the few
// arithmetic instructions it can skip); the blended values are bit-identical.
GHR_HD bool fwd_step(PixFwd& s, bool live, float pxf, float pyf, const f4& r0, const f4& r1, const f4& r2,
                     const f4& r3, uint32_t pos1)
{
    const float dx = r0.x - pxf, dy = r0.y - pyf;
    // forward.cu:361, source order, unfused (decision input)
    const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
    const float alpha = fminf(0.99f, r1.y * fast_exp(power));
    const bool c = live && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
    const float test_T = s.T * (1 - alpha);
    const bool fin = c && test_T < 0.0001f;
    if (!c || fin) return fin;
    const float w = alpha * s.T;
    s.C[0] = fma_(r1.z, w, s.C[0]);
    s.C[1] = fma_(r1.w, w, s.C[1]);
    s.C[2] = fma_(r2.x, w, s.C[2]);
    s.C[3] = fma_(r2.y, w, s.C[3]);
    s.C[4] = fma_(r2.z, w, s.C[4]);
    s.C[5] = fma_(r2.w, w, s.C[5]);
    s.C[6] = fma_(r3.x, w, s.C[6]);
    s.C[7] = fma_(r3.y, w, s.C[7]);
    s.C[8] = fma_(r3.z, w, s.C[8]);
    s.C[9] = fma_(r3.w, w, s.C[9]);
    s.T = test_T;
    s.last = pos1;
    return false;
}

__global__ void __launch_bounds__(GHR_BLOCK) k_render_fwd(int W, int H, int gx, uint32_t T_tiles,
                                                          const uint32_t* __restrict__ tile_start,
                                                          const uint32_t* __restrict__ point_list,
                                                          const f4* __restrict__ rec, const float* __restrict__ bg,
                                                          float* __restrict__ out_color, float* __restrict__ final_T,
                                                          uint32_t* __restrict__ n_contrib, uint32_t cap,
                                                          unsigned long long* __restrict__ cell_mask,
                                                          uint32_t* __restrict__ cell_last,
                                                          const uint32_t* __restrict__ tile_order, float* ginst)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ f4 s_r0[GHR_BLOCK], s_r1[GHR_BLOCK], s_r2[GHR_BLOCK], s_r3[GHR_BLOCK], s_bb[GHR_BLOCK], s_ep[GHR_BLOCK];

    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : xcd_tile(blockIdx.x, T_tiles);  // heaviest first (k_tile_scan)
    if (tile >= T_tiles) return;  // grid padding
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, l = lane & 15;
    // this lane's pixel: cell (wave, grp) of the tile, 4x4 pixels, lane l -> (l & 3, l >> 2)
    const int px = tx * GHR_TILE_X + 4 * grp + (l & 3), py = ty * GHR_TILE_Y + 4 * wave + (l >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float wx0 = (float)(tx * GHR_TILE_X);
    const float cy0 = (float)(ty * GHR_TILE_Y + 4 * wave);

    const uint32_t beg = min(tile_start[tile], cap), end = min(tile_start[tile + 1], cap);  // cap: ghr_forward_stage2
    const uint32_t n = end - beg;

    GHR_PROF_DECL;
    PixFwd st;
    st.T = 1.0f;
    st.last = 0;
#pragma unroll
    for (int c = 0; c < GHR_C; c++) st.C[c] = 0.f;
    bool done = !inside;

    // software-pipelined gather: registers hold the next batch while the current one is composited
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 g0 = zero4, g1 = zero4, g2 = zero4, g3 = zero4;
#define GHR_GATHER(base_)                                     \
    do {                                                      \
        const uint32_t i_ = (base_) + tid;                    \
        if (i_ < n) {                                         \
            const uint32_t id_ = point_list[beg + i_];        \
            const f4* r_ = rec + 4 * (size_t)id_;             \
            g0 = r_[0]; g1 = r_[1]; g2 = r_[2]; g3 = r_[3];   \
        }                                                     \
    } while (0)
    if (n > 0) GHR_GATHER(0u);

    for (uint32_t base = 0; base < n; base += GHR_BLOCK) {
        // forward.cu:335-337: stop when the whole tile is done.  The barrier also protects the LDS planes.
        GHR_PROF(0);
        if (__syncthreads_count(done) == GHR_BLOCK) break;
        GHR_PROF(1);
        s_r0[tid] = g0; s_r1[tid] = g1; s_r2[tid] = g2; s_r3[tid] = g3;
        s_bb[tid] = alpha_bbox(g0, g1);
        s_ep[tid] = ellipse_params(g0, g1);
        __syncthreads();
        GHR_PROF(2);
        if (base + GHR_BLOCK < n) GHR_GATHER(base + GHR_BLOCK);

        const uint32_t cnt = min((uint32_t)GHR_BLOCK, n - base);
        for (uint32_t sub = 0; sub < cnt; sub += 64) {
            const unsigned long long alive = __builtin_amdgcn_ballot_w64(!done);
            if (alive == 0) break;  // wave-uniform: all 64 pixels finished
            const uint32_t e = sub + lane;
            const uint32_t ec = e < cnt ? e : 0;
            unsigned long long todo = cell_masks(s_bb[ec], s_ep[ec], s_r0[ec], e < cnt, wx0, cy0, grp);
            GHR_PROF(3);
            // word (base + sub) / 64 of this cell: every position a pixel of the cell can count in n_contrib lies in a
            // word written here (a cell that is finished, or a tile that stops early, has all its n_contrib behind it)
            // (uniform base + 32-bit byte offset: one address register; the kernel sits at the 80-VGPR occupancy step)
            if (l == 0)
                *reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(cell_mask) +
                    (128u * (uint32_t)(mask_word0(beg, tile) + ((base + sub) >> 6)) + 8u * (uint32_t)(4 * wave + grp))) = todo;
            if (((alive >> (16 * grp)) & 0xffffull) == 0) todo = 0;  // this cell is finished
            while (todo) {  // divergent per GROUP (all 16 lanes of a DPP row share `todo`)
                // (the entry's LDS byte offset straight from the bit index -- (ctz << 4) + 16 sub is ONE v_lshl_add -- and its
                // list position from the scalar base + sub + 1: the kernel is VALU-bound, +8 VALU per step = +17 %)
                const uint32_t bit = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                uint32_t off;  // (written out: the compiler turns (bit << 4) + 16 sub back into an OR and a shift)
                asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(off) : "v"(bit), "s"(16u * sub));
                done |= fwd_step(st, !done, pxf, pyf, *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(s_r0) + off),
                                 *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(s_r1) + off),
                                 *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(s_r2) + off),
                                 *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(s_r3) + off), (base + sub + 1u) + bit);
            }
            GHR_PROF(4);
        }
    }

#undef GHR_GATHER

    {   // the cell's largest n_contrib: positions at or beyond it are dead for all its pixels (backward.cu:490-492)
        uint32_t lm = inside ? st.last : 0u;
        lm = max(lm, (uint32_t)__shfl_xor((int)lm, 1));
        lm = max(lm, (uint32_t)__shfl_xor((int)lm, 2));
        lm = max(lm, (uint32_t)__shfl_xor((int)lm, 4));
        lm = max(lm, (uint32_t)__shfl_xor((int)lm, 8));
        if (l == 0) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(cell_last) + (64u * tile + 4u * (uint32_t)(4 * wave + grp))) = lm;
    }
    if (inside) {  // forward.cu:393-399
        const size_t pix = (size_t)W * py + px;
        const size_t plane = (size_t)W * H;
        final_T[pix] = st.T;
        n_contrib[pix] = st.last;
#pragma unroll
        for (int c = 0; c < GHR_C; c++) out_color[c * plane + pix] = st.C[c] + st.T * bg[c];
    }
    // The tile's gradient lines are its n consecutive lines from `beg` (they lie in list order): when the caller hands the
    // backward pass's scratch over they are zeroed HERE, as the workgroup's last act (the backward render kernel then starts
    // accumulating at once).  This kernel is bound by instruction issue and leaves two thirds of the memory pipe idle; the
    // stores are behind every load of the workgroup, so nothing waits for their acknowledgement.  (Rounds 3-5 had them in the
    // tile sort: 83 MB that had become more than half of that kernel's time once its network was register-blocked.)
    if (ginst != nullptr) {
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t i = tid; i < 4u * n; i += GHR_BLOCK)
            __builtin_nontemporal_store(zero, reinterpret_cast<f4*>(ginst) + 4 * (size_t)beg + i);
    }
    GHR_PROF(5);
#ifdef GHR_K7_PROF
    GHR_PROF_END(6);
#endif
#endif
}

}  // namespace ghr
