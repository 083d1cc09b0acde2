// ghr_render_bwd.h -- K8: back-to-front gradient walk, one 16x16 tile per workgroup.
// Follows R:cuda_rasterizer/backward.cu:403-561 (renderCUDA).
//
// The reference issues 16 global atomicAdds per contributing (pixel, Gaussian) pair (backward.cu:527,549-558).
// Here (same "cell group" mapping as k_render_fwd: wave = 16x4 pixel strip, each 16-lane DPP row = one 4x4 cell
// walking its own list of the batch entries whose alpha box touches the cell) the 16 gradient components
// {mean2D.x, .y, conic.a, .b, .c, opacity, colors[10]} of one list entry are
//   1. reduced across the 16 lanes of the cell with a register "halving butterfly" of DPP row operations
//      (row_ror:8, row_half_mirror, quad_perm): 16 -> 8 -> 4 -> 2 -> 1 values per lane, lane l ends up holding the
//      cell's total of component l (45 VALU, no LDS, no cross-row traffic),
//   2. accumulated across the tile's 16 cells straight into the (tile, Gaussian) INSTANCE's own 64-byte gradient slot
//      (rect4_slot) with ONE workgroup-scope global_atomic_add_f32 per lane: the 16 lanes of a cell hit one cache line,
//      and because only this workgroup ever touches the slot the atomic is resolved in the XCD's L2 (no sc1 / fabric
//      round trip).  The slot line is zero-filled by the staging thread of the entry before the barrier that precedes
//      the walk.  Measured on MI355X (cfg3): an LDS table with ds_add_f32 cost 24 % of the kernel (LDS float atomics
//      retire ~1.8 cycles per lane), device-scope atomics on a per-Gaussian line 40 %, this form is free.
//   3. summed per Gaussian over its instance slots in a fixed order by the per-Gaussian backward (k_geom_bwd /
//      k_project_bwd) -- no cross-tile float atomics.
//
// The per-channel recurrences of the reference (accum_rec[ch], last_color[ch], backward.cu:519-523) are linear in the
// channel index and only ever used through sum_ch(. * dL_dpixel[ch]); they are carried as ONE scalar
//   S = sum_ch accum_rec[ch]*dL_dpixel[ch],   S <- last_alpha*last_cdot + (1-last_alpha)*S,  cdot = sum_ch c[ch]*dL[ch]
// which is the same real-number value (fp32 rounding differs at the 1e-7 level) and frees 20 VGPRs.
//
// bwd_step is branch-free: a pair that does not contribute (position >= n_contrib, power > 0, alpha < 1/255) runs the
// same arithmetic with alpha = G = 0, which leaves T untouched (T * rcp(1) == T), turns the S update into the exact
// flush S <- last_alpha*last_cdot + (1-last_alpha)*S the next contributing entry would have performed (then
// S <- 0*x + 1*S, exact), and makes all 16 gradient terms exact zeros -- no divergence, no zero-fill of g[].
#pragma once
#include "ghr_device.h"

#define GHR_COMMON_MIN 8  // batch entries common to a strip's four cells from which the wave-wide path pays

namespace ghr {

typedef float f2 __attribute__((ext_vector_type(2)));  // pairs of channels: v_pk_mul_f32 / v_pk_fma_f32 (2 flops/lane/issue)

struct PixBwd {
    float T, T_final, S, last_alpha, last_cdot, bgdot;
    f2 dL[GHR_C / 2];  // dL/dpixel of the 10 channels, channel pairs (0,1) (2,3) ...
};

// One list entry applied to one pixel (backward.cu:494-558).  `live` = the entry lies below the pixel's n_contrib
// (backward.cu:490-492).  Always writes the 16 per-pair gradient terms to g[] (exact zeros when the pair does not
// contribute) and returns whether it contributed.
GHR_HD bool bwd_step(PixBwd& s, bool live, float pxf, float pyf, const f4& r0, const f4& r1, const f4& r2,
                     const f4& r3, float u, float v, float* g)
{
    const float dx = r0.x - pxf, dy = r0.y - pyf;
    const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;  // unfused (decision input)
    const float G_raw = fast_exp(power);
    const float o = r1.y;
    const float alpha_raw = fminf(0.99f, o * G_raw);
    const bool c = live && !(power > 0.0f) && !(alpha_raw < 1.0f / 255.0f);
    const float alpha = c ? alpha_raw : 0.0f;
    const float G = c ? G_raw : 0.0f;

    const float inv1ma = fast_rcp(1.f - alpha);
    s.T = s.T * inv1ma;  // backward.cu:507
    const float w = alpha * s.T;

    // colour . dL/dpixel over the 10 channels, two channels per packed FMA
    f2 cd = f2{r1.z, r1.w} * s.dL[0];
    cd = __builtin_elementwise_fma(f2{r2.x, r2.y}, s.dL[1], cd);
    cd = __builtin_elementwise_fma(f2{r2.z, r2.w}, s.dL[2], cd);
    cd = __builtin_elementwise_fma(f2{r3.x, r3.y}, s.dL[3], cd);
    cd = __builtin_elementwise_fma(f2{r3.z, r3.w}, s.dL[4], cd);
    // a pair that does not contribute must not leak a non-finite feature of its Gaussian into S or g[] (0 * inf)
    const float cdot = c ? cd.x + cd.y : 0.0f;
    // backward.cu:519-523 collapsed to scalars (see header)
    s.S = fma_(s.last_alpha, s.last_cdot, (1.f - s.last_alpha) * s.S);
    s.last_cdot = cdot;
    s.last_alpha = alpha;
    float dL_dalpha = (cdot - s.S) * s.T;                                 // :523,:529
    dL_dalpha = fma_(-s.T_final * inv1ma, s.bgdot, dL_dalpha);            // :535-538

    // The six geometric terms of backward.cu:542-558 are linear in the pixel sums of Q, Q dx, Q dy, Q dx u, Q dx v,
    // Q dy v (Q = G dL/dalpha; (u, v) = pixel - tile origin): the line carries those sums and the per-Gaussian gather
    // recovers the reference's terms (ghr_device.h, LineAcc) -- 6 VALU here instead of 18.
    (void)o;
    const float Q = G * dL_dalpha;       // :558 (dL/dopacity term)
    const float qx = Q * dx, qy = Q * dy;
    g[0] = qx;
    g[1] = qy;
    g[2] = qx * u;
    g[3] = qx * v;
    g[4] = qy * v;
    g[5] = Q;
#pragma unroll
    for (int i = 0; i < GHR_C / 2; i++) {  // :508,:527
        const f2 gc = s.dL[i] * w;
        g[6 + 2 * i] = gc.x;
        g[7 + 2 * i] = gc.y;
    }
    return c;
}

#if defined(__HIP_DEVICE_COMPILE__)
// value of `v` in the DPP partner lane.  ctrl: 0 = row_ror:8 (l ^ 8), 1 = row_half_mirror (l ^ 7),
// 2 = quad_perm [2,3,0,1] (l ^ 2), 3 = quad_perm [1,0,3,2] (l ^ 1)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v)
{
    constexpr int ctrl = CTRL == 0 ? 0x128 : (CTRL == 1 ? 0x141 : (CTRL == 2 ? 0x4E : 0xB1));
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), ctrl, 0xf, 0xf, false));
}

// Sum 16 per-lane values over the 16 lanes of a DPP row.  On return lane l (0..15 within its row) holds the row's
// total of component l.  Each stage pairs a lane with a partner that differs in one more bit of l, keeps half of its
// values and receives the partner's copy of the same half: 8 + 4 + 2 + 1 exchanges.
// The first two stages (partners l ^ 8 and l ^ 7) split the lanes along whole 4-lane banks, so "keep mine / take the
// partner's" is expressed with the DPP bank write mask instead of v_cndmask: two v_add_f32_dpp per output
//     h = g_lo + partner(g_lo)   written by the banks that keep the low half
//     h = g_hi + partner(g_hi)   written by the other banks
// (hand-written: the compiler cannot derive the masked form).  s_nop 1 covers the "VALU write -> DPP read" hazard at
// the block boundaries, which the hazard recogniser does not see through inline asm.  33 VALU in total.
__device__ __forceinline__ float row_reduce16(const float* g, int l)
{
    float h0, h1, h2, h3, h4, h5, h6, h7;
#define GHR_PAIR(d, lo, hi, ctrl, mlo, mhi)                                                   \
    "v_add_f32_dpp " d ", " lo ", " lo " " ctrl " row_mask:0xf bank_mask:" mlo "\n\t"          \
    "v_add_f32_dpp " d ", " hi ", " hi " " ctrl " row_mask:0xf bank_mask:" mhi "\n\t"
    asm volatile("s_nop 1\n\t"  // partner l ^ 8: banks 0,1 keep components 0..7, banks 2,3 keep 8..15
                 GHR_PAIR("%0", "%8", "%16", "row_ror:8", "0x3", "0xc")
                 GHR_PAIR("%1", "%9", "%17", "row_ror:8", "0x3", "0xc")
                 GHR_PAIR("%2", "%10", "%18", "row_ror:8", "0x3", "0xc")
                 GHR_PAIR("%3", "%11", "%19", "row_ror:8", "0x3", "0xc")
                 GHR_PAIR("%4", "%12", "%20", "row_ror:8", "0x3", "0xc")
                 GHR_PAIR("%5", "%13", "%21", "row_ror:8", "0x3", "0xc")
                 GHR_PAIR("%6", "%14", "%22", "row_ror:8", "0x3", "0xc")
                 GHR_PAIR("%7", "%15", "%23", "row_ror:8", "0x3", "0xc")
                 : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(h4), "=&v"(h5), "=&v"(h6), "=&v"(h7)
                 : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]),
                   "v"(g[8]), "v"(g[9]), "v"(g[10]), "v"(g[11]), "v"(g[12]), "v"(g[13]), "v"(g[14]), "v"(g[15]));
    float q0, q1, q2, q3;
    asm volatile(  // partner l ^ 7 (flips bit 2): banks 0,2 keep the lower four of their eight, banks 1,3 the upper
                 GHR_PAIR("%0", "%4", "%8", "row_half_mirror", "0x5", "0xa")
                 GHR_PAIR("%1", "%5", "%9", "row_half_mirror", "0x5", "0xa")
                 GHR_PAIR("%2", "%6", "%10", "row_half_mirror", "0x5", "0xa")
                 GHR_PAIR("%3", "%7", "%11", "row_half_mirror", "0x5", "0xa")
                 "s_nop 1"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                 : "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(h4), "v"(h5), "v"(h6), "v"(h7));
#undef GHR_PAIR
    // the last two stages pair lanes inside a quad (no bank granularity): select + add
    const bool b1 = (l & 2) != 0, b0 = (l & 1) != 0;
    const float k0 = b1 ? q2 : q0, s0 = b1 ? q0 : q2, k1 = b1 ? q3 : q1, s1 = b1 ? q1 : q3;
    const float e0 = k0 + dpp_get<2>(s0), e1 = k1 + dpp_get<2>(s1);  // partner l ^ 2
    const float keep = b0 ? e1 : e0, send = b0 ? e0 : e1;            // partner l ^ 1
    return keep + dpp_get<3>(send);
}

// the 64-bit mask held by the lanes of the DPP row that starts at `first_lane`, as a wave-uniform value
__device__ __forceinline__ unsigned long long rowmask(unsigned long long m, int first_lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m, first_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m >> 32), first_lane);
    return ((unsigned long long)hi << 32) | lo;
}

// Sum 16 per-lane values over all 64 lanes of the wave (used for the entries that all four cells of the strip visit: a
// splat that covers the whole strip).  permlane32_swap (16 -> 8 values), permlane16_swap (8 -> 4), then the row butterfly.
// On return lane L holds the total of component 8*b5 + 4*b4 + 2*b3 + b2 (b_i = bit i of L), replicated over its quad.
__device__ __forceinline__ float wave_reduce16(const float* g, int lane)
{
    float h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {  // lanes 0-31 keep component i, lanes 32-63 keep component 8+i
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(g[i]), __float_as_uint(g[8 + i]), false, false);
        h[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    float q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {  // even rows keep h[i], odd rows keep h[4+i]
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[i]), __float_as_uint(h[4 + i]), false, false);
        q[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
    const float d0 = q[0] + dpp_get<0>(q[0]), d1 = q[1] + dpp_get<0>(q[1]);
    const float d2 = q[2] + dpp_get<0>(q[2]), d3 = q[3] + dpp_get<0>(q[3]);
    const float e0 = b3 ? d2 : d0, e1 = b3 ? d3 : d1;  // lanes with bit 3 keep q[2], q[3]
    const float f0 = e0 + dpp_get<1>(e0), f1 = e1 + dpp_get<1>(e1);
    float v = b2 ? f1 : f0;                            // lanes with bit 2 keep e1
    v += dpp_get<2>(v);
    v += dpp_get<3>(v);
    return v;
}
#endif

__global__ void __launch_bounds__(GHR_BLOCK) k_render_bwd(int W, int H, int gx, uint32_t T_tiles,
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
    __shared__ f4 s_r0[GHR_BLOCK], s_r1[GHR_BLOCK], s_r2[GHR_BLOCK], s_r3[GHR_BLOCK], s_bb[GHR_BLOCK], s_ep[GHR_BLOCK];
    __shared__ uint32_t s_slot[GHR_BLOCK];
    __shared__ uint32_t s_max[4];

    const uint32_t tile = xcd_tile(blockIdx.x, T_tiles);
    if (tile >= T_tiles) return;  // grid padding
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, l = lane & 15;
    // this lane's pixel: cell (wave, grp) of the tile, 4x4 pixels, lane l -> (l & 3, l >> 2)
    const int px = tx * GHR_TILE_X + 4 * grp + (l & 3), py = ty * GHR_TILE_Y + 4 * wave + (l >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix = (size_t)W * py + px, plane = (size_t)W * H;
    const float wx0 = (float)(tx * GHR_TILE_X);
    const float cy0 = (float)(ty * GHR_TILE_Y + 4 * wave);

    // cap = the capacity the forward ran with = lines in ginst.  It only bites when the forward was launched with a
    // capacity below the true instance count (speculative launch whose results the caller discards): then, like K7,
    // stay inside the lists and inside the gradient buffer.
    const uint32_t beg = min(tile_start[tile], cap);
    const uint32_t n = min(tile_start[tile + 1], cap) - beg;

    PixBwd st;
    st.T_final = inside ? final_T[pix] : 0.f;
    st.T = st.T_final;
    st.S = 0.f;
    st.last_alpha = 0.f;
    st.last_cdot = 0.f;
    st.bgdot = 0.f;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
#pragma unroll
    for (int c = 0; c < GHR_C; c++) {
        const float d = inside ? dL_dpix[c * plane + pix] : 0.f;
        if (c & 1) st.dL[c / 2].y = d; else st.dL[c / 2].x = d;
        st.bgdot = fma_(bg[c], d, st.bgdot);
    }
    const float pu = (float)(4 * grp + (l & 3)), pv = (float)(4 * wave + (l >> 2));  // pixel - tile origin

    // Entries at list positions >= max(n_contrib) are skipped by every pixel concerned (backward.cu:490-492):
    // the tile starts its walk at the tile maximum, and every cell drops the entries above its own maximum.
    uint32_t gmax = last_contributor;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, off));
    uint32_t m = max(gmax, (uint32_t)__shfl_xor((int)gmax, 16));
    m = max(m, (uint32_t)__shfl_xor((int)m, 32));
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    const uint32_t n_eff = min(n, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));

    for (uint32_t base = 0; base < n_eff; base += GHR_BLOCK) {
        const uint32_t cnt = min((uint32_t)GHR_BLOCK, n_eff - base);
        __syncthreads();  // previous batch fully consumed (LDS planes)
        if ((uint32_t)tid < cnt) {
            // walk back to front: batch entry j is list position n_eff-1-(base+j)
            const uint32_t id = point_list[beg + (n_eff - 1 - (base + tid))];
            const f4* r = rec + 4 * (size_t)id;
            const f4 a0 = r[0], a1 = r[1];
            const uint32_t slot = min(beg + (n_eff - 1 - (base + tid)), cap - 1u);  // lines lie in list order
            f4* dst = reinterpret_cast<f4*>(ginst) + 4 * (size_t)slot;  // zero the instance's gradient line
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
            s_slot[tid] = slot;
            s_r0[tid] = a0; s_r1[tid] = a1; s_r2[tid] = r[2]; s_r3[tid] = r[3];
            s_bb[tid] = alpha_bbox(a0, a1);
            s_ep[tid] = ellipse_params(a0, a1);
        }
        __syncthreads();  // also orders the zero-fill (vmcnt(0) + workgroup fence) before the atomics below

        // per-GROUP ordered lists (one 64-bit mask per 64 entries) of the batch entries whose alpha >= 1/255 region
        // touches the group's cell, and -- wave-uniform -- the entries that ALL FOUR cells of the strip visit
        unsigned long long todo[4], common[4];
        int n_common = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t sub = 64u * k;
            todo[k] = 0ull;
            common[k] = 0ull;
            if (sub < cnt) {  // wave-uniform
                const uint32_t e = sub + lane;
                const uint32_t ec = e < cnt ? e : 0;
                unsigned long long m_ = cell_masks(s_bb[ec], s_ep[ec], s_r0[ec], e < cnt, wx0, cy0, grp);
                // entry j sits at list position n_eff-1-(base+j); positions >= gmax are dead for this cell
                const long long jmin = (long long)n_eff - (long long)gmax - (long long)(base + sub);
                if (jmin > 0) m_ = jmin >= 64 ? 0ull : (m_ & (~0ull << jmin));
                todo[k] = m_;
                common[k] = rowmask(m_, 0) & rowmask(m_, 16) & rowmask(m_, 32) & rowmask(m_, 48);
                n_common += __builtin_popcountll(common[k]);
            }
        }
        auto cell_pass = [&](uint32_t j) {  // one entry, one cell (16 lanes)
            const uint32_t pos = n_eff - 1 - (base + j);  // 0-based list position == reference's `contributor`
            float g[16];
            bwd_step(st, pos < last_contributor, pxf, pyf, s_r0[j], s_r1[j], s_r2[j], s_r3[j], pu, pv, g);
            const float v = row_reduce16(g, l);
            // lane l adds component l of the cell's total: 16 lanes -> one 64-B line, resolved in this XCD's L2
            __hip_atomic_fetch_add(ginst + 16 * (size_t)s_slot[j] + l, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (n_common < GHR_COMMON_MIN) {
            // Needle lists (the common case for strands): every cell walks ITS list over the whole batch without
            // waiting for the other three at the 64-entry boundaries (measured on cfg3: 12 % fewer wave passes than
            // re-synchronising per 64 entries).  Divergent per GROUP: the 16 lanes of a DPP row share k / cur.
            int k = 0;
            unsigned long long cur = todo[0];
            for (;;) {
                while (cur == 0ull && k < 3) {
                    k++;
                    cur = k == 1 ? todo[1] : (k == 2 ? todo[2] : todo[3]);
                }
                if (cur == 0ull) break;
                const uint32_t j = 64u * k + (uint32_t)__builtin_ctzll(cur);
                cur &= cur - 1;
                cell_pass(j);
            }
        } else {
            // Entries that all four cells of the strip visit (a splat covering the strip) are walked by the whole wave
            // at once -- scalar loop, broadcast LDS reads, one 64-lane reduction and ONE line of atomics instead of
            // four on the same line (which serialise in L2: measured 38 % of the kernel on isotropic blobs).  The
            // per-pixel order is preserved: between two common entries every cell first finishes its private entries.
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t sub = 64u * k;
                unsigned long long cm = common[k], rest = todo[k] & ~cm;
                for (;;) {
                    const int nc = cm ? __builtin_ctzll(cm) : 64;  // next common entry (wave-uniform)
                    unsigned long long mine = nc < 64 ? (rest & ((1ull << nc) - 1ull)) : rest;
                    rest &= ~mine;
                    while (mine) {  // divergent per GROUP
                        const uint32_t j = sub + (uint32_t)__builtin_ctzll(mine);
                        mine &= mine - 1;
                        cell_pass(j);
                    }
                    if (nc == 64) break;
                    cm &= cm - 1;
                    const uint32_t j = sub + (uint32_t)nc;  // wave-uniform
                    const uint32_t pos = n_eff - 1 - (base + j);
                    float g[16];
                    bwd_step(st, pos < last_contributor, pxf, pyf, s_r0[j], s_r1[j], s_r2[j], s_r3[j], pu, pv, g);
                    const float v = wave_reduce16(g, lane);
                    if ((lane & 3) == 0)
                        __hip_atomic_fetch_add(ginst + 16 * (size_t)s_slot[j] +
                                                   (((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 +
                                                    ((lane >> 2) & 1)),
                                               v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    // list entries no pixel of the tile ever reached (positions >= n_eff): their slots must read as zero
    for (uint32_t i = n_eff + tid; i < n; i += GHR_BLOCK) {
        f4* dst = reinterpret_cast<f4*>(ginst) + 4 * (size_t)min(beg + i, cap - 1u);
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
    }
#endif
}

}  // namespace ghr
