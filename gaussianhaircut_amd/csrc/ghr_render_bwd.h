// ghr_render_bwd.h -- K8: back-to-front gradient walk, one 16x16 tile per workgroup.
// Follows R:cuda_rasterizer/backward.cu:403-561 (renderCUDA).
//
// The reference issues 16 global atomicAdds per contributing (pixel, Gaussian) pair (backward.cu:527,549-558).
// Here the 16 gradient components {mean2D.x, .y, conic.a, .b, .c, opacity, colors[10]} of one list entry are
//   1. reduced across the 64 lanes of a wavefront with a register "halving butterfly":
//        v_permlane32_swap (16 -> 8 regs), v_permlane16_swap (8 -> 4), DPP row_ror:8 (4 -> 2), DPP row_half_mirror
//        (2 -> 1), then two DPP quad_perm adds; lane 4*c ends up holding the wave's total of component c (~35 VALU),
//   2. accumulated across the tile's 4 wavefronts with one 16-lane ds_add_f32 into a per-batch LDS table,
//   3. written once per (tile, Gaussian) INSTANCE as one 64-byte line (4 plain b128 stores) into the instance's own
//      gradient slot (rect4_slot); the per-Gaussian backward sums a Gaussian's slots in a fixed order.
// => no global float atomics at all: the 16 dword-granular device-scope atomics per instance (measured: 40 % of this
//    kernel's time on MI355X, every one a fabric transaction) are gone, and the cross-tile sum has a fixed order (the
//    only order-dependent float sums left are the 4-wave ds_add_f32 accumulations inside one tile).
//
// The per-channel recurrences of the reference (accum_rec[ch], last_color[ch], backward.cu:519-523) are linear in the
// channel index and only ever used through sum_ch(. * dL_dpixel[ch]); they are carried as ONE scalar
//   S = sum_ch accum_rec[ch]*dL_dpixel[ch],   S <- last_alpha*last_cdot + (1-last_alpha)*S,  cdot = sum_ch c[ch]*dL[ch]
// which is the same real-number value (fp32 rounding differs at the 1e-7 level) and frees 20 VGPRs.
#pragma once
#include "ghr_device.h"

namespace ghr {

struct PixBwd {
    float T, T_final, S, last_alpha, last_cdot, bgdot;
    float dL[GHR_C];
};

// One list entry applied to one pixel (backward.cu:494-558).  Writes the 16 per-pair gradient terms to g[] and
// returns true if the pair contributes; g[] is untouched otherwise.
GHR_HD bool bwd_step(PixBwd& s, float pxf, float pyf, const f4& r0, const f4& r1, const f4& r2, const f4& r3,
                     float ddelx_dx, float ddely_dy, float* g)
{
    const float dx = r0.x - pxf, dy = r0.y - pyf;
    const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;  // unfused (decision input)
    if (power > 0.0f) return false;
    const float G = fast_exp(power);
    const float o = r1.y;
    const float alpha = fminf(0.99f, o * G);
    if (alpha < 1.0f / 255.0f) return false;

    const float inv1ma = fast_rcp(1.f - alpha);
    s.T = s.T * inv1ma;  // backward.cu:507
    const float w = alpha * s.T;

    float cdot = r1.z * s.dL[0];
    cdot = fma_(r1.w, s.dL[1], cdot);
    cdot = fma_(r2.x, s.dL[2], cdot);
    cdot = fma_(r2.y, s.dL[3], cdot);
    cdot = fma_(r2.z, s.dL[4], cdot);
    cdot = fma_(r2.w, s.dL[5], cdot);
    cdot = fma_(r3.x, s.dL[6], cdot);
    cdot = fma_(r3.y, s.dL[7], cdot);
    cdot = fma_(r3.z, s.dL[8], cdot);
    cdot = fma_(r3.w, s.dL[9], cdot);
    // backward.cu:519-523 collapsed to scalars (see header)
    s.S = fma_(s.last_alpha, s.last_cdot, (1.f - s.last_alpha) * s.S);
    s.last_cdot = cdot;
    s.last_alpha = alpha;
    float dL_dalpha = (cdot - s.S) * s.T;                                 // :523,:529
    dL_dalpha = fma_(-s.T_final * inv1ma, s.bgdot, dL_dalpha);            // :535-538

    const float dL_dG = o * dL_dalpha;  // :542
    const float gdx = G * dx, gdy = G * dy;
    const float dG_ddelx = -gdx * r0.z - gdy * r0.w;
    const float dG_ddely = -gdy * r1.x - gdx * r0.w;
    g[0] = dL_dG * dG_ddelx * ddelx_dx;  // :549
    g[1] = dL_dG * dG_ddely * ddely_dy;  // :550
    g[2] = -0.5f * gdx * dx * dL_dG;     // :553
    g[3] = -0.5f * gdx * dy * dL_dG;     // :554 (half of d/db; the Python wrapper doubles it)
    g[4] = -0.5f * gdy * dy * dL_dG;     // :555
    g[5] = G * dL_dalpha;                // :558
#pragma unroll
    for (int c = 0; c < GHR_C; c++) g[6 + c] = w * s.dL[c];  // :508,:527
    return true;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel)
{
    // ctrl: 0 = row_ror:8, 1 = row_half_mirror, 2 = quad_perm [2,3,0,1], 3 = quad_perm [1,0,3,2]
    uint32_t u = __float_as_uint(v), p;
    switch (ctrl_sel) {
        case 0: p = __builtin_amdgcn_update_dpp(0u, u, 0x128, 0xf, 0xf, false); break;
        case 1: p = __builtin_amdgcn_update_dpp(0u, u, 0x141, 0xf, 0xf, false); break;
        case 2: p = __builtin_amdgcn_update_dpp(0u, u, 0x4E, 0xf, 0xf, false); break;
        default: p = __builtin_amdgcn_update_dpp(0u, u, 0xB1, 0xf, 0xf, false); break;
    }
    return v + __uint_as_float(p);
}

// Sum 16 per-lane values over the 64 lanes of the wave.  On return lane L holds the total of component
// comp(L) = 8*b5 + 4*b4 + 2*b3 + b2 (b_i = bit i of L), replicated over the 4 lanes of its quad.
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
    float d0 = dpp_add(q[0], 0), d1 = dpp_add(q[1], 0), d2 = dpp_add(q[2], 0), d3 = dpp_add(q[3], 0);
    const float e0 = b3 ? d2 : d0, e1 = b3 ? d3 : d1;  // lanes with bit3 keep q[2],q[3]
    const float f0 = dpp_add(e0, 1), f1 = dpp_add(e1, 1);
    float v = b2 ? f1 : f0;  // lanes with bit2 keep e1
    v = dpp_add(v, 2);
    v = dpp_add(v, 3);
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
                                                          const rect4* __restrict__ rects, float* ginst)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ f4 s_r0[GHR_BLOCK], s_r1[GHR_BLOCK], s_r2[GHR_BLOCK], s_r3[GHR_BLOCK], s_bb[GHR_BLOCK];
    __shared__ uint32_t s_slot[GHR_BLOCK];
    __shared__ float s_acc[GHR_BLOCK * 16];
    __shared__ uint32_t s_max[4];

    const uint32_t tile = xcd_tile(blockIdx.x, T_tiles);
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = tx * GHR_TILE_X + (tid & 15), py = ty * GHR_TILE_Y + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix = (size_t)W * py + px, plane = (size_t)W * H;
    const float sx0 = (float)(tx * GHR_TILE_X), sx1 = sx0 + 15.0f;
    const float sy0 = (float)(ty * GHR_TILE_Y + 4 * wave), sy1 = sy0 + 3.0f;

    const uint32_t beg = tile_start[tile];
    const uint32_t n = tile_start[tile + 1] - beg;

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
        st.dL[c] = inside ? dL_dpix[c * plane + pix] : 0.f;
        st.bgdot = fma_(bg[c], st.dL[c], st.bgdot);
    }
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;  // backward.cu:464-465

    // Entries at list positions >= max_pixels(n_contrib) are skipped by every pixel (backward.cu:490-492): start
    // the walk there instead of at the end of the list.
    uint32_t m = last_contributor;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    const uint32_t n_eff = min(n, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));

    // component handled by this lane after wave_reduce16
    const int comp = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);

    for (uint32_t base = 0; base < n_eff; base += GHR_BLOCK) {
        const uint32_t cnt = min((uint32_t)GHR_BLOCK, n_eff - base);
        __syncthreads();  // previous batch fully consumed (LDS planes + accumulators)
        if ((uint32_t)tid < cnt) {
            // walk back to front: batch entry j is list position n_eff-1-(base+j)
            const uint32_t id = point_list[beg + (n_eff - 1 - (base + tid))];
            const f4* r = rec + 4 * (size_t)id;
            const f4 a0 = r[0], a1 = r[1];
            s_slot[tid] = rect4_slot(rects[id], tx, ty);
            s_r0[tid] = a0; s_r1[tid] = a1; s_r2[tid] = r[2]; s_r3[tid] = r[3];
            s_bb[tid] = alpha_bbox(a0, a1);
        }
        {
            f4* z = reinterpret_cast<f4*>(s_acc) + 4 * tid;
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            z[0] = zero; z[1] = zero; z[2] = zero; z[3] = zero;
        }
        __syncthreads();

        // per-wave ordered list (ballot mask) of the entries whose alpha>=1/255 box touches this wave's 16x4 strip
        for (uint32_t sub = 0; sub < cnt; sub += 64) {
            const uint32_t e = sub + lane;
            const bool hit = e < cnt && bbox_hits(s_bb[e], sx0, sx1, sy0, sy1);
            unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
            while (todo) {
                const uint32_t j = sub + (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t pos = n_eff - 1 - (base + j);  // 0-based list position == reference's `contributor`
                float g[16];
#pragma unroll
                for (int i = 0; i < 16; i++) g[i] = 0.f;
                bool c = false;
                if (pos < last_contributor)
                    c = bwd_step(st, pxf, pyf, s_r0[j], s_r1[j], s_r2[j], s_r3[j], ddelx_dx, ddely_dy, g);
                if (__builtin_amdgcn_ballot_w64(c) != 0) {  // wave-uniform
                    const float v = wave_reduce16(g, lane);
                    if ((lane & 3) == 0) atomicAdd(&s_acc[j * 16 + comp], v);  // ds_add_f32, 16 lanes, 16 banks
                }
            }
        }
        __syncthreads();

        if ((uint32_t)tid < cnt) {  // every staged instance writes its line (zeros when nothing contributed)
            const f4* a4 = reinterpret_cast<const f4*>(s_acc) + 4 * tid;
            f4* dst = reinterpret_cast<f4*>(ginst) + 4 * (size_t)s_slot[tid];
            dst[0] = a4[0]; dst[1] = a4[1]; dst[2] = a4[2]; dst[3] = a4[3];
        }
    }
    // list entries no pixel of the tile ever reached (positions >= n_eff): their slots must read as zero
    for (uint32_t i = n_eff + tid; i < n; i += GHR_BLOCK) {
        const uint32_t id = point_list[beg + i];
        f4* dst = reinterpret_cast<f4*>(ginst) + 4 * (size_t)rect4_slot(rects[id], tx, ty);
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        dst[0] = zero; dst[1] = zero; dst[2] = zero; dst[3] = zero;
    }
#endif
}

}  // namespace ghr
