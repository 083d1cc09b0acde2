// ghr_device.h -- shared device helpers for the gfx950 rasterizer kernels.
//
// Arithmetic discipline: the translation unit is compiled with -ffp-contract=off, so `a*b+c` is two rounded fp32
// operations unless written as __builtin_fmaf.  Everything that feeds a DISCRETE decision of the reference
// (near cull, radius ceil, tile rect truncation, depth key bits, power>0, alpha<1/255, T<1e-4) is written in the
// reference's source order without fusion, so those decisions are reproducible on a CPU; accumulations that only
// feed continuous outputs use explicit FMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GHR_TILE_X 16
#define GHR_TILE_Y 16
#define GHR_BLOCK 256
#define GHR_C 10
#define GHR_REC 16  // floats per packed per-Gaussian render record (one 64-B line)

// Pure math helpers are host+device so that tests/hostsim can run the SAME per-Gaussian / per-pixel code on
// the CPU and compare it with the oracle without a GPU (test scaffolding; the product never runs on the host).
#define GHR_HD __host__ __device__ __forceinline__

namespace ghr {

typedef float f4 __attribute__((ext_vector_type(4)));  // native 16-B vector: b128 loads/stores, SSA-friendly

// ---- reference helpers, R:cuda_rasterizer/auxiliary.h ------------------------------------------------------------

// auxiliary.h:41-44 (double literals => evaluated in double, narrowed once)
GHR_HD float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:58-66
GHR_HD void xform4x3(const float* m, float px, float py, float pz, float& ox,
                                         float& oy, float& oz)
{
    ox = m[0] * px + m[4] * py + m[8] * pz + m[12];
    oy = m[1] * px + m[5] * py + m[9] * pz + m[13];
    oz = m[2] * px + m[6] * py + m[10] * pz + m[14];
}

GHR_HD int imin(int a, int b) { return a < b ? a : b; }
GHR_HD int imax(int a, int b) { return a > b ? a : b; }

// auxiliary.h:46-56 -- radius as int, float divide truncated toward zero, clamped to the grid.
GHR_HD void tile_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1,
                                          int& y1)
{
    x0 = imin(gx, imax(0, (int)((px - radius) / GHR_TILE_X)));
    y0 = imin(gy, imax(0, (int)((py - radius) / GHR_TILE_Y)));
    x1 = imin(gx, imax(0, (int)((px + radius + GHR_TILE_X - 1) / GHR_TILE_X)));
    y1 = imin(gy, imax(0, (int)((py + radius + GHR_TILE_Y - 1) / GHR_TILE_Y)));
}

// 3x3 in glm's column-major storage m[c][r]; product term order of glm's operator* (k = 0,1,2).
struct m3 { float m[3][3]; };
GHR_HD m3 mul(const m3& a, const m3& b)
{
    m3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) o.m[c][r] = a.m[0][r] * b.m[c][0] + a.m[1][r] * b.m[c][1] + a.m[2][r] * b.m[c][2];
    return o;
}
GHR_HD m3 transpose(const m3& a)
{
    m3 o;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) o.m[c][r] = a.m[r][c];
    return o;
}

// Per-Gaussian tile rect + gradient-slot base, one 16-B record: {x0 | x1<<16, y0 | y1<<16, base_lo, base_hi}.
// The Gaussian's (y - y0) * (x1 - x0) + (x - x0)-th tile instance owns gradient slot base_lo + base_hi + that ordinal,
// where base_lo is the exclusive prefix of the rect areas inside the Gaussian's 256-thread K1 workgroup and base_hi the
// exclusive prefix over the workgroups before it (k_tile_scan; copied in by k_scatter).  No atomics: a single
// allocation counter would serialise every wavefront of K1 on one address (measured: ~90 us at 500k Gaussians).
typedef uint4 rect4;
GHR_HD rect4 make_rect4(int x0, int y0, int x1, int y1, uint32_t base)
{
    return rect4{(uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16), base, 0u};
}
GHR_HD uint32_t rect4_area(const rect4& r)
{
    return ((r.x >> 16) - (r.x & 0xffffu)) * ((r.y >> 16) - (r.y & 0xffffu));
}
// gradient slot of Gaussian `r` in tile (tx, ty); the tile must lie inside the rect
GHR_HD uint32_t rect4_slot(const rect4& r, int tx, int ty)
{
    const uint32_t x0 = r.x & 0xffffu, x1 = r.x >> 16, y0 = r.y & 0xffffu;
    return r.z + r.w + ((uint32_t)ty - y0) * (x1 - x0) + ((uint32_t)tx - x0);
}

// ---- instance gradient lines ---------------------------------------------------------------------------------------
// k_render_bwd / k_render_bwd_scan leave ONE 64-B line per (tile, Gaussian) instance.  Its first six floats are not the
// reference's gradient terms but the raw pixel sums they are linear in (Q = G * dL/dalpha of a contributing pair,
// d = mean - pixel, (u, v) = pixel - tile origin):
//     L0 = sum Q dx     L1 = sum Q dy     L2 = sum Q dx u     L3 = sum Q dx v     L4 = sum Q dy v     L5 = sum Q
// followed by the ten colour sums  sum alpha T dL/dpixel[ch]  (backward.cu:527).  u and v do not depend on the
// Gaussian, which is what lets the scan kernel form the sums as small f32 matrix products (ghr_render_bwd2.h); the
// second-order sums are recovered here per instance with e = mean - tile origin (d = e - (u, v)):
//     sum Q dx^2 = ex L0 - L2     sum Q dx dy = ey L0 - L3     sum Q dy^2 = ey L1 - L4
// (first-order cancellation only: |e| <= a few tiles against |d| of a contributing pixel), and the reference's terms
// (backward.cu:542-558) follow per Gaussian from the conic (a, b, c) and opacity o:
//     dmean2D.x = o (-a S0 - b S1) W/2   dmean2D.y = o (-c S1 - b S0) H/2   dconic = -o/2 (Sxx, Sxy, Syy)   dopacity = S5
struct LineAcc {
    float s0, s1, sxx, sxy, syy, s5;
    f4 c0, c1, c2;  // c0.zw, c1, c2: the ten colour sums (c0.xy unused)
};
GHR_HD void line_acc_init(LineAcc& A)
{
    A.s0 = A.s1 = A.sxx = A.sxy = A.syy = A.s5 = 0.f;
    A.c0 = f4{0.f, 0.f, 0.f, 0.f}; A.c1 = A.c0; A.c2 = A.c0;
}
// add the line (l0..l3) of the instance in tile (tx, ty); mx, my = the Gaussian's pixel mean
GHR_HD void line_acc_add(LineAcc& A, const f4& l0, const f4& l1, const f4& l2, const f4& l3, float mx, float my, uint32_t tx,
                         uint32_t ty)
{
    const float ex = mx - (float)(GHR_TILE_X * tx), ey = my - (float)(GHR_TILE_Y * ty);
    A.s0 += l0.x;
    A.s1 += l0.y;
    A.sxx += __builtin_fmaf(ex, l0.x, -l0.z);
    A.sxy += __builtin_fmaf(ey, l0.x, -l0.w);
    A.syy += __builtin_fmaf(ey, l0.y, -l1.x);
    A.s5 += l1.y;
    A.c0 += l1; A.c1 += l2; A.c2 += l3;
}
// the 16 gradient terms of the reference's backward.cu:527,549-558, summed over the Gaussian's instances
GHR_HD void line_acc_finish(const LineAcc& A, const f4& r0, const f4& r1, float half_w, float half_h, float* ga)
{
    const float a = r0.z, b = r0.w, c = r1.x, o = r1.y;
    ga[0] = o * (-a * A.s0 - b * A.s1) * half_w;
    ga[1] = o * (-c * A.s1 - b * A.s0) * half_h;
    const float h = -0.5f * o;
    ga[2] = h * A.sxx; ga[3] = h * A.sxy; ga[4] = h * A.syy;
    ga[5] = A.s5;
    ga[6] = A.c0.z; ga[7] = A.c0.w; ga[8] = A.c1.x; ga[9] = A.c1.y; ga[10] = A.c1.z; ga[11] = A.c1.w;
    ga[12] = A.c2.x; ga[13] = A.c2.y; ga[14] = A.c2.z; ga[15] = A.c2.w;
}

// Sum of a Gaussian's per-instance gradient lines in tile-ordinal order (deterministic), converted to the reference's
// gradient terms.  r0 / r1: the first two 16-B pieces of the Gaussian's render record (pixel mean, conic, opacity).
// `rows`: number of lines `ginst` holds.  A Gaussian whose lines would reach past it (only possible when the forward ran
// with a capacity below the true instance count, whose results the caller discards) reads nothing.
GHR_HD void gather_inst_grads(const float* ginst, const uint32_t* inst_line, const rect4& r, const f4& r0, const f4& r1,
                              float half_w, float half_h, float* ga, uint32_t rows = 0xffffffffu)
{
    uint32_t cnt = rect4_area(r);
    if ((uint64_t)r.z + r.w + cnt > (uint64_t)rows) cnt = 0;
    // the Gaussian's instances are numbered r.z + r.w + ordinal(tile in its rect) (rect4_slot); their gradient lines lie
    // in TILE-LIST order (line of an instance = its position in the sorted lists, where K8 finds it without a lookup):
    // inst_line[instance] is that position, written by the tile sort
    const uint32_t* il = inst_line + ((size_t)r.z + r.w);
    const f4* lines = reinterpret_cast<const f4*>(ginst);
    const uint32_t x0 = r.x & 0xffffu, wdt = (r.x >> 16) - x0, y0 = r.y & 0xffffu;
    LineAcc A;
    line_acc_init(A);
    // four lines (16 independent 16-B loads) are requested per round trip: the loop is pure memory latency, and one
    // line per trip cost cnt_max-of-the-wave serialized trips.  Lines past `cnt` add +0; the sum order (ascending ordinal)
    // is unchanged.  Positions past `cnt` READ the instance's last line again (a cache hit) and drop it: predicated loads
    // (`in ? p[q] : z`) compiled to one EXEC-masked branch per load with a full vmcnt(0) drain in the middle of the batch --
    // a third round trip per pass (round 5).
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    uint32_t tx = 0, ty = 0;  // position of the instance's tile inside the rect
    for (uint32_t k = 0; k < cnt; k += 4) {  // (inside the loop cnt >= 1 and rows >= cnt: line 0 exists)
        uint32_t ln[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t v_ = il[k + j < cnt ? k + j : cnt - 1u];
            ln[j] = v_ < rows ? v_ : 0u;  // (a stale entry after a too small capacity: memory-safe, the result is discarded)
        }
        f4 v[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f4* p = lines + 4 * (size_t)ln[j];
#pragma unroll
            for (int q = 0; q < 4; q++) v[4 * j + q] = p[q];
        }
#pragma unroll
        for (int j = 1; j < 4; j++)
            if (!(k + j < cnt)) v[4 * j] = v[4 * j + 1] = v[4 * j + 2] = v[4 * j + 3] = z;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            line_acc_add(A, v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3], r0.x, r0.y, x0 + tx, y0 + ty);
            if (++tx == wdt) { tx = 0; ty++; }
        }
    }
    line_acc_finish(A, r0, r1, half_w, half_h, ga);
}

// Rects with more tiles than this are not walked by their own lane: a lane's loop over a 20 x 20-tile splat (one atomic
// or one 64-B line per step, dependent latency each) used to set the duration of the whole kernel (cfg2: the largest of
// 97 k random blobs covers 484 tiles; k_preprocess 130 us, k_scatter 99 us, k_geom_bwd 79 us).  The wave walks them
// together, 64 tiles per step.  Thresholds measured on cfg2 (8 / 16 / 32): counting + scatter 73 / 96 / 106 us, gather
// 92 / 39 / 29 us (its cooperative step ends in a 64-lane butterfly over 16 values); cfg3's needles (<= 9 tiles) do not care.
#define GHR_BIG_RECT 8     // tile counting, scatter
#define GHR_BIG_GATHER 32  // gradient-line gather

#if defined(__HIP_DEVICE_COMPILE__)
// Sum of the instance gradient lines for every lane's Gaussian.  Small rects: per lane (gather_inst_grads, ascending
// ordinal).  Big rects: one at a time by the whole wave -- lane l sums the lines l, l+64, ... (coalesced), then a
// butterfly over the 64 lanes; the order differs from the sequential one but is fixed.  All lanes of the wave call it.
__device__ __forceinline__ void gather_inst_grads_wave(const float* ginst, const uint32_t* inst_line, const rect4& r,
                                                       const f4& r0, const f4& r1, float half_w, float half_h, float* ga,
                                                       uint32_t rows)
{
    const int lane = threadIdx.x & 63;
    const uint32_t cnt = rect4_area(r);
    const bool big = cnt > GHR_BIG_GATHER;
    rect4 small = r;
    if (big) { small.x = 0u; small.y = 0u; }  // empty rect: nothing to read
    gather_inst_grads(ginst, inst_line, small, r0, r1, half_w, half_h, ga, rows);
    unsigned long long todo = __builtin_amdgcn_ballot_w64(big);
    while (todo) {  // wave-uniform
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t first = (uint32_t)__shfl((int)(r.z + r.w), src);
        uint32_t n = (uint32_t)__shfl((int)cnt, src);
        if ((uint64_t)first + n > (uint64_t)rows) n = 0;  // see gather_inst_grads
        const uint32_t rx = (uint32_t)__shfl((int)r.x, src), ry = (uint32_t)__shfl((int)r.y, src);
        const float mx = __shfl(r0.x, src), my = __shfl(r0.y, src);
        const uint32_t x0 = rx & 0xffffu, wdt = (rx >> 16) - x0, y0 = ry & 0xffffu;
        LineAcc A;
        line_acc_init(A);
        for (uint32_t k = lane; k < n; k += 64) {
            const uint32_t ln = inst_line[(size_t)first + k];
            const f4* q = reinterpret_cast<const f4*>(ginst) + 4 * (size_t)(ln < rows ? ln : 0u);
            line_acc_add(A, q[0], q[1], q[2], q[3], mx, my, x0 + k % wdt, y0 + k / wdt);
        }
        float v[16] = {A.s0, A.s1, A.sxx, A.sxy, A.syy, A.s5, A.c0.z, A.c0.w, A.c1.x, A.c1.y, A.c1.z, A.c1.w,
                       A.c2.x, A.c2.y, A.c2.z, A.c2.w};
#pragma unroll
        for (int c = 0; c < 16; c++)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v[c] += __shfl_xor(v[c], off);
        if (lane == src) {
            LineAcc S;
            S.s0 = v[0]; S.s1 = v[1]; S.sxx = v[2]; S.sxy = v[3]; S.syy = v[4]; S.s5 = v[5];
            S.c0 = f4{0.f, 0.f, v[6], v[7]}; S.c1 = f4{v[8], v[9], v[10], v[11]}; S.c2 = f4{v[12], v[13], v[14], v[15]};
            line_acc_finish(S, r0, r1, half_w, half_h, ga);
        }
    }
}

// Exclusive prefix sum of n over the 256 threads of the workgroup (thread order); *total = sum.  Every thread of the
// workgroup must call it.  s_tmp: 4 LDS words.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t n, uint32_t* s_tmp, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    const uint32_t w0 = s_tmp[0], w1 = s_tmp[1], w2 = s_tmp[2], w3 = s_tmp[3];
    const uint32_t before = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    *total = w0 + w1 + w2 + w3;
    return before + incl - n;
}

// Big rects (more than GHR_BIG_RECT tiles) of a 256-thread workgroup, LOAD-BALANCED (round 5): the workgroup scans their
// areas and every thread takes the instances j = tid, tid + 256, ... of the concatenation -- owner by binary search over the
// scanned offsets, tile from the ordinal inside the owner's rect -- so a workgroup with a handful of 20 x 20-tile splats
// spends total / 256 steps per thread, all of them independent, instead of walking the rects one after the other with one
// wave each (cfg2, 100k blobs of 10.6 tiles on average: k_preprocess 40 us for 380 workgroups; the per-wave walk had
// replaced a per-LANE walk that was 3x slower still).  Shared by the tile counting (K1) and the scatter.
struct BigRects {
    uint32_t off[GHR_BLOCK + 1];  // exclusive prefix of the big rects' areas in thread order; off[256] = total
    uint32_t xy[GHR_BLOCK];       // x0 | y0 << 16
    uint32_t w[GHR_BLOCK];        // rect width in tiles
    uint32_t klo[GHR_BLOCK], khi[GHR_BLOCK];  // (scatter) the owner's 64-bit key
    uint32_t tmp[4];
};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t n, uint32_t* s_tmp, uint32_t* total);
// Every thread of the workgroup calls it with its own rect (area 0: not a big rect).  Returns the total number of instances;
// afterwards (total > 0) thread-independent lookups big_rect_instance(j) are valid until the next barrier-separated reuse.
__device__ __forceinline__ uint32_t big_rects_setup(BigRects& s, uint32_t area, int x0, int y0, int w, uint32_t klo = 0u,
                                                   uint32_t khi = 0u)
{
    uint32_t total;
    const uint32_t off = block_excl_scan_256(area, s.tmp, &total);
    if (total == 0u) return 0u;  // (workgroup-uniform)
    const int tid = threadIdx.x;
    s.off[tid] = off;
    s.xy[tid] = (uint32_t)x0 | ((uint32_t)y0 << 16);
    s.w[tid] = (uint32_t)w;
    s.klo[tid] = klo;
    s.khi[tid] = khi;
    if (tid == 0) s.off[GHR_BLOCK] = total;
    __syncthreads();
    return total;
}
// instance j of the concatenation: its owner thread (the last one whose offset is <= j: it has a non-empty rect) and tile
__device__ __forceinline__ uint32_t big_rect_instance(const BigRects& s, uint32_t j, int gx, uint32_t& owner)
{
    uint32_t lo = 0u, hi = GHR_BLOCK;  // off[lo] <= j < off[hi]
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s.off[mid] <= j) lo = mid; else hi = mid;
    }
    owner = lo;
    const uint32_t ord = j - s.off[lo], w = s.w[lo], xy = s.xy[lo];
    const uint32_t ky = ord / w, kx = ord - ky * w;
    return ((xy >> 16) + ky) * (uint32_t)gx + (xy & 0xffffu) + kx;
}
#endif

// Run-aggregated atomicAdd(&counter[t], 1) for every lane with `active`: a run of ADJACENT lanes that target the same
// counter issues ONE atomic (strand Gaussians that are neighbours in memory are neighbours on screen: a wave's 64
// increments collapse to a handful; incoherent inputs degrade to one atomic per lane plus ~15 VALU).  Returns the value
// the counter had before THIS lane's increment (as a per-lane atomicAdd would).  Every lane of the wave must call it.
__device__ __forceinline__ uint32_t wave_inc(uint32_t* counter, uint32_t t, bool active, bool want_result = true)
{
    const int lane = threadIdx.x & 63;
    const uint32_t key = active ? t : 0xffffffffu;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    const bool leader = lane == 0 || prev != key;
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(leader);
    const unsigned long long upto = (2ull << lane) - 1ull;             // bits 0..lane (lane 63: all ones)
    const int lead = 63 - __builtin_clzll(lm & upto);                   // this lane's run leader (bit 0 is always set)
    const unsigned long long after = lm & ~((2ull << lead) - 1ull);     // leaders above the run
    const int end = after ? __builtin_ctzll(after) : 64;
    uint32_t base = 0;
    if (leader && active) base = atomicAdd(&counter[t], (uint32_t)(end - lead));
    if (!want_result) return 0u;
    base = (uint32_t)__shfl((int)base, lead);
    return base + (uint32_t)(lane - lead);
}
// wave_inc in two halves, so that SEVERAL returning atomics of a lane are in flight together (k_scatter: a lane's chain of
// dependent round trips was the kernel): _issue sends the run leaders' atomics and returns the raw result (valid in the
// leader lanes only, not yet waited for) with the lane's leader in `lead`; _result broadcasts it inside the run.
__device__ __forceinline__ uint32_t wave_inc_issue(uint32_t* counter, uint32_t t, bool active, int& lead)
{
    const int lane = threadIdx.x & 63;
    const uint32_t key = active ? t : 0xffffffffu;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    const bool leader = lane == 0 || prev != key;
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(leader);
    const unsigned long long upto = (2ull << lane) - 1ull;
    lead = 63 - __builtin_clzll(lm & upto);
    const unsigned long long after = lm & ~((2ull << lead) - 1ull);
    const int end = after ? __builtin_ctzll(after) : 64;
    uint32_t base = 0;
    if (leader && active) base = atomicAdd(&counter[t], (uint32_t)(end - lead));
    return base;
}
__device__ __forceinline__ uint32_t wave_inc_result(uint32_t base, int lead)
{
    const int lane = threadIdx.x & 63;
    return (uint32_t)__shfl((int)base, lead) + (uint32_t)(lane - lead);
}
#endif

// ---- chip mapping ----------------------------------------------------------------------------------------------------

// Workgroup b is dispatched to XCD b % 8 (observed, speed only).  The row-major tile list is cut into runs of
// GHR_XCD_RUN tiles that are dealt round-robin to the XCDs: neighbouring tiles (which share Gaussians) hit the same
// private L2, and every XCD gets a slice of every image region (one contiguous band per XCD left the XCDs that own
// the empty top / bottom of the frame idle while the middle ones worked: SQ busy 78-83 % of K7 / K8).
// The grid is padded to a whole number of runs per XCD: xcd_grid(n) workgroups, those with xcd_tile() >= n are idle.
// Measured on cfg3 (1080p): K8 289 -> 261 us, K7 132 -> 119 us, tile sort 48 -> 44 us; run lengths 1 / 8 / 32 / 128 are
// within 2 % of each other, 8 keeps small images (256 tiles) spread over all XCDs.
#ifndef GHR_XCD_RUN
#define GHR_XCD_RUN 8
#endif
GHR_HD uint32_t xcd_grid(uint32_t n)
{
    const uint32_t runs = (n + GHR_XCD_RUN - 1) / GHR_XCD_RUN;
    return (runs + 7u) / 8u * 8u * GHR_XCD_RUN;
}
GHR_HD uint32_t xcd_tile(uint32_t b, uint32_t n)
{
    (void)n;
    const uint32_t xcd = b & 7u, k = b >> 3, j = k / GHR_XCD_RUN, o = k % GHR_XCD_RUN;
    return (j * 8u + xcd) * GHR_XCD_RUN + o;
}

// Cell masks (written by k_render_fwd, read by k_render_bwd_cells): one group of 16 64-bit words (one per 4x4-pixel
// cell of the tile) per 64 list positions.  Tile t's groups start at mask_word0(tile_start[t], t): floor(start / 64) + t
// leaves every tile ceil(n / 64) groups of its own; mask_groups(R, T) groups in all.
GHR_HD size_t mask_word0(uint32_t tile_beg, uint32_t tile) { return (size_t)(tile_beg >> 6) + tile; }
GHR_HD size_t mask_groups(size_t R, size_t T) { return (R >> 6) + T + 1; }

// Phase timing for kernel experiments (tools/kbench.py against a -DGHR_K8_PROF build): shader-clock cycles per phase of
// every wave, one slot per wave (no atomics), summed on the host.  Compiled out of the product.
#ifdef GHR_K8_PROF
#define GHR_PROF_SLOTS 65536
__device__ unsigned long long g_k8_prof[8 * GHR_PROF_SLOTS];
__device__ unsigned long long g_k8_tl[2 * GHR_PROF_SLOTS];  // per wave: start timestamp, HW_ID | XCC_ID << 32
#define GHR_PROF_DECL                                                           \
    unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_t0 = prof_t; \
    unsigned long long prof_a[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GHR_PROF(i)                                                 \
    do {                                                            \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        prof_a[i] += t_ - prof_t;                                   \
        prof_t = t_;                                                \
    } while (0)
#define GHR_PROF_COUNT(i, n) do { prof_a[i] += (unsigned long long)(n); } while (0)
#define GHR_PROF_END(i)                                                                   \
    do {                                                                                  \
        prof_a[i] = __builtin_amdgcn_s_memtime() - prof_t0;                               \
        const uint32_t w_ = (blockIdx.x * 4u + (threadIdx.x >> 6)) % GHR_PROF_SLOTS;      \
        if ((threadIdx.x & 63) == 0) {                                                    \
            for (int q_ = 0; q_ < 8; q_++) g_k8_prof[8 * w_ + q_] = prof_a[q_];           \
            g_k8_tl[2 * w_] = prof_t0;                                                    \
            g_k8_tl[2 * w_ + 1] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |  \
                                  ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
        }                                                                                 \
    } while (0)
#else
#define GHR_PROF_DECL
#define GHR_PROF(i)
#define GHR_PROF_COUNT(i, n)
#define GHR_PROF_END(i)
#endif

// v_exp_f32 / v_rcp_f32 (1 ulp each); exp(x) = 2^(x*log2 e) carries a few ulp more from the rounded product.
GHR_HD float fast_exp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
#else
    return exp2f(x * 1.4426950408889634f);
#endif
}
// v_sqrt_f32 (1 ulp) instead of the correctly rounded expansion (~8 VALU)
GHR_HD float fast_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
GHR_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
GHR_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// v_log_f32 (log2, 1 ulp) * ln 2 instead of the correctly rounded logf expansion (~25 VALU)
GHR_HD float fast_log(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 0.6931471805599453f * __builtin_amdgcn_logf(x);
#else
    return logf(x);
#endif
}

// ---- exact culling of list entries per pixel strip ---------------------------------------------------------------------
// A splat only contributes to pixels where alpha = min(.99, o*exp(power)) >= 1/255 (forward.cu:369-371), i.e. inside the
// ellipse  d^T C d <= 2 ln(255 o).  The reference walks every pixel of every tile in the 3-sigma SQUARE of the largest
// eigenvalue (forward.cu:254-260), so for strand-aligned (needle) Gaussians most visited pixels fail that test.  This
// returns a CONSERVATIVE axis-aligned box {xmin, xmax, ymin, ymax} (pixel coordinates) of that ellipse; pixels outside
// it are guaranteed to be skipped by the per-pixel test, so not visiting them leaves every output bit-identical.
// Margins: 1 % + 0.01 px on the half extents (>= 2e-5 relative slack on alpha, vs ~5e-7 error of the fast exp);
// degenerate cases (opacity within 1e-3 of the threshold, non-positive-definite conic, non-finite) return the full plane.
GHR_HD f4 alpha_bbox(const f4& r0, const f4& r1)
{
    const float BIG = 3.0e38f;
    const float o = r1.y, cx = r0.z, cy = r0.w, cz = r1.x;
    if (o < 0.999f * (1.0f / 255.0f)) return f4{BIG, -BIG, BIG, -BIG};  // alpha <= o < 1/255 everywhere: empty
    // hardware log / rcp / sqrt (1 ulp each) on the device: the margins below exceed their error by four orders of
    // magnitude, and a cull only has to be conservative
    const float L = fast_log(255.0f * o);
    const float det = cx * cz - cy * cy;
    if (!(L >= 1.0e-3f) || !(det > 0.0f) || !(cx > 0.0f) || !(cz > 0.0f)) return f4{-BIG, BIG, -BIG, BIG};
    const float k = 2.0f * L * fast_rcp(det);
    const float hx = 1.01f * fast_sqrt(k * cz) + 0.01f, hy = 1.01f * fast_sqrt(k * cx) + 0.01f;
    if (!(hx < BIG) || !(hy < BIG)) return f4{-BIG, BIG, -BIG, BIG};
    return f4{r0.x - hx, r0.x + hx, r0.y - hy, r0.y + hy};
}
GHR_HD bool bbox_hits(const f4& bb, float x0, float x1, float y0, float y1)
{
    return !(bb.y < x0 || bb.x > x1 || bb.w < y0 || bb.z > y1);
}

// Second, tighter stage of the cull: the alpha >= 1/255 region is the ELLIPSE q(d) = cx dx^2 + 2 cy dx dy + cz dy^2
// <= thr = 2 ln(255 o) (d = mean - pixel), and its bounding box over-covers badly for diagonal needles and for blobs
// (corners).  A 4x4 cell spans the full height of its wave's 4-row band, so a cell can hold a contributing pixel only
// if the x-extent of (ellipse ∩ band) overlaps the cell's columns.  That extent is exact and cheap: for a fixed dy the
// ellipse covers dx in (-cy dy -+ sqrt(cx thr - det dy^2)) / cx, and the right / left-most points over the band are
// reached at dy = clamp(-+k, band) with k = cy hx / cz the dy of the ellipse's own extreme points.
// ellipse_params returns {cx*thr, det, 1/cx, k}; thr carries a 2 % + 0.05 margin (fp32 evaluation order differs from
// the per-pixel `power`), the interval another 0.02 px.  Conics that are not comfortably positive definite or extremely
// anisotropic (cancellation in det) get det = -1 (box test only).
GHR_HD f4 ellipse_params(const f4& r0, const f4& r1)
{
    const float o = r1.y, cx = r0.z, cy = r0.w, cz = r1.x;
    const float det = cx * cz - cy * cy;
    const float L = fast_log(255.0f * fmaxf(o, 1.0e-30f));
    if (!(L >= 1.0e-3f) || !(cx > 0.0f) || !(cz > 0.0f) || !(det > 1.0e-4f * cx * cz) || !(L < 100.0f))
        return f4{0.f, -1.f, 0.f, 0.f};
    const float thr = 2.04f * L + 0.05f;
    const float hx = fast_sqrt(thr * cz * fast_rcp(det));
    const float k = cy * hx * fast_rcp(cz);
    if (!(hx < 3.0e38f) || !(fabsf(k) < 3.0e38f)) return f4{0.f, -1.f, 0.f, 0.f};
    return f4{cx * thr, det, fast_rcp(cx), k};
}
// x-extent [lo, hi] (in d = mean - pixel) of the ellipse restricted to the band dy in [ay, by]; the band must
// intersect the ellipse's y-extent (the box test guarantees it).  Degenerate conics: the whole axis.
GHR_HD void ellipse_band_extent(float cy, const f4& ep, float ay, float by, float& lo, float& hi)
{
    if (!(ep.y > 0.0f)) { lo = -3.0e38f; hi = 3.0e38f; return; }
    const float dyr = fminf(by, fmaxf(ay, -ep.w)), dyl = fminf(by, fmaxf(ay, ep.w));
    const float Dr = fmaxf(ep.x - ep.y * dyr * dyr, 0.0f), Dl = fmaxf(ep.x - ep.y * dyl * dyl, 0.0f);
    hi = (-cy * dyr + fast_sqrt(Dr)) * ep.z + 0.02f;
    lo = (-cy * dyl - fast_sqrt(Dl)) * ep.z - 0.02f;
}
// Full per-cell test used by the render kernels (and, pixel by pixel, by tests/hostsim): box, then ellipse extent.
// (X0, Y0) = pixel coordinates of the cell's first pixel; the cell spans X0..X0+3, Y0..Y0+3.
GHR_HD bool cell_hit(const f4& bb, const f4& ep, const f4& r0, float X0, float Y0)
{
    if (!bbox_hits(bb, X0, X0 + 3.0f, Y0, Y0 + 3.0f)) return false;
    float lo, hi;
    ellipse_band_extent(r0.w, ep, r0.y - (Y0 + 3.0f), r0.y - Y0, lo, hi);
    return !(hi < r0.x - (X0 + 3.0f) || lo > r0.x - X0);
}

#if defined(__HIP_DEVICE_COMPILE__)
// Per-lane 64-bit list of the batch entries sub..sub+63 whose alpha >= 1/255 region can touch this lane's cell: lane L
// tests entry sub+L against the wave's four cells (X0 = wx0 + 4g, shared band Y0 = cy0 .. cy0+3): one band extent,
// four interval overlaps, four ballots; every lane keeps the ballot of its own group.
__device__ __forceinline__ unsigned long long cell_masks(const f4& bb, const f4& ep, const f4& r0, bool valid, float wx0,
                                                         float cy0, int grp)
{
    const bool yhit = valid && !(bb.w < cy0 || bb.z > cy0 + 3.0f);
    float lo, hi;
    ellipse_band_extent(r0.w, ep, r0.y - (cy0 + 3.0f), r0.y - cy0, lo, hi);
    // pixel-space x-interval of the band-restricted ellipse, intersected with the box
    const float xl = fmaxf(bb.x, r0.x - hi), xr = fminf(bb.y, r0.x - lo);
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(yhit && !(xr < wx0 || xl > wx0 + 3.0f));
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(yhit && !(xr < wx0 + 4.0f || xl > wx0 + 7.0f));
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(yhit && !(xr < wx0 + 8.0f || xl > wx0 + 11.0f));
    const unsigned long long m3 = __builtin_amdgcn_ballot_w64(yhit && !(xr < wx0 + 12.0f || xl > wx0 + 15.0f));
    return grp == 0 ? m0 : (grp == 1 ? m1 : (grp == 2 ? m2 : m3));
}
#endif

}  // namespace ghr
