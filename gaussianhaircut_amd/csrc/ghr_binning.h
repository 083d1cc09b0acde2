// ghr_binning.h -- tile binning: offset scan, instance scatter, per-tile depth sort.
//
// Replaces the reference's K2-K6 (R:cuda_rasterizer/rasterizer_impl.cu:70-138,281-321): inclusive scan over P,
// duplicateWithKeys, a global 64-bit (tile|depth) stable radix sort of R pairs (6+ passes x 24 B/instance) and
// identifyTileRanges.  Because the sorted list is only ever consumed tile by tile, the same result is produced
// here with one pass each:
//   k_tile_scan : exclusive scan of the T per-tile counts -> tile_start[T+1]  (== `ranges`, rasterizer_impl.cu:116-138)
//   k_scatter   : each Gaussian appends (depth_bits<<32 | idx) to the lists of the tiles in its rect (8 B/instance)
//   k_tile_sort : one workgroup sorts one tile's list in LDS by (depth_bits, idx) and emits point_list (4 B/instance)
// Equivalence: the reference's key is (tile<<32 | depth_bits), sorted stably, with instances emitted in ascending
// Gaussian index (rasterizer_impl.cu:88-108), so within a tile its order is "depth bits ascending, ties by ascending
// idx" -- exactly the total order of the 64-bit key used here.  point_list and ranges are therefore bit-identical
// to the reference's, independent of the (non-deterministic) order in which k_scatter appends.
#pragma once
#include "ghr_device.h"

namespace ghr {

#define GHR_SCAN_BLOCK 1024
#define GHR_SORT_CAP 1024  // keys one wave sorts in LDS (8.5 KiB); longer lists: k_tile_sort_big / the in-place global path
#define GHR_SORT_BIG_CAP 8192   // keys per LDS block of k_tile_sort_big (64 KiB)
#define GHR_SORT_BIG_BLOCK 1024
#define GHR_SORT_BIG_MIN_AVG 256  // k_tile_sort_big is launched when the lists average at least this many instances
#define GHR_SORT_MID_CAP 4096   // k_tile_sort_mid (round 6): lists of GHR_SORT_CAP + 1 .. this many keys (34.8 KiB of LDS)
#ifndef GHR_SORT_MID_WAVES
#define GHR_SORT_MID_WAVES 8    // per SIMD the register allocation of k_tile_sort_mid aims at (64 VGPRs, one spilled; 6: 66 VGPRs)
#endif
#ifndef GHR_SORT_MID_EMIT
#define GHR_SORT_MID_EMIT 2     // entries whose rect gathers are in flight together when the list is written out (4 in k_tile_sort)
#endif
#ifndef GHR_SORT_MID_SPLIT
#define GHR_SORT_MID_SPLIT 0    // 1: 256 threads for lists up to 2048 keys, 512 for the rest (two launches whose critical paths add:
                                // strand stage 162 + 51 us against 169, cfg5 71 + 35 against 101: profiles/r06q); 0: 512 threads for all
#endif
#define GHR_SORT_WALK_MAX 64    // tiles one workgroup of the dense-tile kernels may have to look at (grid >= T / this)
#define GHR_SORT_DONE 0xffffffffu  // tile_cursor value k_tile_sort_big leaves for k_tile_sort: "this tile is sorted"
#define GHR_SORT_BLOCK 128  // two waves per tile (round 5; 256 threads and a barrier per step before)
#define GHR_SORT_SOLO 256u  // lists up to this long are sorted by wave 0 alone

#if defined(__HIP_DEVICE_COMPILE__)
// Exclusive scan of src[n] into dst[n] (may alias) by one 1024-thread workgroup; returns the total in every thread.
// `zero_src`: reset src[i] to 0 after reading it.
__device__ __forceinline__ uint32_t scan_1024(int n, uint32_t* src, uint32_t* dst, bool zero_src, uint32_t* s_part)
{
    const int tid = threadIdx.x;
    const int per = (n + GHR_SCAN_BLOCK - 1) / GHR_SCAN_BLOCK;
    const int b = tid * per, e = min(n, b + per);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += src[i];
    // inclusive scan of the 1024 partials: shuffles inside each of the 16 wavefronts, then the 16 wave totals
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    __syncthreads();  // s_part may still be read by a previous call
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < GHR_SCAN_BLOCK / 64 ? s_part[lane] : 0u;
#pragma unroll
        for (int off = 1; off < GHR_SCAN_BLOCK / 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)w, off);
            if (lane >= off) w += v;
        }
        if (lane < GHR_SCAN_BLOCK / 64) s_part[64 + lane] = w;  // inclusive prefix of the wave totals
    }
    __syncthreads();
    uint32_t run = incl - sum + (wave ? s_part[64 + wave - 1] : 0u);  // exclusive prefix of this thread's chunk
    for (int i = b; i < e; i++) {
        const uint32_t c = src[i];
        if (zero_src) src[i] = 0u;
        dst[i] = run;
        run += c;
    }
    return s_part[64 + GHR_SCAN_BLOCK / 64 - 1];
}
#endif

// tile_count[2][T] -> tile_start[T+1] (exclusive scan of small + big) and small_cnt[T] (the small rects' counts: their
// instances already have their places, k_scatter appends the big rects' behind them); both planes are reset to 0 -- the second
// one becomes k_scatter's append cursors; publishes R = tile_start[T]) and, in place, slot_blk[nblk] -> exclusive prefix of the gradient slots used by the K1 workgroups.
// Also leaves tile_order[xcd_grid(T)] (round 3): the tile each workgroup of the tile sort and of K7 takes (K8 balances its
// cells inside a tile and gains nothing: measured, MI355X, cfg3 / cfg5 / cfg2: K7 0.115 -> 0.100 / 0.320 -> 0.276 /
// 0.131 -> 0.124 ms, K8 0.190 -> 0.189 / 0.499 -> 0.485 / 0.310 -> 0.313 ms; single-view step 0.930 -> 0.909 ms).
// Workgroup b runs on XCD b % 8 and keeps the tiles xcd_tile() deals to that XCD (neighbouring tiles -- which share
// Gaussians -- stay on one L2), but within the XCD the HEAVIEST lists go first (64 classes of n / 16, counting sort): a
// launch is ~5 rounds of workgroups per CU and in raster order its last round is of average weight -- the model of
// tools/cellstats/schedule_sim.py puts the launch at 1.16x (cfg3) / 1.10x (cfg2) of perfect packing, heaviest-first per
// XCD at 1.09x / 1.04x.  The order inside a class comes from LDS atomics: it changes which workgroup takes a tile, never
// a result.  Padding workgroups get 0xffffffff.
__global__ void __launch_bounds__(GHR_SCAN_BLOCK) k_tile_scan(int T, uint32_t* tile_count, uint32_t* small_cnt,
                                                              uint32_t* tile_start, uint32_t* R_out, uint32_t* slot_blk,
                                                              int nblk, uint32_t* R_mapped, uint32_t* tile_order)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // One workgroup, latency-bound: everything a thread needs is requested in ONE round trip (8 consecutive counts as two
    // 16-B loads; the gradient-slot counts of the K1 workgroups with them), scanned in registers, and the heaviest-first
    // order is built from the counts while they are still there (round 3: 21 -> see DESIGN.md 7; the first form re-read
    // tile_start for the order and walked the lists one element per load).
    __shared__ uint32_t s_part[GHR_SCAN_BLOCK];
    __shared__ uint32_t s_hist[8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t grid = xcd_grid((uint32_t)T);
    if (tile_order)
        for (uint32_t i = tid; i < 8u * 64u; i += GHR_SCAN_BLOCK) (&s_hist[0][0])[i] = 0u;
    // block-wide exclusive scan of one value per thread (s_part: 2 x 16 wave totals); returns the exclusive prefix, *total
    auto block_scan = [&](uint32_t sum, uint32_t* total) -> uint32_t {
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += v;
        }
        __syncthreads();  // s_part may still be read by a previous call
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        if (wave == 0) {
            uint32_t w = lane < GHR_SCAN_BLOCK / 64 ? s_part[lane] : 0u;
#pragma unroll
            for (int off = 1; off < GHR_SCAN_BLOCK / 64; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)w, off);
                if (lane >= off) w += v;
            }
            if (lane < GHR_SCAN_BLOCK / 64) s_part[64 + lane] = w;  // inclusive prefix of the wave totals
        }
        __syncthreads();
        *total = s_part[64 + GHR_SCAN_BLOCK / 64 - 1];
        return incl - sum + (wave ? s_part[64 + wave - 1] : 0u);
    };
    // ---- the K1 workgroups' gradient-slot counts (in place), requested first, used last
    const int per_b = (nblk + GHR_SCAN_BLOCK - 1) / GHR_SCAN_BLOCK;
    uint32_t sb[4] = {0u, 0u, 0u, 0u};
    const bool sb_regs = per_b <= 4;  // up to 4096 K1 workgroups (1 M Gaussians) in registers
    if (sb_regs)
#pragma unroll
        for (int i = 0; i < 4; i++) sb[i] = (i < per_b && tid * per_b + i < nblk) ? slot_blk[tid * per_b + i] : 0u;
    // ---- tile counts -> tile_start, rounds of 8192 tiles (one at 1080p)
    uint32_t carry = 0u;
    uint32_t c[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // (kept past the loop: the order below takes a single round's counts from here)
    for (uint32_t base = 0; base < (uint32_t)T; base += 8u * GHR_SCAN_BLOCK) {
        const uint32_t t0 = base + 8u * (uint32_t)tid;
        // two planes: [0] the small rects' instances (their positions are already handed out; the counts move to small_cnt:
        // k_scatter appends the big rects' instances behind them), [1] the big rects'.  Both planes go back to 0: [1] serves
        // as k_scatter's append cursors, and a workspace whose counters are at zero can be recycled without a zero-fill
        uint32_t* big_count = tile_count + T;
        if ((T & 3) == 0 && t0 + 8u <= (uint32_t)T) {  // (256-B aligned sub-allocations, T a multiple of 4: 16-B accesses)
            const uint4 a = *reinterpret_cast<const uint4*>(tile_count + t0), b = *reinterpret_cast<const uint4*>(tile_count + t0 + 4);
            const uint4 a2 = *reinterpret_cast<const uint4*>(big_count + t0), b2 = *reinterpret_cast<const uint4*>(big_count + t0 + 4);
            c[0] = a.x + a2.x; c[1] = a.y + a2.y; c[2] = a.z + a2.z; c[3] = a.w + a2.w;
            c[4] = b.x + b2.x; c[5] = b.y + b2.y; c[6] = b.z + b2.z; c[7] = b.w + b2.w;
            const uint4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4*>(small_cnt + t0) = a;
            *reinterpret_cast<uint4*>(small_cnt + t0 + 4) = b;
            *reinterpret_cast<uint4*>(tile_count + t0) = z;
            *reinterpret_cast<uint4*>(tile_count + t0 + 4) = z;
            *reinterpret_cast<uint4*>(big_count + t0) = z;
            *reinterpret_cast<uint4*>(big_count + t0 + 4) = z;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (t0 + i < (uint32_t)T) {
                    const uint32_t cs = tile_count[t0 + i];
                    c[i] = cs + big_count[t0 + i];
                    small_cnt[t0 + i] = cs;
                    tile_count[t0 + i] = 0u;
                    big_count[t0 + i] = 0u;
                } else {
                    c[i] = 0u;
                }
            }
        }
        uint32_t sum = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) sum += c[i];
        uint32_t total;
        uint32_t run = carry + block_scan(sum, &total);
        uint32_t st[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { st[i] = run; run += c[i]; }
        if (t0 + 8u <= (uint32_t)T) {   // tile_start[T + 1]: index t0 + 7 <= T - 1
            *reinterpret_cast<uint4*>(tile_start + t0) = uint4{st[0], st[1], st[2], st[3]};
            *reinterpret_cast<uint4*>(tile_start + t0 + 4) = uint4{st[4], st[5], st[6], st[7]};
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (t0 + i < (uint32_t)T) tile_start[t0 + i] = st[i];
        }
        if (tile_order) {   // count the tiles of every (XCD, class of n / 16); see k_tile_scan's header comment
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (t0 + i < (uint32_t)T) atomicAdd(&s_hist[((t0 + i) / GHR_XCD_RUN) & 7u][63u - min(c[i] >> 4, 63u)], 1u);
        }
        carry += total;
    }
    if (tid == 0) {
        tile_start[T] = carry;
        *R_out = carry;
        // the host's copy of the count, stored straight into its pinned (device-mapped) word: visible to the host
        // when the kernel retires, without a separate 4-byte copy command between this kernel and k_scatter
        if (R_mapped) *R_mapped = carry;
    }
    // ---- gradient-slot prefix of the K1 workgroups
    if (sb_regs) {
        uint32_t total;
        uint32_t run = block_scan(sb[0] + sb[1] + sb[2] + sb[3], &total);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < per_b && tid * per_b + i < nblk) slot_blk[tid * per_b + i] = run;
            run += sb[i];
        }
    } else {
        scan_1024(nblk, slot_blk, slot_blk, false, s_part);
    }
    // ---- heaviest-first order per XCD (counting sort; the second pass re-reads the counts from tile_start)
    if (tile_order) {
        __syncthreads();  // histogram complete; tile_start (and tile_start[T]) of this workgroup visible
        if (tid < 8 * 64) {  // wave x: exclusive scan of XCD x's 64 class counts -> first slot of each class
            const int x = tid >> 6;
            const uint32_t c = s_hist[x][lane];
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
                if (lane >= off) incl += v;
            }
            s_hist[x][lane] = incl - c;
            // the XCD's slots behind its last tile belong to padding workgroups
            const uint32_t total_x = (uint32_t)__shfl((int)incl, 63);
            for (uint32_t k = total_x + (uint32_t)lane; k < grid / 8u; k += 64u) tile_order[8u * k + (uint32_t)x] = 0xffffffffu;
        }
        __syncthreads();
        if ((uint32_t)T <= 8u * GHR_SCAN_BLOCK) {  // one round (1080p: 8160 tiles): the counts are still in registers
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t t = 8u * (uint32_t)tid + (uint32_t)i;
                if (t < (uint32_t)T) {
                    const uint32_t x = (t / GHR_XCD_RUN) & 7u;
                    tile_order[8u * atomicAdd(&s_hist[x][63u - min(c[i] >> 4, 63u)], 1u) + x] = t;
                }
            }
        } else {
            for (uint32_t t = tid; t < (uint32_t)T; t += GHR_SCAN_BLOCK) {
                const uint32_t x = (t / GHR_XCD_RUN) & 7u;
                tile_order[8u * atomicAdd(&s_hist[x][63u - min((tile_start[t + 1] - tile_start[t]) >> 4, 63u)], 1u) + x] = t;
            }
        }
    }
#endif
}

__global__ void __launch_bounds__(GHR_BLOCK) k_scatter(int P, int gx, rect4* rects,
                                                       const uint32_t* __restrict__ slot_blk,
                                                       const float* __restrict__ depths,
                                                       const uint32_t* __restrict__ tile_start, uint32_t* tile_count,
                                                       uint32_t T, const uint32_t* __restrict__ small_cnt,
                                                       const uint32_t* __restrict__ pos, uint64_t* keys, uint32_t cap)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // A wavefront serves 32 Gaussians: lane = 32*q + i handles the rect ordinals q, q+2 of Gaussian i -- and q+4, q+6 in a second
    // turn that only waves holding a rect of more than four tiles take (strand needles: 99.6 % of the rects have up to four;
    // 16 Gaussians x 4 lanes x two ordinals before: cfg3 18.5 -> 15.9 us, cfg2's blobs 25.8 -> 25.7, profiles/r05u).
    // Small rects (up to GHR_BIG_RECT = 8 tiles): the instance's place in its tile's list was handed out by K1's counting atomic
    // (count_tiles: pos[i][ordinal]) -- one gather of tile_start, one 8-B store, no atomic (round 5).
    const int lane = threadIdx.x & 63, q = lane >> 5;
    const int idx = (int)((blockIdx.x * (GHR_BLOCK / 64) + (threadIdx.x >> 6)) * 32) + (lane & 31);
    rect4 r = rect4{0u, 0u, 0u, 0u};
    uint32_t sb = 0u, pq0 = 0u, pq1 = 0u;
    float dep = 0.f;
    if (idx < P) {  // (all the loads together: one round trip; the positions of a culled / big rect are never used)
        r = rects[idx];
        sb = slot_blk[idx >> 8];
        dep = depths[idx];
        pq0 = pos[(size_t)GHR_BIG_RECT * idx + q];
        pq1 = pos[(size_t)GHR_BIG_RECT * idx + q + 2];
    }
    if (idx < P && q == 0) rects[idx].w = sb;  // gradient-slot base of the Gaussian's K1 workgroup (idempotent)
    const int x0 = r.x & 0xffffu, x1 = r.x >> 16, y0 = r.y & 0xffffu, y1 = r.y >> 16;
    const int w = x1 - x0, full = (x1 > x0 && y1 > y0) ? w * (y1 - y0) : 0;
    const bool big = full > GHR_BIG_RECT;  // expanded by the whole workgroup below
    const int area = big ? 0 : full;
    const uint64_t key = full ? (((uint64_t)__float_as_uint(dep) << 32) | (uint32_t)idx) : 0ull;
    static_assert(GHR_BIG_RECT == 8, "four ordinals per lane: q, q + 2, q + 4, q + 6");
    auto place = [&](int k0, int k1, uint32_t pa, uint32_t pb) {
        // ordinal k -> tile (row-major inside the rect, as count_tiles walks it); w >= 1 when area > 0
        const bool on0 = k0 < area, on1 = k1 < area;
        const int ky0 = on0 ? k0 / w : 0, kx0 = k0 - ky0 * w;
        const int ky1 = on1 ? k1 / w : 0, kx1 = k1 - ky1 * w;
        const int t0 = (y0 + ky0) * gx + x0 + kx0, t1 = (y0 + ky1) * gx + x0 + kx1;
        const uint32_t s0 = on0 ? tile_start[t0] : 0u, s1 = on1 ? tile_start[t1] : 0u;
        const uint32_t p0 = s0 + pa, p1 = s1 + pb;
        if (on0 && p0 < cap) keys[p0] = key;  // cap: see ghr_forward_stage2
        if (on1 && p1 < cap) keys[p1] = key;
    };
    place(q, q + 2, pq0, pq1);
    if (__builtin_amdgcn_ballot_w64(area > 4) != 0ull) {  // wave-uniform
        uint32_t pq2 = 0u, pq3 = 0u;
        if (area > 4) { pq2 = pos[(size_t)GHR_BIG_RECT * idx + q + 4]; pq3 = pos[(size_t)GHR_BIG_RECT * idx + q + 6]; }
        place(q + 4, q + 6, pq2, pq3);
    }
    // big rects: every Gaussian is held by the two lanes i, i+32 -- the first speaks for it; the workgroup's big
    // rects are expanded together, load-balanced (BigRects), two instances per thread and trip so that their returning
    // atomics are in flight together.  They go BEHIND the tile's small-rect instances: small_cnt[t] of those, then the
    // append cursor (the second plane of tile_count, at 0 on entry: k_tile_scan / the tile sort leave it there)
    uint32_t* cursor = tile_count + T;
    __shared__ BigRects s_big;
    const uint32_t total = big_rects_setup(s_big, (big && q == 0) ? (uint32_t)full : 0u, x0, y0, w, (uint32_t)key,
                                           (uint32_t)(key >> 32));
    for (uint32_t j = threadIdx.x; j < total; j += 2 * GHR_BLOCK) {
        const uint32_t j1 = j + GHR_BLOCK;
        const bool on1 = j1 < total;
        uint32_t o0, o1;
        const uint32_t t0 = big_rect_instance(s_big, j, gx, o0), t1 = big_rect_instance(s_big, on1 ? j1 : j, gx, o1);
        const uint32_t c0 = atomicAdd(&cursor[t0], 1u);
        uint32_t c1 = 0u;
        if (on1) c1 = atomicAdd(&cursor[t1], 1u);
        const uint32_t p0 = tile_start[t0] + small_cnt[t0] + c0, p1 = tile_start[t1] + small_cnt[t1] + c1;
        if (p0 < cap) keys[p0] = ((uint64_t)s_big.khi[o0] << 32) | s_big.klo[o0];
        if (on1 && p1 < cap) keys[p1] = ((uint64_t)s_big.khi[o1] << 32) | s_big.klo[o1];
    }
#endif
}

// Bitonic network in its "flip" form: every compare-exchange moves the smaller key to the lower index, so virtual
// +inf padding beyond n never moves and exchanges whose upper index is >= n are simply skipped (any n, no padding).
// Steps whose partner distance is < 64 only touch one aligned 64-key block per 32 pair indices, and thread t always
// handles the pair indices t, t + nthreads, ... (nthreads a multiple of 64) -- so with WAVE_LOCAL a wavefront owns its blocks for all those steps and
// they are separated by a wave-level fence instead of a workgroup barrier (45 of the 55 steps of a 1024-key sort).
#if defined(__HIP_DEVICE_COMPILE__)
#define GHR_SYNC() __syncthreads()
#define GHR_SYNC_WAVE()                                           \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)
// one wave's own global stores complete (and visible to its later loads) before it goes on: no barrier -- the other waves
// of the workgroup may have left already
#define GHR_SYNC_GLOBAL_WAVE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#else
#define GHR_SYNC() ((void)0)  // tests/hostsim runs the network with one "thread": steps are already ordered
#define GHR_SYNC_WAVE() ((void)0)
#endif
template <bool WAVE_LOCAL, typename KeyPtr>
GHR_HD void bitonic_any_n(KeyPtr k, uint32_t n, int tid, int nthreads)
{
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    const uint32_t half = np2 >> 1;
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        const uint32_t hs = size >> 1;
        const bool flip_local = WAVE_LOCAL && size <= 64;
        if (!flip_local) GHR_SYNC();  // keys written by other waves in the previous (local) steps
        for (uint32_t i = tid; i < half; i += nthreads) {  // flip: l <-> block_end - offset
            const uint32_t blk = i / hs, off = i - blk * hs;
            const uint32_t l = blk * size + off, u = blk * size + (size - 1 - off);
            if (u < n) {
                const uint64_t a = k[l], b = k[u];
                if (b < a) { k[l] = b; k[u] = a; }
            }
        }
        if (flip_local) GHR_SYNC_WAVE(); else GHR_SYNC();
        for (uint32_t j = hs >> 1; j >= 1; j >>= 1) {  // disperse: l <-> l + j
            for (uint32_t i = tid; i < half; i += nthreads) {
                const uint32_t l = 2 * j * (i / j) + (i % j), u = l + j;
                if (u < n) {
                    const uint64_t a = k[l], b = k[u];
                    if (b < a) { k[l] = b; k[u] = a; }
                }
            }
            if (WAVE_LOCAL && j <= 32) GHR_SYNC_WAVE(); else GHR_SYNC();  // j <= 32: this and the next step stay in-block
        }
    }
    GHR_SYNC();
}

// ---- the same kind of network, register-blocked (round 5) ---------------------------------------------------------------
// The per-step form above moves every key through LDS once per compare-exchange step (16 B of LDS traffic per key and step):
// at cfg3's list lengths that traffic -- not latency, not occupancy -- is what the tile sort waits for (profiles/r05u: removing
// dependent global round trips or adding tiles in flight changed nothing).  Here a thread owns R = 2^r keys between one LDS
// read and one LDS write and runs up to r consecutive steps of the network on them in registers: ceil(s / r) passes for a
// stage of s steps instead of s.  Still the flip form (every compare-exchange moves the smaller key to the lower index, so
// the virtual +inf padding beyond n never moves: positions >= n read as +inf and are never written).  Keys are unique
// (depth | Gaussian), so ANY correct network leaves the same list: results are bit-identical to the per-step form's.
GHR_HD void key_ce(uint64_t& lo, uint64_t& hi)  // smaller key to `lo`
{
    const uint64_t a = lo, b = hi;
    const bool sw = b < a;
    lo = sw ? b : a;
    hi = sw ? a : b;
}
// the same with the decision taken beforehand (the kernels decide all exchanges of a butterfly level first and move the keys
// afterwards: one 64-bit compare per exchange instead of the two the compiler makes of key_ce, and independent compares
// between a compare and the selects that wait for its mask)
GHR_HD void key_swap_if(bool sw, uint64_t& lo, uint64_t& hi)
{
    const uint32_t al = (uint32_t)lo, ah = (uint32_t)(lo >> 32), bl = (uint32_t)hi, bh = (uint32_t)(hi >> 32);
    const uint32_t ll = sw ? bl : al, lh = sw ? bh : ah, hl = sw ? al : bl, hh = sw ? ah : bh;
    lo = ((uint64_t)lh << 32) | ll;
    hi = ((uint64_t)hh << 32) | hl;
}
// LDS index of key i: one key of padding per 16 (a thread's contiguous chunk of R keys would otherwise put the lanes of a
// wave R * 8 bytes apart: an R-way bank conflict)
template <bool PAD>
GHR_HD uint32_t key_slot(uint32_t i) { return PAD ? i + (i >> 4) : i; }
#define GHR_KEY_INF 0xffffffffffffffffull

// FULL: all np2 slots exist and the ones past n hold +inf (the kernel's LDS array): no bounds tests at all.  Otherwise
// (tests/hostsim: a plain array of n keys) positions >= n read as +inf and are never written.
// T consecutive disperse steps (partner distances 2^(lj + T - 1) ... 2^lj) on groups of 2^T keys {base + (m << lj)}
template <int T, bool PAD, bool FULL, typename KeyPtr>
GHR_HD void bitonic_disperse_group(KeyPtr k, uint32_t n, uint32_t g, int lj)
{
    constexpr int G = 1 << T;
    const uint32_t base = ((g >> lj) << (lj + T)) | (g & ((1u << lj) - 1u));
    uint64_t v[G];
    uint32_t at[G];
#pragma unroll
    for (int m = 0; m < G; m++) {
        const uint32_t i = base + ((uint32_t)m << lj);
        at[m] = key_slot<PAD>(i);
        v[m] = (FULL || i < n) ? k[at[m]] : GHR_KEY_INF;
    }
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) {
        bool sw[G];
#pragma unroll
        for (int m = 0; m < G; m++)
            if (!(m & d)) sw[m] = v[m + d] < v[m];
#pragma unroll
        for (int m = 0; m < G; m++)
            if (!(m & d)) key_swap_if(sw[m], v[m], v[m + d]);
    }
#pragma unroll
    for (int m = 0; m < G; m++)
        if (FULL || base + ((uint32_t)m << lj) < n) k[at[m]] = v[m];
}
// the flip step of a stage of block size S = 2^ls and the T - 1 disperse steps behind it, on groups of 2^T keys:
// {B + a + m * jl} and their mirror images {B + S - 1 - a - m * jl}, jl = S >> T
template <int T, bool PAD, bool FULL, typename KeyPtr>
GHR_HD void bitonic_flip_group(KeyPtr k, uint32_t n, uint32_t g, int ls)
{
    constexpr int H = 1 << (T - 1);
    const int ljl = ls - T;
    const uint32_t B = (g >> ljl) << ls, a = g & ((1u << ljl) - 1u), top = B + (1u << ls) - 1u - a;
    uint64_t x[H], y[H];  // x[m] ascending in index, y[m] descending
    uint32_t ax[H], ay[H];
#pragma unroll
    for (int m = 0; m < H; m++) {
        const uint32_t il = B + a + ((uint32_t)m << ljl), iu = top - ((uint32_t)m << ljl);
        ax[m] = key_slot<PAD>(il);
        ay[m] = key_slot<PAD>(iu);
        x[m] = (FULL || il < n) ? k[ax[m]] : GHR_KEY_INF;
        y[m] = (FULL || iu < n) ? k[ay[m]] : GHR_KEY_INF;
    }
    {
        bool sw[H];
#pragma unroll
        for (int m = 0; m < H; m++) sw[m] = y[m] < x[m];
#pragma unroll
        for (int m = 0; m < H; m++) key_swap_if(sw[m], x[m], y[m]);
    }
#pragma unroll
    for (int d = H / 2; d >= 1; d >>= 1) {
        bool sx[H], sy[H];
#pragma unroll
        for (int m = 0; m < H; m++)
            if (!(m & d)) { sx[m] = x[m + d] < x[m]; sy[m] = y[m] < y[m + d]; }  // y[m + d] is the LOWER index
#pragma unroll
        for (int m = 0; m < H; m++)
            if (!(m & d)) { key_swap_if(sx[m], x[m], x[m + d]); key_swap_if(sy[m], y[m + d], y[m]); }
    }
#pragma unroll
    for (int m = 0; m < H; m++) {
        if (FULL || B + a + ((uint32_t)m << ljl) < n) k[ax[m]] = x[m];
        if (FULL || top - ((uint32_t)m << ljl) < n) k[ay[m]] = y[m];
    }
}
// all stages up to block size R = 2^r on a thread's contiguous chunk of R keys
template <int r, bool PAD, bool FULL, typename KeyPtr>
GHR_HD void bitonic_chunk_sort(KeyPtr k, uint32_t n, uint32_t c)
{
    constexpr int R = 1 << r;
    uint64_t v[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
        const uint32_t i = c * R + m;
        v[m] = (FULL || i < n) ? k[key_slot<PAD>(i)] : GHR_KEY_INF;
    }
#pragma unroll
    for (int S = 2; S <= R; S <<= 1) {
        bool sw[R];
#pragma unroll
        for (int m = 0; m < R; m++) {  // flip: m <-> block end - offset
            const int off = m & (S - 1);
            if (off < S / 2) sw[m] = v[(m - off) + (S - 1 - off)] < v[m];
        }
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int off = m & (S - 1);
            if (off < S / 2) key_swap_if(sw[m], v[m], v[(m - off) + (S - 1 - off)]);
        }
#pragma unroll
        for (int d = S / 4; d >= 1; d >>= 1) {
#pragma unroll
            for (int m = 0; m < R; m++)
                if (!(m & d)) sw[m] = v[m + d] < v[m];
#pragma unroll
            for (int m = 0; m < R; m++)
                if (!(m & d)) key_swap_if(sw[m], v[m], v[m + d]);
        }
    }
#pragma unroll
    for (int m = 0; m < R; m++) {
        const uint32_t i = c * R + m;
        if (FULL || i < n) k[key_slot<PAD>(i)] = v[m];
    }
}
// `t` disperse steps ending at distance 2^lj, every thread R keys per pass (R >> t groups of 2^t)
template <int r, bool PAD, bool FULL, typename KeyPtr>
GHR_HD void bitonic_disperse_pass(KeyPtr k, uint32_t n, uint32_t np2, int t, int lj, int tid, int nthreads)
{
    constexpr int R = 1 << r;
    for (uint32_t c = tid; c < np2 / R; c += nthreads) {
        if (t == r) {
            bitonic_disperse_group<r, PAD, FULL>(k, n, c, lj);
        } else {
            // (fewer steps left than a thread's keys allow: it takes R >> t smaller groups)
            for (uint32_t q = 0; q < (uint32_t)(R >> t); q++) {
                const uint32_t g = c * (uint32_t)(R >> t) + q;
                if (t == 1) bitonic_disperse_group<1, PAD, FULL>(k, n, g, lj);
                else if (t == 2) { if constexpr (r >= 2) bitonic_disperse_group<2, PAD, FULL>(k, n, g, lj); }
                else if (t == 3) { if constexpr (r >= 3) bitonic_disperse_group<3, PAD, FULL>(k, n, g, lj); }
            }
        }
    }
}
// the disperse steps of distances 2^(left-1) ... 1, r at a time (a synchronisation in front of every pass)
template <int r, bool WAVE, bool PAD, bool FULL, typename KeyPtr>
GHR_HD void bitonic_disperse_tail(KeyPtr k, uint32_t n, uint32_t np2, int left, int tid, int nthreads)
{
    while (left > 0) {
        const int t = left < r ? left : r;
        left -= t;
        if (WAVE) GHR_SYNC_WAVE(); else GHR_SYNC();
        bitonic_disperse_pass<r, PAD, FULL>(k, n, np2, t, left, tid, nthreads);
    }
}
// The whole sort.  r: log2 of the keys a thread owns (1..4); np2 = the power of two >= max(n, 2^r) keys are walked.  WAVE:
// the workgroup is ONE wave (passes are separated by a wave-level fence instead of a barrier).
template <int r, bool WAVE, bool PAD, bool FULL, typename KeyPtr>
GHR_HD void bitonic_blocked(KeyPtr k, uint32_t n, int tid, int nthreads)
{
    constexpr int R = 1 << r;
    uint32_t np2 = R;
    int lp = r;
    while (np2 < n) { np2 <<= 1; lp++; }
    for (uint32_t c = tid; c < np2 / R; c += nthreads) bitonic_chunk_sort<r, PAD, FULL>(k, n, c);
    for (int ls = r + 1; ls <= lp; ls++) {
        if (WAVE) GHR_SYNC_WAVE(); else GHR_SYNC();
        for (uint32_t g = tid; g < np2 / R; g += nthreads) bitonic_flip_group<r, PAD, FULL>(k, n, g, ls);
        bitonic_disperse_tail<r, WAVE, PAD, FULL>(k, n, np2, ls - r, tid, nthreads);  // distances (S >> r) / 2 ... 1
    }
    if (WAVE) GHR_SYNC_WAVE(); else GHR_SYNC();
}

// Both sort kernels also write, for every instance, where its gradient line will be: inst_line[instance] = position in
// the sorted lists (instances are numbered by rect4_slot; the lines lie in list order, ghr_device.h gather_inst_grads).
GHR_HD void sort_emit(uint32_t* point_list, uint32_t* inst_line, const rect4* rects, uint32_t pos, uint32_t id, int tx,
                      int ty, uint32_t cap)
{
    point_list[pos] = id;
    const uint32_t inst = rect4_slot(rects[id], tx, ty);
    if (inst < cap) inst_line[inst] = pos;
}

#if defined(__HIP_DEVICE_COMPILE__)
// NT threads (one wave: NT = 64, separated by wave-level fences; or NT / 64 cooperating waves, separated by barriers) sort a
// list of up to NT << r keys in LDS and write it out (k_tile_sort).  Every phase keeps a thread's (up to) 2^r memory
// operations in flight together: loads never sit under a branch (positions past n read the last key and drop it).
template <int r, int NT, int EMIT = 4>
__device__ __forceinline__ void tile_sort_group(uint64_t* g, uint32_t n, uint32_t s, uint64_t* s_keys, uint32_t* point_list,
                                                uint32_t* inst_line, const rect4* __restrict__ rects, int tx, int ty,
                                                uint32_t cap, int t_)
{
    constexpr int L = 1 << r;
    constexpr bool WAVE = NT == 64;
    uint64_t t[L];
#pragma unroll
    for (int q = 0; q < L; q++) {
        const uint32_t i = (uint32_t)t_ + (uint32_t)NT * q;
        t[q] = g[i < n ? i : n - 1u];
    }
#pragma unroll
    for (int q = 0; q < L; q++) {  // all NT << r slots: the ones past n hold +inf, the network runs without bounds tests
        const uint32_t i = (uint32_t)t_ + (uint32_t)NT * q;
        s_keys[key_slot<true>(i)] = i < n ? t[q] : GHR_KEY_INF;
    }
    if (WAVE) GHR_SYNC_WAVE(); else GHR_SYNC();
    if (n > 1) bitonic_blocked<r, WAVE, true, true>(s_keys, n, t_, NT);
    // (four entries at a time: the rect gather of more would set the kernel's register count)
    constexpr int E = L < EMIT ? L : EMIT;
#pragma unroll
    for (int h = 0; h < L; h += E) {
        uint64_t kq[E];
        rect4 rc[E];
#pragma unroll
        for (int q = 0; q < E; q++) {
            const uint32_t i = (uint32_t)t_ + (uint32_t)NT * (h + q);
            kq[q] = s_keys[key_slot<true>(i < n ? i : n - 1u)];
            rc[q] = rects[(uint32_t)kq[q]];
        }
#pragma unroll
        for (int q = 0; q < E; q++) {
            const uint32_t i = (uint32_t)t_ + (uint32_t)NT * (h + q);
            if (i < n) {
                g[i] = kq[q];
                point_list[s + i] = (uint32_t)kq[q];
                const uint32_t inst = rect4_slot(rc[q], tx, ty);
                if (inst < cap) inst_line[inst] = s + i;
            }
        }
    }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// keys of one aligned block of the list (nb <= C of them) global <-> LDS, C / 64 per lane in flight
template <int C>
__device__ __forceinline__ void sort_block_in(const uint64_t* g, uint32_t nb, uint64_t* s_keys, int lane)
{
    for (int h = 0; h < C / 64; h += 8) {  // (eight loads in flight at a time)
        uint64_t t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = (uint32_t)lane + 64u * (h + q);
            t[q] = g[i < nb ? i : nb - 1u];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = (uint32_t)lane + 64u * (h + q);
            s_keys[key_slot<true>(i)] = i < nb ? t[q] : GHR_KEY_INF;
        }
    }
    GHR_SYNC_WAVE();
}
template <int C>
__device__ __forceinline__ void sort_block_out(uint64_t* g, uint32_t nb, const uint64_t* s_keys, int lane)
{
    GHR_SYNC_WAVE();
#pragma unroll
    for (int q = 0; q < C / 64; q++) {
        const uint32_t i = (uint32_t)lane + 64u * q;
        if (i < nb) g[i] = s_keys[key_slot<true>(i)];
    }
}
// A list longer than the kernel's LDS holds (C keys), by one wave (rare: dense tiles of dense scenes have gone through
// k_tile_sort_big).  Same network: whatever stays inside an aligned block of C keys runs register-blocked in LDS, one load and
// one store of the block per stage; only the steps whose partner distance reaches across blocks go through global memory.
template <int CAP>
__device__ __forceinline__ void tile_sort_wave_long(uint64_t* g, uint32_t n, uint64_t* s_keys, int lane)
{
    constexpr uint32_t C = CAP;
    constexpr int RB = 3;  // 8 keys per lane and pass (two turns per pass at C = 1024: this rare path must not set the
                           // kernel's register count)
    int logc = 0;
    for (uint32_t j = C; j > 1; j >>= 1) logc++;
    for (uint32_t b0 = 0; b0 < n; b0 += C) {
        const uint32_t nb = min(C, n - b0);
        sort_block_in<CAP>(g + b0, nb, s_keys, lane);
        bitonic_blocked<RB, true, true, true>(s_keys, C, lane, 64);
        sort_block_out<CAP>(g + b0, nb, s_keys, lane);
    }
    uint32_t np2 = C;
    while (np2 < n) np2 <<= 1;
    auto global_step = [&](uint32_t size, uint32_t j) {  // j == 0: the flip of a stage of block size `size`; else disperse j
        GHR_SYNC_GLOBAL_WAVE();  // (orders the wave's global accesses of the previous step)
        const uint32_t hs = size >> 1;
        for (uint32_t i0 = lane; i0 < (np2 >> 1); i0 += 256u) {
            uint32_t l[4], u[4];
            uint64_t a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t i = i0 + 64u * q;
                if (j == 0u) { const uint32_t blk = i / hs, off = i - blk * hs; l[q] = blk * size + off; u[q] = blk * size + (size - 1u - off); }
                else { l[q] = 2u * j * (i / j) + (i % j); u[q] = l[q] + j; }
                const bool on = i < (np2 >> 1) && u[q] < n;
                if (!on) { l[q] = 0u; u[q] = 0u; }  // reads key 0 twice, exchanges nothing
                a[q] = g[l[q]];
                b[q] = g[u[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (b[q] < a[q]) { g[l[q]] = b[q]; g[u[q]] = a[q]; }
        }
    };
    for (uint32_t size = 2u * C; size <= np2; size <<= 1) {
        global_step(size, 0u);
        for (uint32_t j = size >> 2; j >= C; j >>= 1) global_step(size, j);
        GHR_SYNC_GLOBAL_WAVE();
        for (uint32_t b0 = 0; b0 < n; b0 += C) {
            const uint32_t nb = min(C, n - b0);
            sort_block_in<CAP>(g + b0, nb, s_keys, lane);
            bitonic_disperse_tail<RB, true, true, true>(s_keys, C, C, logc, lane, 64);
            sort_block_out<CAP>(g + b0, nb, s_keys, lane);
        }
    }
    GHR_SYNC_GLOBAL_WAVE();
}
#endif

// TWO waves per tile (round 5; four before, each moving every key through LDS once per compare-exchange step): lists are sorted
// by the register-blocked network above.  Short lists (<= GHR_SORT_SOLO keys: 7 of 10 tiles of cfg3) by wave 0 alone, no
// barrier at all (wave 1 leaves at once); longer ones by both waves (a wave per tile throughout was measured too: the
// longest lists then set the kernel's duration, profiles/r05u).  LDS holds CAP = 1024 keys
// (8.5 KiB); longer lists: tile_sort_wave_long.
#ifndef GHR_SORT_WAVES
#define GHR_SORT_WAVES 6  // per SIMD: 80 VGPRs, no spills: 26.1 us (8: 64 VGPRs + 8 spilled dwords 26.8; 5: 27.4; profiles/r05u)
#endif
#ifndef GHR_SORT_EMIT
#define GHR_SORT_EMIT 4   // entries whose rect gathers are in flight together when a list is written out
#endif
template <int CAP>
__global__ void __launch_bounds__(GHR_SORT_BLOCK) __attribute__((amdgpu_waves_per_eu(GHR_SORT_WAVES, 8))) k_tile_sort(uint32_t T, const uint32_t* __restrict__ tile_start,
                                                         uint64_t* keys, uint32_t* point_list, uint32_t cap,
                                                         uint32_t* tile_cursor, const rect4* __restrict__ rects,
                                                         uint32_t* inst_line, int gx,
                                                         const uint32_t* __restrict__ tile_order)
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(GHR_SORT_BLOCK == 128 && CAP == 1024, "two waves, 8 keys per thread at most");
    __shared__ uint64_t s_keys[CAP + CAP / 16 + 1];
    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : xcd_tile(blockIdx.x, T);  // heaviest first (k_tile_scan)
    if (tile >= T) return;  // grid padding
    const uint32_t s = min(tile_start[tile], cap);
    const uint32_t n = min(tile_start[tile + 1], cap) - s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (tile_cursor = tile_count[2][T]: [0] is back at 0 since k_tile_scan, [1] the big rects' append cursors / the DONE mark)
    // k_scatter is done with this tile's append cursor: leave it at 0, the state stage 2 expects on entry (stage 2 may be
    // replayed, e.g. after a too small speculative capacity) and a forward pass of its image workspace (a recycled one skips
    // its zero-fill).  Read by both waves, then -- behind a barrier both reach -- reset by thread 0.
    const bool sorted_already = tile_cursor[T + tile] == GHR_SORT_DONE;  // by k_tile_sort_big
    __syncthreads();
    if (tid == 0) tile_cursor[T + tile] = 0u;
    if (n == 0) return;
    if (sorted_already) return;
    uint64_t* g = keys + s;
    const int tx = tile % gx, ty = tile / gx;
    if (n <= GHR_SORT_SOLO) {
        if (wave == 1) return;
        if (n <= 128u) tile_sort_group<1, 64, GHR_SORT_EMIT>(g, n, s, s_keys, point_list, inst_line, rects, tx, ty, cap, lane);
        else tile_sort_group<2, 64, GHR_SORT_EMIT>(g, n, s, s_keys, point_list, inst_line, rects, tx, ty, cap, lane);
        return;
    }
    if (n <= 512u) tile_sort_group<2, 128, GHR_SORT_EMIT>(g, n, s, s_keys, point_list, inst_line, rects, tx, ty, cap, tid);
    else if (n <= 1024u) tile_sort_group<3, 128, GHR_SORT_EMIT>(g, n, s, s_keys, point_list, inst_line, rects, tx, ty, cap, tid);
    else {
        // Rare: a single tile with more instances than fit in LDS (and no k_tile_sort_big launch: sparse scene on average)
        if (wave == 1) return;
        tile_sort_wave_long<CAP>(g, n, s_keys, lane);
        for (uint32_t i = lane; i < n; i += 64)
            sort_emit(point_list, inst_line, rects, s + i, (uint32_t)g[i], tx, ty, cap);
    }
#endif
}

// Dense tiles (more than GHR_SORT_CAP instances; profiles/r02c: at the reference's strand-stage size the in-place global
// network of k_tile_sort set the kernel's duration, 0.74 ms).  A few 1024-thread workgroups walk the tiles; a dense one is
// sorted by the SAME network in blocks of GHR_SORT_BIG_CAP keys: everything of the network that stays inside an aligned
// block runs in LDS (one load / store of the block per pass), only the steps whose partner distance reaches across
// blocks touch global memory -- none up to 8192 keys, 1 of 105 steps at 16 384, 6 of 136 at 65 536.  The tile is then
// marked (tile_cursor = GHR_SORT_DONE) and k_tile_sort, which runs afterwards over all tiles, passes it by.  Launched
// only for dense scenes (host heuristic: average list length); a dense tile it does not see is still sorted by
// k_tile_sort's global path -- same network, same result.
template <typename KeyPtr>
GHR_HD void bitonic_disperse_from(KeyPtr k, uint32_t n, uint32_t count, uint32_t j0, int tid, int nthreads)
{
    // the disperse steps j0, j0/2, .. 1 of the network over `count` (a power of two) slots of which the first n hold keys
    for (uint32_t j = j0; j >= 1; j >>= 1) {
        for (uint32_t i = tid; i < (count >> 1); i += nthreads) {
            const uint32_t l = 2 * j * (i / j) + (i % j), u = l + j;
            if (u < n) {
                const uint64_t a = k[l], b = k[u];
                if (b < a) { k[l] = b; k[u] = a; }
            }
        }
        if (j <= 32) GHR_SYNC_WAVE(); else GHR_SYNC();
    }
    GHR_SYNC();
}

#if defined(__HIP_DEVICE_COMPILE__)
// The tiles blockIdx.x, blockIdx.x + gridDim.x, ... whose list length lies in (lo, hi], found in ONE round trip (thread t looks
// at the t-th of them) and listed in LDS; returns their number.  (Round 6: the dense-tile kernels walked their tiles one
// dependent pair of loads at a time -- 16 of them, ~1 us each, in front of the first key at 1080p.)  Needs T <= gridDim.x *
// GHR_SORT_WALK_MAX (the host sizes the grid).
// `tile_order` (k_tile_scan's heaviest-first order per XCD, may be NULL; `order_len` entries): the workgroup takes every
// (gridDim.x / 8)-th entry of its XCD's list instead -- every list of 1008 keys and more sits at the front of those lists (one
// weight class), so the dense tiles are dealt out evenly, where the strided walk over the raster order hands a workgroup
// anything between none and six of them (they cluster in the image: the hair).  Then gridDim.x must be a multiple of 8.
__device__ __forceinline__ uint32_t dense_tiles_of_workgroup(uint32_t T, const uint32_t* __restrict__ tile_start, uint32_t cap,
                                                             uint32_t lo, uint32_t hi, const uint32_t* __restrict__ tile_order,
                                                             uint32_t order_len, uint32_t* s_list, uint32_t* s_cnt)
{
    if (threadIdx.x == 0) *s_cnt = 0u;
    __syncthreads();
    uint32_t tile = 0xffffffffu;
    if (threadIdx.x < GHR_SORT_WALK_MAX) {
        if (tile_order) {
            const uint32_t i = 8u * ((blockIdx.x >> 3) + threadIdx.x * (gridDim.x >> 3)) + (blockIdx.x & 7u);
            if (i < order_len) tile = tile_order[i];
        } else {
            tile = blockIdx.x + threadIdx.x * gridDim.x;
        }
    }
    if (tile < T) {
        const uint32_t s = min(tile_start[tile], cap);
        const uint32_t n = min(tile_start[tile + 1], cap) - s;
        if (n > lo && n <= hi) s_list[atomicAdd(s_cnt, 1u)] = tile;
    }
    __syncthreads();
    return *s_cnt;
}
#endif

// Lists of GHR_SORT_CAP + 1 .. GHR_SORT_MID_CAP keys (round 6).  k_tile_sort_big gives such a tile 1024 threads, 64 KiB of LDS
// and one barrier per compare-exchange step -- two workgroups per CU, most of their threads without a pair to exchange; at
// 2 M strand Gaussians (BASELINE configs[4]) and in the strand stage most dense tiles are of this size and the kernel took
// 207 / 269 us per view.  Here: NT threads sort lists of (lo, 8 NT] keys on the register-blocked network of k_tile_sort (8 keys
// per thread between two LDS trips: 26 passes for 2048 keys where the per-step form takes 66 steps) in 8.5 NT bytes of LDS --
// 512 threads for up to 4096 keys (34.8 KiB of LDS, four workgroups per CU); the 256-thread instantiation for up to 2048 keys
// in front of it (GHR_SORT_MID_SPLIT) measured slower -- two launches whose critical paths add.
// The kernel is bound by a list's chain of passes (VALU 37 % busy, LDS 26 %: profiles/r06q): what pays is more lists in flight
// per CU -- 64 VGPRs for eight waves per SIMD (the list is written out two rect gathers at a time instead of four: no spill),
// and the grid sized so that every workgroup is resident: 181 -> 169 us (strand stage), 117 -> 101 us (cfg5) against three
// workgroups per CU at 66 VGPRs.
template <int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(GHR_SORT_MID_WAVES, 8))) k_tile_sort_mid(uint32_t T, const uint32_t* __restrict__ tile_start,
                                                                   uint64_t* keys, uint32_t* point_list, uint32_t cap,
                                                                   uint32_t* tile_cursor, const rect4* __restrict__ rects,
                                                                   uint32_t* inst_line, int gx,
                                                                   const uint32_t* __restrict__ tile_order, uint32_t order_len,
                                                                   uint32_t lo)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr uint32_t CAPM = (uint32_t)NT << 3;  // 8 keys per thread
    static_assert(CAPM <= GHR_SORT_MID_CAP && NT >= GHR_SORT_WALK_MAX, "k_tile_sort_mid: list length / tile walk");
    __shared__ uint64_t s_keys[CAPM + CAPM / 16 + 1];
    __shared__ uint32_t s_list[GHR_SORT_WALK_MAX], s_cnt;
    const uint32_t cnt = dense_tiles_of_workgroup(T, tile_start, cap, lo, CAPM, tile_order, order_len, s_list, &s_cnt);
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t tile = s_list[k];
        const uint32_t s = min(tile_start[tile], cap);
        const uint32_t n = min(tile_start[tile + 1], cap) - s;
        __syncthreads();  // (the previous tile's keys have left LDS)
        tile_sort_group<3, NT, GHR_SORT_MID_EMIT>(keys + s, n, s, s_keys, point_list, inst_line, rects, (int)(tile % (uint32_t)gx),
                               (int)(tile / (uint32_t)gx), cap, (int)threadIdx.x);
        if (threadIdx.x == 0) tile_cursor[T + tile] = GHR_SORT_DONE;
    }
#endif
}

// `min_n`: lists up to this long are somebody else's (GHR_SORT_CAP, or GHR_SORT_MID_CAP behind k_tile_sort_mid)
__global__ void __launch_bounds__(GHR_SORT_BIG_BLOCK) k_tile_sort_big(uint32_t T, const uint32_t* __restrict__ tile_start,
                                                                   uint64_t* keys, uint32_t* point_list, uint32_t cap,
                                                                   uint32_t* tile_cursor, const rect4* __restrict__ rects,
                                                                   uint32_t* inst_line, int gx, uint32_t min_n,
                                                                   const uint32_t* __restrict__ tile_order, uint32_t order_len)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint64_t s_keys[GHR_SORT_BIG_CAP];
    __shared__ uint32_t s_list[GHR_SORT_WALK_MAX], s_cnt;
    const int tid = threadIdx.x;
    constexpr uint32_t B = GHR_SORT_BIG_CAP;
    const uint32_t n_dense = dense_tiles_of_workgroup(T, tile_start, cap, min_n, 0xffffffffu, tile_order, order_len, s_list, &s_cnt);
    for (uint32_t kk = 0; kk < n_dense; kk++) {
        const uint32_t tile = s_list[kk];
        const uint32_t s = min(tile_start[tile], cap);
        const uint32_t n = min(tile_start[tile + 1], cap) - s;
        uint64_t* g = keys + s;
        const uint32_t nb = (n + B - 1) / B;
        // every block on its own: the network's steps up to size B
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t cnt = min(B, n - b * B);
            __syncthreads();
            for (uint32_t i = tid; i < cnt; i += GHR_SORT_BIG_BLOCK) s_keys[i] = g[b * B + i];
            __syncthreads();
            bitonic_any_n<true>(s_keys, cnt, tid, GHR_SORT_BIG_BLOCK);
            for (uint32_t i = tid; i < cnt; i += GHR_SORT_BIG_BLOCK) g[b * B + i] = s_keys[i];
        }
        if (nb > 1) {
            uint32_t np2 = 1;
            while (np2 < n) np2 <<= 1;
            const uint32_t half = np2 >> 1;
            for (uint32_t size = 2 * B; size <= np2; size <<= 1) {
                const uint32_t hs = size >> 1;
                __syncthreads();  // (one workgroup, one CU: the barrier orders its global accesses, as in k_tile_sort)
                for (uint32_t i = tid; i < half; i += GHR_SORT_BIG_BLOCK) {  // flip, across blocks
                    const uint32_t blk = i / hs, off = i - blk * hs;
                    const uint32_t l = blk * size + off, u = blk * size + (size - 1 - off);
                    if (u < n) {
                        const uint64_t a = g[l], bb = g[u];
                        if (bb < a) { g[l] = bb; g[u] = a; }
                    }
                }
                __syncthreads();
                for (uint32_t j = hs >> 1; j >= B; j >>= 1) {  // disperse steps that still reach across blocks
                    for (uint32_t i = tid; i < half; i += GHR_SORT_BIG_BLOCK) {
                        const uint32_t l = 2 * j * (i / j) + (i % j), u = l + j;
                        if (u < n) {
                            const uint64_t a = g[l], bb = g[u];
                            if (bb < a) { g[l] = bb; g[u] = a; }
                        }
                    }
                    __syncthreads();
                }
                for (uint32_t b = 0; b < nb; b++) {  // the rest of this size's disperse steps: inside the blocks
                    const uint32_t cnt = min(B, n - b * B);
                    for (uint32_t i = tid; i < cnt; i += GHR_SORT_BIG_BLOCK) s_keys[i] = g[b * B + i];
                    __syncthreads();
                    bitonic_disperse_from(s_keys, cnt, B, B >> 1, tid, GHR_SORT_BIG_BLOCK);
                    for (uint32_t i = tid; i < cnt; i += GHR_SORT_BIG_BLOCK) g[b * B + i] = s_keys[i];
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += GHR_SORT_BIG_BLOCK)
            sort_emit(point_list, inst_line, rects, s + i, (uint32_t)g[i], (int)(tile % (uint32_t)gx), (int)(tile / (uint32_t)gx), cap);
        if (tid == 0) tile_cursor[T + tile] = GHR_SORT_DONE;
    }
#endif
}

}  // namespace ghr
