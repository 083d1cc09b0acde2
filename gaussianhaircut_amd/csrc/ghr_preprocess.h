// ghr_preprocess.h -- K1 (per-Gaussian preprocess) fused with the per-tile instance count; markVisible.
// Follows R:cuda_rasterizer/forward.cu:155-282 (preprocessCUDA), :74-152 (computeCov2D / computeCov3D) and
// R:cuda_rasterizer/auxiliary.h:139-164 (in_frustum).  One thread per Gaussian, HBM-bound:
//   reads  xyz 12 + conic 12 (or scales/rot/cov3D) + opacity 4 + colors 40 B
//   writes one 64-B packed record {x, y, a, b | c, opacity, f0, f1 | f2..f5 | f6..f9}, depth 4, rect 8, radii 4 B.
#pragma once
#include "ghr_device.h"

namespace ghr {

struct PreArgs {
    int P, W, H, gx, gy;
    const float* means3D;
    const float* colors;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* conic_precomp;
    const float* view;
    const float* proj;
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y;
    // outputs
    f4* rec;
    float* depths;
    rect4* rects;
    float* cov3D;  // mode B only
    int* radii;
    uint32_t* tile_count;  // [2][T] per-tile instance counts: of the small rects (positions handed out), of the big ones
    uint32_t* slot_blk;    // [ceil(P/256)] gradient slots used by each K1 workgroup (scanned by k_tile_scan)
    uint32_t* pos;         // [P][GHR_BIG_RECT] position of each small-rect instance inside its tile's list (see count_tiles)
};

// forward.cu:118-152.  The quaternion is used as given (normalisation is commented out at :127).
GHR_HD void cov3d_from_scale_rot(const float* s3, float mod, const float* q4, float* cov3D)
{
    m3 S = {};
    S.m[0][0] = mod * s3[0];
    S.m[1][1] = mod * s3[1];
    S.m[2][2] = mod * s3[2];
    const float r = q4[0], x = q4[1], y = q4[2], z = q4[3];
    m3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    const m3 M = mul(S, R);
    const m3 Sigma = mul(transpose(M), M);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

// Shared by forward.cu:74-113 and backward.cu:166-194: clamped view-space mean, J, W, T = W*J, Vrk.
struct Cov2DCtx {
    float tx, ty, tz, txtz, tytz, limx, limy;
    m3 J, Wm, T, Vrk;
};
GHR_HD void cov2d_setup(Cov2DCtx& c, float mx, float my, float mz, float fx, float fy,
                                            float tan_fovx, float tan_fovy, const float* cov3D,
                                            const float* view)
{
    xform4x3(view, mx, my, mz, c.tx, c.ty, c.tz);
    c.limx = 1.3f * tan_fovx;
    c.limy = 1.3f * tan_fovy;
    c.txtz = c.tx / c.tz;
    c.tytz = c.ty / c.tz;
    c.tx = fminf(c.limx, fmaxf(-c.limx, c.txtz)) * c.tz;
    c.ty = fminf(c.limy, fmaxf(-c.limy, c.tytz)) * c.tz;
    c.J = {};
    c.J.m[0][0] = fx / c.tz; c.J.m[0][2] = -(fx * c.tx) / (c.tz * c.tz);
    c.J.m[1][1] = fy / c.tz; c.J.m[1][2] = -(fy * c.ty) / (c.tz * c.tz);
    c.Wm.m[0][0] = view[0]; c.Wm.m[0][1] = view[4]; c.Wm.m[0][2] = view[8];
    c.Wm.m[1][0] = view[1]; c.Wm.m[1][1] = view[5]; c.Wm.m[1][2] = view[9];
    c.Wm.m[2][0] = view[2]; c.Wm.m[2][1] = view[6]; c.Wm.m[2][2] = view[10];
    c.T = mul(c.Wm, c.J);
    c.Vrk.m[0][0] = cov3D[0]; c.Vrk.m[0][1] = cov3D[1]; c.Vrk.m[0][2] = cov3D[2];
    c.Vrk.m[1][0] = cov3D[1]; c.Vrk.m[1][1] = cov3D[3]; c.Vrk.m[1][2] = cov3D[4];
    c.Vrk.m[2][0] = cov3D[2]; c.Vrk.m[2][1] = cov3D[4]; c.Vrk.m[2][2] = cov3D[5];
}
// cov = T^t * Vrk^t * T, +0.3 on the diagonal (forward.cu:106-112); returns (xx, xy, yy).
GHR_HD void cov2d_eval(const Cov2DCtx& c, float& a, float& b, float& d)
{
    const m3 A = mul(transpose(c.T), transpose(c.Vrk));
    const m3 Cm = mul(A, c.T);
    a = Cm.m[0][0] + 0.3f;
    b = Cm.m[0][1];
    d = Cm.m[1][1] + 0.3f;
}

// Per-Gaussian body of K1.  Returns false when the Gaussian is culled (radii/rect already zeroed).
GHR_HD bool preprocess_one(const PreArgs& a, int idx, int& x0, int& y0, int& x1, int& y1)
{
    a.radii[idx] = 0;  // forward.cu:190-191
    a.rects[idx] = rect4{0u, 0u, 0u, 0u};

    const float mx = a.means3D[3 * idx], my = a.means3D[3 * idx + 1], mz = a.means3D[3 * idx + 2];
    float vx, vy, vz;
    xform4x3(a.view, mx, my, mz, vx, vy, vz);
    if (vz <= 0.2f) return false;  // auxiliary.h:154 -- silent cull, never trap

    // forward.cu:203-205
    const float* pm = a.proj;
    const float hx = pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12];
    const float hy = pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13];
    const float hw = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float projx = hx * p_w, projy = hy * p_w;

    float cva, cvb, cvc, cx, cy, cz, det;
    if (a.conic_precomp == nullptr) {  // mode B, forward.cu:214-239
        float c3[6];
        if (a.cov3D_precomp != nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = a.cov3D_precomp[6 * idx + i];
        } else {
            const float s3[3] = {a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]};
            const float q4[4] = {a.rotations[4 * idx], a.rotations[4 * idx + 1], a.rotations[4 * idx + 2],
                                 a.rotations[4 * idx + 3]};
            cov3d_from_scale_rot(s3, a.scale_modifier, q4, c3);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) a.cov3D[6 * idx + i] = c3[i];
        Cov2DCtx c;
        cov2d_setup(c, mx, my, mz, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, a.view);
        cov2d_eval(c, cva, cvb, cvc);
        det = (cva * cvc - cvb * cvb);
        if (det == 0.0f) return false;
        const float det_inv = 1.f / det;
        cx = cvc * det_inv;
        cy = -cvb * det_inv;
        cz = cva * det_inv;
    } else {  // mode A, forward.cu:240-248
        cx = a.conic_precomp[3 * idx];
        cy = a.conic_precomp[3 * idx + 1];
        cz = a.conic_precomp[3 * idx + 2];
        const float det_inv = (cx * cz - cy * cy);
        if (det_inv == 0.0f) return false;
        det = 1.f / det_inv;
        cva = cz * det;
        cvb = -cy * det;
        cvc = cx * det;
    }
    // forward.cu:254-262
    const float mid = 0.5f * (cva + cvc);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda1 = mid + sq, lambda2 = mid - sq;
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    const float pixx = ndc2pix(projx, a.W), pixy = ndc2pix(projy, a.H);
    tile_rect(pixx, pixy, (int)my_radius, a.gx, a.gy, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) == 0) return false;

    // forward.cu:275-281, packed: one 64-B line per Gaussian, read back as 4 x b128 by the render kernels.
    const float* col = a.colors + (size_t)GHR_C * idx;
    f4* r = a.rec + 4 * (size_t)idx;
    r[0] = f4{pixx, pixy, cx, cy};
    r[1] = f4{cz, a.opacities[idx], col[0], col[1]};
    r[2] = f4{col[2], col[3], col[4], col[5]};
    r[3] = f4{col[6], col[7], col[8], col[9]};
    a.depths[idx] = vz;
    a.radii[idx] = (int)my_radius;
    a.rects[idx] = make_rect4(x0, y0, x1, y1, 0u);  // the caller fills in the gradient-slot base
    return true;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Counts every tile of every thread's rect (empty rect: x1 == x0) -- and, since round 5, HANDS OUT the instance's place in
// its tile's list at the same time: the counting atomic of a small rect (up to GHR_BIG_RECT tiles) returns the count before
// it, which IS a unique position among the tile's small-rect instances (the list is sorted afterwards: any unique
// position will do), stored in pos_row[ordinal].  k_scatter then places those instances without an atomic of its own
// (it used to repeat the 1.3 M atomics, returning ones, on the append cursors: throughput-bound, profiles/r05j).
// Small rects: run-aggregated (wave_inc_issue / _result) -- lanes walk their k-th tile in lockstep, so neighbouring
// Gaussians with equal rects share one atomic; all of a lane's up to 8 atomics are in flight before the first result is used.
// Big rects: counted into the SECOND plane (tile_count + T), load-balanced over the workgroup (BigRects); k_scatter appends
// them behind the small ones with cursors of their own.  EVERY thread of the 256-thread workgroup calls both halves; `s` is
// workgroup scratch nobody else touches between the barrier in front of count_tiles_finish and the end of the kernel.
// (in two halves, so that a kernel can do its stores between sending the atomics and needing their results)
struct TileCountPending {
    uint32_t pb[GHR_BIG_RECT];
    int pl[GHR_BIG_RECT];
    int max_area, area;
};
__device__ __forceinline__ void count_tiles_issue(uint32_t* tile_count, int gx, int x0, int y0, int x1, int y1, TileCountPending& c)
{
    const int w = x1 - x0, full = w * (y1 - y0);
    const bool big = full > GHR_BIG_RECT;
    c.area = big ? 0 : full;
    int max_area = c.area;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) max_area = max(max_area, __shfl_xor(max_area, off));
    c.max_area = max_area;
    int kx = 0, t = y0 * gx + x0;  // row-major walk over the rect without a division per step
#pragma unroll
    for (int k = 0; k < GHR_BIG_RECT; k++) {
        c.pb[k] = 0u; c.pl[k] = 0;
        if (k < max_area) {  // wave-uniform
            c.pb[k] = wave_inc_issue(tile_count, (uint32_t)t, k < c.area, c.pl[k]);
            if (++kx == w) { kx = 0; t += gx - w + 1; } else t++;
        }
    }
}
__device__ __forceinline__ void count_tiles_finish(uint32_t* tile_count, uint32_t T, int gx, int x0, int y0, int x1, int y1,
                                                   BigRects& s, uint32_t* pos_row, const TileCountPending& c)
{
    uint32_t pv[GHR_BIG_RECT];
#pragma unroll
    for (int k = 0; k < GHR_BIG_RECT; k++) pv[k] = k < c.max_area ? wave_inc_result(c.pb[k], c.pl[k]) : 0u;
    // (two 16-B stores per Gaussian -- the row is 32 B, 32-B aligned -- instead of one 4-B store per ordinal; the slots past
    // the rect's area hold nothing anybody reads)
    static_assert(GHR_BIG_RECT == 8, "pos rows are two uint4");
    if (c.area > 0) reinterpret_cast<uint4*>(pos_row)[0] = uint4{pv[0], pv[1], pv[2], pv[3]};
    if (c.area > 4) reinterpret_cast<uint4*>(pos_row)[1] = uint4{pv[4], pv[5], pv[6], pv[7]};
    const int w = x1 - x0, full = w * (y1 - y0);
    const uint32_t total = big_rects_setup(s, full > GHR_BIG_RECT ? (uint32_t)full : 0u, x0, y0, w);
    for (uint32_t j = threadIdx.x; j < total; j += GHR_BLOCK) {
        uint32_t owner;
        atomicAdd(&tile_count[T + big_rect_instance(s, j, gx, owner)], 1u);
    }
}
#endif

__global__ void __launch_bounds__(GHR_BLOCK) k_preprocess(PreArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int idx = blockIdx.x * GHR_BLOCK + threadIdx.x;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    __shared__ uint32_t s_scan[4];
    __shared__ BigRects s_big;
    const bool ok = idx < a.P && preprocess_one(a, idx, x0, y0, x1, y1);
    // Per-tile instance counts (replaces the tiles_touched scan + duplicateWithKeys offsets,
    // rasterizer_impl.cu:281,88): tile lists are laid out tile-major, so counts are all binning needs.
    TileCountPending tc;
    count_tiles_issue(a.tile_count, a.gx, x0, y0, x1, y1, tc);
    uint32_t blk_total;
    const uint32_t base = block_excl_scan_256(ok ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u, s_scan, &blk_total);
    if (ok) a.rects[idx].z = base;
    if (threadIdx.x == 0) a.slot_blk[blockIdx.x] = blk_total;
    count_tiles_finish(a.tile_count, (uint32_t)(a.gx * a.gy), a.gx, x0, y0, x1, y1, s_big,
                       a.pos + (size_t)GHR_BIG_RECT * (idx < a.P ? idx : 0), tc);
#endif
}

// rasterizer_impl.cu:54-66 (checkFrustum): only the near test is live (auxiliary.h:154).
__global__ void __launch_bounds__(GHR_BLOCK) k_mark_visible(int P, const float* __restrict__ means3D,
                                                            const float* __restrict__ view, uint8_t* present)
{
    const int idx = blockIdx.x * GHR_BLOCK + threadIdx.x;
    if (idx >= P) return;
    float vx, vy, vz;
    xform4x3(view, means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2], vx, vy, vz);
    present[idx] = !(vz <= 0.2f);
}

}  // namespace ghr
