// ghr_loss.h -- fused loss of the stage-1 step: masked L1 + (1 - SSIM) + mask L1 + orientation, forward and backward.
//
// Reference: src/train_gaussians.py:126-140 with src/utils/loss_utils.py:19-26 (l1_loss) and :91-121 (ssim: 11x11
// Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over the map):
//     Ll1   = mean(|image - gt| * m)                       m = gt_mask[1:] (foreground), broadcast over RGB
//     Lssim = 1 - mean(ssim_map(image * m, gt * m))
//     Lmask = mean(|mask - gt_mask|)
//     Lorient = or_loss(orient_angle, gt_angle, orient_conf, weight = gt_orient_conf, mask = gt_mask[:1])
//               (loss_utils.py:31-47) with orient_angle derived from the rendered 2D strand direction exactly as
//               src/gaussian_renderer/__init__.py:100-105 (normalize, mirror, clamp, acos / pi); NaN -> 0
//               (train_gaussians.py:134)
//     loss  = w_l1 * Ll1 + w_ssim * Lssim + w_mask * Lmask + w_orient * Lorient      (run.sh:112-115: w_orient = 0.1)
// PyTorch runs this as 10 MIOpen depthwise convolutions + ~40 elementwise kernels per step (measured 10.2 ms at
// 1080p on MI355X = 69 % of the step once projection was fused).  Here: one forward kernel (separable 11-tap window
// staged through LDS, 32x16 pixel tiles with a 5-pixel halo) that also emits the three per-pixel partial derivatives
// of the SSIM map, and one backward kernel that convolves those maps back (same window, adjoint of a symmetric
// zero-padded convolution) and adds the L1 terms.  HBM-bound: ~25 B/pixel/channel forward, ~30 B backward.
#pragma once
#include "ghr_device.h"

namespace ghr {

#define GHR_SSIM_R 5
#define GHR_SSIM_T 16
#define GHR_SSIM_E (GHR_SSIM_T + 2 * GHR_SSIM_R)  // 26
#define GHR_LOSS_SLOTS 256
#define GHR_LOSS_TERMS 5  // partial sums per slot: |image-gt|*m, ssim_map, |mask-gt_mask|, orient num, orient den

__device__ __constant__ float c_ssim_w[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                             2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                             3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};

// SSIM value and its partial derivatives w.r.t. (mu1, E[x^2], E[xy]) of the first image; mu2 / E[y^2] are constants.
GHR_HD float ssim_point(float mu1, float mu2, float e11, float e22, float e12, float& dm_dmu1, float& dm_de11,
                        float& dm_de12)
{
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
    // v_rcp_f32 (1 ulp): a correctly rounded division is ~10 VALU, and this runs per pixel and channel
    const float iB1 = fast_rcp(B1), iB2 = fast_rcp(B2);
    const float f1 = A1 * iB1, f2 = A2 * iB2;
    dm_de11 = -f1 * f2 * iB2;                 // via sigma1_sq
    dm_de12 = 2.f * f1 * iB2;                 // via sigma12
    dm_dmu1 = f2 * (2.f * mu2 * B1 - 2.f * mu1 * A1) * iB1 * iB1   // via A1/B1
              + dm_de11 * (-2.f * mu1)                              // sigma1_sq = e11 - mu1^2
              + dm_de12 * (-mu2);                                   // sigma12 = e12 - mu1 mu2
    return f1 * f2;
}

struct LossArgs {
    int W, H;
    const float* image;     // [3,H,W] rendered
    const float* mask;      // [2,H,W] rendered (hair label, foreground)
    const float* dir2d;     // [2,H,W] rendered 2D strand direction (x, y), or NULL (no orientation term)
    const float* oconf;     // [1,H,W] rendered orientation confidence
    const float* gt_image;  // [3,H,W]
    const float* gt_mask;   // [2,H,W]; channel 1 masks the colour terms, channel 0 the orientation term
    const float* gt_angle;  // [1,H,W] orientation angle / pi in [0,1)
    const float* gt_oconf;  // [1,H,W] per-pixel weight of the orientation term
    int mask_colours;       // 0: colour terms on the whole image (strand stage)
    float* maps;            // [3 kinds][3 ch][H*W]: dm/dmu1, dm/dE[x^2], dm/dE[xy]
    float* sums;            // [GHR_LOSS_SLOTS][GHR_LOSS_TERMS] partial sums, zeroed by the caller; block b adds into
                            // slot b % SLOTS (one hot address would serialise ~73k atomics)
    const float* gt_stats;  // [2][3][H*W] window moments of the masked ground truth (mu2, E[y^2]) or NULL
    float* stats_out;       // k_loss_gt_stats: where those moments go
};

// Orientation term of ONE pixel (gaussian_renderer/__init__.py:100-105 + loss_utils.py:31-47), value and the partial
// derivatives w.r.t. the rendered direction (d0, d1) and confidence.  `l` excludes the weight gt_oconf.
struct OrientPix { float l, dl_dd0, dl_dd1, dl_dconf; };
GHR_HD OrientPix orient_pixel(float d0, float d1, float conf, float gt_angle, float m)
{
    const float PI = 3.14159265358979323846f;
    const float INV_PI = 0.31830988618379067154f;
    const float nrm = fast_sqrt(d0 * d0 + d1 * d1);
    const float den = fmaxf(nrm, 1e-12f);                 // F.normalize(dim=0), eps = 1e-12
    const float iden = fast_rcp(den);
    const float u0 = d0 * iden, u1 = d1 * iden;
    const float mirror = u0 < 0.f ? -1.f : 1.f;
    const float lo = -1.f + 1e-3f, hi = 1.f - 1e-3f;
    const float uc = fminf(hi, fmaxf(lo, u1));
    const float c = uc * mirror;
    const float angle = acosf(c) * INV_PI;
    const float diff = angle - gt_angle;
    const float a0 = fabsf(diff), a1 = fabsf(diff - 1.f), a2 = fabsf(diff + 1.f);
    float lmin = a0, arg = diff;
    if (a1 < lmin) { lmin = a1; arg = diff - 1.f; }
    if (a2 < lmin) { lmin = a2; arg = diff + 1.f; }
    OrientPix o;
    o.l = (lmin * PI * conf - logf(conf + 1e-7f)) * m;
    o.dl_dconf = (lmin * PI - fast_rcp(conf + 1e-7f)) * m;
    const float sgn = arg > 0.f ? 1.f : (arg < 0.f ? -1.f : 0.f);
    const float dl_dangle = PI * conf * m * sgn;
    const float dl_dc = -dl_dangle * INV_PI * fast_rcp(fast_sqrt(1.f - c * c));
    const float dl_du1 = (u1 >= lo && u1 <= hi) ? dl_dc * mirror : 0.f;   // clamp passes gradient inside the range
    if (nrm > 1e-12f) {  // u = d / |d|:  du1/dd0 = -d0 d1 / |d|^3,  du1/dd1 = d0^2 / |d|^3
        const float i3 = iden * iden * iden;  // nrm > eps here, so den == nrm
        o.dl_dd0 = dl_du1 * (-d0 * d1 * i3);
        o.dl_dd1 = dl_du1 * (d0 * d0 * i3);
    } else {             // clamped denominator: u = d / eps
        o.dl_dd0 = 0.f;
        o.dl_dd1 = dl_du1 * 1e12f;
    }
    return o;
}

// Tile geometry of both kernels: a 256-thread block produces a 32 x 16 pixel tile of one colour channel from a
// (32 + 10) x (16 + 10) input window.  Horizontal pass: one thread = 4 adjacent outputs of one row from 16 inputs read
// as four ds_read_b128 (rows are 16-B aligned); vertical pass: one thread = 2 vertically adjacent outputs of one
// column (12 rows x maps of conflict-free ds_read_b32).  One barrier per phase and ONE for all partial sums.
#define GHR_L_TW 32
#define GHR_L_TH 16
#define GHR_L_EW (GHR_L_TW + 2 * GHR_SSIM_R)  // 42
#define GHR_L_EH (GHR_L_TH + 2 * GHR_SSIM_R)  // 26
#define GHR_L_XS 44  // row stride (floats) of the input window: 16-B aligned rows, columns 42, 43 are zero padding
#define GHR_L_HS 36  // row stride of the horizontally filtered maps: 16-B aligned, (4 r + c) mod 32 banks

#if defined(__HIP_DEVICE_COMPILE__)
// Sums NV per-thread values over the 256 threads of the block with one barrier; valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum_n(float* v, float (*s_red)[8])
{
#pragma unroll
    for (int k = 0; k < NV; k++)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; k++) s_red[tid >> 6][k] = v[k];
    __syncthreads();
    if (tid == 0)
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k];
}
#endif

// Forward body in three variants (MODE):
//   0  all five window moments (x, y, x^2, y^2, xy) of the masked render x and ground truth y
//   1  the ground-truth moments (mu2 = w*y, E[y^2] = w*y^2) come from a.gt_stats, computed once per camera by MODE 2
//      (ground truth and mask are constants of a training view): three moments instead of five in both passes and in
//      LDS -- 40 % of the window arithmetic
//   2  computes only those two moments into a.stats_out (no loss terms)
// The arithmetic of each moment is identical in all variants, so MODE 1 + MODE 2 reproduce MODE 0 bit for bit.
template <int MODE>
__device__ __forceinline__ void loss_fwd_body(const LossArgs& a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool HAVE_X = MODE != 2;              // x, x^2, xy
    constexpr bool HAVE_Y = MODE != 1;              // y, y^2
    constexpr int NM = MODE == 0 ? 5 : (MODE == 1 ? 3 : 2);
    // plane index of each moment (x, y, xx, yy, xy) in s_h
    constexpr int PX = 0, PY = MODE == 0 ? 1 : 0, PXX = MODE == 0 ? 2 : 1, PYY = MODE == 0 ? 3 : 1, PXY = MODE == 0 ? 4 : 2;
    __shared__ __attribute__((aligned(16))) float s_x[GHR_L_EH][GHR_L_XS], s_y[GHR_L_EH][GHR_L_XS];
    __shared__ __attribute__((aligned(16))) float s_h[NM][GHR_L_EH][GHR_L_HS];
    __shared__ float s_red[4][8];
    const int ch = blockIdx.z;
    const int W = a.W, H = a.H;
    const size_t N = (size_t)W * H;
    const int bx = blockIdx.x * GHR_L_TW, by = blockIdx.y * GHR_L_TH;
    const int tid = threadIdx.x;
    const float* img = a.image + ch * N;
    const float* gt = a.gt_image + ch * N;
    const float* m = a.gt_mask + N;  // gt_mask[1]

    {   // window load: all (up to 15) global loads of a thread are issued before the first LDS store -- the phase is
        // latency-bound, one dependent round trip per loop iteration cost ~2/3 of the kernel
        constexpr int NIT = (GHR_L_EH * GHR_L_XS + 255) / 256;  // 5
        float vi[NIT], vg[NIT], vm[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + 256 * it;
            const int ly = i / GHR_L_XS, lx = i - ly * GHR_L_XS;
            const int gx = bx + lx - GHR_SSIM_R, gy = by + ly - GHR_SSIM_R;
            const bool in = i < GHR_L_EH * GHR_L_XS && lx < GHR_L_EW && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t p = in ? (size_t)gy * W + gx : 0;
            vm[it] = in ? (a.mask_colours ? m[p] : 1.0f) : 0.f;
            vi[it] = (HAVE_X && in) ? img[p] : 0.f;
            vg[it] = in ? gt[p] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + 256 * it;
            if (i < GHR_L_EH * GHR_L_XS) {
                const int ly = i / GHR_L_XS, lx = i - ly * GHR_L_XS;
                if (HAVE_X) s_x[ly][lx] = vi[it] * vm[it];
                s_y[ly][lx] = vg[it] * vm[it];
            }
        }
    }
    __syncthreads();
    // horizontal 11-tap pass: 26 rows x 8 groups of 4 columns
    if (tid < GHR_L_EH * (GHR_L_TW / 4)) {
        const int ly = tid >> 3, c0 = (tid & 7) * 4;
        float xs[16], ys[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f4 vy = *reinterpret_cast<const f4*>(&s_y[ly][c0 + 4 * q]);
            ys[4 * q] = vy.x; ys[4 * q + 1] = vy.y; ys[4 * q + 2] = vy.z; ys[4 * q + 3] = vy.w;
            if (HAVE_X) {
                const f4 vx = *reinterpret_cast<const f4*>(&s_x[ly][c0 + 4 * q]);
                xs[4 * q] = vx.x; xs[4 * q + 1] = vx.y; xs[4 * q + 2] = vx.z; xs[4 * q + 3] = vx.w;
            } else {
                xs[4 * q] = xs[4 * q + 1] = xs[4 * q + 2] = xs[4 * q + 3] = 0.f;
            }
        }
        float xx[14], yy[14], xy[14];
#pragma unroll
        for (int k = 0; k < 14; k++) { xx[k] = xs[k] * xs[k]; yy[k] = ys[k] * ys[k]; xy[k] = xs[k] * ys[k]; }
        float h[5][4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            float h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = c_ssim_w[k];
                if (HAVE_X) { h0 = fma_(w, xs[o + k], h0); h2 = fma_(w, xx[o + k], h2); h4 = fma_(w, xy[o + k], h4); }
                if (HAVE_Y) { h1 = fma_(w, ys[o + k], h1); h3 = fma_(w, yy[o + k], h3); }
            }
            h[0][o] = h0; h[1][o] = h1; h[2][o] = h2; h[3][o] = h3; h[4][o] = h4;
        }
        if (HAVE_X) {
            *reinterpret_cast<f4*>(&s_h[PX][ly][c0]) = f4{h[0][0], h[0][1], h[0][2], h[0][3]};
            *reinterpret_cast<f4*>(&s_h[PXX][ly][c0]) = f4{h[2][0], h[2][1], h[2][2], h[2][3]};
            *reinterpret_cast<f4*>(&s_h[PXY][ly][c0]) = f4{h[4][0], h[4][1], h[4][2], h[4][3]};
        }
        if (HAVE_Y) {
            *reinterpret_cast<f4*>(&s_h[PY][ly][c0]) = f4{h[1][0], h[1][1], h[1][2], h[1][3]};
            *reinterpret_cast<f4*>(&s_h[PYY][ly][c0]) = f4{h[3][0], h[3][1], h[3][2], h[3][3]};
        }
    }
    __syncthreads();
    // vertical pass: column tx, rows 2*tr and 2*tr + 1
    const int tx = tid & 31, tr = tid >> 5;
    float acc[2][NM];
#pragma unroll
    for (int o = 0; o < 2; o++)
#pragma unroll
        for (int k = 0; k < NM; k++) acc[o][k] = 0.f;
#pragma unroll
    for (int r = 0; r < 12; r++) {
        float v[NM];
#pragma unroll
        for (int k = 0; k < NM; k++) v[k] = s_h[k][2 * tr + r][tx];
        if (r < 11) {
            const float w = c_ssim_w[r];
#pragma unroll
            for (int k = 0; k < NM; k++) acc[0][k] = fma_(w, v[k], acc[0][k]);
        }
        if (r >= 1) {
            const float w = c_ssim_w[r - 1];
#pragma unroll
            for (int k = 0; k < NM; k++) acc[1][k] = fma_(w, v[k], acc[1][k]);
        }
    }
    const int gx = bx + tx;
    if (MODE == 2) {
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int gy = by + 2 * tr + o;
            if (gx < W && gy < H) {
                const size_t p = (size_t)gy * W + gx;
                a.stats_out[(0 * 3 + ch) * N + p] = acc[o][PY];
                a.stats_out[(1 * 3 + ch) * N + p] = acc[o][PYY];
            }
        }
        return;
    }
    const bool orient = ch == 2 && a.dir2d != nullptr;
    float sums[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // |image-gt|*m, ssim, |mask-gt_mask|, orientation num, den
#pragma unroll
    for (int o = 0; o < 2; o++) {
        const int gy = by + 2 * tr + o;
        if (gx < W && gy < H) {
            const size_t p = (size_t)gy * W + gx;
            float d0, d1, d2;
            const float mu2 = MODE == 1 ? a.gt_stats[(0 * 3 + ch) * N + p] : acc[o][PY];
            const float e22 = MODE == 1 ? a.gt_stats[(1 * 3 + ch) * N + p] : acc[o][PYY];
            sums[1] += ssim_point(acc[o][PX], mu2, acc[o][PXX], e22, acc[o][PXY], d0, d1, d2);
            a.maps[(0 * 3 + ch) * N + p] = d0;
            a.maps[(1 * 3 + ch) * N + p] = d1;
            a.maps[(2 * 3 + ch) * N + p] = d2;
            sums[0] += fabsf(img[p] - gt[p]) * (a.mask_colours ? m[p] : 1.0f);
            if (ch < 2) sums[2] += fabsf(a.mask[ch * N + p] - a.gt_mask[ch * N + p]);
            if (orient) {  // the blocks of the third colour channel also carry the orientation term
                const float w = a.gt_oconf[p];
                const OrientPix op = orient_pixel(a.dir2d[p], a.dir2d[N + p], a.oconf[p], a.gt_angle[p], a.gt_mask[p]);
                sums[3] += op.l * w;
                sums[4] += w;
            }
        }
    }
    block_sum_n<5>(sums, s_red);
    if (tid == 0) {
        const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * 97u) % GHR_LOSS_SLOTS;
        atomicAdd(&a.sums[GHR_LOSS_TERMS * slot + 0], sums[0]);
        atomicAdd(&a.sums[GHR_LOSS_TERMS * slot + 1], sums[1]);
        if (ch < 2) atomicAdd(&a.sums[GHR_LOSS_TERMS * slot + 2], sums[2]);
        if (orient) {
            atomicAdd(&a.sums[GHR_LOSS_TERMS * slot + 3], sums[3]);
            atomicAdd(&a.sums[GHR_LOSS_TERMS * slot + 4], sums[4]);
        }
    }
#endif
}

// grid (ceil(W/32), ceil(H/16), 3 colour channels), block 256
__global__ void __launch_bounds__(256) k_loss_fwd(LossArgs a) { loss_fwd_body<0>(a); }
__global__ void __launch_bounds__(256) k_loss_fwd_cached(LossArgs a) { loss_fwd_body<1>(a); }
__global__ void __launch_bounds__(256) k_loss_gt_stats(LossArgs a) { loss_fwd_body<2>(a); }

struct LossBwdArgs {
    int W, H;
    const float* image;
    const float* mask;
    const float* dir2d;      // may be NULL
    const float* oconf;
    const float* gt_image;
    const float* gt_mask;
    const float* gt_angle;
    const float* gt_oconf;
    int mask_colours;
    const float* maps;
    const float* aux;        // {sum of orientation weights, orientation-term-is-NaN flag} from k_loss_finalize
    const float* grad_loss;  // device scalar dL/dloss (may be null => 1)
    float w_l1, w_ssim, w_mask, w_orient;
    float* d_image;  // [3,H,W]
    float* d_mask;   // [2,H,W]
    float* d_dir2d;  // [2,H,W] or NULL
    float* d_oconf;  // [1,H,W] or NULL
    float* zero_a;   // optional planes to zero-fill (channels of a packed [10,H,W] gradient no loss term touches)
    float* zero_b;
};

__global__ void __launch_bounds__(256) k_loss_bwd(LossBwdArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float s_m[3][GHR_L_EH][GHR_L_XS];
    __shared__ __attribute__((aligned(16))) float s_h[3][GHR_L_EH][GHR_L_HS];
    const int ch = blockIdx.z;
    const int W = a.W, H = a.H;
    const size_t N = (size_t)W * H;
    const int bx = blockIdx.x * GHR_L_TW, by = blockIdx.y * GHR_L_TH;
    const int tid = threadIdx.x;
    {   // window load, all global loads first (see k_loss_fwd)
        constexpr int NIT = (GHR_L_EH * GHR_L_XS + 255) / 256;
        float v[NIT][3];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + 256 * it;
            const int ly = i / GHR_L_XS, lx = i - ly * GHR_L_XS;
            const int gx = bx + lx - GHR_SSIM_R, gy = by + ly - GHR_SSIM_R;
            const bool in = i < GHR_L_EH * GHR_L_XS && lx < GHR_L_EW && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t p = in ? (size_t)gy * W + gx : 0;
#pragma unroll
            for (int k = 0; k < 3; k++) v[it][k] = in ? a.maps[(k * 3 + ch) * N + p] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = tid + 256 * it;
            if (i < GHR_L_EH * GHR_L_XS) {
                const int ly = i / GHR_L_XS, lx = i - ly * GHR_L_XS;
#pragma unroll
                for (int k = 0; k < 3; k++) s_m[k][ly][lx] = v[it][k];
            }
        }
    }
    __syncthreads();
    if (tid < GHR_L_EH * (GHR_L_TW / 4)) {  // horizontal pass, 4 outputs per thread
        const int ly = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f4 t = *reinterpret_cast<const f4*>(&s_m[k][ly][c0 + 4 * q]);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
            float h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int t = 0; t < 11; t++) h[o] = fma_(c_ssim_w[t], v[o + t], h[o]);
            *reinterpret_cast<f4*>(&s_h[k][ly][c0]) = f4{h[0], h[1], h[2], h[3]};
        }
    }
    __syncthreads();
    const int tx = tid & 31, tr = tid >> 5;
    float c[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
    for (int r = 0; r < 12; r++) {
        const float v0 = s_h[0][2 * tr + r][tx], v1 = s_h[1][2 * tr + r][tx], v2 = s_h[2][2 * tr + r][tx];
        if (r < 11) {
            const float w = c_ssim_w[r];
            c[0][0] = fma_(w, v0, c[0][0]); c[0][1] = fma_(w, v1, c[0][1]); c[0][2] = fma_(w, v2, c[0][2]);
        }
        if (r >= 1) {
            const float w = c_ssim_w[r - 1];
            c[1][0] = fma_(w, v0, c[1][0]); c[1][1] = fma_(w, v1, c[1][1]); c[1][2] = fma_(w, v2, c[1][2]);
        }
    }
    const int gx = bx + tx;
    const float up = a.grad_loss ? a.grad_loss[0] : 1.0f;
    // the per-pixel divisions by the (uniform) element counts as multiplications by their reciprocals
    const float inv3N = 1.0f / (3.0f * (float)N), inv2N = 1.0f / (2.0f * (float)N);
#pragma unroll
    for (int o = 0; o < 2; o++) {
        const int gy = by + 2 * tr + o;
        if (!(gx < W && gy < H)) continue;
        const size_t p = (size_t)gy * W + gx;
        const float mm = a.mask_colours ? a.gt_mask[N + p] : 1.0f;
        const float im = a.image[ch * N + p], g = a.gt_image[ch * N + p];
        const float x = im * mm, y = g * mm;
        // d(mean ssim)/dx(p), then Lssim = 1 - mean  and x = image * m
        const float dssim_dx = (c[o][0] + 2.f * x * c[o][1] + y * c[o][2]) * inv3N;
        const float diff = im - g;
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        a.d_image[ch * N + p] = up * (a.w_l1 * sgn * mm * inv3N - a.w_ssim * dssim_dx * mm);
        if (ch < 2) {
            const float dm = a.mask[ch * N + p] - a.gt_mask[ch * N + p];
            const float sm = dm > 0.f ? 1.f : (dm < 0.f ? -1.f : 0.f);
            a.d_mask[ch * N + p] = up * a.w_mask * sm * inv2N;
            float* z = ch == 0 ? a.zero_a : a.zero_b;
            if (z) z[p] = 0.f;
        } else if (a.d_dir2d != nullptr) {
            float g0 = 0.f, g1 = 0.f, gc = 0.f;
            if (a.dir2d != nullptr && a.w_orient != 0.f && a.aux[1] == 0.f) {
                const OrientPix op = orient_pixel(a.dir2d[p], a.dir2d[N + p], a.oconf[p], a.gt_angle[p], a.gt_mask[p]);
                const float sc = up * a.w_orient * a.gt_oconf[p] * fast_rcp(a.aux[0]);
                g0 = sc * op.dl_dd0; g1 = sc * op.dl_dd1; gc = sc * op.dl_dconf;
            }
            a.d_dir2d[p] = g0;
            a.d_dir2d[N + p] = g1;
            a.d_oconf[p] = gc;
        }
    }
#endif
}

}  // namespace ghr
