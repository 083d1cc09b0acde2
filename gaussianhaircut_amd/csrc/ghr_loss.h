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
// zero-padded convolution) and adds the L1 terms.  ~25 B/pixel/channel forward, ~30 B backward.
// Two forms of each kernel, bit-identical per output: 32 x 16 tiles (any image) and, for 16-B aligned rows, one wave
// marching down a 32-column strip (second half of this file; the one the trainer's 1080p images take).
#pragma once
#include "ghr_device.h"

namespace ghr {

#define GHR_SSIM_R 5
#define GHR_SSIM_T 16
#define GHR_SSIM_E (GHR_SSIM_T + 2 * GHR_SSIM_R)  // 26
#define GHR_LOSS_TERMS 5  // partial sums per slot: |image-gt|*m, ssim_map, |mask-gt_mask|, orient num, orient den
#define GHR_LOSS_AUX 8    // floats in front of the slots: {sum of the orientation weights, orientation-term-is-NaN flag, pad}
// Every workgroup of a forward kernel (tile form: a 32 x 16 tile of one channel; marching form: one wave = a strip segment of
// one channel) owns ONE slot of five partial sums and stores it (round 5; until then the workgroups added into 256 shared
// slots with atomics, which needed a zero-fill launch in front of every forward pass): nothing to initialise, and the fold
// in k_loss_finalize has a fixed order -- the loss value is the same bits run after run.
GHR_HD size_t loss_slots_tile(int W, int H) { return (size_t)3 * ((W + 31) / 32) * ((H + 15) / 16); }
GHR_HD size_t loss_slots_march(int W, int H, int seg) { return (size_t)3 * ((W + 31) / 32) * ((H + seg - 1) / seg); }

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
    float* sums;            // [GHR_LOSS_TERMS][n_slots] partial sums, one slot per workgroup (plain stores)
    const float* gt_stats;  // [2][3][H*W] window moments of the masked ground truth (mu2, E[y^2]) or NULL
    float* stats_out;       // k_loss_gt_stats: where those moments go
    int seg;                // marching kernels: rows of a strip per wave (a multiple of GHR_LM_ROWS)
    uint32_t n_slots;       // workgroups of the forward kernel = slots of `sums`, laid out [term][slot]
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
    if (tid == 0) {  // (sums[2] is 0 in the third channel's blocks, sums[3], sums[4] without the orientation term)
        float* dst = a.sums + ((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z));
#pragma unroll
        for (int k = 0; k < GHR_LOSS_TERMS; k++) dst[(size_t)k * a.n_slots] = sums[k];
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
    int seg;         // marching kernel: rows of a strip per wave (a multiple of GHR_LM_ROWS)
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

// ======================================================================================================================
// Marching form of the three kernels above for images whose rows are 16-B aligned (W % 4 == 0, 16-B aligned planes; the
// host picks it: ghr_capi.hip).  Same arithmetic in the same order per output -- every map, moment and gradient element is
// bit-identical to the tile kernels' -- at half their instructions per pixel and without a single workgroup barrier:
//   * ONE WAVE owns a strip of 32 columns and walks down `seg` rows of it, 8 output rows per pass, with its own LDS
//     (12-14 KB).  Nothing is shared between waves, so a wave never waits for another one: a
//     256-thread block marching down a 32-column strip (built first, same arithmetic) ran its SIMDs at 42 % VALU
//     utilisation -- a third of a wave's cycles went to the four barriers of a pass -- and was no faster than the tiles;
//   * the horizontally filtered rows stay in LDS: a pass filters only its 16 NEW input rows (one lane = four columns of one
//     row) and keeps the last ten for the next one, where a 32 x 16 tile filters 26 rows for 16 (1.6x the arithmetic, on 208
//     of 256 threads) and loads a 2.1x window;
//   * the vertical pass gives a lane four rows of one column from 14 LDS rows (3.5 reads per output and plane, not 6);
//   * the input window is the ALIGNED superset of the strip's 42 columns (32 + the 5-column halo on either side):
//     bx-8 .. bx+39, GHR_LM_WQ = 12 float4 per row -- 16-B loads from a 32-bit byte offset against a scalar base, never under a branch (clamped address,
//     then a select), and issued one pass ahead (the registers are carried around the loop); the tile kernels issue 15
//     one-float loads per thread, each under its own branch with its own 64-bit address (their forward runs 348 float
//     instructions among 365 other VALU and 306 scalar ones);
//   * the partial sums are carried down the strip and reduced once per wave by DPP (four steps inside a row, then four
//     v_readlane) instead of 30 ds_bpermute round trips per tile.
#define GHR_LM_TW 32                                  // columns of a strip: one 128-B line per row and plane
#define GHR_LM_ROWS 8                                 // output rows per pass
#define GHR_LM_WQ 12                                  // float4 per window row (columns bx-8 .. bx+39)
#define GHR_LM_WS 52                                  // row stride (floats) of the staged window
#define GHR_LM_HS 36                                  // row stride (floats) of the filtered planes
#define GHR_LM_SR (2 * GHR_SSIM_R)                     // staging rows: the ten rows above a segment are the largest batch
#define GHR_LM_HR (GHR_LM_ROWS + 2 * GHR_SSIM_R)      // rows of the filtered planes: 10 kept + 8 new

#if defined(__HIP_DEVICE_COMPILE__)
// base + 32-bit byte offset.  (The empty asm keeps the offset's zero-extension in the basic block of the access: the
// instruction selector only folds "scalar base + 32-bit lane offset" into the load when it sees both there; an offset
// computed in another block otherwise becomes a 64-bit VALU addition per access.)
template <typename T>
__device__ __forceinline__ T ldb(const void* base, uint32_t byte_off)
{
    asm("" : "+v"(byte_off));
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void stb(void* base, uint32_t byte_off, T v)
{
    asm("" : "+v"(byte_off));
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// acc = fma(w, x, acc) with the (uniform) window weight in a scalar register, as ONE v_fmac_f32.  Written out because the
// SLP vectoriser otherwise pairs the window FMAs into v_pk_fma_f32, which on this chip costs 1.65 plain instructions for two
// FMAs but needs every weight duplicated into an aligned SGPR pair and the operands shuffled into VGPR pairs:
// the marching kernels then spilled 50-80 scalar registers into VGPR lanes (v_readlane / v_writelane around every use).
__device__ __forceinline__ void fmac_s(float& acc, float w, float x)
{
    asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "s"(w), "v"(x));
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 64 lanes, valid in every lane (wave-uniform result)
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror
    v += dpp_move<0x140>(v);   // row_mirror: every lane of a row of 16 holds the row's sum
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
// LDS written by one lane, read by another lane of the same (only) wave of the workgroup: the hardware keeps a wave's LDS
// operations in order; this keeps the compiler from moving them across, and is no instruction
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// Float4 i of a batch of input rows starting at image row r0 (columns bx-8 .. bx+39): its row in the batch, its float4 in
// the row, and the byte offset of its pixels inside a plane (0 when outside the image or the batch)
struct WinPos { int row, q; bool in; uint32_t off; };
__device__ __forceinline__ WinPos win_pos(int i, int nvec, int bx, int r0, int W, int H)
{
    WinPos w;
    static_assert(GHR_LM_WQ == 12, "win_pos divides by GHR_LM_WQ with the multiply-shift below: 43691 / 2^19 = 1 / 12");
    w.row = (int)(((uint32_t)i * 43691u) >> 19);  // i / 12 for i < 2^13
    w.q = i - GHR_LM_WQ * w.row;
    const int gx = bx - 8 + 4 * w.q, gy = r0 + w.row;
    w.in = i < nvec && gx >= 0 && gx < W && gy >= 0 && gy < H;  // W % 4 == 0: a float4 is inside or outside as a whole
    w.off = w.in ? 4u * ((uint32_t)gy * (uint32_t)W + (uint32_t)gx) : 0u;
    return w;
}
__device__ __forceinline__ f4 sel4(bool c, const f4& v) { return f4{c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f}; }
// "These values exist HERE": keeps a computation from being sunk into the lane-masked region that stores its result.  A
// loaded register first used inside such a region is still pending on the path that skips the region (the compiler branches
// around a masked region when no lane is active), and every pending load it has lost track of costs a vmcnt(0) later on.
__device__ __forceinline__ void touch(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void touch4(f4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// floats 3..16 of five consecutive float4
__device__ __forceinline__ void row14(const float* p, float* o)
{
    const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4), c = *reinterpret_cast<const f4*>(p + 8),
             d = *reinterpret_cast<const f4*>(p + 12), e = *reinterpret_cast<const f4*>(p + 16);
    o[0] = a.w; o[1] = b.x; o[2] = b.y; o[3] = b.z; o[4] = b.w; o[5] = c.x; o[6] = c.y; o[7] = c.z; o[8] = c.w;
    o[9] = d.x; o[10] = d.y; o[11] = d.z; o[12] = d.w; o[13] = e.x;
}
// Vertical 11-tap pass for the four rows 4 rg .. 4 rg + 3 of column tx, from rows 4 rg .. 4 rg + 13 of the filtered planes
// (taps in ascending order for every output, like the tile kernels)
template <int NM>
__device__ __forceinline__ void vpass4(const float (*s_h)[GHR_LM_HR][GHR_LM_HS], int rg, int tx, float (*acc)[NM])
{
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int k = 0; k < NM; k++) acc[o][k] = 0.f;
#pragma unroll
    for (int r = 0; r < 14; r++) {
        float v[NM];
#pragma unroll
        for (int k = 0; k < NM; k++) v[k] = s_h[k][4 * rg + r][tx];
#pragma unroll
        for (int o = 0; o < 4; o++)
            if (r >= o && r - o < 11) {
                const float w = c_ssim_w[r - o];
#pragma unroll
                for (int k = 0; k < NM; k++) fmac_s(acc[o][k], w, v[k]);
            }
    }
}
// rows 8..17 of the filtered planes become rows 0..9 of the next pass: 10 rows x 8 float4 per plane.  The two ranges overlap:
// every read is issued before the first write (one wave: its LDS operations execute in program order)
template <int NM>
__device__ __forceinline__ void keep_last_rows(float (*s_h)[GHR_LM_HR][GHR_LM_HS], int lane)
{
    const int row = lane >> 3, c = (lane & 7) * 4;  // rows 0..7; rows 8, 9 by the lanes 0..15 again
    f4 t[NM][2];
#pragma unroll
    for (int k = 0; k < NM; k++) {
        t[k][0] = *reinterpret_cast<const f4*>(&s_h[k][GHR_LM_ROWS + row][c]);
        t[k][1] = *reinterpret_cast<const f4*>(&s_h[k][GHR_LM_ROWS + 8 + (row & 1)][c]);
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NM; k++) {
        *reinterpret_cast<f4*>(&s_h[k][row][c]) = t[k][0];
        if (lane < 16) *reinterpret_cast<f4*>(&s_h[k][8 + row][c]) = t[k][1];
    }
}
#endif

// MODE as in loss_fwd_body.  CH2: the wave works on the third colour channel (which also carries the orientation term)
// rather than on one of the first two (which carry the mask term): a template parameter, not a branch, so that the memory
// operations of a pass are the same sequence on every path -- the waits the compiler derives for the prefetched batch are
// then exact counts instead of drains.
template <int MODE, bool CH2>
__device__ __forceinline__ void loss_fwd_march(const LossArgs& a, float (*s_x)[GHR_LM_WS], float (*s_y)[GHR_LM_WS],
                                               float (*s_h)[GHR_LM_HR][GHR_LM_HS])
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool HAVE_X = MODE != 2;
    constexpr bool HAVE_Y = MODE != 1;
    constexpr int NM = MODE == 0 ? 5 : (MODE == 1 ? 3 : 2);
    constexpr int PX = 0, PY = MODE == 0 ? 1 : 0, PXX = MODE == 0 ? 2 : 1, PYY = MODE == 0 ? 3 : 1, PXY = MODE == 0 ? 4 : 2;
    const int ch = CH2 ? 2 : (int)blockIdx.z;
    const int W = a.W, H = a.H;
    const uint32_t N = (uint32_t)W * (uint32_t)H;
    // strips in runs per XCD (workgroup b runs on XCD b % 8; the grid's x extent is a multiple of 8): neighbouring strips
    // share their halo columns' lines in one L2
    const int nst = (W + GHR_LM_TW - 1) / GHR_LM_TW, sx = (int)(blockIdx.x & 7u) * ((nst + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (sx >= nst) return;
    const int bx = sx * GHR_LM_TW;
    const int y0 = blockIdx.y * a.seg, y1 = min(y0 + a.seg, H);
    const int lane = threadIdx.x;
    const float* img = a.image + (size_t)ch * N;
    const float* gt = a.gt_image + (size_t)ch * N;
    const float* m = a.gt_mask + N;  // gt_mask[1]
    const bool masked = a.mask_colours != 0;  // uniform

    float sums[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // |image-gt|*m, ssim, |mask-gt_mask|, orientation num, den
    // input rows r0 .. r0 + nvec / 12 - 1 -> rows 0.. of s_x (render x mask) and s_y (ground truth x mask), in two halves:
    // fetch() issues the loads of a batch while the previous one is still being worked on (the registers below are carried
    // around the loop), commit() stores it once the staging rows are free
    WinPos wp[2];
    f4 vi[2], vg[2], vm[2];
    auto fetch = [&](int r0, int nvec) {
#pragma unroll
        for (int it = 0; it < 2; it++) {  // (never under a branch: what lies outside the batch reads offset 0)
            wp[it] = win_pos(lane + 64 * it, nvec, bx, r0, W, H);
            vm[it] = masked ? ldb<f4>(m, wp[it].off) : f4{1.f, 1.f, 1.f, 1.f};
            if (HAVE_X) vi[it] = ldb<f4>(img, wp[it].off);
            vg[it] = ldb<f4>(gt, wp[it].off);
        }
    };
    // r0: image row of the batch's first row.  The L1 term |image - gt| * m is summed HERE, from the batch's own loads (every
    // pixel of the segment's rows and the strip's columns is fetched exactly once as part of a batch): three loads per pixel
    // less in the epilogue -- the kernel is bound by its L1 accesses (profiles/r03h)
    auto commit = [&](int nvec, int r0) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const f4 mm = sel4(wp[it].in, vm[it]);
            f4 x = {0.f, 0.f, 0.f, 0.f};
            if (HAVE_X) {
                x = sel4(wp[it].in, vi[it]);
                const int gy = r0 + wp[it].row;
                const bool own = wp[it].in && wp[it].q >= 2 && wp[it].q < 2 + GHR_LM_TW / 4 && gy >= y0 && gy < y1;
                const f4 gg = sel4(wp[it].in, vg[it]);
                const float l1 = (fabsf(x.x - gg.x) * mm.x + fabsf(x.y - gg.y) * mm.y) + (fabsf(x.z - gg.z) * mm.z + fabsf(x.w - gg.w) * mm.w);
                sums[0] += own ? l1 : 0.f;
                x = f4{x.x * mm.x, x.y * mm.y, x.z * mm.z, x.w * mm.w};
            }
            f4 g = sel4(wp[it].in, vg[it]);
            g = f4{g.x * mm.x, g.y * mm.y, g.z * mm.z, g.w * mm.w};
            if (HAVE_X) touch4(x);
            touch4(g);
            if (lane + 64 * it < nvec) {
                if (HAVE_X) *reinterpret_cast<f4*>(&s_x[wp[it].row][4 * wp[it].q]) = x;
                *reinterpret_cast<f4*>(&s_y[wp[it].row][4 * wp[it].q]) = g;
            }
        }
    };
    // horizontal 11-tap pass over the staged rows 0 .. nrows-1 -> rows dst0.. of the filtered planes: lane = (row, 4 columns)
    auto hpass = [&](int nrows, int dst0, int row0 = 0) {
        const int ly = row0 + (lane & 7), c0 = (lane >> 3) * 4;
        if (ly < nrows) {
            float xs[14], ys[14];
            row14(&s_y[ly][c0], ys);
            if (HAVE_X) row14(&s_x[ly][c0], xs);
            else {
#pragma unroll
                for (int k = 0; k < 14; k++) xs[k] = 0.f;
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
                    if (HAVE_X) { fmac_s(h0, w, xs[o + k]); fmac_s(h2, w, xx[o + k]); fmac_s(h4, w, xy[o + k]); }
                    if (HAVE_Y) { fmac_s(h1, w, ys[o + k]); fmac_s(h3, w, yy[o + k]); }
                }
                h[0][o] = h0; h[1][o] = h1; h[2][o] = h2; h[3][o] = h3; h[4][o] = h4;
            }
            const int d = dst0 + ly;
            if (HAVE_X) {
                *reinterpret_cast<f4*>(&s_h[PX][d][c0]) = f4{h[0][0], h[0][1], h[0][2], h[0][3]};
                *reinterpret_cast<f4*>(&s_h[PXX][d][c0]) = f4{h[2][0], h[2][1], h[2][2], h[2][3]};
                *reinterpret_cast<f4*>(&s_h[PXY][d][c0]) = f4{h[4][0], h[4][1], h[4][2], h[4][3]};
            }
            if (HAVE_Y) {
                *reinterpret_cast<f4*>(&s_h[PY][d][c0]) = f4{h[1][0], h[1][1], h[1][2], h[1][3]};
                *reinterpret_cast<f4*>(&s_h[PYY][d][c0]) = f4{h[3][0], h[3][1], h[3][2], h[3][3]};
            }
        }
    };

    const int tx = lane & 31, rg = lane >> 5;
    const int gx = bx + tx;
    const bool orient = MODE != 2 && CH2 && a.dir2d != nullptr;  // uniform
    // The loop is rotated so that a batch is committed at the END of the pass before it, in straight-line code behind the
    // pass's stores: the wait for its loads then is an exact count (vmcnt = the stores issued since) instead of a drain of
    // every outstanding store at the top of each pass (which is what the counter arithmetic at a loop header amounts to).
    // the ten input rows above the segment -> filtered rows 0..9
    fetch(y0 - GHR_SSIM_R, 2 * GHR_SSIM_R * GHR_LM_WQ);
    commit(2 * GHR_SSIM_R * GHR_LM_WQ, y0 - GHR_SSIM_R);
    wave_lds_fence();
    hpass(2 * GHR_SSIM_R, 0);
    hpass(2 * GHR_SSIM_R, 0, 8);
    fetch(y0 + GHR_SSIM_R, GHR_LM_ROWS * GHR_LM_WQ);
    wave_lds_fence();
    commit(GHR_LM_ROWS * GHR_LM_WQ, y0 + GHR_SSIM_R);
    for (int yb = y0; yb < y1; yb += GHR_LM_ROWS) {
        wave_lds_fence();
        hpass(GHR_LM_ROWS, 2 * GHR_SSIM_R);
        // The next batch, then the epilogue's loads (neither depends on the window), in flight across the vertical pass.
        // In THIS order: the batch is committed at the end of the pass, behind the pass's stores, and what the compiler can
        // prove about the memory counter there is "at least the epilogue's loads were issued after the batch's" -- enough
        // to wait for the batch without draining the stores.  With the batch issued last it waited for vmcnt(0): a full
        // store round trip per pass (measured: 112 us per kernel, SIMDs 25 % busy).
        const bool more = yb + GHR_LM_ROWS < y1;  // uniform
        fetch(yb + GHR_LM_ROWS + GHR_SSIM_R, more ? GHR_LM_ROWS * GHR_LM_WQ : 0);
        bool ok[4];
        uint32_t p4[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int gy = yb + 4 * rg + o;
            ok[o] = gx < W && gy < y1;
            p4[o] = ok[o] ? 4u * ((uint32_t)gy * (uint32_t)W + (uint32_t)gx) : 0u;
        }
        float l_mu2[4], l_e22[4], l_mk[4], l_gmk[4];
        if (MODE != 2) {
#pragma unroll
            for (int o = 0; o < 4; o++) {
                if (MODE == 1) {
                    l_mu2[o] = ldb<float>(a.gt_stats + (size_t)(0 * 3 + ch) * N, p4[o]);
                    l_e22[o] = ldb<float>(a.gt_stats + (size_t)(1 * 3 + ch) * N, p4[o]);
                }
                if (!CH2) {
                    l_mk[o] = ldb<float>(a.mask + (size_t)ch * N, p4[o]);
                    l_gmk[o] = ldb<float>(a.gt_mask + (size_t)ch * N, p4[o]);
                }
            }
        }
        wave_lds_fence();
        float acc[4][NM];
        vpass4<NM>(s_h, rg, tx, acc);
        if (MODE == 2) {
#pragma unroll
            for (int o = 0; o < 4; o++) { touch(acc[o][PY]); touch(acc[o][PYY]); }
#pragma unroll
            for (int o = 0; o < 4; o++)
                if (ok[o]) {
                    stb<float>(a.stats_out + (size_t)(0 * 3 + ch) * N, p4[o], acc[o][PY]);
                    stb<float>(a.stats_out + (size_t)(1 * 3 + ch) * N, p4[o], acc[o][PYY]);
                }
        } else {
            float d0[4], d1[4], d2[4];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const float mu2 = MODE == 1 ? l_mu2[o] : acc[o][PY];
                const float e22 = MODE == 1 ? l_e22[o] : acc[o][PYY];
                const float sv = ssim_point(acc[o][PX], mu2, acc[o][PXX], e22, acc[o][PXY], d0[o], d1[o], d2[o]);
                sums[1] += ok[o] ? sv : 0.f;
                if (!CH2) sums[2] += ok[o] ? fabsf(l_mk[o] - l_gmk[o]) : 0.f;
            }
            if (CH2 && orient) {  // the waves of the third colour channel also carry the orientation term (before the stores:
                                  // waiting for its loads then does not wait for them)
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const float w = ldb<float>(a.gt_oconf, p4[o]);
                    const OrientPix op = orient_pixel(ldb<float>(a.dir2d, p4[o]), ldb<float>(a.dir2d + N, p4[o]),
                                                      ldb<float>(a.oconf, p4[o]), ldb<float>(a.gt_angle, p4[o]),
                                                      ldb<float>(a.gt_mask, p4[o]));
                    sums[3] += ok[o] ? op.l * w : 0.f;
                    sums[4] += ok[o] ? w : 0.f;
                }
            }
            // (ALL the values first, then nothing but the stores under the lane masks)
#pragma unroll
            for (int o = 0; o < 4; o++) { touch(d0[o]); touch(d1[o]); touch(d2[o]); }
#pragma unroll
            for (int k = 0; k < 5; k++) touch(sums[k]);
#pragma unroll
            for (int o = 0; o < 4; o++) {
                if (ok[o]) {
                    stb<float>(a.maps + (size_t)(0 * 3 + ch) * N, p4[o], d0[o]);
                    stb<float>(a.maps + (size_t)(1 * 3 + ch) * N, p4[o], d1[o]);
                    stb<float>(a.maps + (size_t)(2 * 3 + ch) * N, p4[o], d2[o]);
                }
            }
        }
        wave_lds_fence();  // every read of the vertical pass and of the last horizontal pass is done
        if (more) keep_last_rows<NM>(s_h, lane);
        commit(more ? GHR_LM_ROWS * GHR_LM_WQ : 0, yb + GHR_LM_ROWS + GHR_SSIM_R);
    }
    if (MODE == 2) return;
#pragma unroll
    for (int k = 0; k < 5; k++) sums[k] = wave_sum(sums[k]);
    if (lane == 0) {  // this wave's slot: strip sx (< nst: the grid's padding strips have returned), segment, channel
        float* dst = a.sums + ((size_t)sx + (size_t)nst * (blockIdx.y + (size_t)gridDim.y * (unsigned)ch));
#pragma unroll
        for (int k = 0; k < GHR_LOSS_TERMS; k++) dst[(size_t)k * a.n_slots] = sums[k];
    }
#endif
}

// grid (8 ceil(ceil(W/32) / 8), ceil(H/seg), 3 colour channels), block 64: one wave per strip segment
template <int MODE>
__device__ __forceinline__ void loss_fwd_march_any(const LossArgs& a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NM = MODE == 0 ? 5 : (MODE == 1 ? 3 : 2);
    __shared__ __attribute__((aligned(16))) float s_x[GHR_LM_SR][GHR_LM_WS], s_y[GHR_LM_SR][GHR_LM_WS];
    __shared__ __attribute__((aligned(16))) float s_h[NM][GHR_LM_HR][GHR_LM_HS];
    if (blockIdx.z == 2) loss_fwd_march<MODE, true>(a, s_x, s_y, s_h);
    else loss_fwd_march<MODE, false>(a, s_x, s_y, s_h);
#endif
}
__global__ void __launch_bounds__(64) k_loss_fwd_v(LossArgs a) { loss_fwd_march_any<0>(a); }
__global__ void __launch_bounds__(64) k_loss_fwd_cached_v(LossArgs a) { loss_fwd_march_any<1>(a); }
__global__ void __launch_bounds__(64) k_loss_gt_stats_v(LossArgs a) { loss_fwd_march_any<2>(a); }

template <bool CH2>
__device__ __forceinline__ void loss_bwd_march(const LossBwdArgs& a, float (*s_m)[GHR_LM_SR][GHR_LM_WS],
                                               float (*s_h)[GHR_LM_HR][GHR_LM_HS])
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int ch = CH2 ? 2 : (int)blockIdx.z;
    const int W = a.W, H = a.H;
    const uint32_t N = (uint32_t)W * (uint32_t)H;
    // strips in runs per XCD (workgroup b runs on XCD b % 8; the grid's x extent is a multiple of 8): neighbouring strips
    // share their halo columns' lines in one L2
    const int nst = (W + GHR_LM_TW - 1) / GHR_LM_TW, sx = (int)(blockIdx.x & 7u) * ((nst + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (sx >= nst) return;
    const int bx = sx * GHR_LM_TW;
    const int y0 = blockIdx.y * a.seg, y1 = min(y0 + a.seg, H);
    const int lane = threadIdx.x;

    // (fetch / commit: see loss_fwd_march)
    WinPos wp[2];
    f4 v[2][3];
    auto fetch = [&](int r0, int nvec) {
#pragma unroll
        for (int it = 0; it < 2; it++) {  // (never under a branch: what lies outside the batch reads offset 0)
            wp[it] = win_pos(lane + 64 * it, nvec, bx, r0, W, H);
#pragma unroll
            for (int k = 0; k < 3; k++) v[it][k] = ldb<f4>(a.maps + (size_t)(k * 3 + ch) * N, wp[it].off);
        }
    };
    auto commit = [&](int nvec) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            f4 t[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { t[k] = sel4(wp[it].in, v[it][k]); touch4(t[k]); }
            if (lane + 64 * it < nvec)
#pragma unroll
                for (int k = 0; k < 3; k++) *reinterpret_cast<f4*>(&s_m[k][wp[it].row][4 * wp[it].q]) = t[k];
        }
    };
    auto hpass = [&](int nrows, int dst0, int row0 = 0) {
        const int ly = row0 + (lane & 7), c0 = (lane >> 3) * 4;
        if (ly < nrows) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float t[14];
                row14(&s_m[k][ly][c0], t);
                float h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < 4; o++)
#pragma unroll
                    for (int j = 0; j < 11; j++) fmac_s(h[o], c_ssim_w[j], t[o + j]);
                *reinterpret_cast<f4*>(&s_h[k][dst0 + ly][c0]) = f4{h[0], h[1], h[2], h[3]};
            }
        }
    };

    const int tx = lane & 31, rg = lane >> 5;
    const int gx = bx + tx;
    const bool masked = a.mask_colours != 0;
    const bool orient_out = CH2 && a.d_dir2d != nullptr;                                       // uniform
    const bool orient_on = orient_out && a.dir2d != nullptr && a.w_orient != 0.f;              // uniform
    const float up = a.grad_loss ? a.grad_loss[0] : 1.0f;
    const float inv3N = 1.0f / (3.0f * (float)N), inv2N = 1.0f / (2.0f * (float)N);
    const bool orient_live = orient_on && a.aux[1] == 0.f;  // uniform
    const float inv_wsum = orient_live ? fast_rcp(a.aux[0]) : 0.f;
    // first two channels: the plane to zero-fill, or (none asked for) the mask gradient's own plane written twice
    float* zplane = nullptr;
    if (!CH2) {
        zplane = ch == 0 ? a.zero_a : a.zero_b;
    }
    const bool zfill = zplane != nullptr;  // uniform
    float* d_mask_ch = CH2 ? nullptr : a.d_mask + (size_t)ch * N;
    float* zdst = zfill ? zplane : d_mask_ch;

    // (rotated loop: see loss_fwd_march)
    fetch(y0 - GHR_SSIM_R, 2 * GHR_SSIM_R * GHR_LM_WQ);
    commit(2 * GHR_SSIM_R * GHR_LM_WQ);
    wave_lds_fence();
    hpass(2 * GHR_SSIM_R, 0);
    hpass(2 * GHR_SSIM_R, 0, 8);
    fetch(y0 + GHR_SSIM_R, GHR_LM_ROWS * GHR_LM_WQ);
    wave_lds_fence();
    commit(GHR_LM_ROWS * GHR_LM_WQ);
    for (int yb = y0; yb < y1; yb += GHR_LM_ROWS) {
        wave_lds_fence();
        hpass(GHR_LM_ROWS, 2 * GHR_SSIM_R);
        // the next batch BEFORE the epilogue's loads (see loss_fwd_march)
        const bool more = yb + GHR_LM_ROWS < y1;  // uniform
        fetch(yb + GHR_LM_ROWS + GHR_SSIM_R, more ? GHR_LM_ROWS * GHR_LM_WQ : 0);
        bool ok[4];
        uint32_t p4[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const int gy = yb + 4 * rg + o;
            ok[o] = gx < W && gy < y1;
            p4[o] = ok[o] ? 4u * ((uint32_t)gy * (uint32_t)W + (uint32_t)gx) : 0u;
        }
        float l_im[4], l_g[4], l_m[4], l_mk[4], l_gmk[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            l_im[o] = ldb<float>(a.image + (size_t)ch * N, p4[o]);
            l_g[o] = ldb<float>(a.gt_image + (size_t)ch * N, p4[o]);
            l_m[o] = masked ? ldb<float>(a.gt_mask + N, p4[o]) : 1.0f;
            if (!CH2) {
                l_mk[o] = ldb<float>(a.mask + (size_t)ch * N, p4[o]);
                l_gmk[o] = ldb<float>(a.gt_mask + (size_t)ch * N, p4[o]);
            }
        }
        wave_lds_fence();
        float c[4][3];
        vpass4<3>(s_h, rg, tx, c);
        float di[4], dmk[4], zval[4];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const float mm = l_m[o];
            const float im = l_im[o], g = l_g[o];
            const float x = im * mm, y = g * mm;
            // d(mean ssim)/dx(p), then Lssim = 1 - mean  and x = image * m
            const float dssim_dx = (c[o][0] + 2.f * x * c[o][1] + y * c[o][2]) * inv3N;
            const float diff = im - g;
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            di[o] = up * (a.w_l1 * sgn * mm * inv3N - a.w_ssim * dssim_dx * mm);
            dmk[o] = 0.f;
            if (!CH2) {
                const float dm = l_mk[o] - l_gmk[o];
                const float sm = dm > 0.f ? 1.f : (dm < 0.f ? -1.f : 0.f);
                dmk[o] = up * a.w_mask * sm * inv2N;
            }
            zval[o] = zfill ? 0.f : dmk[o];
        }
        float g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f}, gc[4] = {0.f, 0.f, 0.f, 0.f};
        if (CH2 && orient_live) {  // (before the stores: waiting for its loads then does not wait for them)
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const OrientPix op = orient_pixel(ldb<float>(a.dir2d, p4[o]), ldb<float>(a.dir2d + N, p4[o]),
                                                  ldb<float>(a.oconf, p4[o]), ldb<float>(a.gt_angle, p4[o]),
                                                  ldb<float>(a.gt_mask, p4[o]));
                const float sc = up * a.w_orient * ldb<float>(a.gt_oconf, p4[o]) * inv_wsum;
                g0[o] = sc * op.dl_dd0; g1[o] = sc * op.dl_dd1; gc[o] = sc * op.dl_dconf;
            }
        }
        // (ALL the values first, then nothing but the stores under the lane masks)
#pragma unroll
        for (int o = 0; o < 4; o++) {
            touch(di[o]);
            if (!CH2) { touch(dmk[o]); touch(zval[o]); }
            else { touch(g0[o]); touch(g1[o]); touch(gc[o]); }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if (ok[o]) {
                stb<float>(a.d_image + (size_t)ch * N, p4[o], di[o]);
                if (!CH2) {
                    stb<float>(d_mask_ch, p4[o], dmk[o]);
                    stb<float>(zdst, p4[o], zval[o]);
                } else if (orient_out) {
                    stb<float>(a.d_dir2d, p4[o], g0[o]);
                    stb<float>(a.d_dir2d + N, p4[o], g1[o]);
                    stb<float>(a.d_oconf, p4[o], gc[o]);
                }
            }
        }
        wave_lds_fence();
        if (more) keep_last_rows<3>(s_h, lane);
        commit(more ? GHR_LM_ROWS * GHR_LM_WQ : 0);
    }
#endif
}

__global__ void __launch_bounds__(64) k_loss_bwd_v(LossBwdArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float s_m[3][GHR_LM_SR][GHR_LM_WS];
    __shared__ __attribute__((aligned(16))) float s_h[3][GHR_LM_HR][GHR_LM_HS];
    if (blockIdx.z == 2) loss_bwd_march<true>(a, s_m, s_h);
    else loss_bwd_march<false>(a, s_m, s_h);
#endif
}

}  // namespace ghr
