// ghr_strands.h -- strand polylines -> one Gaussian per segment, and the hand-derived backward (round 6).
//
// Reference: initialize_gaussians_hair (src/scene/gaussian_model_strands.py:435-452; line for line the same in
// gaussian_model_latent_strands.py), run at the top of EVERY strand-stage iteration (src/train_strands.py:98-104):
//   pts      = origins + cat(0, cumsum(dirs, dim = 1))              [S, n_seg + 1, 3]
//   xyz      = (pts[:, 1:] + pts[:, :-1]) * 0.5                     mid-point of the segment
//   rotation = parallel_transport(x^, dir) = (1 + b.x, 0, -b.z, b.y),  b = dir / max(|dir|, 1e-12)
//              (src/utils/general_utils.py:150-160: un-normalised quaternion that turns the x axis onto the segment)
//   scaling  = (|dir| / 2, scale, scale)
// In PyTorch that is ~25 kernels forward and ~30 backward over 3 M rows (cumsum over an outer dimension, cat, norm, cross,
// flip ...): 1.4 ms of a 4.3-ms iteration at the reference's 30 000 strands x 99 segments.  Here: one kernel each way.
//
// The prefix sum of a strand is taken SEQUENTIALLY (one thread per (strand, component), like ATen's outer-dimension scan
// and like the CPU): the mid-points are bit-identical to the PyTorch form.  A workgroup stages the directions of `spb`
// consecutive strands (one contiguous slab) in LDS, 3 spb threads walk them, then all threads emit the rows coalesced.
#pragma once
#include "ghr_device.h"

namespace ghr {

#define GHR_STRAND_BLOCK 256
#ifndef GHR_STRAND_MAX_SEG  // include/ghr.h states the same number for the callers
#define GHR_STRAND_MAX_SEG 2048  // 2 x n_seg x 12 B of LDS for ONE strand must fit (48 KB)
#endif
#define GHR_STRAND_LDS_TARGET 32768   // bytes per workgroup the strands-per-block choice aims for
#define GHR_STRAND_NORM_EPS 1e-12f    // F.normalize's eps

GHR_HD int strands_per_block(int n_seg)
{
    int spb = GHR_STRAND_LDS_TARGET / (2 * 12 * n_seg);
    if (spb > GHR_STRAND_BLOCK / 3) spb = GHR_STRAND_BLOCK / 3;  // 3 spb scanning threads
    return spb < 1 ? 1 : spb;
}

// One (strand, component): d[k * 3] are the component's segment values; mid[k * 3] receives the mid-points.
GHR_HD void strand_scan_fwd(const float* d, float* mid, float origin, int n_seg)
{
    float acc = 0.f;
    float prev = origin + 0.f;  // pts[:, 0] = origins + zeros
    for (int k = 0; k < n_seg; k++) {
        acc = acc + d[3 * k];           // torch.cumsum: running sum in list order
        const float next = origin + acc;
        mid[3 * k] = (next + prev) * 0.5f;
        prev = next;
    }
}

// d xyz_k / d dirs_j = 1 (j < k), 1/2 (j == k): out_j = g_j / 2 + sum_{k > j} g_k, walked from the strand's tip.
GHR_HD void strand_scan_bwd(const float* g, float* out, int n_seg)
{
    float suffix = 0.f;
    for (int k = n_seg - 1; k >= 0; k--) {
        const float gk = g[3 * k];
        out[3 * k] = 0.5f * gk + suffix;
        suffix = suffix + gk;
    }
}

GHR_HD void strand_row_fwd(float dx, float dy, float dz, float scale, float* rot, float* scaling)
{
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float den = nrm > GHR_STRAND_NORM_EPS ? nrm : GHR_STRAND_NORM_EPS;
    rot[0] = 1.0f + dx / den;
    rot[1] = 0.f;
    rot[2] = -(dz / den);
    rot[3] = dy / den;
    scaling[0] = nrm * 0.5f;
    scaling[1] = scale;
    scaling[2] = scale;
}

// Cotangents of one row's rotation (4) and of scaling[0] -> the row's direction.
GHR_HD void strand_row_bwd(float dx, float dy, float dz, const float* d_rot, float d_s0, float* out)
{
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (d_rot != nullptr) {
        const float bx = d_rot[0], by = d_rot[3], bz = -d_rot[2];  // cotangent of b = dir / den
        if (nrm >= GHR_STRAND_NORM_EPS) {  // clamp_min passes the gradient at and above the bound: den = |dir|
            const float inv = 1.0f / nrm;
            const float t = (bx * dx + by * dy + bz * dz) * inv * inv * inv;
            gx = bx * inv - dx * t; gy = by * inv - dy * t; gz = bz * inv - dz * t;
        } else {
            const float inv = 1.0f / GHR_STRAND_NORM_EPS;
            gx = bx * inv; gy = by * inv; gz = bz * inv;
        }
    }
    if (nrm > 0.f) {  // torch.norm's backward: grad * x / norm, 0 where the norm is 0
        const float t = 0.5f * d_s0 / nrm;
        gx += t * dx; gy += t * dy; gz += t * dz;
    }
    out[0] = gx; out[1] = gy; out[2] = gz;
}

struct StrandArgs {
    int S, n_seg, spb;
    const float* origins;  // [S, 3]
    const float* dirs;     // [S, n_seg, 3]
    float scale;
    float* xyz;            // [S n_seg, 3]
    float* rot;            // [S n_seg, 4]
    float* scaling;        // [S n_seg, 3]
};

__global__ void __launch_bounds__(GHR_STRAND_BLOCK) k_strand_build(StrandArgs a)
{
    extern __shared__ float strand_lds[];  // [2][spb n_seg 3]: directions | mid-points
    const int tid = threadIdx.x, row = 3 * a.n_seg;
    const int s0 = blockIdx.x * a.spb;
    const int nb = (a.S - s0 < a.spb) ? a.S - s0 : a.spb;
    float* d = strand_lds;
    float* mid = strand_lds + a.spb * row;
    const size_t base = (size_t)s0 * row;
    const int n = nb * row;
    for (int i = tid; i < n; i += GHR_STRAND_BLOCK) d[i] = a.dirs[base + i];
    __syncthreads();
    if (tid < 3 * nb) {
        const int s = tid / 3, c = tid - 3 * s;
        strand_scan_fwd(d + s * row + c, mid + s * row + c, a.origins[(size_t)(s0 + s) * 3 + c], a.n_seg);
    }
    __syncthreads();
    for (int i = tid; i < n; i += GHR_STRAND_BLOCK) a.xyz[base + i] = mid[i];
    const int rows = nb * a.n_seg;
    const size_t r0 = (size_t)s0 * a.n_seg;
    for (int i = tid; i < rows; i += GHR_STRAND_BLOCK) {
        float q[4], sc[3];
        strand_row_fwd(d[3 * i], d[3 * i + 1], d[3 * i + 2], a.scale, q, sc);
        *reinterpret_cast<f4*>(a.rot + (r0 + i) * 4) = f4{q[0], q[1], q[2], q[3]};
        float* so = a.scaling + (r0 + i) * 3;
        so[0] = sc[0]; so[1] = sc[1]; so[2] = sc[2];
    }
}

struct StrandBwdArgs {
    int S, n_seg, spb;
    const float* dirs;       // [S, n_seg, 3]
    const float* d_xyz;      // [S n_seg, 3] or NULL
    const float* d_rot;      // [S n_seg, 4] or NULL
    const float* d_scaling;  // [S n_seg, 3] or NULL (only column 0 depends on the directions)
    const float* d_dir_rows; // [S n_seg, 3] or NULL: cotangent of the direction rows themselves (self._dir, a view of dirs),
                             // added LAST: (row + suffix sums) + d_dir_rows, the order autograd sums the two paths in
    float* d_dirs;           // [S, n_seg, 3], assigned
};

__global__ void __launch_bounds__(GHR_STRAND_BLOCK) k_strand_build_bwd(StrandBwdArgs a)
{
    extern __shared__ float strand_lds[];  // [2][spb n_seg 3]: d xyz | its suffix sums
    const int tid = threadIdx.x, row = 3 * a.n_seg;
    const int s0 = blockIdx.x * a.spb;
    const int nb = (a.S - s0 < a.spb) ? a.S - s0 : a.spb;
    float* g = strand_lds;
    float* acc = strand_lds + a.spb * row;
    const size_t base = (size_t)s0 * row;
    const int n = nb * row;
    if (a.d_xyz != nullptr) {
        for (int i = tid; i < n; i += GHR_STRAND_BLOCK) g[i] = a.d_xyz[base + i];
        __syncthreads();
        if (tid < 3 * nb) {
            const int s = tid / 3, c = tid - 3 * s;
            strand_scan_bwd(g + s * row + c, acc + s * row + c, a.n_seg);
        }
        __syncthreads();
    }
    const int rows = nb * a.n_seg;
    const size_t r0 = (size_t)s0 * a.n_seg;
    for (int i = tid; i < rows; i += GHR_STRAND_BLOCK) {
        const float* dp = a.dirs + base + 3 * i;
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.d_rot != nullptr) {
            const f4 t = *reinterpret_cast<const f4*>(a.d_rot + (r0 + i) * 4);
            q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
        }
        const float ds0 = a.d_scaling != nullptr ? a.d_scaling[(r0 + i) * 3] : 0.f;
        float o[3];
        strand_row_bwd(dp[0], dp[1], dp[2], a.d_rot != nullptr ? q : nullptr, ds0, o);
        if (a.d_xyz != nullptr) { o[0] += acc[3 * i]; o[1] += acc[3 * i + 1]; o[2] += acc[3 * i + 2]; }
        if (a.d_dir_rows != nullptr) {
            const float* e = a.d_dir_rows + base + 3 * i;
            o[0] += e[0]; o[1] += e[1]; o[2] += e[2];
        }
        float* out = a.d_dirs + base + 3 * i;
        out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
    }
}

}  // namespace ghr
