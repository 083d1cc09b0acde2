// ghr_geom_bwd.h -- per-Gaussian backward epilogue: unpacks the 64-B gradient lines written by k_render_bwd into
// the reference's output tensors and, in kernel-geometry mode (B), runs the geometry backward:
//   K9  computeCov2DCUDA  R:cuda_rasterizer/backward.cu:144-274   dL/dconic -> dL/dcov3D, dL/dmean3D (assign)
//   K10 preprocessCUDA    R:cuda_rasterizer/backward.cu:346-400   dL/dmean2D -> dL/dmean3D (add), dL/dcov3D -> scale/rot
// One fused pass (the reference launches K9 and K10 separately and pre-zeroes 34 floats/Gaussian with torch::zeros,
// rasterize_points.cu:160-168); every output element is written here, so the caller may pass uninitialised buffers.
// In pipeline mode (A) K9 is skipped and K10 is a no-op in the reference (backward.cu:371,398,588): outputs are 0.
#pragma once
#include "ghr_preprocess.h"

namespace ghr {

struct GeomBwdArgs {
    int P;
    const float* means3D;
    const int* radii;
    const float* scales;
    const float* rotations;
    const float* cov3D;          // cov3D actually used in forward (precomp or geom workspace); mode B only
    const float* conic_precomp;  // != NULL => mode A
    const float* view;
    const float* proj;
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y;
    const float* ginst;  // [R][16] per-instance gradient lines in tile-list order (format: ghr_device.h LineAcc)
    const uint32_t* inst_line;  // [R] line of every instance (instances numbered by rect4_slot), from the tile sort
    uint32_t ginst_rows;        // lines in ginst / entries in inst_line (bound of the gather)
    const rect4* rects;
    const f4* rec;       // [P][4] packed render records (pixel mean, conic, opacity: needed to unpack the lines)
    float half_w, half_h;  // 0.5 W, 0.5 H (backward.cu:464-465)
    float* dL_dmeans2D;  // [P][3]
    float* dL_dconic;    // [P][4]
    float* dL_dconic3;   // [P][3] or NULL: (d/da, d/db, d/dc) -- what the reference's Python wrapper restacks dL_dconic into
    float* dL_dopacity;  // [P]
    float* dL_dcolors;   // [P][C]
    float* dL_dmeans3D;  // [P][3]
    float* dL_dcov3D;    // [P][6]
    float* dL_dscales;   // [P][3]
    float* dL_drots;     // [P][4]
};

// backward.cu:278-341
GHR_HD void cov3d_bwd(const float* s3, float mod, const float* q4, const float* dcov, float* dscale, float* drot)
{
    const float r = q4[0], x = q4[1], y = q4[2], z = q4[3];
    m3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    const float s[3] = {mod * s3[0], mod * s3[1], mod * s3[2]};
    m3 S = {};
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    const m3 M = mul(S, R);
    m3 dSig;
    dSig.m[0][0] = dcov[0]; dSig.m[0][1] = 0.5f * dcov[1]; dSig.m[0][2] = 0.5f * dcov[2];
    dSig.m[1][0] = 0.5f * dcov[1]; dSig.m[1][1] = dcov[3]; dSig.m[1][2] = 0.5f * dcov[4];
    dSig.m[2][0] = 0.5f * dcov[2]; dSig.m[2][1] = 0.5f * dcov[4]; dSig.m[2][2] = dcov[5];
    m3 M2;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) M2.m[c][k] = 2.0f * M.m[c][k];
    const m3 dM = mul(M2, dSig);
    const m3 Rt = transpose(R);
    m3 dMt = transpose(dM);
#pragma unroll
    for (int k = 0; k < 3; k++)
        dscale[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) dMt.m[k][c] *= s[k];
#define GHR_D(a, b) dMt.m[a][b]
    drot[0] = 2 * z * (GHR_D(0, 1) - GHR_D(1, 0)) + 2 * y * (GHR_D(2, 0) - GHR_D(0, 2)) + 2 * x * (GHR_D(1, 2) - GHR_D(2, 1));
    drot[1] = 2 * y * (GHR_D(1, 0) + GHR_D(0, 1)) + 2 * z * (GHR_D(2, 0) + GHR_D(0, 2)) + 2 * r * (GHR_D(1, 2) - GHR_D(2, 1)) - 4 * x * (GHR_D(2, 2) + GHR_D(1, 1));
    drot[2] = 2 * x * (GHR_D(1, 0) + GHR_D(0, 1)) + 2 * r * (GHR_D(2, 0) - GHR_D(0, 2)) + 2 * z * (GHR_D(1, 2) + GHR_D(2, 1)) - 4 * y * (GHR_D(2, 2) + GHR_D(0, 0));
    drot[3] = 2 * r * (GHR_D(0, 1) - GHR_D(1, 0)) + 2 * x * (GHR_D(2, 0) + GHR_D(0, 2)) + 2 * y * (GHR_D(1, 2) + GHR_D(2, 1)) - 4 * z * (GHR_D(1, 1) + GHR_D(0, 0));
#undef GHR_D
}

GHR_HD void geom_bwd_one(const GeomBwdArgs& a, int idx, const float* g)
{
    const float gmx = g[0], gmy = g[1], gca = g[2], gcb = g[3], gcc = g[4];
    a.dL_dmeans2D[3 * idx] = gmx;
    a.dL_dmeans2D[3 * idx + 1] = gmy;
    a.dL_dmeans2D[3 * idx + 2] = 0.f;
    a.dL_dconic[4 * idx] = gca;
    a.dL_dconic[4 * idx + 1] = gcb;
    a.dL_dconic[4 * idx + 2] = 0.f;
    a.dL_dconic[4 * idx + 3] = gcc;
    if (a.dL_dconic3 != nullptr) {  // diff_gaussian_rasterization/__init__.py:149-153: [xx, 2 * xy, yy]
        a.dL_dconic3[3 * idx] = gca;
        a.dL_dconic3[3 * idx + 1] = 2.f * gcb;
        a.dL_dconic3[3 * idx + 2] = gcc;
    }
    a.dL_dopacity[idx] = g[5];
#pragma unroll
    for (int c = 0; c < GHR_C; c++) a.dL_dcolors[(size_t)GHR_C * idx + c] = g[6 + c];

    float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.conic_precomp == nullptr && a.radii[idx] > 0) {
        const float mx = a.means3D[3 * idx], my = a.means3D[3 * idx + 1], mz = a.means3D[3 * idx + 2];
        // ---- K9, backward.cu:159-273
        Cov2DCtx c;
        cov2d_setup(c, mx, my, mz, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, a.cov3D + 6 * (size_t)idx, a.view);
        const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
        const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
        float ca, cb, cc;
        cov2d_eval(c, ca, cb, cc);
        const m3& T = c.T;
        const m3& V = c.Vrk;
        const m3& Wm = c.Wm;
        const float denom = ca * cc - cb * cb;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * gca + 2 * cb * cc * gcb + (denom - ca * cc) * gcc);
            dL_dc = denom2inv * (-ca * ca * gcc + 2 * ca * cb * gcb + (denom - ca * cc) * gca);
            dL_db = denom2inv * 2 * (cb * cc * gca - (denom + 2 * cb * cb) * gcb + ca * cb * gcc);
            dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
            dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
            dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
            dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
            dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
            dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
        }
        const float dT00 = 2 * (T.m[0][0] * V.m[0][0] + T.m[0][1] * V.m[0][1] + T.m[0][2] * V.m[0][2]) * dL_da +
                           (T.m[1][0] * V.m[0][0] + T.m[1][1] * V.m[0][1] + T.m[1][2] * V.m[0][2]) * dL_db;
        const float dT01 = 2 * (T.m[0][0] * V.m[1][0] + T.m[0][1] * V.m[1][1] + T.m[0][2] * V.m[1][2]) * dL_da +
                           (T.m[1][0] * V.m[1][0] + T.m[1][1] * V.m[1][1] + T.m[1][2] * V.m[1][2]) * dL_db;
        const float dT02 = 2 * (T.m[0][0] * V.m[2][0] + T.m[0][1] * V.m[2][1] + T.m[0][2] * V.m[2][2]) * dL_da +
                           (T.m[1][0] * V.m[2][0] + T.m[1][1] * V.m[2][1] + T.m[1][2] * V.m[2][2]) * dL_db;
        const float dT10 = 2 * (T.m[1][0] * V.m[0][0] + T.m[1][1] * V.m[0][1] + T.m[1][2] * V.m[0][2]) * dL_dc +
                           (T.m[0][0] * V.m[0][0] + T.m[0][1] * V.m[0][1] + T.m[0][2] * V.m[0][2]) * dL_db;
        const float dT11 = 2 * (T.m[1][0] * V.m[1][0] + T.m[1][1] * V.m[1][1] + T.m[1][2] * V.m[1][2]) * dL_dc +
                           (T.m[0][0] * V.m[1][0] + T.m[0][1] * V.m[1][1] + T.m[0][2] * V.m[1][2]) * dL_db;
        const float dT12 = 2 * (T.m[1][0] * V.m[2][0] + T.m[1][1] * V.m[2][1] + T.m[1][2] * V.m[2][2]) * dL_dc +
                           (T.m[0][0] * V.m[2][0] + T.m[0][1] * V.m[2][1] + T.m[0][2] * V.m[2][2]) * dL_db;
        const float dJ00 = Wm.m[0][0] * dT00 + Wm.m[0][1] * dT01 + Wm.m[0][2] * dT02;
        const float dJ02 = Wm.m[2][0] * dT00 + Wm.m[2][1] * dT01 + Wm.m[2][2] * dT02;
        const float dJ11 = Wm.m[1][0] * dT10 + Wm.m[1][1] * dT11 + Wm.m[1][2] * dT12;
        const float dJ12 = Wm.m[2][0] * dT10 + Wm.m[2][1] * dT11 + Wm.m[2][2] * dT12;
        const float tz = 1.f / c.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float h_x = a.focal_x, h_y = a.focal_y;
        const float dtx = x_grad_mul * -h_x * tz2 * dJ02;
        const float dty = y_grad_mul * -h_y * tz2 * dJ12;
        const float dtz = -h_x * tz2 * dJ00 - h_y * tz2 * dJ11 + (2 * h_x * c.tx) * tz3 * dJ02 + (2 * h_y * c.ty) * tz3 * dJ12;
        const float* vm = a.view;  // transformVec4x3Transpose, auxiliary.h:89-97
        dmean[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
        dmean[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        dmean[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;

        // ---- K10, backward.cu:371-391
        const float* pm = a.proj;
        const float hw = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12]) * m_w * m_w;
        const float mul2 = (pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13]) * m_w * m_w;
        dmean[0] += (pm[0] * m_w - pm[3] * mul1) * gmx + (pm[1] * m_w - pm[3] * mul2) * gmy;
        dmean[1] += (pm[4] * m_w - pm[7] * mul1) * gmx + (pm[5] * m_w - pm[7] * mul2) * gmy;
        dmean[2] += (pm[8] * m_w - pm[11] * mul1) * gmx + (pm[9] * m_w - pm[11] * mul2) * gmy;

        if (a.scales != nullptr) {  // backward.cu:398-399
            const float s3[3] = {a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]};
            const float q4[4] = {a.rotations[4 * idx], a.rotations[4 * idx + 1], a.rotations[4 * idx + 2],
                                 a.rotations[4 * idx + 3]};
            cov3d_bwd(s3, a.scale_modifier, q4, dcov, dscale, drot);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) a.dL_dmeans3D[3 * idx + i] = dmean[i];
#pragma unroll
    for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * idx + i] = dcov[i];
#pragma unroll
    for (int i = 0; i < 3; i++) a.dL_dscales[3 * idx + i] = dscale[i];
#pragma unroll
    for (int i = 0; i < 4; i++) a.dL_drots[4 * idx + i] = drot[i];
}

__global__ void __launch_bounds__(GHR_BLOCK) k_geom_bwd(GeomBwdArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int idx = blockIdx.x * GHR_BLOCK + threadIdx.x;
    rect4 r = make_rect4(0, 0, 0, 0, 0u);
    if (idx < a.P) r = a.rects[idx];
    f4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
    if (idx < a.P && rect4_area(r) != 0u) { r0 = a.rec[4 * (size_t)idx]; r1 = a.rec[4 * (size_t)idx + 1]; }
    float ga[16];
    gather_inst_grads_wave(a.ginst, a.inst_line, r, r0, r1, a.half_w, a.half_h, ga, a.ginst_rows);  // every lane of the wave takes part
    if (idx < a.P) geom_bwd_one(a, idx, ga);
#endif
}

}  // namespace ghr
