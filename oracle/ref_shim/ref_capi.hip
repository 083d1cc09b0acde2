// ref_capi.hip -- C-ABI door into the REFERENCE's own rasterizer (CudaRasterizer::Rasterizer, hipified from
// /root/reference/ext/diff_gaussian_rasterization_hair/cuda_rasterizer into oracle/_ref/src by oracle/Makefile.ref).
// TEST INFRASTRUCTURE ONLY: it produces the golden vectors that pin oracle/ghr_oracle.c (tests/golden/
// make_reference_cuda_golden.py); the product never links or loads it.  It replaces the torch glue of
// R:rasterize_points.cu:35-206 (tensor allocation, resize lambdas) with hipMalloc'ed workspaces kept in one static
// context, and adds copy-outs of the internal state the parity tests compare (R:rasterizer_impl.cu:155-194).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <functional>
#include <string>
#include "rasterizer.h"
#include "rasterizer_impl.h"

namespace {
struct Buf {
    char* p = nullptr;
    size_t n = 0;
    char* grow(size_t want)
    {
        if (want > n) {
            if (p) (void)hipFree(p);
            p = nullptr;
            if (hipMalloc(&p, want) != hipSuccess) { n = 0; return nullptr; }
            n = want;
        }
        return p;
    }
};
struct Ctx {
    Buf geom, binning, img;
    int P = 0, W = 0, H = 0, R = 0;
    std::string err;
} g;
std::function<char*(size_t)> grower(Buf& b) { return [&b](size_t n) { return b.grow(n); }; }
int fail(const char* what) { g.err = what; return -1; }
}  // namespace

extern "C" {

const char* ghr_ref_last_error() { return g.err.c_str(); }

// R:rasterize_points.cu:35-123.  All pointers are device pointers; absent optionals are NULL (the reference gets
// data_ptr() == nullptr from an empty tensor).  Returns num_rendered or a negative value.
int ghr_ref_forward(int P, int W, int H, const float* bg, const float* means3D, const float* colors,
                    const float* opacity, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* conic_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, int prefiltered,
                    float* out_color, int* radii)
{
    g.P = P; g.W = W; g.H = H; g.R = 0;
    if (P == 0) return 0;
    // means2D_precomp: the reference's Python always passes a tensor (screenspace_points) and preprocessCUDA projects the
    // mean itself exactly when the pointer is NON-null (forward.cu:203-212: the other branch would read through it), so
    // any non-null pointer reproduces what the reference runs; its values are never read.
    try {
        g.R = CudaRasterizer::Rasterizer::forward(grower(g.geom), grower(g.binning), grower(g.img), P, /*D*/ 0, /*M*/ 0, bg,
                                                  W, H, means3D, /*means2D_precomp*/ means3D, /*shs*/ nullptr, colors,
                                                  opacity, scales, scale_modifier, rotations, cov3D_precomp,
                                                  conic_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                                                  prefiltered != 0, out_color, radii, /*debug*/ false);
    } catch (const std::exception& e) {
        return fail(e.what());
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail("device error in forward");
    return g.R;
}

// Copy-outs of the forward's internal state (device -> caller's DEVICE buffers), parsed exactly as the reference's
// backward re-parses them (fromChunk, R:rasterizer_impl.cu:376-378).
int ghr_ref_state(float* depths, float* means2D /*[P,2]*/, float* conic_opacity /*[P,4]*/, uint32_t* tiles_touched,
                  uint32_t* point_offsets, float* final_T /*[N]*/, uint32_t* n_contrib /*[N]*/,
                  uint32_t* ranges /*[T,2]*/, uint32_t* point_list /*[R]*/, uint64_t* keys /*[R]*/)
{
    using namespace CudaRasterizer;
    const int P = g.P, N = g.W * g.H;
    if (P == 0) return 0;
    char *gp = g.geom.p, *ip = g.img.p, *bp = g.binning.p;  // fromChunk advances the pointer it is handed (char*&)
    GeometryState geom = GeometryState::fromChunk(gp, P);
    ImageState img = ImageState::fromChunk(ip, N);
    BinningState bin = BinningState::fromChunk(bp, g.R);
    const int T = ((g.W + 15) / 16) * ((g.H + 15) / 16);
    auto cp = [](void* d, const void* s, size_t n) { return !d || n == 0 || hipMemcpy(d, s, n, hipMemcpyDeviceToDevice) == hipSuccess; };
    bool ok = cp(depths, geom.depths, 4u * P) && cp(means2D, geom.means2D, 8u * P) &&
              cp(conic_opacity, geom.conic_opacity, 16u * P) && cp(tiles_touched, geom.tiles_touched, 4u * P) &&
              cp(point_offsets, geom.point_offsets, 4u * P) && cp(final_T, img.accum_alpha, 4u * N) &&
              cp(n_contrib, img.n_contrib, 4u * N) && cp(ranges, img.ranges, 8u * T) &&
              cp(point_list, bin.point_list, 4u * (size_t)g.R) && cp(keys, bin.point_list_keys, 8u * (size_t)g.R);
    if (!ok || hipDeviceSynchronize() != hipSuccess) return fail("state copy failed");
    return 0;
}

// R:rasterize_points.cu:125-206 with the caller providing the (zero-filled) gradient tensors.
int ghr_ref_backward(const float* bg, const float* means3D, const int* radii, const float* colors, const float* scales,
                     float scale_modifier, const float* rotations, const float* cov3D_precomp,
                     const float* conic_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                     float tan_fovx, float tan_fovy, const float* dL_dpix, float* dL_dmean2D /*[P,3]*/,
                     float* dL_dconic /*[P,4]*/, float* dL_dopacity, float* dL_dcolor /*[P,C]*/, float* dL_dmean3D,
                     float* dL_dcov3D, float* dL_dscale, float* dL_drot)
{
    if (g.P == 0) return 0;
    try {
        CudaRasterizer::Rasterizer::backward(g.P, 0, 0, g.R, bg, g.W, g.H, means3D, nullptr, colors, scales, scale_modifier,
                                             rotations, cov3D_precomp, conic_precomp, viewmatrix, projmatrix, campos,
                                             tan_fovx, tan_fovy, radii, g.geom.p, g.binning.p, g.img.p, dL_dpix, dL_dmean2D,
                                             dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, /*dL_dsh*/ nullptr,
                                             dL_dscale, dL_drot, false);
    } catch (const std::exception& e) {
        return fail(e.what());
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail("device error in backward");
    return 0;
}

int ghr_ref_mark_visible(int P, float* means3D, float* viewmatrix, float* projmatrix, bool* present)
{
    if (P == 0) return 0;
    CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, present);
    return hipDeviceSynchronize() == hipSuccess ? 0 : fail("device error in markVisible");
}

}  // extern "C"
