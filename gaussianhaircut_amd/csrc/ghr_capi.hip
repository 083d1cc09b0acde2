// ghr_capi.hip -- C ABI of libghr_hip.so (declared in include/ghr.h).  Host-side orchestration only:
// workspace carving, argument checks, kernel launches on the caller's stream.  Replaces
// R:rasterize_points.cu (torch glue) + R:cuda_rasterizer/rasterizer_impl.cu:155-441 (state carving, forward, backward).
#include "../../include/ghr.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ghr_binning.h"
#include "ghr_device.h"
#include "ghr_adam.h"
#include "ghr_geom_bwd.h"
#include "ghr_loss.h"
#include "ghr_preprocess.h"
#include "ghr_project.h"
#include "ghr_render_bwd.h"
#include "ghr_render_bwd2.h"
#include "ghr_render_bwd3.h"
#include "ghr_render_fwd.h"
#include "ghr_strands.h"

namespace {

thread_local char g_err[512] = "";
// Process-wide (NOT thread_local): torch's autograd engine calls ghr_backward from its own worker thread.
hipEvent_t g_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // fwd start/stop, bwd start/stop

// K8 variant (same gradient-line format; GHR_K8=cell selects the fallback, read at every call so that a test can switch):
//   2 = cell-list form (k_render_bwd_cells, default): one wave per 4x4 cell from the forward pass's hit masks, records
//       gathered straight into LDS two chunks ahead;
//   0 = cell-group form (k_render_bwd, round 1): lane = pixel, 16-lane butterflies; the fallback where the cell-list
//       form's 32-bit byte offsets do not reach (b3_fits: >= 2^26 rows / instances).
int g_deterministic = 0;  // ghr_set_deterministic

// Which per-tile kernels take their tiles in k_tile_scan's heaviest-first order instead of the XCD-interleaved raster
// order (xcd_tile): bit 0 tile sort, bit 1 K7, bit 2 K8.  GHR_TILE_ORDER=<mask> overrides (A/B knob, like GHR_K8).
#ifndef GHR_TILE_ORDER_DEFAULT
#define GHR_TILE_ORDER_DEFAULT 3
#endif
const uint32_t* order_ptr(const uint32_t* p, int bit)
{
    const char* e = std::getenv("GHR_TILE_ORDER");
    const int mask = e ? std::atoi(e) : GHR_TILE_ORDER_DEFAULT;
    return (mask >> bit) & 1 ? p : nullptr;
}

int k8_variant()
{
    const char* e = std::getenv("GHR_K8");
    return (e && std::strcmp(e, "cell") == 0) ? 0 : 2;
}

int launch_k8(size_t rows, uint32_t T, hipStream_t s, int W, int H, int gx, uint32_t T_tiles, const uint32_t* tile_start,
               const uint32_t* point_list, const ghr::f4* rec, const float* bg, const float* final_T,
               const uint32_t* n_contrib, const float* dL_dpix, const ghr::rect4* rects, float* ginst, uint32_t cap,
               const unsigned long long* cell_mask, const uint32_t* cell_last, bool prezeroed,
               const uint32_t* tile_order)
{
    const dim3 grid(ghr::xcd_grid(T)), block(GHR_BLOCK);
    int v = k8_variant();
    if (g_deterministic) v = 2;  // the ordered walk exists in the cell-list form
    if (v == 2 && !ghr::b3_fits(rows, cap, (size_t)W, (size_t)H)) {
        if (g_deterministic) return -1;  // the caller reports it: determinism is never dropped silently
        v = 0;  // its 32-bit offsets
    }
    switch (v) {
    case 2:
        hipLaunchKernelGGL(ghr::k_render_bwd_cells, grid, dim3(GHR_B3_THREADS), 0, s, W, H, gx, T_tiles, tile_start, point_list, rec, bg,
                           final_T, n_contrib, dL_dpix, rects, ginst, cap, cell_mask, cell_last, g_deterministic,
                           prezeroed ? 1 : 0, tile_order);
        break;
    default:
        hipLaunchKernelGGL(ghr::k_render_bwd, grid, block, 0, s, W, H, gx, T_tiles, tile_start, point_list, rec, bg,
                           final_T, n_contrib, dL_dpix, rects, ginst, cap);
    }
    return 0;
}
#define GHR_E_DETERMINISTIC_MSG "ghr_set_deterministic(1) cannot be honoured at this size (the ordered walk needs 32-bit byte offsets: < 2^26 rows / instances)"

int fail(int code, const char* fmt, const char* detail = "")
{
    std::snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define GHR_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail(GHR_E_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

// Device-visible alias of the host word that receives num_rendered, when that word lives in pinned (hipHostMalloc /
// hipHostRegister) memory; nullptr for pageable memory, which then gets a stream-ordered 4-byte copy instead.
uint32_t* mapped_word(uint32_t* host)
{
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) {
        (void)hipGetLastError();  // not an error for the caller: fall back to the copy
        return nullptr;
    }
    return (uint32_t*)dev;
}

constexpr size_t ALIGN = 256;
inline size_t up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

// Sub-allocation of the three caller-owned workspaces (cf. obtain()/fromChunk, rasterizer_impl.h:21-73).
struct Geom {
    ghr::f4* rec;
    float* depths;
    ghr::rect4* rects;
    uint32_t* slot_blk;  // [ceil(P/256)]
    uint32_t* pos;       // [P][GHR_BIG_RECT]: place of each small-rect instance in its tile's list (count_tiles)
    float* cov3D;
};
struct Img {
    float* final_T;
    uint32_t* n_contrib;
    uint32_t* tile_count;  // [2][T]: per-tile instance counts of the small / the big rects; [1] then the big rects' append cursors
    uint32_t* tile_start;  // [T+1]
    uint32_t* small_cnt;   // [T]: instances of small rects per tile (k_tile_scan; k_scatter appends the big rects' behind them)
    uint32_t* R_dev;
    uint32_t* cell_last;   // [16 T]: largest n_contrib of each 4x4-pixel cell
    uint32_t* tile_order;  // [xcd_grid(T)]: tile of each workgroup of the per-tile kernels (k_tile_scan: heaviest first)
};
struct Bin {
    uint64_t* keys;
    uint32_t* point_list;
    unsigned long long* cell_mask;  // [mask_groups(R, T)][16]
    uint32_t* inst_line;            // [R]: gradient line (= list position) of every instance (numbered by rect4_slot)
};

size_t carve_geom(char* base, size_t P, bool mode_b, Geom* g)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += up(bytes); return p; };
    ghr::f4* rec = (ghr::f4*)take(P * 64);
    float* depths = (float*)take(P * 4);
    ghr::rect4* rects = (ghr::rect4*)take(P * 16);
    uint32_t* slot_blk = (uint32_t*)take(((P + GHR_BLOCK - 1) / GHR_BLOCK) * 4);
    uint32_t* pos = (uint32_t*)take(P * 4 * GHR_BIG_RECT);
    float* cov3D = mode_b ? (float*)take(P * 24) : nullptr;
    if (g) *g = Geom{rec, depths, rects, slot_blk, pos, cov3D};
    return off + ALIGN;
}
size_t carve_img(char* base, size_t N, size_t T, Img* im)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += up(bytes); return p; };
    float* final_T = (float*)take(N * 4);
    uint32_t* n_contrib = (uint32_t*)take(N * 4);
    uint32_t* tile_count = (uint32_t*)take(2 * T * 4);
    uint32_t* tile_start = (uint32_t*)take((T + 1) * 4);
    uint32_t* small_cnt = (uint32_t*)take(T * 4);
    uint32_t* R_dev = (uint32_t*)take(4);
    uint32_t* cell_last = (uint32_t*)take(T * 16 * 4);
    uint32_t* tile_order = (uint32_t*)take((size_t)ghr::xcd_grid((uint32_t)T) * 4);
    if (im) *im = Img{final_T, n_contrib, tile_count, tile_start, small_cnt, R_dev, cell_last, tile_order};
    return off + ALIGN;
}
size_t carve_bin(char* base, size_t R, size_t T, Bin* b)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += up(bytes); return p; };
    uint64_t* keys = (uint64_t*)take(R * 8);
    uint32_t* pl = (uint32_t*)take(R * 4);
    unsigned long long* cm = (unsigned long long*)take(ghr::mask_groups(R, T) * 16 * 8);
    uint32_t* il = (uint32_t*)take(R * 4);
    if (b) *b = Bin{keys, pl, cm, il};
    return off + ALIGN;
}
inline char* align_base(const void* p) { return (char*)(((uintptr_t)p + ALIGN - 1) / ALIGN * ALIGN); }

int check_dims(const ghr_view_args* a)
{
    if (!a) return fail(GHR_E_INVALID, "ghr_view_args is NULL");
    if (a->P < 0 || a->W <= 0 || a->H <= 0) return fail(GHR_E_INVALID, "bad P/W/H");
    if (a->C != GHR_NUM_CHANNELS) return fail(GHR_E_INVALID, "C must equal GHR_NUM_CHANNELS (10)");
    if ((a->W + GHR_TILE - 1) / GHR_TILE > 65535 || (a->H + GHR_TILE - 1) / GHR_TILE > 65535)
        return fail(GHR_E_INVALID, "image too large for 16-bit tile coordinates");
    return GHR_OK;
}

int check_view(const ghr_view_args* a)
{
    if (int rc = check_dims(a)) return rc;
    if (a->P == 0) return GHR_OK;
    if (!a->colors) return fail(GHR_E_NOCOLORS, "For non-RGB, provide precomputed Gaussian colors!");
    if (!a->means3D || !a->opacities || !a->background || !a->viewmatrix || !a->projmatrix)
        return fail(GHR_E_INVALID, "means3D/opacities/background/viewmatrix/projmatrix must be non-NULL");
    if (!a->conic_precomp && !a->cov3D_precomp && !(a->scales && a->rotations))
        return fail(GHR_E_INVALID, "kernel-geometry mode needs cov3D_precomp or scales+rotations");
    return GHR_OK;
}

int finish(hipStream_t s, int debug)
{
    GHR_HIP(hipGetLastError());
    if (debug) GHR_HIP(hipStreamSynchronize(s));
    return GHR_OK;
}

inline int grid_x(int W) { return (W + GHR_TILE - 1) / GHR_TILE; }

}  // namespace

extern "C" {

const char* ghr_last_error(void) { return g_err; }
int ghr_abi_version(void) { return GHR_ABI_VERSION; }

int ghr_forward_sizes(int32_t P, int32_t W, int32_t H, int32_t mode_b, size_t* geom_bytes, size_t* img_bytes)
{
    if (P < 0 || W <= 0 || H <= 0 || !geom_bytes || !img_bytes) return fail(GHR_E_INVALID, "ghr_forward_sizes: bad args");
    const size_t T = (size_t)grid_x(W) * grid_x(H);
    *geom_bytes = carve_geom(nullptr, (size_t)P, mode_b != 0, nullptr);
    *img_bytes = carve_img(nullptr, (size_t)W * H, T, nullptr);
    return GHR_OK;
}

int ghr_binning_size(uint32_t R, int32_t W, int32_t H, size_t* bin_bytes)
{
    if (!bin_bytes || W <= 0 || H <= 0) return fail(GHR_E_INVALID, "ghr_binning_size: bad args");
    *bin_bytes = carve_bin(nullptr, (size_t)R, (size_t)grid_x(W) * grid_x(H), nullptr);
    return GHR_OK;
}

int ghr_forward_stage1(void* stream, const ghr_view_args* a, void* geom_ws, void* img_ws, int32_t* radii,
                       uint32_t* R_host)
{
    if (int rc = check_view(a)) return rc;
    if (!R_host) return fail(GHR_E_INVALID, "R_host is NULL");
    hipStream_t s = (hipStream_t)stream;
    if (a->P == 0) { *R_host = 0; return GHR_OK; }
    if (!geom_ws || !img_ws || !radii) return fail(GHR_E_INVALID, "workspace/radii is NULL");
    const bool mode_b = a->conic_precomp == nullptr;
    const int gx = grid_x(a->W), gy = grid_x(a->H);
    const int T = gx * gy;
    Geom g; Img im;
    carve_geom(align_base(geom_ws), (size_t)a->P, mode_b, &g);
    carve_img(align_base(img_ws), (size_t)a->W * a->H, (size_t)T, &im);

    // (a recycled workspace -- ghr_view_args.img_ws_recycled -- has its counters at zero already: k_tile_sort left them there)
    if (!a->img_ws_recycled) GHR_HIP(hipMemsetAsync(im.tile_count, 0, sizeof(uint32_t) * 2 * (size_t)T, s));
    else if (a->debug) {
        std::vector<uint32_t> h(2 * (size_t)T);
        GHR_HIP(hipMemcpyAsync(h.data(), im.tile_count, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost, s));
        GHR_HIP(hipStreamSynchronize(s));
        for (uint32_t v : h)
            if (v != 0) return fail(GHR_E_INVALID, "ghr_view_args.img_ws_recycled is set but the workspace's per-tile counters are not zero");
    }
    ghr::PreArgs pa;
    pa.P = a->P; pa.W = a->W; pa.H = a->H; pa.gx = gx; pa.gy = gy;
    pa.means3D = a->means3D; pa.colors = a->colors; pa.opacities = a->opacities;
    pa.scales = a->scales; pa.rotations = a->rotations;
    pa.cov3D_precomp = a->cov3D_precomp; pa.conic_precomp = a->conic_precomp;
    pa.view = a->viewmatrix; pa.proj = a->projmatrix;
    pa.scale_modifier = a->scale_modifier; pa.tan_fovx = a->tan_fovx; pa.tan_fovy = a->tan_fovy;
    pa.focal_y = a->H / (2.0f * a->tan_fovy);  // rasterizer_impl.cu:224-225
    pa.focal_x = a->W / (2.0f * a->tan_fovx);
    pa.rec = g.rec; pa.depths = g.depths; pa.rects = g.rects; pa.cov3D = g.cov3D; pa.radii = radii;
    pa.tile_count = im.tile_count; pa.slot_blk = g.slot_blk; pa.pos = g.pos;
    hipLaunchKernelGGL(ghr::k_preprocess, dim3((a->P + GHR_BLOCK - 1) / GHR_BLOCK), dim3(GHR_BLOCK), 0, s, pa);
    uint32_t* R_mapped = mapped_word(R_host);
    hipLaunchKernelGGL(ghr::k_tile_scan, dim3(1), dim3(GHR_SCAN_BLOCK), 0, s, T, im.tile_count, im.small_cnt, im.tile_start, im.R_dev,
                       g.slot_blk, (a->P + GHR_BLOCK - 1) / GHR_BLOCK, R_mapped, im.tile_order);
    if (!R_mapped) GHR_HIP(hipMemcpyAsync(R_host, im.R_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return finish(s, a->debug);
}

int ghr_forward_stage2(void* stream, const ghr_view_args* a, uint32_t R, void* geom_ws, void* img_ws, void* bin_ws,
                       float* out_color, float* grad_scratch)
{
    if (int rc = check_dims(a)) return rc;  // stage 2 only reads P, W, H, C, background (+ debug)
    if (!out_color) return fail(GHR_E_INVALID, "out_color is NULL");
    if (a->P > 0 && !a->background) return fail(GHR_E_INVALID, "background is NULL");
    hipStream_t s = (hipStream_t)stream;
    const int gx = grid_x(a->W), gy = grid_x(a->H);
    const int T = gx * gy;
    if (a->P == 0) {
        // Nothing to splat: the reference skips the whole forward and returns the zero-filled image
        // (rasterize_points.cu:70,87); keep that.
        GHR_HIP(hipMemsetAsync(out_color, 0, sizeof(float) * (size_t)a->C * a->W * a->H, s));
        return finish(s, a->debug);
    }
    if (!geom_ws || !img_ws || (R > 0 && !bin_ws)) return fail(GHR_E_INVALID, "workspace is NULL");
    if (ghr::mask_groups((size_t)R, (size_t)T) * 128 >= ((size_t)1 << 32))
        return fail(GHR_E_INVALID, "too many instances for the 32-bit offsets of the cell masks");
    // the cov3D plane (mode B) lies behind everything stage 2 touches, so the carve is mode-independent here
    Geom g; Img im; Bin b;
    carve_geom(align_base(geom_ws), (size_t)a->P, false, &g);
    carve_img(align_base(img_ws), (size_t)a->W * a->H, (size_t)T, &im);
    carve_bin(bin_ws ? align_base(bin_ws) : nullptr, (size_t)R, (size_t)T, &b);
    if (R > 0) {
        // append cursors are 0 on entry: k_tile_scan leaves them there and k_tile_sort resets them (replay-safe)
        const int scatter_blocks = (a->P + 127) / 128;  // a wave serves 32 Gaussians
        hipLaunchKernelGGL(ghr::k_scatter, dim3(scatter_blocks), dim3(GHR_BLOCK), 0, s, a->P, gx,
                           g.rects, g.slot_blk, g.depths, im.tile_start, im.tile_count, (uint32_t)T, im.small_cnt, g.pos, b.keys, R);
        // dense scenes (long lists on average) first get their dense tiles sorted in big LDS blocks; the regular kernel
        // then passes those by.  R is the capacity here, an upper bound of the count: a guess that is too high only
        // costs an idle 3-us launch.
        const bool dense = (size_t)R >= (size_t)GHR_SORT_BIG_MIN_AVG * (size_t)T;
        if (dense) {
            // (round 6) lists of 1025 .. 4096 keys by 512-thread workgroups with the register-blocked network, longer ones
            // by the 1024-thread kernel; a workgroup looks at up to GHR_SORT_WALK_MAX tiles
            // by up to GHR_SORT_WALK_MAX entries of k_tile_scan's heaviest-first order (the dense tiles sit at its front: dealt
            // out evenly) or, without the order, of the raster order
            const uint32_t* order = order_ptr(im.tile_order, 0);
            const uint32_t order_len = ghr::xcd_grid((uint32_t)T);
            const unsigned walk = ((unsigned)(((order ? order_len : (uint32_t)T) + GHR_SORT_WALK_MAX - 1) / GHR_SORT_WALK_MAX) + 7u) & ~7u;
            uint32_t big_min = GHR_SORT_CAP;
            if (std::getenv("GHR_NO_SORT_MID") == nullptr) {
                // (every workgroup resident: 256 CUs x 8 resp. 4 workgroups)
                uint32_t lo = GHR_SORT_CAP;
                if (GHR_SORT_MID_SPLIT) {
                    hipLaunchKernelGGL(ghr::k_tile_sort_mid<256>, dim3(std::max(2048u, walk)), dim3(256), 0, s, (uint32_t)T,
                                       im.tile_start, b.keys, b.point_list, R, im.tile_count, g.rects, b.inst_line, gx, order,
                                       order_len, lo);
                    lo = 2048u;
                }
                hipLaunchKernelGGL(ghr::k_tile_sort_mid<512>, dim3(std::max(1024u, walk)), dim3(512), 0, s, (uint32_t)T,
                                   im.tile_start, b.keys, b.point_list, R, im.tile_count, g.rects, b.inst_line, gx, order,
                                   order_len, lo);
                big_min = GHR_SORT_MID_CAP;
            }
            hipLaunchKernelGGL(ghr::k_tile_sort_big, dim3(std::max(512u, walk)), dim3(GHR_SORT_BIG_BLOCK), 0, s, (uint32_t)T,
                               im.tile_start, b.keys, b.point_list, R, im.tile_count, g.rects, b.inst_line, gx, big_min, order,
                               order_len);
        }
        hipLaunchKernelGGL(ghr::k_tile_sort<1024>, dim3(ghr::xcd_grid((uint32_t)T)), dim3(GHR_SORT_BLOCK), 0, s, (uint32_t)T,
                           im.tile_start, b.keys, b.point_list, R, im.tile_count, g.rects, b.inst_line, gx,
                           order_ptr(im.tile_order, 0));
    }
    if (g_ev[0]) GHR_HIP(hipEventRecord(g_ev[0], s));
    hipLaunchKernelGGL(ghr::k_render_fwd, dim3(ghr::xcd_grid((uint32_t)T)), dim3(GHR_BLOCK), 0, s, a->W, a->H, gx,
                       (uint32_t)T, im.tile_start, b.point_list, g.rec, a->background, out_color, im.final_T,
                       im.n_contrib, R, b.cell_mask, im.cell_last, order_ptr(im.tile_order, 1), grad_scratch);
    if (g_ev[1]) GHR_HIP(hipEventRecord(g_ev[1], s));
    return finish(s, a->debug);
}

int ghr_backward(void* stream, const ghr_view_args* a, uint32_t R, const int32_t* radii, const void* geom_ws,
                 const void* img_ws, const void* bin_ws, const float* dL_dpix, float* grad_scratch,
                 float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D,
                 float* dL_dcov3D, float* dL_dscales, float* dL_drotations, int32_t prezeroed)
{
    return ghr_backward_ex(stream, a, R, radii, geom_ws, img_ws, bin_ws, dL_dpix, grad_scratch, dL_dmeans2D, dL_dconic,
                           dL_dopacity, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drotations, prezeroed, nullptr);
}

int ghr_backward_ex(void* stream, const ghr_view_args* a, uint32_t R, const int32_t* radii, const void* geom_ws,
                    const void* img_ws, const void* bin_ws, const float* dL_dpix, float* grad_scratch,
                    float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D,
                    float* dL_dcov3D, float* dL_dscales, float* dL_drotations, int32_t prezeroed, float* dL_dconic3)
{
    // backward reads colours / opacity from the packed records in geom_ws, not from `a`
    if (int rc = check_dims(a)) return rc;
    if (a->P > 0 && (!a->means3D || !a->viewmatrix || !a->projmatrix || !a->background))
        return fail(GHR_E_INVALID, "ghr_backward: means3D/viewmatrix/projmatrix/background must be non-NULL");
    if (a->P > 0 && !a->conic_precomp && !a->cov3D_precomp && !(a->scales && a->rotations))
        return fail(GHR_E_INVALID, "kernel-geometry mode needs cov3D_precomp or scales+rotations");
    hipStream_t s = (hipStream_t)stream;
    if (a->P == 0) return GHR_OK;
    if (!radii || !geom_ws || !img_ws || (R > 0 && !bin_ws) || !dL_dpix || (R > 0 && !grad_scratch) || !dL_dmeans2D ||
        !dL_dconic || !dL_dopacity || !dL_dcolors || !dL_dmeans3D || !dL_dcov3D || !dL_dscales || !dL_drotations)
        return fail(GHR_E_INVALID, "ghr_backward: NULL buffer");
    const bool mode_b = a->conic_precomp == nullptr;
    const int gx = grid_x(a->W), gy = grid_x(a->H);
    const int T = gx * gy;
    Geom g; Img im; Bin b;
    carve_geom(align_base(geom_ws), (size_t)a->P, mode_b, &g);
    carve_img(align_base(img_ws), (size_t)a->W * a->H, (size_t)T, &im);
    carve_bin(bin_ws ? align_base(bin_ws) : nullptr, (size_t)R, (size_t)T, &b);

    if (g_ev[2]) GHR_HIP(hipEventRecord(g_ev[2], s));
    if (R > 0 &&
        launch_k8((size_t)a->P, (uint32_t)T, s, a->W, a->H, gx, (uint32_t)T, (const uint32_t*)im.tile_start,
                  (const uint32_t*)b.point_list, (const ghr::f4*)g.rec, a->background, (const float*)im.final_T,
                  (const uint32_t*)im.n_contrib, dL_dpix, (const ghr::rect4*)g.rects, grad_scratch, R,
                  (const unsigned long long*)b.cell_mask, (const uint32_t*)im.cell_last, prezeroed != 0,
                  order_ptr((const uint32_t*)im.tile_order, 2)))
        return fail(GHR_E_INVALID, GHR_E_DETERMINISTIC_MSG);
    if (g_ev[3]) GHR_HIP(hipEventRecord(g_ev[3], s));
    ghr::GeomBwdArgs ga;
    ga.P = a->P; ga.means3D = a->means3D; ga.radii = radii; ga.scales = a->scales; ga.rotations = a->rotations;
    ga.cov3D = g.cov3D; ga.conic_precomp = a->conic_precomp; ga.view = a->viewmatrix; ga.proj = a->projmatrix;
    ga.scale_modifier = a->scale_modifier; ga.tan_fovx = a->tan_fovx; ga.tan_fovy = a->tan_fovy;
    ga.focal_y = a->H / (2.0f * a->tan_fovy);
    ga.focal_x = a->W / (2.0f * a->tan_fovx);
    ga.ginst = grad_scratch; ga.inst_line = b.inst_line; ga.ginst_rows = R;
    ga.rects = g.rects; ga.rec = g.rec; ga.half_w = 0.5f * a->W; ga.half_h = 0.5f * a->H;
    ga.dL_dmeans2D = dL_dmeans2D; ga.dL_dconic = dL_dconic; ga.dL_dconic3 = dL_dconic3; ga.dL_dopacity = dL_dopacity; ga.dL_dcolors = dL_dcolors;
    ga.dL_dmeans3D = dL_dmeans3D; ga.dL_dcov3D = dL_dcov3D; ga.dL_dscales = dL_dscales; ga.dL_drots = dL_drotations;
    hipLaunchKernelGGL(ghr::k_geom_bwd, dim3((a->P + GHR_BLOCK - 1) / GHR_BLOCK), dim3(GHR_BLOCK), 0, s, ga);
    return finish(s, a->debug);
}

namespace {
int fill_model(const ghr_model_args* m, ghr::ModelArgs* a)
{
    if (!m) return fail(GHR_E_INVALID, "ghr_model_args is NULL");
    if (m->P < 0 || m->W <= 0 || m->H <= 0) return fail(GHR_E_INVALID, "bad P/W/H");
    if (m->sh_degree < 0 || m->sh_degree > 3 || m->sh_coeffs < (m->sh_degree + 1) * (m->sh_degree + 1) ||
        m->sh_coeffs > GHR_SH_MAX)
        return fail(GHR_E_INVALID, "bad sh_degree / sh_coeffs");
    // K = (max_sh_degree + 1)^2 (include/ghr.h): the kernels' 16-B staging of features_rest counts on rows of >= 9 floats
    if (m->sh_coeffs != 1 && m->sh_coeffs != 4 && m->sh_coeffs != 9 && m->sh_coeffs != 16)
        return fail(GHR_E_INVALID, "ghr_model_args: sh_coeffs must be (max_sh_degree + 1)^2, i.e. 1, 4, 9 or 16");
    if (m->mode != 0 && m->mode != 1) return fail(GHR_E_INVALID, "ghr_model_args: mode must be 0 or 1");
    if (m->row0 < 0 || (m->row0 & (GHR_BLOCK - 1))) return fail(GHR_E_INVALID, "ghr_model_args: row0 must be a multiple of 256");
    const bool need_act = m->mode == 0;  // mode 1: opacity / label / confidence pointers are optional
    if (m->P > 0 && (!m->xyz || !m->log_scales || !m->rotations || (need_act && (!m->opacity_logit || !m->label_logit ||
                     !m->orient_conf_log)) || !m->features_dc || (m->sh_coeffs > 1 && !m->features_rest) ||
                     !m->viewmatrix || !m->projmatrix || !m->campos))
        return fail(GHR_E_INVALID, "ghr_model_args: NULL parameter tensor");
    a->P = m->P; a->W = m->W; a->H = m->H; a->gx = grid_x(m->W); a->gy = grid_x(m->H);
    a->sh_degree = m->sh_degree; a->sh_coeffs = m->sh_coeffs;
    a->mode = m->mode; a->row0 = m->row0;
    a->xyz = m->xyz; a->log_scales = m->log_scales; a->rotations = m->rotations;
    a->opacity_logit = m->opacity_logit; a->label_logit = m->label_logit; a->orient_conf_log = m->orient_conf_log;
    a->dir3d = m->mode == 1 ? m->dir3d : nullptr;
    a->const_opacity = m->const_opacity; a->const_label = m->const_label; a->const_conf = m->const_conf;
    a->features_dc = m->features_dc; a->features_rest = m->features_rest;
    a->view = m->viewmatrix; a->proj = m->projmatrix; a->campos = m->campos;
    a->scale_modifier = m->scale_modifier; a->tan_fovx = m->tan_fovx; a->tan_fovy = m->tan_fovy;
    a->focal_y = m->H / (2.0f * m->tan_fovy);
    a->focal_x = m->W / (2.0f * m->tan_fovx);
    a->conic_eps = m->conic_eps;
    if ((m->fovx_dev != nullptr) != (m->fovy_dev != nullptr)) return fail(GHR_E_INVALID, "ghr_model_args: fovx_dev and fovy_dev: both or neither");
    a->fovx = m->fovx_dev; a->fovy = m->fovy_dev;
    a->rec = nullptr; a->depths = nullptr; a->rects = nullptr; a->radii = nullptr; a->means2D = nullptr;
    a->tile_count = nullptr; a->slot_blk = nullptr; a->pos = nullptr;
    return GHR_OK;
}
inline int n_blocks(int rows) { return (rows + GHR_BLOCK - 1) / GHR_BLOCK; }
}  // namespace

int ghr_model_forward_segment(void* stream, const ghr_model_args* m, int32_t rows_total, int32_t first, void* geom_ws,
                              void* img_ws, int32_t* radii, float* means2D_out)
{
    ghr::ModelArgs a;
    if (int rc = fill_model(m, &a)) return rc;
    if (rows_total < 0 || (long long)a.row0 + a.P > rows_total) return fail(GHR_E_INVALID, "segment exceeds rows_total");
    hipStream_t s = (hipStream_t)stream;
    if (rows_total == 0) return GHR_OK;
    if (!geom_ws || !img_ws || !radii) return fail(GHR_E_INVALID, "workspace/radii is NULL");
    const int T = a.gx * a.gy;
    Geom g; Img im;
    carve_geom(align_base(geom_ws), (size_t)rows_total, false, &g);
    carve_img(align_base(img_ws), (size_t)a.W * a.H, (size_t)T, &im);
    // (a recycled workspace -- ghr_model_args.img_ws_recycled -- has its counters at zero already: k_tile_sort left them there)
    if (first && !m->img_ws_recycled) GHR_HIP(hipMemsetAsync(im.tile_count, 0, sizeof(uint32_t) * 2 * (size_t)T, s));
    if (first && m->img_ws_recycled && m->debug) {
        // the promise is otherwise taken on trust: under `debug` the counters are read back and must all be zero
        std::vector<uint32_t> h(2 * (size_t)T);
        GHR_HIP(hipMemcpyAsync(h.data(), im.tile_count, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost, s));
        GHR_HIP(hipStreamSynchronize(s));
        for (uint32_t v : h)
            if (v != 0)
                return fail(GHR_E_INVALID, "ghr_model_args.img_ws_recycled is set but the workspace's per-tile counters are not zero "
                                           "(was it through stage 1 AND stage 2 of a pass with P > 0 at the same W x H?)");
    }
    // rows between the end of this segment and the next multiple of 256 are padding: culled, no gradient slots
    const int end = a.row0 + a.P;
    const int pad_end = (int)std::min<long long>((long long)n_blocks(end) * GHR_BLOCK, rows_total);
    if (pad_end > end) {
        GHR_HIP(hipMemsetAsync(g.rects + end, 0, sizeof(ghr::rect4) * (size_t)(pad_end - end), s));
        GHR_HIP(hipMemsetAsync(radii + end, 0, sizeof(int32_t) * (size_t)(pad_end - end), s));
    }
    if (a.P == 0) return finish(s, m->debug);
    a.rec = g.rec; a.depths = g.depths; a.rects = g.rects; a.radii = radii; a.means2D = means2D_out;
    a.tile_count = im.tile_count; a.slot_blk = g.slot_blk; a.pos = g.pos;
    if (a.sh_coeffs > 1) hipLaunchKernelGGL(ghr::k_project<true>, dim3(n_blocks(a.P)), dim3(GHR_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(ghr::k_project<false>, dim3(n_blocks(a.P)), dim3(GHR_BLOCK), 0, s, a);
    return finish(s, m->debug);
}

int ghr_model_forward_finish(void* stream, int32_t rows_total, int32_t W, int32_t H, int32_t debug, void* geom_ws,
                             void* img_ws, uint32_t* R_host)
{
    if (rows_total < 0 || W <= 0 || H <= 0) return fail(GHR_E_INVALID, "bad rows_total/W/H");
    if (!R_host) return fail(GHR_E_INVALID, "R_host is NULL");
    hipStream_t s = (hipStream_t)stream;
    if (rows_total == 0) { *R_host = 0; return GHR_OK; }
    if (!geom_ws || !img_ws) return fail(GHR_E_INVALID, "workspace is NULL");
    const int T = grid_x(W) * grid_x(H);
    Geom g; Img im;
    carve_geom(align_base(geom_ws), (size_t)rows_total, false, &g);
    carve_img(align_base(img_ws), (size_t)W * H, (size_t)T, &im);
    uint32_t* R_mapped = mapped_word(R_host);
    hipLaunchKernelGGL(ghr::k_tile_scan, dim3(1), dim3(GHR_SCAN_BLOCK), 0, s, T, im.tile_count, im.small_cnt, im.tile_start, im.R_dev,
                       g.slot_blk, n_blocks(rows_total), R_mapped, im.tile_order);
    if (!R_mapped) GHR_HIP(hipMemcpyAsync(R_host, im.R_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return finish(s, debug);
}

int ghr_model_forward_stage1(void* stream, const ghr_model_args* m, void* geom_ws, void* img_ws, int32_t* radii,
                             float* means2D_out, uint32_t* R_host)
{
    if (!m) return fail(GHR_E_INVALID, "ghr_model_args is NULL");
    if (m->row0 != 0) return fail(GHR_E_INVALID, "ghr_model_forward_stage1: row0 must be 0 (use the segment calls)");
    if (!R_host) return fail(GHR_E_INVALID, "R_host is NULL");
    if (int rc = ghr_model_forward_segment(stream, m, m->P, 1, geom_ws, img_ws, radii, means2D_out)) return rc;
    return ghr_model_forward_finish(stream, m->P, m->W, m->H, m->debug, geom_ws, img_ws, R_host);
}

int ghr_render_backward(void* stream, int32_t rows_total, int32_t W, int32_t H, uint32_t R, const float* background,
                        const void* geom_ws, const void* img_ws, const void* bin_ws, const float* dL_dpix,
                        float* grad_scratch, int32_t prezeroed)
{
    if (rows_total < 0 || W <= 0 || H <= 0) return fail(GHR_E_INVALID, "bad rows_total/W/H");
    hipStream_t s = (hipStream_t)stream;
    if (rows_total == 0 || R == 0) return GHR_OK;
    if (!background || !geom_ws || !img_ws || !bin_ws || !dL_dpix || !grad_scratch)
        return fail(GHR_E_INVALID, "ghr_render_backward: NULL buffer");
    const int gx = grid_x(W), T = gx * grid_x(H);
    Geom g; Img im; Bin b;
    carve_geom(align_base(geom_ws), (size_t)rows_total, false, &g);
    carve_img(align_base(img_ws), (size_t)W * H, (size_t)T, &im);
    carve_bin(align_base(bin_ws), (size_t)R, (size_t)T, &b);
    if (g_ev[2]) GHR_HIP(hipEventRecord(g_ev[2], s));
    if (launch_k8((size_t)rows_total, (uint32_t)T, s, W, H, gx, (uint32_t)T, (const uint32_t*)im.tile_start,
                  (const uint32_t*)b.point_list, (const ghr::f4*)g.rec, background, (const float*)im.final_T,
                  (const uint32_t*)im.n_contrib, dL_dpix, (const ghr::rect4*)g.rects, grad_scratch, R,
                  (const unsigned long long*)b.cell_mask, (const uint32_t*)im.cell_last, prezeroed != 0,
                  order_ptr((const uint32_t*)im.tile_order, 2)))
        return fail(GHR_E_INVALID, GHR_E_DETERMINISTIC_MSG);
    if (g_ev[3]) GHR_HIP(hipEventRecord(g_ev[3], s));
    return finish(s, 0);
}

int ghr_model_backward_segment(void* stream, const ghr_model_args* m, int32_t rows_total, const int32_t* radii,
                               const void* geom_ws, const float* grad_scratch, float* d_means2D, float* d_xyz,
                               float* d_log_scales, float* d_rotations, float* d_opacity_logit, float* d_label_logit,
                               float* d_orient_conf_log, float* d_features_dc, float* d_features_rest, float* d_dir3d,
                               int32_t accumulate, int32_t* nan_flag, uint32_t grad_rows, const void* bin_ws,
                               uint32_t R)
{
    ghr::ModelArgs a;
    if (int rc = fill_model(m, &a)) return rc;
    if (rows_total < 0 || (long long)a.row0 + a.P > rows_total) return fail(GHR_E_INVALID, "segment exceeds rows_total");
    if (!bin_ws && R > 0) return fail(GHR_E_INVALID, "ghr_model_backward_segment: bin_ws is NULL");
    hipStream_t s = (hipStream_t)stream;
    if (a.P == 0) return GHR_OK;
    const bool need_act = a.mode == 0;
    const bool cam_only = m->cam_only != 0;
    if (cam_only && !m->cam_partial) return fail(GHR_E_INVALID, "ghr_model_backward_segment: cam_only without cam_partial");
    if (m->cam_partial && (m->cam_slot0 < 0 || (long long)m->cam_slot0 + ghr_camera_slots(a.P) > m->cam_slots))
        return fail(GHR_E_INVALID, "ghr_model_backward_segment: the segment's camera columns exceed cam_slots");
    if (!radii || !geom_ws)
        return fail(GHR_E_INVALID, "ghr_model_backward_segment: NULL buffer");
    const bool factored_sh = m->d_rgb != nullptr;  // ABI 19: the SH gradients leave as d_rgb, their own buffers may be NULL
    // (the SH gradient buffers may be NULL when nothing is stored there: the factored form, or the update carried by this
    // call -- ghr_adam_fuse -- without earlier views' gradients to add)
    const bool sh_unstored = factored_sh || (m->adam_fuse != nullptr && !accumulate);
    if (!cam_only && (!d_means2D || !d_xyz || !d_log_scales || !d_rotations || (!sh_unstored && !d_features_dc) ||
        (need_act && (!d_opacity_logit || !d_label_logit || !d_orient_conf_log)) ||
        (!sh_unstored && a.sh_coeffs > 1 && !d_features_rest)))
        return fail(GHR_E_INVALID, "ghr_model_backward_segment: NULL buffer");
    if (factored_sh && (cam_only || m->adam_fuse))
        return fail(GHR_E_INVALID, "ghr_model_backward_segment: d_rgb with cam_only / adam_fuse");
    Geom g;
    carve_geom(align_base(geom_ws), (size_t)rows_total, false, &g);
    a.radii = const_cast<int*>(radii);
    a.rects = g.rects;
    a.rec = g.rec;  // the gather unpacks the gradient lines with the pixel mean / conic / opacity k_project stored
    Bin b;
    carve_bin(bin_ws ? align_base(bin_ws) : nullptr, (size_t)R, (size_t)a.gx * a.gy, &b);
    ghr::ModelGrads mg;
    mg.inst_line = b.inst_line;
    mg.ginst = grad_scratch; mg.ginst_rows = grad_rows ? (grad_rows < R ? grad_rows : R) : R; mg.d_means2D = d_means2D; mg.d_xyz = d_xyz; mg.d_log_scales = d_log_scales;
    mg.d_rotations = d_rotations; mg.d_opacity_logit = d_opacity_logit; mg.d_label_logit = d_label_logit;
    mg.d_orient_conf_log = d_orient_conf_log; mg.d_features_dc = d_features_dc; mg.d_features_rest = d_features_rest;
    mg.d_rgb = m->d_rgb;
    mg.d_dir3d = a.mode == 1 ? d_dir3d : nullptr;
    mg.accumulate = accumulate; mg.nan_flag = cam_only ? nullptr : nan_flag;
    mg.cam_partial = m->cam_partial; mg.cam_slot0 = (uint32_t)m->cam_slot0; mg.cam_stride = (uint32_t)m->cam_slots;
    mg.cam_only = cam_only ? 1 : 0; mg.detach_means2D = m->detach_means2D != 0 ? 1 : 0;
    const int n_dens = (m->dens_grad_accum != nullptr) + (m->dens_denom != nullptr) + (m->dens_max_radii2D != nullptr);
    if (n_dens != 0 && n_dens != 3)
        return fail(GHR_E_INVALID, "ghr_model_backward_segment: dens_grad_accum / dens_denom / dens_max_radii2D: all three or none");
    mg.dens_grad_accum = m->dens_grad_accum; mg.dens_denom = m->dens_denom; mg.dens_max_radii = m->dens_max_radii2D;
    mg.dens_count = nullptr; mg.dens_cap = R;
    if (n_dens == 3 && m->dens_img_ws) {
        Img im;
        carve_img(align_base(m->dens_img_ws), (size_t)a.W * a.H, (size_t)a.gx * a.gy, &im);
        mg.dens_count = im.R_dev;
    }
    mg.overflow_is_bad = 0;
    mg.adam.on = 0;
    const ghr_adam_fuse* af = m->adam_fuse;
    if (af) {
        if (cam_only) return fail(GHR_E_INVALID, "ghr_adam_fuse: not with a cam_only segment");
        if (a.mode == 1 && accumulate) return fail(GHR_E_INVALID, "ghr_adam_fuse: a strand segment carries the update only as the step's single view");
        if (af->n <= 0 || !af->p_in || !af->m_in || !af->v_in || !af->p_out || !af->m_out || !af->v_out || !af->state || !af->flag ||
            !af->flag_next || af->n_groups <= 0 || af->n_groups > GHR_ADAM_MAX_GROUPS || !af->group_end_host || !af->lr_host)
            return fail(GHR_E_INVALID, "ghr_adam_fuse: NULL buffer / bad group table");
        if (nan_flag != af->flag) return fail(GHR_E_INVALID, "ghr_adam_fuse: nan_flag of the backward call must be adam_fuse->flag");
        const float* arrays[GHR_ADAM_FUSE_ARRAYS] = {a.xyz, a.log_scales, a.rotations, a.opacity_logit, a.label_logit,
                                                     a.orient_conf_log, a.features_dc, a.features_rest};
        // (mode 1, a strand segment: only the SH features are raw parameters of the optimizer; the other groups of its flat
        // buffer -- strand directions, confidence -- get their gradients through autograd and are stepped by the caller)
        const long long w6 = a.mode == 1 ? 0 : 1;
        const long long width[GHR_ADAM_FUSE_ARRAYS] = {3 * w6, 3 * w6, 4 * w6, w6, w6, w6, 3, 3LL * (a.sh_coeffs - 1)};
        long long covered = 0;
        for (int k = 0; k < GHR_ADAM_FUSE_ARRAYS; k++) {
            const long long len = width[k] * a.P;
            if (len == 0) { mg.adam.lr[k] = 0.f; mg.adam.group[k] = 0; continue; }
            const long long off = arrays[k] - af->p_in;
            if (off < 0 || off + len > af->n)
                return fail(GHR_E_INVALID, "ghr_adam_fuse: a raw-parameter array does not lie inside p_in");
            int gi = 0;
            while (gi < af->n_groups - 1 && off >= af->group_end_host[gi]) gi++;
            if (off + len > af->group_end_host[gi])
                return fail(GHR_E_INVALID, "ghr_adam_fuse: a raw-parameter array straddles two parameter groups");
            mg.adam.group[k] = gi;
            mg.adam.lr[k] = af->lr_host[gi];
            covered += len;
        }
        if (a.mode == 0 && covered != af->n)
            return fail(GHR_E_INVALID, "ghr_adam_fuse: the eight raw-parameter arrays must tile p_in (n floats)");
        mg.adam.p_base = af->p_in; mg.adam.m_in = af->m_in; mg.adam.v_in = af->v_in;
        mg.adam.p_out = af->p_out; mg.adam.m_out = af->m_out; mg.adam.v_out = af->v_out;
        mg.adam.state = af->state; mg.adam.beta1 = af->beta1; mg.adam.beta2 = af->beta2; mg.adam.eps = af->eps;
        mg.adam.on = 1;
    }
    if (m->dens_img_ws && (af || m->overflow_raises_flag)) {
        // (steps with the fused optimizer update: EVERY view's backward checks its instance count and raises the step's flag)
        Img im;
        carve_img(align_base(m->dens_img_ws), (size_t)a.W * a.H, (size_t)a.gx * a.gy, &im);
        mg.dens_count = im.R_dev; mg.dens_cap = R; mg.overflow_is_bad = 1;
    }
    const dim3 grid((a.P + GHR_PBW_BLOCK - 1) / GHR_PBW_BLOCK), block(GHR_PBW_BLOCK);
    if (af) {
        if (mg.cam_partial) hipLaunchKernelGGL((ghr::k_project_bwd<true, true>), grid, block, 0, s, a, mg);
        else hipLaunchKernelGGL((ghr::k_project_bwd<false, true>), grid, block, 0, s, a, mg);
        // (a strand segment leaves the finish to the caller -- ghr_adam_fused_finish -- who first steps the groups whose
        // gradients are still on their way through autograd and adds their non-finite mark to the step's flag)
        if (a.mode == 0)
            hipLaunchKernelGGL(ghr::k_adam_fused_finish, dim3(1024), dim3(256), 0, s, (long long)af->n, af->p_in, af->m_in,
                               af->v_in, af->p_out, af->m_out, af->v_out, af->state, (const int*)af->flag, af->flag_next);
    } else if (mg.cam_partial) hipLaunchKernelGGL((ghr::k_project_bwd<true, false>), grid, block, 0, s, a, mg);
    else hipLaunchKernelGGL((ghr::k_project_bwd<false, false>), grid, block, 0, s, a, mg);
    return finish(s, m->debug);
}

int ghr_adam_fused_finish(void* stream, const ghr_adam_fuse* af)
{
    if (!af || af->n < 0 || !af->p_in || !af->m_in || !af->v_in || !af->p_out || !af->m_out || !af->v_out || !af->state ||
        !af->flag || !af->flag_next)
        return fail(GHR_E_INVALID, "ghr_adam_fused_finish: bad ghr_adam_fuse");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ghr::k_adam_fused_finish, dim3(1024), dim3(256), 0, s, (long long)af->n, af->p_in, af->m_in, af->v_in,
                       af->p_out, af->m_out, af->v_out, af->state, (const int*)af->flag, af->flag_next);
    return finish(s, 0);
}

int32_t ghr_camera_slots(int32_t P) { return P > 0 ? (P + GHR_PBW_BLOCK - 1) / GHR_PBW_BLOCK : 0; }

int ghr_camera_grad_fold(void* stream, const float* cam_partial, int32_t cam_slots, float* d_cam, const float* fovx_dev,
                         const float* fovy_dev)
{
    if (!d_cam || cam_slots < 0 || (cam_slots > 0 && !cam_partial) || ((fovx_dev != nullptr) != (fovy_dev != nullptr)))
        return fail(GHR_E_INVALID, "ghr_camera_grad_fold: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (cam_slots == 0) {
        GHR_HIP(hipMemsetAsync(d_cam, 0, sizeof(float) * GHR_CAM_GRADS, s));
        return finish(s, 0);
    }
    hipLaunchKernelGGL(ghr::k_cam_fold, dim3(GHR_CAM_PARTIALS), dim3(GHR_CAM_FOLD_BLOCK), 0, s, cam_partial, (uint32_t)cam_slots,
                       d_cam, fovx_dev, fovy_dev);
    return finish(s, 0);
}

int ghr_sh_grad_from_views(void* stream, int32_t P, int32_t sh_degree, int32_t sh_coeffs, const float* xyz, int32_t n_views,
                           const float* campos, int64_t campos_stride, const float* g_views, int64_t view_stride,
                           float* d_features_dc, float* d_features_rest, int32_t accumulate, int32_t* nan_flag,
                           int64_t flag_offset)
{
    if (P < 0 || n_views < 0 || sh_degree < 0 || sh_degree > 3 || view_stride < 0 || campos_stride < 0 ||
        (nan_flag != nullptr && (flag_offset < 0 || (n_views > 1 && flag_offset >= view_stride))) ||
        !(sh_coeffs == 1 || sh_coeffs == 4 || sh_coeffs == 9 || sh_coeffs == 16) || (sh_degree + 1) * (sh_degree + 1) > sh_coeffs)
        return fail(GHR_E_INVALID, "ghr_sh_grad_from_views: bad sizes");
    if (P == 0) return GHR_OK;
    if (!xyz || !d_features_dc || (sh_coeffs > 1 && !d_features_rest) || (n_views > 0 && (!campos || !g_views)) ||
        (n_views > 1 && view_stride < 3 * (int64_t)P))
        return fail(GHR_E_INVALID, "ghr_sh_grad_from_views: NULL buffer / overlapping views");
    hipStream_t s = (hipStream_t)stream;
    ghr::ShViewsArgs a;
    a.P = P; a.sh_degree = sh_degree; a.sh_coeffs = sh_coeffs; a.n_views = n_views; a.xyz = xyz; a.campos = campos;
    a.campos_stride = (size_t)campos_stride; a.nan_flag = nan_flag; a.flag_offset = flag_offset;
    a.g = g_views; a.view_stride = (size_t)view_stride; a.d_dc = d_features_dc; a.d_rest = d_features_rest;
    a.accumulate = accumulate != 0;
    hipLaunchKernelGGL(ghr::k_sh_grad_from_views, dim3((P + GHR_PBW_BLOCK - 1) / GHR_PBW_BLOCK), dim3(GHR_PBW_BLOCK), 0, s, a);
    return finish(s, 0);
}

static_assert(GHR_STRAND_MAX_SEG * 24 <= 48 * 1024, "one strand's two LDS rows");

int ghr_strand_build(void* stream, int32_t S, int32_t n_seg, const float* origins, const float* dirs, float scale, float* xyz,
                     float* rotation, float* scaling)
{
    if (S < 0 || n_seg < 0 || n_seg > GHR_STRAND_MAX_SEG) return fail(GHR_E_INVALID, "ghr_strand_build: bad strand shape");
    if (S == 0 || n_seg == 0) return GHR_OK;
    if ((int64_t)S * n_seg > (int64_t)INT32_MAX / 4) return fail(GHR_E_INVALID, "ghr_strand_build: too many segments");
    if (!origins || !dirs || !xyz || !rotation || !scaling) return fail(GHR_E_INVALID, "ghr_strand_build: NULL buffer");
    hipStream_t s = (hipStream_t)stream;
    ghr::StrandArgs a;
    a.S = S; a.n_seg = n_seg; a.spb = ghr::strands_per_block(n_seg);
    a.origins = origins; a.dirs = dirs; a.scale = scale; a.xyz = xyz; a.rot = rotation; a.scaling = scaling;
    const size_t lds = (size_t)2 * a.spb * n_seg * 3 * sizeof(float);
    hipLaunchKernelGGL(ghr::k_strand_build, dim3((S + a.spb - 1) / a.spb), dim3(GHR_STRAND_BLOCK), lds, s, a);
    return finish(s, 0);
}

int ghr_strand_build_backward(void* stream, int32_t S, int32_t n_seg, const float* dirs, const float* d_xyz,
                              const float* d_rotation, const float* d_scaling, float* d_dirs)
{
    return ghr_strand_build_backward_ex(stream, S, n_seg, dirs, d_xyz, d_rotation, d_scaling, nullptr, d_dirs);
}

int ghr_strand_build_backward_ex(void* stream, int32_t S, int32_t n_seg, const float* dirs, const float* d_xyz,
                                 const float* d_rotation, const float* d_scaling, const float* d_dir_rows, float* d_dirs)
{
    if (S < 0 || n_seg < 0 || n_seg > GHR_STRAND_MAX_SEG) return fail(GHR_E_INVALID, "ghr_strand_build_backward: bad strand shape");
    if (S == 0 || n_seg == 0) return GHR_OK;
    if ((int64_t)S * n_seg > (int64_t)INT32_MAX / 4) return fail(GHR_E_INVALID, "ghr_strand_build_backward: too many segments");
    if (!dirs || !d_dirs) return fail(GHR_E_INVALID, "ghr_strand_build_backward: NULL buffer");
    hipStream_t s = (hipStream_t)stream;
    ghr::StrandBwdArgs a;
    a.S = S; a.n_seg = n_seg; a.spb = ghr::strands_per_block(n_seg);
    a.dirs = dirs; a.d_xyz = d_xyz; a.d_rot = d_rotation; a.d_scaling = d_scaling; a.d_dir_rows = d_dir_rows; a.d_dirs = d_dirs;
    const size_t lds = (size_t)2 * a.spb * n_seg * 3 * sizeof(float);
    hipLaunchKernelGGL(ghr::k_strand_build_bwd, dim3((S + a.spb - 1) / a.spb), dim3(GHR_STRAND_BLOCK), lds, s, a);
    return finish(s, 0);
}

int ghr_model_backward(void* stream, const ghr_model_args* m, uint32_t R, const int32_t* radii, const void* geom_ws,
                       const void* img_ws, const void* bin_ws, const float* dL_dpix, float* grad_scratch,
                       float* d_means2D, float* d_xyz, float* d_log_scales, float* d_rotations,
                       float* d_opacity_logit, float* d_label_logit, float* d_orient_conf_log, float* d_features_dc,
                       float* d_features_rest, int32_t accumulate, int32_t* nan_flag, int32_t prezeroed)
{
    if (!m) return fail(GHR_E_INVALID, "ghr_model_args is NULL");
    if (m->row0 != 0) return fail(GHR_E_INVALID, "ghr_model_backward: row0 must be 0 (use the segment calls)");
    if (m->P == 0) return GHR_OK;
    if (!dL_dpix || (R > 0 && (!grad_scratch || !bin_ws)) || !img_ws || !m->background)
        return fail(GHR_E_INVALID, "ghr_model_backward: NULL buffer");
    if (int rc = ghr_render_backward(stream, m->P, m->W, m->H, R, m->background, geom_ws, img_ws, bin_ws, dL_dpix,
                                     grad_scratch, prezeroed))
        return rc;
    return ghr_model_backward_segment(stream, m, m->P, radii, geom_ws, grad_scratch, d_means2D, d_xyz, d_log_scales,
                                      d_rotations, d_opacity_logit, d_label_logit, d_orient_conf_log, d_features_dc,
                                      d_features_rest, nullptr, accumulate, nan_flag, R, bin_ws, R);
}

namespace ghr {
// One 1024-thread workgroup folds the 5 x n_slots partial sums the forward kernel's workgroups stored (ghr_loss.h; laid out
// [term][slot]): thread t takes the slots t, t + 1024, ... of every term (all of a round's loads in flight together), a DPP
// sum inside each wave, and one thread per term adds the sixteen wave totals in double -- a fixed order, so the loss value
// does not depend on how the forward kernel was scheduled.  (Built for latency: the kernel is a 5-us stop between the loss
// forward and backward passes; a first form with double-precision butterflies took 12-17 us.)
__global__ void __launch_bounds__(1024) k_loss_finalize(const float* slots, uint32_t n_slots, float w_l1, float w_ssim,
                                                        float w_mask, float w_orient, float n_pix, float* aux, float* out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float s_part[GHR_LOSS_TERMS][16];
    __shared__ double s_tot[GHR_LOSS_TERMS];
    float s[GHR_LOSS_TERMS] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (uint32_t base = 0; base < n_slots; base += 8u * 1024u) {
        float v[GHR_LOSS_TERMS][8];
#pragma unroll
        for (int k = 0; k < GHR_LOSS_TERMS; k++)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t i = base + 1024u * u + threadIdx.x;
                v[k][u] = i < n_slots ? slots[(size_t)k * n_slots + i] : 0.f;
            }
#pragma unroll
        for (int k = 0; k < GHR_LOSS_TERMS; k++)
            s[k] += ((v[k][0] + v[k][1]) + (v[k][2] + v[k][3])) + ((v[k][4] + v[k][5]) + (v[k][6] + v[k][7]));
    }
#pragma unroll
    for (int k = 0; k < GHR_LOSS_TERMS; k++) {
        const float w = wave_sum(s[k]);
        if ((threadIdx.x & 63) == 0) s_part[k][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x < GHR_LOSS_TERMS) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) t += (double)s_part[threadIdx.x][w];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s0 = s_tot[0], s1 = s_tot[1], s2 = s_tot[2], s3 = s_tot[3], s4 = s_tot[4];
        float lo = 0.f, bad = 0.f;
        if (w_orient != 0.f) {
            lo = (float)(s3 / s4);
            if (lo != lo) { lo = 0.f; bad = 1.f; }  // train_gaussians.py:134: a NaN orientation loss is dropped
        }
        aux[0] = (float)s4;
        aux[1] = bad;
        out[0] = (float)(w_l1 * (s0 / (3.0 * n_pix)) + w_ssim * (1.0 - s1 / (3.0 * n_pix)) + w_mask * (s2 / (2.0 * n_pix))) +
                 w_orient * lo;
    }
#endif
}
}  // namespace ghr

#ifndef GHR_ADAM_BLOCKS
#define GHR_ADAM_BLOCKS 65536
#endif
// The marching form of the loss kernels (ghr_loss.h) needs 16-B aligned image rows
// Rows of a strip per wave.  Measured on 1080p (profiles/r03h): 16 .. 40 rows give the same kernel time, 64 is 6 % slower,
// 128 10 %, 272 60 % -- the kernels live on the number of waves in flight, the ten extra rows a segment filters for its
// first output row are cheap
static int loss_march_seg(const ghr_loss_args* l, bool backward)
{
    const char* e = std::getenv(backward ? "GHR_LOSS_SEG_B" : "GHR_LOSS_SEG_F");  // measurement knob
    int seg = e ? atoi(e) : 0;
    if (seg <= 0) seg = 32;
    (void)l;
    return (seg + GHR_LM_ROWS - 1) / GHR_LM_ROWS * GHR_LM_ROWS;
}
static dim3 loss_march_grid(const ghr_loss_args* l, int seg)
{
    return dim3(8 * (((l->W + GHR_LM_TW - 1) / GHR_LM_TW + 7) / 8), (l->H + seg - 1) / seg, 3);
}
static bool loss_vec_ok(const ghr_loss_args* l, const void* p0 = nullptr, const void* p1 = nullptr)
{
    const bool off = std::getenv("GHR_LOSS_SCALAR") != nullptr;  // test / measurement knob (read per call): the tile kernels
    if (off || (l->W & 3) || (size_t)l->W * (size_t)l->H >= ((size_t)1 << 30)) return false;
    const void* ps[] = {l->image, l->gt_image, l->gt_mask, l->gt_stats, p0, p1};
    for (const void* p : ps)
        if (((uintptr_t)p & 15u) != 0) return false;
    return true;
}

size_t ghr_loss_sums_floats(int32_t W, int32_t H)
{
    if (W <= 0 || H <= 0) return 0;
    // the larger of the two kernel forms' slot counts (marching form at its shortest segment, GHR_LM_ROWS rows)
    const size_t n = std::max(ghr::loss_slots_tile(W, H), ghr::loss_slots_march(W, H, GHR_LM_ROWS));
    return GHR_LOSS_AUX + GHR_LOSS_TERMS * n;
}

int ghr_loss_forward(void* stream, const ghr_loss_args* l, float* maps, float* sums, float* loss_out)
{
    if (!l || l->W <= 0 || l->H <= 0 || !l->image || !l->mask || !l->gt_image || !l->gt_mask || !maps || !sums || !loss_out)
        return fail(GHR_E_INVALID, "ghr_loss_forward: bad args");
    const bool orient = l->w_orient != 0.f;
    if (orient && (!l->dir2d || !l->orient_conf || !l->gt_orient_angle || !l->gt_orient_conf))
        return fail(GHR_E_INVALID, "ghr_loss_forward: w_orient != 0 needs dir2d / orient_conf / gt_orient_angle / gt_orient_conf");
    hipStream_t s = (hipStream_t)stream;
    // sums = {aux[GHR_LOSS_AUX] | one slot of five partial sums per workgroup of the forward kernel}: nothing to zero
    ghr::LossArgs a{l->W, l->H, l->image, l->mask, orient ? l->dir2d : nullptr, l->orient_conf, l->gt_image, l->gt_mask,
                    l->gt_orient_angle, l->gt_orient_conf, l->unmasked_colours ? 0 : 1, maps, sums + GHR_LOSS_AUX, l->gt_stats,
                    nullptr, loss_march_seg(l, false), 0u};
    const dim3 grid((l->W + GHR_L_TW - 1) / GHR_L_TW, (l->H + GHR_L_TH - 1) / GHR_L_TH, 3);
    const bool vec = loss_vec_ok(l, maps);
    const dim3 grid_v = loss_march_grid(l, a.seg);
    const size_t n_slots = vec ? ghr::loss_slots_march(l->W, l->H, a.seg) : ghr::loss_slots_tile(l->W, l->H);
    a.n_slots = (uint32_t)n_slots;
    if (l->gt_stats) {
        if (vec) hipLaunchKernelGGL(ghr::k_loss_fwd_cached_v, grid_v, dim3(64), 0, s, a);
        else hipLaunchKernelGGL(ghr::k_loss_fwd_cached, grid, dim3(256), 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL(ghr::k_loss_fwd_v, grid_v, dim3(64), 0, s, a);
        else hipLaunchKernelGGL(ghr::k_loss_fwd, grid, dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(ghr::k_loss_finalize, dim3(1), dim3(1024), 0, s, sums + GHR_LOSS_AUX, (uint32_t)n_slots, l->w_l1,
                       l->w_ssim, l->w_mask, orient ? l->w_orient : 0.f, (float)l->W * (float)l->H, sums, loss_out);
    return finish(s, 0);
}

int ghr_loss_gt_stats(void* stream, const ghr_loss_args* l, float* stats_out)
{
    if (!l || l->W <= 0 || l->H <= 0 || !l->gt_image || !l->gt_mask || !stats_out)
        return fail(GHR_E_INVALID, "ghr_loss_gt_stats: bad args");
    hipStream_t s = (hipStream_t)stream;
    ghr::LossArgs a{l->W, l->H, l->gt_image, nullptr, nullptr, nullptr, l->gt_image, l->gt_mask, nullptr, nullptr,
                    l->unmasked_colours ? 0 : 1, nullptr, nullptr, nullptr, stats_out, loss_march_seg(l, false), 0u};
    const dim3 grid((l->W + GHR_L_TW - 1) / GHR_L_TW, (l->H + GHR_L_TH - 1) / GHR_L_TH, 3);
    if (loss_vec_ok(l, stats_out)) hipLaunchKernelGGL(ghr::k_loss_gt_stats_v, loss_march_grid(l, a.seg), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(ghr::k_loss_gt_stats, grid, dim3(256), 0, s, a);
    return finish(s, 0);
}

int ghr_loss_backward(void* stream, const ghr_loss_args* l, const float* maps, const float* sums,
                      const float* grad_loss, float* d_image, float* d_mask, float* d_dir2d, float* d_orient_conf,
                      float* zero_plane_a, float* zero_plane_b)
{
    if (!l || l->W <= 0 || l->H <= 0 || !l->image || !l->mask || !l->gt_image || !l->gt_mask || !maps || !sums ||
        !d_image || !d_mask || ((d_dir2d == nullptr) != (d_orient_conf == nullptr)))
        return fail(GHR_E_INVALID, "ghr_loss_backward: bad args");
    const bool orient = l->w_orient != 0.f;
    if (orient && (!l->dir2d || !l->orient_conf || !l->gt_orient_angle || !l->gt_orient_conf || !d_dir2d))
        return fail(GHR_E_INVALID, "ghr_loss_backward: w_orient != 0 needs the orientation inputs and d_dir2d / d_orient_conf");
    hipStream_t s = (hipStream_t)stream;
    ghr::LossBwdArgs a{l->W, l->H, l->image, l->mask, orient ? l->dir2d : nullptr, l->orient_conf, l->gt_image,
                       l->gt_mask, l->gt_orient_angle, l->gt_orient_conf, l->unmasked_colours ? 0 : 1, maps,
                       sums,  // aux: {sum of the orientation weights, NaN flag} (k_loss_finalize)
                       grad_loss, l->w_l1, l->w_ssim, l->w_mask, orient ? l->w_orient : 0.f, d_image, d_mask, d_dir2d,
                       d_orient_conf, zero_plane_a, zero_plane_b, loss_march_seg(l, true)};
    const dim3 grid((l->W + GHR_L_TW - 1) / GHR_L_TW, (l->H + GHR_L_TH - 1) / GHR_L_TH, 3);
    if (loss_vec_ok(l, maps)) hipLaunchKernelGGL(ghr::k_loss_bwd_v, loss_march_grid(l, a.seg), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(ghr::k_loss_bwd, grid, dim3(256), 0, s, a);
    return finish(s, 0);
}

namespace {
int adam_step_range(void* stream, int64_t n, int64_t begin, int64_t count, const float* p_in, const float* m_in,
                    const float* v_in, float* p, float* g, float* m, float* v, int32_t* state, int32_t* flag, int32_t nan_mark,
                    int32_t n_groups, const int64_t* group_end_host, const float* lr_host, double beta1, double beta2,
                    float eps, int32_t nan_guard, int32_t zero_grad, int32_t last, uint32_t skip_mask);
}

int ghr_adam_step_range(void* stream, int64_t n, int64_t begin, int64_t count, float* p, float* g, float* m, float* v,
                        int32_t* state, int32_t n_groups, const int64_t* group_end_host, const float* lr_host,
                        double beta1, double beta2, float eps, int32_t nan_guard, int32_t zero_grad, int32_t last,
                        uint32_t skip_mask)
{
    return adam_step_range(stream, n, begin, count, nullptr, nullptr, nullptr, p, g, m, v, state, nullptr, 0, n_groups,
                           group_end_host, lr_host, beta1, beta2, eps, nan_guard, zero_grad, last, skip_mask);
}

int ghr_adam_step_range_to(void* stream, int64_t n, int64_t begin, int64_t count, const float* p_in, const float* m_in,
                           const float* v_in, float* p_out, float* g, float* m_out, float* v_out, int32_t* state,
                           int32_t* flag, int32_t nan_mark, int32_t n_groups, const int64_t* group_end_host,
                           const float* lr_host, double beta1, double beta2, float eps, int32_t zero_grad, uint32_t skip_mask)
{
    if (!p_in || !m_in || !v_in) return fail(GHR_E_INVALID, "ghr_adam_step_range_to: NULL input buffer");
    if (nan_mark && !flag) return fail(GHR_E_INVALID, "ghr_adam_step_range_to: nan_mark needs the flag word");
    if (p_in == p_out || m_in == m_out || v_in == v_out)
        return fail(GHR_E_INVALID, "ghr_adam_step_range_to: in and out buffers must differ (ghr_adam_step_range updates in place)");
    // (nan_guard 2: whoever produced the gradients keeps the flag; last 0: the caller's own finish advances the counter)
    return adam_step_range(stream, n, begin, count, p_in, m_in, v_in, p_out, g, m_out, v_out, state, flag, nan_mark, n_groups,
                           group_end_host, lr_host, beta1, beta2, eps, 2, zero_grad, 0, skip_mask);
}

namespace {
int adam_step_range(void* stream, int64_t n, int64_t begin, int64_t count, const float* p_in, const float* m_in,
                    const float* v_in, float* p, float* g, float* m, float* v, int32_t* state, int32_t* flag, int32_t nan_mark,
                    int32_t n_groups, const int64_t* group_end_host, const float* lr_host, double beta1, double beta2,
                    float eps, int32_t nan_guard, int32_t zero_grad, int32_t last, uint32_t skip_mask)
{
    if (n < 0 || begin < 0 || count < 0 || begin + count > n || !p || !g || !m || !v || !state || n_groups <= 0 ||
        n_groups > GHR_ADAM_MAX_GROUPS || !group_end_host || !lr_host)
        return fail(GHR_E_INVALID, "ghr_adam_step: bad args");
    if (nan_guard == 1 && (begin != 0 || count != n))
        return fail(GHR_E_INVALID, "ghr_adam_step_range: the scanning NaN guard needs the whole buffer (use 0 or 2)");
    hipStream_t s = (hipStream_t)stream;
    if (count > 0) {
        ghr::AdamArgs a;
        a.begin = begin; a.n = begin + count; a.p = p; a.g = g; a.m = m; a.v = v; a.state = state; a.n_groups = n_groups;
        for (int i = 0; i < n_groups; i++) { a.end[i] = group_end_host[i]; a.lr[i] = lr_host[i]; }
        a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.zero_grad = zero_grad; a.skip_mask = skip_mask;
        a.p_in = p_in; a.m_in = m_in; a.v_in = v_in; a.flag = flag; a.nan_mark = nan_mark ? flag : nullptr;
        const int blocks = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
        if (nan_guard == 1)
            hipLaunchKernelGGL(ghr::k_adam_nan_flag, dim3(blocks), dim3(256), 0, s, g, (long long)n, state);
        // four elements per thread where the range and the buffers allow 16-B accesses (GHR_ADAM_SCALAR: the scalar kernel)
        const bool v4 = (begin & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v) |
                                              ((uintptr_t)p_in) | ((uintptr_t)m_in) | ((uintptr_t)v_in)) & 15u) == 0 &&
                        std::getenv("GHR_ADAM_SCALAR") == nullptr;
        if (v4) {
            const long long n4 = (count + 3) / 4;
            const long long cap4 = GHR_ADAM_BLOCKS;
            const int blocks4 = (int)((n4 + 255) / 256 < cap4 ? (n4 + 255) / 256 : cap4);
            hipLaunchKernelGGL(ghr::k_adam_v4, dim3(blocks4 > 0 ? blocks4 : 1), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL(ghr::k_adam, dim3(blocks), dim3(256), 0, s, a);
        }
    }
    if (last && n > 0) hipLaunchKernelGGL(ghr::k_adam_finish, dim3(1), dim3(64), 0, s, state, skip_mask, n_groups);
    return finish(s, 0);
}
}  // namespace

int ghr_adam_step(void* stream, int64_t n, float* p, float* g, float* m, float* v, int32_t* state, int32_t n_groups,
                  const int64_t* group_end_host, const float* lr_host, double beta1, double beta2, float eps,
                  int32_t nan_guard, int32_t zero_grad, uint32_t skip_mask)
{
    return ghr_adam_step_range(stream, n, 0, n, p, g, m, v, state, n_groups, group_end_host, lr_host, beta1, beta2,
                               eps, nan_guard, zero_grad, 1, skip_mask);
}

int ghr_adam_relay_rows(void* stream, int32_t n_groups, const int32_t* width_host, int64_t P_old, int64_t P_new,
                        const int64_t* take, const uint8_t* fresh, const int64_t* child, const float* const* override_host,
                        const float* p_in, const float* m_in, const float* v_in, float* p_out, float* m_out, float* v_out)
{
    if (n_groups <= 0 || n_groups > GHR_ADAM_MAX_GROUPS || !width_host || P_old < 0 || P_new < 0)
        return fail(GHR_E_INVALID, "ghr_adam_relay_rows: bad sizes");
    if (P_new == 0) return GHR_OK;
    if (!take || !fresh || !p_in || !m_in || !v_in || !p_out || !m_out || !v_out)
        return fail(GHR_E_INVALID, "ghr_adam_relay_rows: NULL buffer");
    ghr::RelayArgs a;
    a.P_old = P_old; a.P_new = P_new; a.n_groups = n_groups;
    long long off_old = 0, end_new = 0;
    for (int g = 0; g < n_groups; g++) {
        if (width_host[g] <= 0) return fail(GHR_E_INVALID, "ghr_adam_relay_rows: a group without columns");
        a.width[g] = width_host[g];
        a.off_old[g] = off_old;
        off_old += (long long)width_host[g] * P_old;
        end_new += (long long)width_host[g] * P_new;
        a.end_new[g] = end_new;
        a.override_[g] = override_host ? override_host[g] : nullptr;
        if (a.override_[g] && !child) return fail(GHR_E_INVALID, "ghr_adam_relay_rows: override rows without child indices");
    }
    for (int g = n_groups; g < GHR_ADAM_MAX_GROUPS; g++) { a.width[g] = 1; a.off_old[g] = 0; a.end_new[g] = end_new; a.override_[g] = nullptr; }
    a.take = (const long long*)take; a.fresh = fresh; a.child = (const long long*)child;
    a.p_in = p_in; a.m_in = m_in; a.v_in = v_in; a.p_out = p_out; a.m_out = m_out; a.v_out = v_out;
    hipStream_t s = (hipStream_t)stream;
    const long long blocks_ll = (end_new + 255) / 256;
    const int blocks = (int)(blocks_ll < 16384 ? blocks_ll : 16384);
    hipLaunchKernelGGL(ghr::k_relay_rows, dim3(blocks), dim3(256), 0, s, a);
    return finish(s, 0);
}

int ghr_adam_nan_scan(void* stream, const float* g, int64_t count, int32_t* state)
{
    if (count < 0 || !state || (count > 0 && !g)) return fail(GHR_E_INVALID, "ghr_adam_nan_scan: bad args");
    if (count == 0) return GHR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(ghr::k_adam_nan_flag, dim3(blocks), dim3(256), 0, s, g, (long long)count, state);
    return finish(s, 0);
}

int ghr_mark_visible(void* stream, int32_t P, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    if (P < 0) return fail(GHR_E_INVALID, "P < 0");
    if (P == 0) return GHR_OK;
    if (!means3D || !viewmatrix || !present) return fail(GHR_E_INVALID, "ghr_mark_visible: NULL buffer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ghr::k_mark_visible, dim3((P + GHR_BLOCK - 1) / GHR_BLOCK), dim3(GHR_BLOCK), 0, s, P, means3D,
                       viewmatrix, present);
    return finish(s, 0);
}

int ghr_selftest_wave(void* stream, const float* in, float* out)
{
    if (!in || !out) return fail(GHR_E_INVALID, "ghr_selftest_wave: NULL buffer");
    hipLaunchKernelGGL(ghr::k_wave_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    return finish((hipStream_t)stream, 1);
}

int ghr_selftest_math(void* stream, int32_t n, const float* in, float* out)
{
    if (n < 0 || (n > 0 && (!in || !out))) return fail(GHR_E_INVALID, "ghr_selftest_math: bad arguments");
    if (n == 0) return GHR_OK;
    hipLaunchKernelGGL(ghr::k_math_selftest, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, in, out);
    return finish((hipStream_t)stream, 1);
}

int ghr_set_deterministic(int32_t on)
{
    const int prev = g_deterministic;
    g_deterministic = on != 0;
    return prev;
}

int ghr_set_profile_events(void* fwd_start, void* fwd_stop, void* bwd_start, void* bwd_stop)
{
    g_ev[0] = (hipEvent_t)fwd_start; g_ev[1] = (hipEvent_t)fwd_stop;
    g_ev[2] = (hipEvent_t)bwd_start; g_ev[3] = (hipEvent_t)bwd_stop;
    return GHR_OK;
}

#ifdef GHR_K8_PROF
// kernel-experiment builds only (not declared in include/ghr.h): per-wave phase cycles of the instrumented K8
int ghr_debug_prof(unsigned long long* out, int n_slots, int reset)
{
    if (n_slots < 0 || n_slots > GHR_PROF_SLOTS) return GHR_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return GHR_E_HIP;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ghr::g_k8_prof), 64 * (size_t)n_slots) != hipSuccess) return GHR_E_HIP;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ghr::g_k8_prof)) != hipSuccess) return GHR_E_HIP;
        if (hipMemset(p, 0, 64 * (size_t)GHR_PROF_SLOTS) != hipSuccess) return GHR_E_HIP;
    }
    return GHR_OK;
}
int ghr_debug_timeline(unsigned long long* out, int n_slots)
{
    if (!out || n_slots < 0 || n_slots > GHR_PROF_SLOTS) return GHR_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return GHR_E_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ghr::g_k8_tl), 16 * (size_t)n_slots) != hipSuccess) return GHR_E_HIP;
    return GHR_OK;
}
#endif

int ghr_ws_inspect(int32_t P, int32_t W, int32_t H, int32_t mode_b, uint32_t R, const void* geom_ws,
                   const void* img_ws, const void* bin_ws, ghr_ws_view* out)
{
    if (!out || P < 0 || W <= 0 || H <= 0) return fail(GHR_E_INVALID, "ghr_ws_inspect: bad args");
    const size_t T = (size_t)grid_x(W) * grid_x(H);
    Geom g; Img im; Bin b;
    carve_geom(align_base(geom_ws), (size_t)P, mode_b != 0, &g);
    carve_img(align_base(img_ws), (size_t)W * H, T, &im);
    carve_bin(bin_ws ? align_base(bin_ws) : nullptr, (size_t)R, (size_t)T, &b);
    out->rec = (const float*)g.rec;
    out->depths = g.depths;
    out->rects = (const uint32_t*)g.rects;
    out->cov3D = g.cov3D;
    out->final_T = im.final_T;
    out->n_contrib = im.n_contrib;
    out->tile_start = im.tile_start;
    out->keys = b.keys;
    out->point_list = b.point_list;
    return GHR_OK;
}

}  // extern "C"
