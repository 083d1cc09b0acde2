// ghr_hostsim.cpp -- TEST SCAFFOLDING (never shipped, never on the product path).
//
// Runs the product's own `__host__ __device__` per-Gaussian / per-pixel functions (gaussianhaircut_amd/csrc/*.h:
// preprocess_one, bitonic_any_n, fwd_step, bwd_step, geom_bwd_one, xcd_tile) sequentially on the CPU, with the
// same orchestration as csrc/ghr_capi.hip, so that the `-m "not gpu"` suite can compare the kernels' arithmetic
// with the oracle before any GPU time is spent.  What it cannot cover (LDS staging, wave reductions, atomics,
// barriers) is covered by the `-m gpu` parity tests.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ghr.h"
#include "../../gaussianhaircut_amd/csrc/ghr_adam.h"
#include "../../gaussianhaircut_amd/csrc/ghr_binning.h"
#include "../../gaussianhaircut_amd/csrc/ghr_geom_bwd.h"
#include "../../gaussianhaircut_amd/csrc/ghr_loss.h"
#include "../../gaussianhaircut_amd/csrc/ghr_preprocess.h"
#include "../../gaussianhaircut_amd/csrc/ghr_project.h"
#include "../../gaussianhaircut_amd/csrc/ghr_render_bwd.h"
#include "../../gaussianhaircut_amd/csrc/ghr_render_fwd.h"
#include "../../gaussianhaircut_amd/csrc/ghr_strands.h"

namespace {
struct Sim {
    int P, W, H, gx, gy, T;
    bool mode_b;
    std::vector<ghr::f4> rec;
    std::vector<float> depths, cov3D, final_T;
    std::vector<ghr::rect4> rects;
    std::vector<int> radii;
    std::vector<uint32_t> tile_start, point_list, n_contrib;
    std::vector<uint64_t> keys;
    uint32_t R;
};
}  // namespace

static float* g_sim_d_rgb = nullptr;

extern "C" {

void* ghrsim_forward(const ghr_view_args* a, int32_t* radii_out, float* out_color)
{
    Sim* s = new Sim();
    s->P = a->P; s->W = a->W; s->H = a->H;
    s->gx = (a->W + 15) / 16; s->gy = (a->H + 15) / 16; s->T = s->gx * s->gy;
    s->mode_b = a->conic_precomp == nullptr;
    const int P = a->P, T = s->T;
    s->rec.assign((size_t)4 * P, ghr::f4{0, 0, 0, 0});
    s->depths.assign(P, 0.f);
    s->rects.assign(P, ghr::rect4{0u, 0u, 0u, 0u});
    s->cov3D.assign((size_t)6 * P, 0.f);
    s->radii.assign(P, 0);
    std::vector<uint32_t> count(T, 0u);
    uint32_t slot_alloc = 0;

    ghr::PreArgs pa;
    pa.P = P; pa.W = a->W; pa.H = a->H; pa.gx = s->gx; pa.gy = s->gy;
    pa.means3D = a->means3D; pa.colors = a->colors; pa.opacities = a->opacities;
    pa.scales = a->scales; pa.rotations = a->rotations; pa.cov3D_precomp = a->cov3D_precomp;
    pa.conic_precomp = a->conic_precomp; pa.view = a->viewmatrix; pa.proj = a->projmatrix;
    pa.scale_modifier = a->scale_modifier; pa.tan_fovx = a->tan_fovx; pa.tan_fovy = a->tan_fovy;
    pa.focal_y = a->H / (2.0f * a->tan_fovy);
    pa.focal_x = a->W / (2.0f * a->tan_fovx);
    pa.rec = s->rec.data(); pa.depths = s->depths.data(); pa.rects = s->rects.data(); pa.cov3D = s->cov3D.data();
    pa.radii = s->radii.data(); pa.tile_count = count.data();
    for (int idx = 0; idx < P; idx++) {
        int x0, y0, x1, y1;
        if (!ghr::preprocess_one(pa, idx, x0, y0, x1, y1)) continue;
        s->rects[idx].z = slot_alloc;  // k_preprocess: wave_alloc on the slot counter (any disjoint assignment is valid)
        slot_alloc += (uint32_t)((x1 - x0) * (y1 - y0));
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) count[y * s->gx + x]++;
    }
    // k_tile_scan
    s->tile_start.assign(T + 1, 0u);
    for (int t = 0; t < T; t++) s->tile_start[t + 1] = s->tile_start[t] + count[t];
    s->R = s->tile_start[T];
    // k_scatter (appended in DESCENDING idx order on purpose: the sort must not depend on append order)
    s->keys.assign(s->R, 0ull);
    s->point_list.assign(s->R, 0u);
    std::vector<uint32_t> cursor(T, 0u);
    for (int idx = P - 1; idx >= 0; idx--) {
        const ghr::rect4 r = s->rects[idx];
        const int x0 = r.x & 0xffffu, x1 = r.x >> 16, y0 = r.y & 0xffffu, y1 = r.y >> 16;
        if (x1 <= x0 || y1 <= y0) continue;
        uint32_t db;
        std::memcpy(&db, &s->depths[idx], 4);
        const uint64_t key = ((uint64_t)db << 32) | (uint32_t)idx;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const int t = y * s->gx + x;
                s->keys[s->tile_start[t] + cursor[t]++] = key;
            }
    }
    // k_tile_sort
    for (int t = 0; t < T; t++) {
        const uint32_t b = s->tile_start[t], n = s->tile_start[t + 1] - b;
        if (n > 1) ghr::bitonic_any_n<false>(s->keys.data() + b, n, 0, 1);
        for (uint32_t i = 0; i < n; i++) s->point_list[b + i] = (uint32_t)s->keys[b + i];
    }
    // k_render_fwd
    const size_t N = (size_t)a->W * a->H;
    s->final_T.assign(N, 0.f);
    s->n_contrib.assign(N, 0u);
    for (int t = 0; t < T; t++) {
        const int tx = t % s->gx, ty = t / s->gx;
        const uint32_t beg = s->tile_start[t], n = s->tile_start[t + 1] - beg;
        for (int tid = 0; tid < 256; tid++) {
            const int px = tx * 16 + (tid & 15), py = ty * 16 + (tid >> 4);
            if (!(px < a->W && py < a->H)) continue;
            ghr::PixFwd st;
            st.T = 1.f; st.last = 0;
            for (int c = 0; c < GHR_C; c++) st.C[c] = 0.f;
            bool done = false;
            // k_render_fwd's cull: the 4x4-pixel cell this pixel belongs to
            const float sx0 = (float)(tx * 16 + 4 * ((tid & 15) >> 2)), sy0 = (float)(ty * 16 + 4 * (tid >> 6));
            for (uint32_t j = 0; j < n && !done; j++) {
                const ghr::f4* r = s->rec.data() + 4 * (size_t)s->point_list[beg + j];
                if (!ghr::cell_hit(ghr::alpha_bbox(r[0], r[1]), ghr::ellipse_params(r[0], r[1]), r[0], sx0, sy0)) continue;  // k_render_fwd's cell cull
                done = ghr::fwd_step(st, true, (float)px, (float)py, r[0], r[1], r[2], r[3], j + 1);
            }
            const size_t pix = (size_t)a->W * py + px;
            s->final_T[pix] = st.T;
            s->n_contrib[pix] = st.last;
            for (int c = 0; c < GHR_C; c++) out_color[c * N + pix] = st.C[c] + st.T * a->background[c];
        }
    }
    std::memcpy(radii_out, s->radii.data(), sizeof(int) * P);
    return s;
}

uint32_t ghrsim_num_rendered(void* h) { return ((Sim*)h)->R; }
const float* ghrsim_rec(void* h) { return (const float*)((Sim*)h)->rec.data(); }
const float* ghrsim_depths(void* h) { return ((Sim*)h)->depths.data(); }
const uint32_t* ghrsim_tile_start(void* h) { return ((Sim*)h)->tile_start.data(); }
const uint32_t* ghrsim_point_list(void* h) { return ((Sim*)h)->point_list.data(); }
const uint64_t* ghrsim_keys(void* h) { return ((Sim*)h)->keys.data(); }
const float* ghrsim_final_T(void* h) { return ((Sim*)h)->final_T.data(); }
const uint32_t* ghrsim_n_contrib(void* h) { return ((Sim*)h)->n_contrib.data(); }
void ghrsim_free(void* h) { delete (Sim*)h; }

void ghrsim_backward(void* h, const ghr_view_args* a, const float* dL_dpix, float* dL_dmeans2D, float* dL_dconic,
                     float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales,
                     float* dL_drotations)
{
    Sim* s = (Sim*)h;
    const int P = s->P, T = s->T;
    const size_t N = (size_t)a->W * a->H;
    // one 64-B line per Gaussian-tile instance, in tile-list order as K8 writes them (inst_line maps a Gaussian's
    // instances, numbered by rect4_slot, to their lines: what the tile sort leaves); the per-pixel sum inside
    // an instance is order-free on the GPU (wave butterfly + ds_add), so it is accumulated in double here
    std::vector<double> acc((size_t)16 * s->R, 0.0);
    for (int t = 0; t < T; t++) {
        const int tx = t % s->gx, ty = t / s->gx;
        const uint32_t beg = s->tile_start[t], n = s->tile_start[t + 1] - beg;
        for (int tid = 0; tid < 256; tid++) {
            const int px = tx * 16 + (tid & 15), py = ty * 16 + (tid >> 4);
            if (!(px < a->W && py < a->H)) continue;
            const size_t pix = (size_t)a->W * py + px;
            ghr::PixBwd st;
            st.T_final = s->final_T[pix]; st.T = st.T_final; st.S = 0.f; st.last_alpha = 0.f; st.last_cdot = 0.f;
            st.bgdot = 0.f;
            for (int c = 0; c < GHR_C; c++) {
                const float d = dL_dpix[c * N + pix];
                if (c & 1) st.dL[c / 2].y = d; else st.dL[c / 2].x = d;
                st.bgdot = ghr::fma_(a->background[c], d, st.bgdot);
            }
            const uint32_t last = s->n_contrib[pix];
            // the kernel drops entries above the CELL's max n_contrib; emulate with the cell of this pixel
            uint32_t cell_max = 0;
            for (int q = 0; q < 16; q++) {
                const int qx = tx * 16 + 4 * ((tid & 15) >> 2) + (q & 3), qy = ty * 16 + 4 * (tid >> 6) + (q >> 2);
                if (qx < a->W && qy < a->H) cell_max = std::max(cell_max, s->n_contrib[(size_t)a->W * qy + qx]);
            }
            for (uint32_t k = 0; k < n; k++) {
                const uint32_t pos = n - 1 - k;
                if (!(pos < cell_max)) continue;
                const uint32_t id = s->point_list[beg + pos];
                const ghr::f4* r = s->rec.data() + 4 * (size_t)id;
                {
                    const float sx0 = (float)(tx * 16 + 4 * ((tid & 15) >> 2)), sy0 = (float)(ty * 16 + 4 * (tid >> 6));
                    if (!ghr::cell_hit(ghr::alpha_bbox(r[0], r[1]), ghr::ellipse_params(r[0], r[1]), r[0], sx0, sy0)) continue;  // k_render_bwd's cell cull
                }
                float g[16];
                // branch-free step: non-contributing visits (pos >= n_contrib of THIS pixel included) run with alpha = 0
                if (ghr::bwd_step(st, pos < last, (float)px, (float)py, r[0], r[1], r[2], r[3], (float)(tid & 15), (float)(tid >> 4), g)) {
                    const size_t line = (size_t)beg + pos;
                    for (int i = 0; i < 16; i++) acc[16 * line + i] += (double)g[i];
                }
            }
        }
    }
    std::vector<float> ginst((size_t)16 * s->R + 16);
    for (size_t i = 0; i < acc.size(); i++) ginst[i] = (float)acc[i];
    std::vector<uint32_t> inst_line((size_t)s->R + 1, 0u);
    for (int t = 0; t < T; t++) {
        const uint32_t beg = s->tile_start[t], n = s->tile_start[t + 1] - beg;
        for (uint32_t pos = 0; pos < n; pos++)
            inst_line[ghr::rect4_slot(s->rects[s->point_list[beg + pos]], t % s->gx, t / s->gx)] = beg + pos;
    }
    ghr::GeomBwdArgs ga;
    ga.P = P; ga.means3D = a->means3D; ga.radii = s->radii.data(); ga.scales = a->scales; ga.rotations = a->rotations;
    ga.cov3D = s->cov3D.data(); ga.conic_precomp = a->conic_precomp; ga.view = a->viewmatrix; ga.proj = a->projmatrix;
    ga.scale_modifier = a->scale_modifier; ga.tan_fovx = a->tan_fovx; ga.tan_fovy = a->tan_fovy;
    ga.focal_y = a->H / (2.0f * a->tan_fovy);
    ga.focal_x = a->W / (2.0f * a->tan_fovx);
    ga.ginst = ginst.data(); ga.inst_line = inst_line.data(); ga.ginst_rows = (uint32_t)s->R;
    ga.rects = s->rects.data(); ga.rec = s->rec.data();
    ga.half_w = 0.5f * a->W; ga.half_h = 0.5f * a->H;
    ga.dL_dmeans2D = dL_dmeans2D; ga.dL_dconic = dL_dconic; ga.dL_dconic3 = nullptr; ga.dL_dopacity = dL_dopacity; ga.dL_dcolors = dL_dcolors;
    ga.dL_dmeans3D = dL_dmeans3D; ga.dL_dcov3D = dL_dcov3D; ga.dL_dscales = dL_dscales; ga.dL_drots = dL_drotations;
    for (int idx = 0; idx < P; idx++) {
        float g16[16];
        ghr::gather_inst_grads(ga.ginst, ga.inst_line, ga.rects[idx], ga.rec[4 * (size_t)idx], ga.rec[4 * (size_t)idx + 1],
                               ga.half_w, ga.half_h, g16, ga.ginst_rows);
        ghr::geom_bwd_one(ga, idx, g16);
    }
}

// ---- fused projection (ghr_project.h): forward state + colours, and raw-parameter gradients from packed gacc ----
// out_rec: [P][16], out_radii [P], out_means2D [P][3]
void ghrsim_project_forward(const ghr::ModelArgs* a_in, float* out_rec, int* out_radii, float* out_means2D,
                            float* out_depths)
{
    ghr::ModelArgs a = *a_in;
    const int P = a.P;
    a.gx = (a.W + 15) / 16; a.gy = (a.H + 15) / 16;
    std::vector<ghr::f4> rec((size_t)4 * P, ghr::f4{0, 0, 0, 0});
    std::vector<ghr::rect4> rects(P);
    std::vector<uint32_t> count((size_t)a.gx * a.gy, 0u);
    a.rec = rec.data(); a.depths = out_depths; a.rects = rects.data(); a.radii = out_radii; a.means2D = out_means2D;
    a.tile_count = count.data();
    const int row = 3 * (a.sh_coeffs - 1);
    for (int i = 0; i < P; i++) {
        int x0, y0, x1, y1;
        out_depths[i] = 0.f;
        ghr::project_one(a, i, a.features_rest + (size_t)i * row, x0, y0, x1, y1);
    }
    std::memcpy(out_rec, rec.data(), sizeof(float) * 16 * (size_t)P);
}

void ghrsim_project_backward3(const ghr::ModelArgs* a_in, const int* radii, const float* gacc, float* d_means2D,
                              float* d_xyz, float* d_ls, float* d_rot, float* d_op, float* d_label, float* d_conf,
                              float* d_fdc, float* d_frest, float* d_dir, float* cam, int detach_means2D);
void ghrsim_project_backward2(const ghr::ModelArgs* a_in, const int* radii, const float* gacc, float* d_means2D,
                              float* d_xyz, float* d_ls, float* d_rot, float* d_op, float* d_label, float* d_conf,
                              float* d_fdc, float* d_frest, float* d_dir);
void ghrsim_project_backward(const ghr::ModelArgs* a_in, const int* radii, const float* gacc, float* d_means2D,
                             float* d_xyz, float* d_ls, float* d_rot, float* d_op, float* d_label, float* d_conf,
                             float* d_fdc, float* d_frest)
{
    ghrsim_project_backward2(a_in, radii, gacc, d_means2D, d_xyz, d_ls, d_rot, d_op, d_label, d_conf, d_fdc, d_frest,
                             nullptr);
}
// + the strand-direction gradient of mode 1 (d_dir may be NULL)
void ghrsim_project_backward2(const ghr::ModelArgs* a_in, const int* radii, const float* gacc, float* d_means2D,
                              float* d_xyz, float* d_ls, float* d_rot, float* d_op, float* d_label, float* d_conf,
                              float* d_fdc, float* d_frest, float* d_dir)
{
    ghrsim_project_backward3(a_in, radii, gacc, d_means2D, d_xyz, d_ls, d_rot, d_op, d_label, d_conf, d_fdc, d_frest, d_dir,
                             nullptr, 0);
}
// + the per-Gaussian camera cotangents: cam [P][GHR_CAM_PARTIALS] (may be NULL), layout in ghr_project.h
void ghrsim_project_backward3(const ghr::ModelArgs* a_in, const int* radii, const float* gacc, float* d_means2D,
                              float* d_xyz, float* d_ls, float* d_rot, float* d_op, float* d_label, float* d_conf,
                              float* d_fdc, float* d_frest, float* d_dir, float* cam, int detach_means2D)
{
    ghr::ModelArgs a = *a_in;
    a.gx = (a.W + 15) / 16; a.gy = (a.H + 15) / 16;
    a.radii = const_cast<int*>(radii);
    ghr::ModelGrads g;
    g.ginst = nullptr; g.d_means2D = d_means2D; g.d_xyz = d_xyz; g.d_log_scales = d_ls; g.d_rotations = d_rot;
    g.d_opacity_logit = d_op; g.d_label_logit = d_label; g.d_orient_conf_log = d_conf; g.d_features_dc = d_fdc;
    g.d_features_rest = d_frest; g.d_dir3d = d_dir; g.d_rgb = g_sim_d_rgb;
    g.accumulate = 0; g.nan_flag = nullptr;
    g.cam_partial = nullptr; g.cam_slot0 = 0; g.cam_stride = 0; g.cam_only = 0; g.detach_means2D = detach_means2D;
    g.dens_grad_accum = nullptr; g.dens_denom = nullptr; g.dens_max_radii = nullptr; g.dens_count = nullptr; g.dens_cap = 0;
    g.overflow_is_bad = 0; g.adam.on = 0;
    const int row = 3 * (a.sh_coeffs - 1);
    for (int i = 0; i < a.P; i++)
        ghr::project_bwd_one(a, g, i, gacc + 16 * (size_t)i, a.features_rest + (size_t)i * row, d_frest + (size_t)i * row,
                             cam ? cam + (size_t)GHR_CAM_PARTIALS * i : nullptr);
}

// (set before a ghrsim_project_backward* call: that call also leaves d_rgb [P,3]; NULL = off)
void ghrsim_set_d_rgb(float* p) { g_sim_d_rgb = p; }

void ghrsim_sh_grad_from_views(int P, int deg, int K, const float* xyz, int n_views, const float* campos, const float* g,
                               long long view_stride, float* d_dc, float* d_rest)
{
    const int row = 3 * (K - 1);
    std::vector<float> tmp((size_t)(row > 0 ? row : 1));
    for (int i = 0; i < P; i++) {
        ghr::sh_grad_from_views_one(deg, K, xyz + 3 * (size_t)i, n_views, campos, 3, g, (size_t)view_stride, (size_t)i,
                                    d_dc + 3 * (size_t)i, tmp.data());
        for (int k = 0; k < row; k++) d_rest[(size_t)i * row + k] = tmp[k];
    }
}

int ghrsim_sizeof_model_args(void) { return (int)sizeof(ghr::ModelArgs); }

// Conservativeness of alpha_bbox: returns the number of pixels in [0,W)x[0,H) that fwd_step's own arithmetic would
// blend (alpha >= 1/255, power <= 0) but that lie OUTSIDE the box.  Must be 0.
int ghrsim_bbox_violations(const float* rec16, int n, int W, int H)
{
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const ghr::f4* r = (const ghr::f4*)(rec16 + 16 * (size_t)i);
        const ghr::f4 bb = ghr::alpha_bbox(r[0], r[1]);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float dx = r[0].x - (float)x, dy = r[0].y - (float)y;
                const float power = -0.5f * (r[0].z * dx * dx + r[1].x * dy * dy) - r[0].w * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf(0.99f, r[1].y * ghr::fast_exp(power));
                if (alpha < 1.0f / 255.0f) continue;
                if (!ghr::bbox_hits(bb, (float)x, (float)x, (float)y, (float)y)) bad++;
                // ... and the 4x4 cell that holds the pixel must pass the box + ellipse test of the render kernels
                const float X0 = (float)(x & ~3), Y0 = (float)(y & ~3);
                if (!ghr::cell_hit(bb, ghr::ellipse_params(r[0], r[1]), r[0], X0, Y0)) bad++;
            }
    }
    return bad;
}

// xcd_tile over the padded grid must visit every tile of [0, n) exactly once (padding workgroups map to >= n)
int ghrsim_xcd_bijective(uint32_t n)
{
    std::vector<uint8_t> seen(n, 0);
    uint32_t visited = 0;
    const uint32_t grid = ghr::xcd_grid(n);
    if (grid < n || grid >= n + 8u * GHR_XCD_RUN) return 0;
    for (uint32_t b = 0; b < grid; b++) {
        const uint32_t t = ghr::xcd_tile(b, n);
        if (t >= n) continue;
        if (seen[t]) return 0;
        seen[t] = 1;
        visited++;
    }
    return visited == n;
}

// sorts `n` keys with the product's network; returns 1 if the result is ascending
int ghrsim_bitonic(uint64_t* keys, uint32_t n)
{
    if (n > 1) ghr::bitonic_any_n<false>(keys, n, 0, 1);
    for (uint32_t i = 1; i < n; i++)
        if (keys[i - 1] > keys[i]) return 0;
    return 1;
}

// the register-blocked form of the network (k_tile_sort's wave path): r = log2 of the keys a thread owns, `threads`
// emulated one after the other inside each pass, `pad` = the LDS padding of the kernel (keys then live at key_slot<true>(i):
// the caller's buffer must hold n + n / 16 + 1 keys).  Returns 1 if the result is ascending.
int ghrsim_bitonic_blocked(uint64_t* keys, uint32_t n, int r, int threads)
{
    if (n > 1) {
        for (int tid = 0; tid < 1; tid++) {  // (passes are separated by barriers: one "thread" walking all groups is the same)
            switch (r) {
            case 1: ghr::bitonic_blocked<1, false, false, false>(keys, n, 0, 1); break;
            case 2: ghr::bitonic_blocked<2, false, false, false>(keys, n, 0, 1); break;
            case 3: ghr::bitonic_blocked<3, false, false, false>(keys, n, 0, 1); break;
            default: ghr::bitonic_blocked<4, false, false, false>(keys, n, 0, 1); break;
            }
        }
    }
    (void)threads;
    for (uint32_t i = 1; i < n; i++)
        if (keys[i - 1] > keys[i]) return 0;
    return 1;
}

// ---- per-pixel / per-element arithmetic of the fused loss and of fused Adam (csrc/ghr_loss.h, csrc/ghr_adam.h) --------
// out[i] = {l, dl/dd0, dl/dd1, dl/dconf} of the orientation term of one pixel
void ghrsim_orient_pixel(int n, const float* d0, const float* d1, const float* conf, const float* gt, const float* m,
                         float* out)
{
    for (int i = 0; i < n; i++) {
        const ghr::OrientPix o = ghr::orient_pixel(d0[i], d1[i], conf[i], gt[i], m[i]);
        out[4 * i] = o.l; out[4 * i + 1] = o.dl_dd0; out[4 * i + 2] = o.dl_dd1; out[4 * i + 3] = o.dl_dconf;
    }
}

// out[i] = {ssim, d/dmu1, d/dE[x^2], d/dE[xy]}
void ghrsim_ssim_point(int n, const float* mu1, const float* mu2, const float* e11, const float* e22, const float* e12,
                       float* out)
{
    for (int i = 0; i < n; i++) {
        float a, b, c;
        out[4 * i] = ghr::ssim_point(mu1[i], mu2[i], e11[i], e22[i], e12[i], a, b, c);
        out[4 * i + 1] = a; out[4 * i + 2] = b; out[4 * i + 3] = c;
    }
}

// one Adam step with k_adam's scalar preparation (step = 1-based step number)
void ghrsim_adam(int n, float* p, const float* g, float* m, float* v, float lr, double beta1, double beta2, float eps,
                 int step)
{
    const double bias1 = 1.0 - pow(beta1, (double)step);
    const float bias2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2), b2 = (float)beta2;
    for (int i = 0; i < n; i++) ghr::adam_update(p[i], g[i], m[i], v[i], (float)((double)lr / bias1), w1, b2, w2, eps, bias2_sqrt);
}

// ---- strand polylines -> segment Gaussians (csrc/ghr_strands.h), with k_strand_build's slab / thread decomposition ---------
void ghrsim_strand_build(int S, int n_seg, const float* origins, const float* dirs, float scale, float* xyz, float* rot,
                         float* scaling)
{
    const int row = 3 * n_seg;
    for (int s = 0; s < S; s++) {
        for (int c = 0; c < 3; c++)
            ghr::strand_scan_fwd(dirs + (size_t)s * row + c, xyz + (size_t)s * row + c, origins[3 * s + c], n_seg);
        for (int k = 0; k < n_seg; k++) {
            const size_t i = (size_t)s * n_seg + k;
            ghr::strand_row_fwd(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], scale, rot + 4 * i, scaling + 3 * i);
        }
    }
}

void ghrsim_strand_build_backward(int S, int n_seg, const float* dirs, const float* d_xyz, const float* d_rot,
                                  const float* d_scaling, float* d_dirs)
{
    const int row = 3 * n_seg;
    std::vector<float> acc((size_t)row, 0.f);
    for (int s = 0; s < S; s++) {
        if (d_xyz != nullptr)
            for (int c = 0; c < 3; c++) ghr::strand_scan_bwd(d_xyz + (size_t)s * row + c, acc.data() + c, n_seg);
        for (int k = 0; k < n_seg; k++) {
            const size_t i = (size_t)s * n_seg + k;
            float o[3];
            ghr::strand_row_bwd(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], d_rot != nullptr ? d_rot + 4 * i : nullptr,
                                d_scaling != nullptr ? d_scaling[3 * i] : 0.f, o);
            for (int c = 0; c < 3; c++) d_dirs[3 * i + c] = o[c] + (d_xyz != nullptr ? acc[3 * k + c] : 0.f);
        }
    }
}

}  // extern "C"
