// ghr_project.h -- fused "model -> rasterizer state" kernel and its hand-derived backward.
//
// Replaces, for the free-Gaussian model, the ~60 small PyTorch kernels render() runs per view over all P rows
// (measured on MI355X: 16 aten::bmm calls on P x 3 x 3 alone cost 68 ms of a 95 ms step at P = 500k):
//   src/scene/gaussian_model.py:107-141   activations (exp / sigmoid / normalize)
//   :230-315  get_covariance, get_covariance_2d, get_conic           :317-342  get_mean_2d, get_depths
//   :344-393  get_direction_2d ("strand direction" channels)         :143-228  filter_points
//   src/utils/sh_utils.py:57-112 eval_sh ; src/gaussian_renderer/__init__.py:58-83 colour assembly + mask gathers
// and, in backward, everything autograd would do for that graph -- including the strand-direction term
// d(dir2D)/d(rotation, scaling, xyz) and the SH view-direction term -- in ONE pass over the Gaussians.
//
// Forward = K1 with the conic / features computed in-register from the raw parameters (same cull, radius, rect and
// packed record as k_preprocess, so binning / render kernels are unchanged).  The arithmetic follows the PYTHON
// pipeline (that is what the reference actually runs; its in-kernel geometry is dormant):
//   conic = (c, -b, a) / (det + eps)   with eps = 1e-12 (gaussian_model.py:312), then K1's own
//   cov = conic^-1 for the radius (forward.cu:242-257), so radii match the reference's mode-A chain.
// Backward differentiates exactly that pipeline (torch semantics: clamp() passes gradient only inside the range,
// normalize() is differentiated, clamp_min(sh + 0.5, 0) masks), NOT the dormant CUDA K9/K10.
#pragma once
#include "ghr_adam.h"
#include "ghr_preprocess.h"

namespace ghr {

#define GHR_SH_MAX 16  // (3 + 1)^2 coefficients per colour channel

// Two parametrisations share the kernels:
//   mode 0  the free-Gaussian GaussianModel: raw parameters, activations applied here (exp / sigmoid), 2D direction =
//           longest principal axis (gaussian_model.py:344-393);
//   mode 1  explicit linear-space Gaussians, what render_hair() feeds the rasterizer in the strand stage
//           (gaussian_renderer/__init__.py:116-214, gaussian_model_strands.py:230-452): scaling / opacity / label /
//           orientation confidence are already activated (a NULL pointer means the constant in const_*), the 2D direction
//           is normalize(dir3d) @ T (zero when dir3d is NULL: the frozen head).
// A call covers ONE segment of `P` Gaussians whose outputs land at workspace rows row0 .. row0+P-1 (row0 a multiple of
// 256), so the frozen head and the trainable strands are projected by two launches into one rasterizer state.
struct ModelArgs {
    int P, W, H, gx, gy;
    int sh_degree;   // active degree (0..3)
    int sh_coeffs;   // K = (max_degree + 1)^2 coefficients stored per channel (<= 16)
    int mode;        // 0 / 1, see above
    int row0;        // first workspace row of the segment
    const float* xyz;             // [P,3]
    const float* log_scales;      // [P,3]  mode 0: scaling = exp(.) (gaussian_model.py:108-109); mode 1: scaling
    const float* rotations;       // [P,4]  raw quaternion (r,x,y,z), normalised here (general_utils.py:79-83)
    const float* opacity_logit;   // [P]    mode 0: sigmoid(.); mode 1: opacity or NULL
    const float* label_logit;     // [P]    mode 0: sigmoid(.); mode 1: label or NULL
    const float* orient_conf_log; // [P]    mode 0: exp(.);     mode 1: confidence or NULL
    const float* dir3d;           // [P,3]  mode 1: strand direction or NULL
    float const_opacity, const_label, const_conf;  // mode 1 values for NULL pointers
    const float* features_dc;     // [P,1,3]
    const float* features_rest;   // [P,K-1,3]
    const float* view;            // [16] world_view_transform
    const float* proj;            // [16] full_proj_transform
    const float* campos;          // [3]
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y, conic_eps;
    const float* fovx;            // device scalars FoVx, FoVy (radians) or NULL, NULL: override tan_fov* / focal_* above --
    const float* fovy;            // a trainable FoV changes every step: tan(FoV / 2) is taken here, the host never waits for it
    // forward outputs
    f4* rec;
    float* depths;
    rect4* rects;
    int* radii;
    float* means2D;  // [rows,3] NDC (viewspace_points values), may be null
    uint32_t* tile_count;  // [2][T] counts (small rects / big rects: count_tiles)
    uint32_t* pos;         // [rows][GHR_BIG_RECT] list positions of the small-rect instances (count_tiles)
    uint32_t* slot_blk;    // [ceil(rows/256)] gradient slots per workgroup
};  // rec / depths / rects / radii / means2D / slot_blk are indexed by WORKSPACE ROW (row0 + idx)

struct ModelGrads {
    const float* ginst;  // [R][16] per-instance packed gradients from K8, in tile-list order
    const uint32_t* inst_line;  // [R] line of every instance (instances numbered by rect4_slot), from the tile sort
    uint32_t ginst_rows; // lines in ginst (bound of the gather)
    float* d_means2D;   // [P,3]  dL/d(NDC mean) (densification signal), z = 0
    float* d_xyz;       // [P,3]
    float* d_log_scales;// [P,3]
    float* d_rotations; // [P,4]
    float* d_opacity_logit;   // [P]
    float* d_label_logit;     // [P]
    float* d_orient_conf_log; // [P]
    float* d_features_dc;     // [P,1,3]
    float* d_features_rest;   // [P,K-1,3]
    float* d_rgb;             // [P,3] optional, ASSIGNED: dL/d(rgb) behind the colour clamp -- the view's SH gradients in factored
                              // form (d sh[k][c] = basis_k(dir) d_rgb[c]: k_sh_grad_from_views); d_features_dc / _rest may then be NULL
    float* d_dir3d;           // [P,3] mode 1 (NULL: not wanted).  In mode 1 d_log_scales / d_opacity_logit / d_label_logit /
                              // d_orient_conf_log receive the gradients of the LINEAR quantities and may be NULL
    int accumulate;           // != 0: parameter gradients are ADDED to the output buffers (d_means2D is always assigned)
    int* nan_flag;            // optional: set to 1 when any parameter gradient value written is NaN or +-inf
    // Camera gradients (k_project_bwd<true>): every workgroup leaves the sums over its 64 Gaussians of the GHR_CAM_PARTIALS
    // camera cotangents in slot cam_slot0 + blockIdx.x of a component-major table [GHR_CAM_PARTIALS][cam_stride];
    // k_cam_fold adds the slots up in a fixed order.
    float* cam_partial;
    uint32_t cam_slot0, cam_stride;
    int cam_only;             // != 0: nothing but d_means2D and the camera partials is written (frozen head segment)
    int detach_means2D;       // != 0: the NDC means are constants of the graph (render_hair() detaches the head's,
                              // gaussian_renderer/__init__.py:136): no gradient through full_proj_transform
    // Per-iteration densification statistics of the stage-1 loop (train_gaussians.py:161-165, gaussian_model.py:739-741), folded
    // into this pass (all three or none): for every Gaussian the view sees (radius > 0)
    //   xyz_gradient_accum += |d_means2D.xy|,  denom += 1,  max_radii2D = max(max_radii2D, radius)
    float* dens_grad_accum;   // [P] (tensor [P,1])
    float* dens_denom;        // [P] (tensor [P,1])
    float* dens_max_radii;    // [P] float, like the reference's
    AdamFuse adam;               // adam.on != 0 (mode 0, the step's LAST backward on one rank): the optimizer update is applied
                                 // here, from registers, instead of storing the gradients (ghr_adam.h)
    const uint32_t* dens_count;  // optional: the view's instance count on the device (k_tile_scan's R_dev) ...
    uint32_t dens_cap;           // ... the statistics update is skipped when it exceeds this capacity: the view was rasterized with
                                 // a guessed capacity that turned out too small (include/ghr.h, ghr_forward_stage2), its gradients
                                 // are invalid and the caller recomputes the view -- statistics must not be counted twice
    int overflow_is_bad;         // != 0 (steps with the fused optimizer update): such a view also raises nan_flag, so that the
                                 // update it would poison is undone like one with non-finite gradients
};

// Camera cotangents a Gaussian contributes (the reference's projection graph is differentiable w.r.t. the camera:
// gaussian_model.py:258-266,279-294 view matrix inside t, W and J, tan(FoV / 2) inside focal and the clamp limits; :332-335
// full_proj_transform; gaussian_renderer/__init__.py:59 camera_center; cameras.py:85-151 makes them functions of trainable
// residuals).  Only the entries that can be non-zero are carried:
//   [0..11]  d view[4 r + c], r = 0..3, c = 0..2  (column 3 of world_view_transform is never read)      at 3 r + c
//   [12..23] d proj[4 r + c], r = 0..3, c in {0, 1, 3}  (NDC z carries no gradient, rasterize_points.cu:160) at 12 + 3 r + {0,1,2}
//   [24..25] d tan(FoVx / 2), d tan(FoVy / 2)     [26..28] d camera_center     [29..31] zero
#ifndef GHR_CAM_PARTIALS  // (include/ghr.h states the same two numbers for the callers)
#define GHR_CAM_PARTIALS 32
#define GHR_CAM_GRADS 37  // what k_cam_fold writes: view[16] | proj[16] | camera_center[3] | tanfov[2]
#endif

// Whether a parameter-gradient value about to be stored is NON-FINITE (project_bwd_store raises nan_flag for it).  The
// reference's guard looks for NaN only (train_gaussians.py:174-177); raising the flag for +-inf as well costs nothing (an
// infinite gradient turns the parameter into NaN in the very next Adam update anyway) and closes the data-parallel hole
// in which +inf on one rank and -inf on another meet as a NaN only inside the all-reduced sum.
GHR_HD bool nonfinite(float v) { return !(fabsf(v) <= 3.402823466e38f); }

GHR_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// SH basis values (sh_utils.py:57-112 / auxiliary.h:21-39) for a unit direction.
GHR_HD void sh_basis(int deg, float x, float y, float z, float* b)
{
#pragma unroll
    for (int k = 1; k < GHR_SH_MAX; k++) b[k] = 0.f;  // inactive coefficients contribute 0 (and get 0 gradient)
    b[0] = 0.28209479177387814f;
    if (deg > 0) {
        const float C1 = 0.4886025119029199f;
        b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = 1.0925484305920792f * xy;
            b[5] = -1.0925484305920792f * yz;
            b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
            b[7] = -1.0925484305920792f * xz;
            b[8] = 0.5462742152960396f * (xx - yy);
            if (deg > 2) {
                b[9] = -0.5900435899266435f * y * (3.f * xx - yy);
                b[10] = 2.890611442640554f * xy * z;
                b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
                b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
                b[14] = 1.445305721320277f * z * (xx - yy);
                b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
            }
        }
    }
}
// sum_k v[k] * d(basis_k)/d(x,y,z)  (cf. backward.cu:58-122, written per basis function so nothing is stored)
GHR_HD void sh_basis_grad_dot(int deg, float x, float y, float z, const float* v, float& ox, float& oy, float& oz)
{
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (deg > 0) {
        const float C1 = 0.4886025119029199f;
        gy += -C1 * v[1];
        gz += C1 * v[2];
        gx += -C1 * v[3];
        if (deg > 1) {
            const float c0 = 1.0925484305920792f, c2 = 0.31539156525252005f, c4 = 0.5462742152960396f;
            gx += c0 * y * v[4];            gy += c0 * x * v[4];
            gy += -c0 * z * v[5];           gz += -c0 * y * v[5];
            gx += -2.f * c2 * x * v[6];     gy += -2.f * c2 * y * v[6];   gz += 4.f * c2 * z * v[6];
            gx += -c0 * z * v[7];           gz += -c0 * x * v[7];
            gx += 2.f * c4 * x * v[8];      gy += -2.f * c4 * y * v[8];
            if (deg > 2) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                const float k0 = -0.5900435899266435f, k1 = 2.890611442640554f, k2 = -0.4570457994644658f,
                            k3 = 0.3731763325901154f, k5 = 1.445305721320277f;
                gx += k0 * 6.f * xy * v[9];                   gy += k0 * 3.f * (xx - yy) * v[9];
                gx += k1 * yz * v[10];                        gy += k1 * xz * v[10];       gz += k1 * xy * v[10];
                gx += k2 * -2.f * xy * v[11];                 gy += k2 * (4.f * zz - xx - 3.f * yy) * v[11];
                gz += k2 * 8.f * yz * v[11];
                gx += k3 * -6.f * xz * v[12];                 gy += k3 * -6.f * yz * v[12];
                gz += k3 * (6.f * zz - 3.f * xx - 3.f * yy) * v[12];
                gx += k2 * (4.f * zz - 3.f * xx - yy) * v[13]; gy += k2 * -2.f * xy * v[13];
                gz += k2 * 8.f * xz * v[13];
                gx += k5 * 2.f * xz * v[14];                  gy += k5 * -2.f * yz * v[14]; gz += k5 * (xx - yy) * v[14];
                gx += k0 * 3.f * (xx - yy) * v[15];           gy += k0 * -6.f * xy * v[15];
            }
        }
    }
    ox = gx; oy = gy; oz = gz;
}

// The camera's matrices and position are the same for every thread and not written while the kernels run: read through the
// constant address space they become scalar loads (s_load -> SGPRs, counted in lgkmcnt).  As plain global loads they were
// vector loads of one address by 64 lanes, each a stop at vmcnt in the middle of the arithmetic -- and vmcnt retires in
// order, so such a stop also waits for everything requested before it (the coefficient slab still on its way).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) float* uniform_floats;
#define GHR_UNIFORM(p) ((uniform_floats)(uintptr_t)(p))
#else
typedef const float* uniform_floats;
#define GHR_UNIFORM(p) (p)
#endif

// One Gaussian's raw parameters (everything per-Gaussian but the higher SH coefficients, which the kernels stage through
// LDS).  The kernels request them at the very top, next to the coefficient slab, so that the projection arithmetic finds
// them in registers: loaded where they are used they were two to three more dependent HBM round trips inside a workgroup's
// compute phase (round 5 phase profile: profiles/r05m).  NULL pointers of mode 1 read as their constants / zeros.
struct RawIn {
    float xyz[3], ls[3], q[4];
    float op, lab, conf;  // opacity / label / orientation confidence as stored (logits and log in mode 0)
    float dc[3];
    float dir[3];         // mode 1 strand direction (zeros without one)
};

// No branches: an absent array reads xyz instead (always there, always long enough) and the value is dropped.  A load
// behind a branch -- even a uniform one -- may or may not have been issued, and the compiler can then only wait for OLDER
// loads with vmcnt(0): the kernels want to use these values while the loads they issue next are still in flight.
GHR_HD void load_raw(const ModelArgs& a, int idx, RawIn& in)
{
    const float* dir = a.dir3d ? a.dir3d : a.xyz;
    const float* op = a.opacity_logit ? a.opacity_logit : a.xyz;
    const float* lab = a.label_logit ? a.label_logit : a.xyz;
    const float* conf = a.orient_conf_log ? a.orient_conf_log : a.xyz;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        in.xyz[i] = a.xyz[3 * idx + i];
        in.ls[i] = a.log_scales[3 * idx + i];
        in.dc[i] = a.features_dc[3 * (size_t)idx + i];
        const float d = dir[3 * idx + i];
        in.dir[i] = a.dir3d ? d : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) in.q[i] = a.rotations[4 * idx + i];
    const float vo = op[idx], vl = lab[idx], vc = conf[idx];
    in.op = a.opacity_logit ? vo : a.const_opacity;
    in.lab = a.label_logit ? vl : a.const_label;
    in.conf = a.orient_conf_log ? vc : a.const_conf;
}

// Everything both passes need, recomputed from the raw parameters (cheaper than storing it: 61 floats in, ~300 flop).
struct ProjCtx {
    float s[3], s0[3];    // scaling * modifier, scaling
    float qn[4], qlen;    // normalised quaternion, |q|
    float ax[3][3];       // ax[i] = i-th principal axis = row i of the reference's R (general_utils.py:104-112)
    float t[3];           // view-space mean (unclamped)
    float clx, cly;       // clamp(tx/tz), clamp(ty/tz)
    bool inx, iny;        // inside the 1.3*tanfov clamp range
    float u[3], v[3];     // T[:,0], T[:,1] with T = W @ J (gaussian_model.py:279-290)
    float p[3], r[3];     // p_i = ax_i . u, r_i = ax_i . v
    float a, b, c, det, k;  // cov2D (with +0.3), det, k = 1/(det+eps)
    int jmax;             // longest axis (gaussian_model.py:384-388); use sel3() -- never index an array with it
                          // (dynamic indexing sends the arrays to scratch / LDS)
    float Wc[3][3];       // Wc[c] = column c of view[:3,:3]
    float j00, j20, j11, j21, txp, typ;
    float tfx, tfy, fx, fy; // tan(FoV / 2) and focal lengths in use (ModelArgs' own or from ModelArgs::tanfov)
};

GHR_HD float sel3(const float* v, int j) { return j == 0 ? v[0] : (j == 1 ? v[1] : v[2]); }

GHR_HD void proj_setup(const ModelArgs& a, const RawIn& in, ProjCtx& c)
{
    const float mx = in.xyz[0], my = in.xyz[1], mz = in.xyz[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        c.s0[i] = a.mode == 0 ? expf(in.ls[i]) : in.ls[i];
        c.s[i] = c.s0[i] * a.scale_modifier;
    }
    const float q0 = in.q[0], q1 = in.q[1], q2 = in.q[2], q3 = in.q[3];
    c.qlen = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    const float il = 1.0f / c.qlen;
    const float w = q0 * il, x = q1 * il, y = q2 * il, z = q3 * il;
    c.qn[0] = w; c.qn[1] = x; c.qn[2] = y; c.qn[3] = z;
    c.ax[0][0] = 1 - 2 * (y * y + z * z); c.ax[0][1] = 2 * (x * y + w * z); c.ax[0][2] = 2 * (x * z - w * y);
    c.ax[1][0] = 2 * (x * y - w * z); c.ax[1][1] = 1 - 2 * (x * x + z * z); c.ax[1][2] = 2 * (y * z + w * x);
    c.ax[2][0] = 2 * (x * z + w * y); c.ax[2][1] = 2 * (y * z - w * x); c.ax[2][2] = 1 - 2 * (x * x + y * y);
    const uniform_floats V = GHR_UNIFORM(a.view);  // V[4*row + col]; t = xyz @ V[:3,:3] + V[3,:3]
#pragma unroll
    for (int col = 0; col < 3; col++) {
        c.t[col] = mx * V[col] + my * V[4 + col] + mz * V[8 + col] + V[12 + col];
        c.Wc[col][0] = V[col]; c.Wc[col][1] = V[4 + col]; c.Wc[col][2] = V[8 + col];
    }
    const float tz = c.t[2];
    if (a.fovx != nullptr) {
        // (a uniform branch around scalar loads: they count in lgkmcnt, not in the vmcnt the kernels' waits are written for)
        c.tfx = tanf(GHR_UNIFORM(a.fovx)[0] * 0.5f);   // torch.tan(viewpoint_camera.FoVx * 0.5), gaussian_model.py:258-259
        c.tfy = tanf(GHR_UNIFORM(a.fovy)[0] * 0.5f);
        // focal = dim / (2 tan), gaussian_model.py:264-265 / rasterizer_impl.cu:224-225 -- the expression the host evaluates
        c.fx = a.W / (2.0f * c.tfx);
        c.fy = a.H / (2.0f * c.tfy);
    } else {
        c.tfx = a.tan_fovx; c.tfy = a.tan_fovy; c.fx = a.focal_x; c.fy = a.focal_y;
    }
    const float limx = 1.3f * c.tfx, limy = 1.3f * c.tfy;
    const float txtz = c.t[0] / tz, tytz = c.t[1] / tz;
    c.inx = !(txtz < -limx || txtz > limx);
    c.iny = !(tytz < -limy || tytz > limy);
    c.clx = fminf(limx, fmaxf(-limx, txtz));
    c.cly = fminf(limy, fmaxf(-limy, tytz));
    c.txp = c.clx * tz;
    c.typ = c.cly * tz;
    c.j00 = c.fx / tz;
    c.j11 = c.fy / tz;
    c.j20 = -(c.fx * c.txp) / (tz * tz);
    c.j21 = -(c.fy * c.typ) / (tz * tz);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        c.u[i] = c.Wc[0][i] * c.j00 + c.Wc[2][i] * c.j20;
        c.v[i] = c.Wc[1][i] * c.j11 + c.Wc[2][i] * c.j21;
    }
    float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        c.p[i] = c.ax[i][0] * c.u[0] + c.ax[i][1] * c.u[1] + c.ax[i][2] * c.u[2];
        c.r[i] = c.ax[i][0] * c.v[0] + c.ax[i][1] * c.v[1] + c.ax[i][2] * c.v[2];
        const float s2 = c.s[i] * c.s[i];
        sa += s2 * c.p[i] * c.p[i];
        sb += s2 * c.p[i] * c.r[i];
        sc += s2 * c.r[i] * c.r[i];
    }
    c.a = sa + 0.3f;
    c.b = sb;
    c.c = sc + 0.3f;
    c.det = c.a * c.c - c.b * c.b;
    c.k = 1.0f / (c.det + a.conic_eps);
    c.jmax = 0;  // first maximum == argsort(descending)[0] up to ties
    float smax = c.s0[0];
    if (c.s0[1] > smax) { c.jmax = 1; smax = c.s0[1]; }
    if (c.s0[2] > smax) { c.jmax = 2; }
}

// SH coefficient k (0 = DC) of channel ch.  `rest` points at THIS Gaussian's (K-1) x 3 block of features_rest (staged
// in LDS by the kernels, global memory in the host-sim); coefficients beyond K read as 0.
GHR_HD float sh_coeff(const ModelArgs& a, const RawIn& in, const float* rest, int k, int ch)
{
    if (k == 0) return in.dc[ch];
    return k < a.sh_coeffs ? rest[(k - 1) * 3 + ch] : 0.f;
}

// What the forward leaves per Gaussian (k_project stores it itself: the record through an LDS transpose)
struct ProjOut {
    f4 rec[4];      // packed render record (zero when culled)
    float depth;    // view z (culled: not stored)
    int radius;     // 0 when culled
    float ndc[3];   // get_mean_2d values, written for every row
};

// Forward for one Gaussian, nothing stored, in two parts: everything but the colour (needs the raw parameters only -- the
// kernel runs it while the coefficient slab is still on its way), then the colour (needs the higher SH coefficients).
// project_geom returns false when culled (record all zero); the colour is only evaluated for survivors.
GHR_HD bool project_geom(const ModelArgs& a, const RawIn& in, int& x0, int& y0, int& x1, int& y1, ProjOut& o)
{
    o.radius = 0;
    o.depth = 0.f;
    o.rec[0] = o.rec[1] = o.rec[2] = o.rec[3] = f4{0.f, 0.f, 0.f, 0.f};
    ProjCtx c;
    proj_setup(a, in, c);
    const float mx = in.xyz[0], my = in.xyz[1], mz = in.xyz[2];

    // get_mean_2d (gaussian_model.py:332-335); proj is used row-vector style: hom = xyz @ P[:3,:] + P[3,:]
    const uniform_floats pm = GHR_UNIFORM(a.proj);
    const float hx = mx * pm[0] + my * pm[4] + mz * pm[8] + pm[12];
    const float hy = mx * pm[1] + my * pm[5] + mz * pm[9] + pm[13];
    const float hz = mx * pm[2] + my * pm[6] + mz * pm[10] + pm[14];
    const float hw = mx * pm[3] + my * pm[7] + mz * pm[11] + pm[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float ndcx = hx * p_w, ndcy = hy * p_w;
    o.ndc[0] = ndcx; o.ndc[1] = ndcy; o.ndc[2] = hz * p_w;

    // filter_points (gaussian_model.py:166-172) == K1's cull (auxiliary.h:154)
    if (!(c.t[2] > 0.2f)) return false;
    if (c.det == 0.0f) return false;
    // get_conic (gaussian_model.py:311-313)
    const float cx = c.c * c.k, cy = -c.b * c.k, cz = c.a * c.k;
    // K1 mode A from here (forward.cu:242-262)
    const float det_inv = (cx * cz - cy * cy);
    if (det_inv == 0.0f) return false;
    const float det = 1.f / det_inv;
    const float cva = cz * det, cvc = cx * det;
    const float mid = 0.5f * (cva + cvc);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(mid + sq, mid - sq)));
    const float pixx = ndc2pix(ndcx, a.W), pixy = ndc2pix(ndcy, a.H);
    tile_rect(pixx, pixy, (int)my_radius, a.gx, a.gy, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) == 0) return false;

    float label, conf, opac, d2x, d2y;
    if (a.mode == 0) {
        label = sigmoidf_(in.lab);
        conf = expf(in.conf);
        opac = sigmoidf_(in.op);
        const float sj = sel3(c.s0, c.jmax);
        d2x = sj * sel3(c.p, c.jmax);
        d2y = sj * sel3(c.r, c.jmax);
    } else {
        label = in.lab;
        conf = in.conf;
        opac = in.op;
        d2x = 0.f;
        d2y = 0.f;
        if (a.dir3d) {  // normalize(dir) @ T (gaussian_model_strands.py:430-431; F.normalize eps = 1e-12)
            const float dx_ = in.dir[0], dy_ = in.dir[1], dz_ = in.dir[2];
            const float in_ = 1.0f / fmaxf(sqrtf(dx_ * dx_ + dy_ * dy_ + dz_ * dz_), 1e-12f);
            d2x = (dx_ * c.u[0] + dy_ * c.u[1] + dz_ * c.u[2]) * in_;
            d2y = (dx_ * c.v[0] + dy_ * c.v[1] + dz_ * c.v[2]) * in_;
        }
    }

    o.rec[0] = f4{pixx, pixy, cx, cy};
    o.rec[1] = f4{cz, opac, 0.0f, 0.0f};   // colour: project_colour
    o.rec[2] = f4{0.0f, label, 1.0f, d2x};
    o.rec[3] = f4{d2y, 0.0f, conf, c.t[2]};
    o.depth = c.t[2];
    o.radius = (int)my_radius;
    return true;
}

// colours (gaussian_renderer/__init__.py:58-74) into the record of a Gaussian that passed project_geom
GHR_HD void project_colour(const ModelArgs& a, const RawIn& in, const float* rest, ProjOut& o)
{
    const uniform_floats cam = GHR_UNIFORM(a.campos);
    const float dxv = in.xyz[0] - cam[0], dyv = in.xyz[1] - cam[1], dzv = in.xyz[2] - cam[2];
    const float dl = 1.0f / sqrtf(dxv * dxv + dyv * dyv + dzv * dzv);
    float basis[GHR_SH_MAX];
    sh_basis(a.sh_degree, dxv * dl, dyv * dl, dzv * dl, basis);
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < GHR_SH_MAX; k++) acc += basis[k] * sh_coeff(a, in, rest, k, ch);
        rgb[ch] = fmaxf(acc + 0.5f, 0.0f);
    }
    o.rec[1].z = rgb[0];
    o.rec[1].w = rgb[1];
    o.rec[2].x = rgb[2];
}

GHR_HD bool project_core(const ModelArgs& a, const RawIn& in, const float* rest, int& x0, int& y0, int& x1, int& y1,
                         ProjOut& o)
{
    const bool ok = project_geom(a, in, x0, y0, x1, y1, o);
    if (ok) project_colour(a, in, rest, o);
    return ok;
}

// project_core + the stores of one Gaussian (tests/hostsim; k_project stores for itself).  Returns false when culled.
GHR_HD bool project_one(const ModelArgs& a, int idx, const float* rest, int& x0, int& y0, int& x1, int& y1)
{
    const size_t row = (size_t)a.row0 + idx;
    ProjOut o;
    RawIn in;
    load_raw(a, idx, in);
    const bool ok = project_core(a, in, rest, x0, y0, x1, y1, o);
    if (a.means2D) { a.means2D[3 * row] = o.ndc[0]; a.means2D[3 * row + 1] = o.ndc[1]; a.means2D[3 * row + 2] = o.ndc[2]; }
    a.radii[row] = o.radius;
    a.rects[row] = ok ? make_rect4(x0, y0, x1, y1, 0u) : rect4{0u, 0u, 0u, 0u};  // the caller fills in the gradient-slot base
    if (ok) {
        f4* r = a.rec + 4 * row;
        r[0] = o.rec[0]; r[1] = o.rec[1]; r[2] = o.rec[2]; r[3] = o.rec[3];
        a.depths[row] = o.depth;
    }
    return ok;
}

// Backward for one Gaussian: packed rasterizer gradients `ga[16]` (already summed over the Gaussian's tile instances)
// -> raw-parameter gradients.  Writes (or accumulates into) every output element except d_rest, which is handed back in
// the caller's staging block.  Returns whether any value stored was NaN.
// `rest` / `d_rest`: this Gaussian's (K-1) x 3 blocks of features_rest and of its gradient (LDS in the kernel).
// `in` / `radius`: load_raw(a, idx) and radii[row0 + idx] (requested by the kernel long before they are needed).
// In three parts, like the forward: everything that does not touch the higher SH coefficients (the kernel runs it while the
// coefficient slab is still on its way), the SH colour, the stores.
struct ProjBwdOut {
    float dxyz[3], dls[3], dq[4];
    float dlo, dll, dlc;
    float ddc[3], ddir[3];
    float grgb[3];  // dL/d(rgb) behind the colour clamp: every SH gradient of the view is basis_k x grgb (ModelGrads.d_rgb)
};

// CAM: also the Gaussian's camera cotangents into cam[GHR_CAM_PARTIALS] (layout above; campos by project_bwd_sh).
// detach_m2d: the NDC mean is a constant of the graph (no gradient through proj, none to xyz from it).
template <bool CAM>
GHR_HD void project_bwd_geom(const ModelArgs& a, const RawIn& in, int radius, const float* ga, ProjBwdOut& o, float* cam,
                             bool detach_m2d)
{
    float dxyz[3] = {0, 0, 0}, dls[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0};
    float dlo = 0, dll = 0, dlc = 0;
    float ddir[3] = {0, 0, 0};
    const float gmx = detach_m2d ? 0.f : ga[0], gmy = detach_m2d ? 0.f : ga[1];
    if (CAM) {
#pragma unroll
        for (int i = 0; i < GHR_CAM_PARTIALS; i++) cam[i] = 0.f;
    }

    if (radius > 0) {
        ProjCtx c;
        proj_setup(a, in, c);
        const float mx = in.xyz[0], my = in.xyz[1], mz = in.xyz[2];
        const float gA = ga[2], gB = 2.0f * ga[3], gC = ga[4];  // wrapper's [xx, 2*xy, yy] restack (__init__.py:149-153)
        const float gop = ga[5];
        const float* gc = ga + 6;  // colours: rgb 0-2, label 3, one 4, dir2D 5-7, conf 8, depth 9

        // ---- activations (mode 1: the inputs are the activated quantities)
        if (a.mode == 0) {
            const float o = sigmoidf_(in.op);
            dlo = gop * o * (1.f - o);
            const float l = sigmoidf_(in.lab);
            dll = gc[3] * l * (1.f - l);
            dlc = gc[8] * expf(in.conf);
        } else {
            dlo = gop;
            dll = gc[3];
            dlc = gc[8];
        }

        // ---- conic = (c, -b, a) * k, k = 1/(det + eps)
        const float S = gA * c.c - gB * c.b + gC * c.a;
        const float k2S = c.k * c.k * S;
        const float La = gC * c.k - k2S * c.c;
        const float Lc = gA * c.k - k2S * c.a;
        const float Lb = -gB * c.k + 2.f * k2S * c.b;

        // ---- a,b,c = sum_i s_i^2 {p_i^2, p_i r_i, r_i^2};  dir2D = s0_j (p_j, r_j)
        float Lp[3], Lr[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float s2 = c.s[i] * c.s[i];
            Lp[i] = s2 * (2.f * c.p[i] * La + c.r[i] * Lb);
            Lr[i] = s2 * (2.f * c.r[i] * Lc + c.p[i] * Lb);
            // d/d(log s_i): s_i = exp(ls_i) * mod  =>  ds_i/dls_i = s_i
            dls[i] = 2.f * s2 * (c.p[i] * c.p[i] * La + c.p[i] * c.r[i] * Lb + c.r[i] * c.r[i] * Lc);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {  // dir2D = s0_j (p_j, r_j): only the longest axis j receives these terms
            const float on = (a.mode == 0 && i == c.jmax) ? 1.f : 0.f;
            const float sj = c.s0[i] * on;
            Lp[i] += sj * gc[5];
            Lr[i] += sj * gc[6];
            dls[i] += sj * (c.p[i] * gc[5] + c.r[i] * gc[6]);
        }
        // ---- p_i = ax_i . u, r_i = ax_i . v
        float G[3][3], Lu[3] = {0, 0, 0}, Lv[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int m = 0; m < 3; m++) {
                G[i][m] = Lp[i] * c.u[m] + Lr[i] * c.v[m];
                Lu[m] += Lp[i] * c.ax[i][m];
                Lv[m] += Lr[i] * c.ax[i][m];
            }
        if (a.mode != 0) {
            // d/d(log s) -> d/ds for the linear scaling of mode 1 (s = s0 * modifier, s0 > 0)
#pragma unroll
            for (int i = 0; i < 3; i++) dls[i] = dls[i] / c.s0[i];
            if (a.dir3d) {  // dir2D = normalize(dir) . (u, v): cotangents for u, v and for dir (through the normalisation)
                const float dx_ = in.dir[0], dy_ = in.dir[1], dz_ = in.dir[2];
                const float len_ = sqrtf(dx_ * dx_ + dy_ * dy_ + dz_ * dz_);
                const float in_ = 1.0f / fmaxf(len_, 1e-12f);
                const float dh[3] = {dx_ * in_, dy_ * in_, dz_ * in_};
                float Ldh[3];
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    Lu[m] += gc[5] * dh[m];
                    Lv[m] += gc[6] * dh[m];
                    Ldh[m] = gc[5] * c.u[m] + gc[6] * c.v[m];
                }
                if (len_ > 1e-12f) {
                    const float dotp = Ldh[0] * dh[0] + Ldh[1] * dh[1] + Ldh[2] * dh[2];
#pragma unroll
                    for (int m = 0; m < 3; m++) ddir[m] = (Ldh[m] - dh[m] * dotp) * in_;
                } else {
#pragma unroll
                    for (int m = 0; m < 3; m++) ddir[m] = Ldh[m] * in_;
                }
            }
        }
        // ---- rotation: rows of R(q_hat), then the normalisation q_hat = q/|q|
        {
            const float w = c.qn[0], x = c.qn[1], y = c.qn[2], z = c.qn[3];
            float dn[4];
            dn[0] = 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
            dn[1] = 2 * y * (G[0][1] + G[1][0]) + 2 * z * (G[0][2] + G[2][0]) + 2 * w * (G[1][2] - G[2][1]) -
                    4 * x * (G[1][1] + G[2][2]);
            dn[2] = 2 * x * (G[0][1] + G[1][0]) + 2 * w * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) -
                    4 * y * (G[0][0] + G[2][2]);
            dn[3] = 2 * w * (G[0][1] - G[1][0]) + 2 * x * (G[0][2] + G[2][0]) + 2 * y * (G[1][2] + G[2][1]) -
                    4 * z * (G[0][0] + G[1][1]);
            const float dotp = dn[0] * w + dn[1] * x + dn[2] * y + dn[3] * z;
            const float il = 1.0f / c.qlen;
#pragma unroll
            for (int m = 0; m < 4; m++) dq[m] = (dn[m] - c.qn[m] * dotp) * il;
        }
        // ---- u, v -> Jacobian entries -> view-space mean t
        float Lt[3] = {0, 0, 0};
        {
            const float Lj00 = Lu[0] * c.Wc[0][0] + Lu[1] * c.Wc[0][1] + Lu[2] * c.Wc[0][2];
            const float Lj20 = Lu[0] * c.Wc[2][0] + Lu[1] * c.Wc[2][1] + Lu[2] * c.Wc[2][2];
            const float Lj11 = Lv[0] * c.Wc[1][0] + Lv[1] * c.Wc[1][1] + Lv[2] * c.Wc[1][2];
            const float Lj21 = Lv[0] * c.Wc[2][0] + Lv[1] * c.Wc[2][1] + Lv[2] * c.Wc[2][2];
            const float tz = c.t[2], itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
            const float fx = c.fx, fy = c.fy;
            const float Ltxp = Lj20 * (-fx * itz2), Ltyp = Lj21 * (-fy * itz2);
            Lt[2] = Lj00 * (-fx * itz2) + Lj11 * (-fy * itz2) + Lj20 * (2.f * fx * c.txp * itz3) +
                    Lj21 * (2.f * fy * c.typ * itz3);
            // tx' = clamp(tx/tz) * tz (torch.clamp: gradient 1 inside the range, 0 outside)
            Lt[0] = c.inx ? Ltxp : 0.f;
            Lt[1] = c.iny ? Ltyp : 0.f;
            Lt[2] += (c.inx ? 0.f : c.clx * Ltxp) + (c.iny ? 0.f : c.cly * Ltyp);
            Lt[2] += gc[9];  // depth channel = view z
            if (CAM) {
                // W = view[:3,:3] inside T = W @ J (gaussian_model.py:290-292): u_i = W[i][0] j00 + W[i][2] j20, v_i = W[i][1] j11 + W[i][2] j21
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    cam[3 * i + 0] = Lu[i] * c.j00;
                    cam[3 * i + 1] = Lv[i] * c.j11;
                    cam[3 * i + 2] = Lu[i] * c.j20 + Lv[i] * c.j21;
                }
                // focal = dim / (2 tan): d focal / d tan = -focal / tan; the clamp limits 1.3 tan receive the gradient of a
                // clamped tx / tz (torch.clamp with tensor bounds: to `max` where x > max, to `min` = -lim where x < min)
                const float Lfx = Lj00 * itz - Lj20 * (c.txp * itz2);
                const float Lfy = Lj11 * itz - Lj21 * (c.typ * itz2);
                const float Llimx = c.inx ? 0.f : (c.clx > 0.f ? Ltxp * tz : -(Ltxp * tz));
                const float Llimy = c.iny ? 0.f : (c.cly > 0.f ? Ltyp * tz : -(Ltyp * tz));
                cam[24] = 1.3f * Llimx - Lfx * (fx / c.tfx);
                cam[25] = 1.3f * Llimy - Lfy * (fy / c.tfy);
            }
        }
        const uniform_floats V = GHR_UNIFORM(a.view);
#pragma unroll
        for (int row = 0; row < 3; row++) dxyz[row] += V[4 * row] * Lt[0] + V[4 * row + 1] * Lt[1] + V[4 * row + 2] * Lt[2];
        if (CAM) {
            // t = xyz @ view[:3,:3] + view[3,:3] (gaussian_model.py:268; the depth channel is its z, :339-342)
            const float m[3] = {mx, my, mz};
#pragma unroll
            for (int row = 0; row < 3; row++)
#pragma unroll
                for (int col = 0; col < 3; col++) cam[3 * row + col] += m[row] * Lt[col];
#pragma unroll
            for (int col = 0; col < 3; col++) cam[9 + col] = Lt[col];
        }

        // ---- NDC mean
        {
            const uniform_floats pm = GHR_UNIFORM(a.proj);
            const float hx = mx * pm[0] + my * pm[4] + mz * pm[8] + pm[12];
            const float hy = mx * pm[1] + my * pm[5] + mz * pm[9] + pm[13];
            const float hw = mx * pm[3] + my * pm[7] + mz * pm[11] + pm[15];
            const float w_ = 1.0f / (hw + 0.0000001f);
            const float Lhx = gmx * w_, Lhy = gmy * w_, Lhw = -w_ * w_ * (gmx * hx + gmy * hy);
#pragma unroll
            for (int row = 0; row < 3; row++) dxyz[row] += pm[4 * row] * Lhx + pm[4 * row + 1] * Lhy + pm[4 * row + 3] * Lhw;
            if (CAM) {  // p_hom = xyz @ proj[:3,:] + proj[3] (gaussian_model.py:333)
                const float m[3] = {mx, my, mz};
#pragma unroll
                for (int row = 0; row < 3; row++) {
                    cam[12 + 3 * row + 0] = m[row] * Lhx;
                    cam[12 + 3 * row + 1] = m[row] * Lhy;
                    cam[12 + 3 * row + 2] = m[row] * Lhw;
                }
                cam[21] = Lhx; cam[22] = Lhy; cam[23] = Lhw;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { o.dxyz[i] = dxyz[i]; o.dls[i] = dls[i]; o.ddir[i] = ddir[i]; o.ddc[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; i++) o.dq[i] = dq[i];
    o.dlo = dlo; o.dll = dll; o.dlc = dlc;
}

// ---- SH colour (clamp_min(sh + .5, 0)) incl. the view-direction dependence on xyz (added to o.dxyz behind the geometry
// terms).  d_rest may alias rest: every element is read (cf) before it is overwritten, channel by channel.
template <bool CAM>
GHR_HD void project_bwd_sh(const ModelArgs& a, const RawIn& in, int radius, const float* ga, const float* rest, float* d_rest,
                           ProjBwdOut& o, float* cam)
{
    const int K = a.sh_coeffs;
    // (the camera-centre cotangents as three SSA values stored ONCE behind the branch: stored into cam[] on both paths the
    // compiler sank the stores into one with a selected ADDRESS, and the array went to scratch -- 32 B per lane in both CAM
    // instantiations until round 6)
    float cpg0 = 0.f, cpg1 = 0.f, cpg2 = 0.f;
    if (radius > 0) {
        const float mx = in.xyz[0], my = in.xyz[1], mz = in.xyz[2];
        const float* gc = ga + 6;  // colours: rgb 0-2
        float* dxyz = o.dxyz;
        float* ddc = o.ddc;
        {
            const uniform_floats cpos = GHR_UNIFORM(a.campos);
            const float dxv = mx - cpos[0], dyv = my - cpos[1], dzv = mz - cpos[2];
            const float len = sqrtf(dxv * dxv + dyv * dyv + dzv * dzv), il = 1.0f / len;
            const float x = dxv * il, y = dyv * il, z = dzv * il;
            float basis[GHR_SH_MAX], vk[GHR_SH_MAX];
            sh_basis(a.sh_degree, x, y, z, basis);
#pragma unroll
            for (int k = 0; k < GHR_SH_MAX; k++) vk[k] = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                float cf[GHR_SH_MAX];
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < GHR_SH_MAX; k++) {
                    cf[k] = sh_coeff(a, in, rest, k, ch);
                    acc += basis[k] * cf[k];
                }
                const float gch = (acc + 0.5f >= 0.0f) ? gc[ch] : 0.f;  // clamp_min backward: grad where x >= min
                o.grgb[ch] = gch;
                ddc[ch] = basis[0] * gch;
#pragma unroll
                for (int k = 0; k < GHR_SH_MAX; k++) {
                    if (k > 0 && k < K) d_rest[(k - 1) * 3 + ch] = basis[k] * gch;
                    vk[k] += gch * cf[k];
                }
            }
            float Ld[3];
            sh_basis_grad_dot(a.sh_degree, x, y, z, vk, Ld[0], Ld[1], Ld[2]);
            const float dotp = Ld[0] * x + Ld[1] * y + Ld[2] * z;  // d(normalize)
            const float dd[3] = {(Ld[0] - x * dotp) * il, (Ld[1] - y * dotp) * il, (Ld[2] - z * dotp) * il};
#pragma unroll
            for (int m = 0; m < 3; m++) dxyz[m] += dd[m];
            cpg0 = -dd[0]; cpg1 = -dd[1]; cpg2 = -dd[2];  // dir = xyz - camera_center (gaussian_renderer/__init__.py:59)
        }
    } else {
        for (int k = 0; k < 3 * (K - 1); k++) d_rest[k] = 0.f;
        o.grgb[0] = o.grgb[1] = o.grgb[2] = 0.f;
    }
    if (CAM) { cam[26] = cpg0; cam[27] = cpg1; cam[28] = cpg2; }
}

// Writes (or accumulates into) every output element except d_rest, which stays in the caller's staging block.  Returns
// whether any value stored was non-finite.
// The per-array step size lr / (1 - beta1^t) and sqrt(1 - beta2^t) of the fused update, t = the group's own step number: the
// expressions of k_adam (ghr_adam.h), evaluated by the lane whose index is the array's.
GHR_HD void adam_fuse_coef(const AdamFuse& f, int arr, float& ss, float& b2)
{
    const int grp = f.group[arr];
    const int step = f.state[0] + 1 - f.state[2 + grp];
    const double bias1 = 1.0 - pow(f.beta1, (double)step);
    ss = (float)((double)f.lr[arr] / bias1);
    b2 = (float)sqrt(1.0 - pow(f.beta2, (double)step));
}

GHR_HD bool project_bwd_store(const ModelArgs& a, const ModelGrads& g, int idx, const float* ga, const ProjBwdOut& o,
                              int radius, const RawIn* in = nullptr, const float* ss = nullptr, const float* b2 = nullptr)
{
    if (g.cam_only) return false;  // a frozen segment: only its camera cotangents are wanted
    const int acc = g.accumulate;
    const size_t row = (size_t)a.row0 + idx;
    g.d_means2D[3 * row] = ga[0];
    g.d_means2D[3 * row + 1] = ga[1];
    g.d_means2D[3 * row + 2] = 0.f;
    if (g.dens_grad_accum != nullptr && radius > 0 &&
        (g.dens_count == nullptr || *reinterpret_cast<const volatile uint32_t*>(g.dens_count) <= g.dens_cap)) {
        // torch.norm(grad[:, :2], dim=-1) on this build: sqrt(x * x + y * y) with two roundings before the add (ATen reduces the
        // two-element rows as a vector, one square per lane, then adds) -- tools/gpu/norm_probe.py: 0 of 733 815 rows differ;
        // the fused-multiply-add forms differ in 8 % of the rows by one ulp.  This TU is compiled -ffp-contract=off.
        g.dens_grad_accum[idx] += sqrtf(ga[0] * ga[0] + ga[1] * ga[1]);
        g.dens_denom[idx] += 1.0f;
        g.dens_max_radii[idx] = fmaxf(g.dens_max_radii[idx], (float)radius);
    }
    // the values in output order; accumulating, ALL old values are requested before the first is needed (element by
    // element -- read, add, store -- the 22 of them were 22 dependent round trips per Gaussian in every view but the first)
    float v[22];
    float* p[22];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        v[i] = o.dxyz[i];          p[i] = g.d_xyz + 3 * idx + i;
        v[3 + i] = o.dls[i];       p[3 + i] = g.d_log_scales + 3 * idx + i;
        v[13 + i] = o.ddir[i];     p[13 + i] = g.d_dir3d ? g.d_dir3d + 3 * idx + i : nullptr;
        v[16 + i] = o.ddc[i];      p[16 + i] = g.d_features_dc ? g.d_features_dc + 3 * (size_t)idx + i : nullptr;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { v[6 + i] = o.dq[i]; p[6 + i] = g.d_rotations + 4 * idx + i; }
    v[10] = o.dlo; p[10] = g.d_opacity_logit ? g.d_opacity_logit + idx : nullptr;
    v[11] = o.dll; p[11] = g.d_label_logit ? g.d_label_logit + idx : nullptr;
    v[12] = o.dlc; p[12] = g.d_orient_conf_log ? g.d_orient_conf_log + idx : nullptr;
    const int n = 19;
    if (acc) {
        float old[19];
#pragma unroll
        for (int i = 0; i < n; i++) old[i] = p[i] ? *p[i] : 0.f;
#pragma unroll
        for (int i = 0; i < n; i++) v[i] += old[i];
    }
    bool bad = false;
    if (in != nullptr && g.adam.on && a.mode == 1) {
        // ---- strand segment (round 6): of this Gaussian's raw values only the SH features are parameters of the optimizer --
        // position, scale, rotation, direction, confidence are functions of the strand polylines, their gradients go on
        // through autograd (ghr_strand_build_backward) and are stored as usual.  The DC colour is updated here, the higher
        // bands by slab_out_adam.
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (p[i]) {
                *p[i] = v[i];
                bad |= nonfinite(v[i]);
            }
        const AdamFuse& f = g.adam;
        const float w1 = (float)(1.0 - f.beta1), w2 = (float)(1.0 - f.beta2), bt2 = (float)f.beta2;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const ptrdiff_t off = (a.features_dc + 3 * (size_t)idx + i) - f.p_base;
            float pp = in->dc[i], mm = f.m_in[off], vv = f.v_in[off];
            const float gv = v[16 + i];
            bad |= nonfinite(gv);
            adam_update(pp, gv, mm, vv, ss[6], w1, bt2, w2, f.eps, b2[6]);
            f.p_out[off] = pp;
            f.m_out[off] = mm;
            f.v_out[off] = vv;
        }
        return bad;
    }
    if (in != nullptr && g.adam.on) {
        // ---- the optimizer update instead of the gradient stores (mode 0: sixteen values in six arrays + the DC colour).  The
        // gradient is what the store below would have left in the flat buffer; the parameter is the raw value this thread loaded
        // at the top of the kernel; m and v come from the `in` set, everything goes to the `out` set (ghr_adam.h).
        const AdamFuse& f = g.adam;
        const float w1 = (float)(1.0 - f.beta1), w2 = (float)(1.0 - f.beta2), bt2 = (float)f.beta2;
        // value index -> (array, parameter pointer, parameter value)
        const float* q[16];
        float pv[16], gv[16];
        int arr[16];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            q[i] = a.xyz + 3 * idx + i;                 pv[i] = in->xyz[i];       gv[i] = v[i];        arr[i] = 0;
            q[3 + i] = a.log_scales + 3 * idx + i;      pv[3 + i] = in->ls[i];    gv[3 + i] = v[3 + i]; arr[3 + i] = 1;
            q[13 + i] = a.features_dc + 3 * (size_t)idx + i; pv[13 + i] = in->dc[i]; gv[13 + i] = v[16 + i]; arr[13 + i] = 6;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { q[6 + i] = a.rotations + 4 * idx + i; pv[6 + i] = in->q[i]; gv[6 + i] = v[6 + i]; arr[6 + i] = 2; }
        q[10] = a.opacity_logit + idx;   pv[10] = in->op;   gv[10] = v[10]; arr[10] = 3;
        q[11] = a.label_logit + idx;     pv[11] = in->lab;  gv[11] = v[11]; arr[11] = 4;
        q[12] = a.orient_conf_log + idx; pv[12] = in->conf; gv[12] = v[12]; arr[12] = 5;
#pragma unroll
        for (int h = 0; h < 16; h += 8) {  // eight values at a time: sixteen loads in flight, then their updates and stores
            float mm[8], vv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const ptrdiff_t off = q[h + i] - f.p_base;
                mm[i] = f.m_in[off];
                vv[i] = f.v_in[off];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const ptrdiff_t off = q[h + i] - f.p_base;
                float pp = pv[h + i];
                bad |= nonfinite(gv[h + i]);
                adam_update(pp, gv[h + i], mm[i], vv[i], ss[arr[h + i]], w1, bt2, w2, f.eps, b2[arr[h + i]]);
                f.p_out[off] = pp;
                f.m_out[off] = mm[i];
                f.v_out[off] = vv[i];
            }
        }
        return bad;
    }
#pragma unroll
    for (int i = 0; i < n; i++)
        if (p[i]) {
            *p[i] = v[i];
            bad |= nonfinite(v[i]);
        }
    if (g.d_rgb != nullptr) {  // the view's SH gradients in factored form (ASSIGNED: one table per view)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            g.d_rgb[3 * (size_t)idx + i] = o.grgb[i];
            bad |= nonfinite(o.grgb[i]);
        }
    }
    return bad;
}

GHR_HD bool project_bwd_core(const ModelArgs& a, const ModelGrads& g, int idx, const RawIn& in, int radius, const float* ga,
                             const float* rest, float* d_rest, float* cam = nullptr)
{
    ProjBwdOut o;
    if (cam) {
        project_bwd_geom<true>(a, in, radius, ga, o, cam, g.detach_means2D != 0);
        project_bwd_sh<true>(a, in, radius, ga, rest, d_rest, o, cam);
    } else {
        project_bwd_geom<false>(a, in, radius, ga, o, nullptr, g.detach_means2D != 0);
        project_bwd_sh<false>(a, in, radius, ga, rest, d_rest, o, nullptr);
    }
    return project_bwd_store(a, g, idx, ga, o, radius);
}

// project_bwd_core with its inputs loaded on the spot (tests/hostsim)
GHR_HD bool project_bwd_one(const ModelArgs& a, const ModelGrads& g, int idx, const float* ga, const float* rest,
                            float* d_rest, float* cam = nullptr)
{
    RawIn in;
    load_raw(a, idx, in);
    return project_bwd_core(a, g, idx, in, a.radii[(size_t)a.row0 + idx], ga, rest, d_rest, cam);
}

// features_rest is [P, K-1, 3]: one thread's 3(K-1) floats are contiguous but 180 B apart from its neighbour's, so
// direct per-thread loads touch 64 cache lines per instruction.  A block's slab (256 x 3(K-1) floats) IS contiguous
// and 16-B aligned: move it through LDS with coalesced b128 accesses; per-thread LDS reads at an odd stride (45) are
// conflict-free.
#define GHR_REST_MAX (3 * (GHR_SH_MAX - 1))  // 45

#if defined(__HIP_DEVICE_COMPILE__)
#define GHR_SLAB_IT ((GHR_BLOCK * GHR_REST_MAX / 4 + GHR_BLOCK - 1) / GHR_BLOCK)  // 12 b128 accesses per thread
// Global -> LDS without a stop in registers (round 5): thread t's 16-B pieces t, t + 256, ... go straight to their place
// (`global_load_lds_dwordx4`: 1 KB contiguous per wave and instruction on both sides), all (up to 12) in flight together
// and counted in vmcnt.  The 48 VGPRs the pieces used to wait in are what lets the kernels hold a Gaussian's raw
// parameters from the first instruction on (RawIn) at the same occupancy.  The caller waits (slab_wait) before the barrier
// that publishes the slab.
template <int BLK>
__device__ __forceinline__ void slab_dma(float* dst, const float* src, size_t n_floats, int tid)
{
    // src starts 16-B aligned (256 * 3(K-1) * 4 bytes per block is a multiple of 16)
    const uint32_t n4 = (uint32_t)(n_floats / 4);
    const int wave0 = tid & ~63;
#pragma unroll
    for (int it = 0; it < GHR_SLAB_IT; it++) {
        const uint32_t i = tid + BLK * it;
        if (i < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (size_t)i),
                                             (__attribute__((address_space(3))) void*)(dst + 4 * (wave0 + BLK * it)),
                                             16, 0, 2 /* nt: read once */);
    }
    // (the scalar tail of a partial last block)
    for (size_t i = 4 * (size_t)n4 + tid; i < n_floats; i += BLK) dst[i] = src[i];
}
__device__ __forceinline__ void slab_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Global -> registers -> LDS (k_project): ALL of a thread's (up to 12) loads are issued back to back; a rolled
// `d4[i] = s4[i]` loop compiles to load / s_waitcnt vmcnt(0) / ds_write per trip, i.e. 12 serialized HBM round trips.
__device__ __forceinline__ void slab_load(f4 (&v)[GHR_SLAB_IT], const float* src, size_t n_floats, int tid)
{
    // No predication: pieces past the end re-read the last one (dropped by slab_to_lds).  A load under `if (i < n4)` is a
    // branch around it, and behind loads that may or may not have been issued the compiler can only wait with vmcnt(0) --
    // for the caller's OLDER loads too, which it wants to use while these are in flight.  n_floats >= 4.
    const uint32_t n4 = (uint32_t)(n_floats / 4);
    const f4* s4 = reinterpret_cast<const f4*>(src);
#pragma unroll
    for (int it = 0; it < GHR_SLAB_IT; it++) {
        const uint32_t i = tid + GHR_BLOCK * it;
        v[it] = __builtin_nontemporal_load(s4 + (i < n4 ? i : n4 - 1u));
    }
}
__device__ __forceinline__ void slab_to_lds(float* dst, const f4 (&v)[GHR_SLAB_IT], const float* src, size_t n_floats,
                                            int tid)
{
    const uint32_t n4 = (uint32_t)(n_floats / 4);
    f4* d4 = reinterpret_cast<f4*>(dst);
#pragma unroll
    for (int it = 0; it < GHR_SLAB_IT; it++) {
        const uint32_t i = tid + GHR_BLOCK * it;
        if (i < n4) d4[i] = v[it];
    }
    for (size_t i = 4 * (size_t)n4 + tid; i < n_floats; i += GHR_BLOCK) dst[i] = src[i];
}
// LDS gradient slab -> global, assigning or accumulating; returns whether a stored value was NaN
template <int BLK>
__device__ __forceinline__ bool slab_out(float* dst, const float* src, size_t n_floats, int tid, int accumulate)
{
    const uint32_t n4 = (uint32_t)(n_floats / 4);
    const f4* s4 = reinterpret_cast<const f4*>(src);
    f4* d4 = reinterpret_cast<f4*>(dst);
    bool bad = false;
    f4 old[GHR_SLAB_IT];
    if (accumulate) {  // the read half of the read-modify-write, all loads in flight together (see slab_load)
#pragma unroll
        for (int it = 0; it < GHR_SLAB_IT; it++) {
            const uint32_t i = tid + BLK * it;
            if (i < n4) old[it] = d4[i];
        }
    }
#pragma unroll
    for (int it = 0; it < GHR_SLAB_IT; it++) {
        const uint32_t i = tid + BLK * it;
        if (i < n4) {
            f4 v = s4[i];
            if (accumulate) v += old[it];
            d4[i] = v;
            bad |= nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w);
        }
    }
    for (size_t i = 4 * n4 + tid; i < n_floats; i += BLK) {
        float v = src[i];
        if (accumulate) v += dst[i];
        dst[i] = v;
        bad |= nonfinite(v);
    }
    return bad;
}
// slab_out for the fused optimizer update: the gradient slab in LDS (+ what earlier views left in the flat gradient buffer
// when `accumulate`) meets p, m, v of the `in` set, 16 B at a time, and the updated values go to the `out` set.  `dst_grad` /
// `param`: the block's pieces of d_features_rest / features_rest.  Returns whether a gradient value was non-finite.
template <int BLK>
__device__ __forceinline__ bool slab_out_adam(const float* dst_grad, const float* param, const float* src, size_t n_floats,
                                              int tid, int accumulate, const AdamFuse& f, float ss, float b2)
{
    const uint32_t n4 = (uint32_t)(n_floats / 4);
    const f4* s4 = reinterpret_cast<const f4*>(src);
    const f4* g4 = reinterpret_cast<const f4*>(dst_grad);
    const ptrdiff_t off0 = param - f.p_base;
    const f4 *p4 = reinterpret_cast<const f4*>(param), *m4 = reinterpret_cast<const f4*>(f.m_in + off0),
             *v4 = reinterpret_cast<const f4*>(f.v_in + off0);
    f4 *po = reinterpret_cast<f4*>(f.p_out + off0), *mo = reinterpret_cast<f4*>(f.m_out + off0),
       *vo = reinterpret_cast<f4*>(f.v_out + off0);
    const float w1 = (float)(1.0 - f.beta1), w2 = (float)(1.0 - f.beta2), bt2 = (float)f.beta2;
    bool bad = false;
    constexpr int HALF = GHR_SLAB_IT / 2;
#pragma unroll
    for (int h = 0; h < GHR_SLAB_IT; h += HALF) {  // six pieces at a time: 18 (24) 16-B loads in flight
        f4 P[HALF], M[HALF], V[HALF], O[HALF];
#pragma unroll
        for (int it = 0; it < HALF; it++) {
            const uint32_t i = tid + BLK * (h + it);
            const uint32_t ic = i < n4 ? i : 0u;
            P[it] = __builtin_nontemporal_load(p4 + ic);
            M[it] = __builtin_nontemporal_load(m4 + ic);
            V[it] = __builtin_nontemporal_load(v4 + ic);
            if (accumulate) O[it] = g4[ic];
        }
#pragma unroll
        for (int it = 0; it < HALF; it++) {
            const uint32_t i = tid + BLK * (h + it);
            if (i < n4) {
                f4 G = s4[i];
                if (accumulate) G += O[it];
                bad |= nonfinite(G.x) | nonfinite(G.y) | nonfinite(G.z) | nonfinite(G.w);
                float pp[4] = {P[it].x, P[it].y, P[it].z, P[it].w}, mm[4] = {M[it].x, M[it].y, M[it].z, M[it].w},
                      vv[4] = {V[it].x, V[it].y, V[it].z, V[it].w};
                const float gg[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
                for (int e = 0; e < 4; e++) adam_update(pp[e], gg[e], mm[e], vv[e], ss, w1, bt2, w2, f.eps, b2);
                __builtin_nontemporal_store(f4{pp[0], pp[1], pp[2], pp[3]}, po + i);
                __builtin_nontemporal_store(f4{mm[0], mm[1], mm[2], mm[3]}, mo + i);
                __builtin_nontemporal_store(f4{vv[0], vv[1], vv[2], vv[3]}, vo + i);
            }
        }
    }
    for (size_t i = 4 * (size_t)n4 + tid; i < n_floats; i += BLK) {  // (the scalar tail of a partial last block)
        float G = src[i];
        if (accumulate) G += dst_grad[i];
        bad |= nonfinite(G);
        float pp = param[i], mm = f.m_in[off0 + i], vv = f.v_in[off0 + i];
        adam_update(pp, G, mm, vv, ss, w1, bt2, w2, f.eps, b2);
        f.p_out[off0 + i] = pp; f.m_out[off0 + i] = mm; f.v_out[off0 + i] = vv;
    }
    return bad;
}
#endif

// REST: the model has SH coefficients beyond the DC term (a template parameter, not a test of sh_coeffs: see load_raw)
template <bool REST>
__global__ void __launch_bounds__(GHR_BLOCK) k_project(ModelArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float s_rest[GHR_BLOCK * GHR_REST_MAX];
    const int row = REST ? 3 * (a.sh_coeffs - 1) : 0;
    const int base = blockIdx.x * GHR_BLOCK;
    const int nb = min(GHR_BLOCK, a.P - base);
    const int idx = base + threadIdx.x;
    // Issue order = return order: the raw parameters first, the coefficient slab behind them (registers, not slab_dma:
    // 50.5 us against 54.3 for this kernel, which has the registers to spare: profiles/r05o).  Everything but the colour --
    // cull, conic, radius, tile rect -- is computed while the slab is on its way, and the counting atomics (they hand out
    // the instances' list positions: count_tiles; results needed at the very end) go out before the slab is even waited for.
    RawIn in;
    load_raw(a, min(idx, a.P - 1), in);
    f4 v[GHR_SLAB_IT];
    if (REST) slab_load(v, a.features_rest + (size_t)base * row, (size_t)nb * row, threadIdx.x);
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    ProjOut o;
    const bool ok = idx < a.P && project_geom(a, in, x0, y0, x1, y1, o);
    TileCountPending tc;
    count_tiles_issue(a.tile_count, a.gx, x0, y0, x1, y1, tc);
    if (REST) slab_to_lds(s_rest, v, a.features_rest + (size_t)base * row, (size_t)nb * row, threadIdx.x);
    __syncthreads();
    if (ok) project_colour(a, in, s_rest + threadIdx.x * row, o);
    __shared__ uint32_t s_scan[4];
    uint32_t blk_total;
    const uint32_t slot0 = block_excl_scan_256(ok ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u, s_scan, &blk_total);
    // (the scan's barrier is also the one behind which nobody reads the coefficient slab any more)
    // From here on the kernel only stores.  Stores count in vmcnt like loads: the counting atomics' results, collected at the
    // very end, would wait for the acknowledgement of every store below (the phase profile's "atomics collected": 10 % of a
    // wave's time).  They returned long ago -- under the slab's round trip and the colour arithmetic -- so they are taken in
    // here, with a wait that costs nothing (0x0f70 = vmcnt(0), expcnt / lgkmcnt untouched), and the kernel's tail is stores only.
    __builtin_amdgcn_s_waitcnt(0x0f70);
    // The record is 64 B per Gaussian: stored by its own thread it is four 16-B pieces at a 64-B stride per instruction (2.47 M
    // partial write requests at 500k Gaussians, profiles/r02a).  Through LDS instead: thread t stores the 16-B pieces
    // t, t + 256, ... of the workgroup's 16 KB of records -- 1 KB contiguous per wave and instruction.  Culled rows get zeros.
    f4* s_rec = reinterpret_cast<f4*>(s_rest);
    if (idx < a.P) {
        const size_t rowi = (size_t)a.row0 + idx;
#pragma unroll
        for (int q = 0; q < 4; q++) s_rec[4 * threadIdx.x + q] = o.rec[q];
        if (a.means2D) { a.means2D[3 * rowi] = o.ndc[0]; a.means2D[3 * rowi + 1] = o.ndc[1]; a.means2D[3 * rowi + 2] = o.ndc[2]; }
        a.radii[rowi] = o.radius;
        // (one 16-B store with the gradient-slot base inside the workgroup already in place; .w comes from k_scatter)
        rect4 r = rect4{0u, 0u, 0u, 0u};
        if (ok) { r = make_rect4(x0, y0, x1, y1, 0u); r.z = slot0; }
        a.rects[rowi] = r;
        if (ok) a.depths[rowi] = o.depth;
    }
    if (threadIdx.x == 0) a.slot_blk[blockIdx.x + (a.row0 >> 8)] = blk_total;
    __syncthreads();
    {
        f4* dst = a.rec + 4 * ((size_t)a.row0 + base);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = threadIdx.x + GHR_BLOCK * k;
            if (j < 4 * nb) dst[j] = s_rec[j];
        }
    }
    __syncthreads();  // the record stores have read the staging area: it now serves the big rects' expansion
    count_tiles_finish(a.tile_count, (uint32_t)(a.gx * a.gy), a.gx, x0, y0, x1, y1, *reinterpret_cast<BigRects*>(s_rest),
                       a.pos + (size_t)GHR_BIG_RECT * ((size_t)a.row0 + (idx < a.P ? idx : 0)), tc);
#endif
}

// One WAVE per workgroup (round 5): nothing in this kernel needs more than its own 64 Gaussians, the LDS budget (11.5 KB of
// coefficient slab per wave) and the register budget (134 VGPRs: three waves per SIMD) allow the same 12 waves per CU as
// 256-thread workgroups did, and single waves are dispatched as soon as any slot frees up instead of four at a time:
// 79.3 -> 76.5 us (128 threads: 77.5; forcing 128 VGPRs for a fourth wave per SIMD spills and loses: profiles/r05o).
#define GHR_PBW_BLOCK 64
#if defined(__HIP_DEVICE_COMPILE__)
// Sum of each of the N (a power of two <= 32) values v[.] over the 64 lanes of the wave in N - 1 + (6 - log2 N) exchanges instead of
// 6 N: at every halving step a lane keeps one half of its values and hands the other half to its partner (lane ^ 1, 2, 4, ...),
// so the number of live values halves while the number of lanes summed doubles; the remaining steps are plain butterflies.
// Returns, on every lane, the wave's total of component cam_butterfly_component<N>(lane).  The order of the additions is fixed.
template <int N>
__device__ __forceinline__ float cam_butterfly(float (&v)[N], int lane)
{
    constexpr int LOG = N == 32 ? 5 : (N == 16 ? 4 : (N == 8 ? 3 : (N == 4 ? 2 : 1)));
    static_assert((1 << LOG) == N, "N must be 2, 4, 8, 16 or 32");
#pragma unroll
    for (int s = 0; s < LOG; s++) {  // (a canonical loop: fully unrolled, every v[.] index a constant -- registers, not scratch)
        const int n = N >> (s + 1);
        const bool hi = (lane >> s) & 1;
#pragma unroll
        for (int i = 0; i < n; i++) {
            // (both values pinned in registers first: left to itself the compiler selects between the two ADDRESSES and the
            // array moves to scratch)
            float lo_v = v[i], hi_v = v[i + n];
            asm volatile("" : "+v"(lo_v), "+v"(hi_v));
            const float keep = hi ? hi_v : lo_v;
            const float send = hi ? lo_v : hi_v;
            v[i] = keep + __shfl_xor(send, 1 << s);
        }
    }
    float r = v[0];
#pragma unroll
    for (int w = N; w < 64; w <<= 1) r += __shfl_xor(r, w);
    return r;
}
// lane bit s chose the half of size N >> (s + 1): the component is the bit-reversed low log2(N) bits of the lane
template <int N>
__device__ __forceinline__ int cam_butterfly_component(int lane)
{
    int c = 0;
#pragma unroll
    for (int s = 0; (N >> (s + 1)) >= 1; s++) c |= ((lane >> s) & 1) ? (N >> (s + 1)) : 0;
    return c;
}
#endif

// CAM: the camera cotangents as well (ModelGrads::cam_partial); the default instantiation carries none of it.
#if defined(__HIP_DEVICE_COMPILE__)
template <bool CAM, bool ADAM>
__device__ __forceinline__ void project_bwd_body(const ModelArgs& a, const ModelGrads& g)
{
    constexpr int BLK = GHR_PBW_BLOCK;
    __shared__ __attribute__((aligned(16))) float s_rest[BLK * GHR_REST_MAX];  // coefficients in, gradients out
    const int row = 3 * (a.sh_coeffs - 1);
    const int base = blockIdx.x * BLK;
    const int nb = min(BLK, a.P - base);
    const int idx = base + threadIdx.x;
    // ONE first round trip for everything that does not depend on something loaded (round 5; the phase profile of round 4's
    // form, profiles/r05o, showed seven dependent trips per workgroup: rect -> record -> line numbers -> lines, then inside the
    // compute phase radius -> raw parameters -> activations): the rect, the record k_project wrote (zeros for culled rows, so
    // no need to know the rect first), the raw parameters, the radius.  Two more trips follow: the instances' line numbers
    // and the lines.  The coefficient slab (46 KB per workgroup, the bulk of the kernel's reads) is requested BEHIND them,
    // straight into LDS, and only waited for in front of the SH part: vmcnt retires in order, so requested first it made the
    // small dependent trips wait for the big transfer, and the geometry backward (two thirds of the arithmetic) needs none of it.
    const int idc = min(idx, a.P - 1);
    const size_t rowc = (size_t)a.row0 + idc;
    rect4 r = a.rects[rowc];
    f4 q0 = a.rec[4 * rowc], q1 = a.rec[4 * rowc + 1];  // pixel mean / conic / opacity
    RawIn in;
    load_raw(a, idc, in);
    const int radius = a.radii[rowc];
    if (idx >= a.P) r = make_rect4(0, 0, 0, 0, 0u);
    // ADAM: the per-array coefficients of the fused update, one array per lane, under the first round trip of the loads above
    float ss[GHR_ADAM_FUSE_ARRAYS], b2[GHR_ADAM_FUSE_ARRAYS];
    if constexpr (ADAM) {
        float ss_l = 0.f, b2_l = 0.f;
        if ((threadIdx.x & 63) < GHR_ADAM_FUSE_ARRAYS) adam_fuse_coef(g.adam, threadIdx.x & 63, ss_l, b2_l);
#pragma unroll
        for (int k = 0; k < GHR_ADAM_FUSE_ARRAYS; k++) { ss[k] = __shfl(ss_l, k); b2[k] = __shfl(b2_l, k); }
    }
    float ga[16];
    gather_inst_grads_wave(g.ginst, g.inst_line, r, q0, q1, 0.5f * a.W, 0.5f * a.H, ga, g.ginst_rows);
    // (LDS-DMA, not registers: with the slab's 48 registers on top this kernel needs 180 VGPRs -- two waves per SIMD -- or
    // spills at 168: 109 us against 89.6, profiles/r05o)
    // (everything requested so far has arrived -- the gather used it -- but the compiler cannot know on every path: told
    // here, with a wait that costs nothing, or it would protect the first use of `radius` below with a vmcnt(0) that
    // waits for the slab.  0x0f70 = vmcnt(0), expcnt / lgkmcnt untouched.)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (row > 0) slab_dma<BLK>(s_rest, a.features_rest + (size_t)base * row, (size_t)nb * row, threadIdx.x);
    ProjBwdOut o;
    float cam[CAM ? GHR_CAM_PARTIALS : 1];
    if (CAM) {
#pragma unroll
        for (int i = 0; i < (CAM ? GHR_CAM_PARTIALS : 1); i++) cam[i] = 0.f;
    }
    if (idx < a.P) project_bwd_geom<CAM>(a, in, radius, ga, o, cam, g.detach_means2D != 0);
    float cam_pos[4] = {0.f, 0.f, 0.f, 0.f};
    float cam_geo = 0.f;
    if constexpr (CAM) {
        // the 26 cotangents of the geometry part are summed over the wave HERE, while the coefficient slab is still on its way
        // (the exchanges are LDS-crossbar operations: lgkmcnt, not the vmcnt the slab is waited for with), so that only the
        // wave total and the three camera-centre terms of the SH part stay live to the end of the kernel
        // (in place: project_bwd_sh writes cam[26..28] afterwards, nothing else of cam[] is read again)
        cam_geo = cam_butterfly<GHR_CAM_PARTIALS>(cam, threadIdx.x & 63);
    }
    slab_wait();
    __syncthreads();
    bool bad = false;
    if (CAM) cam[CAM ? 26 : 0] = cam[CAM ? 27 : 0] = cam[CAM ? 28 : 0] = 0.f;   // (lanes past the end of the segment)
    if (idx < a.P) {
        project_bwd_sh<CAM>(a, in, radius, ga, s_rest + threadIdx.x * row, s_rest + threadIdx.x * row, o, cam);
        if constexpr (ADAM) bad = project_bwd_store(a, g, idx, ga, o, radius, &in, ss, b2);
        else bad = project_bwd_store(a, g, idx, ga, o, radius);
    }
    __syncthreads();
    if constexpr (ADAM) {
        if (row > 0)
            bad |= slab_out_adam<BLK>(g.d_features_rest + (size_t)base * row, a.features_rest + (size_t)base * row, s_rest,
                                      (size_t)nb * row, threadIdx.x, g.accumulate, g.adam, ss[7], b2[7]);
    } else if (row > 0 && !g.cam_only && g.d_features_rest != nullptr)
        bad |= slab_out<BLK>(g.d_features_rest + (size_t)base * row, s_rest, (size_t)nb * row, threadIdx.x, g.accumulate);
    if (g.overflow_is_bad && g.dens_count != nullptr && *reinterpret_cast<const volatile uint32_t*>(g.dens_count) > g.dens_cap)
        bad = true;
    if (g.nan_flag != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(g.nan_flag, 1);
    if constexpr (CAM) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < 3; i++) cam_pos[i] = cam[26 + i];
        const float pos_tot = cam_butterfly<4>(cam_pos, lane);
        float* col = g.cam_partial + g.cam_slot0 + blockIdx.x;
        if (lane < GHR_CAM_PARTIALS) {
            const int c = cam_butterfly_component<GHR_CAM_PARTIALS>(lane);
            if (c < 26 || c >= 29) col[(size_t)c * g.cam_stride] = cam_geo;   // (rows 29..31: zeros)
        } else if (lane < GHR_CAM_PARTIALS + 4) {
            const int c = cam_butterfly_component<4>(lane);
            if (c < 3) col[(size_t)(26 + c) * g.cam_stride] = pos_tot;
        }
    }
}
#endif

// ADAM: the optimizer update instead of the gradient stores (ModelGrads::adam; mode 0 only)
template <bool CAM, bool ADAM>
__global__ void __launch_bounds__(GHR_PBW_BLOCK) k_project_bwd(ModelArgs a, ModelGrads g);
template <>
__global__ void __launch_bounds__(GHR_PBW_BLOCK) k_project_bwd<false, false>(ModelArgs a, ModelGrads g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    project_bwd_body<false, false>(a, g);
#endif
}
// (the LDS footprint allows three waves per SIMD either way: told so, the compiler takes the 168 registers that go with them
// instead of holding these instantiations to the default one's 134 and spilling)
template <>
__global__ void __launch_bounds__(GHR_PBW_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_project_bwd<true, false>(ModelArgs a, ModelGrads g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    project_bwd_body<true, false>(a, g);
#endif
}
template <>
__global__ void __launch_bounds__(GHR_PBW_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_project_bwd<false, true>(ModelArgs a, ModelGrads g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    project_bwd_body<false, true>(a, g);
#endif
}
template <>
__global__ void __launch_bounds__(GHR_PBW_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_project_bwd<true, true>(ModelArgs a, ModelGrads g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    project_bwd_body<true, true>(a, g);
#endif
}

// ---- SH gradients from their per-view factors (round 6: the data-parallel gradient message) ------------------------------------
// A view's gradient of the SH coefficients of Gaussian i is the outer product basis(dir_i) (x) d_rgb_i (project_bwd_sh): 48
// floats that are determined by 3 -- the view direction is a function of the camera centre and xyz, which every rank holds.
// Ranks therefore exchange d_rgb per view (12 B per Gaussian and view, all-gather) instead of summing 192 B per Gaussian
// (all-reduce), and every rank rebuilds  d sh[k][c] = sum_v basis_k(dir_{v,i}) d_rgb_{v,i}[c]  here, views in list order: the
// products are the ones project_bwd_sh forms and the sum is taken in the order a single rank accumulating the same views takes
// it, so the result has that run's bits.  A view in which the Gaussian has no gradient (culled: d_rgb = 0) is skipped -- its
// direction may not even be defined (a Gaussian at the camera centre).
GHR_HD void sh_grad_from_views_one(int deg, int K, const float* xyz3, int n_views, const float* campos, size_t campos_stride,
                                   const float* g, size_t view_stride, size_t idx, float* dc, float* rest)
{
    float acc[3 * GHR_SH_MAX];
#pragma unroll
    for (int i = 0; i < 3 * GHR_SH_MAX; i++) acc[i] = 0.f;
    const float mx = xyz3[0], my = xyz3[1], mz = xyz3[2];
    for (int v = 0; v < n_views; v++) {
        const float* gv = g + (size_t)v * view_stride + 3 * idx;
        const float g0 = gv[0], g1 = gv[1], g2 = gv[2];
        if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;  // (a NaN compares unequal: it is carried through)
        const float* cpos = campos + campos_stride * v;
        const float dxv = mx - cpos[0], dyv = my - cpos[1], dzv = mz - cpos[2];  // as project_bwd_sh
        const float len = sqrtf(dxv * dxv + dyv * dyv + dzv * dzv), il = 1.0f / len;
        const float x = dxv * il, y = dyv * il, z = dzv * il;
        float basis[GHR_SH_MAX];
        sh_basis(deg, x, y, z, basis);
        const float gg[3] = {g0, g1, g2};
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
#pragma unroll
            for (int k = 0; k < GHR_SH_MAX; k++)
                if (k < K) acc[3 * k + ch] = acc[3 * k + ch] + basis[k] * gg[ch];
    }
    dc[0] = acc[0]; dc[1] = acc[1]; dc[2] = acc[2];
#pragma unroll
    for (int k = 1; k < GHR_SH_MAX; k++)
        if (k < K) { rest[3 * (k - 1)] = acc[3 * k]; rest[3 * (k - 1) + 1] = acc[3 * k + 1]; rest[3 * (k - 1) + 2] = acc[3 * k + 2]; }
}

struct ShViewsArgs {
    int P, sh_degree, sh_coeffs, n_views;
    const float* xyz;      // [P,3]
    const float* campos;   // view v: 3 floats at campos + v * campos_stride (device)
    size_t campos_stride;  // floats
    const float* g;        // view v: [P,3] at g + v * view_stride
    size_t view_stride;    // floats
    int* nan_flag;         // optional: raised when the float at g + v * view_stride + flag_offset of any view is not zero --
    long long flag_offset; // the ranks' own non-finite flags, carried by the gathered rows (saves a 4-byte all-reduce)
    float* d_dc;           // [P,1,3]
    float* d_rest;         // [P,K-1,3]
    int accumulate;        // != 0: added to what the two arrays hold; 0: assigned
};

__global__ void __launch_bounds__(GHR_PBW_BLOCK) k_sh_grad_from_views(ShViewsArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BLK = GHR_PBW_BLOCK;
    __shared__ __attribute__((aligned(16))) float s_rest[BLK * GHR_REST_MAX];
    const int row = 3 * (a.sh_coeffs - 1);
    const int base = blockIdx.x * BLK;
    const int nb = min(BLK, a.P - base);
    const int idx = base + threadIdx.x;
    if (a.nan_flag != nullptr && blockIdx.x == 0)
        for (int v = threadIdx.x; v < a.n_views; v += BLK)
            if (a.g[(size_t)v * a.view_stride + (size_t)a.flag_offset] != 0.f) atomicOr(a.nan_flag, 1);
    if (idx < a.P) {
        float dc[3];
        sh_grad_from_views_one(a.sh_degree, a.sh_coeffs, a.xyz + 3 * (size_t)idx, a.n_views, a.campos, a.campos_stride, a.g,
                               a.view_stride, (size_t)idx, dc, s_rest + threadIdx.x * row);
        float* o = a.d_dc + 3 * (size_t)idx;
        if (a.accumulate) { dc[0] += o[0]; dc[1] += o[1]; dc[2] += o[2]; }
        o[0] = dc[0]; o[1] = dc[1]; o[2] = dc[2];
    }
    __syncthreads();
    if (row > 0) slab_out<BLK>(a.d_rest + (size_t)base * row, s_rest, (size_t)nb * row, threadIdx.x, a.accumulate);
#endif
}

// Adds up the per-workgroup camera partials (component-major [GHR_CAM_PARTIALS][n_slots]) in a fixed order, in double, and
// writes d_cam[GHR_CAM_GRADS] = d view[16] | d proj[16] | d camera_center[3] | d tanfov[2] (entries the projection never
// reads -- column 3 of the view matrix, column 2 of the projection matrix -- are written as zeros).  One workgroup per component.
// fovx / fovy (both or neither): the last two entries become dL/dFoVx, dL/dFoVy: d tan(FoV / 2) / d FoV = (1 + tan^2) / 2.
#define GHR_CAM_FOLD_BLOCK 1024
__global__ void __launch_bounds__(GHR_CAM_FOLD_BLOCK) k_cam_fold(const float* partial, uint32_t n_slots, float* d_cam,
                                                                 const float* fovx, const float* fovy)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ double s_sum[GHR_CAM_FOLD_BLOCK / 64];
    const int comp = blockIdx.x;
    const float* p = partial + (size_t)comp * n_slots;
    double acc = 0.0;
    // eight loads in flight per thread and trip (a rolled load - convert - add loop is one dependent round trip per element)
    for (uint32_t base = 0; base < n_slots; base += 8u * GHR_CAM_FOLD_BLOCK) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t i = base + (uint32_t)j * GHR_CAM_FOLD_BLOCK + threadIdx.x;
            v[j] = p[i < n_slots ? i : n_slots - 1u];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t i = base + (uint32_t)j * GHR_CAM_FOLD_BLOCK + threadIdx.x;
            acc += i < n_slots ? (double)v[j] : 0.0;
        }
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) acc += __shfl_xor(acc, w);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < GHR_CAM_FOLD_BLOCK / 64; w++) tot += s_sum[w];
        float v = (float)tot;
        if (comp < 12) d_cam[4 * (comp / 3) + comp % 3] = v;
        else if (comp < 24) { const int c = (comp - 12) % 3; d_cam[16 + 4 * ((comp - 12) / 3) + (c == 2 ? 3 : c)] = v; }
        else if (comp < 26) {
            if (fovx != nullptr) {
                const float t = tanf((comp == 24 ? fovx[0] : fovy[0]) * 0.5f);
                v = v * (0.5f * (1.0f + t * t));
            }
            d_cam[35 + comp - 24] = v;
        } else if (comp < 29) d_cam[32 + comp - 26] = v;
        else if (comp == 29) {
#pragma unroll
            for (int r = 0; r < 4; r++) { d_cam[4 * r + 3] = 0.f; d_cam[16 + 4 * r + 2] = 0.f; }
        }
    }
#endif
}

}  // namespace ghr
