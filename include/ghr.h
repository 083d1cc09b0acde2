/*
 * ghr.h -- C ABI of libghr_hip.so, the MI355X (gfx950) strand-aligned Gaussian tile rasterizer.
 *
 * Drop-in boundary for the reference's native module `diff_gaussian_rasterization._C`
 * (ext/diff_gaussian_rasterization_hair/ext.cpp:15-19; signatures rasterize_points.h:18-69):
 *
 *   _C.rasterize_gaussians           (rasterize_points.cu:35-123)   -> ghr_forward_stage1 + ghr_forward_stage2
 *   _C.rasterize_gaussians_backward  (rasterize_points.cu:125-206)  -> ghr_backward
 *   _C.mark_visible                  (rasterize_points.cu:208-227)  -> ghr_mark_visible
 *   CudaRasterizer::required<State>  (rasterizer_impl.h:67-73)      -> ghr_forward_sizes / ghr_binning_size
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.  All pointers are DEVICE pointers unless the
 *    name ends in `_host`.  `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *  - Nullable pointers select the mode exactly like the reference's empty-tensor => nullptr convention
 *    (diff_gaussian_rasterization/__init__.py:210-222; forward.cu:215,228):
 *        conic_precomp != NULL            : "pipeline mode" (A): conic supplied by the caller (what render() does)
 *        conic_precomp == NULL            : "kernel-geometry mode" (B): cov3D_precomp, or scales+rotations, in-kernel
 *  - Matrices are the reference's: 16 floats, element [4*c + r] multiplies component c into output r
 *    (auxiliary.h:58-77), i.e. world_view_transform / full_proj_transform of src/scene/cameras.py:72-80.
 *  - Feature channels: `colors` is [P, C] row-major, C == GHR_NUM_CHANNELS (config.h:15).  Like the reference
 *    (rasterizer_impl.cu:244-247) the library refuses to run without precomputed colors.
 *  - Every function returns 0 on success, a negative GHR_E_* code on failure; ghr_last_error() returns a
 *    thread-local description.  No exceptions cross the ABI.  Kernels never trap: a Gaussian that fails the near
 *    test is culled even when `prefiltered` is set (the reference __trap()s, auxiliary.h:154-162).
 *  - Ownership: the caller owns every buffer (workspaces included) and must keep geom/img/bin workspaces alive
 *    and unmodified between forward and backward of the same view (the reference does this with
 *    ctx.save_for_backward, __init__.py:102).  The library allocates nothing and never synchronises, except that
 *    `debug != 0` makes each call hipStreamSynchronize + check errors before returning (auxiliary.h:166-173).
 */
#ifndef GHR_H
#define GHR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GHR_ABI_VERSION 20
#define GHR_NUM_CHANNELS 10 /* R:cuda_rasterizer/config.h:15 */
#define GHR_TILE 16         /* R:cuda_rasterizer/config.h:16-17 (BLOCK_X, BLOCK_Y) */
#define GHR_ADAM_STATE 18  /* ints of the fused Adam's device state */
#define GHR_GRAD_STRIDE 16  /* floats per Gaussian-tile instance in the gradient scratch of ghr_backward */
#define GHR_CAM_PARTIALS 32 /* rows of the camera-gradient partial table (ghr_model_args.cam_partial) */
#define GHR_CAM_GRADS 37    /* floats ghr_camera_grad_fold writes: d view[16] | d proj[16] | d camera_center[3] | d tanfov[2] */
#define GHR_STRAND_MAX_SEG 2048 /* longest strand (segments) ghr_strand_build takes */

#define GHR_OK 0
#define GHR_E_INVALID (-1)  /* bad argument (NULL where required, C != GHR_NUM_CHANNELS, ...) */
#define GHR_E_NOCOLORS (-2) /* "For non-RGB, provide precomputed Gaussian colors!" */
#define GHR_E_HIP (-3)      /* a HIP runtime call or kernel launch failed */

/* Inputs shared by forward and backward of one view (argument lists of rasterize_points.h:18-69). */
typedef struct ghr_view_args {
    int32_t P;                  /* number of Gaussians handed to the op (after the Python-side mask) */
    int32_t W, H;               /* image_width, image_height */
    int32_t C;                  /* feature channels; must be GHR_NUM_CHANNELS */
    const float* background;    /* [C] */
    const float* means3D;       /* [P,3] */
    const float* colors;        /* [P,C] colors_precomp */
    const float* opacities;     /* [P] (tensor [P,1]) */
    const float* scales;        /* [P,3] or NULL */
    const float* rotations;     /* [P,4] (r,x,y,z), NOT normalised in-kernel (forward.cu:127), or NULL */
    const float* cov3D_precomp; /* [P,6] or NULL */
    const float* conic_precomp; /* [P,3] (a, b, c) or NULL */
    const float* viewmatrix;    /* [16] */
    const float* projmatrix;    /* [16] */
    float scale_modifier;
    float tan_fovx, tan_fovy;
    int32_t prefiltered;        /* accepted for API parity; culling is always silent */
    int32_t debug;              /* != 0: synchronise + check after the call */
    /* ABI 17 (ghr_forward_stage1 only).  != 0: img_ws is the image workspace of an EARLIER forward pass of the same W x H whose
     * stage 1 AND stage 2 ran (P > 0), ordered before this call, untouched since: its per-tile counters are back at zero and
     * stage 1 skips its zero-fill (the promise of ghr_model_args.img_ws_recycled, for the rasterizer op itself).  0: any buffer. */
    int32_t img_ws_recycled;
} ghr_view_args;

const char* ghr_last_error(void);
int ghr_abi_version(void);

/* Workspace sizes (host only).  geom: per-Gaussian state (packed 64-B render records, depths, tile rects,
 * [mode B: cov3D]); img: per-pixel final_T / n_contrib + per-tile counts, list offsets and the largest n_contrib of
 * each 4x4-pixel cell. */
int ghr_forward_sizes(int32_t P, int32_t W, int32_t H, int32_t mode_b, size_t* geom_bytes, size_t* img_bytes);
/* bin: per-instance (Gaussian x tile) sort keys + the sorted point list + (ABI 12) the forward pass's cell masks:
 * per 64 list positions of a tile and per 4x4-pixel cell, which entries can touch the cell (2 B per instance + 128 B
 * per tile, hence W and H). */
int ghr_binning_size(uint32_t R, int32_t W, int32_t H, size_t* bin_bytes);

/* Stage 1 = preprocess (K1) + per-tile instance count + tile offset scan.  Writes radii[P] (int32, 0 = culled)
 * and asynchronously copies num_rendered R to *R_host (pinned host memory; valid once `stream` reaches the
 * point after this call, e.g. after hipStreamSynchronize -- the reference blocks on the same 4 bytes,
 * rasterizer_impl.cu:284-285). */
int ghr_forward_stage1(void* stream, const ghr_view_args* a, void* geom_ws, void* img_ws, int32_t* radii,
                       uint32_t* R_host);

/* Stage 2 = instance scatter + per-tile depth sort (== the reference's global (tile|depth) stable radix sort,
 * rasterizer_impl.cu:293-321) + front-to-back compositing (K7).  out_color is [C,H,W].
 * R is the CAPACITY (in instances) of bin_ws (ghr_binning_size(R)) and fixes its layout: it must be >= the instance
 * count stage 1 reported for the result to be valid, and the same value must be passed to the backward call.  A caller
 * may therefore launch stage 2 speculatively with a capacity guessed from the previous frame BEFORE reading *R_host
 * (no GPU bubble behind the host round trip) and relaunch it only if the true count turned out larger: instances
 * beyond the capacity are dropped without touching memory outside bin_ws.  Stage 2 may be replayed.
 * grad_scratch (ABI 12, may be NULL): the scratch the backward call over this state will be given (>= R lines).  Its
 * lines are then zeroed here, by the render kernel as its last act, so that the backward call can be told `prezeroed` (below). */
int ghr_forward_stage2(void* stream, const ghr_view_args* a, uint32_t R, void* geom_ws, void* img_ws, void* bin_ws,
                       float* out_color, float* grad_scratch);

/* Backward (K8 + K9 + K10).  dL_dpix is [C,H,W].  R: the capacity stage 2 was run with (layout of bin_ws).
 * grad_scratch: GHR_GRAD_STRIDE floats (one 64-B gradient line) per Gaussian-tile instance actually reported by stage 1
 * (may be NULL when that count is 0): every line is zero-filled and accumulated by K8 and the lines of a Gaussian are
 * summed in a fixed order -- no cross-tile float atomics.
 * prezeroed (ABI 13; the library keeps NO record of buffers -- it is stateless between calls): 0 = grad_scratch is
 * uninitialised, K8 zero-fills it.  != 0 = the CALLER guarantees that grad_scratch is exactly as
 * ghr_forward_stage2(..., bin_ws, ..., grad_scratch) of THIS state left it: same scratch, zeroed by that stage 2,
 * and nothing written to it since -- in particular no other backward call (of this or any other state).  Sharing
 * rule: when several states share one scratch buffer, at most the state whose stage 2 was the LAST to be handed the
 * buffer may be backwarded with prezeroed != 0, and only as the first backward that touches the buffer after it; every
 * other backward over the shared buffer passes 0.  A second backward over the same state passes 0.  A caller that launched stage 2
 * speculatively and has not read the count yet passes R lines instead: no line index >= R is ever touched, so a count
 * above the capacity (whose results the caller must discard and recompute) cannot write outside the buffer.
 * Outputs (all fully written, no pre-zeroing needed), shapes of rasterize_points.cu:160-168:
 *   dL_dmeans2D [P,3] (z = 0), dL_dconic [P,2,2] ([0][0], [0][1] = HALF of d/db as in backward.cu:554, [1][1]),
 *   dL_dopacity [P], dL_dcolors [P,C], dL_dmeans3D [P,3], dL_dcov3D [P,6], dL_dscales [P,3], dL_drotations [P,4]. */
int ghr_backward(void* stream, const ghr_view_args* a, uint32_t R, const int32_t* radii, const void* geom_ws,
                 const void* img_ws, const void* bin_ws, const float* dL_dpix, float* grad_scratch,
                 float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D,
                 float* dL_dcov3D, float* dL_dscales, float* dL_drotations, int32_t prezeroed);
/* ABI 14: ghr_backward with one more (optional) output, dL_dconic3 [P,3] = (dL_dconic[0][0], 2 * dL_dconic[0][1],
 * dL_dconic[1][1]): the gradient w.r.t. conic_precomp as the reference's Python wrapper hands it to autograd
 * (diff_gaussian_rasterization/__init__.py:149-153 restacks dL_dconic with three slices, a doubling and a stack: two more
 * kernels per backward pass of the drop-in op).  NULL: exactly ghr_backward. */
int ghr_backward_ex(void* stream, const ghr_view_args* a, uint32_t R, const int32_t* radii, const void* geom_ws,
                    const void* img_ws, const void* bin_ws, const float* dL_dpix, float* grad_scratch,
                    float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D,
                    float* dL_dcov3D, float* dL_dscales, float* dL_drotations, int32_t prezeroed, float* dL_dconic3);

/* ---- fused model path ------------------------------------------------------------------------
 * One kernel computes, from the RAW parameters of the reference's GaussianModel, everything render() would build with
 * ~60 PyTorch kernels before calling the rasterizer (src/gaussian_renderer/__init__.py:29-83):
 *   activations (gaussian_model.py:107-141), get_conic (:230-315), get_mean_2d / get_depths (:317-342),
 *   get_direction_2d (:344-393), eval_sh + clamp (sh_utils.py:57-112), the 10-channel feature vector, filter_points
 *   (:143-228) -- and feeds K1's cull / radius / tile rect directly.  ghr_model_backward runs K8 and then maps the
 *   packed per-Gaussian gradients to raw-parameter gradients (what autograd does for that graph), in one pass.
 * Semantics are the PYTHON pipeline's (torch.clamp / normalize / clamp_min gradients, conic eps), tolerance 1e-4. */
struct ghr_adam_fuse;
typedef struct ghr_model_args {
    int32_t P, W, H;
    int32_t sh_degree;            /* active SH degree 0..3 */
    int32_t sh_coeffs;            /* K = (max_sh_degree + 1)^2: 1, 4, 9 or 16 (anything else: GHR_E_INVALID) */
    const float* xyz;             /* [P,3]   _xyz */
    const float* log_scales;      /* [P,3]   _scaling (exp activation) */
    const float* rotations;       /* [P,4]   _rotation, raw (normalised in-kernel like build_rotation) */
    const float* opacity_logit;   /* [P]     _opacity (sigmoid) */
    const float* label_logit;     /* [P]     _label (sigmoid) */
    const float* orient_conf_log; /* [P]     _orient_conf (exp) */
    const float* features_dc;     /* [P,1,3] */
    const float* features_rest;   /* [P,K-1,3] */
    const float* viewmatrix;      /* [16] */
    const float* projmatrix;      /* [16] */
    const float* campos;          /* [3] */
    const float* background;      /* [C] */
    float scale_modifier, tan_fovx, tan_fovy;
    float conic_eps;              /* 1e-12 (gaussian_model.py:312); strand stage 1e-7 (gaussian_model_strands.py:355) */
    int32_t debug;
    /* ---- explicit linear-space Gaussians: what render_hair() feeds the rasterizer in the strand stage
     * (src/gaussian_renderer/__init__.py:116-214).  mode 1: log_scales holds the scaling, opacity_logit / label_logit /
     * orient_conf_log the ACTIVATED values (NULL = the constant below), dir3d the strand direction whose normalised
     * projection normalize(dir) @ T is the 2D direction (NULL = zero direction: frozen head Gaussians).
     * row0: first workspace row of this segment (ghr_model_forward_segment); a multiple of 256. */
    int32_t mode;
    int32_t row0;
    const float* dir3d;           /* [P,3] or NULL */
    float const_opacity, const_label, const_conf;
    /* ABI 16.  != 0: img_ws is the image workspace of an EARLIER forward pass of the same W x H whose stage 1 AND stage 2 were
     * both launched, ordered before this call on the stream (or through events), and which nobody has written since.  Its
     * per-tile counters are then back at zero -- k_tile_scan turns the counts into append cursors starting at 0 and stage 2's
     * tile sort resets every cursor -- so stage 1 does not launch its zero-fill (one ~5-us launch per view).  0: any buffer.
     * A pass with P == 0 does NOT qualify (both stages return before touching the workspace).  With `debug` set the library
     * reads the counters back and fails with GHR_E_INVALID if any is non-zero; otherwise the promise is taken on trust. */
    int32_t img_ws_recycled;
    /* ---- ABI 17: trainable cameras.  The reference's projection graph is differentiable w.r.t. the camera
     * (src/scene/gaussian_model.py:258-266,279-294,332-335; src/gaussian_renderer/__init__.py:59), whose tensors are functions
     * of trainable pose / FoV residuals (src/scene/cameras.py:85-151) stepped by an optimizer of their own
     * (src/train_gaussians.py:45-60,183-196; on by default: src/arguments/__init__.py:61-62).
     * fovx_dev, fovy_dev (both or neither): DEVICE scalars FoVx, FoVy in radians.  Non-NULL: forward and backward kernels take
     * tan(FoV / 2) from there (and derive focal = dim / (2 tan) themselves) instead of tan_fovx / tan_fovy above, so a FoV that
     * changes every step never costs a device-to-host read; ghr_camera_grad_fold can then return dL/dFoV directly.
     * cam_partial (backward calls only; NULL = no camera gradients): table of GHR_CAM_PARTIALS rows of cam_slots floats.  The
     * segment's backward writes one column per 64 Gaussians -- columns cam_slot0 .. cam_slot0 + ghr_camera_slots(P) - 1, every
     * row of them -- and ghr_camera_grad_fold adds the columns up.  Several segments of one view share a table.
     * cam_only != 0: a frozen segment (render_hair()'s head Gaussians): its backward writes NOTHING but its camera columns (the
     * parameter-gradient and d_means2D pointers may be NULL).
     * detach_means2D != 0: the segment's NDC means are constants of the graph (render_hair() detaches the head's,
     * src/gaussian_renderer/__init__.py:136): no gradient flows through projmatrix for it. */
    const float* fovx_dev;
    const float* fovy_dev;
    float* cam_partial;
    int32_t cam_slot0, cam_slots;
    int32_t cam_only;
    int32_t detach_means2D;
    /* Per-iteration densification statistics of the stage-1 loop (src/train_gaussians.py:161-165,
     * src/scene/gaussian_model.py:739-741), folded into the backward call (all three non-NULL, or all NULL): for every Gaussian
     * of the segment the view sees (radii > 0)   xyz_gradient_accum += |d_means2D.xy|,   denom += 1,
     * max_radii2D = max(max_radii2D, radii).  All [P] floats (the reference's [P,1], [P,1], [P]).  Read-modify-write: calls that
     * share the arrays must be ordered (they are wherever they share the flat gradient buffer). */
    float* dens_grad_accum;
    float* dens_denom;
    float* dens_max_radii2D;
    /* optional with the three above: the view's image workspace.  The update then only happens if the instance count stage 1 left
     * there is within the capacity R the backward call is given -- a view rasterized speculatively with too small a capacity
     * (ghr_forward_stage2) has invalid gradients and is recomputed by the caller: its statistics must not be counted twice. */
    const void* dens_img_ws;
    /* != 0 with dens_img_ws (statistics pointers may be NULL): a view whose instance count exceeds the capacity R raises nan_flag
     * -- what every view of a step whose LAST backward carries adam_fuse must do (see ghr_adam_fuse). */
    int32_t overflow_raises_flag;
    /* ---- the optimizer update fused into this backward call (mode 0, the LAST backward of a single-rank gradient step): see
     * ghr_adam_fuse below.  NULL: gradients are stored / accumulated as documented. */
    const struct ghr_adam_fuse* adam_fuse;
    /* ---- ABI 19 (backward calls, mode 0 or 1; NULL = off): d_rgb [P,3], ASSIGNED: dL/d(rgb) of every Gaussian behind the colour
     * clamp (sh_utils.py:57-112 + clamp_min(.. + 0.5, 0), src/gaussian_renderer/__init__.py:58-63).  This view's gradient of the
     * SH coefficients is the outer product basis_k(dir) x d_rgb[c] -- 48 floats determined by 3 and by the view direction, which
     * any holder of xyz and the camera centre can recompute: ghr_sh_grad_from_views.  With d_rgb set, d_features_dc and
     * d_features_rest of the call may be NULL (nothing is stored for them); non-finite d_rgb raises nan_flag. */
    float* d_rgb;
} ghr_model_args;

/* Adam fused into the projection backward (src/scene/gaussian_model.py:431-444 stepped at src/train_gaussians.py:174-181).
 * The last view's backward of a step holds every parameter gradient of the step in registers -- its own terms plus what the
 * earlier views accumulated in the gradient buffers it is given with accumulate != 0 -- and the raw parameters too: with
 * adam_fuse set it applies ghr_adam_step's update there and stores NO gradients (the gradient buffers are left as they were:
 * 244 B per Gaussian neither written nor read back by a separate optimizer pass).
 * Two sets of flat buffers of n floats each: the raw-parameter pointers of ghr_model_args must point into p_in (all eight
 * arrays; together they must tile it: the reference's eight parameter groups), m_in / v_in are the moments at the same offsets;
 * the updated p, m, v go to p_out / m_out / v_out at the same offsets, and the caller swaps the roles of the two sets for the
 * next step.  The skip-on-non-finite rule is global: nan_flag of the backward call must be `flag`; after the projection kernel
 * the call launches a second kernel that, when *flag != 0, copies in -> out (the update is undone: 732 B per Gaussian, a rare
 * event) and otherwise advances state[0]; it clears *flag_next, never *flag -- alternate two words between steps (every view of
 * a step raises the same word).  With dens_img_ws set, a view whose instance count exceeds the capacity R raises the flag as
 * well (a speculative forward pass that overflowed must not reach the parameters).
 * state / n_groups / group_end_host / lr_host / beta1 / beta2 / eps: as ghr_adam_step (no group may be marked to sit out). */
typedef struct ghr_adam_fuse {
    int64_t n;
    const float* p_in;
    const float* m_in;
    const float* v_in;
    float* p_out;
    float* m_out;
    float* v_out;
    int32_t* state;
    int32_t* flag;
    int32_t* flag_next;
    int32_t n_groups;
    const int64_t* group_end_host;
    const float* lr_host;
    double beta1, beta2;
    float eps;
} ghr_adam_fuse;
/* ABI 19: a strand segment (mode 1, render_hair(): src/train_strands.py:98-160) may carry the update too, as the step's single
 * view (accumulate == 0).  Of its raw values only features_dc / features_rest are parameters of the optimizer's flat buffer
 * (the other groups there -- strand directions, confidence: gaussian_model_strands.py:578-589 -- receive their gradients through
 * autograd, after this call): the kernel updates those two arrays into the `out` set, stores every other gradient as usual
 * (d_features_dc / d_features_rest may be NULL) and launches NO finish kernel.  The caller then (1) adds the non-finite mark
 * of the late gradients to *flag (ghr_adam_nan_scan with state = flag - 1), (2) brings the remaining ranges of the `out` set up
 * to date (ghr_adam_step_range_to with flag, ABI 20; or: copy in -> out, ghr_adam_step_range on the `out` buffers with
 * nan_guard = 2 and state[1] = *flag), (3) calls
 * ghr_adam_fused_finish: the same k_adam_fused_finish as above (out := in for everything when the flag is up, step counter). */
int ghr_adam_fused_finish(void* stream, const ghr_adam_fuse* adam_fuse);

/* Stage 1 of the fused path: replaces ghr_forward_stage1 (then call ghr_forward_stage2 with a ghr_view_args that
 * carries P, W, H, C and background; every other field may be NULL).  means2D_out [P,3] (NDC) may be NULL. */
int ghr_model_forward_stage1(void* stream, const ghr_model_args* m, void* geom_ws, void* img_ws, int32_t* radii,
                             float* means2D_out, uint32_t* R_host);

/* Backward of the fused path.  Outputs: d_means2D [P,3] (always assigned), d_xyz [P,3], d_log_scales [P,3],
 * d_rotations [P,4], d_opacity_logit [P], d_label_logit [P], d_orient_conf_log [P], d_features_dc [P,1,3],
 * d_features_rest [P,K-1,3].  accumulate == 0: every parameter-gradient element is assigned; != 0: added to what the
 * buffers hold (they may be the optimizer's own flat gradient buffer: no separate accumulation pass).
 * nan_flag (nullable, device int): set to 1 when any stored parameter-gradient value is NaN (the stage-1 loop's NaN
 * guard, src/train_gaussians.py:174-181, without a scan over the gradients). */
/* ---- segmented form (strand stage): several ghr_model_args segments, each covering rows row0 .. row0+P-1 of ONE
 * rasterizer state of rows_total rows (workspaces sized with ghr_forward_sizes(rows_total, ...); radii / means2D_out
 * [rows_total(,3)]).  Call ghr_model_forward_segment for every segment (first != 0 on the first), then
 * ghr_model_forward_finish (tile scan + *R_host), then ghr_forward_stage2 with P = rows_total as usual.  Rows between
 * the end of a segment and the next multiple of 256 are culled padding.
 * radii / means2D_out (and d_means2D below) are the caller's per-row arrays and are indexed by workspace row, row0 + i: a call
 * touches the rows of ITS segment only (plus, in radii, the zero-fill of the padding rows behind it), so a caller that wants
 * its outputs without the padding -- the reference's [head rows, strand rows] indexing -- may hand a later segment base
 * pointers displaced by the padding in front of it (round 6: the Python side does; the three arrays are then written compact,
 * earlier segments first).  The same displaced `radii` must be passed to that segment's ghr_model_backward_segment. */
int ghr_model_forward_segment(void* stream, const ghr_model_args* m, int32_t rows_total, int32_t first, void* geom_ws,
                              void* img_ws, int32_t* radii, float* means2D_out);
int ghr_model_forward_finish(void* stream, int32_t rows_total, int32_t W, int32_t H, int32_t debug, void* geom_ws,
                             void* img_ws, uint32_t* R_host);
/* Backward in two steps: K8 over the whole state (rows_total rows), then the per-Gaussian chain for the segments that
 * need gradients.  d_means2D is [rows_total,3]; the parameter gradients are per segment ([P,...]); in mode 1
 * d_log_scales / d_opacity_logit / d_label_logit / d_orient_conf_log are the gradients of the linear quantities and,
 * like d_dir3d, may be NULL.  grad_rows: number of lines in grad_scratch (0: R), see ghr_backward.  bin_ws / R: the binning
 * workspace and the capacity ghr_forward_stage2 ran with (ABI 12: the gradient lines lie in tile-list order, the per-Gaussian
 * chain finds a Gaussian's lines through the index the tile sort left in bin_ws). */
int ghr_render_backward(void* stream, int32_t rows_total, int32_t W, int32_t H, uint32_t R, const float* background,
                        const void* geom_ws, const void* img_ws, const void* bin_ws, const float* dL_dpix,
                        float* grad_scratch, int32_t prezeroed /* see ghr_backward */);
int ghr_model_backward_segment(void* stream, const ghr_model_args* m, int32_t rows_total, const int32_t* radii,
                               const void* geom_ws, const float* grad_scratch, float* d_means2D, float* d_xyz,
                               float* d_log_scales, float* d_rotations, float* d_opacity_logit, float* d_label_logit,
                               float* d_orient_conf_log, float* d_features_dc, float* d_features_rest, float* d_dir3d,
                               int32_t accumulate, int32_t* nan_flag, uint32_t grad_rows, const void* bin_ws,
                               uint32_t R);

/* ABI 17.  Columns of ghr_model_args.cam_partial a segment of P Gaussians fills (host only). */
int32_t ghr_camera_slots(int32_t P);
/* Adds the cam_slots columns of a partial table up (fixed order, double accumulation) into d_cam[GHR_CAM_GRADS]:
 * dL/d world_view_transform [4,4] (column 3 zero: never read) | dL/d full_proj_transform [4,4] (column 2 zero: the NDC z carries
 * no gradient) | dL/d camera_center [3] | dL/d {tan(FoVx / 2), tan(FoVy / 2)} -- the gradients autograd hands the camera tensors
 * in the reference's render() (torch.clamp's tensor bounds 1.3 tan included).  fovx_dev / fovy_dev (both or neither; the
 * pointers of ghr_model_args): the last two entries are dL/dFoVx, dL/dFoVy instead (x (1 + tan^2(FoV / 2)) / 2). */
int ghr_camera_grad_fold(void* stream, const float* cam_partial, int32_t cam_slots, float* d_cam, const float* fovx_dev,
                         const float* fovy_dev);

/* ABI 19.  SH-coefficient gradients from per-view factors: the data-parallel step's gradient message (an all-reduce
 * of 244 B per Gaussian over xGMI) is 79 % SH gradients, and those are rank-1 per view.  Ranks all-gather d_rgb (12 B per Gaussian
 * and view, ghr_model_args.d_rgb) and their camera centres instead, and every rank calls this:
 *   d_features_dc[i][c]      = sum_v basis_0 d_rgb[v][i][c]
 *   d_features_rest[i][k][c] = sum_v basis_{k+1}(normalize(xyz_i - campos_v)) d_rgb[v][i][c]
 * (both ASSIGNED, or added to what the arrays hold with accumulate != 0), views in list order, the products and the order of the sum being those of one rank accumulating the same views
 * (ghr_model_backward with accumulate != 0): that run's bits.  campos and g_views are DEVICE arrays: view v's camera centre is
 * the 3 floats at campos + v * campos_stride, its [P,3] table starts at g_views + v * view_stride (strides in floats; both may
 * point into the same gathered rows).  sh_degree = the active degree (bands above it get zeros), sh_coeffs = K.
 * nan_flag (optional) with flag_offset: the float at g_views + v * view_stride + flag_offset of every view is that view's
 * owner's "my gradients are not finite" mark (non-zero = raised); any raised mark raises *nan_flag -- the ranks' skip-the-step
 * flags (ghr_adam_step, nan_guard = 2) travel inside the gathered rows instead of through a collective of their own. */
int ghr_sh_grad_from_views(void* stream, int32_t P, int32_t sh_degree, int32_t sh_coeffs, const float* xyz, int32_t n_views,
                           const float* campos, int64_t campos_stride, const float* g_views, int64_t view_stride,
                           float* d_features_dc, float* d_features_rest, int32_t accumulate, int32_t* nan_flag,
                           int64_t flag_offset);

/* ABI 18.  Strand polylines -> one Gaussian per segment: initialize_gaussians_hair (src/scene/gaussian_model_strands.py:435-452,
 * the same lines in gaussian_model_latent_strands.py), run at the top of every strand-stage iteration
 * (src/train_strands.py:98-104).  origins [S,3] strand roots, dirs [S,n_seg,3] segment vectors (n_seg <= GHR_STRAND_MAX_SEG);
 * outputs, row s n_seg + k = segment k of strand s:
 *   xyz [S n_seg,3]      mid-points of pts = origins + cat(0, cumsum(dirs)) -- the sum taken in list order: bit-identical to torch
 *   rotation [S n_seg,4] parallel_transport((1,0,0), dir) = (1 + b.x, 0, -b.z, b.y), b = dir / max(|dir|, 1e-12)
 *                        (src/utils/general_utils.py:150-160; not normalised)
 *   scaling [S n_seg,3]  (|dir| / 2, scale, scale)
 * The direction rows themselves (self._dir) are a view of dirs and need no kernel. */
int ghr_strand_build(void* stream, int32_t S, int32_t n_seg, const float* origins, const float* dirs, float scale, float* xyz,
                     float* rotation, float* scaling);
/* Its backward: cotangents of the three outputs (each may be NULL = zero; of d_scaling only column 0 is read) -> d_dirs
 * [S,n_seg,3], ASSIGNED.  d xyz_k / d dirs_j = 1 for j < k and 1/2 for j == k (a suffix sum along the strand). */
int ghr_strand_build_backward(void* stream, int32_t S, int32_t n_seg, const float* dirs, const float* d_xyz,
                              const float* d_rotation, const float* d_scaling, float* d_dirs);
/* ABI 20: the same with the cotangent of the direction ROWS (self._dir = dirs.reshape(-1, 3): what render_hair() hands the
 * rasterizer as dir3d, and what ghr_model_backward_segment returns as d_dir3d) as a fourth input, [S n_seg,3] or NULL, added to
 * the result last -- one kernel where autograd otherwise sums the two paths into `_dirs` with a pass of its own. */
int ghr_strand_build_backward_ex(void* stream, int32_t S, int32_t n_seg, const float* dirs, const float* d_xyz,
                                 const float* d_rotation, const float* d_scaling, const float* d_dir_rows, float* d_dirs);

int ghr_model_backward(void* stream, const ghr_model_args* m, uint32_t R, const int32_t* radii, const void* geom_ws,
                       const void* img_ws, const void* bin_ws, const float* dL_dpix, float* grad_scratch,
                       float* d_means2D, float* d_xyz, float* d_log_scales, float* d_rotations,
                       float* d_opacity_logit, float* d_label_logit, float* d_orient_conf_log, float* d_features_dc,
                       float* d_features_rest, int32_t accumulate, int32_t* nan_flag,
                       int32_t prezeroed /* see ghr_backward */);

/* ---- fused stage-1 loss (src/train_gaussians.py:126-140; src/utils/loss_utils.py:19-47,91-121) ----------------
 * loss = w_l1 * mean(|image-gt| * m) + w_ssim * (1 - mean(ssim(image*m, gt*m))) + w_mask * mean(|mask-gt_mask|)
 *        + w_orient * or_loss(orient_angle(dir2d), gt_orient_angle, orient_conf, weight = gt_orient_conf, mask = gt_mask[0]),
 * m = gt_mask[1]; orient_angle as in src/gaussian_renderer/__init__.py:100-105; a NaN orientation term is dropped
 * (train_gaussians.py:134).  All planes are H*W floats: image/gt_image 3, mask/gt_mask 2, dir2d 2 (x, y), the rest 1.
 * The rendered planes may point into one packed [10,H,W] rasterizer output (channels 0-2, 3-4, 5-6, 8).
 * w_orient == 0 disables the orientation term (its pointers may then be NULL). */
typedef struct ghr_loss_args {
    int32_t W, H;
    const float* image;
    const float* mask;
    const float* dir2d;
    const float* orient_conf;
    const float* gt_image;
    const float* gt_mask;
    const float* gt_orient_angle;
    const float* gt_orient_conf;
    float w_l1, w_ssim, w_mask, w_orient;
    int32_t unmasked_colours;     /* != 0: L1 / SSIM on the whole image (train_strands.py:128-129) instead of * gt_mask[1] */
    const float* gt_stats;        /* optional [2,3,H,W]: the SSIM window moments of the (masked) ground truth from
                                     ghr_loss_gt_stats -- constants of a training view; with them ghr_loss_forward
                                     convolves three moments instead of five, with bit-identical results */
} ghr_loss_args;
/* maps: 9*H*W floats of scratch kept for backward.  sums: ghr_loss_sums_floats(W, H) device floats of scratch kept for
 * backward -- {sum of orientation weights, NaN flag, pad} + one slot of five partial sums per workgroup of the forward kernel;
 * needs no initialisation (ABI 16: the 256 shared slots of earlier versions were zero-filled by a launch of their own in front
 * of every forward pass; with a slot per workgroup the fold also has a fixed order).  loss_out: device scalar. */
size_t ghr_loss_sums_floats(int32_t W, int32_t H);
int ghr_loss_forward(void* stream, const ghr_loss_args* a, float* maps, float* sums, float* loss_out);
/* stats_out [2,3,H,W] = {w * y, w * y^2} per colour channel, y = gt_image (* gt_mask[1] unless unmasked_colours), w the
 * 11x11 window of loss_utils.py:91-121.  Reads only W, H, gt_image, gt_mask, unmasked_colours of `a`. */
int ghr_loss_gt_stats(void* stream, const ghr_loss_args* a, float* stats_out);
/* grad_loss: device scalar dL/dloss (NULL = 1).  d_image [3,H,W], d_mask [2,H,W] fully written; d_dir2d [2,H,W] and
 * d_orient_conf [1,H,W] (both or neither; zeros when the orientation term is off or was NaN); zero_plane_a/b: optional
 * H*W planes to zero-fill (the channels of a packed [10,H,W] gradient that no loss term touches). */
int ghr_loss_backward(void* stream, const ghr_loss_args* a, const float* maps, const float* sums,
                      const float* grad_loss, float* d_image, float* d_mask, float* d_dir2d, float* d_orient_conf,
                      float* zero_plane_a, float* zero_plane_b);

/* ---- fused Adam (src/scene/gaussian_model.py:431-444; src/train_gaussians.py:174-181) ---------------------------
 * One pass over flat buffers p/g/m/v of n floats split into n_groups contiguous groups (group_end[i] = exclusive end
 * offset, lr[i] = its learning rate; host arrays).  state: GHR_ADAM_STATE (18) device ints {step, nan_flag, steps each
 * group has sat out [16]}, zero-initialised once.  skip_mask bit g: group g takes no update in this step and its own step
 * number stops advancing for it -- what torch.optim.Adam does for a parameter whose grad is None, i.e. for the
 * nn.Parameters the reference's densification / opacity reset has just replaced (gaussian_model.py:581-658; the
 * optimizer step follows at train_gaussians.py:180).
 * nan_guard != 0: skip the whole update (and do not advance step) when any gradient is NaN, on-device;
 * nan_guard == 1 scans the gradients first, nan_guard == 2 trusts state[1] as maintained by the gradient producer
 * (ghr_model_backward's nan_flag).
 * zero_grad != 0: the gradient buffer is zeroed for the next step. */
/* ABI 19.  One re-lay of the flat parameter / moment buffers for a densification event (src/scene/gaussian_model.py:596-741:
 * densify_and_clone, densify_and_split, prune_points with their cat_tensors_to_optimizer / _prune_optimizer re-creations of
 * every parameter and Adam state tensor).  The buffers are group-major: group g holds P rows of width[g] floats.  For every
 * group and every NEW row r (P_new of them):  p_out[g][r] = override[g] != NULL && child[r] >= 0 ? override[g][child[r]] :
 * p_in[g][take[r]];  m_out / v_out[g][r] = fresh[r] ? 0 : m_in / v_in[g][take[r]]  (clones and split children start with zero
 * moments, gaussian_model.py:634-654).  take, child (nullable; int64) and fresh (uint8) are DEVICE arrays of P_new entries;
 * width_host and override_host (nullable; n_groups device pointers, each NULL or [n_children, width[g]]) are HOST arrays. */
int ghr_adam_relay_rows(void* stream, int32_t n_groups, const int32_t* width_host, int64_t P_old, int64_t P_new,
                        const int64_t* take, const uint8_t* fresh, const int64_t* child, const float* const* override_host,
                        const float* p_in, const float* m_in, const float* v_in, float* p_out, float* m_out, float* v_out);
/* ABI 18.  state[1] |= any(isnan(g[0 .. count))): the scan of nan_guard == 1 over a part of the gradients, for a step whose other
 * gradients came from a producer that keeps the flag itself (the strand stage, src/train_strands.py:151-155: the SH features'
 * gradients are assigned by the fused render_hair backward, those of the strand directions arrive through autograd); follow
 * it with ghr_adam_step(nan_guard = 2). */
int ghr_adam_nan_scan(void* stream, const float* g, int64_t count, int32_t* state);
int ghr_adam_step(void* stream, int64_t n, float* p, float* g, float* m, float* v, int32_t* state, int32_t n_groups,
                  const int64_t* group_end_host, const float* lr_host, double beta1, double beta2, float eps,
                  int32_t nan_guard, int32_t zero_grad, uint32_t skip_mask);
/* The same update restricted to the elements [begin, begin + count) of the n-element buffers (pointers, groups and n
 * as for the whole buffer), so a step can be applied chunk by chunk as the chunks of an all-reduce arrive.  Every
 * chunk of a step sees the same step number; `last` != 0 on the final chunk advances the counter and clears the
 * flag.  nan_guard 1 (scan) is only accepted for the whole buffer. */
int ghr_adam_step_range(void* stream, int64_t n, int64_t begin, int64_t count, float* p, float* g, float* m, float* v,
                        int32_t* state, int32_t n_groups, const int64_t* group_end_host, const float* lr_host,
                        double beta1, double beta2, float eps, int32_t nan_guard, int32_t zero_grad, int32_t last,
                        uint32_t skip_mask);

/* ABI 20.  ghr_adam_step_range OUT OF PLACE: p, m, v of [begin, begin + count) are read from the *_in buffers and written to
 * the *_out buffers (same offsets; an element that takes no update -- flag up, or its group in skip_mask -- is copied), the
 * gradients zeroed when zero_grad != 0.  flag != NULL: the skip-the-step word is *flag instead of state[1].  nan_mark != 0 (needs
 * flag): the pass ORs 1 into *flag itself when a gradient of the range is NaN (the check of src/train_strands.py:151-155) -- some
 * of its elements may then have been updated already, which is harmless here and ONLY here: the `in` set is intact and the
 * caller's finish (ghr_adam_fused_finish) copies it over the `out` set when the flag is up.  No separate ghr_adam_nan_scan of the
 * range is needed in front of the call.  The step counter is not advanced.  For the late groups of a fused step (ghr_adam_fuse, strand segment: the strand directions' and the
 * confidence's gradients arrive through autograd after the kernel that carried the SH features' update): their ranges of the
 * `out` set are produced by ONE pass each instead of three copies and an in-place pass (src/train_strands.py:151-160). */
int ghr_adam_step_range_to(void* stream, int64_t n, int64_t begin, int64_t count, const float* p_in, const float* m_in,
                           const float* v_in, float* p_out, float* g, float* m_out, float* v_out, int32_t* state,
                           int32_t* flag, int32_t nan_mark, int32_t n_groups, const int64_t* group_end_host,
                           const float* lr_host, double beta1, double beta2, float eps, int32_t zero_grad, uint32_t skip_mask);

/* present[i] = view-space z > 0.2 (rasterizer_impl.cu:54-66). */
int ghr_mark_visible(void* stream, int32_t P, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present);

/* Profiling hook used by bench.py: raw hipEvent_t handles (as void*) that the NEXT calls (from any thread --
 * autograd runs backward on a worker thread) record on the caller's stream immediately before / after the compositing kernel (ghr_forward_stage2: k_render_fwd) and the
 * gradient-walk kernel (ghr_backward: k_render_bwd).  NULL disables a pair.  Sticky until changed. */
int ghr_set_profile_events(void* fwd_start, void* fwd_stop, void* bwd_start, void* bwd_stop);

/* Debug aid (ABI 12; SURVEY.md 5: the reference has none -- its backward sums with float atomics in scheduling
 * order): on != 0 makes the gradient walk bit-reproducible from run to run.  One wave per tile then works the tile's
 * sixteen cells in index order, so the additions into a gradient line happen in one fixed order (the per-Gaussian sum
 * over lines is ordered anyway); about 3x the kernel time.  Process-wide, sticky; returns the previous setting.  While it
 * is on, a backward call at a size the ordered walk cannot address (>= 2^26 rows or instances) FAILS with GHR_E_INVALID
 * instead of running unordered. */
int ghr_set_deterministic(int32_t on);

/* Self test of the wave-level primitives the scan form of the gradient walk is built from (16-lane DPP row scans,
 * row broadcast, v_mfma_f32_16x16x4_f32 operand / result layout): in[8][64] floats -> out[12][64] floats (device
 * pointers); tests/test_gpu_wave_primitives.py states the expected values. */
int ghr_selftest_wave(void* stream, const float* in, float* out);

/* ABI 15.  Self test of the hardware transcendentals the kernels use where the reference calls libm (exp in the alpha of
 * forward.cu:366 / backward.cu:504, 1 / (1 - alpha) in backward.cu:507, sqrt / log / rcp in the conservative culls): the
 * DEVICE twins of csrc/ghr_device.h -- v_exp_f32(x * log2 e), v_rcp_f32, v_sqrt_f32, v_log_f32 * ln 2 -- which the CPU
 * host-sim tests cannot see (they compile the host twins: expf, 1 / x, sqrtf, logf).  in[n][4] -> out[n][4] (device pointers):
 * out = (fast_exp(in.x), fast_rcp(in.y), fast_sqrt(in.z), fast_log(in.w)); tests/test_gpu_wave_primitives.py bounds their
 * error in ulp -- the margin of 2e-5 the parity tests give a discrete decision (tests/helpers.py) rests on those bounds. */
int ghr_selftest_math(void* stream, int32_t n, const float* in, float* out);

/* Introspection for tests (device pointers into the workspaces; layout is otherwise private). */
typedef struct ghr_ws_view {
    const float* rec;          /* [P][16]: x, y, conic a, b, c, opacity, features[10] */
    const float* depths;       /* [P] */
    const uint32_t* rects;     /* [P][4]: (xmin | xmax<<16), (ymin | ymax<<16), first gradient slot = [2] + [3] */
    const float* cov3D;        /* [P][6] (mode B) or NULL */
    const float* final_T;      /* [H*W] */
    const uint32_t* n_contrib; /* [H*W] */
    const uint32_t* tile_start;/* [T+1] exclusive scan of per-tile instance counts == ranges */
    const uint64_t* keys;      /* [R] per-tile sorted (depth_bits << 32 | gaussian idx) */
    const uint32_t* point_list;/* [R] */
} ghr_ws_view;
int ghr_ws_inspect(int32_t P, int32_t W, int32_t H, int32_t mode_b, uint32_t R, const void* geom_ws,
                   const void* img_ws, const void* bin_ws, ghr_ws_view* out);

#ifdef __cplusplus
}
#endif
#endif /* GHR_H */
