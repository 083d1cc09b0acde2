/*
 * ghr_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, IEEE fp32, no FMA contraction)
 * of the reference's CUDA rasterizer `ext/diff_gaussian_rasterization_hair/cuda_rasterizer/`.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library.
 * The product path (gaussianhaircut_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: PINNED to outputs of the reference's own rasterizer.  The reference ships no tests or golden
 * vectors and no CPU path (SURVEY.md F1,F2), so the pin was made here: oracle/Makefile.ref hipifies the nine files of
 * /root/reference/ext/diff_gaussian_rasterization_hair/cuda_rasterizer (text substitution, algorithm untouched) and
 * compiles them for gfx950 against a minimal glm stand-in (oracle/ref_shim/) into oracle/_ref/libghr_ref.so;
 * tests/golden/make_reference_cuda_golden.py ran it on an MI355X and stored inputs, outputs, internal state and
 * gradients of six cases (modes A, A_sr, B_sr, B_cov) in tests/golden/reference_cuda_golden.npz.
 * tests/test_reference_cuda_golden.py compares this file with them on the CPU: radii, tile counts, offsets, the
 * 64-bit sort keys, sorted point lists, tile ranges, depth bits, pixel means and n_contrib are BIT-IDENTICAL;
 * images agree to 1e-5, gradients to 1e-5 (the reference accumulates with fp32 atomics in arbitrary order); the
 * in-kernel conic of mode B to 4e-7 (the reference binary contracts a*b+c, this file does not).
 * Further pins: (i) analytic closed-form micro-cases, (ii) an independent fp64 PyTorch autograd renderer
 * (tests/test_oracle_autograd.py), (iii) the reference's own Python `filter_points` (tests/test_reference_golden.py).
 *
 * Every function cites the reference lines it follows.  R: = ext/diff_gaussian_rasterization_hair/
 *
 * Arithmetic discipline: compiled with -ffp-contract=off so every `a*b+c` is two rounded fp32 operations in
 * source order, exactly as the C expressions of the reference read (nvcc's own FMA contraction choices are
 * not reproducible here).  3x3 products follow glm's column-major operator* term order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define MAXC 16

/* ---------------------------------------------------------------- helpers (R:cuda_rasterizer/auxiliary.h) */

/* auxiliary.h:41-44 -- evaluated in double because the literals are double, then narrowed. */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:58-66 */
static void transformPoint4x3(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
/* auxiliary.h:68-77 */
static void transformPoint4x4(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
/* auxiliary.h:89-97 */
static void transformVec4x3Transpose(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
    o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
    o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:46-56.  max_radius is an int; (p - r)/16 is a float division truncated toward zero. */
static void getRect(float px, float py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
    rmin[0] = imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* glm column-major 3x3: m[c][r].  operator* term order: a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2]. */
typedef struct { float m[3][3]; } mat3;
static mat3 m3mul(const mat3* a, const mat3* b)
{
    mat3 o;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a->m[0][r] * b->m[c][0] + a->m[1][r] * b->m[c][1] + a->m[2][r] * b->m[c][2];
    return o;
}
static mat3 m3T(const mat3* a)
{
    mat3 o;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) o.m[c][r] = a->m[r][c];
    return o;
}

/* forward.cu:118-152 (computeCov3D).  Quaternion is NOT normalised (forward.cu:127). */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    mat3 S;
    memset(&S, 0, sizeof(S));
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    mat3 M = m3mul(&S, &R);
    mat3 Mt = m3T(&M);
    mat3 Sigma = m3mul(&Mt, &M);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

/* shared by forward.cu:74-113 (computeCov2D) and backward.cu:166-194: builds t (clamped), J, W, T, Vrk. */
typedef struct { float t[3]; float txtz, tytz, limx, limy; mat3 J, W, T, Vrk; } cov2d_ctx;
static void cov2d_setup(const float* mean, float fx, float fy, float tan_fovx, float tan_fovy,
                        const float* cov3D, const float* view, cov2d_ctx* c)
{
    transformPoint4x3(mean, view, c->t);
    c->limx = 1.3f * tan_fovx;
    c->limy = 1.3f * tan_fovy;
    c->txtz = c->t[0] / c->t[2];
    c->tytz = c->t[1] / c->t[2];
    c->t[0] = fminf(c->limx, fmaxf(-c->limx, c->txtz)) * c->t[2];
    c->t[1] = fminf(c->limy, fmaxf(-c->limy, c->tytz)) * c->t[2];
    float tz = c->t[2];
    memset(&c->J, 0, sizeof(mat3));
    c->J.m[0][0] = fx / tz; c->J.m[0][1] = 0.f; c->J.m[0][2] = -(fx * c->t[0]) / (tz * tz);
    c->J.m[1][0] = 0.f; c->J.m[1][1] = fy / tz; c->J.m[1][2] = -(fy * c->t[1]) / (tz * tz);
    c->W.m[0][0] = view[0]; c->W.m[0][1] = view[4]; c->W.m[0][2] = view[8];
    c->W.m[1][0] = view[1]; c->W.m[1][1] = view[5]; c->W.m[1][2] = view[9];
    c->W.m[2][0] = view[2]; c->W.m[2][1] = view[6]; c->W.m[2][2] = view[10];
    c->T = m3mul(&c->W, &c->J);
    c->Vrk.m[0][0] = cov3D[0]; c->Vrk.m[0][1] = cov3D[1]; c->Vrk.m[0][2] = cov3D[2];
    c->Vrk.m[1][0] = cov3D[1]; c->Vrk.m[1][1] = cov3D[3]; c->Vrk.m[1][2] = cov3D[4];
    c->Vrk.m[2][0] = cov3D[2]; c->Vrk.m[2][1] = cov3D[4]; c->Vrk.m[2][2] = cov3D[5];
}
static void cov2d_eval(const cov2d_ctx* c, float* cov)
{
    mat3 Tt = m3T(&c->T), Vt = m3T(&c->Vrk);
    mat3 A = m3mul(&Tt, &Vt);
    mat3 C = m3mul(&A, &c->T);
    C.m[0][0] += 0.3f;
    C.m[1][1] += 0.3f;
    cov[0] = C.m[0][0]; cov[1] = C.m[0][1]; cov[2] = C.m[1][1];
}

/* ---------------------------------------------------------------- K1  forward.cu:155-282 (preprocessCUDA) */
/*
 * Mode A: conic_precomp != NULL (pipeline mode).  Mode B: conic_precomp == NULL, cov3D from cov3D_precomp or
 * scales/rotations.  Outputs follow forward.cu:275-281.  The reference __trap()s on a prefilter violation
 * (auxiliary.h:154-162); here such Gaussians are silently culled.
 * cov3D_out (6P, may be NULL): the cov3D actually used (geomState.cov3D / cov3D_precomp) for the backward.
 */
void ghro_preprocess(int P, int W, int H,
                     const float* means3D, const float* opacities,
                     const float* scales, const float* rotations, float scale_modifier,
                     const float* cov3D_precomp, const float* conic_precomp,
                     const float* viewmatrix, const float* projmatrix,
                     float tan_fovx, float tan_fovy,
                     float* depths, int* radii, float* xy, float* conic_opacity,
                     float* cov3D_out, uint32_t* tiles_touched)
{
    const float focal_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:224-225 */
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        depths[idx] = 0.f;
        xy[2 * idx] = xy[2 * idx + 1] = 0.f;
        conic_opacity[4 * idx] = conic_opacity[4 * idx + 1] = conic_opacity[4 * idx + 2] = conic_opacity[4 * idx + 3] = 0.f;
        const float* p_orig = means3D + 3 * idx;
        /* in_frustum, auxiliary.h:139-164 */
        float p_view[3];
        transformPoint4x3(p_orig, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue;
        /* forward.cu:203-205 */
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = { p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w };

        /* forward.cu:214-223 */
        float cov3D_local[6];
        const float* cov3D = NULL;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * idx;
        else if (scales && rotations) {
            computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3D_local);
            cov3D = cov3D_local;
        }
        if (cov3D_out && cov3D) memcpy(cov3D_out + 6 * idx, cov3D, 6 * sizeof(float));

        float cov[3], conic[3], det;
        if (!conic_precomp) { /* forward.cu:228-239 */
            cov2d_ctx c;
            cov2d_setup(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &c);
            cov2d_eval(&c, cov);
            det = (cov[0] * cov[2] - cov[1] * cov[1]);
            if (det == 0.0f) continue;
            float det_inv = 1.f / det;
            conic[0] = cov[2] * det_inv; conic[1] = -cov[1] * det_inv; conic[2] = cov[0] * det_inv;
        } else { /* forward.cu:240-248 */
            conic[0] = conic_precomp[3 * idx]; conic[1] = conic_precomp[3 * idx + 1]; conic[2] = conic_precomp[3 * idx + 2];
            float det_inv = (conic[0] * conic[2] - conic[1] * conic[1]);
            if (det_inv == 0.0f) continue;
            det = 1.f / det_inv;
            cov[0] = conic[2] * det; cov[1] = -conic[1] * det; cov[2] = conic[0] * det;
        }
        /* forward.cu:254-262 */
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix[2] = { ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H) };
        int rmin[2], rmax[2];
        getRect(pix[0], pix[1], (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        /* forward.cu:275-281 */
        depths[idx] = p_view[2];
        radii[idx] = (int)my_radius;
        xy[2 * idx] = pix[0]; xy[2 * idx + 1] = pix[1];
        conic_opacity[4 * idx] = conic[0]; conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2]; conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
}

/* rasterizer_impl.cu:54-66,141-153 (checkFrustum / markVisible) */
void ghro_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        float p_view[3];
        transformPoint4x3(means3D + 3 * idx, viewmatrix, p_view);
        present[idx] = !(p_view[2] <= 0.2f);
    }
}

/* ---------------------------------------------------------------- binning  rasterizer_impl.cu:35-138,281-321 */

/* rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* rasterizer_impl.cu:281-285: inclusive scan of tiles_touched; returns num_rendered. */
int64_t ghro_scan(int P, const uint32_t* tiles_touched, uint32_t* point_offsets)
{
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
    return P > 0 ? (int64_t)acc : 0;
}

/*
 * duplicateWithKeys (rasterizer_impl.cu:70-111) + stable LSD radix sort of the low 32+bit key bits
 * (cub::DeviceRadixSort::SortPairs, rasterizer_impl.cu:304-312) + identifyTileRanges (:116-138, memset :314).
 * keys_sorted (R u64), point_list (R u32), ranges (2*T u32, zero-initialised here).
 */
void ghro_binning(int P, int W, int H, const float* xy, const float* depths, const int* radii,
                  const uint32_t* point_offsets, int64_t R,
                  uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    if (R <= 0) return;
    uint64_t* k0 = (uint64_t*)malloc(sizeof(uint64_t) * R);
    uint32_t* v0 = (uint32_t*)malloc(sizeof(uint32_t) * R);
    uint64_t* k1 = (uint64_t*)malloc(sizeof(uint64_t) * R);
    uint32_t* v1 = (uint32_t*)malloc(sizeof(uint32_t) * R);
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            int rmin[2], rmax[2];
            getRect(xy[2 * idx], xy[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    k0[off] = key;
                    v0[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    const int end_bit = 32 + (int)getHigherMsb((uint32_t)(gx * gy));
    /* stable LSD radix sort, 16-bit digits, only bits [0,end_bit) participate */
    size_t* hist = (size_t*)malloc(sizeof(size_t) * 65537);
    for (int shift = 0; shift < end_bit; shift += 16) {
        int nb = end_bit - shift < 16 ? end_bit - shift : 16;
        uint64_t mask = ((uint64_t)1 << nb) - 1;
        memset(hist, 0, sizeof(size_t) * 65537);
        for (int64_t i = 0; i < R; i++) hist[((k0[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 65536; d++) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < R; i++) {
            size_t d = (size_t)((k0[i] >> shift) & mask);
            size_t pos = hist[d]++;
            k1[pos] = k0[i];
            v1[pos] = v0[i];
        }
        uint64_t* tk = k0; k0 = k1; k1 = tk;
        uint32_t* tv = v0; v0 = v1; v1 = tv;
    }
    memcpy(keys_sorted, k0, sizeof(uint64_t) * R);
    memcpy(point_list, v0, sizeof(uint32_t) * R);
    /* identifyTileRanges */
    for (int64_t idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == R - 1) ranges[2 * currtile + 1] = (uint32_t)R;
    }
    free(hist); free(k0); free(v0); free(k1); free(v1);
}

/* ---------------------------------------------------------------- K7  forward.cu:287-400 (renderCUDA fwd) */
/*
 * fragile (N bytes, may be NULL): set to 1 for a pixel when any discrete decision on its walk
 * (power>0, alpha<1/255, T(1-alpha)<1e-4) was taken within a relative margin `frag_eps` of its threshold,
 * i.e. where an implementation with a different exp() rounding may legitimately decide differently.
 * The thread-block structure of the reference (batches of 256, __syncthreads_count early-out) does not change
 * per-pixel results: a pixel that is `done` never resumes, so the walk below is per pixel over the tile's list.
 */
void ghro_render_forward(int W, int H, int C,
                         const uint32_t* ranges, const uint32_t* point_list,
                         const float* xy, const float* features, const float* conic_opacity,
                         const float* bg_color,
                         float* out_color, float* final_T, uint32_t* n_contrib,
                         uint8_t* fragile, float frag_eps)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const uint32_t pix_id = (uint32_t)W * py + px;
                const float pixf[2] = { (float)px, (float)py };
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float Cc[MAXC];
                for (int ch = 0; ch < C; ch++) Cc[ch] = 0.f;
                uint8_t frag = 0;
                for (uint32_t e = r0; e < r1; e++) {
                    contributor++;
                    const uint32_t id = point_list[e];
                    const float d[2] = { xy[2 * id] - pixf[0], xy[2 * id + 1] - pixf[1] };
                    const float* con_o = conic_opacity + 4 * id;
                    const float power = -0.5f * (con_o[0] * d[0] * d[0] + con_o[2] * d[1] * d[1]) - con_o[1] * d[0] * d[1];
                    if (power > 0.0f) continue;
                    const float ex = expf(power);
                    const float alpha = fminf(0.99f, con_o[3] * ex);
                    if (fabsf(alpha - 1.0f / 255.0f) <= frag_eps * (1.0f / 255.0f)) frag = 1;
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    if (fabsf(test_T - 0.0001f) <= 16.f * frag_eps * 0.0001f) frag = 1;
                    if (test_T < 0.0001f) break; /* done = true; nothing after this is blended */
                    for (int ch = 0; ch < C; ch++) Cc[ch] += features[id * C + ch] * alpha * T;
                    T = test_T;
                    last_contributor = contributor;
                }
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < C; ch++) out_color[(size_t)ch * H * W + pix_id] = Cc[ch] + T * bg_color[ch];
                if (fragile) fragile[pix_id] = frag;
            }
    }
}

/* ---------------------------------------------------------------- K8  backward.cu:403-561 (renderCUDA bwd) */
/*
 * The reference accumulates with fp32 atomicAdd in a non-deterministic order; the oracle accumulates each
 * Gaussian's sums in double and rounds once (the value every fp32 summation order approximates).
 * dL_dmean2D: P x 3 (z unused, stays 0).  dL_dconic2D: P x 4 viewed as (P,2,2): .x -> [0], .y -> [1], .w -> [3].
 */
void ghro_render_backward(int P, int W, int H, int C,
                          const uint32_t* ranges, const uint32_t* point_list,
                          const float* bg_color, const float* xy, const float* conic_opacity, const float* colors,
                          const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels,
                          float* dL_dmean2D, float* dL_dconic2D, float* dL_dopacity, float* dL_dcolors)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int NG = 6 + C; /* mean2D.x,.y, conic .x,.y,.w, opacity, colors[C] */
    double* acc = (double*)calloc((size_t)P * NG, sizeof(double));
    const float ddelx_dx = (float)(0.5 * W); /* backward.cu:464-465 */
    const float ddely_dy = (float)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int toDo = (int)(r1 - r0);
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H)) continue;
                const uint32_t pix_id = (uint32_t)W * py + px;
                const float pixf[2] = { (float)px, (float)py };
                const float T_final = final_Ts[pix_id];
                float T = T_final;
                uint32_t contributor = (uint32_t)toDo;
                const int last_contributor = (int)n_contrib[pix_id];
                float accum_rec[MAXC], dL_dpixel[MAXC], last_color[MAXC];
                for (int ch = 0; ch < C; ch++) {
                    accum_rec[ch] = 0.f; last_color[ch] = 0.f;
                    dL_dpixel[ch] = dL_dpixels[(size_t)ch * H * W + pix_id];
                }
                float last_alpha = 0.f;
                for (int j = 0; j < toDo; j++) {
                    contributor--;
                    if ((int)contributor >= last_contributor) continue;
                    const uint32_t id = point_list[r1 - 1 - j];
                    const float d[2] = { xy[2 * id] - pixf[0], xy[2 * id + 1] - pixf[1] };
                    const float* con_o = conic_opacity + 4 * id;
                    const float power = -0.5f * (con_o[0] * d[0] * d[0] + con_o[2] * d[1] * d[1]) - con_o[1] * d[0] * d[1];
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = fminf(0.99f, con_o[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    double* a = acc + (size_t)id * NG;
                    for (int ch = 0; ch < C; ch++) {
                        const float c = colors[id * C + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        const float g = dchannel_dcolor * dL_dchannel;
#pragma omp atomic
                        a[6 + ch] += (double)g;
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < C; i++) bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = con_o[3] * dL_dalpha;
                    const float gdx = G * d[0];
                    const float gdy = G * d[1];
                    const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                    const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
                    const float g0 = dL_dG * dG_ddelx * ddelx_dx;
                    const float g1 = dL_dG * dG_ddely * ddely_dy;
                    const float g2 = -0.5f * gdx * d[0] * dL_dG;
                    const float g3 = -0.5f * gdx * d[1] * dL_dG;
                    const float g4 = -0.5f * gdy * d[1] * dL_dG;
                    const float g5 = G * dL_dalpha;
#pragma omp atomic
                    a[0] += (double)g0;
#pragma omp atomic
                    a[1] += (double)g1;
#pragma omp atomic
                    a[2] += (double)g2;
#pragma omp atomic
                    a[3] += (double)g3;
#pragma omp atomic
                    a[4] += (double)g4;
#pragma omp atomic
                    a[5] += (double)g5;
                }
            }
    }
    for (int i = 0; i < P; i++) {
        const double* a = acc + (size_t)i * NG;
        dL_dmean2D[3 * i] = (float)a[0]; dL_dmean2D[3 * i + 1] = (float)a[1]; dL_dmean2D[3 * i + 2] = 0.f;
        dL_dconic2D[4 * i] = (float)a[2]; dL_dconic2D[4 * i + 1] = (float)a[3];
        dL_dconic2D[4 * i + 2] = 0.f; dL_dconic2D[4 * i + 3] = (float)a[4];
        dL_dopacity[i] = (float)a[5];
        for (int ch = 0; ch < C; ch++) dL_dcolors[(size_t)i * C + ch] = (float)a[6 + ch];
    }
    free(acc);
}

/* ---------------------------------------------------------------- K9  backward.cu:144-274 (computeCov2DCUDA) */
/* Mode B only (launched when conics == nullptr, backward.cu:588).  ASSIGNS dL_dmeans (backward.cu:273). */
void ghro_cov2d_backward(int P, const float* means, const int* radii, const float* cov3Ds,
                         float h_x, float h_y, float tan_fovx, float tan_fovy, const float* view,
                         const float* dL_dconics /* P x 4 */, float* dL_dmeans /* P x 3 */, float* dL_dcov /* P x 6 */)
{
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        cov2d_ctx c;
        cov2d_setup(means + 3 * idx, h_x, h_y, tan_fovx, tan_fovy, cov3Ds + 6 * idx, view, &c);
        const float dLc[3] = { dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3] };
        const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0.f : 1.f;
        const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0.f : 1.f;
        float cov[3];
        cov2d_eval(&c, cov);
        const float a = cov[0], b = cov[1], cc = cov[2];
        const mat3* T = &c.T; const mat3* V = &c.Vrk; const mat3* Wm = &c.W;
        float denom = a * cc - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dLc[0] + 2 * b * cc * dLc[1] + (denom - a * cc) * dLc[2]);
            dL_dc = denom2inv * (-a * a * dLc[2] + 2 * a * b * dLc[1] + (denom - a * cc) * dLc[0]);
            dL_db = denom2inv * 2 * (b * cc * dLc[0] - (denom + 2 * b * b) * dLc[1] + a * b * dLc[2]);
            dcov[0] = (T->m[0][0] * T->m[0][0] * dL_da + T->m[0][0] * T->m[1][0] * dL_db + T->m[1][0] * T->m[1][0] * dL_dc);
            dcov[3] = (T->m[0][1] * T->m[0][1] * dL_da + T->m[0][1] * T->m[1][1] * dL_db + T->m[1][1] * T->m[1][1] * dL_dc);
            dcov[5] = (T->m[0][2] * T->m[0][2] * dL_da + T->m[0][2] * T->m[1][2] * dL_db + T->m[1][2] * T->m[1][2] * dL_dc);
            dcov[1] = 2 * T->m[0][0] * T->m[0][1] * dL_da + (T->m[0][0] * T->m[1][1] + T->m[0][1] * T->m[1][0]) * dL_db + 2 * T->m[1][0] * T->m[1][1] * dL_dc;
            dcov[2] = 2 * T->m[0][0] * T->m[0][2] * dL_da + (T->m[0][0] * T->m[1][2] + T->m[0][2] * T->m[1][0]) * dL_db + 2 * T->m[1][0] * T->m[1][2] * dL_dc;
            dcov[4] = 2 * T->m[0][2] * T->m[0][1] * dL_da + (T->m[0][1] * T->m[1][2] + T->m[0][2] * T->m[1][1]) * dL_db + 2 * T->m[1][1] * T->m[1][2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        float dL_dT00 = 2 * (T->m[0][0] * V->m[0][0] + T->m[0][1] * V->m[0][1] + T->m[0][2] * V->m[0][2]) * dL_da +
                        (T->m[1][0] * V->m[0][0] + T->m[1][1] * V->m[0][1] + T->m[1][2] * V->m[0][2]) * dL_db;
        float dL_dT01 = 2 * (T->m[0][0] * V->m[1][0] + T->m[0][1] * V->m[1][1] + T->m[0][2] * V->m[1][2]) * dL_da +
                        (T->m[1][0] * V->m[1][0] + T->m[1][1] * V->m[1][1] + T->m[1][2] * V->m[1][2]) * dL_db;
        float dL_dT02 = 2 * (T->m[0][0] * V->m[2][0] + T->m[0][1] * V->m[2][1] + T->m[0][2] * V->m[2][2]) * dL_da +
                        (T->m[1][0] * V->m[2][0] + T->m[1][1] * V->m[2][1] + T->m[1][2] * V->m[2][2]) * dL_db;
        float dL_dT10 = 2 * (T->m[1][0] * V->m[0][0] + T->m[1][1] * V->m[0][1] + T->m[1][2] * V->m[0][2]) * dL_dc +
                        (T->m[0][0] * V->m[0][0] + T->m[0][1] * V->m[0][1] + T->m[0][2] * V->m[0][2]) * dL_db;
        float dL_dT11 = 2 * (T->m[1][0] * V->m[1][0] + T->m[1][1] * V->m[1][1] + T->m[1][2] * V->m[1][2]) * dL_dc +
                        (T->m[0][0] * V->m[1][0] + T->m[0][1] * V->m[1][1] + T->m[0][2] * V->m[1][2]) * dL_db;
        float dL_dT12 = 2 * (T->m[1][0] * V->m[2][0] + T->m[1][1] * V->m[2][1] + T->m[1][2] * V->m[2][2]) * dL_dc +
                        (T->m[0][0] * V->m[2][0] + T->m[0][1] * V->m[2][1] + T->m[0][2] * V->m[2][2]) * dL_db;
        float dL_dJ00 = Wm->m[0][0] * dL_dT00 + Wm->m[0][1] * dL_dT01 + Wm->m[0][2] * dL_dT02;
        float dL_dJ02 = Wm->m[2][0] * dL_dT00 + Wm->m[2][1] * dL_dT01 + Wm->m[2][2] * dL_dT02;
        float dL_dJ11 = Wm->m[1][0] * dL_dT10 + Wm->m[1][1] * dL_dT11 + Wm->m[1][2] * dL_dT12;
        float dL_dJ12 = Wm->m[2][0] * dL_dT10 + Wm->m[2][1] * dL_dT11 + Wm->m[2][2] * dL_dT12;
        float tz = 1.f / c.t[2];
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dt[3];
        dL_dt[0] = x_grad_mul * -h_x * tz2 * dL_dJ02;
        dL_dt[1] = y_grad_mul * -h_y * tz2 * dL_dJ12;
        dL_dt[2] = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * c.t[0]) * tz3 * dL_dJ02 + (2 * h_y * c.t[1]) * tz3 * dL_dJ12;
        transformVec4x3Transpose(dL_dt, view, dL_dmeans + 3 * idx);
    }
}

/* backward.cu:278-341 (computeCov3D bwd) */
static void computeCov3D_bwd(const float* scale, float mod, const float* rot, const float* dL_dcov3D,
                             float* dL_dscale, float* dL_drot)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    mat3 S;
    memset(&S, 0, sizeof(S));
    float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    mat3 M = m3mul(&S, &R);
    mat3 dS;
    dS.m[0][0] = dL_dcov3D[0]; dS.m[0][1] = 0.5f * dL_dcov3D[1]; dS.m[0][2] = 0.5f * dL_dcov3D[2];
    dS.m[1][0] = 0.5f * dL_dcov3D[1]; dS.m[1][1] = dL_dcov3D[3]; dS.m[1][2] = 0.5f * dL_dcov3D[4];
    dS.m[2][0] = 0.5f * dL_dcov3D[2]; dS.m[2][1] = 0.5f * dL_dcov3D[4]; dS.m[2][2] = dL_dcov3D[5];
    /* dL_dM = 2.0f * M * dL_dSigma : glm evaluates (2.0f * M) * dL_dSigma */
    mat3 M2;
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = 2.0f * M.m[c][rr];
    mat3 dL_dM = m3mul(&M2, &dS);
    mat3 Rt = m3T(&R);
    mat3 dL_dMt = m3T(&dL_dM);
    for (int k = 0; k < 3; k++)
        dL_dscale[k] = Rt.m[k][0] * dL_dMt.m[k][0] + Rt.m[k][1] * dL_dMt.m[k][1] + Rt.m[k][2] * dL_dMt.m[k][2];
    for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) dL_dMt.m[k][rr] *= s[k];
#define D(a, b) dL_dMt.m[a][b]
    dL_drot[0] = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
    dL_drot[1] = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
    dL_drot[2] = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
    dL_drot[3] = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
#undef D
}

/* ---------------------------------------------------------------- K10  backward.cu:346-400 (preprocessCUDA bwd) */
/*
 * conics == NULL (Mode B): mean2D->mean3D term is ADDED to dL_dmeans (:390) and, when scales are given,
 * dL_dcov3D -> dL_dscale / dL_drot (:398-399).  conics != NULL (Mode A): every branch is skipped.
 * The SH branch (:394-395) is dead in this fork (NUM_CHANNELS=10 forces colors_precomp, rasterizer_impl.cu:244-247)
 * and is not restated.
 */
void ghro_preprocess_backward(int P, const float* means, const int* radii,
                              const float* scales, const float* rotations, float scale_modifier,
                              const float* conics, const float* proj,
                              const float* dL_dmean2D /* P x 3 */, float* dL_dmeans /* P x 3 */,
                              const float* dL_dcov3D /* P x 6 */, float* dL_dscale, float* dL_drot)
{
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        if (conics == NULL) {
            const float* m = means + 3 * idx;
            float m_hom[4];
            transformPoint4x4(m, proj, m_hom);
            float m_w = 1.0f / (m_hom[3] + 0.0000001f);
            float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
            float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
            const float* g = dL_dmean2D + 3 * idx;
            float d0 = (proj[0] * m_w - proj[3] * mul1) * g[0] + (proj[1] * m_w - proj[3] * mul2) * g[1];
            float d1 = (proj[4] * m_w - proj[7] * mul1) * g[0] + (proj[5] * m_w - proj[7] * mul2) * g[1];
            float d2 = (proj[8] * m_w - proj[11] * mul1) * g[0] + (proj[9] * m_w - proj[11] * mul2) * g[1];
            dL_dmeans[3 * idx] += d0; dL_dmeans[3 * idx + 1] += d1; dL_dmeans[3 * idx + 2] += d2;
        }
        if (scales && conics == NULL)
            computeCov3D_bwd(scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D + 6 * idx,
                             dL_dscale + 3 * idx, dL_drot + 4 * idx);
    }
}

int ghro_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
