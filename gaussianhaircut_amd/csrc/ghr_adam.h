// ghr_adam.h -- fused multi-group Adam over ONE flat parameter / gradient / moment buffer.
//
// Reference: torch.optim.Adam(l, lr=0.0, eps=1e-15) over 8 parameter groups (src/scene/gaussian_model.py:431-444),
// stepped at src/train_gaussians.py:174-181 behind a NaN guard that skips the whole update.  PyTorch's foreach Adam
// runs ~10 multi-tensor kernels over 61 floats/Gaussian (1.19 ms at 500k on MI355X); here one pass reads p, g, m, v
// and writes p, m, v (28 B/param, the HBM floor), zeroes the gradient for the next step (folds zero_grad), and applies
// the NaN guard on-device (no host sync): if the flag is set nothing is updated and the step counter does not advance,
// exactly like optimizer.zero_grad(set_to_none=True); optimizer.step() in the reference.
#pragma once
#include "ghr_device.h"

namespace ghr {

#define GHR_ADAM_MAX_GROUPS 16
#ifndef GHR_ADAM_STATE  // include/ghr.h states the same number for the callers
#define GHR_ADAM_STATE (2 + GHR_ADAM_MAX_GROUPS)
#endif
static_assert(GHR_ADAM_STATE == 2 + GHR_ADAM_MAX_GROUPS, "GHR_ADAM_STATE of include/ghr.h");

struct AdamArgs {
    long long begin;      // first element of the range this launch updates
    long long n;          // one past its last element (group lookup is by absolute index)
    float* p;             // flat parameters
    float* g;             // flat gradients (zeroed on exit when zero_grad != 0)
    float* m;             // exp_avg
    float* v;             // exp_avg_sq
    int* state;           // [0] = step count (advanced by k_adam_finish when not skipped), [1] = NaN flag,
                          // [2 + g] = steps group g has sat out (GHR_ADAM_STATE ints in all)
    unsigned skip_mask;   // bit g: group g takes no update in this step (its parameters were just replaced: the
                          // reference's new nn.Parameters have grad None and optimizer.step() passes them by)
    int n_groups;
    long long end[GHR_ADAM_MAX_GROUPS];  // exclusive end offset of each group in the flat buffer
    float lr[GHR_ADAM_MAX_GROUPS];
    double beta1, beta2;  // double like torch's Python-side scalars: 1 - beta^step cancels badly in fp32
    float eps;
    int zero_grad;
    // ABI 20 (ghr_adam_step_range_to): out-of-place form.  p_in != NULL: p, m, v are READ from p_in / m_in / v_in and written to
    // p / m / v (same offsets; an element that takes no update is copied).  flag != NULL: the skip-the-step word is *flag
    // instead of state[1] (a fused step's own flag word, ghr_adam_fuse).
    const float* p_in;
    const float* m_in;
    const float* v_in;
    const int* flag;
    int* nan_mark;   // != NULL (out-of-place form only): the pass itself ORs 1 into *nan_mark when a gradient it reads is NaN -- the
                     // `in` set stays intact, so whoever finishes the step can still undo everything (k_adam_fused_finish): no
                     // separate scan of the range in front of the update
};

// state[1] |= any(isnan(g)).  grid-stride.
__global__ void __launch_bounds__(256) k_adam_nan_flag(const float* __restrict__ g, long long n, int* state)
{
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = g[i];
        bad |= (x != x);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(&state[1], 1);
}

GHR_HD void adam_update(float& p, float g, float& m, float& v, float step_size, float w1, float beta2, float w2,
                        float eps, float bias2_sqrt)
{
    // torch/optim/adam.py (_single_tensor_adam): exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(g,g,1-beta2)
    m = m + (g - m) * w1;
    v = v * beta2 + w2 * g * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256) k_adam(AdamArgs a)
{
    // per group: step size lr / (1 - beta1^t) and sqrt(1 - beta2^t) with t = the group's own step number (torch keeps a
    // step counter per parameter: a group that sat out a step -- skip_mask -- lags the others from then on)
    __shared__ float s_ss[GHR_ADAM_MAX_GROUPS], s_b2[GHR_ADAM_MAX_GROUPS];
    const int skip = a.flag ? *a.flag : a.state[1];
    const bool oop = a.p_in != nullptr;
    const float* pi = oop ? a.p_in : a.p;
    const float* mi = oop ? a.m_in : a.m;
    const float* vi = oop ? a.v_in : a.v;
    if ((int)threadIdx.x < a.n_groups) {
        const int step = a.state[0] + 1 - a.state[2 + threadIdx.x];  // this update's step number for the group
        const double bias1 = 1.0 - pow(a.beta1, (double)step);
        s_ss[threadIdx.x] = (float)((double)a.lr[threadIdx.x] / bias1);
        s_b2[threadIdx.x] = (float)sqrt(1.0 - pow(a.beta2, (double)step));
    }
    __syncthreads();
    const float w1 = (float)(1.0 - a.beta1), w2 = (float)(1.0 - a.beta2), b2 = (float)a.beta2;
    bool bad = false;
    for (long long i = a.begin + (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        bool upd = false;
        int gi = 0;
        if (!skip) {
            while (gi < a.n_groups - 1 && i >= a.end[gi]) gi++;
            upd = !((a.skip_mask >> gi) & 1u);
        }
        if (upd || oop) {
            float p = pi[i], m = mi[i], v = vi[i];
            if (upd) {
                const float gv = a.g[i];
                bad |= gv != gv;
                adam_update(p, gv, m, v, s_ss[gi], w1, b2, w2, a.eps, s_b2[gi]);
            }
            a.p[i] = p; a.m[i] = m; a.v[i] = v;
        }
        if (a.zero_grad) a.g[i] = 0.f;
    }
    if (a.nan_mark != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(a.nan_mark, 1);
}

// k_adam, four consecutive elements per thread as 16-B NON-TEMPORAL accesses (range start a multiple of 4, 16-B aligned
// buffers: the host checks), one float4 per thread for buffers up to 16 M float4.  Same arithmetic per element -- bit-
// identical.  Measured alone on 29.2 M parameters (tools/adam_bench.py, profiles/r03k): scalar kernel 0.205 ms (4.55 TB/s
// of its 32 B per element), float4 0.204, float4 + 16 k / 64 k workgroups instead of 4 k 0.192, non-temporal loads and stores
// 0.184, all three 0.173 (5.4 TB/s): every byte is touched once per step, nothing of it belongs in a cache.
__global__ void __launch_bounds__(256) k_adam_v4(AdamArgs a)
{
    __shared__ float s_ss[GHR_ADAM_MAX_GROUPS], s_b2[GHR_ADAM_MAX_GROUPS];
    const int skip = a.flag ? *a.flag : a.state[1];
    const bool oop = a.p_in != nullptr;
    const float* pi = oop ? a.p_in : a.p;
    const float* mi = oop ? a.m_in : a.m;
    const float* vi = oop ? a.v_in : a.v;
    if ((int)threadIdx.x < a.n_groups) {
        const int step = a.state[0] + 1 - a.state[2 + threadIdx.x];
        const double bias1 = 1.0 - pow(a.beta1, (double)step);
        s_ss[threadIdx.x] = (float)((double)a.lr[threadIdx.x] / bias1);
        s_b2[threadIdx.x] = (float)sqrt(1.0 - pow(a.beta2, (double)step));
    }
    __syncthreads();
    const float w1 = (float)(1.0 - a.beta1), w2 = (float)(1.0 - a.beta2), b2 = (float)a.beta2;
    const long long n4 = (a.n - a.begin) >> 2;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    bool bad = false;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        const long long i = a.begin + 4 * q;
        int g0 = 0, g3 = 0;
        bool any = false;
        if (!skip) {
            while (g0 < a.n_groups - 1 && i >= a.end[g0]) g0++;
            g3 = g0;
            while (g3 < a.n_groups - 1 && i + 3 >= a.end[g3]) g3++;
            any = g0 != g3 || !((a.skip_mask >> g0) & 1u);
        }
        if (any || oop) {
            const f4 P = __builtin_nontemporal_load(reinterpret_cast<const f4*>(pi + i)), M = __builtin_nontemporal_load(reinterpret_cast<const f4*>(mi + i)),
                     V = __builtin_nontemporal_load(reinterpret_cast<const f4*>(vi + i));
            float pp[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
            if (any) {
                const f4 G = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.g + i));
                const float gg[4] = {G.x, G.y, G.z, G.w};
                bad |= (G.x != G.x) | (G.y != G.y) | (G.z != G.z) | (G.w != G.w);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int gi = g0;
                    if (g0 != g3) while (gi < a.n_groups - 1 && i + e >= a.end[gi]) gi++;
                    if (!((a.skip_mask >> gi) & 1u)) adam_update(pp[e], gg[e], mm[e], vv[e], s_ss[gi], w1, b2, w2, a.eps, s_b2[gi]);
                }
            }
            __builtin_nontemporal_store(f4{pp[0], pp[1], pp[2], pp[3]}, reinterpret_cast<f4*>(a.p + i));
            __builtin_nontemporal_store(f4{mm[0], mm[1], mm[2], mm[3]}, reinterpret_cast<f4*>(a.m + i));
            __builtin_nontemporal_store(f4{vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<f4*>(a.v + i));
        }
        if (a.zero_grad) __builtin_nontemporal_store(zero, reinterpret_cast<f4*>(a.g + i));
    }
    // the last (n - begin) % 4 elements
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const long long i = a.begin + 4 * n4 + threadIdx.x;
        if (i < a.n) {
            bool upd = false;
            int gi = 0;
            if (!skip) {
                while (gi < a.n_groups - 1 && i >= a.end[gi]) gi++;
                upd = !((a.skip_mask >> gi) & 1u);
            }
            if (upd || oop) {
                float p = pi[i], m = mi[i], v = vi[i];
                if (upd) {
                    const float gv = a.g[i];
                    bad |= gv != gv;
                    adam_update(p, gv, m, v, s_ss[gi], w1, b2, w2, a.eps, s_b2[gi]);
                }
                a.p[i] = p; a.m[i] = m; a.v[i] = v;
            }
            if (a.zero_grad) a.g[i] = 0.f;
        }
    }
    if (a.nan_mark != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(a.nan_mark, 1);
}


// ---- Adam fused into the projection backward (round 6; ghr_project.h k_project_bwd<.., ADAM>) ---------------------------------
// The LAST view's projection backward of a single-rank step holds every parameter gradient of the step in registers (its own
// terms plus what the earlier views left in the flat gradient buffer) and the raw parameters as well: it applies the update
// there instead of writing 244 B of gradient per Gaussian for k_adam_v4 to read back.  The skip-on-non-finite decision of
// src/train_gaussians.py:174-181 is global, so the update goes to a SECOND set of buffers: p, m, v are read from `in`, the
// updated values written to `out`; the host swaps the roles of the two sets after every fused step, and k_adam_fused_finish
// copies in -> out when the step's flag says that the update must not have happened (a rare event: 732 B per Gaussian then).
#define GHR_ADAM_FUSE_ARRAYS 8  // xyz, log_scales, rotations, opacity, label, orient_conf, features_dc, features_rest
struct AdamFuse {
    const float* p_base;     // the flat parameter buffer ModelArgs' raw-parameter pointers point into (the `in` set)
    const float* m_in;       // exp_avg, exp_avg_sq of the `in` set (same offsets as the parameters)
    const float* v_in;
    float* p_out;            // the `out` set
    float* m_out;
    float* v_out;
    const int* state;        // GHR_ADAM_STATE ints: [0] step count, [2 + g] steps group g has sat out
    float lr[GHR_ADAM_FUSE_ARRAYS];   // learning rate of each array's group, in the order above
    int group[GHR_ADAM_FUSE_ARRAYS];  // ... and its group index
    double beta1, beta2;
    float eps;
    int on;                  // 0: no fusion (everything above ignored)
};

// One thread per element of [0, n): out := in for p, m, v when *flag != 0 (the fused update of this step must be undone);
// workgroup 0 also keeps the books of k_adam_finish: the step counter advances unless the step was skipped.  The flag word is
// not cleared here (other workgroups may still have to read it): fused steps alternate between two flag words and clear the
// OTHER one (`flag_next`), which nobody reads before the next step's producers run.
__global__ void __launch_bounds__(256) k_adam_fused_finish(long long n, const float* __restrict__ p_in,
                                                           const float* __restrict__ m_in, const float* __restrict__ v_in,
                                                           float* __restrict__ p_out, float* __restrict__ m_out,
                                                           float* __restrict__ v_out, int* state, const int* flag, int* flag_next)
{
    const int bad = *flag;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!bad) state[0] += 1;
        *flag_next = 0;
    }
    if (!bad) return;
    const long long n4 = n >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        reinterpret_cast<f4*>(p_out)[q] = reinterpret_cast<const f4*>(p_in)[q];
        reinterpret_cast<f4*>(m_out)[q] = reinterpret_cast<const f4*>(m_in)[q];
        reinterpret_cast<f4*>(v_out)[q] = reinterpret_cast<const f4*>(v_in)[q];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = 4 * n4 + threadIdx.x;
        p_out[i] = p_in[i]; m_out[i] = m_in[i]; v_out[i] = v_in[i];
    }
}

// Runs after k_adam on the same stream: advance the step counter unless skipped, note which groups sat the step out,
// clear the flag.
__global__ void k_adam_finish(int* state, unsigned skip_mask, int n_groups)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (!state[1]) {
            state[0] += 1;
            for (int g = 0; g < n_groups; g++)
                if ((skip_mask >> g) & 1u) state[2 + g] += 1;
        }
        state[1] = 0;
    }
}

// ---- one re-lay of the flat parameter / moment buffers for a densification event (round 6) ---------------------------------
// The reference's densify_and_prune (src/scene/gaussian_model.py:596-741) clones, splits and prunes by re-creating every
// nn.Parameter and every Adam state tensor several times (cat_tensors_to_optimizer / _prune_optimizer: boolean-mask gathers).
// scene/densification.py decides the event on per-row scalars and describes the surviving rows as (source row, fresh?, child?);
// this kernel then writes the NEW flat buffers in one pass: for every group g (rows of w_g floats, group-major in both
// layouts) and every new row r
//   p_out[g][r] = override_g ? override_g[child[r]] (child[r] >= 0) : p_in[g][take[r]]
//   m_out[g][r] = fresh[r] ? 0 : m_in[g][take[r]]            (clones and split children start with zero moments)
//   v_out[g][r] likewise
// ~80 PyTorch launches (index_select / where per group and buffer, then the copies of FusedAdam._rebuild) become one.
struct RelayArgs {
    long long P_old, P_new;
    int n_groups;
    int width[GHR_ADAM_MAX_GROUPS];           // floats per row of each group
    long long end_new[GHR_ADAM_MAX_GROUPS];   // exclusive end of each group in the NEW flat layout
    long long off_old[GHR_ADAM_MAX_GROUPS];   // start of each group in the OLD flat layout
    const float* override_[GHR_ADAM_MAX_GROUPS];  // optional per group: [n_children, width] rows for child[r] >= 0
    const long long* take;    // [P_new] source row
    const unsigned char* fresh;  // [P_new]
    const long long* child;   // [P_new] or NULL
    const float* p_in; const float* m_in; const float* v_in;
    float* p_out; float* m_out; float* v_out;
};

__global__ void __launch_bounds__(256) k_relay_rows(RelayArgs a)
{
    const long long n = a.end_new[a.n_groups - 1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int g = 0;
        while (i >= a.end_new[g]) g++;
        const long long beg = g == 0 ? 0 : a.end_new[g - 1];
        const int w = a.width[g];
        const long long r = (i - beg) / w;
        const int c = (int)((i - beg) - r * w);
        const long long src = a.off_old[g] + a.take[r] * w + c;
        float pv = a.p_in[src];
        if (a.override_[g] != nullptr && a.child != nullptr) {
            const long long ch = a.child[r];
            if (ch >= 0) pv = a.override_[g][ch * w + c];
        }
        const bool fr = a.fresh[r] != 0;
        a.p_out[i] = pv;
        a.m_out[i] = fr ? 0.f : a.m_in[src];
        a.v_out[i] = fr ? 0.f : a.v_in[src];
    }
}

}  // namespace ghr
