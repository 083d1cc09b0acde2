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
    const int skip = a.state[1];
    if ((int)threadIdx.x < a.n_groups) {
        const int step = a.state[0] + 1 - a.state[2 + threadIdx.x];  // this update's step number for the group
        const double bias1 = 1.0 - pow(a.beta1, (double)step);
        s_ss[threadIdx.x] = (float)((double)a.lr[threadIdx.x] / bias1);
        s_b2[threadIdx.x] = (float)sqrt(1.0 - pow(a.beta2, (double)step));
    }
    __syncthreads();
    const float w1 = (float)(1.0 - a.beta1), w2 = (float)(1.0 - a.beta2), b2 = (float)a.beta2;
    for (long long i = a.begin + (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        if (!skip) {
            int gi = 0;
            while (gi < a.n_groups - 1 && i >= a.end[gi]) gi++;
            if (!((a.skip_mask >> gi) & 1u)) {
                float p = a.p[i], m = a.m[i], v = a.v[i];
                adam_update(p, a.g[i], m, v, s_ss[gi], w1, b2, w2, a.eps, s_b2[gi]);
                a.p[i] = p; a.m[i] = m; a.v[i] = v;
            }
        }
        if (a.zero_grad) a.g[i] = 0.f;
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

}  // namespace ghr
