// What does an f32 MFMA cost a wave that is bound by VALU issue?  (kernel experiments; not part of the product; round 6)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_issue_cost mfma_issue_cost.hip && ./mfma_issue_cost
// Every wave runs iterations of 32 independent v_fma_f32 (16 chains x 2) with M MFMAs of one kind spread between them
// (independent accumulators: no MFMA waits for another).  8 / 5 / 2 waves per SIMD, every CU busy.  The time per iteration
// against the M = 0 loop, in v_fma issue slots, is what one MFMA takes away from the VALU stream.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CHAINS 16

template <int KIND, int M>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
    float x[CHAINS];
    for (int i = 0; i < CHAINS; i++) x[i] = a + i + threadIdx.x;
    f4v acc4[4];
    f16v acc16[4];
    for (int j = 0; j < 4; j++) {
        acc4[j] = f4v{a, b, a, b};
        for (int i = 0; i < 16; i++) acc16[j][i] = a + i;
    }
    float av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                // M MFMAs per iteration, spread: after chains 3, 7, 11, 15 of the first half
                if (half == 0 && (i & 3) == 3 && (i >> 2) < M) {
                    const int j = i >> 2;
                    if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[j]) : "v"(av), "v"(bv));
                    if (KIND == 1) asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0" : "+v"(acc16[j]) : "v"(av), "v"(bv));
                    if (KIND == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4[j]) : "v"(av), "v"(bv));
                    if (KIND == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16[j]) : "v"(av), "v"(bv));
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < CHAINS; i++) s += x[i];
    for (int j = 0; j < 4; j++) {
        s += acc4[j].x + acc4[j].y + acc4[j].z + acc4[j].w;
        for (int i = 0; i < 16; i++) s += acc16[j][i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* out;
static hipEvent_t e0, e1;

template <int KIND, int M>
static double run(int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, iters = 2048;
    // occupancy through a dynamic-LDS pad
    const size_t lds = (160 * 1024) / wg_per_cu - 1024;
    hipFuncSetAttribute((const void*)k<KIND, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<KIND, M>), dim3(blocks), dim3(256), lds, 0, out, 32, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, M>), dim3(blocks), dim3(256), lds, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-iterations per SIMD
    return ms * 1e6 / ((double)blocks * 4 * iters / 1024.0);
}

template <int KIND>
static void kind(const char* name)
{
    for (int wg : {8, 5, 2}) {
        const double t0 = run<KIND, 0>(wg), t1 = run<KIND, 1>(wg), t2 = run<KIND, 2>(wg), t4 = run<KIND, 4>(wg);
        const double slot = t0 / 32.0;
        printf("UBENCH mfma %-28s %d waves/SIMD: iteration of 32 v_fma %7.2f ns (slot %.3f ns); +1 MFMA %+6.2f ns = %5.2f slots; "
               "+2: %5.2f slots each; +4: %5.2f slots each\n",
               name, wg, t0, slot, t1 - t0, (t1 - t0) / slot, (t2 - t0) / 2 / slot, (t4 - t0) / 4 / slot);
    }
}

int main()
{
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEventCreate(&e0); hipEventCreate(&e1);
    kind<0>("v_mfma_f32_16x16x4_f32");
    kind<1>("v_mfma_f32_16x16x1_4b_f32");
    kind<2>("v_mfma_f32_4x4x1_16b_f32");
    kind<3>("v_mfma_f32_32x32x2_f32");
    return 0;
}
