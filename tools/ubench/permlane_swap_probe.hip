// v_permlane32_swap / v_permlane16_swap (gfx950): what they move and what they cost (kernel experiments; round 6).
//   hipcc --offload-arch=gfx950 -O3 -o permlane_swap_probe permlane_swap_probe.hip && ./permlane_swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_sem(unsigned* o)
{
    const unsigned l = threadIdx.x;
    unsigned x = l, y = 100 + l;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[128 + l] = s[0]; o[192 + l] = s[1];
}

#define CHAINS 16
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, float a)
{
    float x[CHAINS];
    for (int i = 0; i < CHAINS; i++) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i += 2) {
            if (KIND == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 1]));
            if (KIND == 1) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 1]));
            if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %0" : "+v"(x[i]), "+v"(x[i + 1]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < CHAINS; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d);
    std::vector<unsigned> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    const char* nm[4] = {"permlane32_swap vdst", "permlane32_swap src ", "permlane16_swap vdst", "permlane16_swap src "};
    for (int r = 0; r < 4; r++) {
        printf("UBENCH %s (x = lane, y = 100 + lane), first lane of each 16-lane row:", nm[r]);
        for (int row = 0; row < 4; row++) printf(" %3u", h[64 * r + 16 * row]);
        printf("\n");
    }
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 4096;
    auto run = [&](int kind, const char* name, int per_iter) {
        void (*fn)(float*, int, float) = kind == 0 ? k_rate<0> : (kind == 1 ? k_rate<1> : k_rate<2>);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 64, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)blocks * 4 * iters * per_iter / 1024.0;
        printf("UBENCH %-22s %7.3f ms  %5.2f ns per wave-instruction per SIMD (8 waves per SIMD)\n", name, ms, ms * 1e6 / n);
    };
    run(0, "v_permlane32_swap_b32", CHAINS / 2);
    run(1, "v_permlane16_swap_b32", CHAINS / 2);
    run(2, "v_add_f32 (pairs)", CHAINS);
    return 0;
}
