// VALU issue-rate micro-benchmark (kernel experiments; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Every wave runs N iterations of 16 independent dependency chains of one instruction kind; 8 waves per SIMD, every
// CU busy.  Reports cycles per wave-instruction per SIMD (4.0 = one wave64 instruction per 4 clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CHAINS 16

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
    f2 x[CHAINS];
    for (int i = 0; i < CHAINS; i++) x[i] = f2{a + i + threadIdx.x, b + i};
    const f2 m = {a, a}, c = {b, b};
    unsigned long long msk = 0x5555555555555555ull, cm[2] = {0, 0};
    asm volatile("" : "+s"(msk));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(m.x), "v"(c.x));   // v_fma_f32
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));      // v_pk_fma_f32
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(m));       // v_pk_mul_f32
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));       // v_pk_add_f32
            if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(a));        // v_mul_f32
            if (KIND == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x[i].x));
            if (KIND == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i].x));                     // transcendental
            if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i].x) : "v"(m.x) : );
            if (KIND == 9) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(m.x), "s"(msk));
            if (KIND == 10) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(cm[i & 1]) : "v"(x[i].x), "v"(m.x));
            if (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i].y) : "v"(x[i].x));
            if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i].x) : "v"(m.x), "v"(c.x));
            if (KIND == 13) asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i].x));
            if (KIND == 14) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(m.x));
            if (KIND == 15) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i].x) : "v"(m.x));
            if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i].x));
        }
    }
    float s = (float)(cm[0] + cm[1]);
    for (int i = 0; i < CHAINS; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int blocks = 256 * 8, iters = 4096;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_add_f32_dpp",
                           "v_exp_f32", "v_cndmask_b32 vcc", "v_rcp_f32", "v_cndmask_b32_e64 s", "v_cmp_lt_f32_e64",
                           "v_mov_b32", "v_fmac_f32", "v_mul_f32_dpp shr", "v_min_f32", "v_sub_f32"};
    auto run = [&](int kind) {
        void (*fn)(float*, int, float, float) = nullptr;
        switch (kind) {
            case 0: fn = k<0>; break; case 1: fn = k<1>; break; case 2: fn = k<2>; break; case 3: fn = k<3>; break;
            case 4: fn = k<4>; break; case 5: fn = k<5>; break; case 6: fn = k<6>; break; case 7: fn = k<7>; break;
            case 8: fn = k<8>; break; case 9: fn = k<9>; break; case 10: fn = k<10>; break; case 11: fn = k<11>; break;
            case 12: fn = k<12>; break; case 13: fn = k<13>; break; case 14: fn = k<14>; break;
            default: fn = k<15>; break;
        }
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 64, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr_per_simd = (double)blocks * 4 * iters * CHAINS / 1024.0;
        printf("UBENCH %-14s %8.3f ms  %6.2f ns per wave-instruction per SIMD  (= %.2f clocks at 2.4 GHz, %.2f at 2.1 GHz)\n",
               names[kind], ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4,
               ms * 1e6 / wave_instr_per_simd * 2.1);
    };
    for (int kind = 0; kind < 16; kind++) run(kind);
    return 0;
}
