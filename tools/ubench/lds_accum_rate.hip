// LDS accumulation-rate micro-benchmark (kernel experiments; not part of the product).  Round 6, VERDICT r5 next #2:
// the "ds_add_f32 = 1.8 cycles per lane" that ruled out on-chip accumulation of K8's gradient lines twice (profiles/HISTORY.md)
// was an in-kernel figure; here the operation is isolated with K8's own lane -> (entry, component) addressing.
//   hipcc --offload-arch=gfx950 -O3 -o lds_accum_rate lds_accum_rate.hip && ./lds_accum_rate
// A workgroup of four waves owns a table of N lines of STRIDE floats in LDS (N = 384: a tile's list).  Per "chunk" a wave
// adds a 16-entry x 16-component block into the table the way k_render_bwd_cells would: lane (k = lane >> 4, m = lane & 15),
// register r = 0..3 adds component m of entry 4 k + r, i.e. FOUR wave instructions per chunk, each touching four lines.
// The entries of a chunk are pseudo-random list positions (ascending within a chunk, as hit lists are).
//   kind 0  ds_add_f32 (no return)                      table[pos][m]
//   kind 1  ds_add_f32 with a returned value (ds_add_rtn_f32)
//   kind 2  non-atomic read - add - write of the same dword (exclusive owner; what a per-wave private table would do)
//   kind 3  lane (k, c) owns a whole entry quarter: ds_read_b128 + 4 v_add + ds_write_b128 on table[pos][4 c .. 4 c + 3]
//           (16 lanes cover 4 entries per instruction; non-atomic)
// Strides 16 (lines packed), 17 and 20 (padded).  1..5 workgroups per CU (4..20 waves per CU) via a dynamic-LDS pad.
// Reports LDS-pipe time per wave instruction per CU and per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define NLINES 384
#define CHUNKS_PER_LIST 64  // chunk descriptors a wave cycles through

template <int KIND, int STRIDE>
__global__ void __launch_bounds__(256) k(const uint16_t* __restrict__ lists, float* out, int iters)
{
    extern __shared__ float table[];  // NLINES * STRIDE floats (+ the occupancy pad)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, m = lane & 15;
    for (int i = tid; i < NLINES * STRIDE; i += 256) table[i] = 0.f;
    __syncthreads();
    // the wave's chunk descriptors: 16 positions per chunk, in registers of the lanes that use them
    const uint16_t* my = lists + ((size_t)(blockIdx.x * 4 + wave) % 64) * CHUNKS_PER_LIST * 16;
    float v = 1.0f + lane;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        const int c = it & (CHUNKS_PER_LIST - 1);
        uint32_t pos[4];
#pragma unroll
        for (int r = 0; r < 4; r++) pos[r] = my[c * 16 + 4 * kk + r];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (KIND == 0) {
                const uint32_t a = 4u * (pos[r] * STRIDE + m);
                asm volatile("ds_add_f32 %0, %1" : : "v"(a), "v"(v) : "memory");
            } else if (KIND == 1) {
                const uint32_t a = 4u * (pos[r] * STRIDE + m);
                float old;
                asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(a), "v"(v) : "memory");
                acc += old;
            } else if (KIND == 2) {
                float* p = table + pos[r] * STRIDE + m;
                *p = *p + v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            } else {
                // 16 lanes = 4 entries x 4 quarters: lane m handles quarter (m & 3) of entry 4 kk + (m >> 2) in ONE instruction
                // (r loops over ... nothing: one b128 RMW covers the whole chunk; run it once per chunk)
                if (r == 0) {
                    const uint32_t e = my[c * 16 + 4 * kk + (m >> 2)];
                    f4* p = reinterpret_cast<f4*>(table + e * STRIDE + 4 * (m & 3));
                    f4 x = *p;
                    x += f4{v, v, v, v};
                    *p = x;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
            }
        }
    }
    __syncthreads();
    float s = acc;
    for (int i = tid; i < NLINES * STRIDE; i += 256) s += table[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int KIND, int STRIDE>
static void run(const char* name, const uint16_t* d_lists, float* out, int wg_per_cu)
{
    const int iters = 8192;
    // occupancy: dynamic LDS sized so that exactly wg_per_cu workgroups fit into the 160 KB of a CU
    size_t lds = (size_t)NLINES * STRIDE * 4;
    const size_t want = (160 * 1024) / wg_per_cu - 512;
    if (want > lds) lds = want;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k<KIND, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * wg_per_cu * 4;  // four rounds of resident workgroups
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, STRIDE>), dim3(blocks), dim3(256), lds, 0, d_lists, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, STRIDE>), dim3(blocks), dim3(256), lds, 0, d_lists, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("UBENCH %s: launch failed\n", name); return; }
    const double chunks_per_cu = (double)blocks * 4 * iters / 256.0;
    const double ns_chunk = ms * 1e6 / chunks_per_cu;
    const int instr = KIND == 3 ? 1 : 4;
    printf("UBENCH lds %-28s stride %2d  %d wg/CU (%2d waves/CU)  %8.3f ms  %7.2f ns per chunk per CU = %6.1f clocks at 2.4 GHz "
           "(%5.1f per wave instruction, %4.2f per lane)\n",
           name, STRIDE, wg_per_cu, 4 * wg_per_cu, ms, ns_chunk, ns_chunk * 2.4, ns_chunk * 2.4 / instr, ns_chunk * 2.4 / instr / 64);
}

int main()
{
    // 64 lists x 64 chunks x 16 ascending pseudo-random positions in [0, NLINES)
    std::vector<uint16_t> lists(64 * CHUNKS_PER_LIST * 16);
    uint32_t s = 12345u;
    for (size_t c = 0; c < lists.size() / 16; c++) {
        uint32_t base = 0;
        for (int e = 0; e < 16; e++) {
            s = s * 1664525u + 1013904223u;
            base += 1 + (s >> 16) % (2 * NLINES / 16 - 1);  // mean gap NLINES / 16
            lists[16 * c + e] = (uint16_t)(base % NLINES);
        }
    }
    uint16_t* d_lists;
    float* out;
    hipMalloc(&d_lists, lists.size() * 2);
    hipMemcpy(d_lists, lists.data(), lists.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)256 * 5 * 4 * 256 * sizeof(float));
    for (int wg = 1; wg <= 5; wg++) {
        run<0, 16>("ds_add_f32", d_lists, out, wg);
        run<0, 17>("ds_add_f32", d_lists, out, wg);
        run<0, 20>("ds_add_f32", d_lists, out, wg);
        run<1, 16>("ds_add_rtn_f32 + wait", d_lists, out, wg);
        run<2, 16>("read + v_add + write b32", d_lists, out, wg);
        run<2, 17>("read + v_add + write b32", d_lists, out, wg);
        run<3, 16>("read b128 + 4 v_add + write b128", d_lists, out, wg);
        run<3, 20>("read b128 + 4 v_add + write b128", d_lists, out, wg);
    }
    return 0;
}
