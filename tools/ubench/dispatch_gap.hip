// Workgroup hand-over micro-benchmark (kernel experiments; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_gap dispatch_gap.hip && ./dispatch_gap
// Question (round 4, profiles/r04b_k8_timeline.txt): K8's workgroups (256 threads, 29 KB LDS, 86 VGPRs: five per CU) leave
// their CU 3.4 instead of 5 waves per SIMD resident on average; the next workgroup starts a median 11.5 k cycles after one
// ends.  Is that the dispatcher?  Here every workgroup just WAITS a given number of cycles (s_sleep; no memory, no ALU
// contention), with the same resource footprint; 6080 workgroups as in the cfg3 launch.  Ideal launch = rounds x life.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int THREADS, int LDS_BYTES, int WAVES_PER_EU>
__global__ void __launch_bounds__(THREADS, WAVES_PER_EU) k(unsigned long long* out, int life, int jitter)
{
    extern __shared__ char dyn[];
    __shared__ char pad[LDS_BYTES > 0 ? LDS_BYTES : 4];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // workgroup b lives life + (hash(b) % jitter) cycles
    unsigned h = blockIdx.x * 2654435761u;
    const unsigned long long want = (unsigned long long)life + (jitter ? (h >> 8) % (unsigned)jitter : 0u);
    if (threadIdx.x == 0) pad[0] = (char)blockIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < want) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = t0;
        out[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
        out[3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |
                                  ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | ((unsigned long long)pad[0] << 60);
    }
}

template <int THREADS, int LDS_BYTES, int WPE>
void run(const char* name, int blocks, int life, int jitter, int per_cu)
{
    unsigned long long* out;
    hipMalloc(&out, 3 * blocks * sizeof(unsigned long long));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(a);
        k<THREADS, LDS_BYTES, WPE><<<blocks, THREADS>>>(out, life, jitter);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(3 * blocks);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double busy = 0;
    for (int i = 0; i < blocks; i++) busy += (double)(h[3 * i + 1] - h[3 * i]);
    const double mean_life = busy / blocks;
    const double rounds = (double)blocks / (256.0 * per_cu);
    // s_memtime ticks per ms: from one workgroup's life against the kernel time is not possible; report both
    printf("%-34s blocks %5d life %6.0f ticks (asked %d + jitter %d): kernel %.4f ms; ideal = %.2f rounds x life = %.0f ticks\n",
           name, blocks, mean_life, life, jitter, best, rounds, rounds * mean_life);
    hipFree(out);
}

int main(int argc, char** argv)
{
    const int life = argc > 1 ? atoi(argv[1]) : 65000;
    // tick calibration: one round of exactly 1280 workgroups
    run<256, 29 * 1024, 5>("calib 256thr 29KB 5/CU", 1280, life, 0, 5);
    run<256, 29 * 1024, 5>("256thr 29KB 5/CU", 6080, life, 0, 5);
    run<256, 29 * 1024, 5>("256thr 29KB 5/CU jitter", 6080, life / 2, life, 5);
    run<256, 20 * 1024, 5>("256thr 20KB (7 by LDS) jitter", 6080, life / 2, life, 5);
    run<256, 0, 5>("256thr 0KB jitter", 6080, life / 2, life, 5);
    run<64, 7 * 1024, 5>("64thr 7KB jitter (20/CU)", 6080 * 4, life / 2, life, 20);
    run<256, 29 * 1024, 5>("256thr 29KB short life/4 jitter", 6080 * 4, life / 8, life / 4, 5);
    run<1024, 116 * 1024, 5>("1024thr 116KB (1/CU... ) jitter", 1520, life / 2, life, 1);
    return 0;
}
