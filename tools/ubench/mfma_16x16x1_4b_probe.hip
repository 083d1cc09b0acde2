// v_mfma_f32_16x16x1_4b_f32 probe (kernel experiments; not part of the product).  Round 6: could K7's ten colour FMAs per
// hit-step ride the matrix pipe?  Four 16-lane groups of a wave blend four different list entries: block b = group b,
// A = the entry's colour of channel (lane & 15), B = the lane's own weight alpha T.  Questions: where do the 4 x 16 x 16 outputs
// live, is the accumulation a single fused multiply-add per element (bit-equal to fmaf), what does EXEC do to it.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_probe mfma_16x16x1_4b_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const float* a, const float* b, const float* c, float* d, int mode)
{
    const int lane = threadIdx.x;
    f16v acc;
    for (int i = 0; i < 16; i++) acc[i] = c[lane * 16 + i];
    const float av = a[lane], bv = b[lane];
    if (mode == 0) {
        acc = __builtin_amdgcn_mfma_f32_16x16x1f32(av, bv, acc, 0, 0, 0);
    } else if (mode == 1) {  // half of the groups masked off
        if ((lane >> 4) & 1) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(av, bv, acc, 0, 0, 0);
    } else {  // ten dependent accumulations (the step loop's shape)
        for (int i = 0; i < 10; i++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(av + (float)i, bv, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; i++) d[lane * 16 + i] = acc[i];
}

int main()
{
    std::vector<float> a(64), b(64), c(1024), d(1024);
    float *da, *db, *dc, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 4096); hipMalloc(&dd, 4096);
    auto run = [&](int mode) {
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd, mode);
        hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
    };
    // 1. layout: which (block, row) and (block, column) does (lane, register) hold?
    std::vector<int> rowsrc(1024), colsrc(1024);
    for (int l = 0; l < 64; l++) { a[l] = (float)(l + 1); b[l] = 1.f; }
    std::fill(c.begin(), c.end(), 0.f);
    run(0);
    for (int i = 0; i < 1024; i++) rowsrc[i] = (int)d[i] - 1;
    for (int l = 0; l < 64; l++) { a[l] = 1.f; b[l] = (float)(l + 1); }
    run(0);
    for (int i = 0; i < 1024; i++) colsrc[i] = (int)d[i] - 1;
    bool regular = true;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 16; r++) {
            const int blk = r >> 2, row = 4 * (l >> 4) + (r & 3), col = l & 15;
            if (rowsrc[l * 16 + r] != blk * 16 + row || colsrc[l * 16 + r] != blk * 16 + col) regular = false;
        }
    printf("MFMA layout: D[lane][reg] = A[16 (reg >> 2) + 4 (lane >> 4) + (reg & 3)] x B[16 (reg >> 2) + (lane & 15)]: %s\n",
           regular ? "CONFIRMED" : "NO");
    if (!regular)
        for (int l = 0; l < 64; l += 17)
            for (int r = 0; r < 16; r++) printf("  lane %2d reg %2d: A lane %2d, B lane %2d\n", l, r, rowsrc[l * 16 + r], colsrc[l * 16 + r]);
    // 2. rounding: one fused multiply-add per element?
    srand(7);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
    int bad = 0, denorm_bad = 0;
    for (int trial = 0; trial < 50; trial++) {
        for (int l = 0; l < 64; l++) { a[l] = rnd() * (trial % 5 == 4 ? 1e-30f : 1.f); b[l] = rnd() * (trial % 5 == 4 ? 1e-10f : 1.f); }
        for (int i = 0; i < 1024; i++) c[i] = rnd() * (trial % 5 == 4 ? 1e-39f : (trial % 3 == 0 ? 1e-3f : 1.f));
        run(0);
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 16; r++) {
                const float want = fmaf(a[rowsrc[l * 16 + r]], b[colsrc[l * 16 + r]], c[l * 16 + r]);
                if (memcmp(&want, &d[l * 16 + r], 4) != 0) { (trial % 5 == 4 ? denorm_bad : bad)++; }
            }
    }
    printf("MFMA rounding: %d of %d normal-range elements differ from fmaf; %d of %d in the denormal range\n", bad, 40 * 1024,
           denorm_bad, 10 * 1024);
    // 3. EXEC: groups 0 and 2 masked off
    for (int l = 0; l < 64; l++) { a[l] = rnd(); b[l] = rnd(); }
    for (int i = 0; i < 1024; i++) c[i] = rnd();
    run(1);
    int upd_masked = 0, upd_active = 0, n_masked = 0, n_active = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 16; r++) {
            const bool changed = d[l * 16 + r] != c[l * 16 + r];
            if ((l >> 4) & 1) { n_active++; upd_active += changed; } else { n_masked++; upd_masked += changed; }
        }
    printf("MFMA under a branch on (lane >> 4) & 1: %d of %d registers of ACTIVE lanes updated, %d of %d of MASKED lanes\n", upd_active,
           n_active, upd_masked, n_masked);
    // which source lanes fed the active lanes' results: did masked lanes' A / B values take part?
    int full = 0, part = 0;
    for (int l = 0; l < 64; l++)
        if ((l >> 4) & 1)
            for (int r = 0; r < 16; r++) {
                const float want = fmaf(a[rowsrc[l * 16 + r]], b[colsrc[l * 16 + r]], c[l * 16 + r]);
                (memcmp(&want, &d[l * 16 + r], 4) == 0 ? full : part)++;
            }
    printf("MFMA under that branch: %d active-lane results equal the full-EXEC product (A / B of masked lanes read), %d do not\n", full, part);
    return 0;
}
