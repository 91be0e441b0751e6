// tools/probe64k.hip — diagnostic build of the register-resident rfft kernel with per-phase
// realtime stamps (wave 0 of every workgroup).  Not part of the product or of the tests.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -Iinclude tools/probe64k.hip -o tools/probe64k
//   tools/probe64k [batch] [stagger_ticks]
#define DSC_R2C64K_PROBE 1
// -DDSC_R2C64K_STAMPS adds per-phase stamps; -DDSC_R2C64K_SKIP=<bits> removes parts (ablation)
#include "../dsc_amd/csrc/fft_r2c_64k.hip"

#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 8192;
    const int stagger = argc > 2 ? atoi(argv[2]) : 0;
    const int io_on = argc > 3 ? atoi(argv[3]) : 1;      // 0: zero-record descriptors, memory ops dropped (compute-only timing)
    const int use_stamps = argc > 4 ? atoi(argv[4]) : 1;
    float *x; f2 *X; f2 *aux; unsigned long long *stamps;
    CK(hipMalloc(&x, (size_t) batch * 65536 * 4));
    CK(hipMalloc(&X, (size_t) batch * 32769 * 8));
    CK(hipMalloc(&aux, dsc_r2c64k_table_bytes()));
    const int grid = 256;
    CK(hipMalloc(&stamps, (size_t) grid * 64 * 16 * 8));
    std::vector<char> tab(dsc_r2c64k_table_bytes());
    dsc_r2c64k_build_tables(tab.data());
    CK(hipMemcpy(aux, tab.data(), tab.size(), hipMemcpyHostToDevice));
    std::vector<float> hx((size_t) 1 << 24);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float) ((i * 2654435761u >> 8) & 0xffff) / 32768.f - 1.f;
    for (size_t off = 0; off < (size_t) batch * 65536; off += hx.size())
        CK(hipMemcpy(x + off, hx.data(), std::min(hx.size(), (size_t) batch * 65536 - off) * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *) rfft64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(stamps, 0, (size_t) grid * 64 * 16 * 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rfft64k_kernel, dim3(grid), dim3(1024), kLdsBytes, 0, x, X, batch, aux, use_stamps ? stamps : nullptr, io_on);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %.3f ms\n", rep, ms);
    }
    std::vector<unsigned long long> h((size_t) grid * 64 * 16);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    const int iters = batch / grid;
    const char *names[8] = {"issue loads", "pass1 (+load wait)", "xchg1", "pass2", "xchg2", "pass3", "post+issue stores", "next-iter gap"};
    double sum[8] = {0}; long cnt = 0;
    for (int b = 0; b < grid; ++b)
        for (int it = 2; it < iters - 1 && it < 63; ++it) {
            const unsigned long long *s = &h[((size_t) b * 64 + it) * 16];
            const unsigned long long *n = &h[((size_t) b * 64 + it + 1) * 16];
            for (int i = 0; i < 7; ++i) sum[i] += (double) (s[i + 1] - s[i]);
            sum[7] += (double) (n[0] - s[7]);
            ++cnt;
        }
    double tot = 0;
    for (int i = 0; i < 8; ++i) tot += sum[i] / cnt;
    printf("per-row phases, mean over %ld (wg,row) samples, microseconds (100 MHz ticks / 100):\n", cnt);
    for (int i = 0; i < 8; ++i) printf("  %-22s %7.2f us  %5.1f%%\n", names[i], sum[i] / cnt / 100.0, 100.0 * sum[i] / cnt / tot);
    printf("  %-22s %7.2f us\n", "row total", tot / 100.0);
    // spread of row start times across workgroups at iteration 10 (are they in lockstep?)
    std::vector<double> st;
    for (int b = 0; b < grid; ++b) st.push_back((double) h[((size_t) b * 64 + 10) * 16]);
    std::sort(st.begin(), st.end());
    printf("row-10 start spread across WGs: p10-p90 = %.2f us, min-max = %.2f us\n", (st[230] - st[25]) / 100.0, (st.back() - st[0]) / 100.0);
    return 0;
}
