// tools/probe64k.hip — diagnostic build of the register-resident rfft kernel: the same instruction stream
// timed with its global accesses live (io_on = 1) and dropped (io_on = 0: zero-record buffer descriptors),
// i.e. whole kernel vs arithmetic + LDS + barriers only.  Not part of the product or of the tests.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -Iinclude -Idsc_amd/csrc tools/probe64k.hip -o tools/bin/probe64k
//   tools/bin/probe64k [batch]
#define DSC_R2C64K_PROBE 1
#include "../dsc_amd/csrc/fft_r2c_64k.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 8192;
    float *x; f2 *X; f2 *aux;
    CK(hipMalloc(&x, (size_t) batch * 65536 * 4));
    CK(hipMalloc(&X, (size_t) batch * 32769 * 8));
    CK(hipMalloc(&aux, dsc_r2c64k_table_bytes()));
    std::vector<char> tab(dsc_r2c64k_table_bytes());
    dsc_r2c64k_build_tables(tab.data());
    CK(hipMemcpy(aux, tab.data(), tab.size(), hipMemcpyHostToDevice));
    std::vector<float> hx((size_t) 1 << 24);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float) ((i * 2654435761u >> 8) & 0xffff) / 32768.f - 1.f;
    for (size_t off = 0; off < (size_t) batch * 65536; off += hx.size())
        CK(hipMemcpy(x + off, hx.data(), std::min(hx.size(), (size_t) batch * 65536 - off) * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *) rfft64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int io_on = 1; io_on >= 0; --io_on) {
        for (int rep = 0; rep < 60; ++rep)                                   // clock ramp
            hipLaunchKernelGGL(rfft64k_kernel, dim3(256), dim3(1024), kLdsBytes, 0, x, X, batch, aux, 65536, 65536, io_on);
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 20; ++rep)
            hipLaunchKernelGGL(rfft64k_kernel, dim3(256), dim3(1024), kLdsBytes, 0, x, X, batch, aux, 65536, 65536, io_on);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f ms per launch of %d rows\n", io_on ? "whole kernel          " : "global accesses dropped", ms / 20, batch);
    }
    CK(hipFuncSetAttribute((const void *) irfft64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    for (int io_on = 1; io_on >= 0; --io_on) {
        for (int rep = 0; rep < 60; ++rep)
            hipLaunchKernelGGL(irfft64k_kernel, dim3(256), dim3(1024), kLdsBytes, 0, X, x, batch, aux, 32769, 32769, io_on);
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 20; ++rep)
            hipLaunchKernelGGL(irfft64k_kernel, dim3(256), dim3(1024), kLdsBytes, 0, X, x, batch, aux, 32769, 32769, io_on);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("irfft %s: %.3f ms per launch of %d rows\n", io_on ? "whole kernel          " : "global accesses dropped", ms / 20, batch);
    }
    return 0;
}
