// tools/hwid_probe.hip — which bits of HW_REG_HW_ID name the CU a workgroup runs on?  512 workgroups of 256 threads, two per CU (the team
// kernel's launch shape): prints, per XCC, the number of distinct values of bits [15:8] (CU_ID 11:8, SH_ID 12, SE_ID 15:13 on gfx9) and
// how many workgroups share each value (expected: 32 values per XCC, 2 workgroups each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256, 2) void probe(unsigned *out, unsigned *arrived, unsigned total) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // XCC_ID[3:0]
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID[31:0]
        atomicAdd(arrived, 1u);
        unsigned spins = 0;
        while (atomicAdd(arrived, 0u) < total && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(8);     // keep everybody resident at once
    }
    __syncthreads();
}
int main() {
    const unsigned n = 512;
    unsigned *out, *arr;
    CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&arr, 4)); CK(hipMemset(arr, 0, 4));
    CK(hipFuncSetAttribute((const void *) probe, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 48 * 1024, 0, out, arr, n);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * n);
    CK(hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost));
    for (int field = 0; field < 3; ++field) {
        const int lo = field == 0 ? 8 : field == 1 ? 8 : 4, width = field == 0 ? 8 : field == 1 ? 12 : 12;
        std::map<unsigned, std::map<unsigned, int>> per_xcc;
        for (unsigned i = 0; i < n; ++i) per_xcc[h[2 * i] & 15][(h[2 * i + 1] >> lo) & ((1u << width) - 1)]++;
        printf("HW_ID bits [%d:%d]:", lo + width - 1, lo);
        for (auto &x : per_xcc) {
            int mn = 1 << 30, mx = 0;
            for (auto &k : x.second) { mn = k.second < mn ? k.second : mn; mx = k.second > mx ? k.second : mx; }
            printf("  xcc %u: %zu values, %d..%d wg each", x.first, x.second.size(), mn, mx);
        }
        printf("\n");
    }
    printf("first 16 workgroups (xcc, HW_ID):");
    for (int i = 0; i < 16; ++i) printf(" (%u, %08x)", h[2 * i] & 15, h[2 * i + 1]);
    printf("\n");
    return 0;
}
