// tools/valubench.hip — issue rate of scalar vs packed f32 VALU on gfx950 (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096, CHAINS = 16;

template<int MODE> __global__ __launch_bounds__(1024) void k(float *out, float a, float b) {
    f2 v[CHAINS];
    for (int i = 0; i < CHAINS; ++i) v[i] = f2{(float) threadIdx.x + i, (float) i};
    const f2 A = {a, a * 1.0001f}, B = {b, b * 0.999f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(a), "v"(b)); }
            if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(A), "v"(B)); }
            if (MODE == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(A)); }
            if (MODE == 3) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(a)); }
            if (MODE == 4) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(A)); }
            if (MODE == 5) { asm volatile("v_mov_b32 %0, %1" : "+v"(v[i].x) : "v"(a)); }
        }
    }
    float s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int MODE> int run(const char *name, int threads) {
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double) ITER * CHAINS * (threads / 64) / 4;       // wave-instructions per SIMD
    printf("%-14s %4d thr/CU: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, threads, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}
int main() {
    for (int threads : {256, 1024}) {
        run<0>("v_fma_f32", threads); run<1>("v_pk_fma_f32", threads); run<3>("v_add_f32", threads);
        run<2>("v_pk_add_f32", threads); run<4>("v_pk_mul_f32", threads); run<5>("v_mov_b32", threads);
    }
    return 0;
}
