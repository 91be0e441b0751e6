// tools/calib_copy.hip — known-byte-count kernel in the rfft kernel's own access pattern
// (8 B per lane loads of 256 KiB rows, 16 B per lane line-aligned stores), used to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE under-counts
// wide reads; other widths must be calibrated in your own pattern).
//   reads exactly rows * 262144 B, writes exactly rows * 262144 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void calib_rowcopy_l8_s16(const float *x, float *y, int rows) {
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) (x + (size_t) row * 65536), 0, 262144, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (y + (size_t) row * 65536), 0, 262144, 0x00020000);
        float v[64];
        const int t = threadIdx.x;
#pragma unroll
        for (int j = 0; j < 32; ++j) { f2 a = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rin, t * 8, j * 8192, 0)); v[2 * j] = a.x; v[2 * j + 1] = a.y; }
#pragma unroll
        for (int j = 0; j < 16; ++j) { f4 a = {v[4 * j] + 1.f, v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]}; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, a), rout, t * 16, j * 16384, 0); }
    }
}
int main() {
    const int rows = 8192; const size_t bytes = (size_t) rows * 262144;
    float *x, *y;
    if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) return 1;
    hipMemset(x, 0x3c, bytes);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(calib_rowcopy_l8_s16, dim3(256), dim3(1024), 0, 0, x, y, rows);
    hipDeviceSynchronize();
    printf("calib: each launch reads %zu B and writes %zu B\n", bytes, bytes);
    return 0;
}
