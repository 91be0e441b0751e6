// tools/membench.hip — what does the memory system give a persistent 1024-thread workgroup
// per CU that loads a 256 KiB row into registers and stores ~256 KiB back?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// plain streaming copy, 16 B per lane, grid-stride
__global__ void copy16(const f4 *in, f4 *out, size_t n) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) out[i] = in[i];
}

// LW: bytes per lane per load (8/16); SW: per store; row pitch of the output in bytes is a runtime value
template<int LW, int SW, int THREADS, bool PEEL = false>
__global__ __launch_bounds__(THREADS) void rowcopy(const float *x, float *y, int batch, int out_pitch_bytes) {
    constexpr int ROW = 262144;
    constexpr int NL = ROW / (THREADS * LW), NS = ROW / (THREADS * SW);
    for (int row = blockIdx.x; row < batch; row += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) (x + (size_t) row * 65536), 0, ROW, 0x00020000);
        const size_t obase = (size_t) row * out_pitch_bytes;
        // PEEL: rows are only 8-B aligned; start the 16-B stores at the first 16-B boundary (bin 0 or bin L stored alone)
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) ((char *) y + obase + (PEEL ? (obase & 8) : 0)), 0, ROW + 8, 0x00020000);
        float v[ROW / THREADS / 4];
        const int t = threadIdx.x;
        if constexpr (LW == 8) {
#pragma unroll
            for (int j = 0; j < NL; ++j) { f2 a = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rin, t * 8, j * THREADS * 8, 0)); v[2 * j] = a.x; v[2 * j + 1] = a.y; }
        } else {
#pragma unroll
            for (int j = 0; j < NL; ++j) { f4 a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, t * 16, j * THREADS * 16, 0)); v[4 * j] = a.x; v[4 * j + 1] = a.y; v[4 * j + 2] = a.z; v[4 * j + 3] = a.w; }
        }
        if constexpr (SW == 8) {
#pragma unroll
            for (int j = 0; j < NS; ++j) { f2 a = {v[2 * j] + 1.f, v[2 * j + 1]}; __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, a), rout, t * 8, j * THREADS * 8, 0); }
        } else {
#pragma unroll
            for (int j = 0; j < NS; ++j) { f4 a = {v[4 * j] + 1.f, v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]}; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, a), rout, t * 16, j * THREADS * 16, 0); }
        }
    }
}

template<typename F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 10;
}

int main() {
    const int batch = 8192; const size_t in_bytes = (size_t) batch * 262144, out_bytes = (size_t) batch * 262400;
    float *x, *y; CK(hipMalloc(&x, in_bytes)); CK(hipMalloc(&y, out_bytes));
    CK(hipMemset(x, 0x3c, in_bytes));
    auto rep = [&](const char *name, float ms, double bytes) { printf("%-58s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6); };
    const double rb = 2.0 * in_bytes;
    rep("copy16 grid-stride 2048x256", timeit([&] { hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, (const f4 *) x, (f4 *) y, in_bytes / 16); }), rb);
    rep("copy16 grid-stride 256x1024", timeit([&] { hipLaunchKernelGGL(copy16, dim3(256), dim3(1024), 0, 0, (const f4 *) x, (f4 *) y, in_bytes / 16); }), rb);
    rep("rowcopy L8 S8 1024thr pitch 262152 (as the rfft kernel)", timeit([&] { hipLaunchKernelGGL((rowcopy<8, 8, 1024>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262152); }), rb);
    rep("rowcopy L8 S8 1024thr pitch 262144 (aligned)", timeit([&] { hipLaunchKernelGGL((rowcopy<8, 8, 1024>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262144); }), rb);
    rep("rowcopy L16 S8 1024thr pitch 262152", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 8, 1024>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262152); }), rb);
    rep("rowcopy L16 S16 1024thr pitch 262144", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 16, 1024>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262144); }), rb);
    rep("rowcopy L16 S16 1024thr pitch 262400 (+256)", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 16, 1024>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262400); }), rb);
    rep("rowcopy L8 S16 1024thr pitch 262144", timeit([&] { hipLaunchKernelGGL((rowcopy<8, 16, 1024>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262144); }), rb);
    rep("rowcopy L16 S8 1024thr pitch 262144", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 8, 1024>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262144); }), rb);
    rep("rowcopy L16 S16 1024thr pitch 262152 PEEL (16-B aligned interior)", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 16, 1024, true>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262152); }), rb);
    rep("rowcopy L8 S16 1024thr pitch 262152 PEEL", timeit([&] { hipLaunchKernelGGL((rowcopy<8, 16, 1024, true>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262152); }), rb);
    rep("rowcopy L16 S16 1024thr pitch 262152 NO peel (misaligned 16-B)", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 16, 1024, false>), dim3(256), dim3(1024), 0, 0, x, y, batch, 262152); }), rb);
    rep("rowcopy L8 S8 512thr x2/CU pitch 262152", timeit([&] { hipLaunchKernelGGL((rowcopy<8, 8, 512>), dim3(512), dim3(512), 0, 0, x, y, batch, 262152); }), rb);
    rep("rowcopy L16 S16 512thr x2/CU pitch 262144", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 16, 512>), dim3(512), dim3(512), 0, 0, x, y, batch, 262144); }), rb);
    rep("rowcopy L16 S16 256thr x4/CU pitch 262144", timeit([&] { hipLaunchKernelGGL((rowcopy<16, 16, 256>), dim3(1024), dim3(256), 0, 0, x, y, batch, 262144); }), rb);
    rep("rowcopy L8 S8 1024thr grid 512 (2 WG/CU if they fit)", timeit([&] { hipLaunchKernelGGL((rowcopy<8, 8, 1024>), dim3(512), dim3(1024), 0, 0, x, y, batch, 262152); }), rb);
    for (int g : {8, 32, 64, 128, 256}) {
        char name[128]; snprintf(name, sizeof name, "rowcopy L8 S16 1024thr aligned, grid %d (per-CU rate = total/grid)", g);
        float ms = timeit([&] { hipLaunchKernelGGL((rowcopy<8, 16, 1024>), dim3(g), dim3(1024), 0, 0, x, y, batch / 4, 262144); });
        printf("%-58s %.3f ms  %.0f GB/s total, %.1f GB/s per CU, %.2f us per row\n", name, ms, rb / 4 / ms / 1e6, rb / 4 / ms / 1e6 / g, ms * 1e3 / (batch / 4 / g));
    }
    return 0;
}
