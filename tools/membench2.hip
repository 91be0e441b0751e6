// tools/membench2.hip — what does a plain streaming copy reach on this box, and with which shape?
// (the guide records 6.29 TB/s for a float4 copy, tools/membench.hip saw 5.4-5.5.)
// Variants: buffer size (Infinity-Cache resident or not), grid, block, unroll (loads in flight per
// thread), cache policy (default / nt), read-only and write-only streams, hipMemcpyDtoD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template<int U, bool NT>
__global__ void copyU(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n) {
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// each block owns a contiguous chunk (block-contiguous instead of grid-stride)
template<int U, bool NT>
__global__ void copy_chunk(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n) {
    const size_t per = n / gridDim.x;
    const f4 *src = in + per * blockIdx.x;
    f4 *dst = out + per * blockIdx.x;
    for (size_t i = threadIdx.x; i + (U - 1) * blockDim.x < per; i += (size_t) U * blockDim.x) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * blockDim.x) : src[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * blockDim.x); else dst[i + u * blockDim.x] = v[u]; }
    }
}

template<int U>
__global__ void read_only(const f4 *__restrict__ in, float *sink, size_t n) {
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = 1.f;
}

__global__ void write_only(f4 *__restrict__ out, size_t n) {
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

template<typename F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms / reps < best) best = ms / reps;
    }
    CK(hipGetLastError());
    return best;
}

int main() {
    const size_t max_bytes = (size_t) 2 << 30;
    f4 *x, *y; float *sink;
    CK(hipMalloc(&x, max_bytes)); CK(hipMalloc(&y, max_bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0x3c, max_bytes)); CK(hipMemset(y, 0, max_bytes));
    auto rep = [&](const char *name, float ms, double bytes) { printf("%-64s %8.3f ms  %6.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    // warm the clocks
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((copyU<1, false>), dim3(2048), dim3(256), 0, 0, x, y, max_bytes / 16);
    CK(hipDeviceSynchronize());
    char name[160];
    for (size_t mb : {64, 256, 1024, 2048}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        snprintf(name, sizeof name, "copy  %4zu MiB grid-stride 2048x256 U1", mb);
        rep(name, timeit([&] { hipLaunchKernelGGL((copyU<1, false>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    }
    const size_t bytes = max_bytes, n = bytes / 16;
    for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
        snprintf(name, sizeof name, "copy 2048 MiB grid-stride %dx256 U1", g);
        rep(name, timeit([&] { hipLaunchKernelGGL((copyU<1, false>), dim3(g), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    }
    rep("copy 2048 MiB one f4 per thread (grid = n/256)", timeit([&] { hipLaunchKernelGGL((copyU<1, false>), dim3((unsigned) (n / 256)), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 2048x256 U2", timeit([&] { hipLaunchKernelGGL((copyU<2, false>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 2048x256 U4", timeit([&] { hipLaunchKernelGGL((copyU<4, false>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 2048x256 U8", timeit([&] { hipLaunchKernelGGL((copyU<8, false>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 1024x256 U8", timeit([&] { hipLaunchKernelGGL((copyU<8, false>), dim3(1024), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 256x1024 U4", timeit([&] { hipLaunchKernelGGL((copyU<4, false>), dim3(256), dim3(1024), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 512x1024 U4", timeit([&] { hipLaunchKernelGGL((copyU<4, false>), dim3(512), dim3(1024), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 2048x256 U1 nt", timeit([&] { hipLaunchKernelGGL((copyU<1, true>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 2048x256 U4 nt", timeit([&] { hipLaunchKernelGGL((copyU<4, true>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 4096x256 U4 nt", timeit([&] { hipLaunchKernelGGL((copyU<4, true>), dim3(4096), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB 2048x512 U4 nt", timeit([&] { hipLaunchKernelGGL((copyU<4, true>), dim3(2048), dim3(512), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB chunked 2048x256 U4", timeit([&] { hipLaunchKernelGGL((copy_chunk<4, false>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB chunked 2048x256 U4 nt", timeit([&] { hipLaunchKernelGGL((copy_chunk<4, true>), dim3(2048), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB chunked 8192x256 U4 nt", timeit([&] { hipLaunchKernelGGL((copy_chunk<4, true>), dim3(8192), dim3(256), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("copy 2048 MiB chunked 256x1024 U4 nt", timeit([&] { hipLaunchKernelGGL((copy_chunk<4, true>), dim3(256), dim3(1024), 0, 0, x, y, n); }), 2.0 * bytes);
    rep("read-only 2048 MiB 2048x256 U1", timeit([&] { hipLaunchKernelGGL((read_only<1>), dim3(2048), dim3(256), 0, 0, x, sink, n); }), 1.0 * bytes);
    rep("read-only 2048 MiB 2048x256 U4", timeit([&] { hipLaunchKernelGGL((read_only<4>), dim3(2048), dim3(256), 0, 0, x, sink, n); }), 1.0 * bytes);
    rep("read-only 2048 MiB 4096x256 U8", timeit([&] { hipLaunchKernelGGL((read_only<8>), dim3(4096), dim3(256), 0, 0, x, sink, n); }), 1.0 * bytes);
    rep("write-only 2048 MiB 2048x256", timeit([&] { hipLaunchKernelGGL(write_only, dim3(2048), dim3(256), 0, 0, y, n); }), 1.0 * bytes);
    rep("write-only 2048 MiB 8192x256", timeit([&] { hipLaunchKernelGGL(write_only, dim3(8192), dim3(256), 0, 0, y, n); }), 1.0 * bytes);
    rep("hipMemcpyAsync D2D 2048 MiB", timeit([&] { CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)); }), 2.0 * bytes);
    rep("hipMemsetAsync 2048 MiB", timeit([&] { CK(hipMemsetAsync(y, 0, bytes, 0)); }), 1.0 * bytes);
    return 0;
}
