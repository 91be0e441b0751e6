// tools/membench3.hip — the ROW-PER-WORKGROUP streaming shape of the 65536-point kernels (256 resident workgroups of 1024 threads,
// each walking its own 256 KiB rows), and what changes its throughput: where in its row a workgroup starts (all workgroups at
// offset 0 ask the same HBM channels at the same time when rows are 256 KiB apart), how many rows are open at once, the store
// width and the cache policy.  Reads and writes separately and as a copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: copy, 1: read only, 2: write only.  ROT: the workgroup starts its row at piece (blockIdx * ROT) % 16 (pieces of 16 KiB).
// SHARE: SHARE workgroups walk one row together (each takes every SHARE-th piece) -> 256 / SHARE rows open at once.
template<int MODE, bool NT>
__global__ __launch_bounds__(1024) void row_copy(const f4 *__restrict__ in, f4 *__restrict__ out, int rows, int rot, int share, float *sink) {
    const int t = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    const int team = blockIdx.x / share, member = blockIdx.x % share, teams = gridDim.x / share;
    const int start = (int) (((unsigned) blockIdx.x * (unsigned) rot) & 15);
    for (int row = team; row < rows; row += teams) {
        const f4 *src = in + (size_t) row * 16384;
        f4 *dst = out + (size_t) row * 16384;
        f4 v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int piece = (p + start) & 15;
            if (share > 1 && (piece % share) != member) continue;
            if (MODE != 2) v[p] = NT ? __builtin_nontemporal_load(src + piece * 1024 + t) : src[piece * 1024 + t];
            else v[p] = f4{1.f, 2.f, 3.f, (float) p};
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int piece = (p + start) & 15;
            if (share > 1 && (piece % share) != member) continue;
            if (MODE == 1) acc += v[p];
            else if (NT) __builtin_nontemporal_store(v[p], dst + piece * 1024 + t);
            else dst[piece * 1024 + t] = v[p];
        }
    }
    if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = 1.f;
}

// the kernels' real access widths: 8 B per lane loads (32 per thread, 8 KiB per workgroup instruction), 16 B per lane stores
template<bool NT>
__global__ __launch_bounds__(1024) void row_copy_8B(const f2 *__restrict__ in, f4 *__restrict__ out, int rows, int rot) {
    const int t = threadIdx.x;
    const int start = (int) (((unsigned) blockIdx.x * (unsigned) rot) & 31);
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const f2 *src = in + (size_t) row * 32768;
        f4 *dst = out + (size_t) row * 16384;
        f2 v[32];
#pragma unroll
        for (int p = 0; p < 32; ++p) { const int piece = (p + start) & 31; v[p] = NT ? __builtin_nontemporal_load(src + piece * 1024 + t) : src[piece * 1024 + t]; }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int piece = (p + (start >> 1)) & 15;
            const f4 o = {v[2 * p].x, v[2 * p].y, v[2 * p + 1].x, v[2 * p + 1].y};
            if (NT) __builtin_nontemporal_store(o, dst + piece * 1024 + t); else dst[piece * 1024 + t] = o;
        }
    }
}


// in-flight depth: the row moves in chunks of U pieces per thread (U loads, then U stores); PIPE: chunk c+1's loads are issued
// before chunk c's stores (so waiting for them does not wait for the stores: vmcnt retires in order)
template<int U, bool PIPE>
__global__ __launch_bounds__(1024) void row_copy_depth(const f4 *__restrict__ in, f4 *__restrict__ out, int rows) {
    const int t = threadIdx.x;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const f4 *src = in + (size_t) row * 16384;
        f4 *dst = out + (size_t) row * 16384;
        f4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = src[u * 1024 + t];
#pragma unroll
        for (int c = 0; c < 16 / U; ++c) {
            if (PIPE && c + 1 < 16 / U) {
#pragma unroll
                for (int u = 0; u < U; ++u) b[u] = src[((c + 1) * U + u) * 1024 + t];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) dst[(c * U + u) * 1024 + t] = a[u];
            if (!PIPE && c + 1 < 16 / U) {
#pragma unroll
                for (int u = 0; u < U; ++u) b[u] = src[((c + 1) * U + u) * 1024 + t];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = b[u];
        }
    }
}
// 256-thread workgroups, 4 per CU, each walking its own rows (1024 rows open at once, 4 x fewer lanes per row)
template<int U>
__global__ __launch_bounds__(256) void row_copy_small(const f4 *__restrict__ in, f4 *__restrict__ out, int rows) {
    const int t = threadIdx.x;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const f4 *src = in + (size_t) row * 16384;
        f4 *dst = out + (size_t) row * 16384;
        for (int c = 0; c < 64 / U; ++c) {
            f4 a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = src[(c * U + u) * 256 + t];
#pragma unroll
            for (int u = 0; u < U; ++u) dst[(c * U + u) * 256 + t] = a[u];
        }
    }
}


// one f4 per thread with tiny workgroups, but in the ADDRESS ORDER of the row shape: consecutive workgroups take the same 4 KiB
// piece of W consecutive rows, then the next piece of those rows ... (W = 1: plain address order)
__global__ void copy_one_perm(const f4 *__restrict__ in, f4 *__restrict__ out, int W) {
    const unsigned b = blockIdx.x;                          // 64 pieces of 4 KiB per 256 KiB row
    const unsigned per_window = (unsigned) W * 64;
    const unsigned window = b / per_window, in_w = b % per_window;
    const unsigned row = window * W + in_w % W, piece = in_w / W;
    const size_t i = ((size_t) row * 64 + piece) * 256 + threadIdx.x;
    out[i] = in[i];
}


// persistent row copy with a row pitch (in f4) and two row orders: mode 0 row = block + k * grid (256 consecutive rows open),
// mode 1 each block walks its own contiguous band of rows (open rows are rows / grid apart)
__global__ __launch_bounds__(1024) void row_copy_pitch(const f4 *__restrict__ in, f4 *__restrict__ out, int rows, size_t in_pitch, size_t out_pitch, int mode) {
    const int t = threadIdx.x;
    const int per = rows / gridDim.x;
    for (int k = 0; k < per; ++k) {
        const int row = mode == 0 ? blockIdx.x + k * gridDim.x : blockIdx.x * per + k;
        const f4 *src = in + (size_t) row * in_pitch;
        f4 *dst = out + (size_t) row * out_pitch;
        f4 v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = src[p * 1024 + t];
#pragma unroll
        for (int p = 0; p < 16; ++p) dst[p * 1024 + t] = v[p];
    }
}

// non-persistent: one workgroup per row (dispatch order = address order)
template<int MODE>
__global__ __launch_bounds__(1024) void row_copy_np(const f4 *__restrict__ in, f4 *__restrict__ out, float *sink) {
    const int t = threadIdx.x;
    const f4 *src = in + (size_t) blockIdx.x * 16384;
    f4 *dst = out + (size_t) blockIdx.x * 16384;
    f4 v[16];
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p] = MODE != 2 ? src[p * 1024 + t] : f4{1.f, 2.f, 3.f, (float) p};
#pragma unroll
    for (int p = 0; p < 16; ++p) { if (MODE == 1) acc += v[p]; else dst[p * 1024 + t] = v[p]; }
    if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = 1.f;
}

__global__ void write_one(f4 *__restrict__ out) { out[(size_t) blockIdx.x * blockDim.x + threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f}; }
__global__ void read_one(const f4 *__restrict__ in, float *sink) { const f4 v = in[(size_t) blockIdx.x * blockDim.x + threadIdx.x]; if (v.x + v.y == 123.456f) *sink = 1.f; }

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
    const int rows = 8192;
    const size_t bytes = (size_t) rows * 262144, n = bytes / 16;
    f4 *x, *y; float *sink;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0x3c, bytes)); CK(hipMemset(y, 0, bytes));
    auto rep = [&](const char *name, float ms, double b) { printf("%-72s %8.3f ms  %6.0f GB/s\n", name, ms, b / ms / 1e6); fflush(stdout); };
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((row_copy<0, false>), dim3(256), dim3(1024), 0, 0, x, y, rows, 0, 1, sink);
    CK(hipDeviceSynchronize());
    char name[200];
    for (int rot : {0, 1, 3, 5, 7}) {
        snprintf(name, sizeof name, "row copy 256x1024, start piece = %d * block", rot);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<0, false>), dim3(256), dim3(1024), 0, 0, x, y, rows, rot, 1, sink); }), 2.0 * bytes);
        snprintf(name, sizeof name, "row copy 256x1024 nt, start piece = %d * block", rot);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<0, true>), dim3(256), dim3(1024), 0, 0, x, y, rows, rot, 1, sink); }), 2.0 * bytes);
    }
    for (int share : {2, 4, 8, 16}) {
        snprintf(name, sizeof name, "row copy 256x1024, %d workgroups per row", share);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<0, false>), dim3(256), dim3(1024), 0, 0, x, y, rows, 0, share, sink); }), 2.0 * bytes);
    }
    for (int rot : {0, 1, 5}) {
        snprintf(name, sizeof name, "row read  256x1024, start piece = %d * block", rot);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<1, false>), dim3(256), dim3(1024), 0, 0, x, y, rows, rot, 1, sink); }), 1.0 * bytes);
        snprintf(name, sizeof name, "row write 256x1024, start piece = %d * block", rot);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<2, false>), dim3(256), dim3(1024), 0, 0, x, y, rows, rot, 1, sink); }), 1.0 * bytes);
        snprintf(name, sizeof name, "row write 256x1024 nt, start piece = %d * block", rot);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<2, true>), dim3(256), dim3(1024), 0, 0, x, y, rows, rot, 1, sink); }), 1.0 * bytes);
    }
    for (int share : {4, 16}) {
        snprintf(name, sizeof name, "row write 256x1024, %d workgroups per row", share);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<2, false>), dim3(256), dim3(1024), 0, 0, x, y, rows, 0, share, sink); }), 1.0 * bytes);
    }
    for (int rot : {0, 1, 5}) {
        snprintf(name, sizeof name, "row copy 8B loads / 16B stores, start piece = %d * block", rot);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy_8B<false>), dim3(256), dim3(1024), 0, 0, (const f2 *) x, y, rows, rot); }), 2.0 * bytes);
        snprintf(name, sizeof name, "row copy 8B loads / 16B stores nt, start piece = %d * block", rot);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy_8B<true>), dim3(256), dim3(1024), 0, 0, (const f2 *) x, y, rows, rot); }), 2.0 * bytes);
    }
    for (int g : {128, 512}) {
        snprintf(name, sizeof name, "row copy %dx1024 (launch bounds allow 1 per CU)", g);
        rep(name, timeit([&] { hipLaunchKernelGGL((row_copy<0, false>), dim3(g), dim3(1024), 0, 0, x, y, rows, 0, 1, sink); }), 2.0 * bytes);
    }

    rep("row copy depth 1 per thread", timeit([&] { hipLaunchKernelGGL((row_copy_depth<1, false>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 2 per thread", timeit([&] { hipLaunchKernelGGL((row_copy_depth<2, false>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 4 per thread", timeit([&] { hipLaunchKernelGGL((row_copy_depth<4, false>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 8 per thread", timeit([&] { hipLaunchKernelGGL((row_copy_depth<8, false>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 1 per thread, loads ahead of stores", timeit([&] { hipLaunchKernelGGL((row_copy_depth<1, true>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 2 per thread, loads ahead of stores", timeit([&] { hipLaunchKernelGGL((row_copy_depth<2, true>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 4 per thread, loads ahead of stores", timeit([&] { hipLaunchKernelGGL((row_copy_depth<4, true>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 8 per thread, loads ahead of stores", timeit([&] { hipLaunchKernelGGL((row_copy_depth<8, true>), dim3(256), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 4, 512 workgroups (2 per CU)", timeit([&] { hipLaunchKernelGGL((row_copy_depth<4, true>), dim3(512), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy depth 2, 512 workgroups (2 per CU)", timeit([&] { hipLaunchKernelGGL((row_copy_depth<2, true>), dim3(512), dim3(1024), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy 1024 x 256 threads, depth 4", timeit([&] { hipLaunchKernelGGL((row_copy_small<4>), dim3(1024), dim3(256), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy 2048 x 256 threads, depth 4", timeit([&] { hipLaunchKernelGGL((row_copy_small<4>), dim3(2048), dim3(256), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy 2048 x 256 threads, depth 1", timeit([&] { hipLaunchKernelGGL((row_copy_small<1>), dim3(2048), dim3(256), 0, 0, x, y, rows); }), 2.0 * bytes);
    rep("row copy 8192 x 256 threads (one per row), depth 4", timeit([&] { hipLaunchKernelGGL((row_copy_small<4>), dim3(8192), dim3(256), 0, 0, x, y, rows); }), 2.0 * bytes);

    for (int W : {1, 4, 16, 64, 256, 1024}) {
        snprintf(name, sizeof name, "one f4 per thread, copy, row-shaped address order over %d rows", W);
        rep(name, timeit([&] { hipLaunchKernelGGL(copy_one_perm, dim3((unsigned) (n / 256)), dim3(256), 0, 0, x, y, W); }), 2.0 * bytes);
    }

    {
        const int prow = 6144;                              // fewer rows so that padded pitches fit the 2 GiB buffers
        const double pb = 2.0 * prow * 262144.0;
        for (size_t pad : {(size_t) 0, (size_t) 8, (size_t) 64, (size_t) 256, (size_t) 1024, (size_t) 4096, (size_t) 5120}) {
            snprintf(name, sizeof name, "row copy, in and out pitch 256 KiB + %zu B", pad * 16);
            rep(name, timeit([&] { hipLaunchKernelGGL(row_copy_pitch, dim3(256), dim3(1024), 0, 0, x, y, prow, 16384 + pad, 16384 + pad, 0); }), pb);
        }
        rep("row copy, in pitch 256 KiB, out pitch 256 KiB + 16 KiB", timeit([&] { hipLaunchKernelGGL(row_copy_pitch, dim3(256), dim3(1024), 0, 0, x, y, prow, 16384, 16384 + 1024, 0); }), pb);
        rep("row copy, in pitch 256 KiB + 16 KiB, out pitch 256 KiB", timeit([&] { hipLaunchKernelGGL(row_copy_pitch, dim3(256), dim3(1024), 0, 0, x, y, prow, 16384 + 1024, 16384, 0); }), pb);
        rep("row copy, pitch 256 KiB, each block its own band of rows", timeit([&] { hipLaunchKernelGGL(row_copy_pitch, dim3(256), dim3(1024), 0, 0, x, y, prow, 16384, 16384, 1); }), pb);
        rep("row copy, pitch 256 KiB + 16 KiB, each block its own band of rows", timeit([&] { hipLaunchKernelGGL(row_copy_pitch, dim3(256), dim3(1024), 0, 0, x, y, prow, 16384 + 1024, 16384 + 1024, 1); }), pb);
    }
    rep("one workgroup per row (8192 x 1024), copy", timeit([&] { hipLaunchKernelGGL((row_copy_np<0>), dim3(rows), dim3(1024), 0, 0, x, y, sink); }), 2.0 * bytes);
    rep("one workgroup per row (8192 x 1024), read", timeit([&] { hipLaunchKernelGGL((row_copy_np<1>), dim3(rows), dim3(1024), 0, 0, x, y, sink); }), 1.0 * bytes);
    rep("one workgroup per row (8192 x 1024), write", timeit([&] { hipLaunchKernelGGL((row_copy_np<2>), dim3(rows), dim3(1024), 0, 0, x, y, sink); }), 1.0 * bytes);
    rep("one f4 per thread, write", timeit([&] { hipLaunchKernelGGL(write_one, dim3((unsigned) (n / 256)), dim3(256), 0, 0, y); }), 1.0 * bytes);
    rep("one f4 per thread, read", timeit([&] { hipLaunchKernelGGL(read_one, dim3((unsigned) (n / 256)), dim3(256), 0, 0, x, sink); }), 1.0 * bytes);
    rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(y, 0, bytes, 0)); }), 1.0 * bytes);
    return 0;
}
