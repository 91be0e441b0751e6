// tools/icbench.hip — can the 256 MiB Infinity Cache carry the intermediate of a two-pass transform?
// Pair of streaming kernels per chunk: A: X[chunk] -> W (intermediate, S MiB, reused buffer), B: W -> Y[chunk].
// If W stays in the Infinity Cache, HBM sees only X reads and Y writes.  Swept over S; the no-residency reference is
// S = the whole 2 GiB (W as large as X).  Also a fused single launch per chunk pair is not needed: what matters is
// whether W's lines survive between A and B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template<bool NT_IN, bool NT_OUT>
__global__ void copy1(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        f4 v = NT_IN ? __builtin_nontemporal_load(in + i) : in[i];
        v.x += 1.f;
        if (NT_OUT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

int main() {
    const size_t total = (size_t) 2 << 30;
    f4 *X, *Y, *W;
    CK(hipMalloc(&X, total)); CK(hipMalloc(&Y, total)); CK(hipMalloc(&W, total));
    CK(hipMemset(X, 0, total)); CK(hipMemset(W, 0, total));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%-40s %10s %12s %s\n", "intermediate per chunk", "ms/2GiB", "GB/s (4x)", "(bytes moved = 4 x 2 GiB: X read, W write, W read, Y write)");
    for (int variant = 0; variant < 2; ++variant) {
        for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 512, 2048}) {
            const size_t S = mb << 20, n = S / 16, chunks = total / S;
            auto pass = [&] {
                for (size_t c = 0; c < chunks; ++c) {
                    const f4 *x = X + c * n; f4 *y = Y + c * n;
                    f4 *w = (mb == 2048) ? W : W;   // reused buffer (the whole W when S = 2 GiB)
                    if (variant == 0) {
                        hipLaunchKernelGGL((copy1<true, false>), dim3((unsigned) (n / 256)), dim3(256), 0, 0, x, w, n);     // X nt, W cached
                        hipLaunchKernelGGL((copy1<false, true>), dim3((unsigned) (n / 256)), dim3(256), 0, 0, (const f4 *) w, y, n);     // W cached, Y nt
                    } else {
                        hipLaunchKernelGGL((copy1<false, false>), dim3((unsigned) (n / 256)), dim3(256), 0, 0, x, w, n);
                        hipLaunchKernelGGL((copy1<false, false>), dim3((unsigned) (n / 256)), dim3(256), 0, 0, (const f4 *) w, y, n);
                    }
                }
            };
            pass(); pass(); CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(a)); for (int i = 0; i < 4; ++i) pass(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms / 4 < best) best = ms / 4;
            }
            char name[96]; snprintf(name, sizeof name, "%s W = %4zu MiB x %3zu chunks", variant == 0 ? "nt ext / cached W:" : "all default:      ", mb, chunks);
            printf("%-40s %10.3f %12.0f\n", name, best, 4.0 * total / best / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
