// tools/probe_generic.hip — times fft_lines_kernel (R2C_PACKED, f32) directly, with parts compiled
// out via -DDSC_GEN_SKIP (1 stages); diagnostic only.
#include "../dsc_amd/csrc/fft_generic.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096;
    const long long B = argc > 2 ? atoll(argv[2]) : 65536;
    const int L = N / 2;
    float *x; float *X; float *tw, *twr;
    CK(hipMalloc(&x, B * N * 4)); CK(hipMalloc(&X, B * (L + 1) * 8));
    CK(hipMemset(x, 0x3c, B * N * 4));
    std::vector<float> h(2 * L), hr(2 * (L + 1));
    for (int k = 0; k < L; ++k) { h[2 * k] = cos(-2 * M_PI * k / L); h[2 * k + 1] = sin(-2 * M_PI * k / L); }
    for (int k = 0; k <= L; ++k) { hr[2 * k] = cos(-2 * M_PI * k / (2.0 * L)); hr[2 * k + 1] = sin(-2 * M_PI * k / (2.0 * L)); }
    CK(hipMalloc(&tw, h.size() * 4)); CK(hipMalloc(&twr, hr.size() * 4));
    CK(hipMemcpy(tw, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(twr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    dsc_fft_lines_args a;
    a.in = x; a.out = X; a.n_lines = B; a.inner = 1;
    a.lin = dsc_line_layout{N, 1, 1}; a.lout = dsc_line_layout{L + 1, 1, 1};
    a.L = L; a.in_len = N; a.inverse = 0; a.scale = 1.0; a.tw = tw; a.tw_real = twr; a.tw4_len = 0; a.tw4 = nullptr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) dsc_launch_fft_lines(a, DSC_MODE_R2C_PACKED, true, 0);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) dsc_launch_fft_lines(a, DSC_MODE_R2C_PACKED, true, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = (double) B * (N * 4 + (L + 1) * 8);
    printf("N=%d B=%lld skip=%d: %.3f ms  %.0f GB/s\n", N, B, DSC_GEN_SKIP, ms, bytes / ms / 1e6);
    return 0;
}
