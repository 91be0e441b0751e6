// Compile-and-link check of dsc_amd/api/dsc_api.h against libdsc_mi355x.so; with a GPU it also
// runs the README's C++ filterFFT shape (README.md:141-163) on 65536 samples.
#include "dsc_api.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2 || std::atoi(argv[1]) == 0) {
        std::printf("linked: %p\n", (void *) &dsc_rfft);
        return 0;
    }
    dsc::init((size_t) 1 << 30);
    const int n = 65536;
    std::vector<float> s(n), b(n, 0.f);
    for (int i = 0; i < n; ++i) s[i] = std::sin(0.001f * i) + 0.5f * std::sin(1.3f * i);
    for (int i = 0; i < 64; ++i) b[i] = 1.f / 64.f;                 // moving average
    dsc::tensor<float> ts(s.data(), {1, n}), tb(b.data(), n);
    auto S = dsc::rfft(ts);
    auto B = dsc::rfft(tb);
    auto y = dsc::irfft(S * B);
    auto yf = dsc::filter_fft(ts, B);
    auto h1 = y.to_host(), h2 = yf.to_host();
    double diff = 0, ref = 0;
    for (int i = 0; i < n; ++i) { diff += (h1[i] - h2[i]) * (h1[i] - h2[i]); ref += h1[i] * h1[i]; }
    std::printf("fused vs composed rel-L2 %.3e\n", std::sqrt(diff / ref));
    // the README's `[:output_length]` crop (README.md:133), on the device
    auto crop = y.get(DSC_SLICE_ALL(), DSC_SLICE_TO(1000));
    auto hc = crop.to_host();
    bool crop_ok = crop.dim(-1) == 1000 && hc.size() == 1000;
    for (int i = 0; i < 1000 && crop_ok; ++i) crop_ok = hc[i] == h1[i];
    std::printf("crop %s\n", crop_ok ? "ok" : "MISMATCH");
    return std::sqrt(diff / ref) < 1e-5 && crop_ok ? 0 : 1;
}
