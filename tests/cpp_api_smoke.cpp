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

    // operators of dsc/api/dsc_api.h:148-186 against host arithmetic (exact: one IEEE operation per element)
    const int m = 6;
    const float ha[m] = {1.5f, -2.f, 3.25f, 4.f, -5.5f, 6.f}, hb[m] = {0.5f, 4.f, -1.25f, 8.f, 2.f, -3.f};
    dsc::tensor<float> ta(ha, {2, 3}), tbb(hb, {2, 3});
    bool ops_ok = true;
    auto check = [&](const dsc::tensor<float> &t, auto f, const char *what) {
        auto h = t.to_host();
        for (int i = 0; i < m; ++i)
            if (h[i] != f(i)) { ops_ok = false; std::printf("operator %s: element %d is %g, want %g\n", what, i, h[i], f(i)); }
    };
    check(ta + tbb, [&](int i) { return ha[i] + hb[i]; }, "+");
    check(ta - tbb, [&](int i) { return ha[i] - hb[i]; }, "-");
    check(ta * tbb, [&](int i) { return ha[i] * hb[i]; }, "*");
    check(ta / tbb, [&](int i) { return ha[i] / hb[i]; }, "/");
    check(ta * 2.5f, [&](int i) { return ha[i] * 2.5f; }, "* scalar");
    check(2.5f * ta, [&](int i) { return 2.5f * ha[i]; }, "scalar *");
    check(ta + 1.f, [&](int i) { return ha[i] + 1.f; }, "+ scalar");
    check(1.f - ta, [&](int i) { return 1.f - ha[i]; }, "scalar -");
    check(ta / 4.f, [&](int i) { return ha[i] / 4.f; }, "/ scalar");
    {
        dsc::tensor<float> c(ta);                                    // deep copy (dsc_api.h:63-66)
        c /= tbb;                                                     // in place (dsc_api.h:180-183)
        check(c, [&](int i) { return ha[i] / hb[i]; }, "/=");
        check(ta, [&](int i) { return ha[i]; }, "copy is deep");
        dsc::tensor<float> d;
        d = tbb;                                                      // copy assignment (dsc_api.h:74-81)
        d /= d;
        check(d, [&](int) { return 1.f; }, "copy-assign + /=");
        check(tbb, [&](int i) { return hb[i]; }, "copy-assign is deep");
    }
    {
        auto t = dsc::transpose(ta);                                 // dsc_api.h:294-302: [2,3] -> [3,2]
        auto h = t.to_host();
        bool ok = t.dim(0) == 3 && t.dim(1) == 2;
        for (int i = 0; i < 3 && ok; ++i) for (int j = 0; j < 2; ++j) ok = ok && h[i * 2 + j] == ha[j * 3 + i];
        auto t2 = dsc::transpose(ta, 1, 0).to_host();
        for (int i = 0; i < m; ++i) ok = ok && t2[i] == h[i];
        if (!ok) { ops_ok = false; std::printf("transpose MISMATCH\n"); }
        dsc::tensor<float> filled({2, 2}, 7.f), lst{1.f, 2.f, 3.f};
        auto hf = filled.to_host(), hl = lst.to_host();
        if (!(hf.size() == 4 && hf[3] == 7.f && hl.size() == 3 && hl[2] == 3.f)) { ops_ok = false; std::printf("ctor MISMATCH\n"); }
    }
    {
        const dsc_c32 hz[2] = {{1.f, 2.f}, {-3.f, 0.5f}};
        dsc::tensor<dsc_c32> z(hz, 2);
        auto hw = (z * dsc_c32{0.f, 1.f}).to_host();               // times i: (re, im) -> (-im, re)
        if (!(hw[0].real == -2.f && hw[0].imag == 1.f && hw[1].real == -0.5f && hw[1].imag == -3.f)) { ops_ok = false; std::printf("complex scalar * MISMATCH\n"); }
    }
    std::printf("operators %s\n", ops_ok ? "ok" : "MISMATCH");
    return std::sqrt(diff / ref) < 1e-5 && crop_ok && ops_ok ? 0 : 1;
}
