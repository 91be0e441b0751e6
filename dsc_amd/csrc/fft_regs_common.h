// fft_regs_common.h — device building blocks of the register-resident FFT kernels (fft_regs_mid.hip,
// fft_r2c_2pass.hip): complex arithmetic, in-register radix-2 DIF DFTs of 2..32 points with
// compile-time twiddles (results in bit-reversed register order), the LDS-only barrier and buffer
// load / store wrappers.  Header-only, everything in an anonymous namespace.
#pragma once

#include <hip/hip_runtime.h>

#include <utility>

namespace {


template<typename R> struct alignas(2 * sizeof(R)) cpx { R x, y; };

template<typename R> __device__ __forceinline__ cpx<R> operator+(cpx<R> a, cpx<R> b) { return cpx<R>{a.x + b.x, a.y + b.y}; }
template<typename R> __device__ __forceinline__ cpx<R> operator-(cpx<R> a, cpx<R> b) { return cpx<R>{a.x - b.x, a.y - b.y}; }
template<typename R> __device__ __forceinline__ cpx<R> cmul(cpx<R> a, cpx<R> w) { return cpx<R>{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
template<typename R> __device__ __forceinline__ cpx<R> cmulc(cpx<R> a, cpx<R> w) { return cpx<R>{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }

__host__ __device__ constexpr int brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
__host__ __device__ constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

// cos(2 pi q / 64), q = 0..16
__device__ constexpr double kCos64[17] = {
    1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
    0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081,
    0.70710678118654752440, 0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856,
    0.38268343236508977173, 0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199,
    0.0};
__device__ constexpr double root64_re(int q) {
    q &= 63;
    return q <= 16 ? kCos64[q] : q <= 32 ? -kCos64[32 - q] : q <= 48 ? -kCos64[q - 32] : kCos64[64 - q];
}
__device__ constexpr double root64_im(int q) {      // -sin(2 pi q / 64)
    q &= 63;
    return q <= 16 ? -kCos64[16 - q] : q <= 32 ? -kCos64[q - 16] : q <= 48 ? kCos64[48 - q] : kCos64[q - 48];
}

// d * W_M^K (forward) or d * conj(W_M^K) (INV), K < M/2, M <= 32
template<typename R, bool INV, int M, int K>
__device__ __forceinline__ cpx<R> mul_root(cpx<R> d) {
    constexpr R c8 = (R) 0.70710678118654752440;
    if constexpr (K == 0) {
        return d;
    } else if constexpr (4 * K == M) {
        return INV ? cpx<R>{-d.y, d.x} : cpx<R>{d.y, -d.x};
    } else if constexpr (8 * K == M) {
        return INV ? cpx<R>{(d.x - d.y) * c8, (d.x + d.y) * c8} : cpx<R>{(d.x + d.y) * c8, (d.y - d.x) * c8};
    } else if constexpr (8 * K == 3 * M) {
        return INV ? cpx<R>{-(d.x + d.y) * c8, (d.x - d.y) * c8} : cpx<R>{(d.y - d.x) * c8, -(d.x + d.y) * c8};
    } else {
        constexpr R wr = (R) root64_re(K * (64 / M));
        constexpr R wi = (R) (INV ? -root64_im(K * (64 / M)) : root64_im(K * (64 / M)));
        return cpx<R>{d.x * wr - d.y * wi, d.x * wi + d.y * wr};
    }
}

#ifdef DSC_DFT_DIF
template<typename R, bool INV, int M, int G, int... K>
__device__ __forceinline__ void dif_group(cpx<R> (&v)[32], std::integer_sequence<int, K...>) {
    (([&] {
         const cpx<R> u = v[G + K] + v[G + K + M / 2];
         const cpx<R> d = v[G + K] - v[G + K + M / 2];
         v[G + K] = u;
         v[G + K + M / 2] = mul_root<R, INV, M, K>(d);
     }()),
     ...);
}
template<typename R, bool INV, int M, int BASE, int... G>
__device__ __forceinline__ void dif_stage(cpx<R> (&v)[32], std::integer_sequence<int, G...>) {
    (dif_group<R, INV, M, BASE + G * M>(v, std::make_integer_sequence<int, M / 2>{}), ...);
}
// N-point DFT (N = 2 .. 32) of v[BASE .. BASE+N), natural order in; v[BASE + p] returns bin brev(p, log2 N)
template<typename R, bool INV, int N, int BASE = 0>
__device__ __forceinline__ void dft_n(cpx<R> (&v)[32]) {
    if constexpr (N >= 32) dif_stage<R, INV, 32, BASE>(v, std::make_integer_sequence<int, N / 32>{});
    if constexpr (N >= 16) dif_stage<R, INV, 16, BASE>(v, std::make_integer_sequence<int, N / 16>{});
    if constexpr (N >= 8)  dif_stage<R, INV, 8, BASE>(v, std::make_integer_sequence<int, N / 8>{});
    if constexpr (N >= 4)  dif_stage<R, INV, 4, BASE>(v, std::make_integer_sequence<int, N / 4>{});
    dif_stage<R, INV, 2, BASE>(v, std::make_integer_sequence<int, N / 2>{});
}
#else
// The decimation-in-time graph with natural-order input and bit-reversed output (same contract as the DIF form above): stage s
// (half distance H = N >> s) has 2^(s-1) groups, group g multiplies the LOWER input of its butterflies by ONE constant
// e^{-i pi theta_g}, theta_g = brev(g, s-1) / 2^(s-1) — a DFT with frequency offset theta splits into two of offsets theta / 2
// and (1 + theta) / 2 — then adds and subtracts.  A constant twiddle in FRONT of the add / sub costs 6 fused multiply-adds
// instead of the 8 operations of "subtract, then multiply" (Linzer & Feig): w = c (1 + i t), p = b.x - t b.y, q = b.y + t b.x,
// out = a +- c (p, q); the larger of |cos|, |sin| is factored out so that |t| <= 1.
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a, b -> a + w b, a - w b with w = W_64^Q (forward) or its conjugate (INV), 0 <= Q < 32
template<typename R, bool INV, int Q>
__device__ __forceinline__ void dit_butterfly(cpx<R> &a, cpx<R> &b) {
    if constexpr (Q == 0) {
        const cpx<R> u = a + b, d = a - b;
        a = u; b = d;
    } else if constexpr (Q == 16) {                         // w = -i (forward), +i (inverse)
        const cpx<R> u = INV ? cpx<R>{a.x - b.y, a.y + b.x} : cpx<R>{a.x + b.y, a.y - b.x};
        const cpx<R> d = INV ? cpx<R>{a.x + b.y, a.y - b.x} : cpx<R>{a.x - b.y, a.y + b.x};
        a = u; b = d;
    } else {
        constexpr double c = root64_re(Q);
        constexpr double s = INV ? -root64_im(Q) : root64_im(Q);                        // w = c + i s
        if constexpr ((c < 0 ? -c : c) >= (s < 0 ? -s : s)) {
            constexpr R t = (R) (s / c), cc = (R) c;
            R p, q;
            if constexpr (t == (R) 1) { p = b.x - b.y; q = b.y + b.x; }
            else if constexpr (t == (R) -1) { p = b.x + b.y; q = b.y - b.x; }
            else { p = fma_r(-t, b.y, b.x); q = fma_r(t, b.x, b.y); }
            const cpx<R> u = cpx<R>{fma_r(cc, p, a.x), fma_r(cc, q, a.y)};
            const cpx<R> d = cpx<R>{fma_r(-cc, p, a.x), fma_r(-cc, q, a.y)};
            a = u; b = d;
        } else {                                            // w b = s (r b.x - b.y, r b.y + b.x), r = c / s
            constexpr R r = (R) (c / s), ss = (R) s;
            const R pn = fma_r(-r, b.x, b.y);               // -(r b.x - b.y)
            const R q = fma_r(r, b.y, b.x);
            const cpx<R> u = cpx<R>{fma_r(-ss, pn, a.x), fma_r(ss, q, a.y)};
            const cpx<R> d = cpx<R>{fma_r(ss, pn, a.x), fma_r(-ss, q, a.y)};
            a = u; b = d;
        }
    }
}
// stage S = 1 .. log2 N of an N-point transform at v[BASE ..): H = N >> S, group G covers 2 H registers
template<typename R, bool INV, int N, int S, int BASE, int G, int... J>
__device__ __forceinline__ void dit_group(cpx<R> (&v)[32], std::integer_sequence<int, J...>) {
    constexpr int H = N >> S;
    constexpr int Q = (32 * brev(G, S - 1)) >> (S - 1);                                 // W_64 exponent = 32 theta_g
    (dit_butterfly<R, INV, Q>(v[BASE + 2 * H * G + J], v[BASE + 2 * H * G + J + H]), ...);
}
template<typename R, bool INV, int N, int S, int BASE, int... G>
__device__ __forceinline__ void dit_stage(cpx<R> (&v)[32], std::integer_sequence<int, G...>) {
    (dit_group<R, INV, N, S, BASE, G>(v, std::make_integer_sequence<int, (N >> S)>{}), ...);
}
// N-point DFT (N = 2 .. 32) of v[BASE .. BASE+N), natural order in; v[BASE + p] returns bin brev(p, log2 N)
template<typename R, bool INV, int N, int BASE = 0>
__device__ __forceinline__ void dft_n(cpx<R> (&v)[32]) {
    dit_stage<R, INV, N, 1, BASE>(v, std::make_integer_sequence<int, 1>{});
    if constexpr (N >= 4)  dit_stage<R, INV, N, 2, BASE>(v, std::make_integer_sequence<int, 2>{});
    if constexpr (N >= 8)  dit_stage<R, INV, N, 3, BASE>(v, std::make_integer_sequence<int, 4>{});
    if constexpr (N >= 16) dit_stage<R, INV, N, 4, BASE>(v, std::make_integer_sequence<int, 8>{});
    if constexpr (N >= 32) dit_stage<R, INV, N, 5, BASE>(v, std::make_integer_sequence<int, 16>{});
}
#endif
template<typename R, bool INV, int N, int... I>
__device__ __forceinline__ void dft_columns(cpx<R> (&v)[32], std::integer_sequence<int, I...>) {
    (dft_n<R, INV, N, I * N>(v), ...);
}

// LDS-only barrier: does not wait for outstanding global loads / stores
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Buffer accesses: voff (VGPR, bytes) + soff (SGPR / literal, bytes) against a wave-uniform descriptor.
// Out-of-range lanes read zero and their stores are dropped, which is how a partially filled last
// group and the "no such line" waves are handled (num_records = 0).
// POL = cache policy (aux bits): kCached (default) or kStream (non-temporal).  Row data is touched once, so
// streaming helps (+1..6 % measured) — but only where a wave's accesses add up to whole 128-B lines: the
// skewed spectrum rows of the real transforms (pitch L + 1) lose up to 30 % with it and stay on kCached.
constexpr int kCached = 0, kStream = 2;
template<int POL = kCached>
__device__ __forceinline__ cpx<float> buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff, float) {
    const f2 q = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, POL));
    return cpx<float>{q.x, q.y};
}
template<int POL = kCached>
__device__ __forceinline__ cpx<double> buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff, double) {
    const d2 q = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, POL));
    return cpx<double>{q.x, q.y};
}
// real sample widened to complex (dsc_fft on real input casts first: dsc.cpp:1984-1988)
template<int POL = kCached>
__device__ __forceinline__ cpx<float> buf_load_real(__amdgpu_buffer_rsrc_t r, int voff, int soff, float) {
    return cpx<float>{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, POL)), 0.0f};
}
template<int POL = kCached>
__device__ __forceinline__ cpx<double> buf_load_real(__amdgpu_buffer_rsrc_t r, int voff, int soff, double) {
    return cpx<double>{__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, POL)), 0.0};
}
template<int POL = kCached>
__device__ __forceinline__ void buf_store(cpx<float> a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, f2{a.x, a.y}), r, voff, soff, POL);
}
template<int POL = kCached>
__device__ __forceinline__ void buf_store(cpx<double> a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u4 data = __builtin_bit_cast(u4, d2{a.x, a.y});
    __builtin_amdgcn_raw_buffer_store_b128(data, r, voff, soff, POL);
    // The data registers of a 128-bit buffer store must not be rewritten by a VALU instruction in the next two issue slots: on
    // gfx950, with f64 data at two or more waves per SIMD, the last quad of every 16-lane row now and then stores the NEXT value
    // (hipcc pads this hazard only for a literal soffset).  Found by tools/stress_fused.py, and again — one row in a few thousand,
    // bins 396-399, 412-415, 428-431, 444-447: lanes 12-15 of each row of one store instruction — by
    // tests/test_gpu_headline.py::test_every_element_of_the_f64_paths when the 2048-point f64 configuration changed its schedule.
    // A bare s_nop after the store is not enough: the scheduler may move the overwriting instruction in front of it.  The wait
    // states therefore READ the data registers, so that nothing can overwrite them before the nop has issued.
    // (DSC_NO_STORE_HAZARD_PAD exists for tests/test_abi.py only: the build-time scan of check_store_hazard.py must then fail)
#ifndef DSC_NO_STORE_HAZARD_PAD
    asm volatile("s_nop 1" : : "v"(data));
#endif
}
// two neighbouring complex values (16 / 32 bytes) in one go
template<int POL = kCached>
__device__ __forceinline__ void buf_store_pair(cpx<float> a, cpx<float> b, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const u4 data = __builtin_bit_cast(u4, f4{a.x, a.y, b.x, b.y});
    __builtin_amdgcn_raw_buffer_store_b128(data, r, voff, soff, POL);
#ifndef DSC_NO_STORE_HAZARD_PAD
    asm volatile("s_nop 1" : : "v"(data));                // the 128-bit store-data hazard, see above
#endif
}
template<int POL = kCached>
__device__ __forceinline__ void buf_store_pair(cpx<double> a, cpx<double> b, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    buf_store<POL>(a, r, voff, soff);
    buf_store<POL>(b, r, voff, soff + 16);
}
template<int POL = kCached>
__device__ __forceinline__ void buf_load_pair(cpx<float> &a, cpx<float> &b, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 q = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, POL));
    a = cpx<float>{q.x, q.y};
    b = cpx<float>{q.z, q.w};
}
template<int POL = kCached>
__device__ __forceinline__ void buf_load_pair(cpx<double> &a, cpx<double> &b, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    a = buf_load<POL>(r, voff, soff, double{});
    b = buf_load<POL>(r, voff, soff + 16, double{});
}

}  // namespace
