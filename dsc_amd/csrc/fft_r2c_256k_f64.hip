// fft_r2c_256k_f64.hip — 262144-point real FFT in f64 (BASELINE config 5): packed length
// L = 131072 complex = 2 MiB per row, which no CU can hold, so the transform is split once:
//
//   L = 8 x 16384,   j = j1 + 16384 j2   (input),   k = 8 k1 + k2   (output)
//
//   A  radix-8 over j2 (stride 16384: coalesced across j1), times W_L^{j1 k2}            streaming
//   B  8 independent 16384-point c64 FFTs over j1 on contiguous 256 KiB segments, each
//      RESIDENT IN REGISTERS of one 512-thread workgroup (32 complex per thread, two waves
//      per SIMD, up to 256 VGPRs): 16384 = 32 x 32 x 16, the same three-pass /
//      two-LDS-transpose pipeline as the f32 65536-point kernel (fft_r2c_64k.hip)
//   C  packed-real post-pass (dsc_fft.h:199-225) reading Z[k] = seg[k & 7][k >> 3]: eight
//      128-B streams per wave, i.e. the transposition back to natural order costs nothing extra
//
// Three passes over HBM (12 MiB per row against 4 MiB algorithmic) instead of the generic
// path's three LATENCY-bound LDS passes.  The inverse runs C', B (conjugate), A' backwards.
// Reference: dsc_rfft / dsc_irfft for F64 / C64 (dsc/src/dsc.cpp:2102-2260, dsc_fft.h:57-238).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <utility>

namespace {

struct cd { double x, y; };
typedef double d2 __attribute__((ext_vector_type(2)));      // 16-B memory-op type

__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd w) { return cd{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ cd cmulc(cd a, cd w) { return cd{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }   // a * conj(w)
__device__ __forceinline__ cd ld(const cd *p) { const d2 t = *(const d2 *) p; return cd{t.x, t.y}; }
__device__ __forceinline__ void st(cd *p, cd a) { *(d2 *) p = d2{a.x, a.y}; }

constexpr int kL = 131072;          // packed complex length
constexpr int kSeg = 16384;         // points per register-resident FFT
constexpr int kThreadsB = 512;
constexpr int kPitch1 = 33;         // doubles per LDS row, exchange 1 (512 rows x 32 values): odd -> conflict-free b64
constexpr int kPitch2 = 17;         // exchange 2 (1024 rows x 16 values)
constexpr int kPlaneDoubles = 1024 * kPitch2;               // 17408 >= 512 * 33 = 16896
constexpr int kLdsBytesB = kPlaneDoubles * 8 + 1024 * 16;   // plane + W_1024 table (c64)

// aux table layout (cd entries): W_1024^m, m < 1024 | W_16384^m, m < 512 | W_131072^m, m < 16384
constexpr int kAuxW1024 = 0, kAuxW16384 = 1024, kAuxWL = 1536, kAuxEntries = 1536 + 16384;

__host__ __device__ constexpr int brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// cos(2 pi q / 64), q = 0..16, to double precision
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
template<bool INV, int M, int K>
__device__ __forceinline__ cd mul_root(cd d) {
    constexpr double c8 = 0.70710678118654752440;
    if constexpr (K == 0) {
        return d;
    } else if constexpr (4 * K == M) {
        return INV ? cd{-d.y, d.x} : cd{d.y, -d.x};
    } else if constexpr (8 * K == M) {
        return INV ? cd{(d.x - d.y) * c8, (d.x + d.y) * c8} : cd{(d.x + d.y) * c8, (d.y - d.x) * c8};
    } else if constexpr (8 * K == 3 * M) {
        return INV ? cd{-(d.x + d.y) * c8, (d.x - d.y) * c8} : cd{(d.y - d.x) * c8, -(d.x + d.y) * c8};
    } else {
        constexpr double wr = root64_re(K * (64 / M));
        constexpr double wi = INV ? -root64_im(K * (64 / M)) : root64_im(K * (64 / M));
        return cd{d.x * wr - d.y * wi, d.x * wi + d.y * wr};
    }
}

template<bool INV, int TOT, int M, int G, int... K>
__device__ __forceinline__ void dif_group(cd (&v)[TOT], std::integer_sequence<int, K...>) {
    (([&] {
         const cd u = v[G + K] + v[G + K + M / 2];
         const cd d = v[G + K] - v[G + K + M / 2];
         v[G + K] = u;
         v[G + K + M / 2] = mul_root<INV, M, K>(d);
     }()),
     ...);
}
template<bool INV, int TOT, int M, int BASE, int... G>
__device__ __forceinline__ void dif_stage(cd (&v)[TOT], std::integer_sequence<int, G...>) {
    (dif_group<INV, TOT, M, BASE + G * M>(v, std::make_integer_sequence<int, M / 2>{}), ...);
}
// N-point DFT (N = 8, 16, 32) of v[BASE .. BASE+N), natural order in; v[BASE + p] returns bin brev(p, log2 N).
template<bool INV, int N, int TOT, int BASE = 0>
__device__ __forceinline__ void dft_n(cd (&v)[TOT]) {
    if constexpr (N >= 32) dif_stage<INV, TOT, 32, BASE>(v, std::make_integer_sequence<int, N / 32>{});
    if constexpr (N >= 16) dif_stage<INV, TOT, 16, BASE>(v, std::make_integer_sequence<int, N / 16>{});
    if constexpr (N >= 8)  dif_stage<INV, TOT, 8, BASE>(v, std::make_integer_sequence<int, N / 8>{});
    if constexpr (N >= 4)  dif_stage<INV, TOT, 4, BASE>(v, std::make_integer_sequence<int, N / 4>{});
    dif_stage<INV, TOT, 2, BASE>(v, std::make_integer_sequence<int, N / 2>{});
}

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ------------------------------------------------------------------------------------------
// B: 16384-point complex FFT of every contiguous 256 KiB segment, in place, in registers.
//   j = 512 j1 + 16 j2 + j3  (j1, j2 < 32, j3 < 16),   k = k1 + 32 k2 + 1024 k3
//   thread t = 16 j2 + j3 loads seg[512 j1 + t]; pass 1 over j1 (dft32), x W_1024^{j2 k1};
//   exchange 1 -> thread 16 k1 + j3 holds [j2]; pass 2 over j2 (dft32), x W_16384^{j3 k1} W_512^{j3 k2};
//   exchange 2 -> thread tau holds columns k' = tau and tau + 512, [j3] each; pass 3: two dft16.
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// Thread-derived values are rebuilt per phase from the wave number (SGPR) and mbcnt: hipcc would
// otherwise hoist every address out of the persistent loop and spill it (see fft_r2c_64k.hip).
__device__ __forceinline__ int thread_id(int wave_sgpr) {
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const int lane = __builtin_amdgcn_mbcnt_hi(-1, __builtin_amdgcn_mbcnt_lo(-1, zero));
    return (wave_sgpr << 6) | lane;
}
__device__ __forceinline__ cd load_seg(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const d2 t = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return cd{t.x, t.y};
}
__device__ __forceinline__ void store_seg(cd a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d2{a.x, a.y}), r, voff, soff, 0);
}
// plane writes: 32 slots of `slot_stride` doubles; two base registers keep every offset a 16-bit immediate
template<int COMP, int SLOT_STRIDE>
__device__ __forceinline__ void plane_write(double *plane, int wbase, const cd (&v)[32]) {
    double *lo16 = plane + wbase;
    double *hi16 = lo16 + 16 * SLOT_STRIDE;
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const int slot = brev(p, 5);
        const double val = COMP == 0 ? v[p].x : v[p].y;
        if (slot < 16) lo16[slot * SLOT_STRIDE] = val;
        else           hi16[(slot - 16) * SLOT_STRIDE] = val;
    }
}

template<bool INV>
__global__ __launch_bounds__(kThreadsB) void fft16k_f64_kernel(cd *__restrict__ buf, long long n_seg, const cd *__restrict__ aux) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *plane = lds;
    cd *w1024 = (cd *) (lds + kPlaneDoubles);
    for (int i = threadIdx.x; i < 1024; i += kThreadsB) w1024[i] = aux[kAuxW1024 + i];
    __syncthreads();
    const int wave_sgpr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    for (long long seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rseg = __builtin_amdgcn_make_buffer_rsrc((void *) (buf + seg * kSeg), 0, kSeg * 16, 0x00020000);
        cd v[32];
        {
            const int t = thread_id(wave_sgpr);
#pragma unroll
            for (int j1 = 0; j1 < 32; ++j1) v[j1] = load_seg(rseg, t * 16, j1 * 8192);          // seg[512 j1 + t]
        }
        // ---- pass 1 over j1, twiddle W_1024^{j2 k1}
        dft_n<INV, 32, 32>(v);
        cd u[32];
        {
            const int t = thread_id(wave_sgpr);
            const int hi = t >> 4, lo = t & 15;
#pragma unroll
            for (int k1 = 1; k1 < 32; ++k1) {
                const cd w = w1024[hi * k1];
                v[brev(k1, 5)] = INV ? cmulc(v[brev(k1, 5)], w) : cmul(v[brev(k1, 5)], w);
            }
            // ---- exchange 1: (j2, j3)[k1] -> thread 16 k1 + j3, [j2]; row = 16 k1 + j3, col = j2
            const int wbase = lo * kPitch1 + hi;
            plane_write<0, 16 * kPitch1>(plane, wbase, v);
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 32; ++m) u[m].x = plane[t * kPitch1 + m];
            lds_barrier();
            plane_write<1, 16 * kPitch1>(plane, wbase, v);
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 32; ++m) u[m].y = plane[t * kPitch1 + m];
            lds_barrier();
        }
        // ---- pass 2 over j2, twiddle W_16384^{j3 k1} W_512^{j3 k2}
        dft_n<INV, 32, 32>(u);
        {
            const int t = thread_id(wave_sgpr);
            const int hi = t >> 4, lo = t & 15;                    // (k1, j3)
            const cd tw2_base = aux[kAuxW16384 + hi * lo];
            u[0] = INV ? cmulc(u[0], tw2_base) : cmul(u[0], tw2_base);
#pragma unroll
            for (int k2 = 1; k2 < 32; ++k2) {
                const cd w = cmul(tw2_base, w1024[2 * lo * k2]);   // W_512^m = W_1024^{2m}
                u[brev(k2, 5)] = INV ? cmulc(u[brev(k2, 5)], w) : cmul(u[brev(k2, 5)], w);
            }
            // ---- exchange 2: (k1, j3)[k2] -> columns k' = k1 + 32 k2; row = k', col = j3.
            // v[0..15] = column t, v[16..31] = column t + 512, natural j3 order.
            const int wbase = hi * kPitch2 + lo;
            plane_write<0, 32 * kPitch2>(plane, wbase, u);
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 16; ++m) { v[m].x = plane[t * kPitch2 + m]; v[16 + m].x = plane[(t + 512) * kPitch2 + m]; }
            lds_barrier();
            plane_write<1, 32 * kPitch2>(plane, wbase, u);
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 16; ++m) { v[m].y = plane[t * kPitch2 + m]; v[16 + m].y = plane[(t + 512) * kPitch2 + m]; }
            lds_barrier();
        }
        // ---- pass 3 over j3: two 16-point DFTs; v[p] = bin brev(p,4) of column t, v[16+p] of column t+512
        dft_n<INV, 16, 32, 0>(v);
        dft_n<INV, 16, 32, 16>(v);
        {
            const int t = thread_id(wave_sgpr);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                store_seg(v[p], rseg, t * 16, brev(p, 4) * 16384);                // seg[t + 1024 k3]
                store_seg(v[16 + p], rseg, (t + 512) * 16, brev(p, 4) * 16384);
            }
        }
    }
}

// radix-8 DFT of v[0..8) in registers: v[p] returns bin brev(p, 3)
template<bool INV>
__device__ __forceinline__ void dft8(cd (&v)[8]) { dft_n<INV, 8, 8>(v); }

// ------------------------------------------------------------------------------------------
// A (forward): z = packed reals of one row viewed as L complex; out[k2 * 16384 + j1] =
//   W_L^{j1 k2} * sum_{j2} z[j1 + 16384 j2] W_8^{j2 k2}
__global__ void radix8_in_kernel(const cd *__restrict__ x, cd *__restrict__ work, long long n_rows, const cd *__restrict__ aux) {
    const long long total = n_rows * kSeg;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x) {
        const long long row = i >> 14;
        const int j1 = (int) (i & (kSeg - 1));
        const cd *z = x + row * kL + j1;
        cd v[8];
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) v[j2] = ld(z + (long long) kSeg * j2);
        dft8<false>(v);
        const cd w1 = aux[kAuxWL + j1];                      // W_L^{j1}; powers by products of at most three factors
        const cd w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
        const cd w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
        cd *o = work + row * kL + j1;
        st(o, v[0]);                                          // k2 = brev(p, 3)
        st(o + 1LL * kSeg, cmul(v[4], w1));
        st(o + 2LL * kSeg, cmul(v[2], w2));
        st(o + 3LL * kSeg, cmul(v[6], w3));
        st(o + 4LL * kSeg, cmul(v[1], w4));
        st(o + 5LL * kSeg, cmul(v[5], w5));
        st(o + 6LL * kSeg, cmul(v[3], w6));
        st(o + 7LL * kSeg, cmul(v[7], w7));
    }
}

// A' (inverse): z[j1 + 16384 j2] = scale * sum_{k2} conj(W_L^{j1 k2}) work[k2 * 16384 + j1] conj(W_8^{j2 k2})
__global__ void radix8_out_kernel(const cd *__restrict__ work, cd *__restrict__ y, long long n_rows, const cd *__restrict__ aux, double scale) {
    const long long total = n_rows * kSeg;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x) {
        const long long row = i >> 14;
        const int j1 = (int) (i & (kSeg - 1));
        const cd *s = work + row * kL + j1;
        const cd w1 = aux[kAuxWL + j1];
        const cd w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
        const cd w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
        cd v[8];
        v[0] = ld(s);
        v[1] = cmulc(ld(s + 1LL * kSeg), w1);
        v[2] = cmulc(ld(s + 2LL * kSeg), w2);
        v[3] = cmulc(ld(s + 3LL * kSeg), w3);
        v[4] = cmulc(ld(s + 4LL * kSeg), w4);
        v[5] = cmulc(ld(s + 5LL * kSeg), w5);
        v[6] = cmulc(ld(s + 6LL * kSeg), w6);
        v[7] = cmulc(ld(s + 7LL * kSeg), w7);
        dft8<true>(v);
        cd *o = y + row * kL + j1;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const cd r = v[p];
            st(o + (long long) kSeg * brev(p, 3), cd{r.x * scale, r.y * scale});
        }
    }
}

// where bin k of the 131072-point spectrum sits after pass B
__device__ __forceinline__ long long zidx(int k) { return (long long) (k & 7) * kSeg + (k >> 3); }

// C (forward post-pass, dsc_fft.h:199-225).  One thread per PAIR of bins (k, L-k), k = 0..L/2: every
// Z element is read once, both outputs leave as coalesced 16-B stores (ascending / descending).
//   s = a + conj b, d = a - conj b, wq = -(i/2) W_{2L}^k:  X[k] = s/2 + wq d,  X[L-k] = conj(s/2 - wq d)
__global__ void r2c_post_f64_kernel(const cd *__restrict__ work, cd *__restrict__ out, const cd *__restrict__ tw_real) {
    const long long row = blockIdx.y;
    const cd *z = work + row * kL;
    cd *o = out + row * (kL + 1LL);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= kL / 2; k += gridDim.x * blockDim.x) {
        const cd a = ld(z + zidx(k)), b = ld(z + zidx(k == 0 ? 0 : kL - k));
        const cd w = ld(tw_real + k);
        const cd wq = cd{0.5 * w.y, -0.5 * w.x};
        const cd sm = cd{a.x + b.x, a.y - b.y}, d = cd{a.x - b.x, a.y + b.y};
        const cd wd = cmul(d, wq);
        cd xk = cd{0.5 * sm.x + wd.x, 0.5 * sm.y + wd.y};
        cd xm = cd{0.5 * sm.x - wd.x, wd.y - 0.5 * sm.y};
        if (k == 0) { xk.y = 0.0; xm.y = 0.0; }              // dsc_fft.h:221-225 stores exact zeros
        st(o + k, xk);
        st(o + kL - k, xm);
    }
}

// C' (inverse pre-pass, dsc_fft.h:199-228), one thread per pair (k, L-k), k = 0..L/2:
//   Z[k] = s/2 + wq d,  Z[L-k] = conj(s/2 - wq d),  wq = (i/2) conj(W_{2L}^k)
__global__ void c2r_pre_f64_kernel(const cd *__restrict__ in, cd *__restrict__ work, const cd *__restrict__ tw_real) {
    const long long row = blockIdx.y;
    const cd *y = in + row * (kL + 1LL);
    cd *z = work + row * kL;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= kL / 2; k += gridDim.x * blockDim.x) {
        cd a = ld(y + k), b = ld(y + kL - k);
        if (k == 0) { a.y = 0.0; b.y = 0.0; }                 // dsc_fft.h:227-228 reads the real parts only
        const cd w = ld(tw_real + k);
        const cd wq = cd{0.5 * w.y, 0.5 * w.x};
        const cd sm = cd{a.x + b.x, a.y - b.y}, d = cd{a.x - b.x, a.y + b.y};
        const cd wd = cmul(d, wq);
        st(z + zidx(k), cd{0.5 * sm.x + wd.x, 0.5 * sm.y + wd.y});
        if (k != 0) st(z + zidx(kL - k), cd{0.5 * sm.x - wd.x, wd.y - 0.5 * sm.y});
    }
}

inline dim3 flat_grid(long long total) {
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return dim3((unsigned) (blocks < 1 ? 1 : blocks));
}

}  // namespace

size_t dsc_r2c256k_table_bytes() { return (size_t) kAuxEntries * 16; }

void dsc_r2c256k_build_tables(void *host_dst) {
    double *o = (double *) host_dst;
    auto put = [&](int at, long long k, long long n) {
        // quarter-turn reduction keeps the values on the axes exact
        k %= n;
        const long long q = (4 * k) / n, r = 4 * k - q * n;
        const long double a = 1.57079632679489661923132169163975144L * (long double) r / (long double) n;
        const long double cr = r == 0 ? 1.0L : cosl(a), sr = r == 0 ? 0.0L : sinl(a);
        long double c, s;
        switch (q) {
            case 0:  c = cr;  s = -sr; break;
            case 1:  c = -sr; s = -cr; break;
            case 2:  c = -cr; s = sr;  break;
            default: c = sr;  s = cr;  break;
        }
        o[2 * at] = (double) c;
        o[2 * at + 1] = (double) s;
    };
    for (int m = 0; m < 1024; ++m) put(kAuxW1024 + m, m, 1024);
    for (int m = 0; m < 512; ++m) put(kAuxW16384 + m, m, 16384);
    for (int m = 0; m < 16384; ++m) put(kAuxWL + m, m, kL);
}

// x: [rows][262144] f64 -> X: [rows][131073] c64.  work: rows * 2 MiB of scratch.
void dsc_launch_rfft256k_f64(const double *x, void *X, long long rows, void *work, const void *aux, const void *tw_real,
                             int n_cu, hipStream_t stream) {
    if (rows <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) fft16k_f64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesB);
        (void) hipFuncSetAttribute((const void *) fft16k_f64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesB);
        attr_set = true;
    }
    hipLaunchKernelGGL(radix8_in_kernel, flat_grid(rows * kSeg), dim3(256), 0, stream, (const cd *) x, (cd *) work, rows, (const cd *) aux);
    const long long n_seg = rows * 8;
    const int grid = (int) (n_seg < n_cu ? n_seg : n_cu);
    hipLaunchKernelGGL(fft16k_f64_kernel<false>, dim3(grid), dim3(kThreadsB), kLdsBytesB, stream, (cd *) work, n_seg, (const cd *) aux);
    hipLaunchKernelGGL(r2c_post_f64_kernel, dim3(kL / 2 / 256 / 4, (unsigned) rows), dim3(256), 0, stream, (const cd *) work, (cd *) X,
                       (const cd *) tw_real);
}

// X: [rows][131073] c64 -> x: [rows][262144] f64
void dsc_launch_irfft256k_f64(const void *X, double *x, long long rows, void *work, const void *aux, const void *tw_real,
                              int n_cu, hipStream_t stream) {
    if (rows <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) fft16k_f64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesB);
        (void) hipFuncSetAttribute((const void *) fft16k_f64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesB);
        attr_set = true;
    }
    hipLaunchKernelGGL(c2r_pre_f64_kernel, dim3(kL / 2 / 256 / 4, (unsigned) rows), dim3(256), 0, stream, (const cd *) X, (cd *) work,
                       (const cd *) tw_real);
    const long long n_seg = rows * 8;
    const int grid = (int) (n_seg < n_cu ? n_seg : n_cu);
    hipLaunchKernelGGL(fft16k_f64_kernel<true>, dim3(grid), dim3(kThreadsB), kLdsBytesB, stream, (cd *) work, n_seg, (const cd *) aux);
    hipLaunchKernelGGL(radix8_out_kernel, flat_grid(rows * kSeg), dim3(256), 0, stream, (const cd *) work, (cd *) x, rows,
                       (const cd *) aux, 1.0 / (double) kL);          // 2/(2n), dsc_fft.h:232
}
