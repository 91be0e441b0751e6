// fft_generic.hip — batched 1-D FFT over strided lines, any power-of-two length, f32 / f64.
//
// This is the general path behind dsc_fft / dsc_ifft / dsc_rfft / dsc_irfft
// (reference drivers: dsc/src/dsc.cpp:1958-2007, 2102-2171; kernels dsc_fft.h:57-103,
// 156-238).  The reference gathers one line at a time through dsc_axis_iterator, runs a
// recursive radix-2 DIT on the host and scatters the result; here a workgroup owns a tile
// of lines resident in LDS:
//
//   gather (coalesced along whichever of {element, line} has unit stride, zero-pad / crop,
//   real->complex cast or pairing of reals)  ->  Stockham autosort radix-4 (+ one radix-2)
//   stages ping-ponging between two LDS images  ->  fused real post-pass / pre-pass
//   (dsc_fft.h:199-228 rewritten as one formula valid for every bin)  ->  scatter.
//
// One HBM round trip for L <= dsc_fft_lds_max_len().  Longer transforms are composed by the
// host (fft_driver.cpp) from two such passes (four-step: columns + twiddle, rows) plus the
// pack / pre-pass / post-pass helpers at the bottom of this file.
//
// The 65536-point f32 real transforms do not come here: fft_r2c_64k.hip keeps the whole
// transform in registers.
#include "kernels.h"

#include <hip/hip_runtime.h>

#ifndef DSC_GEN_SKIP
#define DSC_GEN_SKIP 0          // diagnostic builds (tools/probe_generic.hip): 1 skip stages, 2 skip gather loads, 4 skip scatter stores
#endif

namespace {

constexpr int kMaxThreads = 1024;           // block size is chosen per launch: 64 .. 1024 threads
constexpr int kTileBytes = 72 * 1024;        // per LDS image; two images per workgroup (144 of 160 KiB)

template<typename T> struct alignas(2 * sizeof(T)) cx { T x, y; };

template<typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template<typename T> __device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template<typename T> __device__ __forceinline__ cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }

// exp(-2 pi i m / len), computed in double (four-step twiddles; not a hot path)
template<typename T> __device__ __forceinline__ cx<T> unit_root(long long m, long long len) {
    double s, c;
    sincospi(-2.0 * (double) m / (double) len, &s, &c);
    return {(T) c, (T) s};
}

__device__ __forceinline__ long long line_base(long long q, long long inner, const dsc_line_layout &l) {
    return (q / inner) * l.outer_stride + (q % inner) * l.inner_stride;
}

// Packed-real relations (dsc_fft.h:199-228), one bin at a time:
//   forward:  X[k] = h1 + w * h2,        h1 = (a + conj b)/2,  h2 = -i (a - conj b)/2,  w = W_{2L}^k
//   inverse:  Z[k] = h1 + conj(w) * h2,  h1 = (a + conj b)/2,  h2 = +i (a - conj b)/2
// with a = in[k], b = in[L-k].  Bin 0, L/2 and L need no special case in the forward
// direction because w is exactly 1, -i, -1 there.
template<typename T> __device__ __forceinline__ cx<T> r2c_bin(cx<T> a, cx<T> b, cx<T> w) {
    const T h1r = (T) 0.5 * (a.x + b.x), h1i = (T) 0.5 * (a.y - b.y);
    const T h2r = (T) 0.5 * (a.y + b.y), h2i = (T) -0.5 * (a.x - b.x);
    return {h1r + w.x * h2r - w.y * h2i, h1i + w.x * h2i + w.y * h2r};
}
template<typename T> __device__ __forceinline__ cx<T> c2r_bin(cx<T> a, cx<T> b, cx<T> w) {
    const T h1r = (T) 0.5 * (a.x + b.x), h1i = (T) 0.5 * (a.y - b.y);
    const T h2r = (T) -0.5 * (a.y + b.y), h2i = (T) 0.5 * (a.x - b.x);
    const T wr = w.x, wi = -w.y;
    return {h1r + wr * h2r - wi * h2i, h1i + wr * h2i + wi * h2r};
}

template<typename T>
struct lines_params {
    const void *in;
    void *out;
    long long n_lines, inner;
    dsc_line_layout lin, lout;
    int L, in_len, inverse;
    int C, P;                 // lines per tile, LDS row pitch (complex elements)
    T scale;
    const cx<T> *tw, *tw_real, *tw4;
    long long tw4_len;
};

// Division-free walk over the (line, element) pairs of a tile.  The block (a power of two) is
// split TC x TE with both factors powers of two; neighbouring threads take neighbouring
// ELEMENTS when the element stride is 1 in memory, neighbouring LINES otherwise.
struct tile_walk {
    int tc, te, TC, TE;
};
__device__ __forceinline__ int pow2_ceil(int x) { return x <= 1 ? 1 : 1 << (32 - __clz(x - 1)); }
__device__ __forceinline__ tile_walk make_walk(int tid, int threads, int nl, int per_line, bool elem_major) {
    tile_walk w;
    if (elem_major) {
        const int te_n = pow2_ceil(per_line);
        w.TE = te_n < threads ? te_n : threads;
        w.TC = threads / w.TE;
        w.te = tid & (w.TE - 1);
        w.tc = tid >> (31 - __clz(w.TE));
    } else {
        const int tc_n = pow2_ceil(nl);
        w.TC = tc_n < threads ? tc_n : threads;
        w.TE = threads / w.TC;
        w.tc = tid & (w.TC - 1);
        w.te = tid >> (31 - __clz(w.TC));
    }
    return w;
}

template<typename T, int MODE>
__global__ __launch_bounds__(kMaxThreads) void fft_lines_kernel(const lines_params<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = cx<T>;
    const int tid = threadIdx.x;
    const int threads = blockDim.x;
    const int L = p.L, P = p.P;
    const int log2L = 31 - __clz(L);
    const long long q0 = (long long) blockIdx.x * p.C;
    const int nl = (int) (p.n_lines - q0 < p.C ? p.n_lines - q0 : p.C);

    C *src = (C *) smem;
    C *dst = src + (size_t) p.C * P;
    // per-line bases, computed once per tile (64-bit divisions are expensive)
    long long *base_in = (long long *) (dst + (size_t) p.C * P);
    long long *base_out = base_in + p.C;
    int *line_in_group = (int *) (base_out + p.C);
    for (int c = tid; c < nl; c += threads) {
        const long long q = q0 + c;
        const long long o = q / p.inner, i = q - o * p.inner;
        base_in[c] = o * p.lin.outer_stride + i * p.lin.inner_stride;
        base_out[c] = o * p.lout.outer_stride + i * p.lout.inner_stride;
        line_in_group[c] = (int) i;
    }
    __syncthreads();

    // ---------------------------------------------------------------- gather
    {
        const int per_line = (MODE == DSC_MODE_C2R_PACKED) ? L + 1 : L;
        const tile_walk w = make_walk(tid, threads, nl, per_line, p.lin.elem_stride == 1 || nl == 1);
        for (int c = w.tc; c < nl; c += w.TC) {
            const long long base = base_in[c];
            for (int e = w.te; e < per_line; e += w.TE) {
                C v = {(T) 0, (T) 0};
                if (MODE == DSC_MODE_C2C || MODE == DSC_MODE_C2R_PACKED) {
                    if (e < p.in_len) v = ((const C *) p.in)[base + (long long) e * p.lin.elem_stride];
                } else if (MODE == DSC_MODE_R2C_CAST) {
                    if (e < p.in_len) v.x = ((const T *) p.in)[base + (long long) e * p.lin.elem_stride];
                } else {    // R2C_PACKED: complex sample e = (x[2e], x[2e+1])
                    const T *x = (const T *) p.in;
                    const int r = 2 * e;
                    if (p.lin.elem_stride == 1 && ((base & 1) == 0) && r + 1 < p.in_len) {
                        v = *(const C *) (x + base + r);
                    } else {
                        if (r < p.in_len)     v.x = x[base + (long long) r * p.lin.elem_stride];
                        if (r + 1 < p.in_len) v.y = x[base + (long long) (r + 1) * p.lin.elem_stride];
                    }
                }
                src[c * P + e] = v;
            }
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- irfft pre-pass
    if (MODE == DSC_MODE_C2R_PACKED) {
        const int total = nl << log2L;
        for (int idx = tid; idx < total; idx += threads) {
            const int c = idx >> log2L, k = idx & (L - 1);
            C a = src[c * P + k], b = src[c * P + L - k];
            if (k == 0) { a.y = (T) 0; b.y = (T) 0; }     // dsc_fft.h:227-228 reads real parts only
            dst[c * P + k] = c2r_bin(a, b, p.tw_real[k]);
        }
        __syncthreads();
        C *t = src; src = dst; dst = t;
    }

    if (!(DSC_GEN_SKIP & 1))
    // ---------------------------------------------------------------- Stockham stages
    // Stage with Ns points already combined per sub-transform and radix R:
    //   v[r]  = src[j + r L/R] * W_{Ns R}^{r k},   k = j mod Ns
    //   dst[(j - k) R + k + r Ns] = DFT_R(v)[r]
    // (natural order in, natural order out; no bit reversal).
    {
        int Ns = 1;
        if (log2L & 1) {                                  // one radix-2 stage first (Ns = 1: no twiddle)
            const int half = L >> 1;
            const int items = nl << (log2L - 1);
            for (int w = tid; w < items; w += threads) {
                const int c = w >> (log2L - 1), j = w & (half - 1);
                const C a = src[c * P + j], b = src[c * P + j + half];
                dst[c * P + 2 * j]     = cadd(a, b);
                dst[c * P + 2 * j + 1] = csub(a, b);
            }
            __syncthreads();
            C *t = src; src = dst; dst = t;
            Ns = 2;
        }
        const T sgn = p.inverse ? (T) -1 : (T) 1;
        for (; Ns < L; Ns <<= 2) {
            const int quarter = L >> 2;
            const int items = nl << (log2L - 2);
            const int tw_step = L / (Ns << 2);
            for (int w = tid; w < items; w += threads) {
                const int c = w >> (log2L - 2), j = w & (quarter - 1);
                const int k = j & (Ns - 1);
                const C *s = src + c * P + j;
                C v0 = s[0], v1 = s[quarter], v2 = s[2 * quarter], v3 = s[3 * quarter];
                if (Ns > 1) {
                    C w1 = p.tw[k * tw_step], w2 = p.tw[2 * k * tw_step], w3 = p.tw[3 * k * tw_step];
                    w1.y *= sgn; w2.y *= sgn; w3.y *= sgn;
                    v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
                }
                const C t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3);
                const C d = csub(v1, v3);
                const C t3 = {sgn * d.y, -sgn * d.x};        // (v1 - v3) * (-i) forward, * (+i) inverse
                C *o = dst + c * P + ((j - k) << 2) + k;
                o[0]      = cadd(t0, t2);
                o[Ns]     = cadd(t1, t3);
                o[2 * Ns] = csub(t0, t2);
                o[3 * Ns] = csub(t1, t3);
            }
            __syncthreads();
            C *t = src; src = dst; dst = t;
        }
    }

    // ---------------------------------------------------------------- scatter
    if (MODE == DSC_MODE_C2C || MODE == DSC_MODE_R2C_CAST) {
        const tile_walk w = make_walk(tid, threads, nl, L, p.lout.elem_stride == 1 || nl == 1);
        for (int c = w.tc; c < nl; c += w.TC) {
            const long long base = base_out[c];
            const long long grp = line_in_group[c];
            for (int k = w.te; k < L; k += w.TE) {
                C v = src[c * P + k];
                if (p.tw4_len) {
                    C tw = p.tw4 ? p.tw4[grp * k] : unit_root<T>(grp * k, p.tw4_len);
                    if (p.inverse) tw.y = -tw.y;
                    v = cmul(v, tw);
                }
                v.x *= p.scale; v.y *= p.scale;
                ((C *) p.out)[base + (long long) k * p.lout.elem_stride] = v;
            }
        }
    } else if (MODE == DSC_MODE_R2C_PACKED) {
        const tile_walk w = make_walk(tid, threads, nl, L + 1, p.lout.elem_stride == 1 || nl == 1);
        for (int c = w.tc; c < nl; c += w.TC) {
            const long long base = base_out[c];
            for (int k = w.te; k <= L; k += w.TE) {
                const C a = src[c * P + (k == L ? 0 : k)];
                const C b = src[c * P + (k == 0 ? 0 : L - k)];
                C v = r2c_bin(a, b, p.tw_real[k]);
                if (k == 0 || k == L) v.y = (T) 0;           // dsc_fft.h:221-225 stores exact zeros
                ((C *) p.out)[base + (long long) k * p.lout.elem_stride] = v;
            }
        }
    } else {    // C2R_PACKED: 2L reals = the L complex samples, scaled (dsc_fft.h:232-236)
        const tile_walk w = make_walk(tid, threads, nl, L, p.lout.elem_stride == 1 || nl == 1);
        T *out = (T *) p.out;
        for (int c = w.tc; c < nl; c += w.TC) {
            const long long base = base_out[c];
            for (int k = w.te; k < L; k += w.TE) {
                C v = src[c * P + k];
                v.x *= p.scale; v.y *= p.scale;
                if (p.lout.elem_stride == 1 && ((base & 1) == 0)) {
                    *(C *) (out + base + 2 * k) = v;
                } else {
                    out[base + (long long) (2 * k) * p.lout.elem_stride]     = v.x;
                    out[base + (long long) (2 * k + 1) * p.lout.elem_stride] = v.y;
                }
            }
        }
    }
}

template<typename T>
void launch_lines(const dsc_fft_lines_args &a, dsc_fft_mode mode, hipStream_t stream) {
    lines_params<T> p;
    p.in = a.in; p.out = a.out;
    p.n_lines = a.n_lines; p.inner = a.inner;
    p.lin = a.lin; p.lout = a.lout;
    p.L = a.L; p.in_len = a.in_len; p.inverse = a.inverse;
    p.scale = (T) a.scale;
    p.tw = (const cx<T> *) a.tw; p.tw_real = (const cx<T> *) a.tw_real;
    p.tw4_len = a.tw4_len;
    p.tw4 = (const cx<T> *) a.tw4;
    p.P = a.L + 2;                                             // room for bin L; keeps rows 16-B aligned
    // Tile = C lines.  Small tiles (<= 16 KiB per LDS image) so that several workgroups share
    // a CU (occupancy hides the gather/scatter latency); a line that is longer than that gets a
    // tile of its own and more threads instead.  Never fewer than ~1024 points per tile when
    // the batch has them.
    const long long image_cap = kTileBytes / (long long) sizeof(cx<T>);
    const long long cap_c = image_cap / p.P > 0 ? image_cap / p.P : 1;
    long long C = (16 * 1024 / (long long) sizeof(cx<T>)) / p.P;
    long long min_c = (1024 + a.L - 1) / a.L;
    // lines that are strided in memory (columns of a four-step, non-last axes) are coalesced ACROSS
    // the lines of a tile: take at least one 128-B cache line worth of neighbours
    const long long line_c = 128 / (long long) sizeof(cx<T>);
    if ((a.lin.elem_stride != 1 || a.lout.elem_stride != 1) && min_c < line_c) min_c = line_c;
    if (C < min_c) C = min_c;
    if (C < 1) C = 1;
    if (C > cap_c) C = cap_c;
    if (C > a.n_lines) C = a.n_lines;
    p.C = (int) C;
    const long long points = C * a.L;
    const int threads = points <= 256 ? 64 : points <= 2048 ? 256 : points <= 8192 ? 512 : 1024;
    const long long tiles = (a.n_lines + C - 1) / C;
    const size_t lds = 2 * (size_t) p.C * p.P * sizeof(cx<T>) + (size_t) p.C * 20;      // two images + per-line base tables
    dim3 grid((unsigned) tiles), block(threads);
    static unsigned long long attr_devices = 0;          // dynamic LDS above 64 KiB must be opted into, once per kernel
    if (dsc_first_use_on_device(attr_devices)) {
        const int max_lds = 2 * kTileBytes + 8192;
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_lines_kernel<T, DSC_MODE_C2C>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_lines_kernel<T, DSC_MODE_R2C_CAST>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_lines_kernel<T, DSC_MODE_R2C_PACKED>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_lines_kernel<T, DSC_MODE_C2R_PACKED>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    }
    switch (mode) {
        case DSC_MODE_C2C:        DSC_LAUNCH((fft_lines_kernel<T, DSC_MODE_C2C>), grid, block, lds, stream, p); break;
        case DSC_MODE_R2C_CAST:   DSC_LAUNCH((fft_lines_kernel<T, DSC_MODE_R2C_CAST>), grid, block, lds, stream, p); break;
        case DSC_MODE_R2C_PACKED: DSC_LAUNCH((fft_lines_kernel<T, DSC_MODE_R2C_PACKED>), grid, block, lds, stream, p); break;
        case DSC_MODE_C2R_PACKED: DSC_LAUNCH((fft_lines_kernel<T, DSC_MODE_C2R_PACKED>), grid, block, lds, stream, p); break;
    }
}

// ------------------------------------------------------------------------------------------
// helpers of the multi-pass path: one thread per complex element of the [n_lines][L] work area

template<typename T, int MODE>
__global__ void fft_pack_kernel(const void *in, cx<T> *work, long long q_first, long long n_lines, long long inner,
                                dsc_line_layout lin, int L, int in_len) {
    const long long total = n_lines * L;
    for (long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long) gridDim.x * blockDim.x) {
        const long long q = idx / L;
        const int e = (int) (idx - q * L);
        const long long base = line_base(q_first + q, inner, lin);
        cx<T> v = {(T) 0, (T) 0};
        if (MODE == DSC_MODE_C2C) {
            if (e < in_len) v = ((const cx<T> *) in)[base + (long long) e * lin.elem_stride];
        } else if (MODE == DSC_MODE_R2C_CAST) {
            if (e < in_len) v.x = ((const T *) in)[base + (long long) e * lin.elem_stride];
        } else {
            const T *x = (const T *) in;
            if (2 * e < in_len)     v.x = x[base + (long long) (2 * e) * lin.elem_stride];
            if (2 * e + 1 < in_len) v.y = x[base + (long long) (2 * e + 1) * lin.elem_stride];
        }
        work[idx] = v;
    }
}

template<typename T>
__global__ void fft_c2r_prepass_kernel(const cx<T> *in, cx<T> *work, long long q_first, long long n_lines, long long inner,
                                       dsc_line_layout lin, int L, int in_len, const cx<T> *tw_real) {
    const long long total = n_lines * L;
    for (long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long) gridDim.x * blockDim.x) {
        const long long q = idx / L;
        const int k = (int) (idx - q * L);
        const long long base = line_base(q_first + q, inner, lin);
        cx<T> a = {(T) 0, (T) 0}, b = {(T) 0, (T) 0};
        if (k < in_len)     a = in[base + (long long) k * lin.elem_stride];
        if (L - k < in_len) b = in[base + (long long) (L - k) * lin.elem_stride];
        if (k == 0) { a.y = (T) 0; b.y = (T) 0; }
        work[idx] = c2r_bin(a, b, tw_real[k]);
    }
}

template<typename T>
__global__ void fft_r2c_postpass_kernel(const cx<T> *work, cx<T> *out, long long q_first, long long n_lines, long long inner,
                                        dsc_line_layout lout, int L, const cx<T> *tw_real) {
    const long long bins = (long long) L + 1;
    const long long total = n_lines * bins;
    for (long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long) gridDim.x * blockDim.x) {
        const long long q = idx / bins;
        const int k = (int) (idx - q * bins);
        const cx<T> *z = work + q * L;
        const cx<T> a = z[k == L ? 0 : k], b = z[k == 0 ? 0 : L - k];
        cx<T> v = r2c_bin(a, b, tw_real[k]);
        if (k == 0 || k == L) v.y = (T) 0;
        out[line_base(q_first + q, inner, lout) + (long long) k * lout.elem_stride] = v;
    }
}

template<typename T, int MODE>
__global__ void fft_unpack_kernel(const cx<T> *work, void *out, long long q_first, long long n_lines, long long inner,
                                  dsc_line_layout lout, int L, T scale) {
    const long long total = n_lines * L;
    for (long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long) gridDim.x * blockDim.x) {
        const long long q = idx / L;
        const int k = (int) (idx - q * L);
        cx<T> v = work[idx];
        v.x *= scale; v.y *= scale;
        const long long base = line_base(q_first + q, inner, lout);
        if (MODE == DSC_MODE_C2R_PACKED) {
            T *o = (T *) out;
            o[base + (long long) (2 * k) * lout.elem_stride]     = v.x;
            o[base + (long long) (2 * k + 1) * lout.elem_stride] = v.y;
        } else {
            ((cx<T> *) out)[base + (long long) k * lout.elem_stride] = v;
        }
    }
}

inline dim3 flat_grid(long long total) {
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned) blocks);
}

}  // namespace

int dsc_fft_lds_max_len(bool single_precision) {
    // largest power of two with (L + 2) complex <= one 72 KiB LDS image
    return single_precision ? 8192 : 4096;
}

void dsc_launch_fft_lines(const dsc_fft_lines_args &a, dsc_fft_mode mode, bool single_precision, hipStream_t stream) {
    if (a.n_lines <= 0) return;
    if (single_precision) launch_lines<float>(a, mode, stream);
    else                  launch_lines<double>(a, mode, stream);
}

void dsc_launch_fft_pack(const void *in, void *work, long long q_first, long long n_lines, long long inner, dsc_line_layout lin,
                         int L, int in_len, dsc_fft_mode mode, bool sp, hipStream_t stream) {
    const dim3 grid = flat_grid(n_lines * L), block(256);
#define PACK(T, M) DSC_LAUNCH((fft_pack_kernel<T, M>), grid, block, 0, stream, in, (cx<T> *) work, q_first, n_lines, inner, lin, L, in_len)
    if (sp) {
        if (mode == DSC_MODE_C2C) PACK(float, DSC_MODE_C2C);
        else if (mode == DSC_MODE_R2C_CAST) PACK(float, DSC_MODE_R2C_CAST);
        else PACK(float, DSC_MODE_R2C_PACKED);
    } else {
        if (mode == DSC_MODE_C2C) PACK(double, DSC_MODE_C2C);
        else if (mode == DSC_MODE_R2C_CAST) PACK(double, DSC_MODE_R2C_CAST);
        else PACK(double, DSC_MODE_R2C_PACKED);
    }
#undef PACK
}

void dsc_launch_fft_c2r_prepass(const void *in, void *work, long long q_first, long long n_lines, long long inner, dsc_line_layout lin,
                                int L, int in_len, const void *tw_real, bool sp, hipStream_t stream) {
    const dim3 grid = flat_grid(n_lines * L), block(256);
    if (sp) DSC_LAUNCH(fft_c2r_prepass_kernel<float>, grid, block, 0, stream, (const cx<float> *) in, (cx<float> *) work,
                               q_first, n_lines, inner, lin, L, in_len, (const cx<float> *) tw_real);
    else    DSC_LAUNCH(fft_c2r_prepass_kernel<double>, grid, block, 0, stream, (const cx<double> *) in, (cx<double> *) work,
                               q_first, n_lines, inner, lin, L, in_len, (const cx<double> *) tw_real);
}

void dsc_launch_fft_r2c_postpass(const void *work, void *out, long long q_first, long long n_lines, long long inner, dsc_line_layout lout,
                                 int L, const void *tw_real, bool sp, hipStream_t stream) {
    const dim3 grid = flat_grid(n_lines * (L + 1LL)), block(256);
    if (sp) DSC_LAUNCH(fft_r2c_postpass_kernel<float>, grid, block, 0, stream, (const cx<float> *) work, (cx<float> *) out,
                               q_first, n_lines, inner, lout, L, (const cx<float> *) tw_real);
    else    DSC_LAUNCH(fft_r2c_postpass_kernel<double>, grid, block, 0, stream, (const cx<double> *) work, (cx<double> *) out,
                               q_first, n_lines, inner, lout, L, (const cx<double> *) tw_real);
}

void dsc_launch_fft_unpack(const void *work, void *out, long long q_first, long long n_lines, long long inner, dsc_line_layout lout,
                           int L, double scale, dsc_fft_mode mode, bool sp, hipStream_t stream) {
    const dim3 grid = flat_grid(n_lines * L), block(256);
    if (sp) {
        if (mode == DSC_MODE_C2R_PACKED)
            DSC_LAUNCH((fft_unpack_kernel<float, DSC_MODE_C2R_PACKED>), grid, block, 0, stream, (const cx<float> *) work, out, q_first, n_lines, inner, lout, L, (float) scale);
        else
            DSC_LAUNCH((fft_unpack_kernel<float, DSC_MODE_C2C>), grid, block, 0, stream, (const cx<float> *) work, out, q_first, n_lines, inner, lout, L, (float) scale);
    } else {
        if (mode == DSC_MODE_C2R_PACKED)
            DSC_LAUNCH((fft_unpack_kernel<double, DSC_MODE_C2R_PACKED>), grid, block, 0, stream, (const cx<double> *) work, out, q_first, n_lines, inner, lout, L, scale);
        else
            DSC_LAUNCH((fft_unpack_kernel<double, DSC_MODE_C2C>), grid, block, 0, stream, (const cx<double> *) work, out, q_first, n_lines, inner, lout, L, scale);
    }
}
