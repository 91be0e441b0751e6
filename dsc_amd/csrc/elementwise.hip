// elementwise.hip — dtype cast and broadcast binary operators (HBM-bound streaming kernels).
//
// Reference: dsc_cast (dsc/src/dsc.cpp:536-597, cast_op dsc/include/dsc_ops.h:12-44) and
// binary_op (dsc/src/dsc.cpp:1186-1245) with mul_op & co. (dsc_ops.h:46-90).  The reference
// walks two dsc_broadcast_iterators per element (dsc_iter.h:67-95); here the output's flat
// index is decomposed once per element and each operand offset is a dot product with its
// broadcast strides (0 on broadcast dims), so equal-shape, row-broadcast and scalar operands
// all stream at the same rate.
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <type_traits>

namespace {

template<typename T> struct alignas(2 * sizeof(T)) cx { T x, y; };

template<typename T> struct elem;   // dtype code -> storage type
template<> struct elem<float>  { static constexpr bool cplx = false; using real = float; };
template<> struct elem<double> { static constexpr bool cplx = false; using real = double; };
template<> struct elem<cx<float>>  { static constexpr bool cplx = true; using real = float; };
template<> struct elem<cx<double>> { static constexpr bool cplx = true; using real = double; };

inline dim3 stream_grid(long long ne) {
    long long blocks = (ne + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;        // 8 blocks per CU, grid-stride beyond that
    if (blocks < 1) blocks = 1;
    return dim3((unsigned) blocks);
}

// Grid of the PACKED kernels (16 bytes per thread in their widest stream): one pack per thread, no loop in practice —
// workgroups are dispatched in address order and end right after their store, so the whole chip sweeps one window of memory:
// 6.2 TB/s for a copy against 4.7 TB/s for 2048 grid-striding workgroups (tools/membench2.hip; mul c32 of equal shapes
// 62.8 -> 81.4 % of the roofline).  With 4 or 8 bytes per thread the same launch is SLOWER than the capped grid (measured:
// abs c32 69.5 -> 66 %, cast f32 -> c32 67.7 -> 55.8 %), hence the packs.  2^23 blocks keep the 32-bit strides at <= 2^31.
inline dim3 pack_grid(long long npack) {
    long long blocks = (npack + 255) / 256;
#ifdef DSC_STREAM_GRID_CAPPED
    if (blocks > 256 * 8) blocks = 256 * 8;
#endif
    if (blocks > (1 << 23)) blocks = 1 << 23;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned) blocks);
}

template<typename T, int V> struct alignas(sizeof(T) * V) packed { T e[V]; };
template<typename A, typename B> constexpr int pack_width() { return 16 / (int) (sizeof(A) > sizeof(B) ? sizeof(A) : sizeof(B)); }
inline bool aligned_to(const void *p, size_t a) { return ((size_t) p & (a - 1)) == 0; }

// cast_op (dsc_ops.h:12-44): complex -> real keeps .real; real -> complex sets imag = 0
template<typename Tin, typename Tout>
__device__ __forceinline__ Tout cast_one(Tin v) {
    using Rout = typename elem<Tout>::real;
    if constexpr (elem<Tout>::cplx) {
        if constexpr (elem<Tin>::cplx) return Tout{(Rout) v.x, (Rout) v.y};
        else                           return Tout{(Rout) v, (Rout) 0};
    } else {
        if constexpr (elem<Tin>::cplx) return (Rout) v.x;
        else                           return (Rout) v;
    }
}

template<typename Tin, typename Tout>
__global__ void cast_kernel(const Tin *in, Tout *out, long long ne) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (long long) gridDim.x * blockDim.x)
        out[i] = cast_one<Tin, Tout>(in[i]);
}

// V consecutive elements per thread, 16 bytes in the wider of the two streams
template<typename Tin, typename Tout, int V>
__global__ void cast_pack_kernel(const Tin *in, Tout *out, unsigned npack) {
    const packed<Tin, V> *pi = (const packed<Tin, V> *) in;
    packed<Tout, V> *po = (packed<Tout, V> *) out;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npack; i += gridDim.x * blockDim.x) {
        const packed<Tin, V> x = pi[i];
        packed<Tout, V> r;
#pragma unroll
        for (int j = 0; j < V; ++j) r.e[j] = cast_one<Tin, Tout>(x.e[j]);
        po[i] = r;
    }
}

template<typename Tin, typename Tout>
void cast_to(const Tin *x, Tout *out, long long ne, hipStream_t s) {
    constexpr int V = pack_width<Tin, Tout>();
    long long done = 0;
    if (ne >= V && aligned_to(x, sizeof(Tin) * V) && aligned_to(out, sizeof(Tout) * V)) {
        const long long npack = ne / V;
        DSC_LAUNCH((cast_pack_kernel<Tin, Tout, V>), pack_grid(npack), dim3(256), 0, s, x, out, (unsigned) npack);
        done = npack * V;
    }
    if (done < ne) DSC_LAUNCH((cast_kernel<Tin, Tout>), stream_grid(ne - done), dim3(256), 0, s, x + done, out + done, ne - done);
}

template<typename Tin>
void cast_from(const void *in, void *out, int out_dtype, long long ne, dim3, hipStream_t s) {
    const Tin *x = (const Tin *) in;
    switch (out_dtype) {
        case 0: cast_to<Tin, float>(x, (float *) out, ne, s); break;
        case 1: cast_to<Tin, double>(x, (double *) out, ne, s); break;
        case 2: cast_to<Tin, cx<float>>(x, (cx<float> *) out, ne, s); break;
        default: cast_to<Tin, cx<double>>(x, (cx<double> *) out, ne, s); break;
    }
}

// add_op / sub_op / mul_op / div_op: dsc_ops.h:46-90
template<typename T, int OP>
__device__ __forceinline__ T apply(T a, T b) {
    if constexpr (elem<T>::cplx) {
        if constexpr (OP == 0) return T{a.x + b.x, a.y + b.y};
        else if constexpr (OP == 1) return T{a.x - b.x, a.y - b.y};
        else if constexpr (OP == 2) return T{(a.x * b.x) - (a.y * b.y), (a.x * b.y) + (a.y * b.x)};
        else {
            const auto den = (b.x * b.x) + (b.y * b.y);
            return T{((a.x * b.x) + (a.y * b.y)) / den, ((a.y * b.x) - (a.x * b.y)) / den};
        }
    } else {
        if constexpr (OP == 0) return a + b;
        else if constexpr (OP == 1) return a - b;
        else if constexpr (OP == 2) return a * b;
        else return a / b;
    }
}

// fast index paths: tensors have at most 2^31 - 1 elements (int ne), so 32-bit arithmetic suffices
template<typename T, int OP, int FAST>
__global__ void binary_fast_kernel(const T *a, const T *b, T *out, unsigned ne, unsigned small_ne) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += gridDim.x * blockDim.x) {
        const unsigned ia = FAST == 3 ? i % small_ne : i;
        const unsigned ib = FAST == 2 ? i % small_ne : i;
        out[i] = apply<T, OP>(a[ia], b[ib]);
    }
}

template<typename T, int OP>
__global__ void binary_kernel(const T *a, const T *b, T *out, const dsc_bcast_args g) {
    const long long s3 = g.out_shape[3];
    const long long s23 = s3 * g.out_shape[2];
    const long long s123 = s23 * g.out_shape[1];
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < g.ne; i += (long long) gridDim.x * blockDim.x) {
        long long ia, ib;
        if (g.a_scalar)      { ia = 0; ib = i; }
        else if (g.b_scalar) { ia = i; ib = 0; }
        else {
            const long long i0 = i / s123, r0 = i - i0 * s123;
            const long long i1 = r0 / s23, r1 = r0 - i1 * s23;
            const long long i2 = r1 / s3, i3 = r1 - i2 * s3;
            ia = i0 * g.a_stride[0] + i1 * g.a_stride[1] + i2 * g.a_stride[2] + i3 * g.a_stride[3];
            ib = i0 * g.b_stride[0] + i1 * g.b_stride[1] + i2 * g.b_stride[2] + i3 * g.b_stride[3];
        }
        out[i] = apply<T, OP>(a[ia], b[ib]);
    }
}

// general broadcast with a long innermost axis: a block owns a piece of one innermost row of the RESULT — one division chain per
// block instead of one per element; along the row each operand advances by its own stride (1, or 0 where it is broadcast)
template<typename T, int OP>
__global__ void binary_rows_kernel(const T *a, const T *b, T *out, const dsc_bcast_args g, unsigned chunks_per_row) {
    const unsigned long long blk = blockIdx.x;
    const unsigned long long row = blk / chunks_per_row;
    const unsigned chunk = (unsigned) (blk - row * chunks_per_row);
    const unsigned long long i01 = row / (unsigned) g.out_shape[2];
    const long long i2 = (long long) (row - i01 * (unsigned) g.out_shape[2]);
    const long long i0 = (long long) (i01 / (unsigned) g.out_shape[1]), i1 = (long long) (i01 - (unsigned long long) i0 * (unsigned) g.out_shape[1]);
    const T *pa = a + (i0 * g.a_stride[0] + i1 * g.a_stride[1] + i2 * g.a_stride[2]);
    const T *pb = b + (i0 * g.b_stride[0] + i1 * g.b_stride[1] + i2 * g.b_stride[2]);
    T *po = out + (long long) row * g.out_shape[3];
    const long long sa = g.a_stride[3], sb = g.b_stride[3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = chunk * 1024 + u * 256 + threadIdx.x;
        if (c >= g.out_shape[3]) return;
        po[c] = apply<T, OP>(pa[c * sa], pb[c * sb]);
    }
}

// equal shapes: 16 bytes per lane and operand (V = 16 / sizeof(T) elements), the widest global access
template<typename T, int OP>
__global__ void binary_same_vec_kernel(const T *a, const T *b, T *out, unsigned nvec) {
    constexpr int V = 16 / sizeof(T);
    struct alignas(16) pack { T e[V]; };
    const pack *pa = (const pack *) a, *pb = (const pack *) b;
    pack *po = (pack *) out;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        const u4v xr = __builtin_nontemporal_load((const u4v *) (pa + i)), yr = __builtin_nontemporal_load((const u4v *) (pb + i));
        const pack x = __builtin_bit_cast(pack, xr), y = __builtin_bit_cast(pack, yr);      // touched once: streaming policy
        pack r;
#pragma unroll
        for (int j = 0; j < V; ++j) r.e[j] = apply<T, OP>(x.e[j], y.e[j]);
        __builtin_nontemporal_store(__builtin_bit_cast(u4v, r), (u4v *) (po + i));
    }
}

// One operand has the output's shape, the other is a scalar or spans the trailing dims (a [B, K] spectrum times a [K] filter):
// 16 bytes per thread of the large operand and of the result; the small operand is read per element (it lives in the caches),
// its index from ONE modulo per thread.  BIG_IS_A: out = big op small, else out = small op big.
template<typename T, int OP, bool BIG_IS_A>
__global__ void binary_small_pack_kernel(const T *big, const T *small, T *out, unsigned npack, unsigned small_ne) {
    constexpr int V = 16 / sizeof(T);
    const packed<T, V> *pbig = (const packed<T, V> *) big;
    packed<T, V> *po = (packed<T, V> *) out;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npack; i += gridDim.x * blockDim.x) {
        const packed<T, V> x = pbig[i];
        unsigned m = small_ne == 1 ? 0u : (unsigned) (((unsigned long long) i * V) % small_ne);
        packed<T, V> r;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const T sv = small[m];
            r.e[j] = BIG_IS_A ? apply<T, OP>(x.e[j], sv) : apply<T, OP>(sv, x.e[j]);
            m = m + 1 >= small_ne ? 0u : m + 1;
        }
        po[i] = r;
    }
}

// The small operand spans the LEADING dims (a [B, K] tensor against a [B, 1] column): element e of the result takes small[e / per];
// one division per thread, then the index steps when a pack crosses a row end.
template<typename T, int OP, bool BIG_IS_A>
__global__ void binary_column_pack_kernel(const T *big, const T *small, T *out, unsigned npack, unsigned per) {
    constexpr int V = 16 / sizeof(T);
    const packed<T, V> *pbig = (const packed<T, V> *) big;
    packed<T, V> *po = (packed<T, V> *) out;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npack; i += gridDim.x * blockDim.x) {
        const packed<T, V> x = pbig[i];
        const unsigned long long e0 = (unsigned long long) i * V;
        unsigned q = (unsigned) (e0 / per), m = (unsigned) (e0 - (unsigned long long) q * per);
        packed<T, V> r;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const T sv = small[q];
            r.e[j] = BIG_IS_A ? apply<T, OP>(x.e[j], sv) : apply<T, OP>(sv, x.e[j]);
            if (++m >= per) { m = 0; ++q; }
        }
        po[i] = r;
    }
}

template<typename T, int OP>
bool binary_fast(const T *pa, const T *pb, T *po, const dsc_bcast_args &g, dim3 grid, hipStream_t s) {
    constexpr unsigned V = 16 / sizeof(T);
    if (g.fast >= 4 && !g.a_scalar && !g.b_scalar) {
        const bool big_is_a = g.fast == 4;
        const T *big = big_is_a ? pa : pb, *small = big_is_a ? pb : pa;
        if (g.ne % V != 0 || !aligned_to(big, 16) || !aligned_to(po, 16) || g.small_ne < (int) V) return false;
        const unsigned npack = (unsigned) (g.ne / V);
        if (big_is_a) DSC_LAUNCH((binary_column_pack_kernel<T, OP, true>), pack_grid(npack), dim3(256), 0, s, big, small, po, npack, (unsigned) g.small_ne);
        else          DSC_LAUNCH((binary_column_pack_kernel<T, OP, false>), pack_grid(npack), dim3(256), 0, s, big, small, po, npack, (unsigned) g.small_ne);
        return true;
    }
    if ((g.a_scalar || g.b_scalar || g.fast == 2 || g.fast == 3) && g.ne % V == 0) {
        const bool big_is_a = g.b_scalar || (!g.a_scalar && g.fast == 2);
        const T *big = big_is_a ? pa : pb, *small = big_is_a ? pb : pa;
        const unsigned sm = (g.a_scalar || g.b_scalar) ? 1u : (unsigned) g.small_ne;
        if (aligned_to(big, 16) && aligned_to(po, 16) && sm >= 1) {
            const unsigned npack = (unsigned) (g.ne / V);
            if (big_is_a) DSC_LAUNCH((binary_small_pack_kernel<T, OP, true>), pack_grid(npack), dim3(256), 0, s, big, small, po, npack, sm);
            else          DSC_LAUNCH((binary_small_pack_kernel<T, OP, false>), pack_grid(npack), dim3(256), 0, s, big, small, po, npack, sm);
            return true;
        }
    }
    if (g.a_scalar || g.b_scalar || g.fast == 0 || g.fast >= 4) return false;
    const unsigned ne = (unsigned) g.ne, sm = (unsigned) g.small_ne;
    if (g.fast == 1 && V > 1 && ne % V == 0 && (((size_t) pa | (size_t) pb | (size_t) po) & 15) == 0) {
        DSC_LAUNCH((binary_same_vec_kernel<T, OP>), pack_grid(ne / V), dim3(256), 0, s, pa, pb, po, ne / V);
        return true;
    }
    if (g.fast == 1) DSC_LAUNCH((binary_fast_kernel<T, OP, 1>), grid, dim3(256), 0, s, pa, pb, po, ne, sm);
    else if (g.fast == 2) DSC_LAUNCH((binary_fast_kernel<T, OP, 2>), grid, dim3(256), 0, s, pa, pb, po, ne, sm);
    else DSC_LAUNCH((binary_fast_kernel<T, OP, 3>), grid, dim3(256), 0, s, pa, pb, po, ne, sm);
    return true;
}

// General broadcast, 16 bytes of the result per thread: along the innermost axis an operand either advances with the result
// (one 16-byte load) or is broadcast (one element); the outer indices come from one division chain per thread.  A block is no
// longer tied to one innermost row, so short rows (512 floats) fill their workgroups too.
template<typename T, int OP>
__global__ void binary_bcast_pack_kernel(const T *a, const T *b, T *out, const dsc_bcast_args g, unsigned packs_per_row, unsigned npack) {
    constexpr int V = 16 / sizeof(T);
    packed<T, V> *po = (packed<T, V> *) out;
    const unsigned s2 = (unsigned) g.out_shape[2], s1 = (unsigned) g.out_shape[1];
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < npack; p += gridDim.x * blockDim.x) {
        const unsigned row = p / packs_per_row, c = (p - row * packs_per_row) * V;
        const unsigned t = row / s2, i2 = row - t * s2;
        const unsigned i0 = t / s1, i1 = t - i0 * s1;
        const T *ra = a + ((long long) i0 * g.a_stride[0] + (long long) i1 * g.a_stride[1] + (long long) i2 * g.a_stride[2]);
        const T *rb = b + ((long long) i0 * g.b_stride[0] + (long long) i1 * g.b_stride[1] + (long long) i2 * g.b_stride[2]);
        packed<T, V> x, y, r;
        if (g.a_stride[3]) x = *(const packed<T, V> *) (ra + c);
        else {
            const T v = ra[0];
#pragma unroll
            for (int j = 0; j < V; ++j) x.e[j] = v;
        }
        if (g.b_stride[3]) y = *(const packed<T, V> *) (rb + c);
        else {
            const T v = rb[0];
#pragma unroll
            for (int j = 0; j < V; ++j) y.e[j] = v;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) r.e[j] = apply<T, OP>(x.e[j], y.e[j]);
        po[p] = r;
    }
}

template<typename T>
bool binary_bcast_pack(const T *pa, const T *pb, T *po, int op, const dsc_bcast_args &g, hipStream_t s) {
    constexpr int V = 16 / sizeof(T);
    if (g.a_scalar || g.b_scalar || g.out_shape[3] % V != 0 || !aligned_to(po, 16) || g.ne / V >= (1LL << 32)) return false;
    auto fits = [&](const T *p, const int *st) {
        if (st[3] == 0) return true;
        if (st[3] != 1 || !aligned_to(p, 16)) return false;
        for (int k = 0; k < 3; ++k) if (st[k] % V != 0) return false;
        return true;
    };
    if (!fits(pa, g.a_stride) || !fits(pb, g.b_stride)) return false;
    const unsigned ppr = (unsigned) (g.out_shape[3] / V), npack = (unsigned) (g.ne / V);
    switch (op) {
        case 0: DSC_LAUNCH((binary_bcast_pack_kernel<T, 0>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, g, ppr, npack); break;
        case 1: DSC_LAUNCH((binary_bcast_pack_kernel<T, 1>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, g, ppr, npack); break;
        case 2: DSC_LAUNCH((binary_bcast_pack_kernel<T, 2>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, g, ppr, npack); break;
        default: DSC_LAUNCH((binary_bcast_pack_kernel<T, 3>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, g, ppr, npack); break;
    }
    return true;
}

template<typename T>
void binary_typed(const void *a, const void *b, void *out, int op, const dsc_bcast_args &g, dim3 grid, hipStream_t s) {
    const T *pa = (const T *) a, *pb = (const T *) b;
    T *po = (T *) out;
    if (op == 0 && binary_fast<T, 0>(pa, pb, po, g, grid, s)) return;
    if (op == 1 && binary_fast<T, 1>(pa, pb, po, g, grid, s)) return;
    if (op == 2 && binary_fast<T, 2>(pa, pb, po, g, grid, s)) return;
    if (op == 3 && binary_fast<T, 3>(pa, pb, po, g, grid, s)) return;
    if (binary_bcast_pack<T>(pa, pb, po, op, g, s)) return;
    const long long rows = g.out_shape[3] > 0 ? g.ne / g.out_shape[3] : 0;
    const unsigned chunks = (unsigned) ((g.out_shape[3] + 1023) / 1024);
    if (!g.a_scalar && !g.b_scalar && g.out_shape[3] >= 64 && rows * chunks < (1LL << 31)) {
        const dim3 rg((unsigned) (rows * chunks));
        switch (op) {
            case 0: DSC_LAUNCH((binary_rows_kernel<T, 0>), rg, dim3(256), 0, s, pa, pb, po, g, chunks); break;
            case 1: DSC_LAUNCH((binary_rows_kernel<T, 1>), rg, dim3(256), 0, s, pa, pb, po, g, chunks); break;
            case 2: DSC_LAUNCH((binary_rows_kernel<T, 2>), rg, dim3(256), 0, s, pa, pb, po, g, chunks); break;
            default: DSC_LAUNCH((binary_rows_kernel<T, 3>), rg, dim3(256), 0, s, pa, pb, po, g, chunks); break;
        }
        return;
    }
    switch (op) {
        case 0: DSC_LAUNCH((binary_kernel<T, 0>), grid, dim3(256), 0, s, pa, pb, po, g); break;
        case 1: DSC_LAUNCH((binary_kernel<T, 1>), grid, dim3(256), 0, s, pa, pb, po, g); break;
        case 2: DSC_LAUNCH((binary_kernel<T, 2>), grid, dim3(256), 0, s, pa, pb, po, g); break;
        default: DSC_LAUNCH((binary_kernel<T, 3>), grid, dim3(256), 0, s, pa, pb, po, g); break;
    }
}

// abs / angle / conj / real / imag: dsc/src/dsc.cpp:1480-1622, functors dsc_ops.h:242-303.
// OP: 0 abs, 1 angle, 2 conj, 3 real, 4 imag.  Tin real or complex, output real (conj: same as input).
template<typename Tin, int OP> struct unary_out { using type = typename elem<Tin>::real; };
template<typename R> struct unary_out<cx<R>, 2> { using type = cx<R>; };

template<typename Tin, int OP>
__device__ __forceinline__ typename unary_out<Tin, OP>::type unary_one(Tin v) {
    using R = typename elem<Tin>::real;
    R re, im;
    if constexpr (elem<Tin>::cplx) { re = v.x; im = v.y; }
    else                           { re = v; im = (R) 0; }
    if constexpr (OP == 0) {
        if constexpr (elem<Tin>::cplx) return sqrt((re * re) + (im * im));
        else                           return re >= 0 ? re : -re;
    } else if constexpr (OP == 1) {
        return atan2(im, re);
    } else if constexpr (OP == 2) {
        if constexpr (elem<Tin>::cplx) return Tin{re, -im};
        else                           return re;
    } else if constexpr (OP == 3) {
        return re;
    } else {
        return im;
    }
}

template<typename Tin, int OP>
__global__ void unary_kernel(const Tin *in, void *out, long long ne) {
    using Tout = typename unary_out<Tin, OP>::type;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (long long) gridDim.x * blockDim.x)
        ((Tout *) out)[i] = unary_one<Tin, OP>(in[i]);
}

template<typename Tin, int OP, int V>
__global__ void unary_pack_kernel(const Tin *in, void *out, unsigned npack) {
    using Tout = typename unary_out<Tin, OP>::type;
    const packed<Tin, V> *pi = (const packed<Tin, V> *) in;
    packed<Tout, V> *po = (packed<Tout, V> *) out;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npack; i += gridDim.x * blockDim.x) {
        const packed<Tin, V> x = pi[i];
        packed<Tout, V> r;
#pragma unroll
        for (int j = 0; j < V; ++j) r.e[j] = unary_one<Tin, OP>(x.e[j]);
        po[i] = r;
    }
}

template<typename Tin, int OP>
void unary_op(const Tin *x, void *out, long long ne, hipStream_t s) {
    using Tout = typename unary_out<Tin, OP>::type;
    constexpr int V = pack_width<Tin, Tout>();
    long long done = 0;
    if (V > 1 && ne >= V && aligned_to(x, sizeof(Tin) * V) && aligned_to(out, sizeof(Tout) * V)) {
        const long long npack = ne / V;
        DSC_LAUNCH((unary_pack_kernel<Tin, OP, V>), pack_grid(npack), dim3(256), 0, s, x, out, (unsigned) npack);
        done = npack * V;
    }
    if (done < ne) {
        if (V == 1) DSC_LAUNCH((unary_kernel<Tin, OP>), pack_grid(ne), dim3(256), 0, s, x, out, ne);       // 16 bytes per element already
        else DSC_LAUNCH((unary_kernel<Tin, OP>), stream_grid(ne - done), dim3(256), 0, s, x + done, (void *) ((Tout *) out + done), ne - done);
    }
}

template<typename Tin>
void unary_typed(const void *in, void *out, int op, long long ne, dim3, hipStream_t s) {
    const Tin *x = (const Tin *) in;
    switch (op) {
        case 0: unary_op<Tin, 0>(x, out, ne, s); break;
        case 1: unary_op<Tin, 1>(x, out, ne, s); break;
        case 2: unary_op<Tin, 2>(x, out, ne, s); break;
        case 3: unary_op<Tin, 3>(x, out, ne, s); break;
        default: unary_op<Tin, 4>(x, out, ne, s); break;
    }
}

}  // namespace

void dsc_launch_cast(const void *in, int in_dtype, void *out, int out_dtype, long long ne, hipStream_t stream) {
    if (ne <= 0) return;
    const dim3 grid = stream_grid(ne);
    switch (in_dtype) {
        case 0: cast_from<float>(in, out, out_dtype, ne, grid, stream); break;
        case 1: cast_from<double>(in, out, out_dtype, ne, grid, stream); break;
        case 2: cast_from<cx<float>>(in, out, out_dtype, ne, grid, stream); break;
        default: cast_from<cx<double>>(in, out, out_dtype, ne, grid, stream); break;
    }
}

void dsc_launch_unary(const void *in, int in_dtype, void *out, int op, long long ne, hipStream_t stream) {
    if (ne <= 0) return;
    const dim3 grid = stream_grid(ne);
    switch (in_dtype) {
        case 0: unary_typed<float>(in, out, op, ne, grid, stream); break;
        case 1: unary_typed<double>(in, out, op, ne, grid, stream); break;
        case 2: unary_typed<cx<float>>(in, out, op, ne, grid, stream); break;
        default: unary_typed<cx<double>>(in, out, op, ne, grid, stream); break;
    }
}

// ---- operands of DIFFERENT dtypes and equal shapes: the casts of binary_op (dsc.cpp:1186-1223 casts both operands to the
// promoted type first) happen in registers — same expression per element (cast_one, then apply), no temporary in HBM.
namespace {
template<typename Ta, typename Tb> struct promoted {          // dsc_dtype.h:73-78
    static constexpr bool cplx = elem<Ta>::cplx || elem<Tb>::cplx;
    static constexpr bool wide = cplx ? (sizeof(Ta) == 16 || sizeof(Tb) == 16)
                                      : (sizeof(Ta) == 8 || sizeof(Tb) == 8);
    using real = typename std::conditional<wide, double, float>::type;
    using type = typename std::conditional<cplx, cx<real>, real>::type;
};

template<typename Ta, typename Tb, int OP>
__global__ void binary_mixed_pack_kernel(const Ta *a, const Tb *b, typename promoted<Ta, Tb>::type *out, unsigned npack) {
    using To = typename promoted<Ta, Tb>::type;
    constexpr int V = 16 / sizeof(To);
    const packed<Ta, V> *pa = (const packed<Ta, V> *) a;
    const packed<Tb, V> *pb = (const packed<Tb, V> *) b;
    packed<To, V> *po = (packed<To, V> *) out;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npack; i += gridDim.x * blockDim.x) {
        const packed<Ta, V> x = pa[i];
        const packed<Tb, V> y = pb[i];
        packed<To, V> r;
#pragma unroll
        for (int j = 0; j < V; ++j) r.e[j] = apply<To, OP>(cast_one<Ta, To>(x.e[j]), cast_one<Tb, To>(y.e[j]));
        po[i] = r;
    }
}

template<typename Ta, typename Tb>
bool mixed_pair(const void *a, const void *b, void *out, int op, long long ne, hipStream_t s) {
    using To = typename promoted<Ta, Tb>::type;
    constexpr int V = 16 / sizeof(To);
    if (ne % V != 0 || !aligned_to(a, sizeof(Ta) * V) || !aligned_to(b, sizeof(Tb) * V) || !aligned_to(out, 16)) return false;
    const unsigned npack = (unsigned) (ne / V);
    const Ta *pa = (const Ta *) a; const Tb *pb = (const Tb *) b; To *po = (To *) out;
    switch (op) {
        case 0: DSC_LAUNCH((binary_mixed_pack_kernel<Ta, Tb, 0>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, npack); break;
        case 1: DSC_LAUNCH((binary_mixed_pack_kernel<Ta, Tb, 1>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, npack); break;
        case 2: DSC_LAUNCH((binary_mixed_pack_kernel<Ta, Tb, 2>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, npack); break;
        default: DSC_LAUNCH((binary_mixed_pack_kernel<Ta, Tb, 3>), pack_grid(npack), dim3(256), 0, s, pa, pb, po, npack); break;
    }
    return true;
}

template<typename Ta>
bool mixed_a(const void *a, const void *b, int b_dtype, void *out, int op, long long ne, hipStream_t s) {
    switch (b_dtype) {
        case 0: return mixed_pair<Ta, float>(a, b, out, op, ne, s);
        case 1: return mixed_pair<Ta, double>(a, b, out, op, ne, s);
        case 2: return mixed_pair<Ta, cx<float>>(a, b, out, op, ne, s);
        default: return mixed_pair<Ta, cx<double>>(a, b, out, op, ne, s);
    }
}
}  // namespace

bool dsc_launch_binary_mixed(const void *a, int a_dtype, const void *b, int b_dtype, void *out, int op, long long ne, hipStream_t stream) {
    if (ne <= 0 || a_dtype == b_dtype) return false;
    switch (a_dtype) {
        case 0: return mixed_a<float>(a, b, b_dtype, out, op, ne, stream);
        case 1: return mixed_a<double>(a, b, b_dtype, out, op, ne, stream);
        case 2: return mixed_a<cx<float>>(a, b, b_dtype, out, op, ne, stream);
        default: return mixed_a<cx<double>>(a, b, b_dtype, out, op, ne, stream);
    }
}

void dsc_launch_binary(const void *a, const void *b, void *out, int dtype, int op, const dsc_bcast_args &g, hipStream_t stream) {
    if (g.ne <= 0) return;
    const dim3 grid = stream_grid(g.ne);
    switch (dtype) {
        case 0: binary_typed<float>(a, b, out, op, g, grid, stream); break;
        case 1: binary_typed<double>(a, b, out, op, g, grid, stream); break;
        case 2: binary_typed<cx<float>>(a, b, out, op, g, grid, stream); break;
        default: binary_typed<cx<double>>(a, b, out, op, g, grid, stream); break;
    }
}

// ---- slice regions: dsc_tensor_get_slice / set_slice (dsc.cpp:868-1169) ---------------------
// The reference walks a dsc_slice_iterator (dsc_iter.h:125-190) per element; here a block owns a piece of
// one innermost row of the region (one division chain per block), or — for narrow rows — a flat
// per-element decomposition.
namespace {

struct alignas(16) b16 { unsigned long long a, b; };
// elements per thread of region_rows_kernel: ONE 16-byte pack (the launch then sweeps memory in address order and every
// workgroup ends right after its store: x[:, :60000] 67 -> 73.5 % of the roofline), four of the narrower elements (with one
// per thread the strided x[:, ::2] drops from 49 to 35 %)
template<typename E> constexpr int region_u() { return sizeof(E) == 16 ? 1 : 4; }

template<typename E, bool SCATTER>
__global__ void region_rows_kernel(const E *src, E *dst, const dsc_region r, unsigned chunks_per_row, long long dense_ne) {
    const unsigned long long blk = blockIdx.x;
    const unsigned long long row = blk / chunks_per_row;
    const unsigned chunk = (unsigned) (blk - row * chunks_per_row);
    const unsigned long long i01 = row / r.count[2];
    const long long i2 = (long long) (row - i01 * r.count[2]);
    const long long i0 = (long long) (i01 / r.count[1]), i1 = (long long) (i01 - (unsigned long long) i0 * r.count[1]);
    const long long base = r.base + i0 * r.stride[0] + i1 * r.stride[1] + i2 * r.stride[2];
    const long long dense0 = (long long) row * r.count[3];
    constexpr int U = region_u<E>();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = chunk * (256 * U) + u * 256 + threadIdx.x;
        if (c >= r.count[3]) return;
        if (SCATTER) dst[base + c * r.stride[3]] = src[(dense0 + c) % dense_ne];
        else         dst[dense0 + c] = src[base + c * r.stride[3]];
    }
}

template<typename E, bool SCATTER>
__global__ void region_flat_kernel(const E *src, E *dst, const dsc_region r, long long dense_ne) {
    const long long s3 = r.count[3], s23 = s3 * r.count[2], s123 = s23 * r.count[1];
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < r.ne; i += (long long) gridDim.x * blockDim.x) {
        const long long i0 = i / s123, r0 = i - i0 * s123;
        const long long i1 = r0 / s23, r1 = r0 - i1 * s23;
        const long long i2 = r1 / s3, i3 = r1 - i2 * s3;
        const long long at = r.base + i0 * r.stride[0] + i1 * r.stride[1] + i2 * r.stride[2] + i3 * r.stride[3];
        if (SCATTER) dst[at] = src[i % dense_ne];
        else         dst[i] = src[at];
    }
}

template<typename E>
void region_typed(const void *src, void *dst, const dsc_region &r, bool scatter, long long dense_ne, hipStream_t s) {
    const E *ps = (const E *) src;
    E *pd = (E *) dst;
    const long long rows = r.ne / r.count[3];
    constexpr int per_block = 256 * region_u<E>();
    if (r.count[3] >= 128 && rows * ((r.count[3] + per_block - 1) / per_block) < (1LL << 31)) {
        const unsigned chunks = (unsigned) ((r.count[3] + per_block - 1) / per_block);
        const dim3 grid((unsigned) (rows * chunks));
        if (scatter) DSC_LAUNCH((region_rows_kernel<E, true>), grid, dim3(256), 0, s, ps, pd, r, chunks, dense_ne);
        else         DSC_LAUNCH((region_rows_kernel<E, false>), grid, dim3(256), 0, s, ps, pd, r, chunks, dense_ne);
    } else {
        const dim3 grid = stream_grid(r.ne);
        if (scatter) DSC_LAUNCH((region_flat_kernel<E, true>), grid, dim3(256), 0, s, ps, pd, r, dense_ne);
        else         DSC_LAUNCH((region_flat_kernel<E, false>), grid, dim3(256), 0, s, ps, pd, r, dense_ne);
    }
}

}  // namespace

void dsc_launch_region_copy(const void *src, void *dst, int elem_bytes, const dsc_region &r, bool scatter, long long dense_ne,
                            hipStream_t stream) {
    if (r.ne <= 0) return;
    // contiguous innermost rows whose ends fall on 16-byte boundaries (x[:, :60000], x[::2], a crop after irfft ...): move 16 bytes
    // per lane instead of one element
    const int V = 16 / elem_bytes;
    if (V > 1 && r.stride[3] == 1 && dense_ne == r.ne && r.count[3] % V == 0 && r.base % V == 0 && (((size_t) src | (size_t) dst) & 15) == 0) {
        bool ok = true;
        for (int k = 0; k < 3; ++k) ok = ok && (r.count[k] == 1 || r.stride[k] % V == 0);
        if (ok) {
            dsc_region w = r;
            w.base = r.base / V;
            w.count[3] = r.count[3] / V;
            for (int k = 0; k < 3; ++k) w.stride[k] = r.stride[k] / V;
            w.ne = r.ne / V;
            region_typed<b16>(src, dst, w, scatter, dense_ne / V, stream);
            return;
        }
    }
    switch (elem_bytes) {
        case 4:  region_typed<unsigned int>(src, dst, r, scatter, dense_ne, stream); break;
        case 8:  region_typed<unsigned long long>(src, dst, r, scatter, dense_ne, stream); break;
        default: region_typed<b16>(src, dst, r, scatter, dense_ne, stream); break;
    }
}

// ---- dsc_transpose of the last two axes (dsc.cpp:764-827 walks a stride iterator per element) ----
namespace {

template<typename E>
__global__ void transpose_last2_kernel(const E *in, E *out, int rows, int cols, unsigned tiles_c, unsigned tiles_r) {
    __shared__ E tile[32][33];
    const unsigned long long blk = blockIdx.x;
    const unsigned long long per = (unsigned long long) tiles_c * tiles_r;
    const unsigned long long b = blk / per;
    const unsigned rem = (unsigned) (blk - b * per);
    const unsigned tr = rem / tiles_c, tc = rem - tr * tiles_c;
    const E *src = in + b * (unsigned long long) rows * cols;
    E *dst = out + b * (unsigned long long) rows * cols;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8 threads
    for (int k = ty; k < 32; k += 8) {
        const int r = tr * 32 + k, c = tc * 32 + tx;
        if (r < rows && c < cols) tile[k][tx] = src[(long long) r * cols + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = tc * 32 + k, r = tr * 32 + tx;
        if (r < rows && c < cols) dst[(long long) c * rows + r] = tile[tx][k];
    }
}

template<typename E>
void transpose_typed(const void *in, void *out, long long batch, int rows, int cols, hipStream_t s) {
    const unsigned tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    const unsigned long long blocks = (unsigned long long) batch * tiles_c * tiles_r;
    DSC_LAUNCH((transpose_last2_kernel<E>), dim3((unsigned) blocks), dim3(256), 0, s, (const E *) in, (E *) out, rows, cols, tiles_c,
                       tiles_r);
}

}  // namespace

// ---- any permutation that moves the LAST axis (dsc_transpose with the reversed default, (2, 0, 1), ...): the plane spanned by the
// input's last axis (c, stride 1 in the input) and the input axis that becomes the output's last axis (a, stride 1 in the output) is
// transposed in 32 x 32 LDS tiles, both sides coalesced; the remaining (at most two) axes are a batch with their own strides.
namespace {

struct tr_plan {
    int na, nc;                 // extents along a and c
    long long sa_in, sc_out;    // input stride of a, output stride of c (elements)
    int nb0, nb1;               // batch extents
    long long b0_in, b0_out, b1_in, b1_out;
};

template<typename E>
__global__ void transpose_plane_kernel(const E *in, E *out, tr_plan p, unsigned tiles_a, unsigned tiles_c) {
    __shared__ E tile[32][33];
    unsigned long long blk = blockIdx.x;
    const unsigned tc = (unsigned) (blk % tiles_c); blk /= tiles_c;
    const unsigned ta = (unsigned) (blk % tiles_a); blk /= tiles_a;
    const unsigned i0 = (unsigned) (blk % (unsigned) p.nb0), i1 = (unsigned) (blk / (unsigned) p.nb0);
    const E *src = in + i0 * p.b0_in + i1 * p.b1_in;
    E *dst = out + i0 * p.b0_out + i1 * p.b1_out;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8 threads
    for (int k = ty; k < 32; k += 8) {
        const int a = ta * 32 + k, c = tc * 32 + tx;
        if (a < p.na && c < p.nc) tile[k][tx] = src[a * p.sa_in + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = tc * 32 + k, a = ta * 32 + tx;
        if (a < p.na && c < p.nc) dst[c * p.sc_out + a] = tile[tx][k];
    }
}

// The same plane transpose with 16-byte global accesses on BOTH sides (4- and 8-byte elements, V = 4 / 2 per access): a
// (16 V) x (16 V) tile, 16 x 16 threads; a thread loads V packs (rows ty + 16 k, columns tx V ..) and stores V packs (output
// rows ty + 16 k, elements tx V ..) gathered from V tile rows.  The 32 x 32 tiles move 128 bytes per row segment in f32
// (41-49 % of the roofline for [256, 512, 1024]); this form moves 256.  Needs na, nc, sa_in, sc_out and the batch strides to
// be multiples of V and 16-byte aligned bases: then a pack is never cut by the edge of the tensor.
template<typename E>
__global__ __launch_bounds__(256) void transpose_plane_vec_kernel(const E *in, E *out, tr_plan p, unsigned tiles_a, unsigned tiles_c) {
    constexpr int V = 16 / (int) sizeof(E), TD = 16 * V;
    __shared__ E tile[TD][TD + 1];
    unsigned long long blk = blockIdx.x;
    const unsigned tc = (unsigned) (blk % tiles_c); blk /= tiles_c;
    const unsigned ta = (unsigned) (blk % tiles_a); blk /= tiles_a;
    const unsigned i0 = (unsigned) (blk % (unsigned) p.nb0), i1 = (unsigned) (blk / (unsigned) p.nb0);
    const E *src = in + i0 * p.b0_in + i1 * p.b1_in;
    E *dst = out + i0 * p.b0_out + i1 * p.b1_out;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int la = ty + 16 * k, a = ta * TD + la, c = tc * TD + tx * V;
        if (a < p.na && c < p.nc) {
            const packed<E, V> q = *(const packed<E, V> *) (src + a * p.sa_in + c);
#pragma unroll
            for (int j = 0; j < V; ++j) tile[la][tx * V + j] = q.e[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int lc = ty + 16 * k, c = tc * TD + lc, a = ta * TD + tx * V;
        if (a < p.na && c < p.nc) {
            packed<E, V> q;
#pragma unroll
            for (int j = 0; j < V; ++j) q.e[j] = tile[tx * V + j][lc];
            *(packed<E, V> *) (dst + c * p.sc_out + a) = q;
        }
    }
}

template<typename E>
bool transpose_plane_vec(const void *in, void *out, const tr_plan &p, hipStream_t s) {
    constexpr int V = 16 / (int) sizeof(E), TD = 16 * V;
    if (V == 1 || !aligned_to(in, 16) || !aligned_to(out, 16)) return false;
    const long long must[] = {p.na, p.nc, p.sa_in, p.sc_out, p.nb0 > 1 ? p.b0_in : 0, p.nb0 > 1 ? p.b0_out : 0, p.nb1 > 1 ? p.b1_in : 0,
                              p.nb1 > 1 ? p.b1_out : 0};
    for (long long m : must) if (m % V != 0) return false;
    const unsigned tiles_a = (p.na + TD - 1) / TD, tiles_c = (p.nc + TD - 1) / TD;
    const unsigned long long blocks = (unsigned long long) tiles_a * tiles_c * p.nb0 * p.nb1;
    if (blocks == 0 || blocks > 0x7fffffffull) return false;
    DSC_LAUNCH((transpose_plane_vec_kernel<E>), dim3((unsigned) blocks), dim3(256), 0, s, (const E *) in, (E *) out, p, tiles_a, tiles_c);
    return true;
}

template<typename E>
void transpose_plane_typed(const void *in, void *out, const tr_plan &p, hipStream_t s) {
    if (transpose_plane_vec<E>(in, out, p, s)) return;
    const unsigned tiles_a = (p.na + 31) / 32, tiles_c = (p.nc + 31) / 32;
    const unsigned long long blocks = (unsigned long long) tiles_a * tiles_c * p.nb0 * p.nb1;
    DSC_LAUNCH((transpose_plane_kernel<E>), dim3((unsigned) blocks), dim3(256), 0, s, (const E *) in, (E *) out, p, tiles_a, tiles_c);
}

}  // namespace

// shape / in_stride: the INPUT's extents and element strides per axis (n_dim <= 4, dense); perm: result axis i = input axis perm[i],
// with perm[n_dim - 1] != n_dim - 1.  Returns false if the launch would not fit (the caller keeps the strided copy).
bool dsc_launch_transpose_moving_last(const void *in, void *out, int elem_bytes, int n_dim, const int *shape, const int *in_stride, const int *perm,
                                      hipStream_t stream) {
    long long out_stride[4] = {1, 1, 1, 1};
    for (int i = n_dim - 2; i >= 0; --i) out_stride[i] = out_stride[i + 1] * shape[perm[i + 1]];
    const int a_axis = perm[n_dim - 1], c_axis = n_dim - 1;            // input axes of the tile plane
    tr_plan p;
    p.na = shape[a_axis]; p.nc = shape[c_axis];
    p.sa_in = in_stride[a_axis];
    p.sc_out = 1;
    p.nb0 = p.nb1 = 1; p.b0_in = p.b0_out = p.b1_in = p.b1_out = 0;
    int nb = 0;
    for (int i = 0; i < n_dim; ++i) {                                   // result axis i <- input axis perm[i]
        const int ax = perm[i];
        if (ax == c_axis) { p.sc_out = out_stride[i]; continue; }
        if (ax == a_axis) continue;
        if (nb == 0) { p.nb0 = shape[ax]; p.b0_in = in_stride[ax]; p.b0_out = out_stride[i]; }
        else         { p.nb1 = shape[ax]; p.b1_in = in_stride[ax]; p.b1_out = out_stride[i]; }
        ++nb;
    }
    const unsigned long long blocks = (unsigned long long) ((p.na + 31) / 32) * ((p.nc + 31) / 32) * p.nb0 * p.nb1;
    if (blocks == 0) return true;
    if (blocks > 0x7fffffffull) return false;
    switch (elem_bytes) {
        case 4:  transpose_plane_typed<unsigned int>(in, out, p, stream); break;
        case 8:  transpose_plane_typed<unsigned long long>(in, out, p, stream); break;
        default: transpose_plane_typed<b16>(in, out, p, stream); break;
    }
    return true;
}

void dsc_launch_transpose_last2(const void *in, void *out, int elem_bytes, long long batch, int rows, int cols, hipStream_t stream) {
    if (batch <= 0 || rows <= 0 || cols <= 0) return;
    if (elem_bytes < 16 && batch < (1LL << 31)) {                       // [batch][rows][cols] -> [batch][cols][rows] as a plane plan
        tr_plan p;
        p.na = rows; p.nc = cols; p.sa_in = cols; p.sc_out = rows;
        p.nb0 = (int) batch; p.b0_in = p.b0_out = (long long) rows * cols;
        p.nb1 = 1; p.b1_in = p.b1_out = 0;
        if (elem_bytes == 4 ? transpose_plane_vec<unsigned int>(in, out, p, stream) : transpose_plane_vec<unsigned long long>(in, out, p, stream)) return;
    }
    switch (elem_bytes) {
        case 4:  transpose_typed<unsigned int>(in, out, batch, rows, cols, stream); break;
        case 8:  transpose_typed<unsigned long long>(in, out, batch, rows, cols, stream); break;
        default: transpose_typed<b16>(in, out, batch, rows, cols, stream); break;
    }
}
