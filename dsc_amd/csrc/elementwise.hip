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

namespace {

template<typename T> struct alignas(2 * sizeof(T)) cx { T x, y; };

template<typename T> struct elem;   // dtype code -> storage type
template<> struct elem<float>  { static constexpr bool cplx = false; using real = float; };
template<> struct elem<double> { static constexpr bool cplx = false; using real = double; };
template<> struct elem<cx<float>>  { static constexpr bool cplx = true; using real = float; };
template<> struct elem<cx<double>> { static constexpr bool cplx = true; using real = double; };

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

template<typename Tin>
void cast_from(const void *in, void *out, int out_dtype, long long ne, dim3 grid, hipStream_t s) {
    const Tin *x = (const Tin *) in;
    switch (out_dtype) {
        case 0: DSC_LAUNCH((cast_kernel<Tin, float>), grid, dim3(256), 0, s, x, (float *) out, ne); break;
        case 1: DSC_LAUNCH((cast_kernel<Tin, double>), grid, dim3(256), 0, s, x, (double *) out, ne); break;
        case 2: DSC_LAUNCH((cast_kernel<Tin, cx<float>>), grid, dim3(256), 0, s, x, (cx<float> *) out, ne); break;
        default: DSC_LAUNCH((cast_kernel<Tin, cx<double>>), grid, dim3(256), 0, s, x, (cx<double> *) out, ne); break;
    }
}

inline dim3 stream_grid(long long ne) {
    long long blocks = (ne + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;        // 8 blocks per CU, grid-stride beyond that
    if (blocks < 1) blocks = 1;
    return dim3((unsigned) blocks);
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

template<typename T, int OP>
bool binary_fast(const T *pa, const T *pb, T *po, const dsc_bcast_args &g, dim3 grid, hipStream_t s) {
    if (g.a_scalar || g.b_scalar || g.fast == 0) return false;
    const unsigned ne = (unsigned) g.ne, sm = (unsigned) g.small_ne;
    constexpr unsigned V = 16 / sizeof(T);
    if (g.fast == 1 && V > 1 && ne % V == 0 && (((size_t) pa | (size_t) pb | (size_t) po) & 15) == 0) {
        DSC_LAUNCH((binary_same_vec_kernel<T, OP>), stream_grid(ne / V), dim3(256), 0, s, pa, pb, po, ne / V);
        return true;
    }
    if (g.fast == 1) DSC_LAUNCH((binary_fast_kernel<T, OP, 1>), grid, dim3(256), 0, s, pa, pb, po, ne, sm);
    else if (g.fast == 2) DSC_LAUNCH((binary_fast_kernel<T, OP, 2>), grid, dim3(256), 0, s, pa, pb, po, ne, sm);
    else DSC_LAUNCH((binary_fast_kernel<T, OP, 3>), grid, dim3(256), 0, s, pa, pb, po, ne, sm);
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
template<typename Tin, int OP>
__global__ void unary_kernel(const Tin *in, void *out, long long ne) {
    using R = typename elem<Tin>::real;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (long long) gridDim.x * blockDim.x) {
        R re, im;
        if constexpr (elem<Tin>::cplx) { const Tin v = in[i]; re = v.x; im = v.y; }
        else                           { re = in[i]; im = (R) 0; }
        if constexpr (OP == 0) {
            if constexpr (elem<Tin>::cplx) ((R *) out)[i] = sqrt((re * re) + (im * im));
            else                           ((R *) out)[i] = re >= 0 ? re : -re;
        } else if constexpr (OP == 1) {
            ((R *) out)[i] = atan2(im, re);
        } else if constexpr (OP == 2) {
            if constexpr (elem<Tin>::cplx) ((Tin *) out)[i] = Tin{re, -im};
            else                           ((R *) out)[i] = re;
        } else if constexpr (OP == 3) {
            ((R *) out)[i] = re;
        } else {
            ((R *) out)[i] = im;
        }
    }
}

template<typename Tin>
void unary_typed(const void *in, void *out, int op, long long ne, dim3 grid, hipStream_t s) {
    const Tin *x = (const Tin *) in;
    switch (op) {
        case 0: DSC_LAUNCH((unary_kernel<Tin, 0>), grid, dim3(256), 0, s, x, out, ne); break;
        case 1: DSC_LAUNCH((unary_kernel<Tin, 1>), grid, dim3(256), 0, s, x, out, ne); break;
        case 2: DSC_LAUNCH((unary_kernel<Tin, 2>), grid, dim3(256), 0, s, x, out, ne); break;
        case 3: DSC_LAUNCH((unary_kernel<Tin, 3>), grid, dim3(256), 0, s, x, out, ne); break;
        default: DSC_LAUNCH((unary_kernel<Tin, 4>), grid, dim3(256), 0, s, x, out, ne); break;
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
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                   // four elements per thread: 1024 per block
        const int c = chunk * 1024 + u * 256 + threadIdx.x;
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
    if (r.count[3] >= 128 && rows * ((r.count[3] + 1023) / 1024) < (1LL << 31)) {
        const unsigned chunks = (unsigned) ((r.count[3] + 1023) / 1024);
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

template<typename E>
void transpose_plane_typed(const void *in, void *out, const tr_plan &p, hipStream_t s) {
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
    switch (elem_bytes) {
        case 4:  transpose_typed<unsigned int>(in, out, batch, rows, cols, stream); break;
        case 8:  transpose_typed<unsigned long long>(in, out, batch, rows, cols, stream); break;
        default: transpose_typed<b16>(in, out, batch, rows, cols, stream); break;
    }
}
