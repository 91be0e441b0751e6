// reduce.hip — sum / mean / max / min along one axis.
//
// Reference: dsc/src/dsc.cpp:1771-1953 (one sequential accumulation in T per output element,
// source index from dsc_axis_iterator) with add_op / max_op / min_op (dsc_ops.h:46-55,
// 318-339).  The tensor is viewed as [outer][axis_n][inner]:
//
//   inner > 1   one thread per output element walks the axis sequentially; neighbouring
//               threads read neighbouring addresses, so the walk is coalesced AND keeps the
//               reference's left-to-right accumulation order (sums are bit-identical).  When
//               that gives too few threads for the chip (few outputs, long axis) the axis is
//               cut into segments reduced in parallel and combined in segment order.
//   inner == 1  (reduction over the contiguous last axis) one workgroup per output element:
//               strided partial accumulators per thread, then a wave shuffle + LDS tree.
//               Sum order differs from the reference here (tolerance in the tests).
//
// max/min select an element, so they are exact either way — including ties and NaNs, which follow
// from the reference's predicates applied left to right (dsc_ops.h:318-339, dsc.h:43-44):
//   max        acc = acc > x ? acc : x          complex: .real only; ties -> the LATER element
//   min real   acc = acc < x ? acc : x          ties -> the later element
//   min cplx   acc = acc.real > x.real ? x : acc   ties -> the EARLIER element
// A NaN compares false: max and the real min TAKE a NaN x and then drop it again at the next element,
// so the result is the max / min of the elements AFTER the last NaN (the NaN itself if it is last);
// the complex min never takes one.  The sequential kernels apply the predicate as is; partial results
// carry "contains a NaN", which makes the ordered combination exact (a segment with a NaN wipes what
// came before it); the tree kernel finds the last NaN of the row first.
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <type_traits>

namespace {

template<typename T> struct alignas(2 * sizeof(T)) cx { T x, y; };

template<typename R, bool CPLX> struct acc_t { R r, i; int idx; };

template<typename R, bool CPLX, int OP>
__device__ __forceinline__ acc_t<R, CPLX> acc_init() {
    const R inf = (R) INFINITY;
    if (OP == 2) return {-inf, -inf, -1};
    if (OP == 3) return {inf, inf, -1};
    return {(R) 0, (R) 0, -1};
}

// The reference's step acc <- op(acc, x), x the NEXT element along the axis (a partial result of a later segment when
// called from the ordered combination).  acc.idx doubles as "this partial contains a NaN" (max, real min).
template<int OP, bool CPLX> constexpr bool nan_wipes() { return OP == 2 || (OP == 3 && !CPLX); }
template<typename R, bool CPLX, int OP>
__device__ __forceinline__ acc_t<R, CPLX> step(acc_t<R, CPLX> acc, acc_t<R, CPLX> x) {
    bool take_x;
    if (OP == 2)    take_x = !(acc.r > x.r);
    else if (!CPLX) take_x = !(acc.r < x.r);
    else            take_x = acc.r > x.r;
    if (nan_wipes<OP, CPLX>() && x.idx > 0) take_x = true;            // x is a partial that saw a NaN: it decides alone
    acc_t<R, CPLX> o = take_x ? x : acc;
    o.idx = (acc.idx > 0 || x.idx > 0 || (nan_wipes<OP, CPLX>() && x.r != x.r)) ? 1 : 0;
    return o;
}

// combine(a, b) for the tree kernel: order-free by position (idx), NaN-free inputs only
template<typename R, bool CPLX, int OP>
__device__ __forceinline__ acc_t<R, CPLX> combine(acc_t<R, CPLX> a, acc_t<R, CPLX> b) {
    if (OP <= 1) return {a.r + b.r, a.i + b.i, 0};
    if (b.idx < 0) return a;
    if (a.idx < 0) return b;
    const bool later_wins = (OP == 2) || !CPLX;
    bool take_b;
    if (OP == 2) take_b = b.r > a.r;
    else         take_b = b.r < a.r;
    if (a.r == b.r) take_b = later_wins ? (b.idx > a.idx) : (b.idx < a.idx);
    return take_b ? b : a;
}

template<typename R, bool CPLX>
__device__ __forceinline__ acc_t<R, CPLX> load_elem(const void *x, long long i, int idx) {
    if (CPLX) { const cx<R> v = ((const cx<R> *) x)[i]; return {v.x, v.y, idx}; }
    return {((const R *) x)[i], (R) 0, idx};
}

template<typename R, bool CPLX, int OP>
__device__ __forceinline__ void store_elem(void *out, long long i, acc_t<R, CPLX> a, int axis_n) {
    if (OP == 1) {      // dsc_mean: sum, then mul_op by (1/axis_n [, 0])   dsc.cpp:1837-1853
        const R s = (R) 1 / (R) axis_n;
        if (CPLX) { const R r = (a.r * s) - (a.i * (R) 0), m = (a.r * (R) 0) + (a.i * s); a.r = r; a.i = m; }
        else      { a.r = a.r * s; }
    }
    if (CPLX) ((cx<R> *) out)[i] = cx<R>{a.r, a.i};
    else      ((R *) out)[i] = a.r;
}

// one thread per V neighbouring outputs (V elements = 16 bytes per load where the inner extent allows it, else V = 1), sequential
// along the axis (reference order)
template<typename R, bool CPLX, int V> struct alignas(sizeof(R) * (CPLX ? 2 : 1) * V) in_pack { R e[(CPLX ? 2 : 1) * V]; };

template<typename R, bool CPLX, int V>
__device__ __forceinline__ void load_pack(const void *x, long long i, acc_t<R, CPLX> (&v)[V]) {
    using E = typename std::conditional<CPLX, cx<R>, R>::type;
    const in_pack<R, CPLX, V> q = *(const in_pack<R, CPLX, V> *) ((const E *) x + i);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = CPLX ? acc_t<R, CPLX>{q.e[2 * k], q.e[2 * k + 1], 0} : acc_t<R, CPLX>{q.e[k], (R) 0, 0};
}

template<typename R, bool CPLX, int OP, int V>
__global__ void reduce_seq_kernel(const void *x, void *out, long long outer, int axis_n, long long inner) {
    const long long n_out = outer * inner, n_thr = n_out / V;
    for (long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < n_thr; t += (long long) gridDim.x * blockDim.x) {
        const long long o = t * V;
        const long long oo = o / inner, ii = o - oo * inner;
        const long long base = oo * axis_n * inner + ii;
        acc_t<R, CPLX> acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { acc[k] = acc_init<R, CPLX, OP>(); acc[k].idx = 0; }
#pragma unroll 8
        for (int j = 0; j < axis_n; ++j) {
            acc_t<R, CPLX> v[V];
            load_pack<R, CPLX, V>(x, base + (long long) j * inner, v);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                if (OP <= 1) { acc[k].r = acc[k].r + v[k].r; acc[k].i = acc[k].i + v[k].i; }
                else acc[k] = step<R, CPLX, OP>(acc[k], v[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) store_elem<R, CPLX, OP>(out, o + k, acc[k], axis_n);
    }
}

// Same walk, but only rows [seg * seg_len, (seg + 1) * seg_len) of the axis, partial result
// (value + position along the axis, for the tie rules) to a workspace: used when there are too
// few outputs to fill the chip (e.g. axis 0 of [4096, 32769]).  Segments are combined in order
// by reduce_combine_kernel, so the result does not depend on scheduling.
template<typename R, bool CPLX, int OP, int V>
__global__ void reduce_seg_kernel(const void *x, R *part_r, R *part_i, int *part_idx, long long outer, int axis_n,
                                  long long inner, int seg_len) {
    const long long n_out = outer * inner, n_thr = n_out / V;
    const int seg = blockIdx.y;
    const int j0 = seg * seg_len, j1 = j0 + seg_len < axis_n ? j0 + seg_len : axis_n;
    for (long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < n_thr; t += (long long) gridDim.x * blockDim.x) {
        const long long o = t * V;
        const long long oo = o / inner, ii = o - oo * inner;
        const long long base = oo * axis_n * inner + ii;
        acc_t<R, CPLX> acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { acc[k] = acc_init<R, CPLX, OP>(); acc[k].idx = 0; }
#pragma unroll 4
        for (int j = j0; j < j1; ++j) {
            acc_t<R, CPLX> v[V];
            load_pack<R, CPLX, V>(x, base + (long long) j * inner, v);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                if (OP <= 1) { acc[k].r = acc[k].r + v[k].r; acc[k].i = acc[k].i + v[k].i; }
                else acc[k] = step<R, CPLX, OP>(acc[k], v[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const long long at = (long long) seg * n_out + o + k;
            part_r[at] = acc[k].r;
            if (CPLX) part_i[at] = acc[k].i;
            if (OP >= 2) part_idx[at] = acc[k].idx;
        }
    }
}

template<typename R, bool CPLX, int OP>
__global__ void reduce_combine_kernel(const R *part_r, const R *part_i, const int *part_idx, void *out, long long n_out,
                                      int n_seg, int axis_n) {
    for (long long o = (long long) blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += (long long) gridDim.x * blockDim.x) {
        acc_t<R, CPLX> acc = acc_init<R, CPLX, OP>();
        acc.idx = 0;
        for (int s = 0; s < n_seg; ++s) {
            const long long at = (long long) s * n_out + o;
            acc_t<R, CPLX> v = {part_r[at], CPLX ? part_i[at] : (R) 0, OP >= 2 ? part_idx[at] : 0};      // idx = the segment saw a NaN
            if (OP <= 1) { acc.r = acc.r + v.r; acc.i = acc.i + v.i; }
            else acc = step<R, CPLX, OP>(acc, v);
        }
        store_elem<R, CPLX, OP>(out, o, acc, axis_n);
    }
}

template<typename R, bool CPLX>
__device__ __forceinline__ acc_t<R, CPLX> shfl_down_acc(acc_t<R, CPLX> a, int delta) {
    acc_t<R, CPLX> b;
    b.r = __shfl_down(a.r, delta, 64);
    b.i = CPLX ? __shfl_down(a.i, delta, 64) : (R) 0;
    b.idx = __shfl_down(a.idx, delta, 64);
    return b;
}

// one 256-thread workgroup per output row (inner == 1).  max / min: the threads' strided partial results are combined by
// position (idx), which is only the reference's left-to-right result when no NaN is involved; NaNs are therefore kept out of
// the candidates and their last position p is found alongside — the row's answer is then the max / min over j > p (a second
// sweep, rows with a NaN only), or element p itself when p is the last element.  The complex min simply skips NaNs.
template<typename R, bool CPLX, int OP>
__device__ __forceinline__ acc_t<R, CPLX> block_combine(acc_t<R, CPLX> acc, acc_t<R, CPLX> (&part)[4]) {
    for (int d = 32; d > 0; d >>= 1) acc = combine<R, CPLX, OP>(acc, shfl_down_acc<R, CPLX>(acc, d));
    __syncthreads();                                       // part[] may still be read from the previous use
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    return combine<R, CPLX, OP>(combine<R, CPLX, OP>(part[0], part[1]), combine<R, CPLX, OP>(part[2], part[3]));
}

template<typename R, bool CPLX, int OP>
__global__ __launch_bounds__(256) void reduce_row_kernel(const void *x, void *out, long long outer, int axis_n) {
    __shared__ acc_t<R, CPLX> part[4];
    __shared__ int last_nan_s[4];
    for (long long row = blockIdx.x; row < outer; row += gridDim.x) {
        const long long base = row * axis_n;
        acc_t<R, CPLX> acc = acc_init<R, CPLX, OP>();
        int last_nan = -1;
        int j = threadIdx.x;
        for (; j + 3 * 256 < axis_n; j += 4 * 256) {       // four loads in flight, consumed in ascending order (max / min c32: 46-61 -> 70 %)
            acc_t<R, CPLX> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load_elem<R, CPLX>(x, base + j + u * 256, j + u * 256);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (OP >= 2 && v[u].r != v[u].r) last_nan = j + u * 256;
                else acc = combine<R, CPLX, OP>(acc, v[u]);
            }
        }
        for (; j < axis_n; j += 256) {
            const acc_t<R, CPLX> v = load_elem<R, CPLX>(x, base + j, j);
            if (OP >= 2 && v.r != v.r) last_nan = j;       // j ascends: the thread's last NaN
            else acc = combine<R, CPLX, OP>(acc, v);
        }
        acc = block_combine<R, CPLX, OP>(acc, part);
        if (OP >= 2 && nan_wipes<OP, CPLX>()) {
            for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_down(last_nan, d, 64); last_nan = o > last_nan ? o : last_nan; }
            if ((threadIdx.x & 63) == 0) last_nan_s[threadIdx.x >> 6] = last_nan;
            __syncthreads();
            int p = last_nan_s[0];
            for (int w = 1; w < 4; ++w) p = last_nan_s[w] > p ? last_nan_s[w] : p;
            if (p >= 0) {                                   // the reference's accumulator forgot everything up to the last NaN
                acc = acc_init<R, CPLX, OP>();
                for (int j = p + 1 + threadIdx.x; j < axis_n; j += 256) acc = combine<R, CPLX, OP>(acc, load_elem<R, CPLX>(x, base + j, j));
                acc = block_combine<R, CPLX, OP>(acc, part);
                if (p == axis_n - 1) acc = load_elem<R, CPLX>(x, base + p, p);
            }
        }
        if (threadIdx.x == 0) store_elem<R, CPLX, OP>(out, row, acc, axis_n);
        __syncthreads();
    }
}

// Short rows (64 .. 2048 elements), many of them: one WAVE per row (four rows per workgroup), shuffles only — a 256-thread workgroup
// per 4 KiB row spends its time in barriers (sum f32 [131072, 1024]: 42 % of the roofline).  Same NaN rules as reduce_row_kernel.
template<typename R, bool CPLX, int OP>
__global__ __launch_bounds__(256) void reduce_row_wave_kernel(const void *x, void *out, long long outer, int axis_n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long row = (long long) blockIdx.x * 4 + w; row < outer; row += (long long) gridDim.x * 4) {
        const long long base = row * axis_n;
        acc_t<R, CPLX> acc = acc_init<R, CPLX, OP>();
        int last_nan = -1;
#pragma unroll 4
        for (int j = lane; j < axis_n; j += 64) {
            const acc_t<R, CPLX> v = load_elem<R, CPLX>(x, base + j, j);
            if (OP >= 2 && v.r != v.r) last_nan = j;
            else acc = combine<R, CPLX, OP>(acc, v);
        }
        for (int d = 32; d > 0; d >>= 1) acc = combine<R, CPLX, OP>(acc, shfl_down_acc<R, CPLX>(acc, d));
        if (OP >= 2 && nan_wipes<OP, CPLX>()) {
            for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_down(last_nan, d, 64); last_nan = o > last_nan ? o : last_nan; }
            const int p = __shfl(last_nan, 0, 64);
            if (p >= 0) {                                   // the reference's accumulator forgot everything up to the last NaN
                acc = acc_init<R, CPLX, OP>();
                for (int j = p + 1 + lane; j < axis_n; j += 64) acc = combine<R, CPLX, OP>(acc, load_elem<R, CPLX>(x, base + j, j));
                for (int d = 32; d > 0; d >>= 1) acc = combine<R, CPLX, OP>(acc, shfl_down_acc<R, CPLX>(acc, d));
                if (p == axis_n - 1) acc = load_elem<R, CPLX>(x, base + p, p);
            }
        }
        if (lane == 0) store_elem<R, CPLX, OP>(out, row, acc, axis_n);
    }
}

template<typename R, bool CPLX, int OP>
void launch_op(const void *x, void *out, long long outer, int axis_n, long long inner, void *ws, size_t ws_bytes, hipStream_t s) {
    const long long n_out_all = outer * inner;
    // 16 bytes per load along the inner extent where every row of it starts on a 16-byte boundary and there are enough outputs
    // left to fill the chip with one pack per thread
    constexpr int VP = 16 / (int) (sizeof(R) * (CPLX ? 2 : 1));
    const bool packs = VP > 1 && inner % VP == 0 && (((size_t) x) & 15) == 0 && n_out_all / VP >= 256 * 512;      // measured: 131072 threads of packs 68 -> 74 %, 65536 threads 74 -> 70 %
    // too few outputs for one thread each to fill the chip, long axis: split the axis
    if (!(inner == 1 && axis_n >= 64) && n_out_all < 256 * 1024 && axis_n >= 128 && ws != nullptr) {
        long long n_seg = (512 * 1024 + n_out_all - 1) / n_out_all;
        if (n_seg > axis_n / 32) n_seg = axis_n / 32;
        if (n_seg > 4096) n_seg = 4096;                 // grid.y (HIP allows 65535); a few thousand segments already fill the chip
        const size_t per = (size_t) n_out_all * (2 * sizeof(R) + sizeof(int));
        if (n_seg * per > ws_bytes) n_seg = (long long) (ws_bytes / per);
        if (n_seg >= 2) {
            const int seg_len = (int) ((axis_n + n_seg - 1) / n_seg);
            n_seg = (axis_n + seg_len - 1) / seg_len;
            R *pr = (R *) ws;
            R *pi = pr + n_seg * n_out_all;
            int *pidx = (int *) (pi + n_seg * n_out_all);
            const unsigned bx = (unsigned) ((n_out_all + 255) / 256);
            // (one output per thread: the pack form needs 131072 packs = 262144+ outputs, which this path — fewer than 262144
            // outputs — never has; only the V = 1 form of the kernel is instantiated)
            DSC_LAUNCH((reduce_seg_kernel<R, CPLX, OP, 1>), dim3(bx, (unsigned) n_seg), dim3(256), 0, s, x, pr, pi, pidx, outer,
                               axis_n, inner, seg_len);
            DSC_LAUNCH((reduce_combine_kernel<R, CPLX, OP>), dim3(bx), dim3(256), 0, s, pr, pi, pidx, out, n_out_all,
                               (int) n_seg, axis_n);
            return;
        }
    }
    if (inner == 1 && axis_n >= 64 && axis_n <= 2048 && outer >= 1024) {
        long long blocks = (outer + 3) / 4;
        if (blocks > 256 * 32) blocks = 256 * 32;
        DSC_LAUNCH((reduce_row_wave_kernel<R, CPLX, OP>), dim3((unsigned) blocks), dim3(256), 0, s, x, out, outer, axis_n);
    } else if (inner == 1 && axis_n >= 64) {
        long long blocks = outer < 256 * 16 ? outer : 256 * 16;
        DSC_LAUNCH((reduce_row_kernel<R, CPLX, OP>), dim3((unsigned) blocks), dim3(256), 0, s, x, out, outer, axis_n);
    } else {
        const long long n_out = outer * inner;
        long long blocks = (n_out / (packs ? VP : 1) + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (packs) DSC_LAUNCH((reduce_seq_kernel<R, CPLX, OP, VP>), dim3((unsigned) blocks), dim3(256), 0, s, x, out, outer, axis_n, inner);
        else       DSC_LAUNCH((reduce_seq_kernel<R, CPLX, OP, 1>), dim3((unsigned) blocks), dim3(256), 0, s, x, out, outer, axis_n, inner);
    }
}

template<typename R, bool CPLX>
void launch_typed(const void *x, void *out, int op, long long outer, int axis_n, long long inner, void *ws, size_t wsb, hipStream_t s) {
    switch (op) {
        case 0: launch_op<R, CPLX, 0>(x, out, outer, axis_n, inner, ws, wsb, s); break;
        case 1: launch_op<R, CPLX, 1>(x, out, outer, axis_n, inner, ws, wsb, s); break;
        case 2: launch_op<R, CPLX, 2>(x, out, outer, axis_n, inner, ws, wsb, s); break;
        default: launch_op<R, CPLX, 3>(x, out, outer, axis_n, inner, ws, wsb, s); break;
    }
}

}  // namespace

void dsc_launch_reduce(const void *x, void *out, int dtype, int op, long long outer, int axis_n, long long inner,
                       void *workspace, size_t workspace_bytes, hipStream_t stream) {
    if (outer * inner <= 0) return;
    switch (dtype) {
        case 0: launch_typed<float, false>(x, out, op, outer, axis_n, inner, workspace, workspace_bytes, stream); break;
        case 1: launch_typed<double, false>(x, out, op, outer, axis_n, inner, workspace, workspace_bytes, stream); break;
        case 2: launch_typed<float, true>(x, out, op, outer, axis_n, inner, workspace, workspace_bytes, stream); break;
        default: launch_typed<double, true>(x, out, op, outer, axis_n, inner, workspace, workspace_bytes, stream); break;
    }
}
