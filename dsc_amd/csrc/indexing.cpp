// indexing.cpp — dsc_tensor_get_idx / get_slice / set_idx / set_slice on device tensors.
//
// Host-side mirror of dsc/src/dsc.cpp:829-1169: the same argument parsing, defaults, shape rules and
// assertions; the element walk of dsc_slice_iterator (dsc_iter.h:125-190) becomes one strided
// gather / scatter launch over the region (elementwise.hip), so the `[:output_length]` crop after
// dsc_irfft (README.md:133) and block placement into padded buffers stay in HBM.
#include "dsc_internal.h"
#include "kernels.h"

#include <cstdarg>
#include <cstdlib>
#include <cstring>

// dsc.h:81
static inline int tensor_dim(const dsc_tensor *x, int dim) { return dim < 0 ? DSC_MAX_DIMS + dim : DSC_MAX_DIMS - x->n_dim + dim; }

// dsc.cpp:883-933
static void parse_slices(const dsc_tensor *x, dsc_slice *parsed, bool *collapse_dim, int slices, va_list args) {
    for (int i = 0; i < slices; ++i) {
        dsc_slice slice = va_arg(args, dsc_slice);
        const int x_dim_i = x->shape[tensor_dim(x, i)];

        if (slice.start == slice.stop && slice.start == slice.step && slice.start != DSC_VALUE_NONE) {   // a single index
            if (collapse_dim != nullptr) collapse_dim[i] = true;
            slice.step = 1;
            if (slice.start < 0) {
                slice.start += x_dim_i;
                slice.stop += x_dim_i + 1;
            } else {
                slice.stop += 1;
            }
        }
        DSC_ASSERT(slice.step != 0);

        if (slice.step == DSC_VALUE_NONE) slice.step = 1;
        if (slice.start == DSC_VALUE_NONE) slice.start = slice.step > 0 ? 0 : x_dim_i - 1;
        if (slice.stop == DSC_VALUE_NONE) slice.stop = slice.step > 0 ? x_dim_i : -x_dim_i - 1;

        if (slice.start < 0) slice.start += x_dim_i;
        if (slice.stop < 0) slice.stop += x_dim_i;

        DSC_ASSERT(abs(slice.stop - slice.start) <= x_dim_i);
        DSC_ASSERT((slice.step > 0 && slice.start < slice.stop) || (slice.step < 0 && slice.start > slice.stop));
        DSC_ASSERT(abs(slice.step) <= x_dim_i);
        // The reference would read out of bounds for a start past the end (e.g. x[5:3:-1] on a dim of 4); on the
        // device that is a fault, so it is refused here.
        DSC_ASSERT(slice.start >= 0 && slice.start < x_dim_i);
        {
            const int n_i = (abs(slice.stop - slice.start) + abs(slice.step) - 1) / abs(slice.step);
            const int last = slice.start + (n_i - 1) * slice.step;               // e.g. x[5:14] on a dim of 10
            DSC_ASSERT(last >= 0 && last < x_dim_i);
        }

        parsed[i] = slice;
    }
}

static inline int slice_count(const dsc_slice &s) {
    const int ne = abs(s.stop - s.start), st = abs(s.step);
    return (ne + st - 1) / st;
}

// The region of x selected by the first n_slices dims (the others in full), in the iterator's order
// (dsc_iter.h:127-141: slot = right-aligned dim).
static dsc_region region_of(const dsc_tensor *x, int n_slices, const dsc_slice *slices) {
    dsc_region r;
    r.base = 0;
    r.ne = 1;
    for (int d = 0; d < DSC_MAX_DIMS; ++d) { r.count[d] = 1; r.stride[d] = 0; }
    for (int i = 0; i < x->n_dim; ++i) {
        const int slot = tensor_dim(x, i);
        if (i < n_slices) {
            r.count[slot] = slice_count(slices[i]);
            r.stride[slot] = (long long) slices[i].step * x->stride[slot];
            r.base += (long long) slices[i].start * x->stride[slot];
        } else {
            r.count[slot] = x->shape[slot];
            r.stride[slot] = x->stride[slot];
        }
        r.ne *= r.count[slot];
    }
    return r;
}

// dsc.cpp:832-866
extern "C" dsc_tensor *dsc_tensor_get_idx(dsc_ctx *ctx, const dsc_tensor *x, int indexes, ...) {
    DSC_ASSERT(x != nullptr);
    DSC_TRACE_OP(ctx, "idx;get", x);
    DSC_ASSERT((unsigned) indexes <= DSC_MAX_DIMS);
    if (indexes > x->n_dim) DSC_LOG_FATAL("too many indexes");
    DSC_ASSERT(indexes >= 1);                       // the reference reads stride[-1] for zero indexes

    int el_idx[DSC_MAX_DIMS];
    va_list args;
    va_start(args, indexes);
    for (int i = 0; i < indexes; ++i) {
        int idx = va_arg(args, int);
        const int x_dim_i = x->shape[tensor_dim(x, i)];
        if (idx < 0) idx += x_dim_i;
        DSC_ASSERT((unsigned) idx < (unsigned) x_dim_i);
        el_idx[i] = idx;
    }
    va_end(args);

    const int out_n_dim = x->n_dim == indexes ? 1 : x->n_dim - indexes;
    int out_shape[DSC_MAX_DIMS] = {1};
    if (x->n_dim > indexes) memcpy(out_shape, &x->shape[DSC_MAX_DIMS - out_n_dim], out_n_dim * sizeof(*x->shape));
    dsc_tensor *out = dsc_new_tensor(ctx, out_n_dim, out_shape, x->dtype, nullptr);

    long long offset = 0;
    for (int i = 0; i < indexes; ++i) offset += (long long) x->stride[tensor_dim(x, i)] * el_idx[i];
    const long long count = x->stride[tensor_dim(x, indexes - 1)];        // elements of the selected sub-tensor
    const size_t esz = dsc_dtype_size(x->dtype);
    HIP_CHECK(hipMemcpyAsync(out->data, (const char *) x->data + offset * esz, count * esz, hipMemcpyDeviceToDevice, ctx->stream));
    return out;
}

// dsc.cpp:935-992
extern "C" dsc_tensor *dsc_tensor_get_slice(dsc_ctx *ctx, const dsc_tensor *x, int slices, ...) {
    DSC_ASSERT(x != nullptr);
    DSC_TRACE_OP(ctx, "slice;get", x);
    DSC_ASSERT((unsigned) slices <= DSC_MAX_DIMS);
    if (slices > x->n_dim) DSC_LOG_FATAL("too many slices");

    dsc_slice el_slices[DSC_MAX_DIMS];
    bool collapse_dim[DSC_MAX_DIMS] = {false};
    va_list args;
    va_start(args, slices);
    parse_slices(x, el_slices, collapse_dim, slices, args);
    va_end(args);

    int out_shape[DSC_MAX_DIMS];
    int out_n_dim = x->n_dim;
    for (int i = 0, out_idx = 0; i < x->n_dim; ++i) {
        if (i < slices) {
            if (collapse_dim[i]) { out_n_dim -= 1; continue; }
            out_shape[out_idx] = slice_count(el_slices[i]);
        } else {
            out_shape[out_idx] = x->shape[tensor_dim(x, i)];
        }
        out_idx += 1;
    }
    DSC_ASSERT(out_n_dim >= 1);                      // every dim collapsed: use dsc_tensor_get_idx (the wrapper does)
    dsc_tensor *out = dsc_new_tensor(ctx, out_n_dim, out_shape, x->dtype, nullptr);

    const dsc_region r = region_of(x, slices, el_slices);
    DSC_ASSERT(r.ne == out->ne);
    dsc_launch_region_copy(x->data, out->data, (int) dsc_dtype_size(x->dtype), r, false, out->ne, ctx->stream);
    return out;
}

// dsc.cpp:1010-1041: a region of a single element takes xb[0]; otherwise xb is consumed cyclically
static void tensor_set(dsc_ctx *ctx, dsc_tensor *xa, const dsc_tensor *xb, int n_slices, const dsc_slice *slices) {
    const dsc_region r = region_of(xa, n_slices, slices);
    dsc_launch_region_copy(xb->data, xa->data, (int) dsc_dtype_size(xa->dtype), r, true, xb->ne, ctx->stream);
}

// dsc.cpp:1043-1106
extern "C" void dsc_tensor_set_idx(dsc_ctx *ctx, dsc_tensor *xa, const dsc_tensor *xb, int indexes, ...) {
    DSC_ASSERT(xa != nullptr);
    DSC_ASSERT(xb != nullptr);
    DSC_TRACE_OP(ctx, "idx;set", xa, xb);
    DSC_ASSERT((unsigned) indexes <= (unsigned) xa->n_dim);
    DSC_ASSERT(xa->dtype == xb->dtype);

    dsc_slice el_slices[DSC_MAX_DIMS];
    va_list args;
    va_start(args, indexes);
    for (int i = 0; i < indexes; ++i) {
        const int idx = va_arg(args, int);
        const int x_dim_i = xa->shape[tensor_dim(xa, i)];
        el_slices[i].start = idx;
        el_slices[i].stop = idx + 1;
        el_slices[i].step = 1;
        if (idx < 0) {
            el_slices[i].start += x_dim_i;
            el_slices[i].stop += x_dim_i;
        }
        DSC_ASSERT(el_slices[i].start >= 0 && el_slices[i].start < x_dim_i);    // unchecked in the reference (host UB)
    }
    va_end(args);

    // dsc.cpp:1071-1088 compares xb with the LEADING dims of xa (`i - indexes` where the trailing ones were meant);
    // the check is kept as written so that the same calls are accepted and refused.
    int xa_sub_shape[DSC_MAX_DIMS];
    for (int i = indexes; i < xa->n_dim; ++i) xa_sub_shape[i - indexes] = xa->shape[tensor_dim(xa, i - indexes)];
    const bool xb_scalar = xb->n_dim == 1 && xb->shape[DSC_MAX_DIMS - 1] == 1;
    const int xa_sub_ndim = xa->n_dim - indexes;
    if (xa_sub_ndim == 0) DSC_ASSERT(xb_scalar);
    if (!xb_scalar) {
        DSC_ASSERT(xb->n_dim == xa_sub_ndim);
        for (int i = 0; i < xa_sub_ndim; ++i) DSC_ASSERT(xa_sub_shape[i] == xb->shape[tensor_dim(xb, i)]);
    }
    tensor_set(ctx, xa, xb, indexes, el_slices);
}

// dsc.cpp:1108-1169
extern "C" void dsc_tensor_set_slice(dsc_ctx *ctx, dsc_tensor *xa, const dsc_tensor *xb, int slices, ...) {
    DSC_ASSERT(xa != nullptr);
    DSC_ASSERT(xb != nullptr);
    DSC_TRACE_OP(ctx, "slice;set", xa, xb);
    DSC_ASSERT((unsigned) slices <= (unsigned) xa->n_dim);
    DSC_ASSERT(xa->dtype == xb->dtype);

    dsc_slice el_slices[DSC_MAX_DIMS];
    va_list args;
    va_start(args, slices);
    parse_slices(xa, el_slices, nullptr, slices, args);
    va_end(args);

    int xa_slice_shape[DSC_MAX_DIMS];
    for (int i = 0; i < xa->n_dim; ++i)
        xa_slice_shape[i] = i < slices ? slice_count(el_slices[i]) : xa->shape[tensor_dim(xa, i)];

    const bool xb_scalar = xb->n_dim == 1 && xb->shape[DSC_MAX_DIMS - 1] == 1;
    if (!xb_scalar) {                                // dsc.cpp:1138-1146
        const int dims_to_compare = xa->n_dim < xb->n_dim ? xa->n_dim : xb->n_dim;
        for (int i = 0; i < dims_to_compare; ++i) {
            const int xb_dim_i = xb->shape[tensor_dim(xb, i)];
            const int xa_slice_i = xa_slice_shape[i];
            DSC_ASSERT(xa_slice_i == 1 || xb_dim_i == 1 || xa_slice_i == xb_dim_i);
        }
    }
    tensor_set(ctx, xa, xb, slices, el_slices);
}

// ------------------------------------------------------------------------------------------------
// dsc.cpp:764-827
extern "C" dsc_tensor *dsc_transpose(dsc_ctx *ctx, const dsc_tensor *x, int axes, ...) {
    DSC_ASSERT(x != nullptr);
    DSC_TRACE_OP(ctx, "op;transpose", x);
    if (x->n_dim == 1) return dsc_view(ctx, x);

    int swap_axes[DSC_MAX_DIMS];
    if (axes == 0) {
        for (int i = 0; i < x->n_dim; ++i) swap_axes[i] = x->n_dim - (i + 1);
    } else {
        DSC_ASSERT(axes == x->n_dim);
        va_list args;
        va_start(args, axes);
        for (int i = 0; i < axes; ++i) {
            const int el = va_arg(args, int);
            DSC_ASSERT((unsigned) el < DSC_MAX_DIMS);
            DSC_ASSERT(el < x->n_dim);                  // the reference indexes past its arrays here
            swap_axes[i] = el;
        }
        va_end(args);
    }

    int swapped_shape[DSC_MAX_DIMS], swapped_stride[DSC_MAX_DIMS];
    for (int i = 0; i < DSC_MAX_DIMS - x->n_dim; ++i) {
        swapped_shape[i] = x->shape[i];
        swapped_stride[i] = x->stride[i];
    }
    for (int i = 0; i < x->n_dim; ++i) {
        const int idx = tensor_dim(x, swap_axes[i]);
        swapped_shape[tensor_dim(x, i)] = x->shape[idx];
        swapped_stride[tensor_dim(x, i)] = x->stride[idx];
    }
    dsc_tensor *out = dsc_new_tensor(ctx, x->n_dim, &swapped_shape[tensor_dim(x, 0)], x->dtype, nullptr);
    const int esz = (int) dsc_dtype_size(x->dtype);

    // leading axes in place and the last two swapped: tiled transpose, coalesced on both sides
    bool last2 = swap_axes[x->n_dim - 1] == x->n_dim - 2 && swap_axes[x->n_dim - 2] == x->n_dim - 1;
    for (int i = 0; i < x->n_dim - 2; ++i) last2 = last2 && swap_axes[i] == i;
    if (last2) {
        const int rows = x->shape[DSC_MAX_DIMS - 2], cols = x->shape[DSC_MAX_DIMS - 1];
        dsc_launch_transpose_last2(x->data, out->data, esz, (long long) x->ne / ((long long) rows * cols), rows, cols, ctx->stream);
        return out;
    }
    // copy_with_stride (dsc.cpp:748-762): out is walked densely, x through the permuted strides
    dsc_region r;
    r.base = 0;
    r.ne = out->ne;
    for (int d = 0; d < DSC_MAX_DIMS; ++d) { r.count[d] = swapped_shape[d]; r.stride[d] = swapped_stride[d]; }
    dsc_launch_region_copy(x->data, out->data, esz, r, false, out->ne, ctx->stream);
    return out;
}

// dsc.cpp:2262-2340: the values are computed on the host in T with the reference's own expressions and copied once
template<typename T>
static void fill_fftfreq(T *v, int n, T d) {
    const T factor = 1 / (n * d);
    const int odd = n & 1;
    const int n2 = odd ? ((n - 1) >> 1) : (n >> 1);
    for (int i = 0; i < (n2 + odd); ++i) v[i] = i * factor;
    for (int i = 0; i < n2; ++i) v[(n2 + odd) + i] = (-n2 + i) * factor;
}
template<typename T>
static void fill_rfftfreq(T *v, int count, int n, T d) {
    const T factor = 1 / (n * d);
    for (int i = 0; i < count; ++i) v[i] = i * factor;
}

static dsc_tensor *freq_entry(dsc_ctx *ctx, int n, double d, dsc_dtype dtype, bool real_bins) {
    DSC_ASSERT(n > 0);
    if (dtype != DSC_F32 && dtype != DSC_F64) DSC_LOG_FATAL("dtype must be real");
    const int count = real_bins ? ((n & 1) ? (((n - 1) >> 1) + 1) : ((n >> 1) + 1)) : n;
    dsc_tensor *out = dsc_tensor_1d(ctx, dtype, count);
    const size_t bytes = (size_t) count * dsc_dtype_size(dtype);
    void *host = malloc(bytes);
    DSC_ASSERT(host != nullptr);
    if (dtype == DSC_F32) {
        if (real_bins) fill_rfftfreq((float *) host, count, n, (float) d);
        else           fill_fftfreq((float *) host, n, (float) d);
    } else {
        if (real_bins) fill_rfftfreq((double *) host, count, n, d);
        else           fill_fftfreq((double *) host, n, d);
    }
    HIP_CHECK(hipMemcpyAsync(out->data, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    free(host);
    return out;
}

extern "C" dsc_tensor *dsc_fftfreq(dsc_ctx *ctx, int n, double d, dsc_dtype dtype) { return freq_entry(ctx, n, d, dtype, false); }
extern "C" dsc_tensor *dsc_rfftfreq(dsc_ctx *ctx, int n, double d, dsc_dtype dtype) { return freq_entry(ctx, n, d, dtype, true); }
