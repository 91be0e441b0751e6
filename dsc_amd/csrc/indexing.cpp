// indexing.cpp — dsc_tensor_get_idx / get_slice / set_idx / set_slice, dsc_transpose, dsc_fftfreq / dsc_rfftfreq on device
// tensors.
//
// Behavioural contract: dsc/src/dsc.cpp:764-1169 and :2262-2340 — which selections are accepted, what they select, the
// shape of the result, what is refused.  Structure here: every entry point turns its variadic arguments into a list of
// per-axis picks (`pick`: first element, count, step), `choose()` folds the picks into ONE strided region of the tensor
// (dsc_region), and a single gather / scatter launch moves it (elementwise.hip) — where the reference walks the selection
// element by element with dsc_slice_iterator (dsc_iter.h:125-190).  The README's `[:output_length]` crop after dsc_irfft
// and the placement of a block into a zero-padded buffer (README.md:113-135) therefore never leave HBM.
#include "dsc_internal.h"
#include "kernels.h"

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// ---- one axis of a selection -------------------------------------------------------------------

struct pick {
    int first = 0;        // first selected element of the axis
    int count = 1;        // how many
    int step = 1;         // distance between them (negative: walking down)
    bool single = false;  // written as a plain index: the axis disappears from a get_slice result
};

struct check { bool holds; const char *rule; };

static void enforce(const check *checks, size_t n, const char *who) {
    for (size_t i = 0; i < n; ++i) {
        if (!checks[i].holds) {
            fprintf(stderr, "%s: %s\n", who, checks[i].rule);
            exit(EXIT_FAILURE);
        }
    }
}

static const dsc_tensor *given(const dsc_tensor *t, const char *who) {
    const check rules[] = {{t != nullptr, "tensor argument is NULL"}};
    enforce(rules, 1, who);
    return t;
}

static inline int wrap(int i, int extent) { return i < 0 ? i + extent : i; }

// A plain (possibly negative) index into an axis of `extent` elements.
static pick pick_index(int index, int extent, const char *who) {
    pick p;
    p.first = wrap(index, extent);
    p.single = true;
    const check rules[] = {{p.first >= 0 && p.first < extent, "index out of range for its axis"}};
    enforce(rules, 1, who);
    return p;
}

// A (start, stop, step) triple with DSC_VALUE_NONE for "not given" (dsc.h:110-117), NumPy conventions: a missing start /
// stop means "from the end the step walks away from"; negative values count from the back.  The wrapper encodes a plain
// index k inside a slice list as start = stop = step = k (dsc_api.h:16, python/dsc/tensor.py:205-216).
static pick pick_slice(dsc_slice s, int extent, const char *who) {
    if (s.start != DSC_VALUE_NONE && s.start == s.stop && s.start == s.step) return pick_index(s.start, extent, who);

    const int step = s.step == DSC_VALUE_NONE ? 1 : s.step;
    const bool up = step > 0;
    const int from = wrap(s.start == DSC_VALUE_NONE ? (up ? 0 : extent - 1) : s.start, extent);
    const int to = s.stop == DSC_VALUE_NONE ? (up ? extent : -1) : wrap(s.stop, extent);      // -1: one before the first element
    const int span = abs(to - from), stride = abs(step);

    pick p;
    p.first = from;
    p.step = step;
    p.count = stride > 0 ? (span + stride - 1) / stride : 0;
    const long long last = (long long) from + (long long) (p.count - 1) * step;
    const check rules[] = {
        {step != 0, "slice step must not be zero"},
        {span <= extent, "slice spans more than its axis"},
        {up ? from < to : from > to, "slice is empty or runs against its step"},
        {stride <= extent, "slice step larger than its axis"},
        // the reference's iterator would read outside the tensor for these (x[5:14] on an axis of 10, x[5:3:-1] on one of
        // 4: both pass its assertions); on the device that is a fault, so they are refused
        {from >= 0 && from < extent, "slice starts outside its axis"},
        {last >= 0 && last < extent, "slice ends outside its axis"},
    };
    enforce(rules, sizeof(rules) / sizeof(rules[0]), who);
    return p;
}

// ---- a whole selection -------------------------------------------------------------------------

// The picks (one per leading axis; the remaining axes are taken whole) folded into one strided region of x, in the
// right-aligned 4-slot layout of dsc_tensor, plus the shape the selection has as a tensor of its own.
struct chosen {
    dsc_region region;
    int shape[DSC_MAX_DIMS];      // extent per axis of x (singles included, as 1)
    int kept_shape[DSC_MAX_DIMS]; // the same without the single-index axes
    int kept = 0;
};

static chosen choose(const dsc_tensor *x, const std::vector<pick> &picks) {
    chosen c;
    c.region.base = 0;
    c.region.ne = 1;
    for (int slot = 0; slot < DSC_MAX_DIMS; ++slot) { c.region.count[slot] = 1; c.region.stride[slot] = 0; }
    for (int axis = 0; axis < x->n_dim; ++axis) {
        const int slot = dsc_axis_slot(x, axis);
        pick p;
        if (axis < (int) picks.size()) p = picks[axis];
        else p.count = x->shape[slot];
        c.region.base += (long long) p.first * x->stride[slot];
        c.region.count[slot] = p.count;
        c.region.stride[slot] = (long long) p.step * x->stride[slot];
        c.region.ne *= p.count;
        c.shape[axis] = p.count;
        if (!p.single) c.kept_shape[c.kept++] = p.count;
    }
    return c;
}

static bool is_one_element(const dsc_tensor *t) { return t->n_dim == 1 && t->shape[DSC_MAX_DIMS - 1] == 1; }

static std::vector<pick> read_indexes(const dsc_tensor *x, int n, va_list args, const char *who) {
    std::vector<pick> picks;
    for (int axis = 0; axis < n; ++axis) picks.push_back(pick_index(va_arg(args, int), x->shape[dsc_axis_slot(x, axis)], who));
    return picks;
}

static std::vector<pick> read_slices(const dsc_tensor *x, int n, va_list args, const char *who) {
    std::vector<pick> picks;
    for (int axis = 0; axis < n; ++axis) picks.push_back(pick_slice(va_arg(args, dsc_slice), x->shape[dsc_axis_slot(x, axis)], who));
    return picks;
}

static void scatter(dsc_ctx *ctx, dsc_tensor *into, const dsc_tensor *values, const chosen &c) {
    // a selection of one element takes values[0]; otherwise `values` is consumed cyclically (dsc.cpp:1010-1041)
    dsc_launch_region_copy(values->data, into->data, (int) dsc_dtype_size(into->dtype), c.region, true, values->ne, ctx->stream);
}

}  // namespace

// x[i, j, ...] with plain indexes: the sub-tensor below them, or a one-element tensor when every axis is indexed (dsc.cpp:832-866)
extern "C" dsc_tensor *dsc_tensor_get_idx(dsc_ctx *ctx, const dsc_tensor *x, int indexes, ...) {
    given(x, __func__);
    DSC_TRACE_OP(ctx, "idx;get", x);
    const check args_ok[] = {{indexes >= 1 && indexes <= DSC_MAX_DIMS, "between 1 and 4 indexes"}, {indexes <= x->n_dim, "too many indexes"}};
    enforce(args_ok, 2, __func__);
    va_list args;
    va_start(args, indexes);
    const std::vector<pick> picks = read_indexes(x, indexes, args, __func__);
    va_end(args);

    const chosen c = choose(x, picks);
    static const int one[1] = {1};
    dsc_tensor *out = c.kept == 0 ? dsc_new_tensor(ctx, 1, one, x->dtype, nullptr) : dsc_new_tensor(ctx, c.kept, c.kept_shape, x->dtype, nullptr);
    // leading indexes of a contiguous tensor select one contiguous run
    const size_t esz = dsc_dtype_size(x->dtype);
    HIP_CHECK(hipMemcpyAsync(out->data, (const char *) x->data + c.region.base * (long long) esz, (size_t) c.region.ne * esz,
                             hipMemcpyDeviceToDevice, ctx->stream));
    return out;
}

// x[a:b:c, k, ...]: a copy of the selection; axes given as a single index are dropped from the result (dsc.cpp:935-992)
extern "C" dsc_tensor *dsc_tensor_get_slice(dsc_ctx *ctx, const dsc_tensor *x, int slices, ...) {
    given(x, __func__);
    DSC_TRACE_OP(ctx, "slice;get", x);
    const check args_ok[] = {{slices >= 0 && slices <= DSC_MAX_DIMS, "at most 4 slices"}, {slices <= x->n_dim, "too many slices"}};
    enforce(args_ok, 2, __func__);
    va_list args;
    va_start(args, slices);
    const std::vector<pick> picks = read_slices(x, slices, args, __func__);
    va_end(args);

    const chosen c = choose(x, picks);
    const check shape_ok[] = {{c.kept >= 1, "every axis given as a single index: use dsc_tensor_get_idx"}};
    enforce(shape_ok, 1, __func__);
    dsc_tensor *out = dsc_new_tensor(ctx, c.kept, c.kept_shape, x->dtype, nullptr);
    DSC_ASSERT(c.region.ne == out->ne);
    dsc_launch_region_copy(x->data, out->data, (int) dsc_dtype_size(x->dtype), c.region, false, out->ne, ctx->stream);
    return out;
}

// xa[i, j, ...] = xb (dsc.cpp:1043-1106)
extern "C" void dsc_tensor_set_idx(dsc_ctx *ctx, dsc_tensor *xa, const dsc_tensor *xb, int indexes, ...) {
    given(xa, __func__), given(xb, __func__);
    DSC_TRACE_OP(ctx, "idx;set", xa, xb);
    const check args_ok[] = {{indexes >= 0 && indexes <= xa->n_dim, "too many indexes"}, {xa->dtype == xb->dtype, "value and destination differ in dtype"}};
    enforce(args_ok, 2, __func__);
    va_list args;
    va_start(args, indexes);
    const std::vector<pick> picks = read_indexes(xa, indexes, args, __func__);
    va_end(args);

    // What the reference accepts as xb (dsc.cpp:1071-1088): one element always; otherwise a tensor of as many axes as xa has
    // left, whose extents equal xa's FIRST extents (not the remaining ones — the check is reproduced as the reference makes
    // it, so that the same calls are accepted and refused).
    const int left = xa->n_dim - indexes;
    if (!is_one_element(xb)) {
        std::vector<check> rules = {{left > 0, "a fully indexed element takes a single value"}, {xb->n_dim == left, "value has the wrong number of axes"}};
        for (int axis = 0; axis < left && axis < xb->n_dim; ++axis)
            rules.push_back({xa->shape[dsc_axis_slot(xa, axis)] == xb->shape[dsc_axis_slot(xb, axis)], "value extents do not match the destination"});
        enforce(rules.data(), rules.size(), __func__);
    }
    scatter(ctx, xa, xb, choose(xa, picks));
}

// xa[a:b:c, ...] = xb (dsc.cpp:1108-1169)
extern "C" void dsc_tensor_set_slice(dsc_ctx *ctx, dsc_tensor *xa, const dsc_tensor *xb, int slices, ...) {
    given(xa, __func__), given(xb, __func__);
    DSC_TRACE_OP(ctx, "slice;set", xa, xb);
    const check args_ok[] = {{slices >= 0 && slices <= xa->n_dim, "too many slices"}, {xa->dtype == xb->dtype, "value and destination differ in dtype"}};
    enforce(args_ok, 2, __func__);
    va_list args;
    va_start(args, slices);
    const std::vector<pick> picks = read_slices(xa, slices, args, __func__);
    va_end(args);

    const chosen c = choose(xa, picks);
    if (!is_one_element(xb)) {
        // axis by axis from the front, over the axes both have: equal, or one of them 1 (dsc.cpp:1138-1146)
        std::vector<check> rules;
        for (int axis = 0; axis < xa->n_dim && axis < xb->n_dim; ++axis) {
            const int have = xb->shape[dsc_axis_slot(xb, axis)], want = c.shape[axis];
            rules.push_back({want == 1 || have == 1 || want == have, "value extents cannot fill the selection"});
        }
        enforce(rules.data(), rules.size(), __func__);
    }
    scatter(ctx, xa, xb, c);
}

// ------------------------------------------------------------------------------------------------
// dsc_transpose(x)              reverse the axes
// dsc_transpose(x, p0, p1, ...) result axis i = x's axis p_i                      (dsc.cpp:764-827)
extern "C" dsc_tensor *dsc_transpose(dsc_ctx *ctx, const dsc_tensor *x, int axes, ...) {
    given(x, __func__);
    DSC_TRACE_OP(ctx, "op;transpose", x);
    const int nd = x->n_dim;
    if (nd == 1) return dsc_view(ctx, x);

    int perm[DSC_MAX_DIMS];
    for (int i = 0; i < nd; ++i) perm[i] = nd - 1 - i;
    if (axes != 0) {
        const check count_ok[] = {{axes == nd, "one axis number per axis of the tensor"}};
        enforce(count_ok, 1, __func__);
        va_list args;
        va_start(args, axes);
        for (int i = 0; i < nd; ++i) {
            perm[i] = va_arg(args, int);
            const check axis_ok[] = {{perm[i] >= 0 && perm[i] < nd, "axis number out of range"}};      // the reference indexes past its arrays here
            enforce(axis_ok, 1, __func__);
        }
        va_end(args);
    }

    // the result is dense; its element (i0, i1, ...) is x's element with the indexes routed through perm: walk the result in
    // order and read x through permuted strides
    dsc_region src;
    src.base = 0;
    int out_shape[DSC_MAX_DIMS];
    for (int slot = 0; slot < DSC_MAX_DIMS; ++slot) { src.count[slot] = 1; src.stride[slot] = 0; }
    for (int i = 0; i < nd; ++i) {
        const int from = dsc_axis_slot(x, perm[i]), to = dsc_axis_slot(x, i);
        out_shape[i] = x->shape[from];
        src.count[to] = x->shape[from];
        src.stride[to] = x->stride[from];
    }
    dsc_tensor *out = dsc_new_tensor(ctx, nd, out_shape, x->dtype, nullptr);
    src.ne = out->ne;
    const int esz = (int) dsc_dtype_size(x->dtype);

    // only the last two axes trade places: LDS-tiled transpose, coalesced on both sides
    bool tail_swap = perm[nd - 1] == nd - 2 && perm[nd - 2] == nd - 1;
    for (int i = 0; i + 2 < nd; ++i) tail_swap = tail_swap && perm[i] == i;
    if (tail_swap) {
        const int rows = x->shape[DSC_MAX_DIMS - 2], cols = x->shape[DSC_MAX_DIMS - 1];
        dsc_launch_transpose_last2(x->data, out->data, esz, (long long) x->ne / ((long long) rows * cols), rows, cols, ctx->stream);
    } else {
        // the last axis moves: tile the plane it spans with the axis that takes its place (both sides coalesced)
        bool tiled = false;
        if (perm[nd - 1] != nd - 1) {
            int shape[DSC_MAX_DIMS], stride[DSC_MAX_DIMS];
            for (int i = 0; i < nd; ++i) { shape[i] = x->shape[dsc_axis_slot(x, i)]; stride[i] = x->stride[dsc_axis_slot(x, i)]; }
            bool distinct = true;                                        // a repeated axis number is not a permutation: the strided copy handles it
            for (int i = 0; i < nd; ++i) for (int k = i + 1; k < nd; ++k) distinct = distinct && perm[i] != perm[k];
            if (distinct) tiled = dsc_launch_transpose_moving_last(x->data, out->data, esz, nd, shape, stride, perm, ctx->stream);
        }
        if (!tiled) dsc_launch_region_copy(x->data, out->data, esz, src, false, out->ne, ctx->stream);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// Sample frequencies of an n-point transform with spacing d (dsc.cpp:2262-2340), evaluated on the host in the output
// precision — bin k maps to the signed bin number k (k < ceil(n / 2)) or k - n, times 1 / (n d) — and copied once.
namespace {

template<typename T>
void frequencies(T *f, int count, int n, T d, bool one_sided) {
    const T unit = 1 / (n * d);
    const int positive = one_sided ? count : n - n / 2;               // ceil(n / 2) non-negative bins in the two-sided layout
    for (int k = 0; k < count; ++k) f[k] = (k < positive ? k : k - n) * unit;
}

dsc_tensor *frequency_tensor(dsc_ctx *ctx, int n, double d, dsc_dtype dtype, bool one_sided) {
    const check rules[] = {{n > 0, "transform length must be positive"}, {dtype == DSC_F32 || dtype == DSC_F64, "dtype must be real"}};
    enforce(rules, 2, one_sided ? "dsc_rfftfreq" : "dsc_fftfreq");
    const int count = one_sided ? n / 2 + 1 : n;
    dsc_tensor *out = dsc_tensor_1d(ctx, dtype, count);
    std::vector<double> host((size_t) count);                         // 8 bytes per value holds either precision
    if (dtype == DSC_F32) frequencies((float *) host.data(), count, n, (float) d, one_sided);
    else                  frequencies(host.data(), count, n, d, one_sided);
    dsc_copy_from_host(ctx, out, host.data(), (size_t) count * dsc_dtype_size(dtype));
    return out;
}

}  // namespace

extern "C" dsc_tensor *dsc_fftfreq(dsc_ctx *ctx, int n, double d, dsc_dtype dtype) { return frequency_tensor(ctx, n, d, dtype, false); }
extern "C" dsc_tensor *dsc_rfftfreq(dsc_ctx *ctx, int n, double d, dsc_dtype dtype) { return frequency_tensor(ctx, n, d, dtype, true); }
