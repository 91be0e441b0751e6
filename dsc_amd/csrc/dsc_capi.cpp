// dsc_capi.cpp — context, HBM arenas, tensors, copies, cast / mul / reductions of the C ABI
// declared in include/dsc_mi355x.h.  FFT entry points live in fft_driver.cpp.
//
// Host-side mirror of dsc/src/dsc.cpp:136-470 (context + tensor creation), :44-115
// (parameter validation), :1273-1284 (dsc_mul) and :1793-1953 (reductions); arithmetic is in
// the .hip files.  Error behaviour is the reference's: print and exit (dsc.h:14-28).
#include "dsc_internal.h"
#include "kernels.h"

#include <cstring>

static int g_device = 0;

// ------------------------------------------------------------------------------ context

extern "C" int dsc_set_device(int device) {
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    if (count == 0) DSC_LOG_FATAL("no HIP device visible: this backend has no CPU fallback");
    DSC_ASSERT(device >= 0 && device < count);
    g_device = device;
    return count;
}

extern "C" dsc_ctx *dsc_ctx_init(size_t main_mem, size_t scratch_mem) {
    DSC_ASSERT(main_mem > 0);
    DSC_ASSERT(scratch_mem > 0);

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        DSC_LOG_FATAL("no HIP device visible: this backend has no CPU fallback");
    HIP_CHECK(hipSetDevice(g_device));

    dsc_ctx *ctx = new dsc_ctx();
    ctx->device = g_device;
    HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreate(&ctx->ev_start));
    HIP_CHECK(hipEventCreate(&ctx->ev_stop));

    main_mem = DSC_ALIGN_UP(main_mem, DSC_DEVICE_ALIGN);
    scratch_mem = DSC_ALIGN_UP(scratch_mem, DSC_DEVICE_ALIGN);
    HIP_CHECK(hipMalloc((void **) &ctx->main_buf, main_mem));
    HIP_CHECK(hipMalloc((void **) &ctx->scratch_buf, scratch_mem));
    ctx->main.init(ctx->main_buf, main_mem);
    ctx->scratch.init(ctx->scratch_buf, scratch_mem);
    memset(ctx->fft_plans, 0, sizeof(ctx->fft_plans));
    ctx->last_fft_path = "none";

    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, g_device));
    ctx->n_cu = prop.multiProcessorCount;

    DSC_LOG_INFO("created new context %p with %ldMB for main and %ldMB for scratch memory on %s (device %d: %s, %d CUs)",
                 (void *) ctx, (long) (main_mem >> 20), (long) (scratch_mem >> 20), "MI355X", g_device,
                 prop.gcnArchName, ctx->n_cu);
    return ctx;
}

// A header the caller has seen: out of circulation until DSC_HEADER_QUARANTINE younger ones have been retired.
static void retire_header(dsc_ctx *ctx, dsc_tensor *t) {
    t->buffer = nullptr;
    t->data = nullptr;
    ctx->quarantine.push_back(t);
    if (ctx->quarantine.size() > DSC_HEADER_QUARANTINE) {
        ctx->tensor_pool.push_back(ctx->quarantine.front());
        ctx->quarantine.pop_front();
    }
}

void dsc_stream_sync(dsc_ctx *ctx) {
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->async_error != nullptr && *ctx->async_error != 0) {
        const unsigned e = *ctx->async_error;
        *ctx->async_error = 0;
        DSC_LOG_FATAL("a team barrier of the fused L2 transform did not complete (code %u): the launch was not fully resident", e);
    }
}

static void release_headers(dsc_ctx *ctx) {
    for (dsc_tensor *t : ctx->live_tensors) {
        dsc_buffer_rec *rec = (dsc_buffer_rec *) t->buffer;
        if (rec != nullptr && --rec->pub.refs == 0) delete rec;
        retire_header(ctx, t);
    }
    ctx->live_tensors.clear();
}

extern "C" void dsc_ctx_free(dsc_ctx *ctx) {
    if (ctx == nullptr) return;
    HIP_CHECK(hipSetDevice(ctx->device));
    dsc_stream_sync(ctx);
    dsc_trace_release(ctx);
    for (hipEvent_t e : ctx->tracer.free_events) HIP_CHECK(hipEventDestroy(e));
    if (ctx->tracer.based) HIP_CHECK(hipEventDestroy(ctx->tracer.base_ev));
    DSC_LOG_INFO("freeing context %p: main mem %ldMB, scratch mem %ldMB", (void *) ctx,
                 (long) (ctx->main.capacity() >> 20), (long) (ctx->scratch.capacity() >> 20));
    release_headers(ctx);
    dsc_peer_release(ctx);
    if (ctx->async_error != nullptr) HIP_CHECK(hipHostFree(ctx->async_error));
    for (dsc_tensor *t : ctx->tensor_pool) delete t;
    for (dsc_tensor *t : ctx->quarantine) delete t;
    for (auto &plan : ctx->fft_plans) { delete plan; plan = nullptr; }
    HIP_CHECK(hipFree(ctx->main_buf));
    HIP_CHECK(hipFree(ctx->scratch_buf));
    HIP_CHECK(hipEventDestroy(ctx->ev_start));
    HIP_CHECK(hipEventDestroy(ctx->ev_stop));
    HIP_CHECK(hipStreamDestroy(ctx->stream));
    delete ctx;
}

// dsc.cpp:287-291: both arenas are reset; every tensor and plan handed out so far is dead.
extern "C" void dsc_ctx_clear(dsc_ctx *ctx) {
    release_headers(ctx);
    for (auto &plan : ctx->fft_plans) { delete plan; plan = nullptr; }
    ctx->main.clear();
    ctx->scratch.reset();
}

extern "C" size_t dsc_used_mem(dsc_ctx *ctx) { return ctx->main.used(); }

extern "C" void dsc_print_mem_usage(dsc_ctx *ctx) {
    const size_t used = ctx->main.used(), total = ctx->main.capacity();
    DSC_LOG_INFO("main memory (%s) usage: %ld/%ld MB (%.1f%%)", "MI355X", (long) (used >> 20), (long) (total >> 20),
                 (double) used / (double) total * 1e2);
}

extern "C" void dsc_synchronize(dsc_ctx *ctx) { dsc_stream_sync(ctx); }

extern "C" void *dsc_stream(dsc_ctx *ctx) { return (void *) ctx->stream; }

extern "C" void dsc_timer_start(dsc_ctx *ctx) { HIP_CHECK(hipEventRecord(ctx->ev_start, ctx->stream)); }

extern "C" float dsc_timer_stop(dsc_ctx *ctx) {
    float ms = 0.f;
    HIP_CHECK(hipEventRecord(ctx->ev_stop, ctx->stream));
    HIP_CHECK(hipEventSynchronize(ctx->ev_stop));
    HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    return ms;
}

extern "C" const char *dsc_last_fft_path(dsc_ctx *ctx) { return ctx->last_fft_path; }

// ------------------------------------------------------------------------------ tensors

// dsc.cpp:342-397.  The header is host memory, the payload an arena block (main arena, or
// the scratch arena for operator temporaries: the reference's DSC_CTX_PUSH, dsc.cpp:31-36).
dsc_tensor *dsc_new_tensor_in(dsc_ctx *ctx, int n_dim, const int *shape, dsc_dtype dtype,
                              dsc_tensor_buffer *buffer, bool in_scratch) {
    DSC_ASSERT((unsigned) n_dim <= DSC_MAX_DIMS);
    DSC_ASSERT(dtype < 4);

    long long ne = 1;
    for (int i = 0; i < n_dim; ++i) {
        DSC_ASSERT(shape[i] > 0);
        ne *= shape[i];
    }
    DSC_ASSERT(ne <= 0x7fffffffLL);               // `int ne` in the ABI (dsc.h:104)

    dsc_tensor *t;
    if (!ctx->tensor_pool.empty()) { t = ctx->tensor_pool.back(); ctx->tensor_pool.pop_back(); }
    else                           { t = new dsc_tensor(); }

    dsc_buffer_rec *rec;
    if (buffer == nullptr) {
        rec = new dsc_buffer_rec();
        rec->pub.refs = 0;
        rec->nbytes = (size_t) ne * dsc_dtype_size(dtype);
        if (in_scratch) { rec->dev = nullptr; t->data = ctx->scratch.alloc(rec->nbytes); }
        else            { rec->dev = ctx->main.alloc(rec->nbytes); t->data = rec->dev; }
    } else {
        rec = (dsc_buffer_rec *) buffer;
        DSC_ASSERT(rec->dev != nullptr);
        DSC_ASSERT((size_t) ne * dsc_dtype_size(dtype) <= DSC_ALIGN_UP(rec->nbytes, DSC_DEVICE_ALIGN));
        t->data = rec->dev;
    }
    rec->pub.refs++;
    t->buffer = &rec->pub;
    t->dtype = dtype;
    t->ne = (int) ne;
    t->n_dim = n_dim;
    t->backend = DSC_BACKEND_MI355X;
    for (int i = 0; i < DSC_MAX_DIMS; ++i)
        t->shape[i] = i < (DSC_MAX_DIMS - n_dim) ? 1 : shape[i - (DSC_MAX_DIMS - n_dim)];
    t->stride[DSC_MAX_DIMS - 1] = 1;
    for (int i = DSC_MAX_DIMS - 2; i >= 0; --i) t->stride[i] = t->stride[i + 1] * t->shape[i + 1];

    ctx->live_tensors.insert(t);
    return t;
}

extern "C" dsc_tensor *dsc_new_tensor(dsc_ctx *ctx, int n_dim, const int *shape, dsc_dtype dtype, dsc_tensor_buffer *buffer) {
    return dsc_new_tensor_in(ctx, n_dim, shape, dtype, buffer, false);
}

// dsc.cpp:399-401 (dsc_new_view, dsc.h:83)
extern "C" dsc_tensor *dsc_view(dsc_ctx *ctx, const dsc_tensor *x) {
    DSC_ASSERT(x != nullptr);
    return dsc_new_tensor(ctx, x->n_dim, &x->shape[DSC_MAX_DIMS - x->n_dim], x->dtype, x->buffer);
}

extern "C" dsc_tensor *dsc_tensor_1d(dsc_ctx *ctx, dsc_dtype dtype, int d1) {
    const int shape[4] = {d1};
    return dsc_new_tensor(ctx, 1, shape, dtype, nullptr);
}
extern "C" dsc_tensor *dsc_tensor_2d(dsc_ctx *ctx, dsc_dtype dtype, int d1, int d2) {
    const int shape[4] = {d1, d2};
    return dsc_new_tensor(ctx, 2, shape, dtype, nullptr);
}
extern "C" dsc_tensor *dsc_tensor_3d(dsc_ctx *ctx, dsc_dtype dtype, int d1, int d2, int d3) {
    const int shape[4] = {d1, d2, d3};
    return dsc_new_tensor(ctx, 3, shape, dtype, nullptr);
}
extern "C" dsc_tensor *dsc_tensor_4d(dsc_ctx *ctx, dsc_dtype dtype, int d1, int d2, int d3, int d4) {
    const int shape[4] = {d1, d2, d3, d4};
    return dsc_new_tensor(ctx, 4, shape, dtype, nullptr);
}

// dsc.cpp:293-303.  Python's __del__ may free twice or after dsc_ctx_clear: those find the header retired (not live)
// and are ignored; the header's address is not handed out again while such stale handles are plausible (retire_header).
extern "C" void dsc_tensor_free(dsc_ctx *ctx, dsc_tensor *x) {
    if (x == nullptr || ctx == nullptr) return;
    auto it = ctx->live_tensors.find(x);
    if (it == ctx->live_tensors.end()) return;
    ctx->live_tensors.erase(it);
    dsc_buffer_rec *rec = (dsc_buffer_rec *) x->buffer;
    if (--rec->pub.refs == 0) {
        if (!rec->external) ctx->main.free(rec->dev);
        delete rec;
    }
    retire_header(ctx, x);
}

// dsc_tensor_from_device_ptr: an ordinary header whose buffer record points at memory the caller owns.
dsc_tensor *dsc_new_tensor_over(dsc_ctx *ctx, void *ptr, size_t nbytes, int n_dim, const int *shape, dsc_dtype dtype) {
    DSC_ASSERT(n_dim > 0 && n_dim <= DSC_MAX_DIMS);
    DSC_ASSERT(dtype < 4);
    long long ne = 1;
    for (int i = 0; i < n_dim; ++i) {
        DSC_ASSERT(shape[i] > 0);
        ne *= shape[i];
    }
    DSC_ASSERT(ne <= 0x7fffffffLL);
    DSC_ASSERT((size_t) ne * dsc_dtype_size(dtype) <= nbytes);
    dsc_buffer_rec *rec = new dsc_buffer_rec();
    rec->pub.refs = 0;
    rec->dev = (char *) ptr;
    rec->nbytes = nbytes;
    rec->external = true;
    return dsc_new_tensor_in(ctx, n_dim, shape, dtype, &rec->pub, false);
}

extern "C" void dsc_copy_from_host(dsc_ctx *ctx, dsc_tensor *dst, const void *src, size_t nbytes) {
    DSC_ASSERT(dst != nullptr && src != nullptr);
    DSC_ASSERT(nbytes <= (size_t) dst->ne * dsc_dtype_size(dst->dtype));
    HIP_CHECK(hipMemcpyAsync(dst->data, src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    dsc_stream_sync(ctx);       // src may be pageable and reused by the caller
}

extern "C" void dsc_copy_to_host(dsc_ctx *ctx, const dsc_tensor *src, void *dst, size_t nbytes) {
    DSC_ASSERT(dst != nullptr && src != nullptr);
    DSC_ASSERT(nbytes <= (size_t) src->ne * dsc_dtype_size(src->dtype));
    HIP_CHECK(hipMemcpyAsync(dst, src->data, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    dsc_stream_sync(ctx);
}

// dsc.cpp:449-479: scalars as one-element 1-D tensors
template<typename T>
static dsc_tensor *wrap_value(dsc_ctx *ctx, dsc_dtype dtype, T val) {
    dsc_tensor *out = dsc_tensor_1d(ctx, dtype, 1);
    dsc_copy_from_host(ctx, out, &val, sizeof(T));
    return out;
}
extern "C" dsc_tensor *dsc_wrap_f32(dsc_ctx *ctx, float val)   { return wrap_value(ctx, DSC_F32, val); }
extern "C" dsc_tensor *dsc_wrap_f64(dsc_ctx *ctx, double val)  { return wrap_value(ctx, DSC_F64, val); }
extern "C" dsc_tensor *dsc_wrap_c32(dsc_ctx *ctx, dsc_c32 val) { return wrap_value(ctx, DSC_C32, val); }
extern "C" dsc_tensor *dsc_wrap_c64(dsc_ctx *ctx, dsc_c64 val) { return wrap_value(ctx, DSC_C64, val); }

// ------------------------------------------------------------------------------ cast

static dsc_tensor *cast_into(dsc_ctx *ctx, dsc_tensor *x, dsc_dtype new_dtype, bool in_scratch) {
    if (x->dtype == new_dtype) return x;
    dsc_tensor *out = dsc_new_tensor_in(ctx, x->n_dim, &x->shape[DSC_MAX_DIMS - x->n_dim], new_dtype, nullptr, in_scratch);
    dsc_launch_cast(x->data, x->dtype, out->data, new_dtype, x->ne, ctx->stream);
    return out;
}

// dsc.cpp:587-597
extern "C" dsc_tensor *dsc_cast(dsc_ctx *ctx, dsc_tensor *x, dsc_dtype new_dtype) {
    DSC_ASSERT(x != nullptr);
    DSC_TRACE_OP(ctx, "op;cast", x, nullptr, (int) new_dtype, 0);
    DSC_ASSERT(new_dtype < 4);
    return cast_into(ctx, x, new_dtype, false);
}

// A scratch-arena temporary is only a header + bump allocation: drop the header again.
static void drop_scratch_tensor(dsc_ctx *ctx, dsc_tensor *t) {
    ctx->live_tensors.erase(t);
    delete (dsc_buffer_rec *) t->buffer;
    ctx->tensor_pool.push_back(t);
}

// ------------------------------------------------------------------------------ mul

static const dsc_dtype k_promote[4][4] = {      // dsc_dtype.h:73-78
    {DSC_F32, DSC_F64, DSC_C32, DSC_C64},
    {DSC_F64, DSC_F64, DSC_C32, DSC_C64},
    {DSC_C32, DSC_C32, DSC_C32, DSC_C64},
    {DSC_C64, DSC_C64, DSC_C64, DSC_C64},
};

// dsc.cpp:44-69 (validate_binary_params) + :1186-1223 (binary_op)
static dsc_tensor *binary_entry(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out, int op) {
    DSC_ASSERT(xa != nullptr);
    DSC_ASSERT(xb != nullptr);
    static const char *names[] = {"dsc_add", "dsc_sub", "dsc_mul", "dsc_div"};
    dsc_trace_scope trace__(ctx, names[op], "op;binary", xa, xb);
    int shape[DSC_MAX_DIMS];
    for (int i = 0; i < DSC_MAX_DIMS; ++i) {       // can_broadcast, dsc.cpp:1174-1184
        DSC_ASSERT(xa->shape[i] == xb->shape[i] || xa->shape[i] == 1 || xb->shape[i] == 1);
        shape[i] = xa->shape[i] > xb->shape[i] ? xa->shape[i] : xb->shape[i];
    }
    const int n_dim = xa->n_dim > xb->n_dim ? xa->n_dim : xb->n_dim;
    const dsc_dtype out_dtype = k_promote[xa->dtype][xb->dtype];

    if (out == nullptr) {
        out = dsc_new_tensor(ctx, n_dim, &shape[DSC_MAX_DIMS - n_dim], out_dtype, nullptr);
    } else {
        DSC_ASSERT(out->dtype == out_dtype);
        DSC_ASSERT(out->n_dim == n_dim);
        DSC_ASSERT(memcmp(out->shape, shape, sizeof(shape)) == 0);
    }

    if (xa->dtype != xb->dtype && xa->ne == out->ne && xb->ne == out->ne &&
        dsc_launch_binary_mixed(xa->data, xa->dtype, xb->data, xb->dtype, out->data, op, out->ne, ctx->stream))
        return out;                                 // equal shapes, different dtypes: promoted in registers

    ctx->scratch.reset();                           // DSC_CTX_PUSH
    dsc_tensor *ca = cast_into(ctx, xa, out_dtype, true);
    dsc_tensor *cb = cast_into(ctx, xb, out_dtype, true);

    dsc_bcast_args g;
    g.ne = out->ne;
    g.a_scalar = xa->n_dim == 1 && xa->shape[DSC_MAX_DIMS - 1] == 1;
    g.b_scalar = !g.a_scalar && xb->n_dim == 1 && xb->shape[DSC_MAX_DIMS - 1] == 1;
    for (int i = 0; i < DSC_MAX_DIMS; ++i) {
        g.out_shape[i] = shape[i];
        g.a_stride[i] = xa->shape[i] < shape[i] ? 0 : xa->stride[i];     // dsc_iter.h:71-73
        g.b_stride[i] = xb->shape[i] < shape[i] ? 0 : xb->stride[i];
    }
    // index fast paths: equal shapes, or one operand spanning exactly the trailing dims of the output
    // (e.g. a [B, K] spectrum times a [K] filter): no per-element index decomposition
    auto trailing = [&](const dsc_tensor *small) {
        bool seen_full = false;                       // shape must be 1,..,1,d_k,..,d_3 with the d's equal to the output's
        for (int i = 0; i < DSC_MAX_DIMS; ++i) {
            if (small->shape[i] == shape[i] && (shape[i] != 1 || seen_full)) seen_full = true;
            else if (small->shape[i] == 1 && !seen_full) continue;
            else if (small->shape[i] != shape[i]) return false;
        }
        return true;
    };
    // ... or exactly the LEADING dims (a [B, K] tensor scaled by a [B, 1] column): index = i / (elements per small element)
    auto leading = [&](const dsc_tensor *small, long long *per) {
        bool ones = false;
        long long inner = 1;
        for (int i = 0; i < DSC_MAX_DIMS; ++i) {
            if (!ones && small->shape[i] == shape[i]) continue;
            if (small->shape[i] != 1) return false;
            ones = true;
            inner *= shape[i];
        }
        *per = inner;
        return ones && inner > 1 && inner < (1LL << 31);
    };
    g.fast = 0;
    g.small_ne = 1;
    long long per = 0;
    if (xa->ne == out->ne && xb->ne == out->ne) g.fast = 1;
    else if (xa->ne == out->ne && trailing(xb)) { g.fast = 2; g.small_ne = xb->ne; }
    else if (xb->ne == out->ne && trailing(xa)) { g.fast = 3; g.small_ne = xa->ne; }
    else if (xa->ne == out->ne && leading(xb, &per)) { g.fast = 4; g.small_ne = (int) per; }
    else if (xb->ne == out->ne && leading(xa, &per)) { g.fast = 5; g.small_ne = (int) per; }
    dsc_launch_binary(ca->data, cb->data, out->data, out_dtype, op, g, ctx->stream);

    if (ca != xa) drop_scratch_tensor(ctx, ca);
    if (cb != xb) drop_scratch_tensor(ctx, cb);
    return out;
}

extern "C" dsc_tensor *dsc_mul(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out) {
    return binary_entry(ctx, xa, xb, out, 2);
}
// dsc.cpp:1247-1271, 1286-1297: the same skeleton with the other functors
extern "C" dsc_tensor *dsc_add(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out) { return binary_entry(ctx, xa, xb, out, 0); }
extern "C" dsc_tensor *dsc_sub(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out) { return binary_entry(ctx, xa, xb, out, 1); }
extern "C" dsc_tensor *dsc_div(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out) { return binary_entry(ctx, xa, xb, out, 3); }

// ------------------------------------------------------------------------------ unary (spectrum consumers)

static dsc_dtype as_real(dsc_dtype t) { return dsc_is_single(t) ? DSC_F32 : DSC_F64; }

static dsc_tensor *unary_entry(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int op, dsc_dtype out_dtype) {
    static const char *names[] = {"dsc_abs", "dsc_angle", "dsc_conj", "dsc_real", "dsc_imag"};
    dsc_trace_scope trace__(ctx, names[op], "op;unary", x);
    if (out == nullptr) {
        out = dsc_new_tensor(ctx, x->n_dim, &x->shape[DSC_MAX_DIMS - x->n_dim], out_dtype, nullptr);
    } else {                                          // dsc.cpp:1490-1494
        DSC_ASSERT(out->dtype == out_dtype);
        DSC_ASSERT(out->n_dim == x->n_dim);
        DSC_ASSERT(memcmp(out->shape, x->shape, sizeof(x->shape)) == 0);
    }
    dsc_launch_unary(x->data, x->dtype, out->data, op, x->ne, ctx->stream);
    return out;
}

// dsc.cpp:1480-1513
extern "C" dsc_tensor *dsc_abs(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out) {
    DSC_ASSERT(x != nullptr);
    return unary_entry(ctx, x, out, 0, as_real(x->dtype));
}
// dsc.cpp:1515-1541
extern "C" dsc_tensor *dsc_angle(dsc_ctx *ctx, const dsc_tensor *x) {
    DSC_ASSERT(x != nullptr);
    return unary_entry(ctx, x, nullptr, 1, as_real(x->dtype));
}
// dsc.cpp:1543-1568: a real input is returned as is
extern "C" dsc_tensor *dsc_conj(dsc_ctx *ctx, dsc_tensor *x) {
    DSC_ASSERT(x != nullptr);
    if (!dsc_is_complex(x->dtype)) return x;
    return unary_entry(ctx, x, nullptr, 2, x->dtype);
}
// dsc.cpp:1570-1594
extern "C" dsc_tensor *dsc_real(dsc_ctx *ctx, dsc_tensor *x) {
    DSC_ASSERT(x != nullptr);
    if (!dsc_is_complex(x->dtype)) return x;
    return unary_entry(ctx, x, nullptr, 3, as_real(x->dtype));
}
// dsc.cpp:1596-1622
extern "C" dsc_tensor *dsc_imag(dsc_ctx *ctx, const dsc_tensor *x) {
    DSC_ASSERT(x != nullptr);
    return unary_entry(ctx, x, nullptr, 4, as_real(x->dtype));
}

// ------------------------------------------------------------------------------ reductions

// dsc.cpp:83-115 (validate_reduce_params).  For keep_dims=false the reference leaves
// 0x01010101 in the leading slots of its scratch shape (memset with 1, :98), which only a
// user-supplied `out` can observe; here those slots are 1 and such an `out` is accepted.
static dsc_tensor *reduce_entry(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int axis, bool keep_dims, int op) {
    DSC_ASSERT(x != nullptr);
    static const char *names[] = {"dsc_sum", "dsc_mean", "dsc_max", "dsc_min"};
    dsc_trace_scope trace__(ctx, names[op], "op;unary", x, nullptr, 0, axis);
    const int slot = dsc_axis_slot(x, axis);
    DSC_ASSERT(slot >= 0 && slot < DSC_MAX_DIMS);

    int out_shape[DSC_MAX_DIMS];
    int out_ndim = x->n_dim;
    if (keep_dims) {
        memcpy(out_shape, x->shape, sizeof(out_shape));
        out_shape[slot] = 1;
    } else {
        out_ndim--;
        const int lead = DSC_MAX_DIMS - out_ndim;
        for (int i = 0; i < lead; ++i) out_shape[i] = 1;
        for (int xi = DSC_MAX_DIMS - x->n_dim, oi = 0; xi < DSC_MAX_DIMS; ++xi) {
            if (xi == slot) continue;
            out_shape[lead + oi++] = x->shape[xi];
        }
    }
    if (out == nullptr) {
        out = dsc_new_tensor(ctx, out_ndim, &out_shape[DSC_MAX_DIMS - out_ndim], x->dtype, nullptr);
    } else {
        DSC_ASSERT(out->dtype == x->dtype);
        DSC_ASSERT(out->n_dim == out_ndim);
        DSC_ASSERT(memcmp(out->shape, out_shape, sizeof(out_shape)) == 0);
    }

    long long outer = 1, inner = 1;
    for (int i = 0; i < slot; ++i) outer *= x->shape[i];
    for (int i = slot + 1; i < DSC_MAX_DIMS; ++i) inner *= x->shape[i];
    // workspace for the segmented path: whatever the scratch arena has, up to 64 MiB
    ctx->scratch.reset();
    size_t ws_bytes = ctx->scratch.capacity() > (64u << 20) ? (64u << 20) : ctx->scratch.capacity() / 2;
    void *ws = ws_bytes >= 4096 ? ctx->scratch.alloc(ws_bytes) : nullptr;
    dsc_launch_reduce(x->data, out->data, x->dtype, op, outer, x->shape[slot], inner, ws, ws_bytes, ctx->stream);
    return out;
}

extern "C" dsc_tensor *dsc_sum (dsc_ctx *c, const dsc_tensor *x, dsc_tensor *o, int axis, bool keep) { return reduce_entry(c, x, o, axis, keep, 0); }
extern "C" dsc_tensor *dsc_mean(dsc_ctx *c, const dsc_tensor *x, dsc_tensor *o, int axis, bool keep) { return reduce_entry(c, x, o, axis, keep, 1); }
extern "C" dsc_tensor *dsc_max (dsc_ctx *c, const dsc_tensor *x, dsc_tensor *o, int axis, bool keep) { return reduce_entry(c, x, o, axis, keep, 2); }
extern "C" dsc_tensor *dsc_min (dsc_ctx *c, const dsc_tensor *x, dsc_tensor *o, int axis, bool keep) { return reduce_entry(c, x, o, axis, keep, 3); }
