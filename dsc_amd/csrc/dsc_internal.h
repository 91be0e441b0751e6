// dsc_internal.h — host-side internals of the MI355X backend (context, HBM arenas,
// plan cache).  Not part of the C ABI (include/dsc_mi355x.h).
#pragma once

#include "dsc_mi355x.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

// Error convention of the reference (dsc/include/dsc.h:14-28): message on stderr, exit.
#define DSC_LOG_FATAL(format, ...)                                        \
    do {                                                                  \
        fprintf(stderr, "%s: " format "\n", __func__, ##__VA_ARGS__);    \
        exit(EXIT_FAILURE);                                               \
    } while (0)

#define DSC_LOG_INFO(format, ...) fprintf(stdout, "%s: " format "\n", __func__, ##__VA_ARGS__)

#define DSC_ASSERT(x)                                                             \
    do {                                                                          \
        if (!(x)) {                                                               \
            fprintf(stderr, "DSC_ASSERT: %s:%d %s\n", __FILE__, __LINE__, #x);    \
            exit(EXIT_FAILURE);                                                   \
        }                                                                         \
    } while (0)

#define HIP_CHECK(call)                                                                       \
    do {                                                                                      \
        hipError_t err_ = (call);                                                             \
        if (err_ != hipSuccess) {                                                             \
            fprintf(stderr, "HIP error %s:%d: %s -> %s\n", __FILE__, __LINE__, #call,        \
                    hipGetErrorString(err_));                                                 \
            exit(EXIT_FAILURE);                                                               \
        }                                                                                     \
    } while (0)

#define DSC_ALIGN_UP(x, y) (((x) + (y) - 1) & ~((size_t) (y) - 1))
#define DSC_DEVICE_ALIGN ((size_t) 256)
#define DSC_MAX_FFT_PLANS 16                     // dsc/src/dsc.cpp:19-21
#define DSC_HEADER_QUARANTINE ((size_t) 65536)   // freed tensor headers (64 B each) kept out of circulation

static inline size_t dsc_dtype_size(dsc_dtype t) {   // dsc_dtype.h:58-63
    static const size_t sz[4] = {4, 8, 8, 16};
    DSC_ASSERT(t < 4);
    return sz[t];
}
static inline bool dsc_is_complex(dsc_dtype t) { return t == DSC_C32 || t == DSC_C64; }
static inline bool dsc_is_single(dsc_dtype t) { return t == DSC_F32 || t == DSC_C32; }

// dsc/include/dsc.h:81 (dsc_tensor_dim): user axis -> slot in the 4-wide arrays
static inline int dsc_axis_slot(const dsc_tensor *x, int axis) {
    return axis < 0 ? DSC_MAX_DIMS + axis : DSC_MAX_DIMS - x->n_dim + axis;
}

// dsc/include/dsc.h:122-132
static inline int dsc_pow2_n(int n) {
    DSC_ASSERT(n > 0);
    int p = n - 1;
    p |= p >> 1; p |= p >> 2; p |= p >> 4; p |= p >> 8; p |= p >> 16;
    return p + 1;
}

// ---------------------------------------------------------------------------------
// HBM arenas.  The reference keeps allocator nodes inside the arena bytes
// (dsc/src/dsc_allocator.cpp:34-49); a device arena cannot, so the bookkeeping is
// host-side and the arena holds payload only.

// Main arena: best-fit free list with coalescing of neighbours on free — the policy of
// dsc_generic_allocator (dsc_allocator.cpp:51-221).
class dsc_main_arena {
public:
    void init(char *base, size_t size);
    char *alloc(size_t nb, bool from_top = false);   // fatal when no block fits (dsc_allocator.cpp:112-114)
    bool fits(size_t nb_a, size_t nb_b = 0) const;   // non-fatal probe: would these two blocks fit now?
    void free(char *p);                  // unknown / already freed pointers are ignored (:152-181)
    void clear();
    size_t used() const { return used_; }
    size_t capacity() const { return size_; }

private:
    char *base_ = nullptr;
    size_t size_ = 0, used_ = 0;
    std::map<size_t, size_t> free_;                 // offset -> size, address ordered
    std::unordered_map<size_t, size_t> live_;       // offset -> size
};

// Scratch arena: bump pointer, reset as a whole (dsc_allocator.cpp:226-304).  Work on one
// in-order stream makes reuse after reset safe without waiting.
class dsc_scratch_arena {
public:
    void init(char *base, size_t size) { base_ = base; size_ = size; top_ = 0; }
    char *alloc(size_t nb);
    void reset() { top_ = 0; }
    size_t capacity() const { return size_; }

private:
    char *base_ = nullptr;
    size_t size_ = 0, top_ = 0;
};

// Our buffer record: `pub.refs` first so that &rec->pub is the ABI's dsc_tensor_buffer*.
struct dsc_buffer_rec {
    dsc_tensor_buffer pub;
    char *dev;                // arena block (NULL for buffers living in scratch)
    size_t nbytes;
    bool external = false;    // caller-owned device memory (dsc_tensor_from_device_ptr): never returned to the arena
};

// A plan = device twiddle tables for one (n, fft_type, precision): dsc_fft.h:18-27.
struct dsc_fft_plan {
    int n;                    // complex length of the transform (pow2)
    int last_used;
    dsc_dtype dtype;          // DSC_F32 or DSC_F64 (twiddle precision)
    dsc_fft_type fft_type;
    void *tw_full;            // W_n^k, k in [0, n): interleaved (cos, sin), device
    void *tw_real;            // REAL plans: W_{2n}^k, k in [0, n/2]: post/pre-pass factors
    void *tw_aux;             // kernel-specific tables (register-resident kernels)
    char *block;              // arena block holding the above
};

// Tracing (dsc.h:159-168, dsc_tracing.h): one record per operator call while recording is on — host begin / end
// timestamps plus a pair of HIP events on the context's stream, so that the dump shows both the (asynchronous) API call
// and the kernels it enqueued.  tracing.cpp.
struct dsc_trace_rec {
    char name[32], cat[16], args[224];
    unsigned long long ts_b, ts_e;       // host, microseconds
    hipEvent_t ev_b, ev_e;
};
struct dsc_tracer {
    bool recording = false;
    bool based = false;
    unsigned long long base_ts = 0;      // host time of base_ev
    hipEvent_t base_ev = nullptr;
    std::vector<dsc_trace_rec> recs;
    std::vector<hipEvent_t> free_events;
};

struct dsc_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev_start, ev_stop;
    char *main_buf, *scratch_buf;
    dsc_main_arena main;
    dsc_scratch_arena scratch;
    dsc_fft_plan *fft_plans[DSC_MAX_FFT_PLANS];
    std::unordered_set<dsc_tensor *> live_tensors;     // tolerate double frees (tensor.py __del__)
    // Header recycling.  A header address handed to the caller is a handle the caller may still hold after freeing it
    // (Python's __del__ frees twice; handles outlive dsc_ctx_clear): reusing the address at once would let such a
    // stale dsc_tensor_free release a NEW tensor's block.  Freed user-visible headers therefore wait in `quarantine`
    // until DSC_HEADER_QUARANTINE younger ones have been freed; only headers that never left the library (scratch
    // temporaries of an operator) go straight back to `tensor_pool`.
    std::deque<dsc_tensor *> quarantine;
    std::vector<dsc_tensor *> tensor_pool;             // headers safe to hand out again
    const char *last_fft_path;
    int n_cu;
    dsc_tracer tracer;
    std::vector<hipStream_t> peer_streams;             // copy lanes of dsc_peer_push (peer.cpp), created on first use
    hipEvent_t peer_ready = nullptr;
    unsigned *async_error = nullptr;                   // pinned: error word of the last team-barrier kernel (fft_xcd_fused.hip), checked at every synchronise
};

// hipStreamSynchronize + the deferred device-side error checks
void dsc_stream_sync(dsc_ctx *ctx);

// internal helpers shared by the C-ABI translation units
dsc_tensor *dsc_new_tensor_in(dsc_ctx *ctx, int n_dim, const int *shape, dsc_dtype dtype,
                              dsc_tensor_buffer *buffer, bool in_scratch);
// header over caller-owned device memory (dsc_tensor_from_device_ptr): the buffer record does not own an arena block
dsc_tensor *dsc_new_tensor_over(dsc_ctx *ctx, void *ptr, size_t nbytes, int n_dim, const int *shape, dsc_dtype dtype);
void dsc_peer_release(dsc_ctx *ctx);

// RAII scope placed at the top of an operator entry point: no cost beyond a branch unless recording.
void dsc_trace_begin(dsc_ctx *ctx, const char *name, const char *cat, const dsc_tensor *a, const dsc_tensor *b, int i0, int i1);
void dsc_trace_end(dsc_ctx *ctx, size_t index);
void dsc_trace_release(dsc_ctx *ctx);
struct dsc_trace_scope {
    dsc_ctx *ctx;
    size_t index;
    bool on;
    dsc_trace_scope(dsc_ctx *c, const char *name, const char *cat, const dsc_tensor *a = nullptr, const dsc_tensor *b = nullptr,
                    int i0 = 0, int i1 = 0) : ctx(c), index(0), on(c != nullptr && c->tracer.recording) {
        if (on) { index = c->tracer.recs.size(); dsc_trace_begin(c, name, cat, a, b, i0, i1); on = c->tracer.recs.size() > index; }
    }
    ~dsc_trace_scope() { if (on) dsc_trace_end(ctx, index); }
};
#define DSC_TRACE_OP(ctx, cat, ...) dsc_trace_scope trace__((ctx), __func__, (cat), ##__VA_ARGS__)
