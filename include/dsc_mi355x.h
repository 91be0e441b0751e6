/*
 * dsc_mi355x.h — C ABI of the MI355X (gfx950) backend for DSC's FFT hot path.
 *
 * Drop-in seam: the `extern "C"` block of the reference's dsc/include/dsc.h:85-428.
 * Every function in section A keeps the reference's name, argument order, argument
 * meaning and error behaviour (invalid arguments / out of memory print to stderr and
 * exit(EXIT_FAILURE), dsc.h:14-28), so the reference's own bindings
 * (python/dsc/_bindings.py, dsc/api/dsc_api.h) load this library unchanged for that
 * subset.  What differs, and why:
 *
 *   - `dsc_tensor.data` is a DEVICE pointer into an HBM arena.  Host code must not
 *     dereference it; section B adds the explicit copy entry points the reference
 *     never needed (python/dsc/tensor.py:305-323, 371-377 memmove / view the pointer).
 *   - `dsc_tensor.backend` is DSC_BACKEND_MI355X (1); the reference only has CPU = 0
 *     (dsc/include/dsc_backend.h:11-13).
 *   - tensor headers, buffer refcounts, allocator nodes and FFT plans live in host
 *     memory next to the context instead of inside the arena
 *     (reference: dsc/src/dsc_allocator.cpp:34-49, dsc/src/dsc.cpp:255-256, 356-361).
 *   - operators are enqueued on the context's HIP stream and return immediately;
 *     every host-visible read (dsc_copy_to_host, dsc_synchronize) waits for them.
 *     Since `data` is not host-readable, this is indistinguishable from the
 *     reference's blocking calls.
 *
 * Plain C: pointers, sizes and PODs only.  No torch / HIP types in any signature.
 * Citations are file:line under the reference tree.
 */
#ifndef DSC_MI355X_H
#define DSC_MI355X_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSC_MAX_DIMS 4                         /* dsc.h:72-76 */

/* dsc_dtype.h:51-56 — `enum dsc_dtype : u8`; values are ABI (python/dsc/dtype.py:15-19) */
typedef uint8_t dsc_dtype;
enum { DSC_F32 = 0, DSC_F64 = 1, DSC_C32 = 2, DSC_C64 = 3 };

/* dsc_fft.h:13-16 — `enum dsc_fft_type : u8` */
typedef uint8_t dsc_fft_type;
enum { DSC_FFT_REAL = 0, DSC_FFT_COMPLEX = 1 };

/* dsc_backend.h:11-13 — `enum dsc_backend_type : u8 { CPU = 0 }`; MI355X is new */
typedef uint8_t dsc_backend_type;
enum { DSC_BACKEND_CPU = 0, DSC_BACKEND_MI355X = 1 };

/* dsc_dtype.h:36-49 — interleaved (real, imag) */
typedef struct { float  real, imag; } dsc_c32;
typedef struct { double real, imag; } dsc_c64;

typedef struct dsc_ctx dsc_ctx;                 /* dsc.cpp:140-145 (opaque) */
typedef struct dsc_fft_plan dsc_fft_plan;       /* dsc_fft.h:18-27 (opaque here) */

/* dsc.cpp:136-138 — `refs` first, as python/dsc/_bindings.py:38-41 mirrors it. */
typedef struct dsc_tensor_buffer {
    int refs;
} dsc_tensor_buffer;

/* dsc.h:96-108 — 64-byte POD; shape right-aligned and padded with 1; stride in
 * ELEMENTS, row-major contiguous (dsc.cpp:373-384); mirrored by _bindings.py:44-54. */
typedef struct dsc_tensor {
    int shape[DSC_MAX_DIMS];
    int stride[DSC_MAX_DIMS];
    dsc_tensor_buffer *buffer;
    void *data;                 /* DEVICE pointer (HBM arena), 256-B aligned */
    int ne;
    int n_dim;
    dsc_dtype dtype;
    dsc_backend_type backend;
} dsc_tensor;

/* ===================================================================== A. reference surface */

/* dsc.h:137, dsc.cpp:150-180.  Allocates the main arena (best-fit free list, as
 * dsc_allocator.cpp:51-221) and the scratch arena (bump allocator, :226-304) in HBM on
 * the current HIP device (see dsc_set_device) and creates the context's stream. */
dsc_ctx *dsc_ctx_init(size_t main_mem, size_t scratch_mem);

/* dsc.h:139-141, dsc.cpp:218-267: 16-slot plan cache keyed on (pow2ceil(n), fft_type,
 * twiddle precision), LRU eviction.  A plan here is the device twiddle tables. */
dsc_fft_plan *dsc_plan_fft(dsc_ctx *ctx, int n, dsc_fft_type fft_type, dsc_dtype dtype);

void   dsc_ctx_free(dsc_ctx *ctx);                          /* dsc.h:146, dsc.cpp:272-285 */
void   dsc_ctx_clear(dsc_ctx *ctx);                         /* dsc.h:148, dsc.cpp:287-291 */
void   dsc_tensor_free(dsc_ctx *ctx, dsc_tensor *x);        /* dsc.h:150, dsc.cpp:293-303; NULL and repeated frees are ignored */
size_t dsc_used_mem(dsc_ctx *ctx);                          /* dsc.h:155, dsc.cpp:310-312 */
void   dsc_print_mem_usage(dsc_ctx *ctx);                   /* dsc.h:157, dsc.cpp:314-322 */

/* dsc.h:159-168, dsc_tracing.h — operator tracing.  While recording, every operator call logs its
 * host-side begin / end AND the span its kernels took on the context's HIP stream (a pair of events); dsc_dump_traces
 * writes a Perfetto / chrome://tracing JSON array with the reference's fields ("name", "cat", "ph", "ts", "pid", "tid",
 * "args") on two tracks: tid 0 = API calls ("B" / "E"), tid 1 = device execution ("X" with "dur").  Always compiled in
 * (the reference needs DSC_ENABLE_TRACING); not recording costs one branch per call. */
void dsc_traces_record(dsc_ctx *ctx, bool record);
void dsc_dump_traces(dsc_ctx *ctx, const char *filename);
void dsc_clear_traces(dsc_ctx *ctx);

/* dsc.h:173-198, dsc.cpp:342-428.  buffer == NULL allocates; otherwise the new tensor
 * shares (and references) `buffer`. */
dsc_tensor *dsc_new_tensor(dsc_ctx *ctx, int n_dim, const int *shape, dsc_dtype dtype, dsc_tensor_buffer *buffer);
dsc_tensor *dsc_view(dsc_ctx *ctx, const dsc_tensor *x);
dsc_tensor *dsc_tensor_1d(dsc_ctx *ctx, dsc_dtype dtype, int dim1);
dsc_tensor *dsc_tensor_2d(dsc_ctx *ctx, dsc_dtype dtype, int dim1, int dim2);
dsc_tensor *dsc_tensor_3d(dsc_ctx *ctx, dsc_dtype dtype, int dim1, int dim2, int dim3);
dsc_tensor *dsc_tensor_4d(dsc_ctx *ctx, dsc_dtype dtype, int dim1, int dim2, int dim3, int dim4);

/* dsc.h:200-210, dsc.cpp:430-470: one-element 1-D tensors holding a scalar. */
dsc_tensor *dsc_wrap_f32(dsc_ctx *ctx, float val);
dsc_tensor *dsc_wrap_f64(dsc_ctx *ctx, double val);
dsc_tensor *dsc_wrap_c32(dsc_ctx *ctx, dsc_c32 val);
dsc_tensor *dsc_wrap_c64(dsc_ctx *ctx, dsc_c64 val);

/* dsc.h:221-223, dsc.cpp:587-597: returns x itself when the dtype already matches. */
dsc_tensor *dsc_cast(dsc_ctx *ctx, dsc_tensor *x, dsc_dtype new_dtype);

/* dsc.h:275-278, dsc.cpp:1273-1284 (+ :44-69, :1174-1245; dsc_ops.h:68-78).
 * NumPy-style broadcasting over the 4 right-aligned dims, result dtype from the
 * promotion table dsc_dtype.h:73-78 (F64 x C32 -> C32).  out may be NULL. */
dsc_tensor *dsc_mul(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out);
/* dsc.h:265-283, dsc.cpp:1247-1297 — same skeleton with add_op / sub_op / div_op (dsc_ops.h:46-90);
 * SURVEY 8f "next" row 2. */
dsc_tensor *dsc_add(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out);
dsc_tensor *dsc_sub(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out);
dsc_tensor *dsc_div(dsc_ctx *ctx, dsc_tensor *xa, dsc_tensor *xb, dsc_tensor *out);

/* dsc.h:325-340, dsc.cpp:1480-1622 (functors dsc_ops.h:242-303) — magnitude / phase / parts of a spectrum
 * (SURVEY 8f "next" row 2).  abs, angle, imag (and real of a complex tensor) return the REAL dtype of x;
 * dsc_conj / dsc_real of a real tensor return x itself, as the reference does. */
dsc_tensor *dsc_abs(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out);
dsc_tensor *dsc_angle(dsc_ctx *ctx, const dsc_tensor *x);
dsc_tensor *dsc_conj(dsc_ctx *ctx, dsc_tensor *x);
dsc_tensor *dsc_real(dsc_ctx *ctx, dsc_tensor *x);
dsc_tensor *dsc_imag(dsc_ctx *ctx, const dsc_tensor *x);

/* dsc.h:110-117 — one slice per leading dimension; DSC_VALUE_NONE (dsc.h:78) in a field selects NumPy's
 * default for it; start == stop == step != NONE means "this single index" and collapses the dimension
 * (how the wrapper spells x[:, 1], tensor.py:106-118). */
#define DSC_VALUE_NONE INT32_MAX
typedef struct dsc_slice {
    int start, stop, step;
} dsc_slice;

/* dsc.h:236-260, dsc.cpp:829-1169 — indexing and slicing ON THE DEVICE (SURVEY 8f "next" row 1): the
 * `[:output_length]` crop after dsc_irfft and the placement of a block into a zero-padded buffer no longer
 * round-trip through the host.  Variadic exactly as the reference: `indexes` ints, or `slices` dsc_slice
 * structs BY VALUE.  get_* return a new contiguous tensor (a fully indexed element is a 1-element 1-D tensor);
 * set_* write xb (same dtype; a 1-element tensor is broadcast, anything else is consumed cyclically in
 * row-major order, dsc.cpp:1010-1041) into the selected region of xa.  Negative indexes / starts / stops count
 * from the end, negative steps walk backwards; out-of-range arguments abort as the reference's asserts do. */
dsc_tensor *dsc_tensor_get_idx(dsc_ctx *ctx, const dsc_tensor *x, int indexes, ...);
dsc_tensor *dsc_tensor_get_slice(dsc_ctx *ctx, const dsc_tensor *x, int slices, ...);
void dsc_tensor_set_idx(dsc_ctx *ctx, dsc_tensor *xa, const dsc_tensor *xb, int indexes, ...);
void dsc_tensor_set_slice(dsc_ctx *ctx, dsc_tensor *xa, const dsc_tensor *xb, int slices, ...);

/* dsc.h:233-235, dsc.cpp:764-827 — permutation of the axes into a new contiguous tensor;
 * `axes` = 0 reverses them, otherwise `axes` == n_dim ints follow.  A 1-D tensor returns a view, as the reference. */
dsc_tensor *dsc_transpose(dsc_ctx *ctx, const dsc_tensor *x, int axes, ...);

/* dsc.h:416-424, dsc.cpp:2262-2340 — bin centre frequencies; dtype must be real.  Computed in
 * the output precision exactly as the reference does, then placed in HBM. */
dsc_tensor *dsc_fftfreq(dsc_ctx *ctx, int n, double d, dsc_dtype dtype);
dsc_tensor *dsc_rfftfreq(dsc_ctx *ctx, int n, double d, dsc_dtype dtype);

/* dsc.h:358-380, dsc.cpp:1771-1953.  Sequential left-to-right accumulation order per
 * output element is NOT reproduced on the GPU (tree order); max/min are exact
 * including the reference's tie rules on the real part (dsc_ops.h:318-339). */
dsc_tensor *dsc_sum (dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int axis, bool keep_dims);
dsc_tensor *dsc_mean(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int axis, bool keep_dims);
dsc_tensor *dsc_max (dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int axis, bool keep_dims);
dsc_tensor *dsc_min (dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int axis, bool keep_dims);

/* dsc.h:392-414, dsc.cpp:1958-2260.  Out-of-place, any axis, every length rounded up
 * to a power of two; n <= 0 means "length of the axis"; irfft's n counts BINS
 * (dsc.cpp:2199-2200).  out may be NULL; if given it must match dtype/n_dim/shape. */
dsc_tensor *dsc_fft  (dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis);
dsc_tensor *dsc_ifft (dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis);
dsc_tensor *dsc_rfft (dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis);
dsc_tensor *dsc_irfft(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis);

/* ===================================================================== B. device seam (new) */

/* Select the HIP device used by the next dsc_ctx_init (one process per GPU: call with
 * LOCAL_RANK).  Returns the number of visible devices. */
int dsc_set_device(int device);

/* Replaces the host memmove in python/dsc/tensor.py:371-377 (from_numpy) and the
 * zero-copy view in :305-323 (numpy()).  nbytes must be <= ne * sizeof(dtype).
 * Both are stream-ordered; dsc_copy_to_host returns after the bytes have landed. */
void dsc_copy_from_host(dsc_ctx *ctx, dsc_tensor *dst, const void *src, size_t nbytes);
void dsc_copy_to_host(dsc_ctx *ctx, const dsc_tensor *src, void *dst, size_t nbytes);

/* Wait for everything enqueued on the context's stream. */
void dsc_synchronize(dsc_ctx *ctx);

/* The context's hipStream_t as an opaque pointer (interop / profiling). */
void *dsc_stream(dsc_ctx *ctx);

/* HIP-event stopwatch on the context's stream: start records an event, stop records a
 * second one, waits for it and returns the elapsed milliseconds between the two. */
void  dsc_timer_start(dsc_ctx *ctx);
float dsc_timer_stop(dsc_ctx *ctx);

/* Fused README filterFFT (README.md:113-135): out = irfft(rfft(s, n) * H) along the last
 * axis, one launch per batch of rows, the spectrum never written to HBM.
 *   s   real [.., ls]   (f32), zero-padded / cropped to n = 2*(H_bins-1) like dsc_rfft
 *   H   complex [H_bins] (c32) filter spectrum, H_bins = n/2 + 1, broadcast over rows
 *   out real [.., n] or NULL
 * Equals dsc_irfft(dsc_mul(dsc_rfft(s, n), H)) within float rounding.  Falls back to
 * exactly that three-op composition for sizes without a fused kernel. */
dsc_tensor *dsc_filter_fft(dsc_ctx *ctx, const dsc_tensor *s, const dsc_tensor *H, dsc_tensor *out);

/* Name of the kernel path the last FFT-family call took ("r2c_64k_regs", "generic_lds",
 * "generic_4step", ...): lets tests assert that the hand-written path really ran. */
const char *dsc_last_fft_path(dsc_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * Section C — multi-GPU reassembly of batch-sharded outputs.
 *
 * No reference counterpart: the reference has one backend (CPU, dsc/include/dsc_backend.h:11-13) and no communication
 * layer.  Rows of a batched transform are independent, so rank r (one process per GPU) transforms rows
 * [r*B/P, (r+1)*B/P) with no exchange on the data path; these entry points exist to put the P shards next to each
 * other afterwards.  The gathered output of config 4 (65536 x 32769 c32) exceeds `int ne` (dsc.h:104) and therefore is
 * a RAW device buffer, not a dsc_tensor.  dsc_amd/shard.py drives them (and the RCCL variants); plain C here.
 */

/* hipIpcMemHandle_t as bytes, so that no HIP type appears in a signature. */
typedef struct dsc_ipc_handle { unsigned char bytes[64]; } dsc_ipc_handle;

/* A device buffer of its own (hipMalloc on the context's device), outside both arenas: the gather destination
 * [P x shard].  NULL on failure (this one reports instead of exiting: the caller sizes it from the world size). */
void *dsc_device_alloc(dsc_ctx *ctx, size_t nbytes);
void  dsc_device_free(dsc_ctx *ctx, void *ptr);

/* A dsc_tensor header over `nbytes` of caller-owned device memory (e.g. this rank's slot of the gather destination, so
 * that dsc_rfft writes its shard in place — no local copy).  The tensor does not own the bytes: dsc_tensor_free
 * releases the header only, and the memory must outlive it. */
dsc_tensor *dsc_tensor_from_device_ptr(dsc_ctx *ctx, void *ptr, size_t nbytes, int n_dim, const int *shape, dsc_dtype dtype);

/* Export `ptr` (a dsc_device_alloc result) to the other ranks' processes / open another rank's export.  The mapped
 * pointer addresses the PEER GPU's memory over xGMI (or the same GPU when ranks share one).  0 on success. */
int   dsc_ipc_export(dsc_ctx *ctx, void *ptr, dsc_ipc_handle *out);
void *dsc_ipc_open(dsc_ctx *ctx, const dsc_ipc_handle *handle);
int   dsc_ipc_close(dsc_ctx *ctx, void *mapped);

/* Direct push: copy `nbytes` from `src` (local) to `dst` (local or a dsc_ipc_open mapping) on copy lane `lane`
 * (0 <= lane < dsc_peer_lanes(): one HIP stream per lane, i.e. one per xGMI link when lane = peer), ordered AFTER
 * everything enqueued so far on the context's stream — the transform that produced `src` — and asynchronous to what
 * is enqueued afterwards: the gather of chunk i overlaps the transform of chunk i+1.  0 on success. */
int   dsc_peer_lanes(void);
int   dsc_peer_push(dsc_ctx *ctx, void *dst, const void *src, size_t nbytes, int lane);
/* Host waits until every push on every lane has landed (then exchange a barrier before peers read). */
int   dsc_peer_wait(dsc_ctx *ctx);

/* The collective north_star names: an RCCL communicator (one rank per process and GPU) and the all-gather that reassembles
 * the output, callable from a C / C++ host (the reference's C++ users, dsc/api/dsc_api.h:24-34) — no Python required.
 * RCCL is loaded on first use (dlopen); a process that never calls these never maps it.  All calls report failures
 * (-1 / NULL + a message on stderr) instead of exiting: the caller decides what a missing peer means.
 *   bootstrap: rank 0 calls dsc_comm_unique_id, the host ships the 128 bytes to the other ranks by its own means, every rank
 *   calls dsc_comm_init_rank (collective: returns once all n_ranks have called it). */
typedef struct dsc_comm dsc_comm;
typedef struct dsc_comm_id { unsigned char bytes[128]; } dsc_comm_id;      /* ncclUniqueId as bytes */
int       dsc_comm_unique_id(dsc_comm_id *out);
dsc_comm *dsc_comm_init_rank(dsc_ctx *ctx, const dsc_comm_id *id, int n_ranks, int rank);
int       dsc_comm_n_ranks(const dsc_comm *comm);
int       dsc_comm_rank(const dsc_comm *comm);
void      dsc_comm_free(dsc_comm *comm);
/* ONE in-place ncclAllGather on the persistent destination dest[n_ranks][rows][row_bytes]: this rank's shard already lies in
 * slot dest[rank] (its transform wrote it there through dsc_tensor_from_device_ptr views).  Enqueued on the context's stream,
 * i.e. ordered after the transforms enqueued so far; dsc_synchronize waits for it.  Every rank passes the same rows / row_bytes. */
int       dsc_shard_allgather(dsc_ctx *ctx, dsc_comm *comm, void *dest, size_t rows, size_t row_bytes);
/* Rows [row0, row0 + n_rows) of every slot in one group of P-1 ncclSend + P-1 ncclRecv (every GPU talks to all peers at once,
 * one xGMI link per peer): the chunk-wise form, so that chunk i travels while chunk i+1 is transformed.  Same ordering rules. */
int       dsc_shard_exchange_rows(dsc_ctx *ctx, dsc_comm *comm, void *dest, size_t rows, size_t row_bytes, size_t row0, size_t n_rows);

#ifdef __cplusplus
}
#endif
#endif /* DSC_MI355X_H */
