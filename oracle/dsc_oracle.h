/*
 * dsc_oracle.h — CPU restatement of the dspcraft/dsc FFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (include/, dsc_amd/)
 * includes, links or executes this.  It is the checker used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  Every entry point here is compared element-wise
 * against the reference itself (oracle/_ref/libdsc_ref.so, compiled by
 * oracle/Makefile from the sources under /root/reference) by
 * tests/test_oracle_vs_ref.py, and against the committed fixtures under
 * tests/golden/ (generated from that same reference build by
 * tests/golden/make_golden.py).  The reference ships no golden vectors of its
 * own (its tests compare against numpy on unseeded random data,
 * python/tests/test_ops.py:32-39, 458-489).
 *
 * Each function cites the reference file:line it restates.  Paths are relative
 * to /root/reference.
 */
#ifndef DSC_ORACLE_H
#define DSC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dsc/include/dsc_dtype.h:51-56 — enum values are ABI. */
enum { ORC_F32 = 0, ORC_F64 = 1, ORC_C32 = 2, ORC_C64 = 3 };

/* dsc/include/dsc_fft.h:13-16 */
enum { ORC_REAL = 0, ORC_COMPLEX = 1 };

#define ORC_MAX_DIMS 4

/* Host-side tensor descriptor: the subset of dsc_tensor (dsc/include/dsc.h:96-108)
 * the hot path reads.  shape is right-aligned and padded with 1, stride is in
 * ELEMENTS, row-major contiguous (dsc/src/dsc.cpp:373-384). */
typedef struct {
    int   shape[ORC_MAX_DIMS];
    int   stride[ORC_MAX_DIMS];
    void *data;
    int   ne;
    int   n_dim;
    int   dtype;
} orc_tensor;

/* Fill shape/stride/ne/n_dim exactly as dsc_new_tensor does (dsc.cpp:353-384). */
void orc_tensor_init(orc_tensor *t, int n_dim, const int *shape, int dtype, void *data);

size_t orc_dtype_size(int dtype);                 /* dsc_dtype.h:58-63 */
int    orc_promote(int dtype_a, int dtype_b);     /* dsc_dtype.h:73-78 */
int    orc_pow2_n(int n);                         /* dsc.h:122-132 */

/* Plan = twiddle table (dsc_fft.h:18-55, 109-135). */
size_t orc_fft_storage(int n, int dtype, int fft_type);
void   orc_init_plan(void *twiddles, int n, int dtype, int fft_type);

/* 1-D kernels on one contiguous buffer (dsc_fft.h:57-103, 156-238).
 * x has n complex (complex_fft) or n+1 complex (real_fft) elements; work same. */
void orc_complex_fft_c32(const float  *tw, float  *x, float  *work, int n, int forward);
void orc_complex_fft_c64(const double *tw, double *x, double *work, int n, int forward);
void orc_real_fft_c32(const float  *tw, float  *x, float  *work, int n, int forward);
void orc_real_fft_c64(const double *tw, double *x, double *work, int n, int forward);

/* Shape rules (dsc.cpp:2019-2047, 2188-2224): fills out_shape[4] (right-aligned)
 * and *out_dtype; returns 0, or -1 where the reference would DSC_LOG_FATAL. */
int orc_fft_out_shape (const orc_tensor *x, int n, int axis, int out_shape[4], int *out_dtype);
int orc_rfft_out_shape(const orc_tensor *x, int n, int axis, int forward, int out_shape[4], int *out_dtype);

/* Drivers (dsc.cpp:1958-2100, 2102-2260).  `out` must already have the shape
 * and dtype given by the *_out_shape functions. */
int orc_fft  (const orc_tensor *x, orc_tensor *out, int n, int axis);
int orc_ifft (const orc_tensor *x, orc_tensor *out, int n, int axis);
int orc_rfft (const orc_tensor *x, orc_tensor *out, int n, int axis);
int orc_irfft(const orc_tensor *x, orc_tensor *out, int n, int axis);

/* dsc_cast (dsc.cpp:536-597, dsc_ops.h:12-44): element-wise dtype conversion. */
void orc_cast(const orc_tensor *x, orc_tensor *out);

/* dsc_mul (dsc.cpp:44-69, 1174-1245, 1273-1284; dsc_ops.h:68-78).
 * out shape = element-wise max of the operand shapes, dtype = orc_promote. */
int orc_mul_out_shape(const orc_tensor *xa, const orc_tensor *xb, int out_shape[4], int *out_n_dim, int *out_dtype);
int orc_mul(const orc_tensor *xa, const orc_tensor *xb, orc_tensor *out);
/* dsc_add / dsc_sub / dsc_mul / dsc_div (dsc.cpp:1247-1297; dsc_ops.h:46-90): same shape and promotion rules. */
enum { ORC_ADD = 0, ORC_SUB = 1, ORC_MUL = 2, ORC_DIV = 3 };
int orc_binary(const orc_tensor *xa, const orc_tensor *xb, orc_tensor *out, int op);

/* dsc_abs / dsc_angle / dsc_conj / dsc_real / dsc_imag (dsc.cpp:1480-1622; dsc_ops.h:242-303) — SURVEY 8f row 2.
 * abs, angle, real, imag produce the REAL dtype of x; conj keeps it.  (dsc_conj / dsc_real of a real tensor
 * return x itself in the reference; here out simply receives a copy.) */
enum { ORC_ABS = 0, ORC_ANGLE = 1, ORC_CONJ = 2, ORC_REALPART = 3, ORC_IMAGPART = 4 };
int orc_unary_out_dtype(int dtype, int op);
int orc_unary(const orc_tensor *x, orc_tensor *out, int op);

/* Reductions (dsc.cpp:83-115, 1774-1953; dsc_ops.h:46-55, 318-339).
 * op: 0 sum, 1 mean, 2 max, 3 min. */
enum { ORC_SUM = 0, ORC_MEAN = 1, ORC_MAX = 2, ORC_MIN = 3 };
int orc_reduce_out_shape(const orc_tensor *x, int axis, int keep_dims, int out_shape[4], int *out_n_dim);
int orc_reduce(const orc_tensor *x, orc_tensor *out, int axis, int op);

#ifdef __cplusplus
}
#endif
#endif
