/*
 * dsc_oracle.c — CPU restatement of the dspcraft/dsc FFT hot path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY: see dsc_oracle.h for the rules and the parity
 * status (pinned against oracle/_ref and tests/golden).  Citations are
 * file:line under /root/reference.
 */
#include "dsc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ basics */

/* dsc/include/dsc_dtype.h:58-63 */
size_t orc_dtype_size(int dtype) {
    switch (dtype) {
        case ORC_F32: return 4;
        case ORC_F64: return 8;
        case ORC_C32: return 8;
        case ORC_C64: return 16;
        default:      return 0;
    }
}

/* dsc/include/dsc_dtype.h:73-78.  Note F64 x C32 -> C32 (not NumPy's C64). */
int orc_promote(int a, int b) {
    static const int table[4][4] = {
        {ORC_F32, ORC_F64, ORC_C32, ORC_C64},
        {ORC_F64, ORC_F64, ORC_C32, ORC_C64},
        {ORC_C32, ORC_C32, ORC_C32, ORC_C64},
        {ORC_C64, ORC_C64, ORC_C64, ORC_C64},
    };
    return table[a][b];
}

/* dsc/include/dsc.h:122-132: smallest power of two >= n. */
int orc_pow2_n(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

/* dsc/src/dsc.cpp:353-384: right-aligned shape padded with 1, element strides. */
void orc_tensor_init(orc_tensor *t, int n_dim, const int *shape, int dtype, void *data) {
    int ne = 1;
    for (int i = 0; i < ORC_MAX_DIMS; ++i) {
        const int lead = ORC_MAX_DIMS - n_dim;
        t->shape[i] = i < lead ? 1 : shape[i - lead];
        ne *= t->shape[i];
    }
    t->stride[ORC_MAX_DIMS - 1] = 1;
    for (int i = ORC_MAX_DIMS - 2; i >= 0; --i) t->stride[i] = t->stride[i + 1] * t->shape[i + 1];
    t->ne = ne;
    t->n_dim = n_dim;
    t->dtype = dtype;
    t->data = data;
}

/* dsc/include/dsc.h:81 (dsc_tensor_dim): user axis -> slot in the 4-wide arrays. */
static int orc_axis_slot(const orc_tensor *x, int axis) {
    return axis < 0 ? ORC_MAX_DIMS + axis : ORC_MAX_DIMS - x->n_dim + axis;
}

/* Line enumerator, restating what dsc_axis_iterator yields
 * (dsc/include/dsc_iter.h:11-65): all 1-D lines along `axis`, visited with the
 * remaining dims as an odometer whose last dim turns fastest.  base = flat
 * element index of the line's first element, step = element stride along axis. */
typedef struct {
    const orc_tensor *t;
    int axis;
    int idx[ORC_MAX_DIMS];
    long base;
    int step;
    int started, done;
} orc_lines;

static void orc_lines_begin(orc_lines *it, const orc_tensor *t, int axis) {
    memset(it, 0, sizeof(*it));
    it->t = t;
    it->axis = axis;
    it->step = t->stride[axis];
}

static int orc_lines_next(orc_lines *it) {
    if (it->done) return 0;
    if (!it->started) {
        it->started = 1;
        it->base = 0;
        return it->t->ne > 0;
    }
    for (int d = ORC_MAX_DIMS - 1; d >= 0; --d) {
        if (d == it->axis) continue;
        if (++it->idx[d] < it->t->shape[d]) {
            it->base += it->t->stride[d];
            return 1;
        }
        it->base -= (long) (it->t->shape[d] - 1) * it->t->stride[d];
        it->idx[d] = 0;
    }
    it->done = 1;
    return 0;
}

/* ------------------------------------------------------------------ plans */

/* dsc/include/dsc_fft.h:109-135 */
size_t orc_fft_storage(int n, int dtype, int fft_type) {
    size_t reals = 0;
    const int sets = fft_type == ORC_REAL ? (n << 1) : n;
    for (int m = 2; m <= sets; m <<= 1) reals += (size_t) m;
    return reals * ((dtype == ORC_F32 || dtype == ORC_C32) ? 4 : 8);
}

/* ------------------------------------------------------------------ kernels */

#define R      float
#define FN(x)  x##_c32
#define R_COS  cosf
#define R_SIN  sinf
#define R_PI   3.14159265358979323846f
#include "dsc_oracle_fft.inc"
#undef R
#undef FN
#undef R_COS
#undef R_SIN
#undef R_PI

#define R      double
#define FN(x)  x##_c64
#define R_COS  cos
#define R_SIN  sin
#define R_PI   3.14159265358979323846
#include "dsc_oracle_fft.inc"
#undef R
#undef FN
#undef R_COS
#undef R_SIN
#undef R_PI

/* dsc/include/dsc_fft.h:137-154 */
void orc_init_plan(void *twiddles, int n, int dtype, int fft_type) {
    if (dtype == ORC_F32 || dtype == ORC_C32) orc_init_plan_c32((float *) twiddles, n, fft_type);
    else                                      orc_init_plan_c64((double *) twiddles, n, fft_type);
}

/* ------------------------------------------------------------------ FFT drivers */

/* dsc/src/dsc.cpp:2019-2047 */
int orc_fft_out_shape(const orc_tensor *x, int n, int axis, int out_shape[4], int *out_dtype) {
    const int slot = orc_axis_slot(x, axis);
    if (slot < 0 || slot >= ORC_MAX_DIMS) return -1;
    const int len = n > 0 ? orc_pow2_n(n) : orc_pow2_n(x->shape[slot]);
    for (int i = 0; i < ORC_MAX_DIMS; ++i) out_shape[i] = i == slot ? len : x->shape[i];
    *out_dtype = x->dtype == ORC_F32 ? ORC_C32 : x->dtype == ORC_F64 ? ORC_C64 : x->dtype;
    return 0;
}

/* dsc/src/dsc.cpp:2188-2216 */
int orc_rfft_out_shape(const orc_tensor *x, int n, int axis, int forward, int out_shape[4], int *out_dtype) {
    const int slot = orc_axis_slot(x, axis);
    if (slot < 0 || slot >= ORC_MAX_DIMS) return -1;
    const int x_n = x->shape[slot];
    int out_n;
    if (forward) {
        const int order = orc_pow2_n(n > 0 ? n : x_n) >> 1;
        if (order < 1) return -1;               /* a length-1 transform: dsc_plan_fft(0) trips DSC_ASSERT(n > 0) in dsc_pow2_n (dsc.h:124) */
        out_n = order + 1;
        if      (x->dtype == ORC_F32) *out_dtype = ORC_C32;
        else if (x->dtype == ORC_F64) *out_dtype = ORC_C64;
        else return -1;                         /* "RFFT input must be real" */
    } else {
        if ((n > 0 ? n : x_n) < 2) return -1;   /* dsc_pow2_n(0): DSC_ASSERT(n > 0) (dsc.h:124, dsc.cpp:2199) */
        const int order = orc_pow2_n((n > 0 ? n : x_n) - 1);
        out_n = order << 1;
        if      (x->dtype == ORC_C32) *out_dtype = ORC_F32;
        else if (x->dtype == ORC_C64) *out_dtype = ORC_F64;
        else return -1;                         /* "IRFFT input must be complex" */
    }
    for (int i = 0; i < ORC_MAX_DIMS; ++i) out_shape[i] = i == slot ? out_n : x->shape[i];
    return 0;
}

/* dsc/src/dsc.cpp:2009-2100 */
static int orc_fft_any(const orc_tensor *x, orc_tensor *out, int n, int axis, int forward) {
    int shape[4], dtype;
    if (orc_fft_out_shape(x, n, axis, shape, &dtype)) return -1;
    if (out->dtype != dtype || memcmp(shape, out->shape, sizeof(shape))) return -1;
    const int slot = orc_axis_slot(x, axis);
    const int x_n = x->shape[slot];
    const int fft_n = shape[slot];
    const int in_complex = x->dtype == ORC_C32 || x->dtype == ORC_C64;
    if (dtype == ORC_C32) orc_exec_fft_c32(x, out, slot, x_n, fft_n, in_complex, forward);
    else                  orc_exec_fft_c64(x, out, slot, x_n, fft_n, in_complex, forward);
    return 0;
}

int orc_fft (const orc_tensor *x, orc_tensor *out, int n, int axis) { return orc_fft_any(x, out, n, axis, 1); }
int orc_ifft(const orc_tensor *x, orc_tensor *out, int n, int axis) { return orc_fft_any(x, out, n, axis, 0); }

/* dsc/src/dsc.cpp:2173-2260 */
static int orc_rfft_any(const orc_tensor *x, orc_tensor *out, int n, int axis, int forward) {
    int shape[4], dtype;
    if (orc_rfft_out_shape(x, n, axis, forward, shape, &dtype)) return -1;
    if (out->dtype != dtype || memcmp(shape, out->shape, sizeof(shape))) return -1;
    const int slot = orc_axis_slot(x, axis);
    const int x_n = x->shape[slot];
    const int out_n = shape[slot];
    const int order = forward ? out_n - 1 : out_n >> 1;
    if (x->dtype == ORC_F32 || x->dtype == ORC_C32) orc_exec_rfft_c32(x, out, slot, x_n, out_n, order, forward);
    else                                            orc_exec_rfft_c64(x, out, slot, x_n, out_n, order, forward);
    return 0;
}

int orc_rfft (const orc_tensor *x, orc_tensor *out, int n, int axis) { return orc_rfft_any(x, out, n, axis, 1); }
int orc_irfft(const orc_tensor *x, orc_tensor *out, int n, int axis) { return orc_rfft_any(x, out, n, axis, 0); }

/* ------------------------------------------------------------------ cast */

/* Element access in double-double form keeps the switchyard small: every dtype
 * round-trips exactly through (double re, double im). */
static void orc_load(const void *data, int dtype, size_t i, double *re, double *im) {
    switch (dtype) {
        case ORC_F32: *re = ((const float  *) data)[i];       *im = 0; break;
        case ORC_F64: *re = ((const double *) data)[i];       *im = 0; break;
        case ORC_C32: *re = ((const float  *) data)[2 * i];   *im = ((const float  *) data)[2 * i + 1]; break;
        default:      *re = ((const double *) data)[2 * i];   *im = ((const double *) data)[2 * i + 1]; break;
    }
}

/* dsc/include/dsc_ops.h:12-44: complex -> real keeps the real part, real ->
 * complex sets imag to 0, width changes are plain C conversions. */
static void orc_store(void *data, int dtype, size_t i, double re, double im) {
    switch (dtype) {
        case ORC_F32: ((float  *) data)[i] = (float) re; break;
        case ORC_F64: ((double *) data)[i] = re; break;
        case ORC_C32: ((float  *) data)[2 * i] = (float) re; ((float  *) data)[2 * i + 1] = (float) im; break;
        default:      ((double *) data)[2 * i] = re;         ((double *) data)[2 * i + 1] = im; break;
    }
}

/* dsc/src/dsc.cpp:536-597 */
void orc_cast(const orc_tensor *x, orc_tensor *out) {
    for (int i = 0; i < out->ne; ++i) {
        double re, im;
        orc_load(x->data, x->dtype, (size_t) i, &re, &im);
        orc_store(out->data, out->dtype, (size_t) i, re, im);
    }
}

/* ------------------------------------------------------------------ mul */

/* dsc/src/dsc.cpp:1174-1184 (can_broadcast) + :50-55 (shape, dtype). */
int orc_mul_out_shape(const orc_tensor *xa, const orc_tensor *xb, int out_shape[4], int *out_n_dim, int *out_dtype) {
    for (int i = 0; i < ORC_MAX_DIMS; ++i) {
        const int a = xa->shape[i], b = xb->shape[i];
        if (!(a == b || a == 1 || b == 1)) return -1;
        out_shape[i] = a > b ? a : b;
    }
    *out_n_dim = xa->n_dim > xb->n_dim ? xa->n_dim : xb->n_dim;
    *out_dtype = orc_promote(xa->dtype, xb->dtype);
    return 0;
}

/* Broadcast walker: dsc/include/dsc_iter.h:67-95 — stride 0 on every dim where
 * the operand is smaller than the output. */
static size_t orc_bcast_index(const orc_tensor *x, const int *out_shape, const int *out_idx) {
    size_t off = 0;
    for (int d = 0; d < ORC_MAX_DIMS; ++d)
        if (x->shape[d] >= out_shape[d]) off += (size_t) out_idx[d] * (size_t) x->stride[d];
    return off;
}

/* One loop per functor (as the reference instantiates binary_op<T, Op> per Op, dsc.cpp:1186-1245):
 * CPLX_RE / CPLX_IM / REAL are the expressions of dsc_ops.h:46-90 in terms of ar, ai, br, bi / va, vb. */
#define ORC_BINARY_LOOP(R, CPLX_RE, CPLX_IM, REAL)                                              \
    {                                                                                           \
    const R *a = (const R *) ca.data, *b = (const R *) cb.data;                                 \
    R *o = (R *) out->data;                                                                     \
    const int a_scalar = xa->n_dim == 1 && xa->shape[ORC_MAX_DIMS - 1] == 1;                    \
    const int b_scalar = xb->n_dim == 1 && xb->shape[ORC_MAX_DIMS - 1] == 1;                    \
    int idx[ORC_MAX_DIMS] = {0, 0, 0, 0};                                                       \
    for (int i = 0; i < out->ne; ++i) {                                                         \
        size_t ia, ib;                                                                          \
        if (a_scalar)      { ia = 0; ib = (size_t) i; }                                         \
        else if (b_scalar) { ia = (size_t) i; ib = 0; }                                         \
        else { ia = orc_bcast_index(&ca, out->shape, idx); ib = orc_bcast_index(&cb, out->shape, idx); } \
        if (cplx) {                                                                             \
            const R ar = a[2 * ia], ai = a[2 * ia + 1], br = b[2 * ib], bi = b[2 * ib + 1];     \
            o[2 * i]     = CPLX_RE;                                                             \
            o[2 * i + 1] = CPLX_IM;                                                             \
        } else {                                                                                \
            const R va = a[ia], vb = b[ib];                                                     \
            o[i] = REAL;                                                                        \
        }                                                                                       \
        for (int d = ORC_MAX_DIMS - 1; d >= 0; --d) {                                           \
            if (++idx[d] < out->shape[d]) break;                                                \
            idx[d] = 0;                                                                         \
        }                                                                                       \
    }                                                                                           \
    }

#define ORC_BINARY_BODY(R)                                                                                        \
    switch (op) {                                                                                                 \
        case ORC_ADD: ORC_BINARY_LOOP(R, ar + br, ai + bi, va + vb) break;                                        \
        case ORC_SUB: ORC_BINARY_LOOP(R, ar - br, ai - bi, va - vb) break;                                        \
        case ORC_MUL: ORC_BINARY_LOOP(R, (ar * br) - (ai * bi), (ar * bi) + (ai * br), va * vb) break;            \
        default:      ORC_BINARY_LOOP(R, ((ar * br) + (ai * bi)) / ((br * br) + (bi * bi)),                       \
                                      ((ai * br) - (ar * bi)) / ((br * br) + (bi * bi)), va / vb) break;          \
    }

/* dsc_add / dsc_sub / dsc_mul / dsc_div: dsc/src/dsc.cpp:1247-1297 via :44-69 and :1186-1245;
 * functors per dsc/include/dsc_ops.h:46-90.  Operands are first cast to the promoted dtype
 * (the reference does it in its scratch arena, :65-68). */
int orc_binary(const orc_tensor *xa, const orc_tensor *xb, orc_tensor *out, int op) {
    int shape[4], n_dim, dtype;
    if (orc_mul_out_shape(xa, xb, shape, &n_dim, &dtype)) return -1;
    if (out->dtype != dtype || memcmp(shape, out->shape, sizeof(shape))) return -1;

    orc_tensor ca = *xa, cb = *xb;
    void *tmp_a = NULL, *tmp_b = NULL;
    if (xa->dtype != dtype) {
        tmp_a = malloc(orc_dtype_size(dtype) * (size_t) xa->ne);
        ca.data = tmp_a; ca.dtype = dtype;
        orc_cast(xa, &ca);
    }
    if (xb->dtype != dtype) {
        tmp_b = malloc(orc_dtype_size(dtype) * (size_t) xb->ne);
        cb.data = tmp_b; cb.dtype = dtype;
        orc_cast(xb, &cb);
    }
    const int cplx = dtype == ORC_C32 || dtype == ORC_C64;
    if (dtype == ORC_F32 || dtype == ORC_C32) { ORC_BINARY_BODY(float) }
    else                                      { ORC_BINARY_BODY(double) }
    free(tmp_a);
    free(tmp_b);
    return 0;
}

int orc_mul(const orc_tensor *xa, const orc_tensor *xb, orc_tensor *out) { return orc_binary(xa, xb, out, ORC_MUL); }

/* ------------------------------------------------------------------ unary (spectrum consumers) */

int orc_unary_out_dtype(int dtype, int op) {
    if (op == ORC_CONJ) return dtype;
    return (dtype == ORC_F32 || dtype == ORC_C32) ? ORC_F32 : ORC_F64;     /* as_real, dsc.cpp:1487 */
}

/* One loop per functor and precision, as the reference instantiates complex_unary<Tin, Tout>(x, out, op). */
#define ORC_UNARY_BODY(R, SQRT, ATAN2)                                                          \
    {                                                                                           \
    const R *xd = (const R *) x->data;                                                          \
    R *od = (R *) out->data;                                                                    \
    for (int i = 0; i < x->ne; ++i) {                                                           \
        const R re = cplx ? xd[2 * i] : xd[i];                                                  \
        const R im = cplx ? xd[2 * i + 1] : (R) 0;                                              \
        switch (op) {                                                                           \
            case ORC_ABS:      od[i] = cplx ? SQRT((re * re) + (im * im)) : (re >= 0 ? re : -re); break;   /* dsc_ops.h:273-286 */ \
            case ORC_ANGLE:    od[i] = cplx ? ATAN2(im, re) : ATAN2((R) 0, re); break;                       /* dsc_ops.h:288-303 */ \
            case ORC_CONJ:     if (cplx) { od[2 * i] = re; od[2 * i + 1] = -im; } else od[i] = re; break;     /* dsc_ops.h:242-249 */ \
            case ORC_REALPART: od[i] = re; break;                                                             /* dsc_ops.h:251-258 */ \
            default:           od[i] = im; break;                                                             /* dsc_ops.h:260-271 */ \
        }                                                                                       \
    }                                                                                           \
    }

/* dsc/src/dsc.cpp:1480-1622 */
int orc_unary(const orc_tensor *x, orc_tensor *out, int op) {
    if (out->dtype != orc_unary_out_dtype(x->dtype, op) || out->ne != x->ne) return -1;
    const int cplx = x->dtype == ORC_C32 || x->dtype == ORC_C64;
    if (x->dtype == ORC_F32 || x->dtype == ORC_C32) ORC_UNARY_BODY(float, sqrtf, atan2f)
    else                                            ORC_UNARY_BODY(double, sqrt, atan2)
    return 0;
}

/* ------------------------------------------------------------------ reductions */

/* dsc/src/dsc.cpp:83-115.  (The reference fills the leading slots of the
 * keep_dims=false shape with memset(…, 1, …), i.e. 0x01010101 per int, which
 * is only observable through a user-supplied `out`; dsc_new_tensor rewrites
 * them to 1.  We produce 1.) */
int orc_reduce_out_shape(const orc_tensor *x, int axis, int keep_dims, int out_shape[4], int *out_n_dim) {
    const int slot = orc_axis_slot(x, axis);
    if (slot < 0 || slot >= ORC_MAX_DIMS) return -1;
    if (keep_dims) {
        memcpy(out_shape, x->shape, sizeof(int) * ORC_MAX_DIMS);
        out_shape[slot] = 1;
        *out_n_dim = x->n_dim;
    } else {
        const int nd = x->n_dim - 1;
        const int lead = ORC_MAX_DIMS - nd;
        for (int i = 0; i < lead; ++i) out_shape[i] = 1;
        int o = 0;
        for (int i = ORC_MAX_DIMS - x->n_dim; i < ORC_MAX_DIMS; ++i) {
            if (i == slot) continue;
            out_shape[lead + o++] = x->shape[i];
        }
        *out_n_dim = nd;
    }
    return 0;
}

#define ORC_REDUCE_BODY(R)                                                                      \
    const R *xd = (const R *) x->data;                                                          \
    R *od = (R *) out->data;                                                                    \
    const R inf = (R) INFINITY;                                                                 \
    orc_lines it;                                                                               \
    orc_lines_begin(&it, x, slot);                                                              \
    for (int i = 0; i < out->ne && orc_lines_next(&it); ++i) {                                  \
        R acc_r, acc_i;                                                                         \
        if (op == ORC_MAX)      acc_r = acc_i = -inf;                                           \
        else if (op == ORC_MIN) acc_r = acc_i = inf;                                            \
        else                    acc_r = acc_i = (R) 0;                                          \
        for (int j = 0; j < axis_n; ++j) {                                                      \
            const size_t idx = (size_t) it.base + (size_t) j * (size_t) it.step;                \
            const R vr = cplx ? xd[2 * idx] : xd[idx];                                          \
            const R vi = cplx ? xd[2 * idx + 1] : (R) 0;                                        \
            if (op == ORC_MAX) {                                                                \
                /* max_op: xa > xb ? xa : xb, complex on .real (dsc_ops.h:318-328) */           \
                if (!(acc_r > vr)) { acc_r = vr; acc_i = vi; }                                  \
            } else if (op == ORC_MIN) {                                                         \
                /* min_op real: xa < xb ? xa : xb; complex: xa.real > xb.real ? xb : xa */      \
                if (cplx) { if (acc_r > vr)    { acc_r = vr; acc_i = vi; } }                    \
                else      { if (!(acc_r < vr)) { acc_r = vr; } }                                \
            } else {                                                                            \
                acc_r = acc_r + vr;                                                             \
                acc_i = acc_i + vi;                                                             \
            }                                                                                   \
        }                                                                                       \
        if (op == ORC_MEAN) {                                                                   \
            /* dsc.cpp:1837-1853: multiply by (1/axis_n [, 0]) with mul_op */                   \
            const R s = (R) 1 / (R) axis_n;                                                     \
            if (cplx) {                                                                         \
                const R r = (acc_r * s) - (acc_i * (R) 0);                                      \
                const R m = (acc_r * (R) 0) + (acc_i * s);                                      \
                acc_r = r; acc_i = m;                                                           \
            } else {                                                                            \
                acc_r = acc_r * s;                                                              \
            }                                                                                   \
        }                                                                                       \
        if (cplx) { od[2 * i] = acc_r; od[2 * i + 1] = acc_i; }                                 \
        else      { od[i] = acc_r; }                                                            \
    }

/* dsc/src/dsc.cpp:1774-1953: one sequential left-to-right accumulation in T per
 * output element, lines taken in axis-iterator order. */
int orc_reduce(const orc_tensor *x, orc_tensor *out, int axis, int op) {
    const int slot = orc_axis_slot(x, axis);
    if (slot < 0 || slot >= ORC_MAX_DIMS || out->dtype != x->dtype) return -1;
    const int axis_n = x->shape[slot];
    const int cplx = x->dtype == ORC_C32 || x->dtype == ORC_C64;
    if (x->dtype == ORC_F32 || x->dtype == ORC_C32) { ORC_REDUCE_BODY(float) }
    else                                            { ORC_REDUCE_BODY(double) }
    return 0;
}
