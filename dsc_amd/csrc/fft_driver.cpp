// fft_driver.cpp — dsc_plan_fft / dsc_fft / dsc_ifft / dsc_rfft / dsc_irfft / dsc_filter_fft.
//
// Host-side mirror of the reference drivers: plan cache dsc/src/dsc.cpp:182-267, shape and
// dtype rules :2009-2071 (fft) and :2173-2244 (rfft), public entry points :2073-2100,
// :2246-2260.  Where the reference loops over lines on the host (exec_fft :1958-2007,
// exec_rfft :2102-2171), this file picks a kernel path for the whole batch:
//
//   r2c_64k_regs / c2r_64k_regs   f32, 65536-point real transform of contiguous rows:
//                                  register-resident, one HBM round trip (fft_r2c_64k.hip)
//   r2c_2pass_regs / c2r_2pass_... real transforms of 65536 (f64) .. 524288 points, contiguous rows: two passes over HBM,
//                                  register-resident rows and column kernels (fft_r2c_2pass.hip)
//   regs_mid                       contiguous full lines, complex length 256 .. 16384 (f32, f64; f32 complex
//                                  also 32768): register-resident, one HBM round trip (fft_regs_mid.hip)
//   generic_lds                    any axis / padding, complex length <= dsc_fft_lds_max_len: one pass (fft_generic.hip)
//   generic_4step                  longer: pack -> columns(+twiddle) -> rows -> post/unpack,
//                                  chunked over lines to fit the scratch arena
#include "dsc_internal.h"
#include "kernels.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

// exp(-2 pi i k / n) with exact values on the axes: quarter-turn reduction, long double.
static void unit_root(long long k, long long n, long double *c, long double *s) {
    k %= n;
    const long long q = (4 * k) / n;
    const long long r = 4 * k - q * n;
    const long double a = 1.57079632679489661923132169163975144L * (long double) r / (long double) n;
    const long double cr = r == 0 ? 1.0L : cosl(a), sr = r == 0 ? 0.0L : sinl(a);
    switch (q) {
        case 0:  *c = cr;  *s = -sr; break;
        case 1:  *c = -sr; *s = -cr; break;
        case 2:  *c = -cr; *s = sr;  break;
        default: *c = sr;  *s = cr;  break;
    }
}

template<typename T>
static void fill_roots(T *dst, long long count, long long n) {
    for (long long k = 0; k < count; ++k) {
        long double c, s;
        unit_root(k, n, &c, &s);
        dst[2 * k] = (T) c;
        dst[2 * k + 1] = (T) s;
    }
}

// dsc.cpp:182-216
static dsc_fft_plan *find_plan(dsc_ctx *ctx, int n, dsc_fft_type fft_type, dsc_dtype twd_dtype) {
    dsc_fft_plan *plan = nullptr;
    for (int i = 0; i < DSC_MAX_FFT_PLANS; ++i) {
        dsc_fft_plan *cached = ctx->fft_plans[i];
        if (cached == nullptr) continue;
        if (cached->n == n && cached->fft_type == fft_type && cached->dtype == twd_dtype) {
            plan = cached;
            plan->last_used = 0;
        } else {
            cached->last_used++;
        }
    }
    return plan;
}

// dsc.cpp:218-267.  The reference's table is the concatenation of every radix-2 stage's
// twiddles in the transform precision (dsc_fft.h:33-55); ours is one table of the n-th
// roots (every stage indexes it with a stride) plus, for REAL plans, the 2n-th roots of the
// packed-real pass, all rounded once from long double.
extern "C" dsc_fft_plan *dsc_plan_fft(dsc_ctx *ctx, int n, dsc_fft_type fft_type, dsc_dtype dtype) {
    DSC_ASSERT(dtype < 4);
    DSC_ASSERT(fft_type == DSC_FFT_REAL || fft_type == DSC_FFT_COMPLEX);
    const int fft_n = dsc_pow2_n(n);
    const dsc_dtype twd = dsc_is_single(dtype) ? DSC_F32 : DSC_F64;

    dsc_fft_plan *plan = find_plan(ctx, fft_n, fft_type, twd);
    if (plan != nullptr) return plan;

    int slot = -1;
    for (int i = 0; i < DSC_MAX_FFT_PLANS; ++i) {
        if (ctx->fft_plans[i] == nullptr) { slot = i; break; }
    }
    if (slot < 0) {                                     // evict the least recently used
        int oldest = -1;
        for (int i = 0; i < DSC_MAX_FFT_PLANS; ++i) {
            if (ctx->fft_plans[i]->last_used > oldest) { oldest = ctx->fft_plans[i]->last_used; slot = i; }
        }
        ctx->main.free(ctx->fft_plans[slot]->block);
        delete ctx->fft_plans[slot];
        ctx->fft_plans[slot] = nullptr;
    }

    const size_t real_sz = twd == DSC_F32 ? 4 : 8;
    const bool regs64k = twd == DSC_F32 && fft_n == 32768;          // the 65536-point real and the 32768-point complex register kernels
    const size_t full_bytes = DSC_ALIGN_UP((size_t) fft_n * 2 * real_sz, DSC_DEVICE_ALIGN);
    const size_t real_bytes = fft_type == DSC_FFT_REAL ? DSC_ALIGN_UP(((size_t) fft_n + 1) * 2 * real_sz, DSC_DEVICE_ALIGN) : 0;
    const size_t aux_bytes = regs64k ? DSC_ALIGN_UP(dsc_r2c64k_table_bytes(), DSC_DEVICE_ALIGN) : 0;
    const size_t total = full_bytes + real_bytes + aux_bytes;

    std::vector<char> host(total, 0);
    if (twd == DSC_F32) {
        fill_roots((float *) host.data(), fft_n, fft_n);
        if (real_bytes) fill_roots((float *) (host.data() + full_bytes), (long long) fft_n + 1, 2LL * fft_n);
    } else {
        fill_roots((double *) host.data(), fft_n, fft_n);
        if (real_bytes) fill_roots((double *) (host.data() + full_bytes), (long long) fft_n + 1, 2LL * fft_n);
    }
    if (regs64k) dsc_r2c64k_build_tables(host.data() + full_bytes + real_bytes);

    plan = new dsc_fft_plan();
    plan->n = fft_n;
    plan->last_used = 0;
    plan->dtype = twd;
    plan->fft_type = fft_type;
    plan->block = ctx->main.alloc(total, true);          // long lived: from the top of the arena
    plan->tw_full = plan->block;
    plan->tw_real = real_bytes ? plan->block + full_bytes : nullptr;
    plan->tw_aux = aux_bytes ? plan->block + full_bytes + real_bytes : nullptr;
    HIP_CHECK(hipMemcpyAsync(plan->block, host.data(), total, hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));       // `host` dies at return
    ctx->fft_plans[slot] = plan;
    return plan;
}

// ---------------------------------------------------------------------------------------------

struct fft_job {
    const dsc_tensor *x;
    dsc_tensor *out;
    int slot;            // transformed axis (0..3)
    int in_len;          // valid samples to read along the axis (input element units)
    int L;               // complex transform length
    dsc_fft_mode mode;
    bool inverse;
    double scale;
};

static void lines_of(const dsc_tensor *t, int slot, long long *n_lines, long long *inner, dsc_line_layout *l) {
    long long in = 1;
    for (int i = slot + 1; i < DSC_MAX_DIMS; ++i) in *= t->shape[i];
    long long outer = 1;
    for (int i = 0; i < slot; ++i) outer *= t->shape[i];
    *inner = in;
    *n_lines = outer * in;
    l->inner_stride = 1;
    l->elem_stride = in;
    l->outer_stride = (long long) t->shape[slot] * in;
}

static void run_four_step(dsc_ctx *ctx, const fft_job &j, bool sp, const dsc_fft_plan *real_plan) {
    const int lds_max = dsc_fft_lds_max_len(sp);
    const int L = j.L;
    // balanced split L = L1 * L2 (columns of length L1, then rows of length L2): both passes then
    // move >= 128-B pieces when their tiles take a cache line of neighbouring lines
    int log2l = 0;
    while ((1 << log2l) < L) ++log2l;
    int L1 = 1 << ((log2l + 1) / 2), L2 = L / L1;
    while (L1 > lds_max) { L1 >>= 1; L2 <<= 1; }
    DSC_ASSERT(L1 >= 2 && L2 >= 2 && L2 <= lds_max);
    const dsc_dtype cdt = sp ? DSC_C32 : DSC_C64;
    const dsc_fft_plan *p1 = dsc_plan_fft(ctx, L1, DSC_FFT_COMPLEX, cdt);
    const dsc_fft_plan *p2 = dsc_plan_fft(ctx, L2, DSC_FFT_COMPLEX, cdt);
    // W_L^m for the inter-pass twiddle: the COMPLEX plan of the full length (its table is gathered
    // from L2; computing it with sincospi in double cost more than the butterflies)
    const dsc_fft_plan *pl = L <= (1 << 22) ? dsc_plan_fft(ctx, L, DSC_FFT_COMPLEX, cdt) : nullptr;

    long long n_lines, inner_in, inner_out;
    dsc_line_layout lin, lout;
    lines_of(j.x, j.slot, &n_lines, &inner_in, &lin);
    lines_of(j.out, j.slot, &n_lines, &inner_out, &lout);

    const size_t csz = dsc_dtype_size(cdt);
    const size_t line_bytes = (size_t) L * csz;
    // Contiguous rows that need no padding can be read / written in place as arrays of complex:
    //   R2C: the 2L reals of a row ARE L packed complex samples;  C2R: likewise on the way out
    const int x_n = j.x->shape[j.slot], out_n = j.out->shape[j.slot];
    const bool direct_in = inner_in == 1 &&
        ((j.mode == DSC_MODE_R2C_PACKED && j.in_len == 2 * L && x_n == 2 * L) || (j.mode == DSC_MODE_C2C && j.in_len == L && x_n == L));
    const bool direct_out = inner_out == 1 &&
        ((j.mode == DSC_MODE_C2R_PACKED && out_n == 2 * L) || ((j.mode == DSC_MODE_C2C || j.mode == DSC_MODE_R2C_CAST) && out_n == L));

    ctx->scratch.reset();
    long long chunk = (long long) ((ctx->scratch.capacity() - 2 * DSC_DEVICE_ALIGN) / (2 * line_bytes));
    if (chunk < 1)
        DSC_LOG_FATAL("scratch arena too small: a %d-point transform needs %.1f MB of scratch", L, 2.0 * line_bytes / 1048576.);
    // Keep a chunk's working set (input rows + two work buffers + output rows) inside the 256 MiB
    // Infinity Cache, so that the intermediate of the column pass is still on-die when the row pass
    // reads it: ~4 line-sized buffers per row.
    {
        static long long cap_bytes = -1;
        if (cap_bytes < 0) {
            const char *e = getenv("DSC_4STEP_CHUNK_MB");
            cap_bytes = (e ? atoll(e) : 192) << 20;
        }
        long long by_cache = cap_bytes / (long long) (4 * line_bytes);
        if (by_cache < 1) by_cache = 1;
        if (chunk > by_cache) chunk = by_cache;
    }
    if (chunk > n_lines) chunk = n_lines;
    char *A = ctx->scratch.alloc((size_t) chunk * line_bytes);
    char *B = ctx->scratch.alloc((size_t) chunk * line_bytes);

    for (long long q = 0; q < n_lines; q += chunk) {
        const long long nl = n_lines - q < chunk ? n_lines - q : chunk;
        const char *src = A;
        if (direct_in) {
            src = (const char *) j.x->data + (size_t) q * line_bytes;
        } else if (j.mode == DSC_MODE_C2R_PACKED) {
            dsc_launch_fft_c2r_prepass(j.x->data, A, q, nl, inner_in, lin, L, j.in_len, real_plan->tw_real, sp, ctx->stream);
        } else {
            dsc_launch_fft_pack(j.x->data, A, q, nl, inner_in, lin, L, j.in_len, j.mode, sp, ctx->stream);
        }

        dsc_fft_lines_args a;
        // columns: L2 lines of length L1 per transform, times W_L^{j2 k1}
        a.in = src; a.out = B;
        a.n_lines = nl * L2; a.inner = L2;
        a.lin = a.lout = dsc_line_layout{L, 1, L2};
        a.L = L1; a.in_len = L1; a.inverse = j.inverse; a.scale = 1.0;
        a.tw = p1->tw_full; a.tw_real = nullptr; a.tw4_len = L; a.tw4 = pl ? pl->tw_full : nullptr;
        dsc_launch_fft_lines(a, DSC_MODE_C2C, sp, ctx->stream);
        // rows: L1 lines of length L2, output k1 + L1 k2
        char *dst = direct_out ? (char *) j.out->data + (size_t) q * line_bytes : A;
        a.in = B; a.out = dst;
        a.n_lines = nl * L1; a.inner = L1;
        a.lin = dsc_line_layout{L, L2, 1};
        a.lout = dsc_line_layout{L, 1, L1};
        a.L = L2; a.in_len = L2; a.tw = p2->tw_full; a.tw4_len = 0; a.tw4 = nullptr;
        a.scale = direct_out ? j.scale : 1.0;
        dsc_launch_fft_lines(a, DSC_MODE_C2C, sp, ctx->stream);

        if (j.mode == DSC_MODE_R2C_PACKED)
            dsc_launch_fft_r2c_postpass(A, j.out->data, q, nl, inner_out, lout, L, real_plan->tw_real, sp, ctx->stream);
        else if (!direct_out)
            dsc_launch_fft_unpack(A, j.out->data, q, nl, inner_out, lout, L, j.scale, j.mode, sp, ctx->stream);
    }
    ctx->last_fft_path = "generic_4step";
}

static void run_job(dsc_ctx *ctx, const fft_job &j) {
    const bool sp = dsc_is_single(j.out->dtype);
    const bool packed = j.mode == DSC_MODE_R2C_PACKED || j.mode == DSC_MODE_C2R_PACKED;
    const dsc_fft_plan *plan = dsc_plan_fft(ctx, j.L, packed ? DSC_FFT_REAL : DSC_FFT_COMPLEX, j.out->dtype);

    long long n_lines, inner;
    dsc_line_layout lin, lout;
    lines_of(j.x, j.slot, &n_lines, &inner, &lin);
    lines_of(j.out, j.slot, &n_lines, &inner, &lout);

    // Long complex transforms along a non-last axis (dsc_fft / dsc_ifft, complex or real input, full lines): four-step in two passes of
    // the column kernel — two streaming passes with whole tile rows instead of the three of the transpose route below; 4096-point lines
    // only from 64 columns (below that the one-pass kernel's 8-column tiles hold whole rows: 34 - 44 % against 10 - 27 %)
    // (33 - 34 % of the roofline against 20 - 22 %; 4096-point c32 lines: against 24 % for the one-pass column kernel with its 64-B pieces).  One full-size
    // temporary in the main arena.  (DSC_COLS_4STEP_MIN: smallest length that takes this route; 0 switches it off.)
    {
        static const long long min_4step = [] { const char *e = getenv("DSC_COLS_4STEP_MIN"); return e ? atoll(e) : 4096LL; }();
        const int x_n = j.x->shape[j.slot], out_n = j.out->shape[j.slot];
        int n1 = 0, n2 = 0;
        if (inner >= 8 && min_4step > 0 && j.L >= min_4step && (j.mode == DSC_MODE_C2C || j.mode == DSC_MODE_R2C_CAST) && x_n == j.L && j.in_len == j.L &&
            out_n == j.L && j.L <= (1 << 22) && (j.L > 4096 || inner >= 64) && dsc_fft_cols_4step_split(j.L, sp, (int) inner, &n1, &n2)) {
            const size_t csz = dsc_dtype_size(j.out->dtype);
            const size_t slice_bytes = (size_t) j.L * inner * csz;                     // pass 1 addresses a whole [n][inner] slice with 32-bit offsets
            const size_t work_bytes = (size_t) j.out->ne * csz;
            if (slice_bytes < 0x7f000000u && (long long) n1 * inner < (1LL << 30) && (n_lines / inner) * n2 < (1LL << 31) && ctx->main.fits(work_bytes, 0)) {
                const dsc_dtype cdt = j.out->dtype;
                const dsc_fft_plan *p1 = dsc_plan_fft(ctx, n1, DSC_FFT_COMPLEX, cdt);
                const dsc_fft_plan *p2 = dsc_plan_fft(ctx, n2, DSC_FFT_COMPLEX, cdt);
                const dsc_fft_plan *pn = dsc_plan_fft(ctx, j.L, DSC_FFT_COMPLEX, cdt);
                dsc_tensor *work = dsc_new_tensor(ctx, j.out->n_dim, &j.out->shape[DSC_MAX_DIMS - j.out->n_dim], cdt, nullptr);
                dsc_launch_fft_cols_4step(j.x->data, work->data, j.out->data, n_lines / inner, (int) inner, n1, n2, j.mode, j.inverse, sp, p1->tw_full,
                                          p2->tw_full, pn->tw_full, j.scale, ctx->stream);
                dsc_tensor_free(ctx, work);              // stream ordered: whoever reuses the block is enqueued after these launches
                ctx->last_fft_path = "cols_4step";
                return;
            }
        }
    }

    // Long REAL transforms along a non-last axis (dsc_rfft / dsc_irfft, full lines, an even number of columns): two neighbouring columns
    // as one complex column through the same four-step, the spectra separated / merged inside its passes (fft_regs_cols.hip) — two
    // streaming passes instead of the transpose route's three.  (DSC_COLS_4STEP_REAL_MIN: smallest real length; 0 switches it off.)
    {
        static const long long min_real = [] { const char *e = getenv("DSC_COLS_4STEP_REAL_MIN"); return e ? atoll(e) : 8192LL; }();
        const int x_n = j.x->shape[j.slot], out_n = j.out->shape[j.slot];
        const long long n = 2LL * j.L;
        const bool full = j.mode == DSC_MODE_R2C_PACKED ? (x_n == n && j.in_len == n && out_n == j.L + 1)
                                                        : (x_n == j.L + 1 && j.in_len == j.L + 1 && out_n == n);
        int n1 = 0, n2 = 0;
        if (packed && full && inner >= 16 && inner % 2 == 0 && min_real > 0 && n >= min_real && n <= (1 << 22) &&
            dsc_fft_cols_4step_split((int) n, sp, 1 << 20, &n1, &n2) && n1 >= 64 && n2 >= 64) {   // balanced: narrower pass-2 tiles would widen the merge kernel's (measured)
            const dsc_dtype cdt = sp ? DSC_C32 : DSC_C64;
            const size_t csz = dsc_dtype_size(cdt);
            const long long slices = n_lines / inner, cc_n = inner / 2;
            const size_t real_slice = (size_t) n * inner * (csz / 2), bins_slice = (size_t) (j.L + 1) * inner * csz;
            const size_t work_bytes = (size_t) slices * n * cc_n * csz;
            if (real_slice < 0x7f000000u && bins_slice < 0x7f000000u && slices * n < (1LL << 31) && slices * n2 < (1LL << 31) &&
                (long long) n1 * cc_n < (1LL << 30) && ctx->main.fits(work_bytes, 0)) {
                const dsc_fft_plan *p1 = dsc_plan_fft(ctx, n1, DSC_FFT_COMPLEX, cdt);
                const dsc_fft_plan *p2 = dsc_plan_fft(ctx, n2, DSC_FFT_COMPLEX, cdt);
                const dsc_fft_plan *pn = dsc_plan_fft(ctx, (int) n, DSC_FFT_COMPLEX, cdt);
                const int shape_w[2] = {(int) (slices * n), (int) cc_n};
                dsc_tensor *work = dsc_new_tensor(ctx, 2, shape_w, cdt, nullptr);
                if (j.mode == DSC_MODE_R2C_PACKED)
                    dsc_launch_rfft_cols_4step(j.x->data, work->data, j.out->data, slices, (int) cc_n, n1, n2, sp, p1->tw_full, p2->tw_full, pn->tw_full,
                                               ctx->stream);
                else                                     // j.scale = 2 / n folds the halves of the packed pre-pass (dsc_fft.h:232); here: 1 / n
                    dsc_launch_irfft_cols_4step(j.x->data, work->data, j.out->data, slices, (int) cc_n, n1, n2, sp, p1->tw_full, p2->tw_full,
                                                pn->tw_full, 0.5 * j.scale, ctx->stream);
                dsc_tensor_free(ctx, work);              // stream ordered
                ctx->last_fft_path = "cols_4step_real";
                return;
            }
        }
    }

    // Strided lines of complex length 32 .. 2048 (4096): the column kernel (lanes = neighbouring lines), one pass over HBM.
    static const bool cols_off = getenv("DSC_NO_COLS") != nullptr;            // A/B aid (tools/bench_axis0.py)
    const bool tiny_cols = dsc_fft_tiny_supports(j.L) && getenv("DSC_NO_TINY") == nullptr;
    if (inner > 1 && !cols_off && inner < (1LL << 30) && (tiny_cols || dsc_fft_regs_cols_supports(j.L, j.mode, sp))) {
        const int x_n = j.x->shape[j.slot], out_n = j.out->shape[j.slot];
        const size_t in_slice = (size_t) x_n * inner * dsc_dtype_size(j.x->dtype), out_slice = (size_t) out_n * inner * dsc_dtype_size(j.out->dtype);
        // largest offsets the kernel forms: over the transform length, not the axis length (zero padding reads past a short axis)
        const size_t rows_in = j.mode == DSC_MODE_R2C_PACKED ? 2 * (size_t) j.L : j.mode == DSC_MODE_C2R_PACKED ? (size_t) j.L + 1 : (size_t) j.L;
        const size_t span = rows_in * inner * dsc_dtype_size(j.x->dtype);
        if (in_slice < 0x7f000000u && out_slice < 0x7f000000u && span < 0x7f000000u) {     // 32-bit buffer offsets
            if (tiny_cols) {
                dsc_launch_fft_tiny_cols(j.x->data, j.out->data, n_lines / inner, (int) inner, j.L, j.mode, j.inverse, sp, j.scale, x_n,
                                         j.in_len < x_n ? j.in_len : x_n, out_n, ctx->stream);
                ctx->last_fft_path = "regs_tiny_cols";
                return;
            }
            dsc_launch_fft_regs_cols(j.x->data, j.out->data, n_lines / inner, (int) inner, j.L, j.mode, j.inverse, sp, plan->tw_full, plan->tw_real,
                                     j.scale, x_n, j.in_len < x_n ? j.in_len : x_n, out_n, ctx->stream);
            ctx->last_fft_path = "regs_cols";
            return;
        }
    }

    // Strided lines (a transform along a non-last axis) of a length the register kernels cover: transpose the axis to the
    // back (32 x 32 LDS tiles), transform contiguous rows, transpose the result back — three streaming passes instead of
    // one latency-bound strided pass (measured 1.3-4x faster from 512 points up; below that the strided LDS kernel wins).
    static const bool via_transpose_off = getenv("DSC_NO_AXIS_TRANSPOSE") != nullptr;
    const bool is_cast = j.mode == DSC_MODE_R2C_CAST;
    const bool last_axis_kernel =                          // is there a register kernel for contiguous rows of this length and mode?
        dsc_fft_regs_mid_supports(j.L, j.mode, sp) || ((packed || j.mode == DSC_MODE_C2C) && dsc_fft_two_pass_supports(j.L, sp)) ||
        (sp && j.L == 32768) || dsc_fft_fused_l2_supports(j.L, sp, packed, j.inverse) || (is_cast && j.L == 262144);
    if (inner > 1 && !via_transpose_off && j.L >= 512 && last_axis_kernel) {
        const int x_n = j.x->shape[j.slot], out_n = j.out->shape[j.slot];
        const long long outer = n_lines / inner;
        // the route needs two full-size temporaries in the main arena; a context sized for x and out only keeps the strided
        // LDS kernel below (the reference needs two line-sized scratch buffers for the same call, dsc.cpp:2115-2116)
        const bool room = ctx->main.fits((size_t) j.x->ne * dsc_dtype_size(j.x->dtype), (size_t) j.out->ne * dsc_dtype_size(j.out->dtype));
        if (outer * inner < (1LL << 31) && room) {
            const int shape_in[2] = {(int) (outer * inner), x_n}, shape_out[2] = {(int) (outer * inner), out_n};
            dsc_tensor *t_in = dsc_new_tensor(ctx, 2, shape_in, j.x->dtype, nullptr);
            dsc_tensor *t_out = dsc_new_tensor(ctx, 2, shape_out, j.out->dtype, nullptr);
            dsc_launch_transpose_last2(j.x->data, t_in->data, (int) dsc_dtype_size(j.x->dtype), outer, x_n, (int) inner, ctx->stream);
            fft_job j2 = j;
            j2.x = t_in;
            j2.out = t_out;
            j2.slot = DSC_MAX_DIMS - 1;
            run_job(ctx, j2);
            dsc_launch_transpose_last2(t_out->data, j.out->data, (int) dsc_dtype_size(j.out->dtype), outer, (int) inner, out_n, ctx->stream);
            dsc_tensor_free(ctx, t_in);                  // stream ordered: whoever reuses the blocks is enqueued after these launches
            dsc_tensor_free(ctx, t_out);
            return;
        }
    }

    // register-resident 32768-point complex transform (c32 rows)
    if (sp && (j.mode == DSC_MODE_C2C || j.mode == DSC_MODE_R2C_CAST) && j.L == 32768 && inner == 1 && plan->tw_aux != nullptr) {
        dsc_launch_fft32k_c32(j.x->data, j.out->data, (int) n_lines, j.x->shape[j.slot], j.in_len, j.inverse, j.mode == DSC_MODE_R2C_CAST, plan->tw_aux, ctx->n_cu,
                              ctx->stream);
        ctx->last_fft_path = "c2c_32k_regs";
        return;
    }
    // register-resident 65536-point real transforms: contiguous full rows only
    if (sp && packed && j.L == 32768 && inner == 1 && plan->tw_aux != nullptr) {
        if (j.mode == DSC_MODE_R2C_PACKED) {               // any row length: shorter rows are zero padded, longer ones cropped
            dsc_launch_rfft64k((const float *) j.x->data, j.out->data, (int) n_lines, j.x->shape[j.slot], j.in_len, plan->tw_aux, ctx->n_cu,
                               ctx->stream);
            ctx->last_fft_path = "r2c_64k_regs";
            return;
        }
        if (j.mode == DSC_MODE_C2R_PACKED) {
            dsc_launch_irfft64k(j.x->data, (float *) j.out->data, (int) n_lines, j.x->shape[j.slot], j.in_len, plan->tw_aux, ctx->n_cu,
                                ctx->stream);
            ctx->last_fft_path = "c2r_64k_regs";
            return;
        }
    }

    // 65536-point complex rows (real length 131072) and 131072-point f64 rows (config 5): one launch, the four-step intermediate stays in the XCD-local L2
    // (fft_xcd_fused.hip)
    static const bool fused_off = getenv("DSC_NO_FUSED_L2") != nullptr;           // A/B aid
    const bool fused_cplx = !packed;
    const bool fused_cast = j.mode == DSC_MODE_R2C_CAST;                           // dsc_fft / dsc_ifft of a real tensor: widened while loading
    const bool fused_fwd = fused_cplx ? !j.inverse : j.mode == DSC_MODE_R2C_PACKED;
    if ((packed || j.mode == DSC_MODE_C2C || fused_cast) && inner == 1 && !fused_off && dsc_fft_fused_l2_supports(j.L, sp, packed, !fused_fwd) &&
        ctx->scratch.capacity() >= dsc_fft_fused_l2_scratch_bytes(j.L, sp) + DSC_DEVICE_ALIGN) {
        const bool cplx = fused_cplx, fwd = fused_fwd;
        ctx->scratch.reset();
        char *blk = ctx->scratch.alloc(dsc_fft_fused_l2_scratch_bytes(j.L, sp));
        if (ctx->async_error == nullptr) {
            DSC_KERNEL_CHECK(hipHostMalloc((void **) &ctx->async_error, sizeof(unsigned), hipHostMallocDefault));
            *ctx->async_error = 0;
        }
        if (dsc_launch_fft_fused_l2(j.x->data, j.out->data, n_lines, j.L, packed, !fwd, fused_cast, sp, blk, ctx->async_error, plan->tw_full, plan->tw_real,
                                    j.x->shape[j.slot], j.in_len, ctx->stream)) {
            ctx->last_fft_path = cplx ? "c2c_fused_l2" : fwd ? "r2c_fused_l2" : "c2r_fused_l2";
            return;
        }
    }

    // long transforms of contiguous rows (config 5 = f64 N = 262144): two passes over HBM, rows kernel + column kernel with
    // the real pass fused (fft_r2c_2pass.hip)
    static const bool two_pass_off = getenv("DSC_NO_TWO_PASS") != nullptr;        // A/B aid (tools/bench_mid.py)
    const bool two_pass_cast = j.mode == DSC_MODE_R2C_CAST && j.L == 262144;        // dsc_fft / dsc_ifft of a real tensor (that length only)
    // dsc_fft / dsc_ifft of a REAL tensor at the other two-pass lengths (524288, 1048576 points): widen the rows into a complex
    // temporary (one streaming pass, 79 % of the roofline) and take the complex two-pass route — 6 % of the roofline on the
    // generic four-step path otherwise.  Needs room for the temporary in the main arena (non-fatal probe).
    if (j.mode == DSC_MODE_R2C_CAST && !two_pass_cast && inner == 1 && !two_pass_off && dsc_fft_two_pass_supports(j.L, sp)) {
        const long long x_n = j.x->shape[j.slot];
        const size_t tmp_bytes = (size_t) n_lines * (size_t) x_n * (sp ? 8 : 16);
        if (n_lines * x_n < (1LL << 31) && n_lines < (1LL << 31) && ctx->main.fits(tmp_bytes)) {
            // x's own shape in the complex dtype: the axis slot stays where it is, so that the recursive call counts the lines of
            // `wide` and of `out` the same way also when trailing unit dimensions follow the axis ([B, N, 1], axis 1)
            dsc_tensor *wide = dsc_new_tensor(ctx, j.x->n_dim, &j.x->shape[DSC_MAX_DIMS - j.x->n_dim], sp ? DSC_C32 : DSC_C64, nullptr);
            dsc_launch_cast(j.x->data, j.x->dtype, wide->data, wide->dtype, n_lines * x_n, ctx->stream);
            fft_job j2 = j;
            j2.x = wide;
            j2.mode = DSC_MODE_C2C;
            run_job(ctx, j2);
            dsc_tensor_free(ctx, wide);                  // stream ordered
            return;
        }
    }
    if ((packed || j.mode == DSC_MODE_C2C || two_pass_cast) && inner == 1 && !two_pass_off && dsc_fft_two_pass_supports(j.L, sp)) {
        const int L = j.L;
        const bool cplx = !packed;                                        // dsc_fft / dsc_ifft of a complex tensor
        const bool fwd = cplx ? !j.inverse : j.mode == DSC_MODE_R2C_PACKED, inv = !fwd;   // any row length: padded / cropped by the row descriptors
        const long long x_n = j.x->shape[j.slot];
        {
            const size_t csz = sp ? 8 : 16;
            const size_t row_bytes = (size_t) L * csz;
            const size_t in_el = two_pass_cast ? csz / 2 : csz;                   // bytes per input element of a complex transform
            const size_t real_row = cplx ? (fwd ? (size_t) x_n * in_el : (size_t) L * csz) : (size_t) (fwd ? x_n : 2 * L) * (csz / 2);       // time-domain side
            const size_t bins_row = cplx ? (fwd ? (size_t) L * csz : (size_t) x_n * in_el) : (size_t) (fwd ? L + 1 : x_n) * csz;               // frequency-domain side
            ctx->scratch.reset();
            long long chunk = (long long) ((ctx->scratch.capacity() - DSC_DEVICE_ALIGN) / row_bytes);
            if (chunk < 1) DSC_LOG_FATAL("scratch arena too small: a %d-point transform needs %.1f MB of scratch per row", 2 * L, row_bytes / 1048576.);
            {
                // rows per launch sequence: as many as the scratch arena holds (cutting the batch into 64-row launches costs 15 %)
                static long long cap_rows = -1;
                if (cap_rows < 0) {
                    const char *e = getenv("DSC_2PASS_CHUNK_ROWS");
                    cap_rows = e ? atoll(e) : (1LL << 40);
                    if (cap_rows < 1) cap_rows = 1;
                }
                if (chunk > cap_rows) chunk = cap_rows;
            }
            if (chunk > n_lines) chunk = n_lines;
            char *work = ctx->scratch.alloc((size_t) chunk * row_bytes);
            for (long long q = 0; q < n_lines; q += chunk) {
                const long long nl = n_lines - q < chunk ? n_lines - q : chunk;
                const char *src = (const char *) j.x->data + (size_t) q * (fwd ? real_row : bins_row);
                char *dst = (char *) j.out->data + (size_t) q * (fwd ? bins_row : real_row);
                if (cplx)
                    dsc_launch_fft_two_pass(src, dst, nl, L, inv, two_pass_cast, sp, work, plan->tw_full, x_n, j.in_len, ctx->stream);
                else
                    dsc_launch_rfft_two_pass(src, dst, nl, L, inv, sp, work, plan->tw_full, plan->tw_real, x_n, j.in_len, ctx->stream);
            }
            ctx->last_fft_path = cplx ? "c2c_2pass_regs" : fwd ? "r2c_2pass_regs" : "c2r_2pass_regs";
            return;
        }
    }

    // complex lengths 2 .. 16: one thread per line (fft_tiny.hip)
    static const bool tiny_off = getenv("DSC_NO_TINY") != nullptr;                // A/B aid
    if (inner == 1 && !tiny_off && dsc_fft_tiny_supports(j.L)) {
        const int x_n = j.x->shape[j.slot];
        const int want = j.mode == DSC_MODE_R2C_PACKED ? 2 * j.L : j.mode == DSC_MODE_C2R_PACKED ? j.L + 1 : j.L;
        const bool full = j.in_len == want && x_n == want;
        if (full || (long long) x_n * 16 * 256 < (1LL << 30)) {                  // byte offsets of a padded group fit 32 bits
            dsc_launch_fft_tiny(j.x->data, j.out->data, n_lines, j.L, j.mode, j.inverse, sp, j.scale, full ? -1 : x_n, j.in_len, ctx->stream);
            ctx->last_fft_path = "regs_tiny";
            return;
        }
    }

    // register-resident mid sizes: contiguous full lines along the last axis
    static const bool regs_mid_off = getenv("DSC_NO_REGS_MID") != nullptr;      // A/B aid (tools/bench_mid.py)
    if (inner == 1 && !regs_mid_off && dsc_fft_regs_small_supports(j.L)) {            // 32 .. 256 points: LDS-staged register kernel
        const int x_n = j.x->shape[j.slot];
        const int want = j.mode == DSC_MODE_R2C_PACKED ? 2 * j.L : j.mode == DSC_MODE_C2R_PACKED ? j.L + 1 : j.L;
        const bool full = j.in_len == want && x_n == want;
        // zero padded / cropped lines (frames of 200 samples transformed at 256 ...): the same kernel gathers line by line; 256-point
        // lines take the mid kernel's PAD form below; the byte offsets of a group must fit 32 bits
        if (full || (long long) x_n * 16 * 256 < (1LL << 30)) {
            dsc_launch_fft_regs_mid(j.x->data, j.out->data, n_lines, j.L, j.mode, j.inverse, sp, plan->tw_full, plan->tw_real, j.scale,
                                    full ? -1 : x_n, j.in_len, ctx->stream);
            ctx->last_fft_path = "regs_small";
            return;
        }
    }
    if (inner == 1 && !regs_mid_off && dsc_fft_regs_mid_supports(j.L, j.mode, sp)) {
        const int x_n = j.x->shape[j.slot];
        const int want = j.mode == DSC_MODE_R2C_PACKED ? 2 * j.L : j.mode == DSC_MODE_C2R_PACKED ? j.L + 1 : j.L;
        const bool full = j.in_len == want && x_n == want;                    // else: zero padded or cropped lines
        // byte pitch of a padded group must fit the kernel's 32-bit offsets
        if (full || (long long) x_n * 16 * 64 < (1LL << 30)) {
            dsc_launch_fft_regs_mid(j.x->data, j.out->data, n_lines, j.L, j.mode, j.inverse, sp, plan->tw_full, plan->tw_real, j.scale,
                                    full ? -1 : x_n, j.in_len, ctx->stream);
            ctx->last_fft_path = "regs_mid";
            return;
        }
    }

    if (j.L <= dsc_fft_lds_max_len(sp)) {
        dsc_fft_lines_args a;
        a.in = j.x->data; a.out = j.out->data;
        a.n_lines = n_lines; a.inner = inner;
        a.lin = lin; a.lout = lout;
        a.L = j.L; a.in_len = j.in_len; a.inverse = j.inverse; a.scale = j.scale;
        a.tw = plan->tw_full; a.tw_real = plan->tw_real; a.tw4_len = 0; a.tw4 = nullptr;
        dsc_launch_fft_lines(a, j.mode, sp, ctx->stream);
        ctx->last_fft_path = "generic_lds";
        return;
    }
    run_four_step(ctx, j, sp, plan);
}

static dsc_tensor *make_out(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, const int *out_shape, dsc_dtype out_dtype) {
    if (out == nullptr)
        return dsc_new_tensor(ctx, x->n_dim, &out_shape[DSC_MAX_DIMS - x->n_dim], out_dtype, nullptr);
    DSC_ASSERT(out->dtype == out_dtype);
    DSC_ASSERT(out->n_dim == x->n_dim);
    DSC_ASSERT(memcmp(out_shape, out->shape, DSC_MAX_DIMS * sizeof(int)) == 0);
    return out;
}

// dsc.cpp:2009-2071
static dsc_tensor *internal_fft(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis, bool forward) {
    DSC_ASSERT(x != nullptr);
    dsc_trace_scope trace__(ctx, forward ? "dsc_fft" : "dsc_ifft", "op;fft", x, nullptr, n, axis);
    const int slot = dsc_axis_slot(x, axis);
    DSC_ASSERT(slot >= 0 && slot < DSC_MAX_DIMS);
    const int x_n = x->shape[slot];
    n = n > 0 ? dsc_pow2_n(n) : dsc_pow2_n(x_n);

    int out_shape[DSC_MAX_DIMS];
    for (int i = 0; i < DSC_MAX_DIMS; ++i) out_shape[i] = i != slot ? x->shape[i] : n;
    dsc_dtype out_dtype = x->dtype;
    if (x->dtype == DSC_F32) out_dtype = DSC_C32;
    else if (x->dtype == DSC_F64) out_dtype = DSC_C64;
    out = make_out(ctx, x, out, out_shape, out_dtype);

    fft_job j;
    j.x = x; j.out = out; j.slot = slot;
    j.in_len = x_n < n ? x_n : n;
    j.L = n;
    j.mode = dsc_is_complex(x->dtype) ? DSC_MODE_C2C : DSC_MODE_R2C_CAST;
    j.inverse = !forward;
    j.scale = forward ? 1.0 : 1.0 / (double) n;          // dsc_fft.h:168-175
    run_job(ctx, j);
    return out;
}

// dsc.cpp:2173-2244
static dsc_tensor *internal_rfft(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis, bool forward) {
    DSC_ASSERT(x != nullptr);
    dsc_trace_scope trace__(ctx, forward ? "dsc_rfft" : "dsc_irfft", "op;fft", x, nullptr, n, axis);
    const int slot = dsc_axis_slot(x, axis);
    DSC_ASSERT(slot >= 0 && slot < DSC_MAX_DIMS);
    const int x_n = x->shape[slot];

    int out_n, order;
    dsc_dtype out_dtype;
    if (forward) {
        order = (n > 0 ? dsc_pow2_n(n) : dsc_pow2_n(x_n)) >> 1;
        out_n = order + 1;
        if (x->dtype == DSC_F32) out_dtype = DSC_C32;
        else if (x->dtype == DSC_F64) out_dtype = DSC_C64;
        else DSC_LOG_FATAL("RFFT input must be real");
    } else {
        DSC_ASSERT((n > 0 ? n : x_n) > 1);
        order = n > 0 ? dsc_pow2_n(n - 1) : dsc_pow2_n(x_n - 1);
        out_n = order << 1;
        if (x->dtype == DSC_C32) out_dtype = DSC_F32;
        else if (x->dtype == DSC_C64) out_dtype = DSC_F64;
        else DSC_LOG_FATAL("IRFFT input must be complex");
    }
    DSC_ASSERT(order >= 1);            // the reference asserts n > 0 in dsc_fft_storage (dsc_fft.h:112)

    int out_shape[DSC_MAX_DIMS];
    for (int i = 0; i < DSC_MAX_DIMS; ++i) out_shape[i] = i != slot ? x->shape[i] : out_n;
    out = make_out(ctx, x, out, out_shape, out_dtype);

    fft_job j;
    j.x = x; j.out = out; j.slot = slot;
    j.L = order;
    j.inverse = !forward;
    if (forward) {
        j.mode = DSC_MODE_R2C_PACKED;
        j.in_len = x_n < 2 * order ? x_n : 2 * order;         // dsc.cpp:2121, 2125-2133
        j.scale = 1.0;
    } else {
        j.mode = DSC_MODE_C2R_PACKED;
        j.in_len = x_n < order + 1 ? x_n : order + 1;         // dsc.cpp:2145, 2149-2157
        j.scale = 2.0 / (double) (order << 1);                 // dsc_fft.h:232
    }
    run_job(ctx, j);
    return out;
}

extern "C" dsc_tensor *dsc_fft(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis) {
    return internal_fft(ctx, x, out, n, axis, true);
}
extern "C" dsc_tensor *dsc_ifft(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis) {
    return internal_fft(ctx, x, out, n, axis, false);
}
extern "C" dsc_tensor *dsc_rfft(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis) {
    return internal_rfft(ctx, x, out, n, axis, true);
}
extern "C" dsc_tensor *dsc_irfft(dsc_ctx *ctx, const dsc_tensor *x, dsc_tensor *out, int n, int axis) {
    return internal_rfft(ctx, x, out, n, axis, false);
}

// README.md:113-135 as one call.  Fused kernel for the 65536-point f32 case, otherwise the
// three-operator composition the reference's users write by hand.
extern "C" dsc_tensor *dsc_filter_fft(dsc_ctx *ctx, const dsc_tensor *s, const dsc_tensor *H, dsc_tensor *out) {
    DSC_ASSERT(s != nullptr && H != nullptr);
    DSC_TRACE_OP(ctx, "op;fft", s, H);
    DSC_ASSERT(dsc_is_complex(H->dtype));
    const int bins = H->shape[DSC_MAX_DIMS - 1];
    DSC_ASSERT(H->ne == bins && bins >= 2);
    const int n = 2 * (bins - 1);
    DSC_ASSERT((n & (n - 1)) == 0);

    const int ls = s->shape[DSC_MAX_DIMS - 1];
    if (s->dtype == DSC_F32 && H->dtype == DSC_C32 && n == 65536) {   // rows shorter than n are zero padded, longer ones cropped
        int out_shape[DSC_MAX_DIMS];
        memcpy(out_shape, s->shape, sizeof(out_shape));
        out_shape[DSC_MAX_DIMS - 1] = n;
        out = make_out(ctx, s, out, out_shape, DSC_F32);
        const dsc_fft_plan *plan = dsc_plan_fft(ctx, 32768, DSC_FFT_REAL, DSC_C32);
        dsc_launch_filter64k((const float *) s->data, H->data, (float *) out->data, s->ne / ls, ls, ls < n ? ls : n, plan->tw_aux, ctx->n_cu,
                             ctx->stream);
        ctx->last_fft_path = "filter_64k_regs";
        return out;
    }
    const bool sp_f = s->dtype == DSC_F32 && H->dtype == DSC_C32, dp_f = s->dtype == DSC_F64 && H->dtype == DSC_C64;
    if ((sp_f || dp_f) && dsc_fft_regs_mid_supports(n / 2, DSC_MODE_R2C_PACKED, sp_f) && n / 2 <= 16384 && (long long) ls * 8 * 64 < (1LL << 30)) {
        int out_shape[DSC_MAX_DIMS];
        memcpy(out_shape, s->shape, sizeof(out_shape));
        out_shape[DSC_MAX_DIMS - 1] = n;
        out = make_out(ctx, s, out, out_shape, s->dtype);
        const dsc_fft_plan *plan = dsc_plan_fft(ctx, n / 2, DSC_FFT_REAL, s->dtype);
        dsc_launch_filter_regs_mid(s->data, H->data, out->data, s->ne / ls, n / 2, sp_f, plan->tw_full, plan->tw_real, ls, ls < n ? ls : n,
                                   ctx->stream);
        ctx->last_fft_path = "filter_mid_regs";
        return out;
    }
    dsc_tensor *S = dsc_rfft(ctx, s, nullptr, n, -1);
    dsc_tensor *P = dsc_mul(ctx, S, const_cast<dsc_tensor *>(H), nullptr);
    out = dsc_irfft(ctx, P, out, -1, -1);
    dsc_tensor_free(ctx, S);
    dsc_tensor_free(ctx, P);
    ctx->last_fft_path = "filter_composed";
    return out;
}
