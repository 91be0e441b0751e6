// fft_regs_cols.hip — register-resident transforms along a NON-LAST axis (strided lines), one HBM round trip.
//
// Reference: the same exec_fft / exec_rfft loops (dsc/src/dsc.cpp:1958-2007, 2102-2171) — dsc_axis_iterator walks any axis
// (dsc_iter.h:11-65), so `dsc.fft(x, axis=0)` is the same call as along the last axis (python/tests/test_ops.py:466-468).
//
// A tensor [outer][len][inner] transformed along `len`: element e of line (o, i) sits at ((o len + e) inner + i), i.e. the
// lines are strided but NEIGHBOURING LINES ARE CONTIGUOUS.  So the lanes of a wave are neighbouring lines ("columns"): a
// workgroup owns CW adjacent columns x T threads per column, thread (t, c) holds 32 elements of column c in registers —
// every global access of a wave is CW contiguous elements (128-512 B) per row, and every LDS exchange is indexed
// [...][c] with c fastest, which is bank-conflict free by construction.  The arithmetic is that of fft_regs_mid.hip:
//
//   three passes  L = 1024 B:  j = T j1 + B j2 + j3, T = 32 B threads per column:  dft32 (j1) . W_1024 . xchg . dft32 (j2) . W_L W_32B . xchg . dft_B (j3)
//   two passes    L = 32 B:    j = B j1 + j3,        T = B:                          dft32 (j1) . W_L . xchg . dft_B (j3)
//
// with the packed-real pre / post pass (dsc_fft.h:194-228) through an LDS staging plane [k][c].  This replaces
// transpose -> last-axis kernel -> transpose (three passes over HBM, 16-22 % of the roofline) and the strided LDS kernel for
// complex lengths 32 .. 2048 (c32 data: 4096).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <utility>

#include "fft_regs_common.h"

namespace {

__device__ __forceinline__ void store_real(float a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), r, voff, soff, kStream);
}
__device__ __forceinline__ void store_real(double a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, a), r, voff, soff, kStream);
}

// The two halves of a four-step transform along an axis of n = n1 n2 points (tensor [S][n][C], dsc_launch_fft_cols_4step below) are this
// kernel with two additions on the complex modes' way out:
//   tw4 / tw4_div   pass 1 (view [S][n2][n1 C], lines over j2): bin k2 of column q is multiplied by W_n^{j1 k2}, j1 = q / tw4_div
//   out_pitch/group pass 2 (view [S n2][n1][C], lines over j1): bin k1 of slice (s, k2) goes to row n2 k1 + k2 of slice s — consecutive
//                   bins are out_pitch elements apart and slice sigma starts at (sigma / group) (out_axis out_pitch) + (sigma % group) inner
// Plain calls pass {inner, 1, nullptr, 1}.
struct cols_remap {
    int out_pitch, group;
    const void *tw4;
    int tw4_div;
};

template<typename R, int B, bool TWO, int CW> struct cols_cfg {
    static constexpr int T = TWO ? B : 32 * B;         // threads per column
    static constexpr int L = 32 * T;                   // complex length
    static constexpr int NT = CW * T;
    static constexpr int COLS = TWO ? 32 : 1024;       // spectrum columns entering the last pass
    static constexpr int CPT = 32 / B;
    static constexpr int PLANE = (L + 1) * CW;         // values: every exchange needs L per column, the real staging L + 1
    static constexpr int TABLE = TWO ? L : 1024;       // W_L^m (two-pass) or W_1024^m
    static constexpr int TABLE_STRIDE = TWO ? 1 : B;
    static constexpr int WAVES_PER_EU = NT >= 1024 ? 4 : sizeof(R) == 8 ? 2 : NT >= 512 ? 4 : 2;
    static constexpr bool PIPE = false;                // persistent, software-pipelined tile loop (three-pass forms): measured, off
};
template<typename R, int B, bool TWO, int CW>
constexpr size_t cols_lds_bytes() { return ((size_t) cols_cfg<R, B, TWO, CW>::PLANE + 2 * cols_cfg<R, B, TWO, CW>::TABLE) * sizeof(R); }

// In: v[j1] = z[T j1 + t] of column c.  Out: v[i B + p] = bin (t + T i) + COLS brev(p).  LDS index = (flat index) * CW + c.
template<typename R, int B, bool TWO, int CW, bool INV>
__device__ __forceinline__ void cols_passes(cpx<R> (&v)[32], R *plane, const cpx<R> *wtab, const cpx<R> *__restrict__ tw_full, int t, int c) {
    using C = cpx<R>;
    using cfg = cols_cfg<R, B, TWO, CW>;
    constexpr int T = cfg::T, CPT = cfg::CPT;
    const int hi = TWO ? 0 : t / B, lo = TWO ? t : t % B;
    C u[32];
    if constexpr (!TWO) {
        dft_n<R, INV, 32>(v);                                       // pass 1 over j1, twiddle W_1024^{j2 k1}
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            const C w = wtab[hi * k1];
            v[brev(k1, 5)] = INV ? cmulc(v[brev(k1, 5)], w) : cmul(v[brev(k1, 5)], w);
        }
        // exchange 1: (j2, j3)[k1] -> thread B k1 + j3, [j2]:  flat = (B k1 + j3) 32 + j2
        R *wr = plane + (lo * 32 + hi) * CW + c;
        const R *rd = plane + (t * 32) * CW + c;
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) wr[k1 * B * 32 * CW] = v[brev(k1, 5)].x;
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m].x = rd[m * CW];
        lds_barrier();
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) wr[k1 * B * 32 * CW] = v[brev(k1, 5)].y;
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m].y = rd[m * CW];
        lds_barrier();
    } else {
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m] = v[m];
    }
    dft_n<R, INV, 32>(u);                                           // pass 2 over j2 (two-pass: over j1)
    if constexpr (TWO && B == 1) {
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = u[brev(k, 5)];
        return;
    }
    if constexpr (!TWO) {
        const C tw2_base = tw_full[hi * lo];                        // W_L^{j3 k1}
        u[0] = INV ? cmulc(u[0], tw2_base) : cmul(u[0], tw2_base);
#pragma unroll
        for (int k2 = 1; k2 < 32; ++k2) {
            const C w = cmul(tw2_base, wtab[(32 / B) * lo * k2]);    // x W_32B^{j3 k2}
            u[brev(k2, 5)] = INV ? cmulc(u[brev(k2, 5)], w) : cmul(u[brev(k2, 5)], w);
        }
    } else {
#pragma unroll
        for (int k2 = 1; k2 < 32; ++k2) {
            const C w = wtab[lo * k2];                               // W_L^{j3 k2}
            u[brev(k2, 5)] = INV ? cmulc(u[brev(k2, 5)], w) : cmul(u[brev(k2, 5)], w);
        }
    }
    // last exchange: flat = (column k') B + j3, k' = k1 + 32 k2 (two-pass: k2); thread t reads columns t + T i
    constexpr int CS = TWO ? 1 : 32;
    R *wr = plane + (hi * B + lo) * CW + c;
    const R *rd = plane + (t * B) * CW + c;
#pragma unroll
    for (int k2 = 0; k2 < 32; ++k2) wr[k2 * CS * B * CW] = u[brev(k2, 5)].x;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int m = 0; m < B; ++m) v[i * B + m].x = rd[(i * T * B + m) * CW];
    lds_barrier();
#pragma unroll
    for (int k2 = 0; k2 < 32; ++k2) wr[k2 * CS * B * CW] = u[brev(k2, 5)].y;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int m = 0; m < B; ++m) v[i * B + m].y = rd[(i * T * B + m) * CW];
    lds_barrier();
    dft_columns<R, INV, B>(v, std::make_integer_sequence<int, CPT>{});
}

// MODE: DSC_MODE_C2C (INV either way), DSC_MODE_R2C_PACKED (forward), DSC_MODE_C2R_PACKED (inverse).
// in / out: tensor base; the transformed axis has in_len valid input elements (reals for R2C_PACKED, bins for C2R, complex
// otherwise; the rest of the transform length reads as zero: dsc.cpp:1990-1998, 2125-2133, 2149-2157) and a pitch of `inner`
// elements between consecutive elements; in_axis / out_axis = the axis lengths of the two tensors (slice pitch = axis * inner).
//
// Round 3.  (1) The three-pass forms (L = 2048, 4096: one workgroup fills a CU) are PERSISTENT and software pipelined like the
// 65536-point kernels: a workgroup walks tiles blockIdx.x, + gridDim.x, ...; the next tile's loads are issued into registers
// the current tile has finished with, ahead of (half of) its stores, so that a CU's load, compute and store phases overlap.
// (2) The inverse packed-real pre-pass handles every pair (k, L - k) ONCE: a thread loads its lower 16 bins AND their partners
// (rows L - k: the same 32 row loads as its own upper half), computes Z[k] and Z[L - k] together and hands Z[L - k] to its owner
// through the staging plane — half the arithmetic, one exchange instead of two, and no second 32-value array (the old form
// spilled 14 - 35 registers).
template<typename R, int B, bool TWO, int CW, int MODE, bool INV>
__global__ __launch_bounds__((cols_cfg<R, B, TWO, CW>::NT), (cols_cfg<R, B, TWO, CW>::WAVES_PER_EU)) void fft_cols_kernel(
    const void *__restrict__ in, void *__restrict__ out, int inner, int tiles_per_slice, int n_tiles, int in_axis, int in_len, int out_axis,
    const cpx<R> *__restrict__ tw_full, const cpx<R> *__restrict__ tw_real, R scale, cols_remap rm) {
    using C = cpx<R>;
    using cfg = cols_cfg<R, B, TWO, CW>;
    constexpr int T = cfg::T, L = cfg::L, NT = cfg::NT, CPT = cfg::CPT, COLS = cfg::COLS, LOGB = ilog2(B);
    constexpr int CB = (int) sizeof(C), RB = (int) sizeof(R);
    constexpr int IB = (MODE == DSC_MODE_R2C_PACKED || MODE == DSC_MODE_R2C_CAST) ? RB : CB;       // bytes per input element
    constexpr int OB = MODE == DSC_MODE_C2R_PACKED ? RB : CB;       // bytes per output element
    constexpr int kOut = 0x7f000000;                                // an offset past every descriptor range: reads 0, stores dropped
    // PIPE: persistent + next tile requested ahead of the stores.  Built and measured in round 3 (profiles/r03_cols_axis0.md): it does
    // not pay — 2048-point c32 lines 0.648 ms against 0.584 ms, 4096-point 1.27 against 1.13 — because these tiles are not limited by
    // the overlap of their phases but by the size of their pieces (16 / 8 columns = 128 / 64 B per row of the tensor); kept, off.
    constexpr bool PIPE = cfg::PIPE && !TWO;
    // the inverse pre-pass with every pair handled once (see above): the f32 forms lose their spills with it (14 - 25 -> 0 - 4),
    // the f64 forms get more (6 - 35 -> 19 - 54: hipcc starts all sixteen pairs at once), so they keep the two-exchange form
    // (later in round 3: the one-exchange pre-pass below needs no second set of loads and measured equal or better — irfft along axis 0,
    // 1024 points 63.1 -> 65.3 %, 512: 67.0 -> 67.8 %, the others within 0.3 points (tools/r03_call_aa.sh) — so the form that loads the
    // partners from memory is off; -DDSC_COLS_PAIR_ONCE brings it back for f32)
#ifdef DSC_COLS_PAIR_ONCE
    constexpr bool PAIR_ONCE = sizeof(R) == 4;
#else
    constexpr bool PAIR_ONCE = false;
#endif
    // The real passes that move only the upper halves through the staging plane (fft_regs_mid.hip, "who owns which bin": thread t of a
    // column holds the bins t + T m, its partners L - k are upper bins of thread T - t of the same column): forward post-pass and, where
    // PAIR_ONCE is off, the inverse pre-pass.
#ifdef DSC_COLS_OLD_REAL_PASSES
    constexpr bool POST_ONCE = false, PRE_ONCE = false;
#else
    constexpr bool POST_ONCE = T >= 2 && !PIPE, PRE_ONCE = T >= 2;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *wtab = (C *) (plane + cfg::PLANE);

    for (int i = threadIdx.x; i < cfg::TABLE; i += NT) wtab[i] = tw_full[(long long) i * cfg::TABLE_STRIDE];
    // Everything derived from the thread id is loop invariant: hipcc hoists it out of the persistent loop (dozens of addresses) and
    // then spills.  Each phase therefore rebuilds (t, c) from an opaque copy of the id, so that nothing outlives its phase.
    auto tid_now = [&]() { int x = (int) threadIdx.x; if constexpr (PIPE) asm volatile("" : "+v"(x)); return x; };

    // descriptor of a tile's input slice (wave uniform) and a lane's column in it; a tile past the end reads zeros
    auto in_rsrc = [&](int tile) {
        const int slice = tile / tiles_per_slice;
        return __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) in + (size_t) slice * in_axis * inner * IB), 0, tile < n_tiles ? in_len * inner * IB : 0, 0x00020000);
    };
    auto col_of = [&](int tile, int c) { return (tile - (tile / tiles_per_slice) * tiles_per_slice) * CW + c; };
    // the loads of one tile, rows [J0, J0 + N) of the 32 a thread holds (see the modes below); C2R: dst[j] = Y[T j + t] for j < 16,
    // dst[16 + j] = its partner Y[L - T j - t]
    auto request = [&](auto &dst, auto j0_tag, auto n_tag, int tile) {
        constexpr int J0 = decltype(j0_tag)::value, N = decltype(n_tag)::value;
        const __amdgpu_buffer_rsrc_t rin = in_rsrc(tile);
        const int tid = tid_now(), c = tid % CW, t = tid / CW;
        const int col = col_of(tile, c);
        const bool live = col < inner;
        if constexpr (MODE == DSC_MODE_R2C_PACKED) {                // z[m] = (x[2m], x[2m + 1]): two rows of the axis
            const int voff = live ? (2 * t * inner + col) * RB : kOut;
            const int row_b = inner * RB, step = 2 * T * inner * RB;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const cpx<R> a = buf_load_real<kStream>(rin, voff, (J0 + j) * step, R{}), b = buf_load_real<kStream>(rin, voff, (J0 + j) * step + row_b, R{});
                dst[j] = C{a.x, b.x};
            }
        } else if constexpr (MODE == DSC_MODE_R2C_CAST) {           // dsc_fft / dsc_ifft of a real tensor: widened while loading
            const int voff = live ? (t * inner + col) * RB : kOut;
            const int step = T * inner * RB;
#pragma unroll
            for (int j = 0; j < N; ++j) dst[j] = buf_load_real<kStream>(rin, voff, (J0 + j) * step, R{});
        } else if constexpr (MODE == DSC_MODE_C2R_PACKED && PAIR_ONCE) {
            const int step = T * inner * CB;
            const int voff_k = live ? (t * inner + col) * CB : kOut;
            const int voff_m = live ? (((L - 15 * T) - t) * inner + col) * CB : kOut;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int jj = J0 + j;
                dst[j] = jj < 16 ? buf_load<kStream>(rin, voff_k, jj * step, R{}) : buf_load<kStream>(rin, voff_m, (31 - jj) * step, R{});
            }
        } else {
            const int voff = live ? (t * inner + col) * CB : kOut;
            const int step = T * inner * CB;
#pragma unroll
            for (int j = 0; j < N; ++j) dst[j] = buf_load<kStream>(rin, voff, (J0 + j) * step, R{});
        }
    };
    auto request_mid = [&](int tile) {                              // C2R, thread 0 of a column: bin L/2, which pairs with itself (two-exchange form: bin L)
        C y = C{(R) 0, (R) 0};
        const int tid = tid_now(), c = tid % CW, t = tid / CW;
        if (MODE == DSC_MODE_C2R_PACKED && t == 0) {
            const int col = col_of(tile, c);
            y = buf_load<kStream>(in_rsrc(tile), col < inner ? col * CB : kOut, (PAIR_ONCE ? L / 2 : L) * inner * CB, R{});
        }
        return y;
    };
    using i0 = std::integral_constant<int, 0>;
    using i16 = std::integral_constant<int, 16>;
    using i32 = std::integral_constant<int, 32>;

    // C2R: the pre-pass twiddle of this thread, requested FIRST — arriving after the data it lets hipcc start all sixteen pairs
    // (their sums and differences) and hold them until it is there: 40 - 90 spilled registers
    C wpre = C{(R) 0, (R) 0};
    if constexpr (MODE == DSC_MODE_C2R_PACKED && sizeof(R) == 4) {
        wpre = tw_real[tid_now() / CW];
        asm volatile("" : "+v"(wpre.x), "+v"(wpre.y));
    }
    C v[32];
    request(v, i0{}, i32{}, (int) blockIdx.x);
    C ymid = C{(R) 0, (R) 0};
    if constexpr (PAIR_ONCE) ymid = request_mid((int) blockIdx.x);
    if constexpr (PIPE) __syncthreads();                            // twiddle table visible (one-tile forms: at the barrier after the pre-pass,
                                                                    // so that the loads are not all waited for at once)
    int tile = blockIdx.x;
    do {                                                            // one tile unless PIPE
        const int next = PIPE ? tile + (int) gridDim.x : n_tiles;
        const int slice = tile / tiles_per_slice;
        // (plain calls: group = 1, out_pitch = inner — slice * out_axis * inner elements, a range of out_axis * inner)
        const size_t out_base = (size_t) (slice / rm.group) * out_axis * rm.out_pitch + (size_t) (slice % rm.group) * inner;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            (void *) ((char *) out + out_base * OB), 0, ((out_axis - 1) * rm.out_pitch + inner) * OB, 0x00020000);

        if constexpr (MODE == DSC_MODE_C2R_PACKED && !PAIR_ONCE && T == 1) {
            // 32-point lines, one thread per column: both partners of every pair live in this thread — no staging, no barrier.
            // (k, L - k) = (j, 32 - j): a = v[j], b = v[32 - j]; W_2L^j = W_64^j, a compile-time constant.
            const C yl = request_mid(tile);                         // bin L
            { const R a = v[0].x, b = yl.x; v[0] = C{(R) 0.5 * (a + b), (R) 0.5 * (a - b)}; }       // bins 0 and L: real parts only (dsc_fft.h:227-232)
#pragma unroll
            for (int j = 1; j < 16; ++j) {
                const R wqx = (R) (0.5 * root64_im(j)), wqy = (R) (0.5 * root64_re(j));             // wq = (i/2) conj(W_64^j)
                const C a = v[j], b = v[32 - j];
                const R sx = (R) 0.5 * (a.x + b.x), sy = (R) 0.5 * (a.y - b.y), dx = a.x - b.x, dy = a.y + b.y;
                const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
                v[j] = C{sx + wdx, sy + wdy};
                v[32 - j] = C{sx - wdx, wdy - sy};
            }
            v[16].y = -v[16].y;                                     // Z[L/2] = conj Y[L/2]
        }
        if constexpr (MODE == DSC_MODE_C2R_PACKED && !PAIR_ONCE && T != 1 && PRE_ONCE) {
            // pair (k, L - k), k = t + T j, j < 16: a = Y[k] (own), b = Y[L-k] = an upper bin of thread T - t; the upper halves go to the
            // staging plane (slot of bin b: b - L/2; x plane rows [0, L/2), y plane [L/2, L)), one barrier, every pair is computed once,
            // Z[L-k] goes back into the slot b came from (this thread is its only reader), second barrier, everybody collects its upper half
            const int tid = tid_now(), c = tid % CW, t = tid / CW;
            R *stage = plane + c;
            const C wbase = tw_real[t];
            C yl = request_mid(tile);                               // bin L (thread 0)
            if (t == 0) { v[0].y = (R) 0; yl.y = (R) 0; }           // bins 0 and L: real parts only (dsc_fft.h:227-228)
            R *own_x = stage + t * CW, *own_y = own_x + (L / 2) * CW;
            R *par_x = stage + ((L / 2 - 15 * T) - t) * CW, *par_y = par_x + (L / 2) * CW;
#pragma unroll
            for (int j = 16; j < 32; ++j) { own_x[T * (j - 16) * CW] = v[j].x; own_y[T * (j - 16) * CW] = v[j].y; }
            if (t == 0) own_y[0] = -v[16].y;                        // bin L/2 pairs with itself: Z[L/2] = conj Y[L/2]
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const C a = v[j];
                C b = C{par_x[T * (15 - j) * CW], par_y[T * (15 - j) * CW]};
                if (j == 0 && t == 0) b = yl;
                const C w = cmul(wbase, C{(R) root64_re(j), (R) root64_im(j)});        // W_2L^{t + T j} = W_2L^t W_64^j
                const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;
                const R sx = (R) 0.5 * (a.x + b.x), sy = (R) 0.5 * (a.y - b.y), dx = a.x - b.x, dy = a.y + b.y;
                const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
                v[j] = C{sx + wdx, sy + wdy};
                if (!(j == 0 && t == 0)) {                          // Z[L] is no bin
                    par_x[T * (15 - j) * CW] = sx - wdx;
                    par_y[T * (15 - j) * CW] = wdy - sy;
                }
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) v[16 + j] = C{own_x[T * j * CW], own_y[T * j * CW]};
        }
        if constexpr (MODE == DSC_MODE_C2R_PACKED && !PAIR_ONCE && T != 1 && !PRE_ONCE) {
            // Z[k] = (a + conj b)/2 + wq (a - conj b), a = Y[k], b = Y[L-k], wq = (i/2) conj(W_2L^k), k = T j1 + t (dsc_fft.h:194-228):
            // every thread for its own 32 bins, b through the staging plane one component at a time
            const int tid = tid_now(), c = tid % CW, t = tid / CW;
            R *stage = plane + c;
            const C wbase = tw_real[t];
            C yl = request_mid(tile);                               // bin L (thread 0), requested here as the round-2 form did
            if (t == 0) { v[0].y = (R) 0; yl.y = (R) 0; }           // bins 0 and L: real parts only (dsc_fft.h:227-228)
            R dx[32];
            R *up = stage + t * CW;
            const R *dn = stage + ((L - 31 * T) - t) * CW;
#pragma unroll
            for (int j1 = 0; j1 < 32; ++j1) up[T * j1 * CW] = v[j1].x;
            if (t == 0) stage[L * CW] = yl.x;
            lds_barrier();
#pragma unroll
            for (int j1 = 0; j1 < 32; ++j1) {
                const R bx = dn[T * (31 - j1) * CW];
                dx[j1] = v[j1].x - bx;
                v[j1].x = v[j1].x + bx;
            }
            lds_barrier();
#pragma unroll
            for (int j1 = 0; j1 < 32; ++j1) up[T * j1 * CW] = v[j1].y;
            if (t == 0) stage[L * CW] = yl.y;
            lds_barrier();
#pragma unroll
            for (int j1 = 0; j1 < 32; ++j1) {
                const R by = dn[T * (31 - j1) * CW];
                const C w = cmul(wbase, C{(R) root64_re(j1), (R) root64_im(j1)});      // W_2L^{t + T j1} = W_2L^t W_64^{j1}
                const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;
                const R sy = v[j1].y - by, dy = v[j1].y + by;
                const R zx = (R) 0.5 * v[j1].x + (dx[j1] * wqx - dy * wqy);
                const R zy = (R) 0.5 * sy + (dx[j1] * wqy + dy * wqx);
                v[j1] = C{zx, zy};
            }
        }
        if constexpr (MODE == DSC_MODE_C2R_PACKED && PAIR_ONCE) {
            const int tid = tid_now(), c = tid % CW, t = tid / CW;
            R *stage = plane + c;                                   // stage[k * CW] = component of bin k of this column
            // pair (k, L - k), k = T j + t, j < 16: a = Y[k] = v[j], b = Y[L - k] = v[16 + j];  s = a + conj b, d = a - conj b,
            // wq = (i/2) conj(W_2L^k):  Z[k] = s/2 + wq d (kept),  Z[L - k] = conj(s/2 - wq d) (handed to its owner)   (dsc_fft.h:194-228)
            const C wbase = wpre;
            if (t == 0) { v[0].y = (R) 0; v[16].y = (R) 0; }        // bins 0 and L: real parts only (dsc_fft.h:227-228)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const C w = cmul(wbase, C{(R) root64_re(j), (R) root64_im(j)});        // W_2L^{t + T j} = W_2L^t W_64^j
                const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;
                const C a = v[j], b = v[16 + j];
                const R sx = (R) 0.5 * (a.x + b.x), sy = (R) 0.5 * (a.y - b.y), dx = a.x - b.x, dy = a.y + b.y;
                const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
                v[j] = C{sx + wdx, sy + wdy};
                v[16 + j] = C{sx - wdx, wdy - sy};
            }
            // Z[L - k] goes to staging index L - k; everybody then reads its own upper half k' = T j' + t, j' = 16 .. 31.  Thread 0
            // pairs with itself (its Z[L - T j] are its own rows 32 - j) and supplies Z[L/2] = conj Y[L/2]; its write at index L is never read.
            R *up = stage + ((L - 15 * T) - t) * CW;                // index L - T j - t = (L - 15 T - t) + T (15 - j)
            const R *dn = stage + (16 * T + t) * CW;
#pragma unroll
            for (int j = 0; j < 16; ++j) up[T * (15 - j) * CW] = v[16 + j].x;
            if (t == 0) stage[(L / 2) * CW] = ymid.x;
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) v[16 + j].x = dn[T * j * CW];
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) up[T * (15 - j) * CW] = v[16 + j].y;
            if (t == 0) stage[(L / 2) * CW] = -ymid.y;
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 16; ++j) v[16 + j].y = dn[T * j * CW];
        }
        if constexpr (PIPE) lds_barrier();      // staging reads (this tile's pre-pass, the previous tile's post-pass) done before the plane is reused
        else __syncthreads();                   // ... and the twiddle table visible

        {
            const int tid = tid_now();
            cols_passes<R, B, TWO, CW, INV>(v, plane, wtab, tw_full, tid / CW, tid % CW);
        }
        const int tid = tid_now(), c = tid % CW, t = tid / CW;
        const int col = col_of(tile, c);
        const bool live = col < inner;
        R *stage = plane + c;                                       // stage[k * CW] = component of bin k of this column

        if constexpr (MODE == DSC_MODE_C2C || MODE == DSC_MODE_R2C_CAST) {
            const int voff = live ? (t * rm.out_pitch + col) * CB : kOut;
            const int step = rm.out_pitch * CB;
            if (rm.tw4 != nullptr) {                                // four-step, pass 1: times W_n^{j1 k}, k the bin, j1 = col / tw4_div
                const C *tw4 = (const C *) rm.tw4;
                const int j1 = col / rm.tw4_div;
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    const int k = t + T * (q / B) + COLS * brev(q % B, LOGB);
                    const C w = tw4[live ? j1 * k : 0];
                    v[q] = INV ? cmulc(v[q], w) : cmul(v[q], w);
                }
            }
            auto put = [&](int q) {                                 // q = i B + p
                const C r = v[q];
                buf_store<kStream>(C{r.x * scale, r.y * scale}, rout, voff, (T * (q / B) + COLS * brev(q % B, LOGB)) * step);
            };
            if constexpr (PIPE) {
                // (the scheduling fences keep hipcc from hoisting all 32 loads in front of the stores: v[] would still be live)
                __builtin_amdgcn_sched_barrier(0);
                C nx[16];
                request(nx, i0{}, i16{}, next);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) put(q);
                __builtin_amdgcn_sched_barrier(0);
                C ny[16];
                request(ny, i16{}, i16{}, next);                    // into the registers the first sixteen stores have read
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 16; q < 32; ++q) put(q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = nx[j]; v[16 + j] = ny[j]; }
            } else {
#pragma unroll
                for (int q = 0; q < 32; ++q) put(q);
            }
        } else if constexpr (MODE == DSC_MODE_C2R_PACKED) {         // sample pair (2k, 2k + 1) = (re, im) of z[k]: two rows of the output axis
            const int voff = live ? (2 * t * inner + col) * RB : kOut;
            const int row_b = inner * RB;
            auto put = [&](int q) {
                const C r = v[q];
                const int soff = 2 * (T * (q / B) + COLS * brev(q % B, LOGB)) * row_b;
                store_real(r.x * scale, rout, voff, soff);
                store_real(r.y * scale, rout, voff, soff + row_b);
            };
            if constexpr (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
                C nx[16];
                request(nx, i0{}, i16{}, next);
                ymid = request_mid(next);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) put(q);
                __builtin_amdgcn_sched_barrier(0);
                C ny[16];
                request(ny, i16{}, i16{}, next);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 16; q < 32; ++q) put(q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = nx[j]; v[16 + j] = ny[j]; }
            } else {
#pragma unroll
                for (int q = 0; q < 32; ++q) put(q);
            }
        } else if constexpr (POST_ONCE) {
            // packed-real post-pass (dsc_fft.h:199-225): only the upper halves travel (see above), one barrier
            const C wbase = tw_real[t];
            auto reg_of = [](int m) constexpr { return (m % CPT) * B + brev(m / CPT, LOGB); };      // register that holds bin t + T m
            R *own_x = stage + t * CW, *own_y = own_x + (L / 2) * CW;
            const R *par_x = stage + ((L / 2 - 15 * T) - t) * CW, *par_y = par_x + (L / 2) * CW;
#pragma unroll
            for (int m = 16; m < 32; ++m) { own_x[T * (m - 16) * CW] = v[reg_of(m)].x; own_y[T * (m - 16) * CW] = v[reg_of(m)].y; }
            const C zmid = v[reg_of(16)];                           // thread 0: bin L/2, which pairs with itself
            lds_barrier();
            const int step = inner * CB;
            const int voff_k = live ? (t * inner + col) * CB : kOut;
            const int voff_m = live ? (((L - 15 * T) - t) * inner + col) * CB : kOut;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const C a = v[reg_of(m)];
                C b = C{par_x[T * (15 - m) * CW], par_y[T * (15 - m) * CW]};
                if (m == 0 && t == 0) b = a;                        // Z[L] := Z[0]
                const C w = cmul(wbase, C{(R) root64_re(m), (R) root64_im(m)});        // W_2L^{t + T m} = W_2L^t W_64^m
                const R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;
                const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
                const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
                C xk = C{(R) 0.5 * sx + wdx, (R) 0.5 * sy + wdy};
                C xm = C{(R) 0.5 * sx - wdx, wdy - (R) 0.5 * sy};
                if (m == 0 && t == 0) { xk.y = (R) 0; xm.y = (R) 0; }           // dsc_fft.h:221-225 stores exact zeros
                buf_store<kStream>(C{xk.x * scale, xk.y * scale}, rout, voff_k, T * m * step);
                buf_store<kStream>(C{xm.x * scale, xm.y * scale}, rout, voff_m, T * (15 - m) * step);
            }
            if (t == 0) buf_store<kStream>(C{zmid.x * scale, -zmid.y * scale}, rout, voff_k, (L / 2) * step);     // k = L/2: a = b, W_2L^{L/2} = -i
        } else {
            // packed-real post-pass (dsc_fft.h:199-225), one thread per PAIR (k, L-k), k = t + T i < L/2, plus k = L/2 (thread 0)
            const C wbase = tw_real[t];
            R ax[16], bx[16], amx = (R) 0;
            R *up = stage + t * CW;
            const R *dn = stage + ((L - 15 * T) - t) * CW;
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int p = 0; p < B; ++p) up[(T * i + COLS * brev(p, LOGB)) * CW] = v[i * B + p].x;
            if (t == 0) stage[L * CW] = v[0].x;                                   // Z[L] := Z[0]
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 16; ++i) { ax[i] = up[T * i * CW]; bx[i] = dn[T * (15 - i) * CW]; }
            if (t == 0) amx = stage[(L / 2) * CW];
            lds_barrier();
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int p = 0; p < B; ++p) up[(T * i + COLS * brev(p, LOGB)) * CW] = v[i * B + p].y;
            if (t == 0) stage[L * CW] = v[0].y;
            if constexpr (PIPE) {                                   // v is in the staging plane: its registers take the next tile
                __builtin_amdgcn_sched_barrier(0);
                request(v, i0{}, i32{}, next);
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier();
            const int step = inner * CB;
            const int voff_k = live ? (t * inner + col) * CB : kOut;
            const int voff_m = live ? (((L - 15 * T) - t) * inner + col) * CB : kOut;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const R ay = up[T * i * CW], by = dn[T * (15 - i) * CW];
                const C w = cmul(wbase, C{(R) root64_re(i), (R) root64_im(i)});        // W_2L^{t + T i} = W_2L^t W_64^i
                const R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;
                const R sx = ax[i] + bx[i], sy = ay - by, dx = ax[i] - bx[i], dy = ay + by;
                const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
                C xk = C{(R) 0.5 * sx + wdx, (R) 0.5 * sy + wdy};
                C xm = C{(R) 0.5 * sx - wdx, wdy - (R) 0.5 * sy};
                if (i == 0 && t == 0) { xk.y = (R) 0; xm.y = (R) 0; }           // dsc_fft.h:221-225 stores exact zeros
                buf_store<kStream>(C{xk.x * scale, xk.y * scale}, rout, voff_k, T * i * step);
                buf_store<kStream>(C{xm.x * scale, xm.y * scale}, rout, voff_m, T * (15 - i) * step);
            }
            if (t == 0) {                                                     // k = L/2: a = b, W_2L^{L/2} = -i
                const R ay = stage[(L / 2) * CW];
                buf_store<kStream>(C{amx * scale, -ay * scale}, rout, voff_k, (L / 2) * step);
            }
        }
    } while (PIPE && (tile += (int) gridDim.x) < n_tiles);
}

// ------------------------------------------------------------------------------------------------
// REAL transforms along an axis of n = n1 n2 >= 4096 points (dsc_rfft / dsc_irfft along a non-last axis, full lines, an even number of
// columns): TWO neighbouring real columns are one complex column — the real tensor [S][n][C] read as complex [S][n][C/2] is
// z = x_a + i x_b — and the complex four-step above does the work; the two spectra are separated (forward) or merged (inverse) where
// the data passes through registers anyway.  Two passes over HBM (16 B per real sample against the algorithmic 8) instead of the
// transpose route's three.
//
// forward   pass 1: the plain complex pass over j2 with the W_n^{j1 k2} twiddle (fft_cols_kernel on the complex view)
//           pass 2: cols_real_split_kernel — lines over j1 of slice k2 give Z[n2 k1 + k2]; X_a[k] = (Z[k] + conj Z[n-k]) / 2,
//                   X_b[k] = -i (Z[k] - conj Z[n-k]) / 2, and n - k = n2 (n1 - 1 - k1) + (n2 - k2) lies in slice n2 - k2.  A workgroup
//                   therefore takes the slice PAIR (k2, n2 - k2), half of its columns each; the partner of the lower bin k1 of one
//                   half is the upper bin n1 - 1 - k1 of the same column in the other half: the upper halves go through the staging
//                   plane once (the ownership of section 4.1b across the two halves), every thread separates its sixteen lower bins
//                   and stores (X_a[k], X_b[k]) — neighbouring columns of row k — in one access.  Slice 0 pairs with itself
//                   (partner n1 - k1, Z[n] := Z[0]) and so does slice n2 / 2; both halves then hold the same slice and only one stores.
// inverse   pass 1: cols_real_merge_kernel — Z[k] = Y_a[k] + i Y_b[k] for k <= n/2 and conj(Y_a[n-k]) + i conj(Y_b[n-k]) above; the rows
//                   above the middle of column block j1 are the lower rows of block n1 - j1, so a workgroup takes that block PAIR and
//                   the halves hand each other the conjugate combinations (every spectrum row is read once); then the inverse pass
//                   over j2 and the conjugate twiddle, into the work tensor
//           pass 2: the plain complex pass over j1 with the output remap; its output rows are the real rows (x_a, x_b interleaved)
template<typename R, int B, bool TWO, int CW>
__global__ __launch_bounds__((cols_cfg<R, B, TWO, CW>::NT), (cols_cfg<R, B, TWO, CW>::WAVES_PER_EU)) void cols_real_split_kernel(
    const cpx<R> *__restrict__ work, cpx<R> *__restrict__ out, int cc_n, int tiles_per_pair, int n2, const cpx<R> *__restrict__ tw_full) {
    using C = cpx<R>;
    using cfg = cols_cfg<R, B, TWO, CW>;
    constexpr int T = cfg::T, L = cfg::L, NT = cfg::NT, CPT = cfg::CPT, LOGB = ilog2(B), H = CW / 2;
    constexpr int CB = (int) sizeof(C);
    constexpr int kOut = 0x7f000000;
    static_assert(T >= 2, "a line of 32 points has its pairs inside one thread: use 64 points and more");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *wtab = (C *) (plane + cfg::PLANE);
    for (int i = threadIdx.x; i < cfg::TABLE; i += NT) wtab[i] = tw_full[(long long) i * cfg::TABLE_STRIDE];

    const int tile = blockIdx.x;
    const int ct = tile % tiles_per_pair, pr = tile / tiles_per_pair;
    const int n_pairs = n2 / 2 + 1;
    const int p = pr % n_pairs, s = pr / n_pairs;
    const int k_a = p, k_b = (n2 - p) % n2;
    const bool self = k_a == k_b;                                   // slices 0 and n2 / 2 pair with themselves
    const int tid = threadIdx.x, c = tid % CW, t = tid / CW, h = c / H;
    const int col = ct * H + (c - h * H);
    const bool live = col < cc_n;
    const int k2 = h ? k_b : k_a;
    const long long n = (long long) L * n2;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) (work + (size_t) s * n * cc_n), 0, (int) (n * cc_n * CB), 0x00020000);
    C v[32];
    {
        const int voff = live ? ((k2 * L + t) * cc_n + col) * CB : kOut;
        const int step = T * cc_n * CB;
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = buf_load<kStream>(rin, voff, j * step, R{});
    }
    __syncthreads();
    cols_passes<R, B, TWO, CW, false>(v, plane, wtab, tw_full, t, c);      // v[reg_of(m)] = Z[n2 (t + T m) + k2] of this column

    auto reg_of = [](int m) constexpr { return (m % CPT) * B + brev(m / CPT, LOGB); };
    R *own_x = plane + c + t * CW, *own_y = own_x + (L / 2) * CW;   // [T (m - 16) CW]: slot of the own upper bin k1 = t + T m
#pragma unroll
    for (int m = 16; m < 32; ++m) { own_x[T * (m - 16) * CW] = v[reg_of(m)].x; own_y[T * (m - 16) * CW] = v[reg_of(m)].y; }
    const C zmid = v[reg_of(16)];                                   // slice 0, thread 0: Z[n/2], which pairs with itself
    lds_barrier();
    const int shift = p == 0 ? 0 : 1;                               // partner of k1: n1 - k1 in slice 0, n1 - 1 - k1 otherwise
    const int pc = c < H ? c + H : c - H;                           // the same column in the other half
    const R *par_x = plane + pc + ((L / 2 - 15 * T) - shift - t) * CW;      // [T (15 - m) CW]: slot of k1' - n1/2 = n1/2 - shift - k1
    const R *par_y = par_x + (L / 2) * CW;
    const long long out_rows = n / 2 + 1;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (out + (size_t) s * out_rows * 2 * cc_n), 0,
                                                                          (int) (out_rows * 2 * cc_n * CB), 0x00020000);
    const bool stores = live && !(self && h == 1);
    const int ovoff = stores ? ((k2 + n2 * t) * 2 * cc_n + 2 * col) * CB : kOut;
    const int ostep = T * n2 * 2 * cc_n * CB;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const C a = v[reg_of(m)];
        C b = C{par_x[T * (15 - m) * CW], par_y[T * (15 - m) * CW]};
        if (m == 0 && t == 0 && p == 0) b = a;                      // Z[n] := Z[0]
        const C xa = C{(R) 0.5 * (a.x + b.x), (R) 0.5 * (a.y - b.y)};
        const C xb = C{(R) 0.5 * (a.y + b.y), (R) -0.5 * (a.x - b.x)};
        buf_store_pair<kStream>(xa, xb, rout, ovoff, m * ostep);
    }
    if (p == 0 && t == 0)                                           // row n/2 = n2 (n1/2): X_a = Re Z, X_b = Im Z
        buf_store_pair<kStream>(C{zmid.x, (R) 0}, C{zmid.y, (R) 0}, rout, ovoff, 16 * ostep);
}

template<typename R, int B, bool TWO, int CW>
__global__ __launch_bounds__((cols_cfg<R, B, TWO, CW>::NT), (cols_cfg<R, B, TWO, CW>::WAVES_PER_EU)) void cols_real_merge_kernel(
    const cpx<R> *__restrict__ in, cpx<R> *__restrict__ work, int cc_n, int n1, int tiles_per_pair, const cpx<R> *__restrict__ tw_full,
    const cpx<R> *__restrict__ twn) {
    // The mirror of the split kernel.  Column block j1 of the view [n2][n1 cc_n] needs the rows k = j1 + n1 j2 of the full spectrum;
    // those above the middle are the conjugates of rows n - k = (n1 - j1) + n1 (n2 - 1 - j2): the LOWER rows of block n1 - j1.  A
    // workgroup takes the block pair (j1, n1 - j1), half of its columns each; every thread loads its sixteen lower rows ONCE, keeps
    // Z = Y_a + i Y_b and hands conj Y_a + i conj Y_b to the other half through the staging plane (one barrier) — each row of the
    // spectrum is read exactly once.  Block 0 pairs with itself (partner n2 - j2; row n/2 is loaded by its thread), block n1/2 too.
    using C = cpx<R>;
    using cfg = cols_cfg<R, B, TWO, CW>;
    constexpr int T = cfg::T, L = cfg::L, NT = cfg::NT, COLS = cfg::COLS, LOGB = ilog2(B), H = CW / 2;      // L = n2
    constexpr int CB = (int) sizeof(C);
    constexpr int kOut = 0x7f000000;
    static_assert(T >= 2, "lines of 64 points and more");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *wtab = (C *) (plane + cfg::PLANE);
    for (int i = threadIdx.x; i < cfg::TABLE; i += NT) wtab[i] = tw_full[(long long) i * cfg::TABLE_STRIDE];

    const int tile = blockIdx.x;
    const int ct = tile % tiles_per_pair, pr = tile / tiles_per_pair;
    const int n_pairs = n1 / 2 + 1;
    const int p = pr % n_pairs, s = pr / n_pairs;
    const int j_a = p, j_b = (n1 - p) % n1;
    const bool self = j_a == j_b;
    const int tid = threadIdx.x, c = tid % CW, t = tid / CW, h = c / H;
    const int col = ct * H + (c - h * H);
    const bool live = col < cc_n;
    const int j1 = h ? j_b : j_a;
    const long long n = (long long) L * n1;
    const int half = (int) (n / 2);
    const long long in_rows = n / 2 + 1;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) (in + (size_t) s * in_rows * 2 * cc_n), 0,
                                                                         (int) (in_rows * 2 * cc_n * CB), 0x00020000);
    C v[32];
    R *own_x = plane + c + t * CW, *own_y = own_x + (L / 2) * CW;   // [T j CW]: slot j2 = t + T j of this column
    {
        const int voff = live ? ((j1 + n1 * t) * 2 * cc_n + 2 * col) * CB : kOut;   // row k = j1 + n1 (t + T j), columns (2 col, 2 col + 1)
        const int step = T * n1 * 2 * cc_n * CB;
        C ya[16], yb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) buf_load_pair<kStream>(ya[j], yb[j], rin, voff, j * step);
        C ym_a = C{(R) 0, (R) 0}, ym_b = ym_a;
        if (p == 0 && t == 0) buf_load_pair<kStream>(ym_a, ym_b, rin, voff, 16 * step);        // row n/2 = n1 (n2/2): block 0, j2 = n2/2
        if (p == 0 && t == 0) { ya[0].y = (R) 0; yb[0].y = (R) 0; }                            // dsc_fft.h:227-228: real parts only at bins 0, n/2
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v[j] = C{ya[j].x - yb[j].y, ya[j].y + yb[j].x};                                     // Z[k] = Y_a + i Y_b
            own_x[T * j * CW] = ya[j].x + yb[j].y;                                              // Z[n - k] = conj Y_a + i conj Y_b
            own_y[T * j * CW] = yb[j].x - ya[j].y;
        }
        v[16] = C{ym_a.x, ym_b.x};                                  // (block 0, thread 0) Z[n/2] = Re Y_a + i Re Y_b; everybody else: overwritten below
    }
    __syncthreads();                                                // staging and the twiddle table visible
    {
        const int shift = p == 0 ? 0 : 1;                           // partner of j2: n2 - j2 in block 0, n2 - 1 - j2 otherwise
        const int pc = c < H ? c + H : c - H;
        const R *par_x = plane + pc + ((L - 31 * T) - shift - t) * CW;      // [T (31 - j) CW]: slot n2 - shift - (t + T j)
        const R *par_y = par_x + (L / 2) * CW;
#pragma unroll
        for (int j = 16; j < 32; ++j) {
            const C z = C{par_x[T * (31 - j) * CW], par_y[T * (31 - j) * CW]};
            if (!(j == 16 && t == 0 && p == 0)) v[j] = z;           // (that one is row n/2, loaded above; its slot index n2/2 is not a slot)
        }
    }
    lds_barrier();                                                  // staging reads done before the passes write the plane
    cols_passes<R, B, TWO, CW, true>(v, plane, wtab, tw_full, t, c);       // v[i B + p] = bin k2 = (t + T i) + COLS brev(p) of column (j1, col)
    const int inner = n1 * cc_n;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (work + (size_t) s * n * cc_n), 0, (int) (n * cc_n * CB), 0x00020000);
    const bool stores = live && !(self && h == 1);
    const int voff = stores ? (t * inner + j1 * cc_n + col) * CB : kOut;
    const int step = inner * CB;
#pragma unroll
    for (int qq = 0; qq < 32; ++qq) {
        const int k2 = t + T * (qq / B) + COLS * brev(qq % B, LOGB);
        const C w = twn[j1 * k2];
        buf_store<kStream>(cmulc(v[qq], w), rout, voff, (T * (qq / B) + COLS * brev(qq % B, LOGB)) * step);
    }
    (void) half;
}

template<typename R, int B, bool TWO, int CW>
void launch_real_split(const void *work, void *out, long long slices, int cc_n, int n2, const void *tw_full, hipStream_t stream) {
    using cfg = cols_cfg<R, B, TWO, CW>;
    constexpr size_t lds = cols_lds_bytes<R, B, TWO, CW>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) cols_real_split_kernel<R, B, TWO, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
    const int tiles = (cc_n + CW / 2 - 1) / (CW / 2);
    const long long grid = slices * (n2 / 2 + 1) * tiles;
    DSC_LAUNCH((cols_real_split_kernel<R, B, TWO, CW>), dim3((unsigned) grid), dim3(cfg::NT), lds, stream, (const cpx<R> *) work, (cpx<R> *) out, cc_n,
               tiles, n2, (const cpx<R> *) tw_full);
}
template<typename R, int B, bool TWO, int CW>
void launch_real_merge(const void *in, void *work, long long slices, int cc_n, int n1, const void *tw_full, const void *twn, hipStream_t stream) {
    using cfg = cols_cfg<R, B, TWO, CW>;
    constexpr size_t lds = cols_lds_bytes<R, B, TWO, CW>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) cols_real_merge_kernel<R, B, TWO, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
    const int tiles = (cc_n + CW / 2 - 1) / (CW / 2);
    const long long grid = slices * (n1 / 2 + 1) * tiles;
    DSC_LAUNCH((cols_real_merge_kernel<R, B, TWO, CW>), dim3((unsigned) grid), dim3(cfg::NT), lds, stream, (const cpx<R> *) in, (cpx<R> *) work,
               cc_n, n1, tiles, (const cpx<R> *) tw_full, (const cpx<R> *) twn);
}

#ifndef DSC_COLS_CW_64          // columns per tile of the f32 forms (A/B knobs, tools/build_file_variant.sh)
#define DSC_COLS_CW_64 128
#endif
#ifndef DSC_COLS_CW_128
#define DSC_COLS_CW_128 64
#endif
#ifndef DSC_COLS_CW_256
#define DSC_COLS_CW_256 32
#endif
#ifndef DSC_COLS_CW_512
#define DSC_COLS_CW_512 32
#endif
#ifndef DSC_COLS_CW_1024
#define DSC_COLS_CW_1024 16
#endif
#ifndef DSC_COLS_CW_2048
#define DSC_COLS_CW_2048 16
#endif
#ifndef DSC_COLS_CW_4096
#define DSC_COLS_CW_4096 8
#endif
template<typename R, int B, bool TWO, int CW, int MODE, bool INV>
void launch_cols_one(const void *in, void *out, long long slices, int inner, int in_axis, int in_len, int out_axis, const void *tw_full,
                     const void *tw_real, double scale, const cols_remap &rm, hipStream_t stream) {
    using cfg = cols_cfg<R, B, TWO, CW>;
    constexpr size_t lds = cols_lds_bytes<R, B, TWO, CW>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_cols_kernel<R, B, TWO, CW, MODE, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
    const int tiles = (inner + CW - 1) / CW;
    const long long n_tiles = slices * tiles;
    long long grid = n_tiles;
    if (cfg::PIPE && !TWO) {                                // persistent: one workgroup per CU walks the tiles
        static int cus[64];
        int dev = 0;
        DSC_KERNEL_CHECK(hipGetDevice(&dev));
        dev &= 63;
        if (cus[dev] == 0) DSC_KERNEL_CHECK(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
        if (grid > cus[dev]) grid = cus[dev];
    }
    DSC_LAUNCH((fft_cols_kernel<R, B, TWO, CW, MODE, INV>), dim3((unsigned) grid), dim3(cfg::NT), lds, stream, in, out, inner, tiles, (int) n_tiles,
               in_axis, in_len, out_axis, (const cpx<R> *) tw_full, (const cpx<R> *) tw_real, (R) scale, rm);
}

template<typename R, int B, bool TWO, int CW>
void launch_cols_mode(dsc_fft_mode mode, bool inverse, const void *in, void *out, long long slices, int inner, int in_axis, int in_len, int out_axis,
                      const void *tw_full, const void *tw_real, double scale, const cols_remap &rm, hipStream_t stream) {
    if (mode == DSC_MODE_R2C_PACKED)      launch_cols_one<R, B, TWO, CW, DSC_MODE_R2C_PACKED, false>(in, out, slices, inner, in_axis, in_len, out_axis, tw_full, tw_real, scale, rm, stream);
    else if (mode == DSC_MODE_C2R_PACKED) launch_cols_one<R, B, TWO, CW, DSC_MODE_C2R_PACKED, true>(in, out, slices, inner, in_axis, in_len, out_axis, tw_full, tw_real, scale, rm, stream);
    else if (mode == DSC_MODE_R2C_CAST && inverse) launch_cols_one<R, B, TWO, CW, DSC_MODE_R2C_CAST, true>(in, out, slices, inner, in_axis, in_len, out_axis, tw_full, tw_real, scale, rm, stream);
    else if (mode == DSC_MODE_R2C_CAST)   launch_cols_one<R, B, TWO, CW, DSC_MODE_R2C_CAST, false>(in, out, slices, inner, in_axis, in_len, out_axis, tw_full, tw_real, scale, rm, stream);
    else if (inverse)                     launch_cols_one<R, B, TWO, CW, DSC_MODE_C2C, true>(in, out, slices, inner, in_axis, in_len, out_axis, tw_full, tw_real, scale, rm, stream);
    else                                  launch_cols_one<R, B, TWO, CW, DSC_MODE_C2C, false>(in, out, slices, inner, in_axis, in_len, out_axis, tw_full, tw_real, scale, rm, stream);
}

}  // namespace

// Complex lengths with a column kernel.  32 .. 2048, and 4096 for c32 data.  Modes: C2C, R2C_CAST (a real tensor through dsc_fft /
// dsc_ifft, widened while loading), R2C_PACKED, C2R_PACKED.
bool dsc_fft_regs_cols_supports(int L, dsc_fft_mode mode, bool single_precision) {
    if (L == 32 || L == 64 || L == 128 || L == 256 || L == 512 || L == 1024 || L == 2048) return true;
    return L == 4096 && single_precision && mode == DSC_MODE_C2C;      // 8 columns per tile: the real modes (4-B samples: 32-B pieces) lose to the transpose route there
}

// Tensor [slices][axis][inner] (contiguous), transform along `axis`: in has in_axis elements along it of which in_len are
// used (the rest of the transform length is zero), out has out_axis.  Element counts are in each side's own element type
// (reals on the real side of the packed modes).  Every slice must stay below 2 GiB (32-bit buffer offsets): the caller checks.
static void launch_cols_len(const void *in, void *out, long long slices, int inner, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                            const void *tw_full, const void *tw_real, double scale, int in_axis, int in_len, int out_axis, const cols_remap &rm,
                            hipStream_t stream) {
    if (slices <= 0 || inner <= 0) return;
#define COLS_ARGS mode, inverse, in, out, slices, inner, in_axis, in_len, out_axis, tw_full, tw_real, scale, rm, stream
    if (single_precision) {
        switch (L) {
            case 32:   launch_cols_mode<float, 1, true, 256>(COLS_ARGS); break;
            case 64:   launch_cols_mode<float, 2, true, DSC_COLS_CW_64>(COLS_ARGS); break;
            case 128:  launch_cols_mode<float, 4, true, DSC_COLS_CW_128>(COLS_ARGS); break;
            case 256:  launch_cols_mode<float, 8, true, DSC_COLS_CW_256>(COLS_ARGS); break;
            case 512:  launch_cols_mode<float, 16, true, DSC_COLS_CW_512>(COLS_ARGS); break;
            case 1024:                                  // real rows are 4 B per column: 32 columns make them whole 128-B lines
                if (mode == DSC_MODE_R2C_PACKED) launch_cols_mode<float, 32, true, 32>(COLS_ARGS);
                else                             launch_cols_mode<float, 32, true, DSC_COLS_CW_1024>(COLS_ARGS);
                break;
            case 2048: launch_cols_mode<float, 2, false, DSC_COLS_CW_2048>(COLS_ARGS); break;
            default:   launch_cols_mode<float, 4, false, DSC_COLS_CW_4096>(COLS_ARGS); break;
        }
    } else {
        switch (L) {
            case 32:   launch_cols_mode<double, 1, true, 256>(COLS_ARGS); break;
            case 64:   launch_cols_mode<double, 2, true, 128>(COLS_ARGS); break;
            case 128:  launch_cols_mode<double, 4, true, 64>(COLS_ARGS); break;
            case 256:  launch_cols_mode<double, 8, true, 32>(COLS_ARGS); break;
            case 512:  launch_cols_mode<double, 16, true, 16>(COLS_ARGS); break;
            case 1024: launch_cols_mode<double, 32, true, 16>(COLS_ARGS); break;
            default:   launch_cols_mode<double, 2, false, 8>(COLS_ARGS); break;
        }
    }
#undef COLS_ARGS
}

void dsc_launch_fft_regs_cols(const void *in, void *out, long long slices, int inner, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                              const void *tw_full, const void *tw_real, double scale, int in_axis, int in_len, int out_axis, hipStream_t stream) {
    launch_cols_len(in, out, slices, inner, L, mode, inverse, single_precision, tw_full, tw_real, scale, in_axis, in_len, out_axis,
                    cols_remap{inner, 1, nullptr, 1}, stream);
}

// Complex transform of n = n1 n2 points along the middle axis of a contiguous [slices][n][inner] tensor (dsc_fft / dsc_ifft along a
// non-last axis, dsc.cpp:1977-1978 + the strided iterator, for lengths beyond the one-pass column kernel) as a four-step transform in
// TWO passes of the column kernel over HBM, j = j1 + n1 j2, k = n2 k1 + k2:
//   pass 1   view [slices][n2][n1 inner], lines over j2 (n2 points, n1 inner apart), times W_n^{j1 k2}           -> work [slices][k2][j1][inner]
//   pass 2   view [slices n2][n1][inner], lines over j1 (n1 points), bin k1 of slice (s, k2) to row n2 k1 + k2   -> out
// Both passes move pieces of a whole tile row (32 - 128 columns); the route through two transposes and a row transform is three
// passes.  mode: DSC_MODE_C2C or DSC_MODE_R2C_CAST (real input, widened while pass 1 loads).  work: slices n inner complex.
// tw1 / tw2: W_{n1}^m / W_{n2}^m (the complex plans of the two lengths), twn: W_n^m, m < n.  The caller checks the 2 GiB slice limit.
// columns per tile of the column kernel at complex length L (the table of launch_cols_len)
static int cols_tile_width(int L, bool single_precision) {
    if (L <= 32) return 256;
    if (L == 64) return DSC_COLS_CW_64;
    if (L == 128) return DSC_COLS_CW_128;
    if (L == 256) return DSC_COLS_CW_256;
    if (single_precision) return L == 512 ? DSC_COLS_CW_512 : L <= 2048 ? 16 : 8;
    return L <= 1024 ? 16 : 8;
}

// n = n1 n2: balanced (n1 <= n2: the longer lines go to pass 1, whose tiles are always whole); when the tensor has fewer columns than a
// pass-2 tile of n1-point lines is wide, bits move from n2 to n1 — longer lines have narrower tiles — as long as n2 stays >= 64
// (measured, tools/bench_axis_small_inner.py: 65536 points x 8 columns 22.6 % with 256 x 256).  cols: complex columns pass 2 sees.
bool dsc_fft_cols_4step_split(int n, bool single_precision, int cols, int *n1, int *n2) {
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    if ((1 << lg) != n || lg < 10) return false;
    int a = 1 << (lg / 2), b = n / a;
    static const int skew = [] { const char *e = getenv("DSC_COLS_4STEP_SKEW"); return e ? atoi(e) : 0; }();      // experiments: n1 >> skew
    for (int i = 0; i < skew && a > 64 && b < 2048; ++i) { a >>= 1; b <<= 1; }
    static const bool widen = getenv("DSC_COLS_4STEP_NO_WIDEN") == nullptr;
    while (widen && cols_tile_width(a, single_precision) > cols && b >= 128 && a < 2048) { a <<= 1; b >>= 1; }
    if (a < 32 || b < 32 || a > 2048 || b > 2048) return false;
    *n1 = a;
    *n2 = b;
    return true;
}

// Real transforms along the middle axis of [slices][n][2 cc_n] reals (see cols_real_split_kernel): cc_n = complex columns = half the real ones.
// rfft: in = the real tensor, out = [slices][n/2 + 1][2 cc_n] complex.  irfft: the other way round, scale = 1 / n.  work: slices n cc_n complex.
template<typename R>
static void launch_real_split_len(int L, const void *work, void *out, long long slices, int cc_n, int n2, const void *tw, hipStream_t st) {
    constexpr bool SP = sizeof(R) == 4;
    switch (L) {
        case 64:   launch_real_split<R, 2, true, 128>(work, out, slices, cc_n, n2, tw, st); break;
        case 128:  launch_real_split<R, 4, true, 64>(work, out, slices, cc_n, n2, tw, st); break;
        case 256:  launch_real_split<R, 8, true, 32>(work, out, slices, cc_n, n2, tw, st); break;
        case 512:  launch_real_split<R, 16, true, SP ? 32 : 16>(work, out, slices, cc_n, n2, tw, st); break;
        case 1024: launch_real_split<R, 32, true, 16>(work, out, slices, cc_n, n2, tw, st); break;
        default:   launch_real_split<R, 2, false, SP ? 16 : 8>(work, out, slices, cc_n, n2, tw, st); break;
    }
}
template<typename R>
static void launch_real_merge_len(int L, const void *in, void *work, long long slices, int cc_n, int n1, const void *tw, const void *twn, hipStream_t st) {
    constexpr bool SP = sizeof(R) == 4;
    switch (L) {
        case 64:   launch_real_merge<R, 2, true, 128>(in, work, slices, cc_n, n1, tw, twn, st); break;
        case 128:  launch_real_merge<R, 4, true, 64>(in, work, slices, cc_n, n1, tw, twn, st); break;
        case 256:  launch_real_merge<R, 8, true, 32>(in, work, slices, cc_n, n1, tw, twn, st); break;
        case 512:  launch_real_merge<R, 16, true, SP ? 32 : 16>(in, work, slices, cc_n, n1, tw, twn, st); break;
        case 1024: launch_real_merge<R, 32, true, 16>(in, work, slices, cc_n, n1, tw, twn, st); break;
        default:   launch_real_merge<R, 2, false, SP ? 16 : 8>(in, work, slices, cc_n, n1, tw, twn, st); break;
    }
}

void dsc_launch_rfft_cols_4step(const void *in, void *work, void *out, long long slices, int cc_n, int n1, int n2, bool single_precision,
                                const void *tw1, const void *tw2, const void *twn, hipStream_t stream) {
    launch_cols_len(in, work, slices, n1 * cc_n, n2, DSC_MODE_C2C, false, single_precision, tw2, nullptr, 1.0, n2, n2, n2,
                    cols_remap{n1 * cc_n, 1, twn, cc_n}, stream);
    if (single_precision) launch_real_split_len<float>(n1, work, out, slices, cc_n, n2, tw1, stream);
    else                  launch_real_split_len<double>(n1, work, out, slices, cc_n, n2, tw1, stream);
}

void dsc_launch_irfft_cols_4step(const void *in, void *work, void *out, long long slices, int cc_n, int n1, int n2, bool single_precision,
                                 const void *tw1, const void *tw2, const void *twn, double scale, hipStream_t stream) {
    if (single_precision) launch_real_merge_len<float>(n2, in, work, slices, cc_n, n1, tw2, twn, stream);
    else                  launch_real_merge_len<double>(n2, in, work, slices, cc_n, n1, tw2, twn, stream);
    launch_cols_len(work, out, slices * n2, cc_n, n1, DSC_MODE_C2C, true, single_precision, tw1, nullptr, scale, n1, n1, n1,
                    cols_remap{n2 * cc_n, n2, nullptr, 1}, stream);
}

void dsc_launch_fft_cols_4step(const void *in, void *work, void *out, long long slices, int inner, int n1, int n2, dsc_fft_mode mode, bool inverse,
                               bool single_precision, const void *tw1, const void *tw2, const void *twn, double scale, hipStream_t stream) {
    const int n = n1 * n2;
    launch_cols_len(in, work, slices, n1 * inner, n2, mode, inverse, single_precision, tw2, nullptr, 1.0, n2, n2, n2,
                    cols_remap{n1 * inner, 1, twn, inner}, stream);
    launch_cols_len(work, out, slices * n2, inner, n1, DSC_MODE_C2C, inverse, single_precision, tw1, nullptr, scale, n1, n1, n1,
                    cols_remap{n2 * inner, n2, nullptr, 1}, stream);
    (void) n;
}
