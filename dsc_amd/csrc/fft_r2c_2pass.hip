// fft_r2c_2pass.hip — long real FFTs in TWO passes over HBM: packed complex length L = L1 x 1024 with
// L1 = 32 B1 in {32, 64, 128, 256, 512, 1024}, i.e. real lengths 65536 (f64 only; f32 has fft_r2c_64k.hip), 131072, 262144
// (BASELINE config 5 in f64) and 524288, f32 and f64.  A row is too big for one CU, so (four-step, DIT):
//
//   j = j1 + L1 j2   (input),      k = 1024 k1 + k2   (output),      j1, k1 < L1,  j2, k2 < 1024
//   rows  A[j1][k2] = W_L^{j1 k2} * sum_{j2} z[j1 + L1 j2] W_1024^{j2 k2}       1024-point FFTs
//   cols  Z[1024 k1 + k2] = sum_{j1} A[j1][k2] W_L1^{j1 k1}                     L1-point FFTs
//   X from Z by the packed-real pass (dsc_fft.h:199-225), FUSED into the column kernel
//
// Both kernels keep their working set (16384 complex) in the registers of one 512-thread workgroup (32 complex
// per thread), with the building blocks of fft_regs_mid.hip:
//   rows kernel  16 adjacent j1 (128/256-B pieces of the input) x 1024 j2; 1024 = 32 x 32, one LDS exchange that
//                also turns the lanes from "j1 fastest" (coalesced strided loads) to "k2 fastest" (contiguous stores).
//                W_L^{j1 k2} = three table values (exact) multiplied together.
//   cols kernel  NC = 512 / B1 columns k2 x all L1 j1; L1 = 32 x B1; lanes = columns, so loads and stores are runs
//                of NC / 2 elements.  The real pass pairs bin (k1, k2) with (L1 - 1 - k1, 1024 - k2): the column
//                set of a workgroup is S_b = [H b + 1, H b + H] (H = NC / 2) plus its mirror M_b, closed under that
//                pairing (column 0, which pairs with itself, takes the place of the duplicate 512 in the last
//                block); partners meet through an LDS staging plane, one component at a time.
// Twice the algorithmic traffic (the ceiling is half the copy rate).  The inverse runs the same two kernels backwards.
// Reference: dsc_rfft / dsc_irfft (dsc/src/dsc.cpp:2102-2260, dsc_fft.h:57-238).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include "fft_regs_common.h"

namespace {

// Cache policy of the INTERMEDIATE (the work array between the two kernels): default (cached) — it is written by one kernel
// and read back by the next, and when the launches are cut into chunks of rows it can stay in the 256 MiB Infinity Cache.
// The external rows are touched once and stay non-temporal (kStream).
#ifndef DSC_2PASS_WORK_POLICY
#define DSC_2PASS_WORK_POLICY 0
#endif
constexpr int kWork = DSC_2PASS_WORK_POLICY;
// The spectrum rows of the REAL transforms have a pitch of L + 1 bins, so a workgroup's runs start and end inside 128-B lines it
// shares with its neighbours: reading them with the default policy (the shared lines are found in the L2) is 4-5 % faster than
// non-temporal (irfft f64 262144: 3.99 -> 3.80 ms, f32 131072: 1.71 -> 1.62 ms); writing them stays non-temporal (cached
// stores: 3.59 -> 3.85 ms).
#ifndef DSC_2PASS_BINS_LOAD_POLICY
#define DSC_2PASS_BINS_LOAD_POLICY kCached
#endif
constexpr int kBinsLoadReal = DSC_2PASS_BINS_LOAD_POLICY;

constexpr int kPQ = 1060;                    // rows kernel: plane pitch per line (values), = 4 mod 32: conflict-free both ways

template<typename R> constexpr int rows_lds_bytes() { return (16 * kPQ + 2 * 1024) * (int) sizeof(R); }          // plane + W_1024
// threads of the column kernel: 512, or 1024 for f32 with L1 = 256 (N = 524288) — with 512 its runs are 32 columns = 256 B,
// skewed on the spectrum side (measured 1.33 ms for that kernel against 0.78 ms for the rows kernel of the same transform)
template<typename R, int B1> constexpr int cols_threads() { return (sizeof(R) == 4 && B1 >= 8) ? 1024 : 512; }
template<typename R, int B1> constexpr int cols_lds_bytes() { return (32 * cols_threads<R, B1>() + 2 * 32 * B1 + cols_threads<R, B1>()) * (int) sizeof(R); }  // plane + W_L1 + slack: the unused partner read of bin 0 lands one row past the plane
template<typename R> constexpr int waves_per_eu() { return sizeof(R) == 8 ? 2 : 4; }     // f32: <= 128 VGPRs, two workgroups per CU

// W_L^{j1 (tau + 32 k3)}, k3 = 4 a + b: base W_L^{j1 tau} times A[a] = W_L^{128 j1 a} times Bq[b] = W_L^{32 j1 b}, all three
// exact table values (two products per twiddle).  Applied to v[brev(k3)] (BREV) or v[k3].
template<typename R, bool CONJ, bool BREV>
__device__ __forceinline__ void four_step_twiddle(cpx<R> (&v)[32], const cpx<R> *twL, int j1, int tau) {
    using C = cpx<R>;
    const C base = twL[j1 * tau];
    const C b1 = twL[32 * j1], b2 = twL[64 * j1], b3 = twL[96 * j1];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const C bq = a == 0 ? base : cmul(base, twL[128 * j1 * a]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const C w = b == 0 ? bq : b == 1 ? cmul(bq, b1) : b == 2 ? cmul(bq, b2) : cmul(bq, b3);
            const int k3 = 4 * a + b;
            const int r = BREV ? brev(k3, 5) : k3;
            v[r] = CONJ ? cmulc(v[r], w) : cmul(v[r], w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rows kernel.  Forward (INV = false): z (row of packed reals) -> work[j1][k2].  Inverse: work -> z * scale.
//   "writer" lanes  tid = 16 t + q : line j1 = 16 a + q, holds index 32 j2' + t  (pieces of 16 elements across q)
//   "reader" lanes  tid = 32 q + tau: line j1 = 16 a + q, holds index tau + 32 k3 (pieces of 32 elements across tau)
// CAST (forward): the external rows hold REAL samples, widened on the way in (dsc_fft of a real tensor, dsc.cpp:1984-1988)
template<typename R, int B1, bool INV, bool CAST = false>
__global__ __launch_bounds__(512, (waves_per_eu<R>())) void two_pass_rows_kernel(const cpx<R> *__restrict__ in, cpx<R> *__restrict__ out,
                                                                              const cpx<R> *__restrict__ twL, R scale, long long ext_pitch_b,
                                                                              int ext_len_b) {
    using C = cpx<R>;
    constexpr int L1 = 32 * B1, L = L1 * 1024, CB = (int) sizeof(C), GROUPS = L1 / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *w1024 = (C *) (plane + 16 * kPQ);
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 512) w1024[i] = twL[i * L1];            // W_1024^m = W_L^{L1 m}
    const long long row = blockIdx.x / GROUPS;
    const int a = blockIdx.x % GROUPS;
    // The external side (forward: the input samples; inverse: the output samples) has a row pitch of ext_pitch_b bytes and
    // ext_len_b valid bytes: shorter rows are zero padded by the descriptor's range check (dsc.cpp:2125-2133).
    const __amdgpu_buffer_rsrc_t rin = INV ? __builtin_amdgcn_make_buffer_rsrc((void *) (in + row * L), 0, L * CB, 0x00020000)
                                           : __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) in + row * ext_pitch_b), 0, ext_len_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = INV ? __builtin_amdgcn_make_buffer_rsrc((void *) ((char *) out + row * ext_pitch_b), 0, ext_len_b, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void *) (out + row * L), 0, L * CB, 0x00020000);
    const int wq = tid & 15, wt = tid >> 4;              // writer mapping
    const int rq = tid >> 5, rtau = tid & 31;            // reader mapping
    constexpr int EB = CAST ? (int) sizeof(R) : CB;      // bytes per external sample
    const int zoff = ((16 * a + wq) + L1 * wt) * EB;     // z[j1 + L1 (32 j2' + t)]: + j2' * 32 L1 elements
    const int aoff = ((16 * a + rq) * 1024 + rtau) * CB; // A[j1][tau + 32 k3]:      + k3 * 32 elements
    const int j1r = 16 * a + rq;
    constexpr int ZSTEP = 32 * L1 * EB, ASTEP = 32 * CB;

    C u[32], v[32];
    if constexpr (!INV) {
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m] = CAST ? buf_load_real<kStream>(rin, zoff, m * ZSTEP, R{}) : buf_load<kStream>(rin, zoff, m * ZSTEP, R{});
        __syncthreads();
        dft_n<R, false, 32>(u);                                               // over j2' -> k2' in u[brev(k2')]
#pragma unroll
        for (int k = 1; k < 32; ++k) u[brev(k, 5)] = cmul(u[brev(k, 5)], w1024[wt * k]);
        R *wr = plane + wq * kPQ + wt;
        const R *rd = plane + rq * kPQ + rtau * 33;
#pragma unroll
        for (int k = 0; k < 32; ++k) wr[k * 33] = u[brev(k, 5)].x;
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 32; ++m) v[m].x = rd[m];
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 32; ++k) wr[k * 33] = u[brev(k, 5)].y;
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 32; ++m) v[m].y = rd[m];
        dft_n<R, false, 32>(v);                                               // over t -> k3 in v[brev(k3)]
        four_step_twiddle<R, false, true>(v, twL, j1r, rtau);
#pragma unroll
        for (int k3 = 0; k3 < 32; ++k3) buf_store<kWork>(v[brev(k3, 5)], rout, aoff, k3 * ASTEP);
    } else {
#pragma unroll
        for (int k3 = 0; k3 < 32; ++k3) v[k3] = buf_load<kWork>(rin, aoff, k3 * ASTEP, R{});
        __syncthreads();
        four_step_twiddle<R, true, false>(v, twL, j1r, rtau);
        dft_n<R, true, 32>(v);                                                // over k3 -> t in v[brev(t)]
#pragma unroll
        for (int t = 1; t < 32; ++t) v[brev(t, 5)] = cmulc(v[brev(t, 5)], w1024[t * rtau]);
        R *wr = plane + rq * kPQ + rtau * 33;
        const R *rd = plane + wq * kPQ + wt;
#pragma unroll
        for (int t = 0; t < 32; ++t) wr[t] = v[brev(t, 5)].x;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 32; ++k) u[k].x = rd[k * 33];
        lds_barrier();
#pragma unroll
        for (int t = 0; t < 32; ++t) wr[t] = v[brev(t, 5)].y;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 32; ++k) u[k].y = rd[k * 33];
        dft_n<R, true, 32>(u);                                                // over tau -> j2' in u[brev(j2')]
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const C r = u[brev(m, 5)];
            buf_store<kStream>(C{r.x * scale, r.y * scale}, rout, zoff, m * ZSTEP);
        }
    }
}

// Staging plane of the cols kernel: [k1][ell], L1 rows of NC values.  Element e's own slot is at relative row
// (B1 i' + 32 k3), its partner's at (L1 - B1) - (B1 i' + 32 k3); two base registers per direction keep every LDS
// offset a 16-bit immediate (the high halves are computed from an opaque copy of the offset, or hipcc folds the two
// bases back into one and materialises an address register for every offset beyond 64 KiB).
template<typename R, int B1>
struct stage_ptrs {
    static constexpr int NC = cols_threads<R, B1>() / B1, L1 = 32 * B1, HALF = L1 / 2;
    R *mine_lo, *mine_hi;
    const R *theirs_lo, *theirs_hi;
    __device__ __forceinline__ stage_ptrs(R *plane, int t, int ell, int ellp, bool col0) {
        int hi = HALF * NC;
        asm volatile("" : "+v"(hi));
        mine_lo = plane + t * NC + ell;
        mine_hi = plane + (t * NC + ell + hi);
        const int th = (B1 - 1 - t) * NC + ellp + (col0 ? NC : 0);
        theirs_lo = plane + th;
        theirs_hi = plane + (th + hi);
    }
    __device__ __forceinline__ R &mine(int rel) const { return rel < HALF ? mine_lo[rel * NC] : mine_hi[(rel - HALF) * NC]; }
    __device__ __forceinline__ R theirs(int rel) const { return rel < HALF ? theirs_lo[rel * NC] : theirs_hi[(rel - HALF) * NC]; }
};

// ------------------------------------------------------------------------------------------------
// cols kernel.  Forward: work[j1][k2] -> X (L + 1 bins).  Inverse: Y (L + 1 bins) -> work[j1][k2].
//   lanes tid = NC t + ell: local column ell (S side 0..H-1, mirror side H..NC-1), slice t of the L1-point axis (j1 = B1 i + t)
//   element e = B1 i' + p of a thread after the transform: k1 = t + B1 i' + 32 k3, k3 = brev(p)
//   REAL = false: plain complex transform (dsc_fft / dsc_ifft): columns NC b + ell, no pairing, rows of L bins
// CAST (inverse complex): the bins rows hold REAL values (dsc_ifft of a real tensor)
template<typename R, int B1, bool INV, bool REAL, bool CAST = false>
__global__ __launch_bounds__((cols_threads<R, B1>()), (waves_per_eu<R>())) void two_pass_cols_kernel(const cpx<R> *__restrict__ in, cpx<R> *__restrict__ out,
                                                                              const cpx<R> *__restrict__ twL, const cpx<R> *__restrict__ tw_real,
                                                                              long long bins_pitch, int bins_len) {
    using C = cpx<R>;
    constexpr int L1 = 32 * B1, L = L1 * 1024, CB = (int) sizeof(C);
    constexpr int NC = cols_threads<R, B1>() / B1, H = NC / 2, BLOCKS = 1024 / NC, CPT = 32 / B1, LOGB = ilog2(B1);
    constexpr int WSTEP = B1 * 1024 * CB;                 // work[(B1 i + t)][col]: + i * B1 rows
    constexpr int BSTEP = 1024 * CB;                      // bin 1024 k1 + col:     + k1 * 1024 bins
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *wl1 = (C *) (plane + 32 * cols_threads<R, B1>());
    const int tid = threadIdx.x;
    for (int i = tid; i < L1; i += cols_threads<R, B1>()) wl1[i] = twL[i * 1024];      // W_L1^m = W_L^{1024 m}
    const long long row = blockIdx.x / BLOCKS;
    const int b = blockIdx.x % BLOCKS;
    const int ell = tid % NC, t = tid / NC;
    const bool last = b == BLOCKS - 1;
    const bool col0 = REAL && last && ell == H;                               // column 0 replaces the duplicate 512
    const int col = !REAL ? NC * b + ell : col0 ? 0 : ell < H ? H * b + 1 + ell : 1024 - H - H * b + (ell - H);
    const int ellp = (last && (ell == H - 1 || ell == H)) ? ell : NC - 1 - ell;   // local column of the pairing partner
    const C *work = INV ? out + row * L : in + row * L;
    // bins_pitch / bins_len (in bins): the inverse reads rows of any length, missing bins as zero (dsc.cpp:2149-2157)
    constexpr int BB = CAST ? (int) sizeof(R) : CB;                           // bytes per external bin
    const C *bins = INV ? (const C *) ((const char *) in + row * bins_pitch * BB) : out + row * bins_pitch;
    const __amdgpu_buffer_rsrc_t rwork = __builtin_amdgcn_make_buffer_rsrc((void *) work, 0, L * CB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbins = __builtin_amdgcn_make_buffer_rsrc((void *) bins, 0, bins_len * BB, 0x00020000);
    const int woff = (t * 1024 + col) * CB;
    const int boff = col * CB;
    constexpr int kBinsLoad = REAL ? kBinsLoadReal : kStream;

    // W_2L^k for this thread's bins k = 1024 k1 + col, k1 = t + B1 i' + 32 k3: W_2L^{col} W_2L1^{t} times the constant
    // W_64^{i' + (32 / B1) k3}
    const C wt0 = REAL ? cmul(tw_real[col], tw_real[1024 * t]) : C{(R) 1, (R) 0};
    const stage_ptrs<R, B1> sp(plane, t, ell, ellp, col0);

    C u[32], v[32];
    if constexpr (!INV) {
#pragma unroll
        for (int i = 0; i < 32; ++i) u[i] = buf_load<kWork>(rwork, woff, i * WSTEP, R{});
        __syncthreads();
        dft_n<R, false, 32>(u);                                               // over i -> k' in u[brev(k')]
        if constexpr (B1 > 1) {
#pragma unroll
            for (int k = 1; k < 32; ++k) u[brev(k, 5)] = cmul(u[brev(k, 5)], wl1[t * k]);
            // exchange: plane[k'][t][ell]; thread (ell, t) then holds k' = t + B1 i', all slices t'
            R *wr = plane + t * NC + ell;
            const R *rd = plane + t * B1 * NC + ell;
#pragma unroll
            for (int k = 0; k < 32; ++k) wr[k * B1 * NC] = u[brev(k, 5)].x;
            lds_barrier();
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int tp = 0; tp < B1; ++tp) v[i * B1 + tp].x = rd[(i * B1 * B1 + tp) * NC];
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 32; ++k) wr[k * B1 * NC] = u[brev(k, 5)].y;
            lds_barrier();
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int tp = 0; tp < B1; ++tp) v[i * B1 + tp].y = rd[(i * B1 * B1 + tp) * NC];
            lds_barrier();
            dft_columns<R, false, B1>(v, std::make_integer_sequence<int, CPT>{});   // v[B1 i' + p] = Z[k1 = t + B1 i' + 32 brev(p)][col]
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = u[brev(k, 5)];                // one thread per column: v[k1]
        }

        if constexpr (!REAL) {
#pragma unroll
            for (int e = 0; e < 32; ++e) buf_store<kStream>(v[e], rbins, boff, (t + B1 * (e / B1) + 32 * brev(e % B1, LOGB)) * BSTEP);
            return;
        }
        // ---- packed-real pass: a = Z[k1][col] (own), b = Z[L1 - 1 - k1][1024 - col] (L1 - k1 in column 0), through the plane
        R bx[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) sp.mine(B1 * (e / B1) + 32 * brev(e % B1, LOGB)) = v[e].x;
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) bx[e] = sp.theirs((L1 - B1) - B1 * (e / B1) - 32 * brev(e % B1, LOGB));
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) sp.mine(B1 * (e / B1) + 32 * brev(e % B1, LOGB)) = v[e].y;
        lds_barrier();
        C wt = wt0;
        asm volatile("" : "+v"(wt.x), "+v"(wt.y));          // the 32 twiddles derived from it must not be computed (and kept) earlier
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int ip = e / B1, k3 = brev(e % B1, LOGB);
            const int k1 = t + B1 * ip + 32 * k3;
            const R by = sp.theirs((L1 - B1) - B1 * ip - 32 * k3);
            const C w = cmul(wt, C{(R) root64_re(ip + (32 / B1) * k3), (R) root64_im(ip + (32 / B1) * k3)});
            const R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;                // -(i/2) W_2L^k
            const R ax = v[e].x, ay = v[e].y;
            const R sx = ax + bx[e], sy = ay - by, dx = ax - bx[e], dy = ay + by;
            C xk = C{(R) 0.5 * sx + (dx * wqx - dy * wqy), (R) 0.5 * sy + (dx * wqy + dy * wqx)};
            if (e == 0 && col0 && t == 0) {                                   // k = 0: X[0], X[L] real (dsc_fft.h:221-225); its "partner" read is unused
                xk = C{ax + ay, (R) 0};
                buf_store<kStream>(C{ax - ay, (R) 0}, rbins, L * CB, 0);
            }
            buf_store<kStream>(xk, rbins, boff, k1 * BSTEP);
        }
    } else {
        // ---- load the bins in the layout the forward kernel leaves them in (natural k3 order), pre-pass (dsc_fft.h:194-228)
#pragma unroll
        for (int e = 0; e < 32; ++e)
            v[e] = CAST ? buf_load_real<kStream>(rbins, col * BB, (t + B1 * (e / B1) + 32 * (e % B1)) * 1024 * BB, R{})
                        : buf_load<kBinsLoad>(rbins, boff, (t + B1 * (e / B1) + 32 * (e % B1)) * BSTEP, R{});
        C ylast = C{(R) 0, (R) 0};
        if (col0 && t == 0) { ylast = buf_load<kBinsLoad>(rbins, L * CB, 0, R{}); v[0].y = (R) 0; }      // real parts only at k = 0 and k = L
        __syncthreads();
        if constexpr (REAL) {
        R bx[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) sp.mine(B1 * (e / B1) + 32 * (e % B1)) = v[e].x;
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) bx[e] = sp.theirs((L1 - B1) - B1 * (e / B1) - 32 * (e % B1));
        if (col0 && t == 0) bx[0] = ylast.x;                                  // bin 0 pairs with bin L
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) sp.mine(B1 * (e / B1) + 32 * (e % B1)) = v[e].y;
        lds_barrier();
        C wt = wt0;
        asm volatile("" : "+v"(wt.x), "+v"(wt.y));
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int ip = e / B1, k3 = e % B1;
            R by = sp.theirs((L1 - B1) - B1 * ip - 32 * k3);
            if (e == 0 && col0 && t == 0) by = (R) 0;
            const C w = cmul(wt, C{(R) root64_re(ip + (32 / B1) * k3), (R) root64_im(ip + (32 / B1) * k3)});
            const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;                 // (i/2) conj(W_2L^k)
            const R ax = v[e].x, ay = v[e].y;
            const R sx = ax + bx[e], sy = ay - by, dx = ax - bx[e], dy = ay + by;
            v[e] = C{(R) 0.5 * sx + (dx * wqx - dy * wqy), (R) 0.5 * sy + (dx * wqy + dy * wqx)};
        }
        }
        if constexpr (B1 > 1) {
            if constexpr (REAL) lds_barrier();
            // ---- inverse L1-point transform over k1 = k' + 32 k3: B1-point over k3 -> t', twiddle, exchange, 32-point over k'
            dft_columns<R, true, B1>(v, std::make_integer_sequence<int, CPT>{});      // v[B1 i' + p]: t' = brev(p)
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int tp = 1; tp < B1; ++tp)
                    v[i * B1 + brev(tp, LOGB)] = cmulc(v[i * B1 + brev(tp, LOGB)], wl1[tp * (t + B1 * i)]);
            R *wr = plane + t * B1 * NC + ell;                                // plane[k' = t + B1 i'][t'][ell]
            const R *rd = plane + t * NC + ell;                               // thread (ell, t) reads all k' of its slice t
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int tp = 0; tp < B1; ++tp) wr[(i * B1 * B1 + tp) * NC] = v[i * B1 + brev(tp, LOGB)].x;
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 32; ++k) u[k].x = rd[k * B1 * NC];
            lds_barrier();
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int tp = 0; tp < B1; ++tp) wr[(i * B1 * B1 + tp) * NC] = v[i * B1 + brev(tp, LOGB)].y;
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 32; ++k) u[k].y = rd[k * B1 * NC];
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) u[k] = v[k];
        }
        dft_n<R, true, 32>(u);                                                // over k' -> i in u[brev(i)]
#pragma unroll
        for (int i = 0; i < 32; ++i) buf_store<kWork>(u[brev(i, 5)], rwork, woff, i * WSTEP);
    }
}

// in_pitch / in_len: pitch and valid length of the INPUT rows in input elements (REAL: reals forward, bins inverse; complex
// transforms: complex samples both ways)
template<typename R, int B1, bool REAL, bool CAST = false>
void launch_pair(const void *in, void *out, long long rows, void *work, const void *tw_full, const void *tw_real, bool inverse,
                 long long in_pitch, int in_len, hipStream_t stream) {
    using C = cpx<R>;
    constexpr int L = 32 * B1 * 1024;
    constexpr int rl = rows_lds_bytes<R>(), cl = cols_lds_bytes<R, B1>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) two_pass_rows_kernel<R, B1, false, CAST>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) two_pass_rows_kernel<R, B1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) two_pass_cols_kernel<R, B1, false, REAL>, hipFuncAttributeMaxDynamicSharedMemorySize, cl));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) two_pass_cols_kernel<R, B1, true, REAL, CAST>, hipFuncAttributeMaxDynamicSharedMemorySize, cl));
    }
    const dim3 grid((unsigned) (rows * 2 * B1));          // L1 / 16 row groups per transform
    constexpr int CT = cols_threads<R, B1>();
    const dim3 cgrid((unsigned) (rows * (1024 / (CT / B1))));   // 1024 / NC column blocks per transform
    constexpr int ext_b = (REAL || CAST) ? (int) sizeof(R) : (int) sizeof(C);  // bytes per external time-domain element
    constexpr long long full_row_b = (long long) L * sizeof(C);
    constexpr int out_bins = REAL ? L + 1 : L;
    if (!inverse) {
        DSC_LAUNCH((two_pass_rows_kernel<R, B1, false, CAST>), grid, dim3(512), rl, stream, (const C *) in, (C *) work, (const C *) tw_full, (R) 1,
                           in_pitch * ext_b, (int) (in_len * ext_b));
        DSC_LAUNCH((two_pass_cols_kernel<R, B1, false, REAL>), cgrid, dim3(CT), cl, stream, (const C *) work, (C *) out, (const C *) tw_full,
                           (const C *) tw_real, (long long) out_bins, out_bins);
    } else {
        DSC_LAUNCH((two_pass_cols_kernel<R, B1, true, REAL, CAST>), cgrid, dim3(CT), cl, stream, (const C *) in, (C *) work, (const C *) tw_full,
                           (const C *) tw_real, in_pitch, in_len);
        DSC_LAUNCH((two_pass_rows_kernel<R, B1, true>), grid, dim3(512), rl, stream, (const C *) work, (C *) out, (const C *) tw_full,
                           (R) (1.0 / (double) L), full_row_b, (int) full_row_b);                 // 2/(2n) (dsc_fft.h:232) = 1/n (:168-175)
    }
}

template<typename R, bool REAL>
void launch_len(int L, bool cast, const void *in, void *out, long long rows, void *work, const void *tw_full, const void *tw_real, bool inverse,
                long long in_pitch, int in_len, hipStream_t stream) {
    switch (L) {
        case 32768:  launch_pair<R, 1, REAL>(in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream); break;
        case 65536:  launch_pair<R, 2, REAL>(in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream); break;
        case 131072: launch_pair<R, 4, REAL>(in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream); break;
        case 524288: launch_pair<R, 16, REAL>(in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream); break;
        case 1048576: launch_pair<R, 32, REAL>(in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream); break;
        default:
            if constexpr (!REAL) { if (cast) { launch_pair<R, 8, false, true>(in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream); break; } }
            launch_pair<R, 8, REAL>(in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream); break;
    }
}

}  // namespace

bool dsc_fft_two_pass_supports(int L, bool single_precision) {
    if (L == 32768) return !single_precision;             // f32: fft_r2c_64k.hip, one pass
    return L == 65536 || L == 131072 || L == 262144 || L == 524288 || L == 1048576;
}

// forward: in = [rows][in_pitch] reals of which in_len <= 2L are transformed (the rest of the length is zero), out =
// [rows][L + 1] bins; inverse: in = [rows][in_pitch] bins of which in_len <= L + 1 are used, out = [rows][2L] reals.
// work: rows * L complex of scratch.  tw_full: W_L^k, k < L; tw_real: W_{2L}^k, k <= L (the REAL plan's own tables).
void dsc_launch_rfft_two_pass(const void *in, void *out, long long rows, int L, bool inverse, bool single_precision, void *work,
                              const void *tw_full, const void *tw_real, long long in_pitch, int in_len, hipStream_t stream) {
    if (rows <= 0) return;
    if (single_precision) launch_len<float, true>(L, false, in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream);
    else                  launch_len<double, true>(L, false, in, out, rows, work, tw_full, tw_real, inverse, in_pitch, in_len, stream);
}

// complex transforms of the same lengths: in = [rows][in_pitch] complex of which in_len <= L are transformed, out = [rows][L]
// cast (L = 262144 only): the input rows hold REAL values (dsc_fft / dsc_ifft of a real tensor); in_pitch / in_len count reals
void dsc_launch_fft_two_pass(const void *in, void *out, long long rows, int L, bool inverse, bool cast, bool single_precision, void *work,
                             const void *tw_full, long long in_pitch, int in_len, hipStream_t stream) {
    if (rows <= 0) return;
    if (single_precision) launch_len<float, false>(L, cast, in, out, rows, work, tw_full, tw_full, inverse, in_pitch, in_len, stream);
    else                  launch_len<double, false>(L, cast, in, out, rows, work, tw_full, tw_full, inverse, in_pitch, in_len, stream);
}
