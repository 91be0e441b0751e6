// fft_r2c_256k_f64_2pass.hip — 262144-point real FFT in f64 (BASELINE config 5) in TWO passes over HBM.
//
// Packed length L = 131072 complex = 2 MiB per row = 128 x 1024 (four-step, decimation in time):
//
//   j = j1 + 128 j2   (input),      k = 1024 k1 + k2   (output),      j1, k1 < 128,  j2, k2 < 1024
//   rows  A[j1][k2] = W_L^{j1 k2} * sum_{j2} z[j1 + 128 j2] W_1024^{j2 k2}      1024-point FFTs
//   cols  Z[1024 k1 + k2] = sum_{j1} A[j1][k2] W_128^{j1 k1}                    128-point FFTs
//   X from Z by the packed-real pass (dsc_fft.h:199-225), FUSED into the column kernel
//
// Both kernels keep their 256 KiB working set in the registers of one 512-thread workgroup (32 c64 per
// thread), with the building blocks of fft_regs_mid.hip:
//   rows kernel  16 adjacent j1 (256-B pieces of the input) x 1024 j2; 1024 = 32 x 32, one LDS exchange that
//                also turns the lanes from "j1 fastest" (coalesced strided loads) to "k2 fastest" (512-B stores)
//   cols kernel  128 columns k2 x 128 j1; 128 = 32 x 4; lanes = columns, so loads and stores are 1 KiB runs.
//                The real pass pairs bin (k1, k2) with (127 - k1, 1024 - k2): the column set of a workgroup is
//                S_b = [64 b + 1, 64 b + 64] plus its mirror M_b = [960 - 64 b, 1023 - 64 b], closed under
//                that pairing (column 0, which pairs with itself, takes the place of the duplicate 512 in
//                b = 7); partners meet through an LDS staging plane, one component at a time.
// 8 MiB of traffic per row against 4 MiB algorithmic (the three-pass version moved 12 MiB).
// The inverse runs the same two kernels backwards (pre-pass fused into the column kernel).
// Reference: dsc_rfft / dsc_irfft for F64 / C64 (dsc/src/dsc.cpp:2102-2260, dsc_fft.h:57-238).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include "fft_regs_common.h"

namespace {

using C = cpx<double>;

constexpr int kL = 131072;
constexpr int kPQ = 1060;                                     // rows kernel: plane pitch per line (doubles), = 4 mod 32
constexpr int kRowsLds = (16 * kPQ + 2 * 1024) * 8;          // plane + W_1024 (c64)
constexpr int kColsLds = (128 * 128 + 2 * 128) * 8;          // plane / staging + W_128 (c64)

// W_L^{j1 (tau + 32 k3)}, k3 = 0..31, applied to v[brev(k3)] (BREV) or v[k3]: base = W_L^{j1 tau}, step = W_4096^{j1};
// powers by products of at most seven factors (f64: far inside the 1e-12 tolerance)
template<bool CONJ, bool BREV>
__device__ __forceinline__ void four_step_twiddle(C (&v)[32], C base, C s) {
    const C p1 = s, p2 = cmul(s, s), p3 = cmul(p2, s), p4 = cmul(p2, p2);
    C q = C{1.0, 0.0};
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const C bq = cmul(base, q);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const C w = b == 0 ? bq : b == 1 ? cmul(bq, p1) : b == 2 ? cmul(bq, p2) : cmul(bq, p3);
            const int k3 = 4 * a + b;
            const int r = BREV ? brev(k3, 5) : k3;
            v[r] = CONJ ? cmulc(v[r], w) : cmul(v[r], w);
        }
        q = cmul(q, p4);
    }
}

// ------------------------------------------------------------------------------------------------
// rows kernel.  Forward (INV = false): z (row of packed reals) -> work[j1][k2].  Inverse: work -> z * scale.
//   "writer" lanes  tid = 16 t + q : line j1 = 16 a + q, holds index 32 j2' + t  (256-B pieces across q)
//   "reader" lanes  tid = 32 q + tau: line j1 = 16 a + q, holds index tau + 32 k3 (512-B pieces across tau)
template<bool INV>
__global__ __launch_bounds__(512, 2) void c5_rows_kernel(const C *__restrict__ in, C *__restrict__ out, const C *__restrict__ twL, double scale) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *plane = lds;
    C *w1024 = (C *) (lds + 16 * kPQ);
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 512) w1024[i] = twL[i * 128];           // W_1024^m = W_L^{128 m}
    const long long row = blockIdx.x >> 3;
    const int a = blockIdx.x & 7;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) (in + row * kL), 0, kL * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (out + row * kL), 0, kL * 16, 0x00020000);
    const int wq = tid & 15, wt = tid >> 4;              // writer mapping
    const int rq = tid >> 5, rtau = tid & 31;            // reader mapping
    const int zoff = ((16 * a + wq) + 128 * wt) * 16;    // z[j1 + 128 (32 j2' + t)]: + j2' * 65536 B
    const int aoff = ((16 * a + rq) * 1024 + rtau) * 16; // A[j1][tau + 32 k3]:        + k3 * 512 B
    const int j1r = 16 * a + rq;

    C u[32], v[32];
    if constexpr (!INV) {
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m] = buf_load<kStream>(rin, zoff, m * 65536, 0.0);
        __syncthreads();
        dft_n<double, false, 32>(u);                                          // over j2' -> k2' in u[brev(k2')]
#pragma unroll
        for (int k = 1; k < 32; ++k) u[brev(k, 5)] = cmul(u[brev(k, 5)], w1024[wt * k]);
        double *wr = plane + wq * kPQ + wt;
        const double *rd = plane + rq * kPQ + rtau * 33;
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
        dft_n<double, false, 32>(v);                                          // over t -> k3 in v[brev(k3)]
        four_step_twiddle<false, true>(v, twL[j1r * rtau], twL[32 * j1r]);
#pragma unroll
        for (int k3 = 0; k3 < 32; ++k3) buf_store<kStream>(v[brev(k3, 5)], rout, aoff, k3 * 512);
    } else {
#pragma unroll
        for (int k3 = 0; k3 < 32; ++k3) v[k3] = buf_load<kStream>(rin, aoff, k3 * 512, 0.0);
        __syncthreads();
        four_step_twiddle<true, false>(v, twL[j1r * rtau], twL[32 * j1r]);
        dft_n<double, true, 32>(v);                                           // over k3 -> t in v[brev(t)]
#pragma unroll
        for (int t = 1; t < 32; ++t) v[brev(t, 5)] = cmulc(v[brev(t, 5)], w1024[t * rtau]);
        double *wr = plane + rq * kPQ + rtau * 33;
        const double *rd = plane + wq * kPQ + wt;
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
        dft_n<double, true, 32>(u);                                           // over tau -> j2' in u[brev(j2')]
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const C r = u[brev(m, 5)];
            buf_store<kStream>(C{r.x * scale, r.y * scale}, rout, zoff, m * 65536);
        }
    }
}


// Staging plane accessors of the cols kernel: element e's own slot is at row offset (4 i' + 32 k3), its partner's at
// (124 - 4 i' - 32 k3), times 128 doubles.  Two base registers per direction keep every offset a 16-bit immediate.
struct stage_ptrs {
    double *mine_lo, *mine_hi;                 // relative rows 0..63 / 64..127
    const double *theirs_lo, *theirs_hi;
};
__device__ __forceinline__ stage_ptrs make_stage(double *plane, int t, int ell, int ellp, bool col0) {
    stage_ptrs p;
    // the high halves are computed from an opaque copy of the index, or hipcc folds the two bases back into one
    // and then materialises a separate address register for every offset beyond 64 KiB
    int hi = 64 * 128;
    asm volatile("" : "+v"(hi));
    p.mine_lo = plane + t * 128 + ell;
    p.mine_hi = plane + (t * 128 + ell + hi);
    const int th = (3 - t) * 128 + ellp + (col0 ? 128 : 0);
    p.theirs_lo = plane + th;
    p.theirs_hi = plane + (th + hi);
    return p;
}
__device__ __forceinline__ double &mine_at(const stage_ptrs &p, int rel) { return rel < 64 ? p.mine_lo[rel * 128] : p.mine_hi[(rel - 64) * 128]; }
__device__ __forceinline__ double theirs_at(const stage_ptrs &p, int rel) { return rel < 64 ? p.theirs_lo[rel * 128] : p.theirs_hi[(rel - 64) * 128]; }

// ------------------------------------------------------------------------------------------------
// cols kernel.  Forward: work[j1][k2] -> X (L + 1 bins).  Inverse: Y (L + 1 bins) -> work[j1][k2].
//   lanes tid = 128 t + ell: local column ell (S side 0..63, mirror side 64..127), quarter t of the 128-point axis
template<bool INV>
__global__ __launch_bounds__(512, 2) void c5_cols_kernel(const C *__restrict__ in, C *__restrict__ out, const C *__restrict__ twL,
                                                        const C *__restrict__ tw_real) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *plane = lds;
    C *w128 = (C *) (lds + 128 * 128);
    const int tid = threadIdx.x;
    if (tid < 128) w128[tid] = twL[tid * 1024];                               // W_128^m = W_L^{1024 m}
    const long long row = blockIdx.x >> 3;
    const int b = blockIdx.x & 7;
    const int ell = tid & 127, t = tid >> 7;
    const bool col0 = b == 7 && ell == 64;                                   // column 0 replaces the duplicate 512
    const int col = col0 ? 0 : ell < 64 ? 64 * b + 1 + ell : 960 - 64 * b + (ell - 64);
    const int ellp = (b == 7 && (ell == 63 || ell == 64)) ? ell : 127 - ell; // local column of the pairing partner
    const C *work = INV ? out + row * kL : in + row * kL;
    const C *bins = INV ? in + row * (kL + 1LL) : out + row * (kL + 1LL);
    const __amdgpu_buffer_rsrc_t rwork = __builtin_amdgcn_make_buffer_rsrc((void *) work, 0, kL * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbins = __builtin_amdgcn_make_buffer_rsrc((void *) bins, 0, (kL + 1) * 16, 0x00020000);
    const int woff = (t * 1024 + col) * 16;                                  // work[(4 i + t)][col]: + i * 65536 B
    const int boff = col * 16;                                               // bin 1024 k1 + col:    + k1 * 16384 B

    // W_2L^k for this thread's bins k = 1024 k1 + col, k1 = t + 4 i' + 32 k3: W_2L^{col} W_256^{t} times the constant W_64^{i' + 8 k3}
    const C wt0 = cmul(tw_real[col], tw_real[1024 * t]);

    C u[32], v[32];
    if constexpr (!INV) {
#pragma unroll
        for (int i = 0; i < 32; ++i) u[i] = buf_load<kStream>(rwork, woff, i * 65536, 0.0);
        __syncthreads();
        dft_n<double, false, 32>(u);                                          // over i -> k' in u[brev(k')]
#pragma unroll
        for (int k = 1; k < 32; ++k) u[brev(k, 5)] = cmul(u[brev(k, 5)], w128[t * k]);
        // exchange: plane[k'][t][ell]; thread (ell, t) then holds k' = t + 4 i', all four quarters t'
        double *wr = plane + t * 128 + ell;
        const double *rd = plane + t * 512 + ell;
#pragma unroll
        for (int k = 0; k < 32; ++k) wr[k * 512] = u[brev(k, 5)].x;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) v[i * 4 + tp].x = rd[i * 2048 + tp * 128];
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 32; ++k) wr[k * 512] = u[brev(k, 5)].y;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) v[i * 4 + tp].y = rd[i * 2048 + tp * 128];
        lds_barrier();
        dft_columns<double, false, 4>(v, std::make_integer_sequence<int, 8>{});  // v[4 i' + p] = Z[k1 = t + 4 i' + 32 brev2(p)][col]

        // ---- packed-real pass: a = Z[k1][col] (own), b = Z[k1p][colp] (partner, through the staging plane)
        // partner of k1 is 127 - k1 (128 - k1 in column 0): see stage_ptrs
        double bx[32];
        const stage_ptrs sp = make_stage(plane, t, ell, ellp, col0);
#pragma unroll
        for (int e = 0; e < 32; ++e) mine_at(sp, 4 * (e >> 2) + 32 * brev(e & 3, 2)) = v[e].x;
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) bx[e] = theirs_at(sp, 124 - 4 * (e >> 2) - 32 * brev(e & 3, 2));
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) mine_at(sp, 4 * (e >> 2) + 32 * brev(e & 3, 2)) = v[e].y;
        lds_barrier();
        C wt = wt0;
        asm volatile("" : "+v"(wt.x), "+v"(wt.y));          // the 32 twiddles derived from it must not be computed (and kept) earlier
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int ip = e >> 2, k3 = brev(e & 3, 2);
            const int k1 = t + 4 * ip + 32 * k3;
            const double by = theirs_at(sp, 124 - 4 * ip - 32 * k3);
            const C w = cmul(wt, C{root64_re(ip + 8 * k3), root64_im(ip + 8 * k3)});
            const double wqx = 0.5 * w.y, wqy = -0.5 * w.x;                  // -(i/2) W_2L^k
            const double ax = v[e].x, ay = v[e].y;
            const double sx = ax + bx[e], sy = ay - by, dx = ax - bx[e], dy = ay + by;
            C xk = C{0.5 * sx + (dx * wqx - dy * wqy), 0.5 * sy + (dx * wqy + dy * wqx)};
            if (e == 0 && col0 && t == 0) {                                   // k = 0: X[0], X[L] real (dsc_fft.h:221-225); its "partner" read is unused
                xk = C{ax + ay, 0.0};
                buf_store<kStream>(C{ax - ay, 0.0}, rbins, kL * 16, 0);
            }
            buf_store<kStream>(xk, rbins, boff, k1 * 16384);
        }
    } else {
        // ---- load the bins in the layout the forward kernel leaves them in, pre-pass (dsc_fft.h:194-228)
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int k1 = t + 4 * (e >> 2) + 32 * (e & 3);                   // natural k3 order: v[4 i' + k3]
            v[e] = buf_load<kStream>(rbins, boff, k1 * 16384, 0.0);
        }
        C ylast = C{0.0, 0.0};
        if (col0 && t == 0) { ylast = buf_load<kStream>(rbins, kL * 16, 0, 0.0); v[0].y = 0.0; }     // real parts only at k = 0 and k = L
        __syncthreads();
        double bx[32];
        const stage_ptrs sp = make_stage(plane, t, ell, ellp, col0);
#pragma unroll
        for (int e = 0; e < 32; ++e) mine_at(sp, 4 * (e >> 2) + 32 * (e & 3)) = v[e].x;
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) bx[e] = theirs_at(sp, 124 - 4 * (e >> 2) - 32 * (e & 3));
        if (col0 && t == 0) bx[0] = ylast.x;                                  // bin 0 pairs with bin L
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 32; ++e) mine_at(sp, 4 * (e >> 2) + 32 * (e & 3)) = v[e].y;
        lds_barrier();
        C wt = wt0;
        asm volatile("" : "+v"(wt.x), "+v"(wt.y));          // the 32 twiddles derived from it must not be computed (and kept) earlier
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int ip = e >> 2, k3 = e & 3;
            double by = theirs_at(sp, 124 - 4 * ip - 32 * k3);
            if (e == 0 && col0 && t == 0) by = 0.0;
            const C w = cmul(wt, C{root64_re(ip + 8 * k3), root64_im(ip + 8 * k3)});
            const double wqx = 0.5 * w.y, wqy = 0.5 * w.x;                   // (i/2) conj(W_2L^k)
            const double ax = v[e].x, ay = v[e].y;
            const double sx = ax + bx[e], sy = ay - by, dx = ax - bx[e], dy = ay + by;
            v[e] = C{0.5 * sx + (dx * wqx - dy * wqy), 0.5 * sy + (dx * wqy + dy * wqx)};
            if ((e & 7) == 7) asm volatile("" ::: "memory");                   // partner reads at most 8 deep: registers
        }
        lds_barrier();
        // ---- inverse 128-point transform over k1 = k' + 32 k3: four-point over k3 -> t', twiddle, exchange, 32-point over k'
        dft_columns<double, true, 4>(v, std::make_integer_sequence<int, 8>{});   // v[4 i' + p]: t' = brev2(p)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tp = 1; tp < 4; ++tp) v[i * 4 + brev(tp, 2)] = cmulc(v[i * 4 + brev(tp, 2)], w128[tp * (t + 4 * i)]);
        double *wr = plane + t * 512 + ell;                                   // plane[k' = t + 4 i'][t'][ell]
        const double *rd = plane + t * 128 + ell;                             // thread (ell, t) reads all k' of its quarter t
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) wr[i * 2048 + tp * 128] = v[i * 4 + brev(tp, 2)].x;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 32; ++k) u[k].x = rd[k * 512];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) wr[i * 2048 + tp * 128] = v[i * 4 + brev(tp, 2)].y;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 32; ++k) u[k].y = rd[k * 512];
        dft_n<double, true, 32>(u);                                           // over k' -> i in u[brev(i)]
#pragma unroll
        for (int i = 0; i < 32; ++i) buf_store<kStream>(u[brev(i, 5)], rwork, woff, i * 65536);
    }
}

void set_attrs() {
    static bool done = false;
    if (done) return;
    (void) hipFuncSetAttribute((const void *) c5_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kRowsLds);
    (void) hipFuncSetAttribute((const void *) c5_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kRowsLds);
    (void) hipFuncSetAttribute((const void *) c5_cols_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kColsLds);
    (void) hipFuncSetAttribute((const void *) c5_cols_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kColsLds);
    done = true;
}

}  // namespace

// x: [rows][262144] f64 -> X: [rows][131073] c64.  work: rows * 2 MiB of scratch.  tw_full: W_L^k, k < L; tw_real: W_{2L}^k, k <= L.
void dsc_launch_rfft256k_f64_2pass(const double *x, void *X, long long rows, void *work, const void *tw_full, const void *tw_real,
                                   hipStream_t stream) {
    if (rows <= 0) return;
    set_attrs();
    const dim3 grid((unsigned) (rows * 8));
    hipLaunchKernelGGL(c5_rows_kernel<false>, grid, dim3(512), kRowsLds, stream, (const C *) x, (C *) work, (const C *) tw_full, 1.0);
    hipLaunchKernelGGL(c5_cols_kernel<false>, grid, dim3(512), kColsLds, stream, (const C *) work, (C *) X, (const C *) tw_full,
                       (const C *) tw_real);
}

// X: [rows][131073] c64 -> x: [rows][262144] f64
void dsc_launch_irfft256k_f64_2pass(const void *X, double *x, long long rows, void *work, const void *tw_full, const void *tw_real,
                                    hipStream_t stream) {
    if (rows <= 0) return;
    set_attrs();
    const dim3 grid((unsigned) (rows * 8));
    hipLaunchKernelGGL(c5_cols_kernel<true>, grid, dim3(512), kColsLds, stream, (const C *) X, (C *) work, (const C *) tw_full,
                       (const C *) tw_real);
    hipLaunchKernelGGL(c5_rows_kernel<true>, grid, dim3(512), kRowsLds, stream, (const C *) work, (C *) x, (const C *) tw_full,
                       1.0 / (double) kL);                                                     // 2/(2n), dsc_fft.h:232
}
