// fft_regs_mid.hip — register-resident transforms of contiguous lines with complex length
// L = 1024 B, B in {2, 4, 8, 16}  (real lengths 4096 .. 32768, complex 2048 .. 16384) and, for
// complex data, B = 32 (L = 32768), f32.
// The same kernels in f64 for L = 256 .. 16384.
//
// The design of fft_r2c_64k.hip / fft_r2c_256k_f64.hip, parameterised by B: a line lives in
// the registers of T = 32 B threads, 32 complex each; G = 16/B (or 8/B) lines share a
// workgroup.  One HBM round trip:
//
//   j = T j1 + B j2 + j3   (j1, j2 < 32, j3 < B),   k = k1 + 32 k2 + 1024 k3
//   thread t = B j2 + j3 loads z[T j1 + t]                 (32 coalesced loads in flight)
//   pass 1  dft32 over j1,  x W_1024^{j2 k1}               exchange 1 (LDS, re then im plane)
//   pass 2  dft32 over j2,  x W_L^{j3 k1} W_{32B}^{j3 k2}  exchange 2
//   pass 3  32/B dft_B over j3; thread t now holds columns k' = t + T i
//
// dsc_rfft: the packed-real pass (dsc_fft.h:199-225) pairs bins k and L-k, which live in
// different threads; the line is staged through LDS one component at a time (L floats, the
// plane's own space) and read back as pairs, so both output streams leave coalesced.
// dsc_irfft: the same staging on the way in (dsc_fft.h:194-228).
//
// Reference: exec_fft / exec_rfft (dsc/src/dsc.cpp:1958-2007, 2102-2171) over
// dsc_complex_fft / dsc_real_fft (dsc/include/dsc_fft.h:57-238).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <utility>

namespace {

template<typename R> struct alignas(2 * sizeof(R)) cpx { R x, y; };

template<typename R> __device__ __forceinline__ cpx<R> operator+(cpx<R> a, cpx<R> b) { return cpx<R>{a.x + b.x, a.y + b.y}; }
template<typename R> __device__ __forceinline__ cpx<R> operator-(cpx<R> a, cpx<R> b) { return cpx<R>{a.x - b.x, a.y - b.y}; }
template<typename R> __device__ __forceinline__ cpx<R> cmul(cpx<R> a, cpx<R> w) { return cpx<R>{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
template<typename R> __device__ __forceinline__ cpx<R> cmulc(cpx<R> a, cpx<R> w) { return cpx<R>{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }

__host__ __device__ constexpr int brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
__host__ __device__ constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

// cos(2 pi q / 64), q = 0..16
__device__ constexpr double kCos64[17] = {
    1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
    0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081,
    0.70710678118654752440, 0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856,
    0.38268343236508977173, 0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199,
    0.0};
__device__ constexpr double root64_re(int q) {
    q &= 63;
    return q <= 16 ? kCos64[q] : q <= 32 ? -kCos64[32 - q] : q <= 48 ? -kCos64[q - 32] : kCos64[64 - q];
}
__device__ constexpr double root64_im(int q) {      // -sin(2 pi q / 64)
    q &= 63;
    return q <= 16 ? -kCos64[16 - q] : q <= 32 ? -kCos64[q - 16] : q <= 48 ? kCos64[48 - q] : kCos64[q - 48];
}

// d * W_M^K (forward) or d * conj(W_M^K) (INV), K < M/2, M <= 32
template<typename R, bool INV, int M, int K>
__device__ __forceinline__ cpx<R> mul_root(cpx<R> d) {
    constexpr R c8 = (R) 0.70710678118654752440;
    if constexpr (K == 0) {
        return d;
    } else if constexpr (4 * K == M) {
        return INV ? cpx<R>{-d.y, d.x} : cpx<R>{d.y, -d.x};
    } else if constexpr (8 * K == M) {
        return INV ? cpx<R>{(d.x - d.y) * c8, (d.x + d.y) * c8} : cpx<R>{(d.x + d.y) * c8, (d.y - d.x) * c8};
    } else if constexpr (8 * K == 3 * M) {
        return INV ? cpx<R>{-(d.x + d.y) * c8, (d.x - d.y) * c8} : cpx<R>{(d.y - d.x) * c8, -(d.x + d.y) * c8};
    } else {
        constexpr R wr = (R) root64_re(K * (64 / M));
        constexpr R wi = (R) (INV ? -root64_im(K * (64 / M)) : root64_im(K * (64 / M)));
        return cpx<R>{d.x * wr - d.y * wi, d.x * wi + d.y * wr};
    }
}

template<typename R, bool INV, int M, int G, int... K>
__device__ __forceinline__ void dif_group(cpx<R> (&v)[32], std::integer_sequence<int, K...>) {
    (([&] {
         const cpx<R> u = v[G + K] + v[G + K + M / 2];
         const cpx<R> d = v[G + K] - v[G + K + M / 2];
         v[G + K] = u;
         v[G + K + M / 2] = mul_root<R, INV, M, K>(d);
     }()),
     ...);
}
template<typename R, bool INV, int M, int BASE, int... G>
__device__ __forceinline__ void dif_stage(cpx<R> (&v)[32], std::integer_sequence<int, G...>) {
    (dif_group<R, INV, M, BASE + G * M>(v, std::make_integer_sequence<int, M / 2>{}), ...);
}
// N-point DFT (N = 2 .. 32) of v[BASE .. BASE+N), natural order in; v[BASE + p] returns bin brev(p, log2 N)
template<typename R, bool INV, int N, int BASE = 0>
__device__ __forceinline__ void dft_n(cpx<R> (&v)[32]) {
    if constexpr (N >= 32) dif_stage<R, INV, 32, BASE>(v, std::make_integer_sequence<int, N / 32>{});
    if constexpr (N >= 16) dif_stage<R, INV, 16, BASE>(v, std::make_integer_sequence<int, N / 16>{});
    if constexpr (N >= 8)  dif_stage<R, INV, 8, BASE>(v, std::make_integer_sequence<int, N / 8>{});
    if constexpr (N >= 4)  dif_stage<R, INV, 4, BASE>(v, std::make_integer_sequence<int, N / 4>{});
    dif_stage<R, INV, 2, BASE>(v, std::make_integer_sequence<int, N / 2>{});
}
template<typename R, bool INV, int N, int... I>
__device__ __forceinline__ void dft_columns(cpx<R> (&v)[32], std::integer_sequence<int, I...>) {
    (dft_n<R, INV, N, I * N>(v), ...);
}

// LDS-only barrier: does not wait for outstanding global loads / stores
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// TWO = false: L = 1024 B, three passes (32 x 32 x B), T = 32 B threads per line.
// TWO = true:  L = 32 B,   two passes  (32 x B),       T = B threads per line (a wave holds 64 / B lines).
template<typename R, int B, bool TWO> struct mid_cfg {
    static constexpr bool DP = sizeof(R) == 8;
    static constexpr int T = TWO ? B : 32 * B;       // threads per line
    static constexpr int L = 32 * T;                 // complex length
    static constexpr int COLS = TWO ? 32 : 1024;     // columns entering the last pass (DFT_B down each)
    // threads per workgroup.  f32: a thread needs ~100 VGPRs and 132-264 B of LDS, so two 512-thread groups (or three
    // of 256) share a CU and overlap each other's load / compute / store phases.  f64: twice the registers (one
    // 512-thread group or several smaller ones per CU) and twice the LDS, which decides the group size.
    static constexpr int NT = DP ? (TWO ? (B >= 32 ? 128 : 256) : (B >= 8 ? 512 : 128))
                                 : (TWO ? 256 : B >= 32 ? 1024 : B >= 8 ? 512 : 256);
    static constexpr int G = NT / T;                 // lines per workgroup
    static constexpr int WAVES_PER_EU = DP ? 2 : (TWO ? 2 : B >= 8 ? 4 : 2);   // f32: <= 128 VGPRs where two 512-thread groups share a CU
    static constexpr int P1 = 33;                    // exchange-1 row pitch (values): odd
    static constexpr int P2 = B + 1;                 // last-exchange row pitch
    static constexpr int SP = L + 1;                 // staging pitch per line (bins 0 .. L)
    static constexpr int PLANE = G * COLS * P2;      // >= G*T*P1 (three-pass) and >= G*SP
    static constexpr int CPT = 32 / B;               // columns per thread in the last pass
    static constexpr int TABLE = TWO ? L : 1024;     // LDS twiddle table: W_L^m (two-pass) or W_1024^m
    static constexpr int TABLE_STRIDE = TWO ? 1 : B;
};

template<typename R, int B, bool TWO>
constexpr size_t mid_lds_bytes() { return ((size_t) mid_cfg<R, B, TWO>::PLANE + 2 * mid_cfg<R, B, TWO>::TABLE) * sizeof(R); }

typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Buffer accesses: voff (VGPR, bytes) + soff (SGPR / literal, bytes) against a wave-uniform descriptor.
// Out-of-range lanes read zero and their stores are dropped, which is how a partially filled last
// group and the "no such line" waves are handled (num_records = 0).
__device__ __forceinline__ cpx<float> buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff, float) {
    const f2 q = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    return cpx<float>{q.x, q.y};
}
__device__ __forceinline__ cpx<double> buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff, double) {
    const d2 q = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return cpx<double>{q.x, q.y};
}
// real sample widened to complex (dsc_fft on real input casts first: dsc.cpp:1984-1988)
__device__ __forceinline__ cpx<float> buf_load_real(__amdgpu_buffer_rsrc_t r, int voff, int soff, float) {
    return cpx<float>{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)), 0.0f};
}
__device__ __forceinline__ cpx<double> buf_load_real(__amdgpu_buffer_rsrc_t r, int voff, int soff, double) {
    return cpx<double>{__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)), 0.0};
}
__device__ __forceinline__ void buf_store(cpx<float> a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, f2{a.x, a.y}), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store(cpx<double> a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d2{a.x, a.y}), r, voff, soff, 0);
}

// MODE: DSC_MODE_C2C, DSC_MODE_R2C_CAST (L reals in), DSC_MODE_R2C_PACKED (forward only), DSC_MODE_C2R_PACKED (inverse only)
template<typename R, int B, bool TWO, int MODE, bool INV>
__global__ __launch_bounds__((mid_cfg<R, B, TWO>::NT), (mid_cfg<R, B, TWO>::WAVES_PER_EU)) void fft_mid_kernel(
    const cpx<R> *__restrict__ in, cpx<R> *__restrict__ out, long long n_lines, const cpx<R> *__restrict__ tw_full,
    const cpx<R> *__restrict__ tw_real, R scale) {
    using C = cpx<R>;
    using cfg = mid_cfg<R, B, TWO>;
    constexpr int T = cfg::T, L = cfg::L, G = cfg::G, NT = cfg::NT, P1 = cfg::P1, P2 = cfg::P2, SP = cfg::SP, CPT = cfg::CPT;
    constexpr int COLS = cfg::COLS;
    constexpr int LOGB = ilog2(B);
    constexpr int CB = (int) sizeof(C);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *wtab = (C *) (plane + cfg::PLANE);

    const int tid = threadIdx.x;
    const int g = T >= 64 ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;      // line within the group
    const int t = tid - g * T;
    const long long line0 = (long long) blockIdx.x * G;
    const long long left = n_lines - line0;
    const int n_valid = left < G ? (int) left : G;                  // lines past the end read zeros, their stores are dropped
    const int hi = TWO ? 0 : t / B, lo = TWO ? t : t % B;
    constexpr int in_pitch = MODE == DSC_MODE_C2R_PACKED ? L + 1 : L;
    constexpr int out_pitch = MODE == DSC_MODE_R2C_PACKED ? L + 1 : L;
    constexpr int IB = MODE == DSC_MODE_R2C_CAST ? (int) sizeof(R) : CB;      // bytes per input element
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) in + line0 * in_pitch * IB), 0, n_valid * in_pitch * IB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (out + line0 * out_pitch), 0, n_valid * out_pitch * CB, 0x00020000);
    const int vin = (g * in_pitch + t) * IB;                       // byte offset of element t of this thread's line
    const int vout = (g * out_pitch + t) * CB;
    R *stage = plane + g * SP;

    for (int i = tid; i < cfg::TABLE; i += NT) wtab[i] = tw_full[(long long) i * cfg::TABLE_STRIDE];    // W_1024^m = W_L^{B m}

    C v[32];
#pragma unroll
    for (int j1 = 0; j1 < 32; ++j1) {                                                        // z[T j1 + t]
        if constexpr (MODE == DSC_MODE_R2C_CAST) v[j1] = buf_load_real(rin, vin, j1 * T * IB, R{});
        else                                     v[j1] = buf_load(rin, vin, j1 * T * CB, R{});
    }

    if constexpr (MODE == DSC_MODE_C2R_PACKED) {
        // Z[k] = (a + conj b)/2 + wq (a - conj b), a = Y[k], b = Y[L-k], wq = (i/2) conj(W_2L^k), for the
        // thread's own k = T j1 + t; b comes through the staging plane, one component at a time.
        const C wbase = tw_real[t];
        C yl = C{(R) 0, (R) 0};
        if (t == 0) { yl = buf_load(rin, vin, L * CB, R{}); v[0].y = (R) 0; yl.y = (R) 0; }   // dsc_fft.h:227-228: real parts only at k = 0
        R dx[32];
        R *up = stage + t;                          // up[T j1]         = stage[k]
        const R *dn = stage + (L - 31 * T) - t;     // dn[T (31 - j1)]  = stage[L - k]
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) up[T * j1] = v[j1].x;
        if (t == 0) stage[L] = yl.x;
        lds_barrier();
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) {
            const R bx = dn[T * (31 - j1)];
            dx[j1] = v[j1].x - bx;
            v[j1].x = v[j1].x + bx;
        }
        lds_barrier();
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) up[T * j1] = v[j1].y;
        if (t == 0) stage[L] = yl.y;
        lds_barrier();
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) {
            const R by = dn[T * (31 - j1)];
            const C w = cmul(wbase, C{(R) root64_re(j1), (R) root64_im(j1)});      // W_2L^{t + T j1} = W_2L^t W_64^{j1}
            const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;
            const R sy = v[j1].y - by, dy = v[j1].y + by;
            const R zx = (R) 0.5 * v[j1].x + (dx[j1] * wqx - dy * wqy);
            const R zy = (R) 0.5 * sy + (dx[j1] * wqy + dy * wqx);
            v[j1] = C{zx, zy};
        }
    }
    __syncthreads();                // twiddle table visible; staging reads done before the plane is reused

    C u[32];
    if constexpr (!TWO) {
        // ---- pass 1 over j1, twiddle W_1024^{j2 k1}
        dft_n<R, INV, 32>(v);
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            const C w = wtab[hi * k1];
            v[brev(k1, 5)] = INV ? cmulc(v[brev(k1, 5)], w) : cmul(v[brev(k1, 5)], w);
        }
        // ---- exchange 1: (j2, j3)[k1] -> thread B k1 + j3, [j2]
        R *wr = plane + (g * T + lo) * P1 + hi;
        const R *rd = plane + tid * P1;
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) wr[k1 * B * P1] = v[brev(k1, 5)].x;
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m].x = rd[m];
        lds_barrier();
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) wr[k1 * B * P1] = v[brev(k1, 5)].y;
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m].y = rd[m];
        lds_barrier();
    } else {
#pragma unroll
        for (int m = 0; m < 32; ++m) u[m] = v[m];                   // two-pass: the loaded index j1 is the pass-2 index
    }
    // ---- pass 2 over j2 (thread = (k1, j3) = (hi, lo)), twiddle W_L^{j3 k1} W_{32B}^{j3 k2}
    dft_n<R, INV, 32>(u);
    {
        if constexpr (!TWO) {
            const C tw2_base = tw_full[hi * lo];
            u[0] = INV ? cmulc(u[0], tw2_base) : cmul(u[0], tw2_base);
#pragma unroll
            for (int k2 = 1; k2 < 32; ++k2) {
                const C w = cmul(tw2_base, wtab[(32 / B) * lo * k2]);
                u[brev(k2, 5)] = INV ? cmulc(u[brev(k2, 5)], w) : cmul(u[brev(k2, 5)], w);
            }
        } else {
#pragma unroll
            for (int k2 = 1; k2 < 32; ++k2) {
                const C w = wtab[lo * k2];                           // W_L^{j3 k2}
                u[brev(k2, 5)] = INV ? cmulc(u[brev(k2, 5)], w) : cmul(u[brev(k2, 5)], w);
            }
        }
        // ---- last exchange: row = column k' = k1 + 32 k2 (two-pass: k2), col = j3; thread t reads columns t + T i
        constexpr int CS = TWO ? 1 : 32;
        R *wr = plane + (g * COLS + hi) * P2 + lo;
        const R *rd = plane + (g * COLS + t) * P2;
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) wr[k2 * CS * P2] = u[brev(k2, 5)].x;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int m = 0; m < B; ++m) v[i * B + m].x = rd[i * T * P2 + m];
        lds_barrier();
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) wr[k2 * CS * P2] = u[brev(k2, 5)].y;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int m = 0; m < B; ++m) v[i * B + m].y = rd[i * T * P2 + m];
        lds_barrier();
    }
    // ---- last pass over j3: CPT DFTs of B points; v[i B + p] = bin k = (t + T i) + COLS brev(p)
    dft_columns<R, INV, B>(v, std::make_integer_sequence<int, CPT>{});

    if constexpr (MODE != DSC_MODE_R2C_PACKED) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int p = 0; p < B; ++p) {
                const C r = v[i * B + p];
                buf_store(C{r.x * scale, r.y * scale}, rout, vout, (T * i + COLS * brev(p, LOGB)) * CB);
            }
    } else {
        // packed-real post-pass (dsc_fft.h:199-225), one thread per PAIR (k, L-k), k = t + T i < L/2,
        // plus k = L/2 (thread 0).  a = Z[k], b = Z[L-k]:
        //   s = a + conj b, d = a - conj b, wq = -(i/2) W_2L^k:  X[k] = s/2 + wq d,  X[L-k] = conj(s/2 - wq d)
        const C wbase = tw_real[t];
        R ax[16], bx[16], amx = (R) 0;
        R *up = stage + t;                          // up[T i]          = stage[k]
        const R *dn = stage + (L - 15 * T) - t;     // dn[T (15 - i)]   = stage[L - k]
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int p = 0; p < B; ++p) up[T * i + COLS * brev(p, LOGB)] = v[i * B + p].x;
        if (t == 0) stage[L] = v[0].x;                                    // Z[L] := Z[0]
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) { ax[i] = up[T * i]; bx[i] = dn[T * (15 - i)]; }
        if (t == 0) amx = stage[L / 2];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int p = 0; p < B; ++p) up[T * i + COLS * brev(p, LOGB)] = v[i * B + p].y;
        if (t == 0) stage[L] = v[0].y;
        lds_barrier();
        const int dn_voff = (g * out_pitch + (L - 15 * T) - t) * CB;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const R ay = up[T * i], by = dn[T * (15 - i)];
            const C w = cmul(wbase, C{(R) root64_re(i), (R) root64_im(i)});        // W_2L^{t + T i} = W_2L^t W_64^i
            const R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;
            const R sx = ax[i] + bx[i], sy = ay - by, dx = ax[i] - bx[i], dy = ay + by;
            const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
            C xk = C{(R) 0.5 * sx + wdx, (R) 0.5 * sy + wdy};
            C xm = C{(R) 0.5 * sx - wdx, wdy - (R) 0.5 * sy};
            if (i == 0 && t == 0) { xk.y = (R) 0; xm.y = (R) 0; }           // dsc_fft.h:221-225 stores exact zeros
            buf_store(C{xk.x * scale, xk.y * scale}, rout, vout, T * i * CB);
            buf_store(C{xm.x * scale, xm.y * scale}, rout, dn_voff, T * (15 - i) * CB);
        }
        if (t == 0) {                                                     // k = L/2: a = b, W_2L^{L/2} = -i
            const R ay = stage[L / 2];
            buf_store(C{amx * scale, -ay * scale}, rout, vout, (L / 2) * CB);
        }
    }
}

template<typename R, int B, bool TWO, int MODE, bool INV>
void launch_one(const void *in, void *out, long long n_lines, const void *tw_full, const void *tw_real, double scale, hipStream_t stream) {
    using cfg = mid_cfg<R, B, TWO>;
    constexpr size_t lds = mid_lds_bytes<R, B, TWO>();
    static bool attr_set = false;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) fft_mid_kernel<R, B, TWO, MODE, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        attr_set = true;
    }
    const long long groups = (n_lines + cfg::G - 1) / cfg::G;
    hipLaunchKernelGGL((fft_mid_kernel<R, B, TWO, MODE, INV>), dim3((unsigned) groups), dim3(cfg::NT), lds, stream, (const cpx<R> *) in,
                       (cpx<R> *) out, n_lines, (const cpx<R> *) tw_full, (const cpx<R> *) tw_real, (R) scale);
}

template<typename R, int B, bool TWO>
void launch_b(const void *in, void *out, long long n_lines, dsc_fft_mode mode, bool inverse, const void *tw_full, const void *tw_real,
              double scale, hipStream_t stream) {
    if (mode == DSC_MODE_R2C_PACKED)      launch_one<R, B, TWO, DSC_MODE_R2C_PACKED, false>(in, out, n_lines, tw_full, tw_real, scale, stream);
    else if (mode == DSC_MODE_C2R_PACKED) launch_one<R, B, TWO, DSC_MODE_C2R_PACKED, true>(in, out, n_lines, tw_full, tw_real, scale, stream);
    else if (mode == DSC_MODE_R2C_CAST && !inverse) launch_one<R, B, TWO, DSC_MODE_R2C_CAST, false>(in, out, n_lines, tw_full, tw_real, scale, stream);
    else if (mode == DSC_MODE_R2C_CAST)   launch_one<R, B, TWO, DSC_MODE_R2C_CAST, true>(in, out, n_lines, tw_full, tw_real, scale, stream);
    else if (inverse)                     launch_one<R, B, TWO, DSC_MODE_C2C, true>(in, out, n_lines, tw_full, tw_real, scale, stream);
    else                                  launch_one<R, B, TWO, DSC_MODE_C2C, false>(in, out, n_lines, tw_full, tw_real, scale, stream);
}

}  // namespace

bool dsc_fft_regs_mid_supports(int L, dsc_fft_mode mode, bool single_precision) {
    if (L == 32768) return single_precision && mode == DSC_MODE_C2C;   // the packed-real 65536-point f32 transforms have their own kernels
    return L == 256 || L == 512 || L == 1024 || L == 2048 || L == 4096 || L == 8192 || L == 16384;
}

template<typename R>
static void launch_len(const void *in, void *out, long long n_lines, int L, dsc_fft_mode mode, bool inverse, const void *tw_full,
                       const void *tw_real, double scale, hipStream_t stream) {
    switch (L) {
        case 256:   launch_b<R, 8, true>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, stream); break;
        case 512:   launch_b<R, 16, true>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, stream); break;
        case 1024:  launch_b<R, 32, true>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, stream); break;
        case 2048:  launch_b<R, 2, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, stream); break;
        case 4096:  launch_b<R, 4, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, stream); break;
        case 8192:  launch_b<R, 8, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, stream); break;
        default:    launch_b<R, 16, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, stream); break;
    }
}

void dsc_launch_fft_regs_mid(const void *in, void *out, long long n_lines, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                             const void *tw_full, const void *tw_real, double scale, hipStream_t stream) {
    if (n_lines <= 0) return;
    if (!single_precision) {
        launch_len<double>(in, out, n_lines, L, mode, inverse, tw_full, tw_real, scale, stream);
    } else if (L == 32768) {
        if (inverse) launch_one<float, 32, false, DSC_MODE_C2C, true>(in, out, n_lines, tw_full, tw_real, scale, stream);
        else         launch_one<float, 32, false, DSC_MODE_C2C, false>(in, out, n_lines, tw_full, tw_real, scale, stream);
    } else {
        launch_len<float>(in, out, n_lines, L, mode, inverse, tw_full, tw_real, scale, stream);
    }
}
