// fft_r2c_64k.hip — 65536-point real FFT (f32) with the whole transform resident in registers.
//
// Replaces, for this size, the reference's per-line loop  gather -> dsc_real_fft -> scatter
// (dsc/src/dsc.cpp:2102-2171, dsc/include/dsc_fft.h:57-103, 178-238) by ONE pass over HBM:
// 4 B/sample in, 8 B/sample out, nothing else touches memory.
//
// Why registers: the packed transform is M = 32768 complex = 256 KiB per row, more than the
// 160 KiB of LDS but half of a CU's 512 KiB vector register file.  One 1024-thread workgroup
// (16 waves, 4 per SIMD, <= 128 VGPRs each) owns one row: 32 complex per thread.
//
//   M = 32 x 32 x 32,   j = 1024 j1 + 32 j2 + j3   (input),   k = k1 + 32 k2 + 1024 k3   (output)
//
//   load    thread t = 32 j2 + j3 reads z[1024 j1 + t], j1 = 0..31       (coalesced 8 B / lane)
//   pass 1  32-point DFT over j1 in registers, times W_1024^{j2 k1}
//   xchg 1  transpose j2 <-> k1 through LDS (re plane, then im plane: 136 KiB each)
//   pass 2  32-point DFT over j2, times W_32768^{j3 (k1 + 32 k2)}
//   xchg 2  transpose j3 <-> k2 through LDS; the reader picks the column k' = k1 + 32 k2 so
//           that lane l and lane 63-l of a wave hold columns k' and 1024-k'
//   pass 3  32-point DFT over j3: Z[k' + 1024 k3]
//   post    packed-real untangling (dsc_fft.h:199-225) needs Z[k] and Z[M-k]: the partner
//           lane holds it, fetched with ds_bpermute (no LDS storage); each lane finishes 16
//           bin pairs and stores X[k] and X[M-k]                            (8 B / lane)
//
// In-register DFTs are radix-2 DIF with compile-time twiddles (output index bit-reversed in
// the register number, which is free: every register index is a constant).  Inter-pass
// twiddles come from an 8 KiB LDS table (W_1024) and two per-thread constants.
// The inverse kernel (dsc_irfft, dsc_fft.h:194-236) is the same pipeline run backwards.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <utility>

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));      // memory-op type only (8-B loads / stores / LDS)

// Arithmetic is done on plain scalar pairs, and this file is built with -fno-slp-vectorize:
// v_pk_*_f32 issues at half the rate of the scalar forms on gfx950 (measured, tools/valubench.hip:
// ~6 vs ~3 cycles per wave-instruction), cannot take literal constants (every twiddle constant
// would occupy a VGPR pair for the whole kernel) and needs v_mov shuffles to line operands up.
struct cf { float x, y; };
__device__ __forceinline__ cf operator+(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf to_cf(f2 a) { return cf{a.x, a.y}; }
__device__ __forceinline__ f2 to_f2(cf a) { return f2{a.x, a.y}; }

constexpr int kM = 32768;            // complex points per row
constexpr int kRowPitch = 34;        // floats per LDS row: 32 + 2 -> conflict-free b64 row reads
constexpr int kPlaneFloats = 1024 * kRowPitch;
constexpr int kLdsBytes = kPlaneFloats * 4 + 1024 * 8;      // exchange plane + W_1024 table
constexpr int kTabEntries = 1024;

// aux table layout (f2 entries): [0,1024) W_1024^m | [1024,2048) W_32768^m | [2048,3072) W_65536^m
constexpr int kAuxW1024 = 0, kAuxW32768 = 1024, kAuxW65536 = 2048, kAuxEntries = 3072;

__host__ __device__ constexpr int br5(int x) {
    return ((x & 1) << 4) | ((x & 2) << 2) | (x & 4) | ((x & 8) >> 2) | ((x & 16) >> 4);
}

// cos / sin of 2 pi q / 64, q = 0..16 (first quadrant; the rest by symmetry)
__device__ constexpr float kCos64[17] = {
    1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f,
    0.92387953251128675613f, 0.88192126434835502971f, 0.83146961230254523708f, 0.77301045336273696081f,
    0.70710678118654752440f, 0.63439328416364549822f, 0.55557023301960222474f, 0.47139673682599764856f,
    0.38268343236508977173f, 0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f,
    0.0f};

// exp(-2 pi i q / 64) for q in [0, 64): (cos, -sin)
__device__ constexpr float root64_re(int q) {
    q &= 63;
    return q <= 16 ? kCos64[q] : q <= 32 ? -kCos64[32 - q] : q <= 48 ? -kCos64[q - 32] : kCos64[64 - q];
}
__device__ constexpr float root64_im(int q) {      // -sin(2 pi q / 64)
    q &= 63;
    return q <= 16 ? -kCos64[16 - q] : q <= 32 ? -kCos64[q - 16] : q <= 48 ? kCos64[48 - q] : kCos64[q - 48];
}

__device__ __forceinline__ cf cmul(cf a, cf w) {
    return cf{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
}
__device__ __forceinline__ cf cmul_conj(cf a, cf w) {     // a * conj(w)
    return cf{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y};
}

// d * W_M^K (forward) or d * conj(W_M^K) (INV), K < M/2, constants folded at compile time
template<bool INV, int M, int K>
__device__ __forceinline__ cf mul_root(cf d) {
    constexpr float c8 = 0.70710678118654752440f;
    if constexpr (K == 0) {
        return d;
    } else if constexpr (4 * K == M) {
        return INV ? cf{-d.y, d.x} : cf{d.y, -d.x};
    } else if constexpr (8 * K == M) {
        return INV ? cf{(d.x - d.y) * c8, (d.x + d.y) * c8} : cf{(d.x + d.y) * c8, (d.y - d.x) * c8};
    } else if constexpr (8 * K == 3 * M) {
        return INV ? cf{-(d.x + d.y) * c8, (d.x - d.y) * c8} : cf{(d.y - d.x) * c8, -(d.x + d.y) * c8};
    } else {
        constexpr float wr = root64_re(K * (64 / M));
        constexpr float wi = INV ? -root64_im(K * (64 / M)) : root64_im(K * (64 / M));
        return cf{d.x * wr - d.y * wi, d.x * wi + d.y * wr};
    }
}

#ifdef DSC_DFT32_DIF
template<bool INV, int M, int G, int... K>
__device__ __forceinline__ void dif_group(cf (&v)[32], std::integer_sequence<int, K...>) {
    (([&] {
         const cf u = v[G + K] + v[G + K + M / 2];
         const cf d = v[G + K] - v[G + K + M / 2];
         v[G + K] = u;
         v[G + K + M / 2] = mul_root<INV, M, K>(d);
     }()),
     ...);
}

template<bool INV, int M, int... G>
__device__ __forceinline__ void dif_stage(cf (&v)[32], std::integer_sequence<int, G...>) {
    (dif_group<INV, M, G * M>(v, std::make_integer_sequence<int, M / 2>{}), ...);
}

// 32-point DFT, natural order in; v[p] returns bin br5(p).
template<bool INV>
__device__ __forceinline__ void dft32(cf (&v)[32]) {
    dif_stage<INV, 32>(v, std::make_integer_sequence<int, 1>{});
    dif_stage<INV, 16>(v, std::make_integer_sequence<int, 2>{});
    dif_stage<INV, 8>(v, std::make_integer_sequence<int, 4>{});
    dif_stage<INV, 4>(v, std::make_integer_sequence<int, 8>{});
    dif_stage<INV, 2>(v, std::make_integer_sequence<int, 16>{});
}


#else
// The same transform as the decimation-in-time graph with natural-order input and bit-reversed output: stage s (half distance
// h = 32 >> s) has 2^(s-1) groups, group g multiplies the LOWER input of its butterflies by ONE constant e^{-i pi theta_g},
// theta_g = bitrev(g) / 2^(s-1) (a DFT with frequency offset theta splits into offsets theta / 2 and (1 + theta) / 2), then
// adds and subtracts.  A constant twiddle in front of the add / sub costs 6 fused multiply-adds instead of the 8 operations of
// "subtract, then multiply" (Linzer & Feig): with w = c (1 + i t), p = b.x - t b.y, q = b.y + t b.x, out = a +- c (p, q); the
// larger of |cos|, |sin| is factored out so that |t| <= 1.  34 of the 80 butterflies carry such a twiddle: 388 VALU
// instructions per 32-point DFT instead of 456 — the fused filter, six of them per row, is bound by the vector ALUs.
constexpr double kCos64d[17] = {
    1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
    0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081,
    0.70710678118654752440, 0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856,
    0.38268343236508977173, 0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199,
    0.0};

// a, b -> a + w b, a - w b with w = W_32^E (forward) or its conjugate (INV), 0 <= E < 16
template<bool INV, int E>
__device__ __forceinline__ void dit_butterfly(cf &a, cf &b) {
    if constexpr (E == 0) {
        const cf u = a + b, d = a - b;
        a = u; b = d;
    } else if constexpr (E == 8) {                          // w = -i (forward), +i (inverse)
        const cf u = INV ? cf{a.x - b.y, a.y + b.x} : cf{a.x + b.y, a.y - b.x};
        const cf d = INV ? cf{a.x + b.y, a.y - b.x} : cf{a.x - b.y, a.y + b.x};
        a = u; b = d;
    } else {
        constexpr double c = E <= 8 ? kCos64d[2 * E] : -kCos64d[32 - 2 * E];            // cos(2 pi E / 32)
        constexpr double sn = (E <= 8 ? kCos64d[16 - 2 * E] : kCos64d[2 * E - 16]);       // sin(2 pi E / 32) > 0
        constexpr double s = INV ? sn : -sn;                                            // w = c + i s
        if constexpr ((c < 0 ? -c : c) >= sn) {
            constexpr float t = (float) (s / c), cc = (float) c;
            float p, q;
            if constexpr (t == 1.0f) { p = b.x - b.y; q = b.y + b.x; }
            else if constexpr (t == -1.0f) { p = b.x + b.y; q = b.y - b.x; }
            else { p = __builtin_fmaf(-t, b.y, b.x); q = __builtin_fmaf(t, b.x, b.y); }
            const cf u = cf{__builtin_fmaf(cc, p, a.x), __builtin_fmaf(cc, q, a.y)};
            const cf d = cf{__builtin_fmaf(-cc, p, a.x), __builtin_fmaf(-cc, q, a.y)};
            a = u; b = d;
        } else {                                            // w b = s (r b.x - b.y, r b.y + b.x), r = c / s
            constexpr float r = (float) (c / s), ss = (float) s;
            const float pn = __builtin_fmaf(-r, b.x, b.y);  // -(r b.x - b.y)
            const float q = __builtin_fmaf(r, b.y, b.x);
            const cf u = cf{__builtin_fmaf(-ss, pn, a.x), __builtin_fmaf(ss, q, a.y)};
            const cf d = cf{__builtin_fmaf(ss, pn, a.x), __builtin_fmaf(-ss, q, a.y)};
            a = u; b = d;
        }
    }
}

constexpr int brev_bits(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// stage S = 1..5: H = 32 >> S, group G covers v[2 H G .. 2 H G + 2 H)
template<bool INV, int S, int G, int... J>
__device__ __forceinline__ void dit_group(cf (&v)[32], std::integer_sequence<int, J...>) {
    constexpr int H = 32 >> S;
    constexpr int E = brev_bits(G, S - 1) * (16 >> (S - 1));                            // W_32 exponent = 16 theta_g
    (dit_butterfly<INV, E>(v[2 * H * G + J], v[2 * H * G + J + H]), ...);
}

template<bool INV, int S, int... G>
__device__ __forceinline__ void dit_stage(cf (&v)[32], std::integer_sequence<int, G...>) {
    (dit_group<INV, S, G>(v, std::make_integer_sequence<int, (32 >> S)>{}), ...);
}

// 32-point DFT, natural order in; v[p] returns bin br5(p).
template<bool INV>
__device__ __forceinline__ void dft32(cf (&v)[32]) {
    dit_stage<INV, 1>(v, std::make_integer_sequence<int, 1>{});
    dit_stage<INV, 2>(v, std::make_integer_sequence<int, 2>{});
    dit_stage<INV, 3>(v, std::make_integer_sequence<int, 4>{});
    dit_stage<INV, 4>(v, std::make_integer_sequence<int, 8>{});
    dit_stage<INV, 5>(v, std::make_integer_sequence<int, 16>{});
}
#endif

typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// LDS transpose of one float plane: every thread writes 32 floats at wbase + slot * 1088
// (slot = br5(p): register p of a DIF output holds logical index br5(p)) and reads back its
// own row of 32 as 16 x b64.  Two base registers cover all 32 slots with 16-bit immediates.
template<int COMP>
__device__ __forceinline__ void plane_write(float *plane, int wbase, const cf (&v)[32]) {
    float *lo16 = plane + wbase;
    float *hi16 = lo16 + 16 * (32 * kRowPitch);
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const int slot = br5(p);
        const float val = COMP == 0 ? v[p].x : v[p].y;
        if (slot < 16) lo16[slot * (32 * kRowPitch)] = val;
        else           hi16[(slot - 16) * (32 * kRowPitch)] = val;
    }
}
template<int COMP>
__device__ __forceinline__ void plane_read(const float *plane, int row, cf (&v)[32]) {
    const f2 *r = (const f2 *) (plane + row * kRowPitch);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const f2 t = r[m];
        if (COMP == 0) { v[2 * m].x = t.x; v[2 * m + 1].x = t.y; }
        else           { v[2 * m].y = t.x; v[2 * m + 1].y = t.y; }
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e.
// every wave would wait at each of the ~12 barriers per row for its outstanding global
// stores (and prefetched loads); nothing here communicates through global memory.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float bperm(int byte_addr, float x) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(x)));
}

// Column of the output spectrum a lane owns in pass 3 / post-pass (see file header).
__device__ __forceinline__ int column_of(int wave, int lane) {
    int kp = lane < 32 ? 32 * wave + lane : 1024 - 32 * wave - (63 - lane);
    return kp == 1024 ? 512 : kp;
}

// Diagnostic builds (tools/probe64k.hip) add an `io_on` argument: 0 gives every buffer
// descriptor zero records, i.e. all global accesses are dropped while the instruction stream
// stays the same (compute-only timing).
#ifdef DSC_R2C64K_PROBE
#define PROBE_ARGS , int io_on
#define PROBE_NULL , 1
#define IO_ON io_on
#else
#define PROBE_ARGS
#define PROBE_NULL
#define IO_ON 1
#endif

// Values derived from threadIdx are loop invariant; hipcc hoists every address built from
// them out of the persistent row loop (dozens of VGPRs) and then spills them, and every
// scratch reload drains vmcnt, i.e. waits for all outstanding global stores.  The thread id
// is therefore rebuilt where it is needed from the wave number (an SGPR) and the hardware
// lane count, behind an opaque zero that keeps the two mbcnt from being hoisted.
__device__ __forceinline__ int thread_id(int wave_sgpr) {
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const int lane = __builtin_amdgcn_mbcnt_hi(-1, __builtin_amdgcn_mbcnt_lo(-1, zero));
    return (wave_sgpr << 6) | lane;
}

// Row data is touched exactly once: non-temporal (aux = 2, `nt`) loads and stores keep it from
// displacing the tables and the next rows in L2 (measured +3.8 % on the rfft kernel).  The
// filter spectrum H is re-read by every row and stays on the default policy.
#ifndef DSC_STREAM_AUX
#define DSC_STREAM_AUX 2
#endif
constexpr int kStream = DSC_STREAM_AUX;
__device__ __forceinline__ cf load_c(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kStream)));
}
// How many pairs of the NEXT row irfft64k_kernel requests before its last pass (see there): measured 0 / 4 / 6 / 8 pairs ->
// 0.914 / 0.868 / 0.879 / 0.920 ms.  With pass 3 pinned (DSC_IRFFT_PIN_PASS3) 4 / 6 / 8 pairs need 116 / 121 / 128 VGPRs without spills
// and measure 0.868 / 0.870 / 0.884 ms: what pays is the rebalanced tail (8 + 16 + 9 loads instead of 17 + 16), not more prefetch.
#ifndef DSC_FILTER_EARLY_LOADS
#define DSC_FILTER_EARLY_LOADS 4      // measured 0 / 4 / 12 / 16: 0.669 / 0.662 / 0.664 / 0.673 ms (the kernel is ALU bound; +1 %)
#endif
#ifndef DSC_IRFFT_PIN_PASS3
#define DSC_IRFFT_PIN_PASS3 1
#endif
#ifndef DSC_IRFFT_EARLY_PAIRS
#define DSC_IRFFT_EARLY_PAIRS 4
#endif
// spectrum rows of the inverse kernel (skewed by 8 B x row; see irfft64k_kernel): cache policy chosen by measurement (default
// policy 0.911 ms; sc0 0.911, nt 0.930, sc0 nt 0.929, sc1 0.937, sc0 sc1 0.936, sc1 nt 0.929)
#ifndef DSC_IRFFT_LOAD_AUX
#define DSC_IRFFT_LOAD_AUX 0
#endif
__device__ __forceinline__ cf load_c_spec(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, DSC_IRFFT_LOAD_AUX)));
}
__device__ __forceinline__ cf load_c_cached(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)));
}
template<int POLICY>
__device__ __forceinline__ void store_c(cf a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, to_f2(a)), r, voff, soff, POLICY);
}

// One LDS transpose = re plane, then im plane.  Barriers sit AFTER each read phase (not before
// each write phase): a wave's LDS writes then overlap the tail of its own butterflies and
// the other waves' arithmetic; the last one leaves the plane free for whoever writes next.
__device__ __forceinline__ void exchange(float *plane, int wbase, int row, const cf (&src)[32], cf (&dst)[32]) {
    plane_write<0>(plane, wbase, src);
    lds_barrier();
    plane_read<0>(plane, row, dst);
    lds_barrier();
    plane_write<1>(plane, wbase, src);
    lds_barrier();
    plane_read<1>(plane, row, dst);
    lds_barrier();
}

// Three passes of the 32768-point complex DFT (forward, or INV = conjugate twiddles).
//   in : v[i] = element 1024 i + t of the input sequence, t = thread id, i = 0..31
//        (forward: time samples z[j]; inverse: bins Z[k] — any thread may hold any column
//         `col`, the exchange routes by column: pass `col` = the column this thread holds)
//   out: v[p] = element col_out + 1024 br5(p) of the output sequence, where the caller picks
//        which output column `col_out` this thread receives.
struct no_hook { __device__ __forceinline__ void operator()() const {} };

// `before_pass3` runs between the second exchange and the last 32-point DFT: from there to the end of the row only v[] (64
// VGPRs) is live, which leaves room to request part of the NEXT row that early (irfft64k_kernel does).
template<bool INV, typename Hook = no_hook>
__device__ __forceinline__ void three_passes(cf (&v)[32], float *plane, const f2 *w1024, const f2 *aux, int wave_sgpr,
                                             bool mirrored_in, bool mirrored_out, Hook before_pass3 = Hook{}) {
    // ---- pass 1 (over the slow index) and twiddle W_1024^{hi r1}, hi = col >> 5
    dft32<INV>(v);
    cf u[32];
    {
        const int t = thread_id(wave_sgpr);
        const int col = mirrored_in ? column_of(t >> 6, t & 63) : t;
        const int hi = col >> 5;
#pragma unroll
        for (int r1 = 1; r1 < 32; ++r1) {
            const cf w = to_cf(w1024[hi * r1]);
            v[br5(r1)] = INV ? cmul_conj(v[br5(r1)], w) : cmul(v[br5(r1)], w);
        }
        // exchange 1: (hi, lo)[r1] -> thread (r1, lo), registers [hi];  LDS row = r1*32 + lo, col = hi
        exchange(plane, (col & 31) * kRowPitch + hi, t, v, u);
    }
    // ---- pass 2 (over the middle index) and twiddle W_32768^{lo r1} W_1024^{lo r2}
    dft32<INV>(u);
    {
        const int t = thread_id(wave_sgpr);
        const int hi = t >> 5, lo = t & 31;              // (r1, lo)
        const cf tw2_base = to_cf(aux[kAuxW32768 + hi * lo]);
        u[0] = INV ? cmul_conj(u[0], tw2_base) : cmul(u[0], tw2_base);
#pragma unroll
        for (int r2 = 1; r2 < 32; ++r2) {
            const cf w = cmul(tw2_base, to_cf(w1024[lo * r2]));
            u[br5(r2)] = INV ? cmul_conj(u[br5(r2)], w) : cmul(u[br5(r2)], w);
        }
        // exchange 2: (r1, lo)[r2] -> output column r1 + 32 r2, registers [lo];  LDS row = column
        const int col_out = mirrored_out ? column_of(t >> 6, t & 63) : t;
        exchange(plane, hi * kRowPitch + lo, col_out, u, v);
    }
    before_pass3();
    // ---- pass 3 (over the fast index)
    dft32<INV>(v);
}

// Packed-real relations on one pair of bins (dsc_fft.h:199-228), p = in[k], q = in[M-k]:
//   s = p + conj q,  d = p - conj q,  out[k] = hs s + wq d,  out[M-k] = conj(hs s - wq d)
// forward: hs = 1/2, wq = -(i/2) W^k;  inverse: hs = 1/(2M), wq = (i/2M) conj(W^k).
__device__ __forceinline__ void real_pair(cf p, cf q, cf wq, float hs, cf &out_k, cf &out_mk) {
    const cf s = cf{p.x + q.x, p.y - q.y};
    const cf d = cf{p.x - q.x, p.y + q.y};
    const cf wd = cmul(d, wq);
    out_k = cf{hs * s.x + wd.x, hs * s.y + wd.y};
    out_mk = cf{hs * s.x - wd.x, wd.y - hs * s.y};
}

// Lane layout of the spectrum side (forward post-pass / inverse pre-pass): lane l and lane
// 63-l of a wave hold columns c and 1024-c; rows a of c pair with rows 31-a of 1024-c.
// Column 0 (wave 0, lane 0) pairs with itself shifted by one row, column 512 (wave 0, lane 63)
// with itself; both use themselves as ds_bpermute partner.
__device__ __forceinline__ int partner_byte_addr(int wave, int lane) {
    return ((wave == 0 && (lane == 63 || lane == 0)) ? lane : 63 - lane) * 4;
}

#ifndef DSC_R2C64K_HELPERS_ONLY       // fft_c2c_32k.hip includes this file for the building blocks only
// ------------------------------------------------------------------------------------------
// forward: x [batch][65536] f32  ->  X [batch][32769] c32
// in_pitch: floats between input rows; in_len <= 65536: valid samples per row — the rest of the transform length reads
// as zero (dsc_rfft's zero padding, dsc.cpp:2125-2133) through the range check of the row's buffer descriptor.
__global__ __launch_bounds__(1024) void rfft64k_kernel(const float *__restrict__ x, f2 *__restrict__ X, int batch,
                                                       const f2 *__restrict__ aux, int in_pitch, int in_len PROBE_ARGS) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *plane = lds;
    f2 *w1024 = (f2 *) (lds + kPlaneFloats);
    w1024[threadIdx.x] = aux[kAuxW1024 + threadIdx.x];
    __syncthreads();

    const int wave_sgpr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // The row loop is software pipelined on the loads: row r+1's 32 loads per lane are issued
    // in the tail of row r, as soon as the spectrum has left the registers for the LDS staging
    // area and BEFORE row r's stores are issued, so that (vmcnt retires in order) pass 1 of
    // the next row never waits behind the previous row's stores.
    cf v[32];
    {
        const int row0 = blockIdx.x;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (x + (size_t) row0 * in_pitch), 0, row0 < batch ? in_len * 4 * IO_ON : 0, 0x00020000);
        const int load_off = thread_id(wave_sgpr) * 8;
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) v[j1] = load_c(r0, load_off, j1 * 8192);
    }

    for (int row = blockIdx.x; row < batch; row += gridDim.x) {
        // row descriptors: wave-uniform base, per-lane 32-bit byte offset, SGPR/immediate steps.
        // The next row's descriptor has zero records past the end of the batch: loads return 0.
        const int next_row = row + gridDim.x;
        const __amdgpu_buffer_rsrc_t rnext = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (x + (size_t) next_row * in_pitch), 0, next_row < batch ? in_len * 4 * IO_ON : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc((void *) (X + (size_t) row * (kM + 1)), 0, (kM + 1) * 8 * IO_ON, 0x00020000);

        three_passes<false>(v, plane, w1024, aux, wave_sgpr, false, true);     // v[p] = Z[k' + 1024 br5(p)]
        // (Requesting part of the next row before pass 3, as irfft64k_kernel does, was tried here too: it needs pass 3's results
        // pinned — the compiler otherwise sinks the butterflies into the post-pass and the merged region takes all 128
        // registers — and then fits 4 loads, which measured 0.842-0.849 ms against 0.845-0.849 ms: nothing.)

        // ---- packed-real post-pass.  Rows 0..15 of this column pair with rows 31..16 of the
        // partner column (odd registers there); fetch them, finish both bins of each pair:
        // xk[k3] = X[k], xm[k3] = X[M-k], k = k' + 1024 k3.
        const int t4 = thread_id(wave_sgpr);
        const int lane = t4 & 63, wave = t4 >> 6;
        const int kp = column_of(wave, lane);
        const int partner_addr = partner_byte_addr(wave, lane);
        cf xk[16], xm[16];
        const cf zmid = v[br5(16)];                        // Z[M/2] in column 0
        if (wave_sgpr == 0) {
            // Column 0 (lane 0, its own partner) pairs row k3 with row 32 - k3 of ITSELF, not 31 - k3: shift its rows 17 .. 31
            // down by one (and row 0 into row 31) BEFORE the exchange, so that the general fetch below is right for it too and
            // the sent rows die with their ds_bpermute — fixing the fetched values up afterwards kept all 32 of them alive
            // across the exchange, in every wave.
#pragma unroll
            for (int r = 16; r < 31; ++r) v[br5(r)] = lane == 0 ? v[br5(r + 1)] : v[br5(r)];
            v[br5(31)] = lane == 0 ? v[br5(0)] : v[br5(31)];
        }
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            const int src = 31 - br5(k3);                  // = br5(31 - k3)
            xm[k3].x = bperm(partner_addr, v[src].x);
            xm[k3].y = bperm(partner_addr, v[src].y);
        }
        {
            const cf wpost = to_cf(aux[kAuxW65536 + kp]);
            const cf post_base = cf{0.5f * wpost.y, -0.5f * wpost.x};          // -(i/2) W_65536^{k'}
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) {
                const cf c = cf{root64_re(k3), root64_im(k3)};                 // W_64^{k3} = W_65536^{1024 k3}
                const cf w = k3 == 0 ? post_base : cmul(post_base, c);
                real_pair(v[br5(k3)], xm[k3], w, 0.5f, xk[k3], xm[k3]);
            }
        }

        // ---- store.  Output rows are 32769 bins long, so a row starts 8 * (row mod 16) bytes
        // past a 128-B line; storing each lane's bins where the FFT left them would cut every
        // 256-B piece across three lines (measured: ~20 % of the HBM rate lost to partial
        // lines).  Instead the spectrum goes through the (now idle) LDS plane half a row at a
        // time and is stored as whole, 128-B aligned lines, 16 B per lane.
        f2 *stage = (f2 *) plane;
        const int skew = (int) ((((size_t) (X + (size_t) row * (kM + 1))) >> 3) & 15);     // bins past a line start
        // half 1: bins [0, 16384 - skew)
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) stage[kp + 1024 * k3] = to_f2(xk[k3]);
        const cf xk15 = xk[15];
        const int load_off = thread_id(wave_sgpr) * 8;
#pragma unroll
        for (int j1 = 0; j1 < 16; ++j1) v[j1] = load_c(rnext, load_off, j1 * 8192);      // xk[] is dead: next row, first half (20 + 12 instead of 16 + 16: measured equal)
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int k = 2 * (t4 + 1024 * m) - skew;       // first bin of this lane's 16-B chunk
            if (m > 0 || k >= 0) {                          // only the row's first chunk can start before bin 0
                const f2 lo2 = stage[k], hi2 = stage[k + 1];
                const f4 q = f4{lo2.x, lo2.y, hi2.x, hi2.y};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q), rout, k * 8, 0, kStream);
            } else if (m == 0 && k == -1) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, stage[0]), rout, 0, 0, kStream);
            }
            if (m & 1) __builtin_amdgcn_sched_barrier(0);       // at most two chunks of staging reads in flight (VGPR budget)
        }
        lds_barrier();
        // half 2: bins [16384 - skew, 32768], staged at index bin - kStage2
        constexpr int kStage2 = kM / 2 - 16;
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) stage[(kM - kStage2) - kp - 1024 * k3] = to_f2(xm[k3]);
        if (kp >= 1009) stage[kp + 15 * 1024 - kStage2] = to_f2(xk15);         // bins 16369..16383
        if (t4 == 0) stage[kM / 2 - kStage2] = f2{zmid.x, -zmid.y};            // bin M/2 = conj Z[M/2] (dsc_fft.h:218)
#pragma unroll
        for (int j1 = 16; j1 < 32; ++j1) v[j1] = load_c(rnext, load_off, j1 * 8192);     // xm[] is dead: next row, second half
        lds_barrier();
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            if (m == 8 && wave != 0) break;                // bins past 32768 + 15 do not exist
            const int k = kM / 2 + 2 * (t4 + 1024 * m) - skew;
            if (m < 8 || k + 1 <= kM) {                     // only the row's last chunks can run past bin M
                const f2 lo2 = stage[k - kStage2], hi2 = stage[k + 1 - kStage2];
                const f4 q = f4{lo2.x, lo2.y, hi2.x, hi2.y};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q), rout, k * 8, 0, kStream);
            } else if (m == 8 && k == kM) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, stage[k - kStage2]), rout, k * 8, 0, kStream);
            }
            if (m & 1) __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                                     // plane free for the next row's exchange 1
    }
}

#endif  // DSC_R2C64K_HELPERS_ONLY

// Inverse packed-real pre-pass with the partner bins LOADED by the lane itself: on entry v[a] = Y[k], v[16 + a] = Y[M - k] for
// the lane's first 16 rows, k = c + 1024 a (Y[M - k] is row 31 - a of the partner column 1024 - c: the same 32 loads per
// lane as reading one's own 32 rows, but the pair is complete without a ds_bpermute).  Each pair gives Z[k] — this lane's
// row a — and Z[M - k] — the PARTNER's row 31 - a, which is the only thing exchanged (32 ds_bpermute instead of 64).
// On exit v[r] = Z[c + 1024 r] / M in natural row order.  y_mid = bin M/2 (column 0 only).
__device__ __forceinline__ void inverse_prepass_direct(cf (&v)[32], cf y_mid, const f2 *aux, int wave_sgpr) {
    constexpr float kScale = 1.0f / (float) kM;            // 2/(2n), dsc_fft.h:232
    const int t1 = thread_id(wave_sgpr);
    const int lane = t1 & 63, wave = t1 >> 6;
    const int c = column_of(wave, lane);
    const int partner_addr = partner_byte_addr(wave, lane);
    if (wave == 0 && lane == 0) { v[0].y = 0.f; v[16].y = 0.f; }          // bins 0 and M: real parts only (dsc_fft.h:227-228)
    const cf wpre = to_cf(aux[kAuxW65536 + c]);
    const cf wq_base = cf{0.5f * kScale * wpre.y, 0.5f * kScale * wpre.x};      // (i/2) conj(W^c) / M
#pragma unroll
    for (int half = 0; half < 2; ++half) {                 // two batches of 8 pairs: VGPR budget, and batch 0 only needs the
        cf zm[8];                                          // loads the pipeline issues first
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = half * 8 + i;
            const cf cj = cf{root64_re(a), -root64_im(a)}; // conj(W_64^a)
            const cf wq = a == 0 ? wq_base : cmul(wq_base, cj);
            real_pair(v[a], v[16 + a], wq, 0.5f * kScale, v[a], zm[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                      // the partner's Z[M - k] of its pair a is my row 31 - a
            const int a = half * 8 + i;
            v[16 + a].x = bperm(partner_addr, zm[i].x);
            v[16 + a].y = bperm(partner_addr, zm[i].y);
        }
    }
    cf w[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) { w[r] = v[r]; w[16 + r] = v[31 - r]; }  // v[16 + a] holds row 31 - a
    if (wave == 0) {
        // Column 0 pairs row a with row 32 - a of ITSELF (and row 0 with bin M): what came back as "row 31 - a" is row
        // 32 - a.  Shift up by one; row 16 = bin M/2 pairs with itself: Z[M/2] = conj Y[M/2] (dsc_fft.h:218).
#pragma unroll
        for (int r = 31; r > 16; --r) w[r] = lane == 0 ? w[r - 1] : w[r];
        w[16] = lane == 0 ? cf{kScale * y_mid.x, -kScale * y_mid.y} : w[16];
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = w[r];
}

// Tail shared by the inverse and the fused kernels: v[p] = z[t + 1024 br5(p)] (time samples of
// one aligned row) leaves through the LDS staging area, half a row (16384 complex) at a time, as
// whole lines, 16 B per lane; as soon as a half has left the registers the next row's loads are
// issued into them (`issue_next(first_half)`), ahead of this row's stores.  Register p holds
// row br5(p): even p = rows 0..15 = first half.  On return the even registers hold what
// issue_next(true) loaded, the odd ones what issue_next(false) loaded.
template<typename IssueNext>
__device__ __forceinline__ void staged_time_store(cf (&v)[32], float *plane, __amdgpu_buffer_rsrc_t rout, int wave_sgpr,
                                                  IssueNext issue_next) {
    f2 *stage = (f2 *) plane;
    const int t4 = thread_id(wave_sgpr);
#pragma unroll
    for (int p = 0; p < 32; p += 2) stage[t4 + 1024 * br5(p)] = to_f2(v[p]);
    issue_next(true);
    lds_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k = 2 * (t4 + 1024 * m);
        const f2 lo2 = stage[k], hi2 = stage[k + 1];
        const f4 q = f4{lo2.x, lo2.y, hi2.x, hi2.y};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q), rout, k * 8, 0, kStream);
        if (m & 1) __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
#pragma unroll
    for (int p = 1; p < 32; p += 2) stage[t4 + 1024 * (br5(p) - 16)] = to_f2(v[p]);
    issue_next(false);
    lds_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int k = 2 * (t4 + 1024 * m);
        const f2 lo2 = stage[k], hi2 = stage[k + 1];
        const f4 q = f4{lo2.x, lo2.y, hi2.x, hi2.y};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q), rout, (kM / 2 + k) * 8, 0, kStream);
        if (m & 1) __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();                                         // plane free for the next row's exchange 1
}

// The pipelined loads land rows 0..15 in the even registers and rows 16..31 in the odd ones:
// back to natural row order (a renaming, every index is a constant).
__device__ __forceinline__ void unzip_rows(cf (&v)[32]) {
    cf nxt[32];
#pragma unroll
    for (int a = 0; a < 16; ++a) { nxt[a] = v[2 * a]; nxt[16 + a] = v[2 * a + 1]; }
#pragma unroll
    for (int a = 0; a < 32; ++a) v[a] = nxt[a];
}

#ifndef DSC_R2C64K_HELPERS_ONLY       // fft_c2c_32k.hip includes this file for the building blocks only
// ------------------------------------------------------------------------------------------
// inverse: X [batch][32769] c32  ->  x [batch][65536] f32        (dsc_irfft, dsc_fft.h:194-236)
//
// The forward pipeline run backwards.  Each lane reads its column's first 16 rows AND their pairing partners (rows 31..16
// of the mirror column), the packed-real pre-pass completes the pairs and exchanges only the results through ds_bpermute, and three
// conjugate-twiddle passes end with thread t holding z[t + 1024 r]: the time samples leave as
// aligned, coalesced 8-B stores.  The 2/(2n) scale is folded into the pre-pass constants.
// in_pitch: bins between input rows; in_len <= 32769 bins per row are used, missing ones read as zero (dsc.cpp:2149-2157)
__global__ __launch_bounds__(1024) void irfft64k_kernel(const f2 *__restrict__ X, float *__restrict__ x, int batch,
                                                        const f2 *__restrict__ aux, int in_pitch, int in_len PROBE_ARGS) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *plane = lds;
    f2 *w1024 = (f2 *) (lds + kPlaneFloats);
    w1024[threadIdx.x] = aux[kAuxW1024 + threadIdx.x];
    __syncthreads();

    const int wave_sgpr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // Software pipelined like the forward kernel: the time samples leave through the LDS
    // staging area (whole lines, 16 B per lane) and the next row's bins are requested as soon
    // as the staging writes have freed the registers, ahead of this row's stores.
    // Loads keep the default cache policy: spectrum rows are skewed by 8 B x row, so neighbouring
    // waves share their boundary lines (nt loads measured 5 % slower on this kernel).
    // Register convention at the top of a row: v[a] = Y[c + 1024 a], v[16 + a] = Y[M - c - 1024 a] (a < 16): see inverse_prepass_direct.
    cf v[32];
    cf y_mid = cf{0.f, 0.f};
    {
        const int row0 = blockIdx.x;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (X + (size_t) row0 * in_pitch), 0, row0 < batch ? in_len * 8 * IO_ON : 0, 0x00020000);
        const int t0 = thread_id(wave_sgpr);
        const int c = column_of(t0 >> 6, t0 & 63);
        const int pv = (kM - c - 15 * 1024) * 8;                                     // Y[M - c - 1024 a] = pv + (15 - a) * 8192
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            v[a] = load_c_spec(r0, c * 8, a * 8192);
            v[16 + a] = load_c_spec(r0, pv, (15 - a) * 8192);
        }
        if (c == 0) y_mid = load_c_spec(r0, (kM / 2) * 8, 0);
    }

    for (int row = blockIdx.x; row < batch; row += gridDim.x) {
        const int next_row = row + gridDim.x;
        const __amdgpu_buffer_rsrc_t rnext = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (X + (size_t) next_row * in_pitch), 0, next_row < batch ? in_len * 8 * IO_ON : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc((void *) (x + (size_t) row * 65536), 0, 65536 * 4 * IO_ON, 0x00020000);

        inverse_prepass_direct(v, y_mid, aux, wave_sgpr);
        // The next row's first kEarly pairs (the ones the pre-pass consumes first) are requested BEFORE pass 3 — from the second
        // exchange on only v[] is live, which leaves room for them — the next eight when the first half of this row has left for
        // the staging area, the rest after the second half.  (kEarly = 0: everything in the tail, as the forward kernel does.)
        constexpr int kEarly = DSC_IRFFT_EARLY_PAIRS;
        cf early[kEarly > 0 ? 2 * kEarly : 1];
        three_passes<true>(v, plane, w1024, aux, wave_sgpr, true, false, [&]() {      // v[p] = z[t + 1024 br5(p)]
            if constexpr (kEarly > 0) {
                const int t3 = thread_id(wave_sgpr);
                const int c3 = column_of(t3 >> 6, t3 & 63);
                const int pv3 = (kM - c3 - 15 * 1024) * 8;
#pragma unroll
                for (int a = 0; a < kEarly; ++a) {
                    early[2 * a] = load_c_spec(rnext, c3 * 8, a * 8192);
                    early[2 * a + 1] = load_c_spec(rnext, pv3, (15 - a) * 8192);
                }
            }
        });

#if DSC_IRFFT_PIN_PASS3
        // pass 3 complete here (the compiler otherwise sinks its butterflies towards the staging writes, where they overlap the
        // loads' registers)
#pragma unroll
        for (int p = 0; p < 32; p += 8)
            asm volatile("" : "+v"(v[p].x), "+v"(v[p].y), "+v"(v[p + 1].x), "+v"(v[p + 1].y), "+v"(v[p + 2].x), "+v"(v[p + 2].y), "+v"(v[p + 3].x),
                         "+v"(v[p + 3].y), "+v"(v[p + 4].x), "+v"(v[p + 4].y), "+v"(v[p + 5].x), "+v"(v[p + 5].y), "+v"(v[p + 6].x), "+v"(v[p + 6].y),
                         "+v"(v[p + 7].x), "+v"(v[p + 7].y));
#endif
        const int t4 = thread_id(wave_sgpr);
        const int c = column_of(t4 >> 6, t4 & 63);
        const int pv = (kM - c - 15 * 1024) * 8;
        staged_time_store(v, plane, rout, wave_sgpr, [&](bool first_half) {
            // consumption order of the pre-pass: the pairs in ascending order (and bin M/2 with the first)
            if (first_half) {
                y_mid = cf{0.f, 0.f};
                if (c == 0) y_mid = load_c_spec(rnext, (kM / 2) * 8, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {              // the even registers are free now
                    v[2 * i] = load_c_spec(rnext, c * 8, (kEarly + i) * 8192);
                    v[16 + 2 * i] = load_c_spec(rnext, pv, (15 - kEarly - i) * 8192);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8 - kEarly; ++i) {     // the odd ones
                    v[2 * i + 1] = load_c_spec(rnext, c * 8, (kEarly + 8 + i) * 8192);
                    v[17 + 2 * i] = load_c_spec(rnext, pv, (7 - kEarly - i) * 8192);
                }
            }
        });
        {   // back to the convention above (a renaming, every index is a constant)
            cf nxt[32];
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                if (a < kEarly)          { nxt[a] = early[2 * a];             nxt[16 + a] = early[2 * a + 1]; }
                else if (a < kEarly + 8) { nxt[a] = v[2 * (a - kEarly)];      nxt[16 + a] = v[16 + 2 * (a - kEarly)]; }
                else                     { nxt[a] = v[2 * (a - kEarly - 8) + 1]; nxt[16 + a] = v[17 + 2 * (a - kEarly - 8)]; }
            }
#pragma unroll
            for (int a = 0; a < 32; ++a) v[a] = nxt[a];
        }
    }
}

// ------------------------------------------------------------------------------------------
// fused README filterFFT (README.md:113-135): y = irfft(rfft(s) * H), H [32769] c32 shared by
// all rows.  The reference runs rfft, mul, irfft as three passes over memory with two
// materialised spectra; here the spectrum never leaves the register file: forward passes,
// post-pass pairs (X[k], X[M-k]), times (H[k], H[M-k]) read from L2, inverse pre-pass on the
// same pair in the same lane, inverse passes.  4 B/sample in, 4 B/sample out.
__global__ __launch_bounds__(1024) void filter64k_kernel(const float *__restrict__ x, const f2 *__restrict__ H,
                                                         float *__restrict__ y, int batch, const f2 *__restrict__ aux, int in_pitch,
                                                         int in_len) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *plane = lds;
    f2 *w1024 = (f2 *) (lds + kPlaneFloats);
    w1024[threadIdx.x] = aux[kAuxW1024 + threadIdx.x];
    __syncthreads();

    const int wave_sgpr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void *) H, 0, (kM + 1) * 8, 0x00020000);
    constexpr float kScale = 1.0f / (float) kM;

    cf v[32];                                              // software pipelined like the other two kernels
    {
        const int row0 = blockIdx.x;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (x + (size_t) row0 * in_pitch), 0, row0 < batch ? in_len * 4 : 0, 0x00020000);
        const int load_off = thread_id(wave_sgpr) * 8;
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) v[j1] = load_c(r0, load_off, j1 * 8192);
    }
    for (int row = blockIdx.x; row < batch; row += gridDim.x) {
        const int next_row = row + gridDim.x;
        const __amdgpu_buffer_rsrc_t rnext = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (x + (size_t) next_row * in_pitch), 0, next_row < batch ? in_len * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc((void *) (y + (size_t) row * 65536), 0, 65536 * 4, 0x00020000);
        three_passes<false>(v, plane, w1024, aux, wave_sgpr, false, true);     // v[p] = Z[c + 1024 br5(p)]

        // ---- post-pass, multiply by H, pre-pass: all on the pair (k, M-k) held by this lane.
        // In place: register br5(r) holds row r of the column before (Z) and after (Z'/M).
        {
            const int t1 = thread_id(wave_sgpr);
            const int lane = t1 & 63, wave = t1 >> 6;
            const int c = column_of(wave, lane);
            const int partner_addr = partner_byte_addr(wave, lane);
            const cf wc = to_cf(aux[kAuxW65536 + c]);
            const cf post_base = cf{0.5f * wc.y, -0.5f * wc.x};                 // -(i/2) W^c
            const cf pre_base = cf{0.5f * kScale * wc.y, 0.5f * kScale * wc.x}; // (i/2M) conj(W^c)
            if (wave == 0) {                               // column 0, row 16: bin M/2 pairs with itself
                const cf zmid = v[br5(16)];
                const cf hmid = load_c_cached(rh, (kM / 2) * 8, 0);
                const cf pmid = cmul(cf{zmid.x, -zmid.y}, hmid);               // X[M/2] H[M/2]
                v[br5(16)] = lane == 0 ? cf{kScale * pmid.x, -kScale * pmid.y} : v[br5(16)];
            }
            // Column 0 pairs row a with row 32 - a of itself (row 0 with itself: X[0], X[M] both come
            // from Z[0]).  Shift its rows 17..31 down by one so that the general "row 31 - a" applies;
            // its own row 16 result (above) is parked meanwhile.
            cf park = v[br5(16)];
            if (wave == 0) {
#pragma unroll
                for (int r = 16; r < 31; ++r) v[br5(r)] = lane == 0 ? v[br5(r + 1)] : v[br5(r)];
                v[br5(31)] = lane == 0 ? v[br5(0)] : v[br5(31)];
            }
#pragma unroll
            for (int part = 0; part < 8; ++part) {         // eight batches of 2 pairs: VGPR budget
                cf q[2], zm[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int a = part * 2 + i;
                    q[i].x = bperm(partner_addr, v[br5(31 - a)].x);
                    q[i].y = bperm(partner_addr, v[br5(31 - a)].y);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int a = part * 2 + i;
                    const cf w64 = cf{root64_re(a), root64_im(a)};
                    cf xk, xm;
                    real_pair(v[br5(a)], q[i], a == 0 ? post_base : cmul(post_base, w64), 0.5f, xk, xm);
                    // bins k = c + 1024 a and M - k, times the filter
                    const cf hk = load_c_cached(rh, c * 8, a * 8192);
                    const cf hm = load_c_cached(rh, (kM - 15 * 1024 - c) * 8, (15 - a) * 8192);
                    cf pk = cmul(xk, hk), pm = cmul(xm, hm);
                    if (a == 0) {                          // dsc_fft.h:227-228: bins 0 and M enter irfft through their real parts
                        pk.y = (c == 0) ? 0.f : pk.y;
                        pm.y = (c == 0) ? 0.f : pm.y;
                    }
                    const cf w64c = cf{root64_re(a), -root64_im(a)};
                    real_pair(pk, pm, a == 0 ? pre_base : cmul(pre_base, w64c), 0.5f * kScale, v[br5(a)], zm[i]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {              // Z'[M-k] belongs to the partner's row 31 - a
                    const int a = part * 2 + i;
                    v[br5(31 - a)].x = bperm(partner_addr, zm[i].x);
                    v[br5(31 - a)].y = bperm(partner_addr, zm[i].y);
                }
                __builtin_amdgcn_sched_barrier(0);         // keep the next batch's H loads and bpermutes out of this one
            }
            if (wave == 0) {                               // undo the shift of column 0
#pragma unroll
                for (int r = 31; r > 16; --r) v[br5(r)] = lane == 0 ? v[br5(r - 1)] : v[br5(r)];
                v[br5(16)] = lane == 0 ? park : v[br5(16)];
            }
        }
        cf z[32];                                          // inverse input in natural row order (a renaming)
#pragma unroll
        for (int r = 0; r < 32; ++r) z[r] = v[br5(r)];
        // the next row's first kEarlyX loads are requested before the inverse transform's last pass (see irfft64k_kernel)
        constexpr int kEarlyX = DSC_FILTER_EARLY_LOADS;
        cf early[kEarlyX > 0 ? kEarlyX : 1];
        three_passes<true>(z, plane, w1024, aux, wave_sgpr, true, false, [&]() {      // z[p] = y[2(t + 1024 br5(p)) .. +1]
            if constexpr (kEarlyX > 0) {
                const int off = thread_id(wave_sgpr) * 8;
#pragma unroll
                for (int j1 = 0; j1 < kEarlyX; ++j1) early[j1] = load_c(rnext, off, j1 * 8192);
            }
        });
        {
            const int load_off = thread_id(wave_sgpr) * 8;
            staged_time_store(z, plane, rout, wave_sgpr, [&](bool first_half) {
                if (first_half) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) z[2 * i] = load_c(rnext, load_off, (kEarlyX + i) * 8192);
                } else {
#pragma unroll
                    for (int i = 0; i < 16 - kEarlyX; ++i) z[2 * i + 1] = load_c(rnext, load_off, (kEarlyX + 16 + i) * 8192);
                }
            });
#pragma unroll
            for (int j1 = 0; j1 < 32; ++j1) v[j1] = j1 < kEarlyX ? early[j1] : j1 < kEarlyX + 16 ? z[2 * (j1 - kEarlyX)] : z[2 * (j1 - kEarlyX - 16) + 1];
        }
    }
}

#endif  // DSC_R2C64K_HELPERS_ONLY

}  // namespace

#ifndef DSC_R2C64K_HELPERS_ONLY
size_t dsc_r2c64k_table_bytes() { return (size_t) kAuxEntries * sizeof(float) * 2; }

void dsc_r2c64k_build_tables(void *host_dst) {
    float *o = (float *) host_dst;
    auto put = [&](int at, long long k, long long n) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double) k / (long double) n;
        o[2 * at] = (float) cosl(a);
        o[2 * at + 1] = (float) sinl(a);
    };
    for (int m = 0; m < kTabEntries; ++m) {
        put(kAuxW1024 + m, m, 1024);
        put(kAuxW32768 + m, m, 32768);
        put(kAuxW65536 + m, m, 65536);
    }
    // exact values on the axes (bin M/2 relies on W_65536^{16384 * ...} only through constants,
    // but keep the tables clean too)
    o[2 * (kAuxW1024 + 256)] = 0.f;  o[2 * (kAuxW1024 + 256) + 1] = -1.f;
    o[2 * (kAuxW1024 + 512)] = -1.f; o[2 * (kAuxW1024 + 512) + 1] = 0.f;
    o[2 * (kAuxW1024 + 768)] = 0.f;  o[2 * (kAuxW1024 + 768) + 1] = 1.f;
}

void dsc_launch_rfft64k(const float *x, void *X, int batch, int in_pitch, int in_len, const void *aux, int n_cu, hipStream_t stream) {
    if (batch <= 0) return;
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) rfft64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    }
    const int grid = batch < n_cu ? batch : n_cu;
    DSC_LAUNCH(rfft64k_kernel, dim3(grid), dim3(1024), kLdsBytes, stream, x, (f2 *) X, batch, (const f2 *) aux, in_pitch, in_len PROBE_NULL);
}

void dsc_launch_irfft64k(const void *X, float *x, int batch, int in_pitch, int in_len, const void *aux, int n_cu, hipStream_t stream) {
    if (batch <= 0) return;
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) irfft64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    }
    const int grid = batch < n_cu ? batch : n_cu;
    DSC_LAUNCH(irfft64k_kernel, dim3(grid), dim3(1024), kLdsBytes, stream, (const f2 *) X, x, batch, (const f2 *) aux, in_pitch, in_len PROBE_NULL);
}
void dsc_launch_filter64k(const float *s, const void *H, float *y, int batch, int in_pitch, int in_len, const void *aux, int n_cu,
                          hipStream_t stream) {
    if (batch <= 0) return;
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) filter64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    }
    const int grid = batch < n_cu ? batch : n_cu;
    DSC_LAUNCH(filter64k_kernel, dim3(grid), dim3(1024), kLdsBytes, stream, s, (const f2 *) H, y, batch, (const f2 *) aux, in_pitch, in_len);
}
#endif  // DSC_R2C64K_HELPERS_ONLY
