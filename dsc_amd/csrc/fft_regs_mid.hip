// fft_regs_mid.hip — register-resident transforms of contiguous lines with complex length
// L = 1024 B, B in {2, 4, 8, 16}  (real lengths 4096 .. 32768, complex 2048 .. 16384) and, for
// complex data, B = 32 (L = 32768), f32.
// The same kernels in f64 for L = 256 .. 16384.
//
// The design of fft_r2c_64k.hip, parameterised by B: a line lives in
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

#include "fft_regs_common.h"

namespace {

// TWO = false: L = 1024 B, three passes (32 x 32 x B), T = 32 B threads per line.
// TWO = true:  L = 32 B,   two passes  (32 x B),       T = B threads per line (a wave holds 64 / B lines).
constexpr int two_pass_stride(int B) {
    const int base = 32 * (B + 1);
    for (int p = 0; p < 64; ++p) {
        const int r = (base + p) % 64;
        if (B >= 64 || (r % B == 0 && ((r / B) & 1))) return base + p;
    }
    return base;
}

// V: 0 = the transforms (fft_mid_kernel), 1 = the fused filter (fft_mid_filter_kernel) — the two want different group sizes at some lengths
template<typename R, int B, bool TWO, int V = 0> struct mid_cfg {
    static constexpr bool DP = sizeof(R) == 8;
    static constexpr int T = TWO ? B : 32 * B;       // threads per line
    static constexpr int L = 32 * T;                 // complex length
    static constexpr int COLS = TWO ? 32 : 1024;     // columns entering the last pass (DFT_B down each)
    // threads per workgroup.  f32: a thread needs ~100 VGPRs and 132-264 B of LDS, so two 512-thread groups (or three
    // of 256) share a CU and overlap each other's load / compute / store phases.  f64: twice the registers (one
    // 512-thread group or several smaller ones per CU) and twice the LDS, which decides the group size.
#ifdef DSC_MID_PADDED_SHORT
    static constexpr bool PACKED = false;
#else
    // Three-pass lines of 2048 points (B = 2): the last exchange keeps its 2-value rows UNPADDED and reads a row as one 8- /
    // 16-byte access (conflict free, like the stride-1 writes) — the padded pitch 3 cost 50 % more LDS and left two 256-thread
    // groups (two waves per SIMD) on a CU in f32; unpadded, two 512-thread groups fit (four per SIMD): irfft N = 4096 64.6 ->
    // 71 %, fft 68 -> 73 %, fused filter 31 -> 38.6 %.  (f32 B = 4 measured too: 1 % slower, stays padded.)  f64 B = 4 (4096-point
    // lines, one 128-thread group per line): 4-value rows unpadded, read as two 16-byte accesses (2-way conflicts) — three groups
    // per CU instead of two: rfft N = 8192 62.5 -> 69.3 %, irfft 60 -> 71.2 %, fft c64 71 -> 78.7 %.
#ifdef DSC_MID_NO_F64_B4
    static constexpr bool PACKED = !TWO && B == 2;
#else
    static constexpr bool PACKED = !TWO && (B == 2 || (DP && B == 4));
#endif
#endif
#ifdef DSC_MID_NO_HALF_TABLE   // (A/B switch)
    static constexpr bool HALF = false;
#else
    // f64 lines of 8192 points (B = 8): two lines per 512-thread group filled the whole LDS (144 KiB plane + 16 KiB table), i.e. ONE
    // group per CU with nothing to overlap its load / compute / store phases with.  One line per 256-thread group and HALF the
    // W_1024 table (W^(m + 512) = -W^m) is exactly 80 KiB: two independent groups per CU.
    static constexpr bool HALF = DP && !TWO && B == 8;
#endif
    // f32 group sizes, round 3 (tools/build_mid_variant.sh, one box, gpurun_out/r3e/mid_nt.txt): 2048-point lines (B = 2) in groups of
    // 256 threads instead of 512 — four independent groups per CU instead of two: fused filter N = 4096 0.701 -> 0.646 ms (38.3 -> 41.6 %),
    // irfft 71.2 -> 73.2 %, rfft 68.7 -> 69.8 % (128: no better); 8192-point lines (B = 8) in groups of 256 for the FILTER only (0.772 ->
    // 0.695 ms, 34.8 -> 38.6 %; the transforms lose 2 % with it); 4096-point lines (B = 4) stay at 256 (128: filter 40.4 -> 37.9 %; 512: everything 25 - 35 % slower).
#ifndef DSC_MID_NT_F32_B2
#define DSC_MID_NT_F32_B2 256
#endif
#ifndef DSC_MID_NT_F32_B4
#define DSC_MID_NT_F32_B4 256
#endif
#ifndef DSC_MID_NT_F32_B8
#define DSC_MID_NT_F32_B8 512
#endif
#ifndef DSC_MID_NT_F32_TWO
#define DSC_MID_NT_F32_TWO 128      // two-pass lines (512, 1024 points): groups of 128 (was 256): irfft N = 2048 66.8 -> 70.5 %, fft c32 L = 1024 68.7 -> 72.8 %, fused filter N = 1024 46.4 -> 50.4 %; 512: worse
#endif
#ifndef DSC_MID_NT_F64_TWO16
#define DSC_MID_NT_F64_TWO16 256
#endif
#ifndef DSC_MID_NT_F32_B8_FILTER
#define DSC_MID_NT_F32_B8_FILTER 256
#endif
    static constexpr int NT = DP ? (TWO ? (B >= 32 ? 128 : DSC_MID_NT_F64_TWO16) : HALF ? 256 : (B >= 8 ? 512 : 128))
                                 : (TWO ? DSC_MID_NT_F32_TWO : B >= 32 ? 1024 : B == 16 ? 512 : B == 8 ? (V == 1 ? DSC_MID_NT_F32_B8_FILTER : DSC_MID_NT_F32_B8)
                                                                                  : B == 4 ? DSC_MID_NT_F32_B4 : (PACKED ? DSC_MID_NT_F32_B2 : 256));
    static constexpr int G = NT / T;                 // lines per workgroup
    // f64 lines of 16384 points: ONE 512-thread group per CU (131 KiB plane) — nothing to overlap its load / compute / store phases
    // with.  PIPE: the group is persistent (lines blockIdx.x, + gridDim.x, ...) and requests its next line into the registers the stores
    // (complex, inverse real) or the last staging write (forward real) have just freed, the way the 65536-point kernels do.
#ifdef DSC_MID_NO_PIPE
    static constexpr bool PIPE = false;
#else
    static constexpr bool PIPE = DP && !TWO && B == 16 && V == 0;
#endif
    static constexpr int WAVES_PER_EU = DP ? 2 : (TWO ? 2 : (B >= 8 || PACKED) ? 4 : 2);   // f32: <= 128 VGPRs where two 512-thread groups share a CU
    static constexpr int P1 = 33;                    // exchange-1 row pitch (values): odd
    static constexpr int P2 = PACKED ? B : B + 1;    // last-exchange row pitch
    static constexpr int SP = L + 1;                 // staging pitch per line (bins 0 .. L)
    // values per line in the last exchange.  Two-pass: padded so that the stride is an ODD multiple of B mod 64 — the lanes of a
    // wave are (line, j3) pairs and then hit 64 different banks (an unpadded 32 (B + 1) is = 32 mod 64: up to 16-way conflicts).
    // f64 takes the SMALLEST such pad: with stride = B mod 64 the 512-point lines (B = 16) carry 48 spare values each, which cost
    // the second workgroup per CU (83.8 -> 79.7 KiB: rfft f64 N = 1024 63.7 -> 67.9 %, irfft 64.3 -> 69.2 %).  In f32 the same
    // change buys a fourth workgroup (41.9 -> 39.9 KiB) and measured 1-2 % SLOWER: f32 keeps stride = B mod 64.
    static constexpr int LSTRIDE = TWO ? (DP ? two_pass_stride(B) : 32 * (B + 1) + (((B - 32 - 32 * B) % 64) + 64) % 64) : 1024 * P2;
    static constexpr int PLANE_MIN = G * LSTRIDE;    // padded: >= G*T*P1 (three-pass) and >= G*SP; unpadded rows: take the largest
    static constexpr int PLANE_X1 = TWO ? 0 : G * T * 33, PLANE_ST = G * (32 * T + 1);
    static constexpr int PLANE = ((PLANE_MIN > PLANE_X1 ? (PLANE_MIN > PLANE_ST ? PLANE_MIN : PLANE_ST) : (PLANE_X1 > PLANE_ST ? PLANE_X1 : PLANE_ST)) + 3) & ~3;
    static constexpr int CPT = 32 / B;               // columns per thread in the last pass
    static constexpr int TABLE = TWO ? L : HALF ? 512 : 1024;     // LDS twiddle table: W_L^m (two-pass) or W_1024^m (HALF: m < 512)
    static constexpr int TABLE_STRIDE = TWO ? 1 : B;
};

template<typename R, int B, bool TWO, int V = 0>
constexpr size_t mid_lds_bytes() { return ((size_t) mid_cfg<R, B, TWO, V>::PLANE + 2 * mid_cfg<R, B, TWO, V>::TABLE) * sizeof(R); }

// W_1024^m from the LDS table; HALF: the table holds m < 512 and W^(m + 512) = -W^m
template<typename R, bool HALF>
__device__ __forceinline__ cpx<R> table_entry(const cpx<R> *wtab, int m) {
    if constexpr (HALF) {
        const cpx<R> w = wtab[m & 511];
        return (m & 512) ? cpx<R>{-w.x, -w.y} : w;
    } else {
        return wtab[m];
    }
}

// The transform proper, shared by fft_mid_kernel and fft_mid_filter_kernel.  In: v[j1] = z[T j1 + t] of line g (natural
// register order).  Out: v[i B + p] = bin k = (t + T i) + COLS brev(p) ("column layout").  Ends with an LDS barrier, i.e.
// the plane is free on return (three-pass) or untouched since the last barrier.
template<typename R, int B, bool TWO, bool INV, int V = 0>
__device__ __forceinline__ void mid_passes(cpx<R> (&v)[32], R *plane, const cpx<R> *wtab, const cpx<R> *__restrict__ tw_full, int g, int t,
                                           int tid) {
    using C = cpx<R>;
    using cfg = mid_cfg<R, B, TWO, V>;
    constexpr int T = cfg::T, P1 = cfg::P1, P2 = cfg::P2, CPT = cfg::CPT;
    const int hi = TWO ? 0 : t / B, lo = TWO ? t : t % B;
    C u[32];
    if constexpr (!TWO) {
        // ---- pass 1 over j1, twiddle W_1024^{j2 k1}
        dft_n<R, INV, 32>(v);
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            const C w = table_entry<R, cfg::HALF>(wtab, hi * k1);
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
    if constexpr (TWO && B == 1) {                                  // one thread per 32-point line: done
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = u[brev(k, 5)];
        return;
    }
    {
        if constexpr (!TWO) {
            const C tw2_base = tw_full[hi * lo];
            u[0] = INV ? cmulc(u[0], tw2_base) : cmul(u[0], tw2_base);
#pragma unroll
            for (int k2 = 1; k2 < 32; ++k2) {
                const C w = cmul(tw2_base, table_entry<R, cfg::HALF>(wtab, (32 / B) * lo * k2));
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
        R *wr = plane + g * cfg::LSTRIDE + hi * P2 + lo;
        const R *rd = plane + g * cfg::LSTRIDE + t * P2;
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) wr[k2 * CS * P2] = u[brev(k2, 5)].x;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            if constexpr (cfg::PACKED) {
                typedef R row_t __attribute__((ext_vector_type(B)));
                const row_t q = *(const row_t *) (rd + i * T * P2);
#pragma unroll
                for (int m = 0; m < B; ++m) v[i * B + m].x = q[m];
            } else {
#pragma unroll
                for (int m = 0; m < B; ++m) v[i * B + m].x = rd[i * T * P2 + m];
            }
        }
        lds_barrier();
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) wr[k2 * CS * P2] = u[brev(k2, 5)].y;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            if constexpr (cfg::PACKED) {
                typedef R row_t __attribute__((ext_vector_type(B)));
                const row_t q = *(const row_t *) (rd + i * T * P2);
#pragma unroll
                for (int m = 0; m < B; ++m) v[i * B + m].y = q[m];
            } else {
#pragma unroll
                for (int m = 0; m < B; ++m) v[i * B + m].y = rd[i * T * P2 + m];
            }
        }
        lds_barrier();
    }
    // ---- last pass over j3: CPT DFTs of B points; v[i B + p] = bin k = (t + T i) + COLS brev(p)
    dft_columns<R, INV, B>(v, std::make_integer_sequence<int, CPT>{});

}

// MODE: DSC_MODE_C2C, DSC_MODE_R2C_CAST (L reals in), DSC_MODE_R2C_PACKED (forward only), DSC_MODE_C2R_PACKED (inverse only)
// PAD: input lines have a pitch of in_pitch_b bytes and in_len_b valid bytes; the rest of the transform length reads as zero
// (zero padding / cropping of dsc_fft / dsc_rfft / dsc_irfft with n != axis length, dsc.cpp:1990-1998, 2125-2133, 2149-2157).
template<typename R, int B, bool TWO, int MODE, bool INV, bool PAD>
__global__ __launch_bounds__((mid_cfg<R, B, TWO>::NT), (mid_cfg<R, B, TWO>::WAVES_PER_EU)) void fft_mid_kernel(
    const cpx<R> *__restrict__ in, cpx<R> *__restrict__ out, long long n_lines, const cpx<R> *__restrict__ tw_full,
    const cpx<R> *__restrict__ tw_real, R scale, int in_pitch_b, int in_len_b) {
    using C = cpx<R>;
    using cfg = mid_cfg<R, B, TWO>;
    constexpr int T = cfg::T, L = cfg::L, G = cfg::G, NT = cfg::NT, SP = cfg::SP, CPT = cfg::CPT;
    constexpr int COLS = cfg::COLS;
    constexpr int LOGB = ilog2(B);
    constexpr int CB = (int) sizeof(C);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *wtab = (C *) (plane + cfg::PLANE);

    const int tid = threadIdx.x;
    const int g = T >= 64 ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;      // line within the group
    int t = tid - g * T;
    constexpr bool PIPE = cfg::PIPE;
    static_assert(!PIPE || G == 1, "the persistent form walks single lines");
    long long line0 = PIPE ? (long long) blockIdx.x : (long long) blockIdx.x * G;
    const long long left = n_lines - line0;
    const int n_valid = left < G ? (int) left : G;                  // lines past the end read zeros, their stores are dropped
    constexpr int in_pitch = MODE == DSC_MODE_C2R_PACKED ? L + 1 : L;
    constexpr int out_pitch = MODE == DSC_MODE_R2C_PACKED ? L + 1 : L;
    constexpr int IB = MODE == DSC_MODE_R2C_CAST ? (int) sizeof(R) : CB;      // bytes per input element
    // measured per mode (tools/bench_mid.py): complex transforms gain 1-3 % from streaming both ways, the inverse real
    // ones from streaming their (aligned) stores at L >= 2048; everything that touches rows of L + 1 bins, and the
    // forward real transform as a whole, is faster cached (see fft_regs_common.h)
    constexpr bool kComplex = MODE == DSC_MODE_C2C || MODE == DSC_MODE_R2C_CAST;
    // the persistent f64 lines (PIPE): streaming loads + cached stores measured best for the real transforms (tools/r03_call_p.sh:
    // rfft 55.6 -> 56.2 %, irfft 50.9 -> 52.4 %; streaming stores: rfft 52 %)
#ifndef DSC_MID_PIPE_REAL_LOAD
#define DSC_MID_PIPE_REAL_LOAD kStream
#endif
#ifndef DSC_MID_PIPE_REAL_STORE
#define DSC_MID_PIPE_REAL_STORE kCached
#endif
    constexpr int LOADP = kComplex ? kStream : cfg::PIPE ? DSC_MID_PIPE_REAL_LOAD : kCached;
    constexpr int STOREP = kComplex || (MODE == DSC_MODE_C2R_PACKED && !TWO) ? kStream : (cfg::PIPE && MODE == DSC_MODE_R2C_PACKED) ? DSC_MID_PIPE_REAL_STORE : kCached;
    const int pitch_b = PAD ? in_pitch_b : in_pitch * IB;
    auto in_rsrc = [&](long long first) {
        return __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) in + first * pitch_b), 0,
                                                 PAD ? (n_valid - 1) * pitch_b + in_len_b : n_valid * in_pitch * IB, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rin = in_rsrc(line0);
    __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (out + line0 * out_pitch), 0, n_valid * out_pitch * CB, 0x00020000);
    int vin = g * pitch_b + t * IB;                                // byte offset of element t of this thread's line
    // element `idx` of the line (units of IB bytes); with PAD, elements past the valid length read zero (their offset is
    // pushed out of the descriptor's range; one line per group: the descriptor ends with the line's valid bytes and does that by
    // itself) and a sample pair cut by an odd length keeps its real part only
    auto load_from = [&](const __amdgpu_buffer_rsrc_t &rs, int idx) -> C {
        constexpr int kOut = 0x7f000000;
        if constexpr (MODE == DSC_MODE_R2C_CAST) {
            return buf_load_real<LOADP>(rs, (!PAD || G == 1 || (t + idx) * IB < in_len_b) ? vin : kOut, idx * IB, R{});
        } else {
            C val = buf_load<LOADP>(rs, (!PAD || G == 1 || (t + idx) * CB < in_len_b) ? vin : kOut, idx * CB, R{});
            if (PAD && MODE == DSC_MODE_R2C_PACKED && (t + idx) * CB + CB > in_len_b) val.y = (R) 0;
            return val;
        }
    };
    auto load_elem = [&](int idx) -> C { return load_from(rin, idx); };
    int vout = (g * out_pitch + t) * CB;
    R *stage = plane + g * SP;

    for (int i = tid; i < cfg::TABLE; i += NT) wtab[i] = tw_full[(long long) i * cfg::TABLE_STRIDE];    // W_1024^m = W_L^{B m}

    C v[32];
#ifdef DSC_MID_OLD_PRE
    constexpr bool PRE_ONCE = false;
#else
    // the inverse pre-pass that moves only the upper halves (see there).  Measured against the two-exchange form (tools/r03_call_s.sh):
    // f32 lines of 16384 points 67.5 -> 70.4 % (it replaced a form that loaded the partners from memory, 4 spilled registers), the
    // persistent f64 lines of 16384 points 52.3 -> 55.0 % (43 spilled registers -> 0), everything else within +- 0.8 points; the two
    // f64 forms that measured 0.7 - 1 point lower with it (lines of 1024 and 2048 points) keep the two-exchange form.
    constexpr bool PRE_ONCE = !(sizeof(R) == 8 && ((TWO && B == 32) || (!TWO && B == 2)));
#endif
#ifdef DSC_MID_OLD_POST
    constexpr bool POST_ONCE = false;
#else
    // measured (tools/r03_call_o.sh): f32 + 0 - 1.7 points at every three-pass length; f64 lines of 2048 - 8192 points lose 1 - 2 points
    // with it (they keep the two-exchange form), the persistent f64 lines of 16384 points need its registers (49 -> 55 %)
    // two-pass lines (the same ownership: COLS = T CPT there too): f32 + 0.3 - 0.6 points, f64 no gain (tools/r03_call_r.sh)
    constexpr bool POST_ONCE = TWO ? (sizeof(R) == 4 && B > 1) : (sizeof(R) == 4 || B == 16);
#endif
    {
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) v[j1] = load_elem(T * j1);                           // z[T j1 + t]
    }
    do {
    const long long next_line = line0 + (long long) gridDim.x;
    const bool more = PIPE && next_line < n_lines;                  // uniform
    const __amdgpu_buffer_rsrc_t rnext = in_rsrc(more ? next_line : line0);
    if constexpr (PIPE) {
        // Everything derived from the thread id is loop invariant: hipcc would hoist it out of the persistent loop (dozens of LDS and
        // buffer addresses) and spill it.  An opaque copy per line keeps the address arithmetic next to its use.
        asm volatile("" : "+v"(t));
        vin = g * pitch_b + t * IB;
        vout = (g * out_pitch + t) * CB;
    }
    if constexpr (MODE == DSC_MODE_C2R_PACKED && PRE_ONCE) {
        // Inverse packed-real pre-pass (dsc_fft.h:194-228) with the ownership the forward post-pass uses: the thread loaded bins t + T j,
        // its lower sixteen k and sixteen upper ones, and the partners L - k of its lower bins are upper bins of thread T - t.  The upper
        // halves go to the staging plane (slot of bin b: b - L/2; x plane [0, L/2), y plane [L/2, L)), one barrier, then per pair
        //   a = Y[k], b = Y[L-k], s = a + conj b, d = a - conj b, wq = (i/2) conj(W_2L^k):  Z[k] = s/2 + wq d (kept),  Z[L-k] = conj(s/2 - wq d)
        // and Z[L-k] goes back into the slot b came from (this thread is its only reader: no barrier); second barrier, everybody collects
        // its upper half.  Against "everybody computes its own 32 bins from two exchanged planes": half the arithmetic, no second
        // 32-value array, two barriers instead of three.
        const C wbase = tw_real[t];
        C yl = C{(R) 0, (R) 0};
        if (t == 0) { yl = load_elem(L); v[0].y = (R) 0; yl.y = (R) 0; }   // dsc_fft.h:227-228: real parts only at k = 0, L
        R *own_x = stage + t, *own_y = stage + L / 2 + t;                 // [T (j - 16)]: slot of the own upper bin t + T j
        R *par_x = stage + (L / 2 - 15 * T) - t;                          // [T (15 - j)]: slot of L - k, k = t + T j
        R *par_y = par_x + L / 2;
#pragma unroll
        for (int j = 16; j < 32; ++j) { own_x[T * (j - 16)] = v[j].x; own_y[T * (j - 16)] = v[j].y; }
        if (t == 0) own_y[0] = -v[16].y;                                  // bin L/2 pairs with itself: Z[L/2] = conj Y[L/2], nobody's partner
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const C a = v[j];
            C b = C{par_x[T * (15 - j)], par_y[T * (15 - j)]};
            if (j == 0 && t == 0) b = yl;                                 // Y[L] (the slot read is somebody else's: ignored)
            const C w = cmul(wbase, C{(R) root64_re(j), (R) root64_im(j)});      // W_2L^{t + T j} = W_2L^t W_64^j
            const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;
            const R sx = (R) 0.5 * (a.x + b.x), sy = (R) 0.5 * (a.y - b.y), dx = a.x - b.x, dy = a.y + b.y;
            const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
            v[j] = C{sx + wdx, sy + wdy};
            if (!(j == 0 && t == 0)) {                                    // Z[L] is no bin
                par_x[T * (15 - j)] = sx - wdx;
                par_y[T * (15 - j)] = wdy - sy;
            }
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[16 + j] = C{own_x[T * j], own_y[T * j]};
    }
    if constexpr (MODE == DSC_MODE_C2R_PACKED && !PRE_ONCE) {
        // Z[k] = (a + conj b)/2 + wq (a - conj b), a = Y[k], b = Y[L-k], wq = (i/2) conj(W_2L^k), for the
        // thread's own k = T j1 + t; b comes through the staging plane, one component at a time.
        const C wbase = tw_real[t];
        C yl = C{(R) 0, (R) 0};
        if (t == 0) { yl = load_elem(L); v[0].y = (R) 0; yl.y = (R) 0; }   // dsc_fft.h:227-228: real parts only at k = 0
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

    mid_passes<R, B, TWO, INV>(v, plane, wtab, tw_full, g, t, tid);

    if constexpr (MODE != DSC_MODE_R2C_PACKED) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
#pragma unroll
            for (int p = 0; p < B; ++p) {
                const C r = v[i * B + p];
                buf_store<STOREP>(C{r.x * scale, r.y * scale}, rout, vout, (T * i + COLS * brev(p, LOGB)) * CB);
            }
            if constexpr (PIPE) {                                  // the next line's elements into the registers just stored
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
#pragma unroll
                    for (int p = 0; p < B; ++p) v[i * B + p] = load_from(rnext, T * (i * B + p));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if constexpr (POST_ONCE) {
        // packed-real post-pass (dsc_fft.h:199-225), three-pass lines, round 3.  The column layout leaves thread t with the bins
        // t + T m, m = 0 .. 31 (COLS = T CPT: m = i + CPT brev(p)) — its sixteen lower bins k AND sixteen upper ones, and the partners
        // L - k of its lower bins are the UPPER bins of thread T - t.  So only the upper halves travel: both components at once (L reals:
        // the plane holds them), ONE barrier, half the LDS traffic of "everybody stages all 32 bins, twice", and no second 32-value array.
        //   a = Z[k] (own register), b = Z[L-k]:  s = a + conj b, d = a - conj b, wq = -(i/2) W_2L^k:  X[k] = s/2 + wq d,  X[L-k] = conj(s/2 - wq d)
        const C wbase = tw_real[t];
        auto reg_of = [](int m) constexpr { return (m % CPT) * B + brev(m / CPT, LOGB); };      // register that holds bin t + T m
        R *wr_x = stage + t, *wr_y = stage + L / 2 + t;                 // [T (m - 16)] = bin t + T m - L/2
        const R *rd_x = stage + (L / 2 - 15 * T) - t;                   // [T (15 - m)] = bin (L - k) - L/2, k = t + T m
        const R *rd_y = rd_x + L / 2;
#pragma unroll
        for (int m = 16; m < 32; ++m) { wr_x[T * (m - 16)] = v[reg_of(m)].x; wr_y[T * (m - 16)] = v[reg_of(m)].y; }
        const C zmid = v[reg_of(16)];                                   // thread 0: bin L/2, which pairs with itself
        if constexpr (PIPE) {                                           // the upper half is dead: the next line's elements into its registers
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int m = 16; m < 32; ++m) v[reg_of(m)] = load_from(rnext, T * reg_of(m));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
        const int dn_voff = (g * out_pitch + (L - 15 * T) - t) * CB;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const C a = v[reg_of(m)];
            C b = C{rd_x[T * (15 - m)], rd_y[T * (15 - m)]};
            if (m == 0 && t == 0) b = a;                                  // Z[L] := Z[0]
            const C w = cmul(wbase, C{(R) root64_re(m), (R) root64_im(m)});        // W_2L^{t + T m} = W_2L^t W_64^m
            const R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;
            const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            const R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
            C xk = C{(R) 0.5 * sx + wdx, (R) 0.5 * sy + wdy};
            C xm = C{(R) 0.5 * sx - wdx, wdy - (R) 0.5 * sy};
            if (m == 0 && t == 0) { xk.y = (R) 0; xm.y = (R) 0; }           // dsc_fft.h:221-225 stores exact zeros
            buf_store<STOREP>(C{xk.x * scale, xk.y * scale}, rout, vout, T * m * CB);
            buf_store<STOREP>(C{xm.x * scale, xm.y * scale}, rout, dn_voff, T * (15 - m) * CB);
            if constexpr (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) v[reg_of(m)] = load_from(rnext, T * reg_of(m));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t == 0) buf_store<STOREP>(C{zmid.x * scale, -zmid.y * scale}, rout, vout, (L / 2) * CB);    // k = L/2: a = b, W_2L^{L/2} = -i
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
        if constexpr (PIPE) {                                      // v is dead from here: the next line travels under the post-pass,
            __builtin_amdgcn_sched_barrier(0);                     // its first half now, the rest as ax / bx free their registers
            if (more) {
#pragma unroll
                for (int j1 = 0; j1 < 16; ++j1) v[j1] = load_from(rnext, T * j1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
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
            buf_store<STOREP>(C{xk.x * scale, xk.y * scale}, rout, vout, T * i * CB);
            buf_store<STOREP>(C{xm.x * scale, xm.y * scale}, rout, dn_voff, T * (15 - i) * CB);
            if constexpr (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) v[16 + i] = load_from(rnext, T * (16 + i));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (t == 0) {                                                     // k = L/2: a = b, W_2L^{L/2} = -i
            const R ay = stage[L / 2];
            buf_store<STOREP>(C{amx * scale, -ay * scale}, rout, vout, (L / 2) * CB);
        }
    }
    if constexpr (PIPE) {
        if (!more) break;
        line0 = next_line;
        rin = rnext;
        rout = __builtin_amdgcn_make_buffer_rsrc((void *) (out + line0 * out_pitch), 0, out_pitch * CB, 0x00020000);
        // no barrier here: the passes end with one (the inverse pre-pass may write the staging plane at once), and the forward
        // post-pass's staging reads are followed by the barrier in front of the next line's passes
    }
    } while (PIPE);
}

// ------------------------------------------------------------------------------------------------
// Fused README filterFFT (README.md:113-135) at the mid sizes: y = irfft(rfft(s, 2L) * H), H [L + 1] bins shared by all
// rows.  Forward passes, packed-real pass on the pair (k, L-k), times (H[k], H[L-k]), inverse pre-pass on the same pair
// in the same thread, one staging round trip to bring the pairs back to the load layout, inverse passes: the spectrum
// never leaves the CU.  s rows may be shorter than 2L (zero padded) or longer (cropped).
template<typename R, int B, bool TWO>
__global__ __launch_bounds__((mid_cfg<R, B, TWO, 1>::NT), (mid_cfg<R, B, TWO, 1>::WAVES_PER_EU)) void fft_mid_filter_kernel(
    const R *__restrict__ s, const cpx<R> *__restrict__ H, cpx<R> *__restrict__ y, long long n_lines, const cpx<R> *__restrict__ tw_full,
    const cpx<R> *__restrict__ tw_real, int in_pitch_b, int in_len_b) {
    using C = cpx<R>;
    using cfg = mid_cfg<R, B, TWO, 1>;
    constexpr int T = cfg::T, L = cfg::L, G = cfg::G, NT = cfg::NT, SP = cfg::SP, CPT = cfg::CPT, COLS = cfg::COLS;
    constexpr int LOGB = ilog2(B), CB = (int) sizeof(C);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R *plane = (R *) lds_raw;
    C *wtab = (C *) (plane + cfg::PLANE);
    const int tid = threadIdx.x;
    const int g = T >= 64 ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;
    const int t = tid - g * T;
    const long long line0 = (long long) blockIdx.x * G;
    const long long left = n_lines - line0;
    const int n_valid = left < G ? (int) left : G;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) s + line0 * in_pitch_b), 0,
                                                                         (n_valid - 1) * in_pitch_b + in_len_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (y + line0 * L), 0, n_valid * L * CB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void *) H, 0, (L + 1) * CB, 0x00020000);
    const int vin = g * in_pitch_b + t * CB, vout = (g * L + t) * CB;
    R *stage = plane + g * SP;
    for (int i = tid; i < cfg::TABLE; i += NT) wtab[i] = tw_full[(long long) i * cfg::TABLE_STRIDE];

    C v[32];
#pragma unroll
    for (int j1 = 0; j1 < 32; ++j1) {                                     // sample pairs past the valid length read zero
        const int eoff = (T * j1 + t) * CB;
        v[j1] = buf_load<kCached>(rin, eoff < in_len_b ? vin : 0x7f000000, T * j1 * CB, R{});
        if (eoff + CB > in_len_b) v[j1].y = (R) 0;
    }
    __syncthreads();
    mid_passes<R, B, TWO, false, 1>(v, plane, wtab, tw_full, g, t, tid);  // v[i B + p] = Z[(t + T i) + COLS brev(p)]

#ifdef DSC_MID_OLD_FILTER_PAIRS
    constexpr bool ONCE = false;
#else
    // two-pass lines, measured (tools/r03_call_r.sh): f32 N = 512 / 1024 / 2048: 48.6 -> 50.2, 50.2 -> 54.1, 44.9 -> 49.3 %; f64 N = 512
    // 57.7 -> 60.1 %, but N = 1024 and 2048 LOSE with it (59.7 -> 49.1, 56.8 -> 50.6 %: the second 32-value array costs their occupancy)
    constexpr bool ONCE = !TWO || sizeof(R) == 4 || B == 8;
#endif
    if constexpr (ONCE) {
        // ---- three-pass lines, round 3.  The column layout leaves thread t with the bins t + T m, m = 0 .. 31: its sixteen lower bins k
        // and sixteen upper ones; the partners L - k of its lower bins are upper bins of thread T - t.  Only the upper halves travel, both
        // components at once (slot of upper bin b: b - L/2; x plane [0, L/2), y plane [L/2, L)): write them, ONE barrier, then per pair
        // read b = Z[L-k], packed-real pass, times (H[k], H[L-k]), inverse pre-pass, keep Z'[k] — the inverse passes want it in register
        // m, the load layout — and put Z'[L-k] back into the slot it came from (read by this thread only: no barrier); second barrier,
        // everybody collects its upper half.  Half the LDS traffic and two barriers instead of five (the form below).
        const C wbase = tw_real[t];
        auto reg_of = [](int m) constexpr { return (m % CPT) * B + brev(m / CPT, LOGB); };      // register that holds bin t + T m
        R *own_x = stage + t, *own_y = stage + L / 2 + t;                 // [T (m - 16)]: slot of the own upper bin t + T m
        R *par_x = stage + (L / 2 - 15 * T) - t;                          // [T (15 - m)]: slot of L - k, k = t + T m
        R *par_y = par_x + L / 2;
#pragma unroll
        for (int m = 16; m < 32; ++m) { own_x[T * (m - 16)] = v[reg_of(m)].x; own_y[T * (m - 16)] = v[reg_of(m)].y; }
        const C zm0 = v[reg_of(16)];                                      // thread 0: bin L/2, which pairs with itself
        lds_barrier();
        C z[32];
        const int hm_voff = ((L - 15 * T) - t) * CB;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const C za = v[reg_of(m)];
            C zb = C{par_x[T * (15 - m)], par_y[T * (15 - m)]};
            if (m == 0 && t == 0) zb = za;                                // Z[L] := Z[0] (the slot read is somebody else's: ignored)
            const C w = cmul(wbase, C{(R) root64_re(m), (R) root64_im(m)});            // W_2L^{t + T m}
            // forward: X[k] = s/2 + wq d, X[L-k] = conj(s/2 - wq d), wq = -(i/2) w          (dsc_fft.h:199-225)
            R sx = za.x + zb.x, sy = za.y - zb.y, dx = za.x - zb.x, dy = za.y + zb.y;
            R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;
            R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
            C xk = C{(R) 0.5 * sx + wdx, (R) 0.5 * sy + wdy};
            C xm = C{(R) 0.5 * sx - wdx, wdy - (R) 0.5 * sy};
            if (m == 0 && t == 0) { xk.y = (R) 0; xm.y = (R) 0; }
            C a = cmul(xk, buf_load<kCached>(rh, t * CB, T * m * CB, R{}));
            C b = cmul(xm, buf_load<kCached>(rh, hm_voff, T * (15 - m) * CB, R{}));
            if (m == 0 && t == 0) { a.y = (R) 0; b.y = (R) 0; }                         // dsc_fft.h:227-228: real parts only at k = 0, L
            // inverse: Z'[k] = s/2 + wq' d, Z'[L-k] = conj(s/2 - wq' d), wq' = (i/2) conj(w)  (dsc_fft.h:194-228)
            sx = a.x + b.x; sy = a.y - b.y; dx = a.x - b.x; dy = a.y + b.y;
            wqx = (R) 0.5 * w.y; wqy = (R) 0.5 * w.x;
            wdx = dx * wqx - dy * wqy; wdy = dx * wqy + dy * wqx;
            z[m] = C{(R) 0.5 * sx + wdx, (R) 0.5 * sy + wdy};
            if (!(m == 0 && t == 0)) {                                    // Z'[L] is no bin
                par_x[T * (15 - m)] = (R) 0.5 * sx - wdx;
                par_y[T * (15 - m)] = wdy - (R) 0.5 * sy;
            }
        }
        if (t == 0) {                                                     // k = L/2: X = conj Z, then Z' = conj(X H): slot 0, nobody's partner
            const C ym = cmul(C{zm0.x, -zm0.y}, buf_load<kCached>(rh, (L / 2) * CB, 0, R{}));
            stage[0] = ym.x;
            stage[L / 2] = -ym.y;
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) z[16 + j] = C{own_x[T * j], own_y[T * j]};
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = z[j];                         // v[j1] = Z'[T j1 + t]
    } else {
        // ---- the pair (k, L-k), k = t + T i, i < 16 (plus k = L/2 in thread 0): a = Z[k], b = Z[L-k] through the staging plane
        const C wbase = tw_real[t];
        R ax[16], bx[16], amx = (R) 0;
        R *up = stage + t;                          // up[T i]        = stage[k]
        R *dn = stage + (L - 15 * T) - t;           // dn[T (15 - i)] = stage[L - k]
    #pragma unroll
        for (int i = 0; i < CPT; ++i)
    #pragma unroll
            for (int p = 0; p < B; ++p) up[T * i + COLS * brev(p, LOGB)] = v[i * B + p].x;
        if (t == 0) stage[L] = v[0].x;
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
        // Every staging slot is read by exactly one thread — the one that owns the pair — so the real parts of the results go
        // back into the same slots at once (no barrier, and only the imaginary parts stay in registers).
        R zky[16], zmy[16];
        C zmid = C{(R) 0, (R) 0};
        const int hm_voff = ((L - 15 * T) - t) * CB;
    #pragma unroll
        for (int i = 0; i < 16; ++i) {
            const R ay = up[T * i], by = dn[T * (15 - i)];
            const C w = cmul(wbase, C{(R) root64_re(i), (R) root64_im(i)});            // W_2L^{t + T i}
            // forward: X[k] = s/2 + wq d, X[L-k] = conj(s/2 - wq d), wq = -(i/2) w          (dsc_fft.h:199-225)
            R sx = ax[i] + bx[i], sy = ay - by, dx = ax[i] - bx[i], dy = ay + by;
            R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;
            R wdx = dx * wqx - dy * wqy, wdy = dx * wqy + dy * wqx;
            C xk = C{(R) 0.5 * sx + wdx, (R) 0.5 * sy + wdy};
            C xm = C{(R) 0.5 * sx - wdx, wdy - (R) 0.5 * sy};
            if (i == 0 && t == 0) { xk.y = (R) 0; xm.y = (R) 0; }
            // times the filter
            C a = cmul(xk, buf_load<kCached>(rh, t * CB, T * i * CB, R{}));
            C b = cmul(xm, buf_load<kCached>(rh, hm_voff, T * (15 - i) * CB, R{}));
            if (i == 0 && t == 0) { a.y = (R) 0; b.y = (R) 0; }                         // dsc_fft.h:227-228: real parts only at k = 0, L
            // inverse: Z'[k] = s/2 + wq' d, Z'[L-k] = conj(s/2 - wq' d), wq' = (i/2) conj(w)  (dsc_fft.h:194-228)
            sx = a.x + b.x; sy = a.y - b.y; dx = a.x - b.x; dy = a.y + b.y;
            wqx = (R) 0.5 * w.y; wqy = (R) 0.5 * w.x;
            wdx = dx * wqx - dy * wqy; wdy = dx * wqy + dy * wqx;
            up[T * i] = (R) 0.5 * sx + wdx;                                             // Re Z'[k]
            dn[T * (15 - i)] = (R) 0.5 * sx - wdx;                                      // Re Z'[L-k]
            zky[i] = (R) 0.5 * sy + wdy;
            zmy[i] = wdy - (R) 0.5 * sy;
        }
        if (t == 0) {                                                         // k = L/2: X = conj Z, then Z' = conj(X H); after the loop: its
            const R ay = stage[L / 2];                                        // slot is nobody's pair (T * 16 = L/2 belongs to i = 16)
            const C ym = cmul(C{amx, -ay}, buf_load<kCached>(rh, (L / 2) * CB, 0, R{}));
            zmid = C{ym.x, -ym.y};
            stage[L / 2] = zmid.x;
        }
        lds_barrier();
        // ---- back to the load layout of the inverse transform: v[j1] = Z'[T j1 + t]
    #pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) v[j1].x = up[T * j1];
        lds_barrier();
    #pragma unroll
        for (int i = 0; i < 16; ++i) { up[T * i] = zky[i]; dn[T * (15 - i)] = zmy[i]; }
        if (t == 0) stage[L / 2] = zmid.y;
        lds_barrier();
    #pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) v[j1].y = up[T * j1];
        lds_barrier();

    }

    // an opaque copy of the table pointer: otherwise the 62 twiddles the forward passes read from the table are kept in
    // registers for the inverse passes (common subexpressions) and the kernel needs 220 VGPRs
    const C *wtab_inv = wtab;
    asm volatile("" : "+v"(wtab_inv));
    mid_passes<R, B, TWO, true, 1>(v, plane, wtab_inv, tw_full, g, t, tid);
    const R scale = (R) (1.0 / (double) L);                               // 2/(2n), dsc_fft.h:232
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int p = 0; p < B; ++p) {
            const C r = v[i * B + p];
            buf_store<kStream>(C{r.x * scale, r.y * scale}, rout, vout, (T * i + COLS * brev(p, LOGB)) * CB);
        }
}

template<typename R, int B, bool TWO>
void launch_filter(const void *s, const void *H, void *y, long long n_lines, const void *tw_full, const void *tw_real, int in_pitch_b,
                   int in_len_b, hipStream_t stream) {
    using cfg = mid_cfg<R, B, TWO, 1>;
    constexpr size_t lds = mid_lds_bytes<R, B, TWO, 1>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_mid_filter_kernel<R, B, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
    const long long groups = (n_lines + cfg::G - 1) / cfg::G;
    DSC_LAUNCH((fft_mid_filter_kernel<R, B, TWO>), dim3((unsigned) groups), dim3(cfg::NT), lds, stream, (const R *) s,
                       (const cpx<R> *) H, (cpx<R> *) y, n_lines, (const cpx<R> *) tw_full, (const cpx<R> *) tw_real, in_pitch_b, in_len_b);
}

template<typename R>
void launch_filter_len(int L, const void *s, const void *H, void *y, long long n_lines, const void *tw_full, const void *tw_real,
                       int pb, int lb, hipStream_t stream) {
    switch (L) {
        case 256:   launch_filter<R, 8, true>(s, H, y, n_lines, tw_full, tw_real, pb, lb, stream); break;
        case 512:   launch_filter<R, 16, true>(s, H, y, n_lines, tw_full, tw_real, pb, lb, stream); break;
        case 1024:  launch_filter<R, 32, true>(s, H, y, n_lines, tw_full, tw_real, pb, lb, stream); break;
        case 2048:  launch_filter<R, 2, false>(s, H, y, n_lines, tw_full, tw_real, pb, lb, stream); break;
        case 4096:  launch_filter<R, 4, false>(s, H, y, n_lines, tw_full, tw_real, pb, lb, stream); break;
        case 8192:  launch_filter<R, 8, false>(s, H, y, n_lines, tw_full, tw_real, pb, lb, stream); break;
        default:    launch_filter<R, 16, false>(s, H, y, n_lines, tw_full, tw_real, pb, lb, stream); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Complex lengths 32 .. 256 (B = 1, 2, 4, 8 threads per line, real lengths 64 .. 512).  Lines this short cannot be read
// coalesced in the "thread t owns elements B j1 + t" pattern, so the group's contiguous block of lines is copied to LDS
// first (flat, 8/16 B per lane) and picked up from there; results take the same way back, and the packed-real passes
// work on the staged lines (one output bin per thread and step, both partners read from LDS).  Row pitch 33 B complex:
// every LDS access of the kernel is conflict free.
// group size of the LDS-staged kernel (lines of 32 .. 256 points), round 3 (gpurun_out/r3m/small_nt.txt, profiles/r03_bench_mid.txt): f32 in groups
// of 128 threads instead of 256: rfft / irfft / fft at N = 128 69.3 / 67.1 / 63.8 -> 72.7 / 70.5 / 67.7 %, N = 256 70.3 / 68.5 / 65.1 -> 72.4 / 70.7 /
// 67.6 %, N = 512 +1 .. 3, N = 64 +0 .. 3 (512 threads: the real transforms fall to 57 %).  f64 stays at 128 (64: irfft N = 512 62.7 -> 51.9 %).
#ifndef DSC_SMALL_NT_F32
#define DSC_SMALL_NT_F32 128
#endif
#ifndef DSC_SMALL_NT_F64
#define DSC_SMALL_NT_F64 128
#endif
// f64 COMPLEX lines in groups of 64 threads (fft c64 L = 32 / 128 / 256: 65.5 / 64.1 / 65.0 -> 66.6 / 66.7 / 74.2 %); the f64 real transforms lose with
// it (irfft N = 512 62.7 -> 51.9 %) and keep 128
#ifndef DSC_SMALL_NT_F64_COMPLEX
#define DSC_SMALL_NT_F64_COMPLEX 64
#endif
template<typename R, int MODE> struct small_cfg {
    static constexpr int NT = sizeof(R) == 8 ? (MODE == DSC_MODE_C2C ? DSC_SMALL_NT_F64_COMPLEX : DSC_SMALL_NT_F64) : DSC_SMALL_NT_F32;
};
template<typename R, int B, int MODE> constexpr size_t small_lds_bytes() { return ((size_t) (small_cfg<R, MODE>::NT / B) * 33 * B + 32 * B + 32 * B + 2) * 2 * sizeof(R); }

// PAD: the input lines have a pitch of in_pitch BYTES of which in_len BYTES are valid (a packed-real line may hold an odd number of
// samples: the pair that straddles its end keeps the first sample only); the rest of the transform length reads as zero
// (zero padding / cropping through n=, dsc.cpp:1990-1998, 2125-2133, 2149-2157) — e.g. frames of 200 samples transformed at 256.
template<typename R, int B, int MODE, bool INV, bool PAD>
__global__ __launch_bounds__((small_cfg<R, MODE>::NT)) void fft_small_kernel(const void *__restrict__ in, void *__restrict__ out, long long n_lines,
                                                                     const cpx<R> *__restrict__ tw_full, const cpx<R> *__restrict__ tw_real,
                                                                     R scale, int in_pitch, int in_len) {
    using C = cpx<R>;
    constexpr int NT = small_cfg<R, MODE>::NT, L = 32 * B, G = NT / B, P = 33 * B, LOGB = ilog2(B);
    constexpr bool REAL_IN = MODE == DSC_MODE_R2C_CAST;
    constexpr int IN_PITCH = MODE == DSC_MODE_C2R_PACKED ? L + 1 : L, OUT_PITCH = MODE == DSC_MODE_R2C_PACKED ? L + 1 : L;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    C *stage = (C *) lds_raw;
    R *plane = (R *) lds_raw;                       // the exchange plane of mid_passes reuses the staging area
    C *wtab = stage + G * P;                        // W_L^m, m < L
    C *wreal = wtab + L;                            // W_2L^k, k <= L
    const int tid = threadIdx.x;
    const int g = tid / B, t = tid % B;
    const long long line0 = (long long) blockIdx.x * G;
    const long long left = n_lines - line0;
    const int n_valid = left < G ? (int) left : G;

    for (int i = tid; i < L; i += NT) wtab[i] = tw_full[i];
    if constexpr (MODE == DSC_MODE_R2C_PACKED || MODE == DSC_MODE_C2R_PACKED)
        for (int i = tid; i <= L; i += NT) wreal[i] = tw_real[i];
    // ---- stage in: the group's block of lines, flat.  All loads are issued before the first LDS write; element
    // e = tid + NT m of the block is (line, j) with e = line IN_PITCH + j, advanced incrementally (one division in all).
    // Lines past the end of the batch read zeros (descriptor range).
    constexpr int CB = (int) sizeof(C), EB = REAL_IN ? (int) sizeof(R) : CB;
    constexpr int STEPS_IN = (G * IN_PITCH + NT - 1) / NT;                  // 32, or 33 for rows of L + 1 bins
    const long long gpitch_b = PAD ? (long long) in_pitch : (long long) IN_PITCH * EB;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) in + line0 * gpitch_b), 0,
                                                                         (int) (PAD ? (n_valid - 1) * gpitch_b + in_len : n_valid * gpitch_b), 0x00020000);
    {
        C tmp[STEPS_IN];
        if constexpr (PAD) {                                              // element (line, j) of the LDS image <- line * in_pitch + j, or zero
            int line = tid / IN_PITCH, j = tid % IN_PITCH;
#pragma unroll
            for (int m = 0; m < STEPS_IN; ++m) {
                const int voff = j * EB < in_len ? line * in_pitch + j * EB : 0x7f000000;
                if constexpr (REAL_IN) tmp[m] = buf_load_real<kCached>(rin, voff, 0, R{});
                else                   tmp[m] = buf_load<kCached>(rin, voff, 0, R{});
                if (!REAL_IN && j * EB + EB > in_len) tmp[m].y = (R) 0;
                j += NT % IN_PITCH;
                line += NT / IN_PITCH;
                if (j >= IN_PITCH) { j -= IN_PITCH; ++line; }
            }
        } else {
#pragma unroll
        for (int m = 0; m < STEPS_IN; ++m) {
            if constexpr (REAL_IN) tmp[m] = buf_load_real<kStream>(rin, tid * EB, m * NT * EB, R{});
            else                   tmp[m] = buf_load<kCached>(rin, tid * EB, m * NT * EB, R{});
        }
        }
        int line = tid / IN_PITCH, j = tid % IN_PITCH;
#pragma unroll
        for (int m = 0; m < STEPS_IN; ++m) {
            if (m * NT + NT <= G * IN_PITCH || tid + m * NT < G * IN_PITCH) stage[line * P + j] = tmp[m];
            j += NT % IN_PITCH;
            line += NT / IN_PITCH;
            if (j >= IN_PITCH) { j -= IN_PITCH; ++line; }
        }
    }
    __syncthreads();
    C v[32];
    const C *mine = stage + g * P;
    if constexpr (MODE == DSC_MODE_C2R_PACKED) {    // pre-pass (dsc_fft.h:194-228) for the thread's own bins k = B j1 + t
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) {
            const int k = B * j1 + t;
            C a = mine[k], b = mine[L - k];
            if (k == 0) { a.y = (R) 0; b.y = (R) 0; }
            const C w = wreal[k];
            const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;
            const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            v[j1] = C{(R) 0.5 * sx + (dx * wqx - dy * wqy), (R) 0.5 * sy + (dx * wqy + dy * wqx)};
        }
    } else {
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1) v[j1] = mine[B * j1 + t];
    }
    __syncthreads();                                // staged lines consumed: the area becomes the exchange plane

    mid_passes<R, B, true, INV>(v, plane, wtab, tw_full, g, t, tid);

    // ---- results back to the staging area in natural order: v[i B + p] = bin (t + B i) + 32 brev(p)
    C *mine_w = stage + g * P;
    const R sc = MODE == DSC_MODE_R2C_PACKED ? (R) 1 : scale;
#pragma unroll
    for (int i = 0; i < 32 / B; ++i)
#pragma unroll
        for (int p = 0; p < B; ++p) {
            const C r = v[i * B + p];
            mine_w[(t + B * i) + 32 * brev(p, LOGB)] = C{r.x * sc, r.y * sc};
        }
    __syncthreads();
    // ---- stage out, flat (stores past the end of the batch are dropped by the descriptor range)
    constexpr int STEPS_OUT = (G * OUT_PITCH + NT - 1) / NT;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) ((char *) out + line0 * OUT_PITCH * CB), 0,
                                                                          n_valid * OUT_PITCH * CB, 0x00020000);
    int line = tid / OUT_PITCH, k = tid % OUT_PITCH;
#pragma unroll
    for (int m = 0; m < STEPS_OUT; ++m) {
        const bool inside = m * NT + NT <= G * OUT_PITCH || tid + m * NT < G * OUT_PITCH;
        C x;
        if constexpr (MODE == DSC_MODE_R2C_PACKED) {                      // dsc_fft.h:199-225, one bin per thread and step
            const C *row = stage + (inside ? line : 0) * P;
            const C a = row[k == L ? 0 : k], b = row[k == 0 || k == L ? 0 : L - k];
            const C w = wreal[k];
            const R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;
            const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            x = C{((R) 0.5 * sx + (dx * wqx - dy * wqy)) * scale, ((R) 0.5 * sy + (dx * wqy + dy * wqx)) * scale};
            if (k == 0 || k == L) x.y = (R) 0;
        } else {
            x = stage[(inside ? line : 0) * P + k];
        }
        buf_store<kCached>(x, rout, inside ? tid * CB : 0x7f000000, m * NT * CB);
        k += NT % OUT_PITCH;
        line += NT / OUT_PITCH;
        if (k >= OUT_PITCH) { k -= OUT_PITCH; ++line; }
    }
}

template<typename R, int B, int MODE, bool INV, bool PAD>
void launch_small_pad(const void *in, void *out, long long n_lines, const void *tw_full, const void *tw_real, double scale, int in_pitch, int in_len,
                      hipStream_t stream) {
    constexpr int G = small_cfg<R, MODE>::NT / B;
    constexpr size_t lds = small_lds_bytes<R, B, MODE>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_small_kernel<R, B, MODE, INV, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
    const long long groups = (n_lines + G - 1) / G;
    DSC_LAUNCH((fft_small_kernel<R, B, MODE, INV, PAD>), dim3((unsigned) groups), dim3(small_cfg<R, MODE>::NT), lds, stream, in, out, n_lines,
                       (const cpx<R> *) tw_full, (const cpx<R> *) tw_real, (R) scale, in_pitch, in_len);
}
// in_pitch < 0: full contiguous lines
template<typename R, int B, int MODE, bool INV>
void launch_small_one(const void *in, void *out, long long n_lines, const void *tw_full, const void *tw_real, double scale, long long in_pitch, int in_len,
                      hipStream_t stream) {
    if (in_pitch < 0) launch_small_pad<R, B, MODE, INV, false>(in, out, n_lines, tw_full, tw_real, scale, 0, 0, stream);
    else              launch_small_pad<R, B, MODE, INV, true>(in, out, n_lines, tw_full, tw_real, scale, (int) in_pitch, in_len, stream);
}

template<typename R, int B>
void launch_small(const void *in, void *out, long long n_lines, dsc_fft_mode mode, bool inverse, const void *tw_full, const void *tw_real,
                  double scale, long long in_pitch, int in_len, hipStream_t stream) {
    if (mode == DSC_MODE_R2C_PACKED)      launch_small_one<R, B, DSC_MODE_R2C_PACKED, false>(in, out, n_lines, tw_full, tw_real, scale, in_pitch, in_len, stream);
    else if (mode == DSC_MODE_C2R_PACKED) launch_small_one<R, B, DSC_MODE_C2R_PACKED, true>(in, out, n_lines, tw_full, tw_real, scale, in_pitch, in_len, stream);
    else if (mode == DSC_MODE_R2C_CAST && !inverse) launch_small_one<R, B, DSC_MODE_R2C_CAST, false>(in, out, n_lines, tw_full, tw_real, scale, in_pitch, in_len, stream);
    else if (mode == DSC_MODE_R2C_CAST)   launch_small_one<R, B, DSC_MODE_R2C_CAST, true>(in, out, n_lines, tw_full, tw_real, scale, in_pitch, in_len, stream);
    else if (inverse)                     launch_small_one<R, B, DSC_MODE_C2C, true>(in, out, n_lines, tw_full, tw_real, scale, in_pitch, in_len, stream);
    else                                  launch_small_one<R, B, DSC_MODE_C2C, false>(in, out, n_lines, tw_full, tw_real, scale, in_pitch, in_len, stream);
}

template<typename R, int B, bool TWO, int MODE, bool INV, bool PAD>
void launch_pad(const void *in, void *out, long long n_lines, const void *tw_full, const void *tw_real, double scale, int in_pitch_b,
                int in_len_b, hipStream_t stream) {
    using cfg = mid_cfg<R, B, TWO>;
    constexpr size_t lds = mid_lds_bytes<R, B, TWO>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_mid_kernel<R, B, TWO, MODE, INV, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
    long long groups = (n_lines + cfg::G - 1) / cfg::G;
    if (cfg::PIPE) {                                        // persistent: one workgroup per CU walks the lines
        static int cus[64];
        int dev = 0;
        DSC_KERNEL_CHECK(hipGetDevice(&dev));
        dev &= 63;
        if (cus[dev] == 0) DSC_KERNEL_CHECK(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
        if (groups > cus[dev]) groups = cus[dev];
    }
    DSC_LAUNCH((fft_mid_kernel<R, B, TWO, MODE, INV, PAD>), dim3((unsigned) groups), dim3(cfg::NT), lds, stream, (const cpx<R> *) in,
                       (cpx<R> *) out, n_lines, (const cpx<R> *) tw_full, (const cpx<R> *) tw_real, (R) scale, in_pitch_b, in_len_b);
}

// in_pitch_b < 0: full contiguous lines (the fast instantiation)
template<typename R, int B, bool TWO, int MODE, bool INV>
void launch_one(const void *in, void *out, long long n_lines, const void *tw_full, const void *tw_real, double scale, int in_pitch_b,
                int in_len_b, hipStream_t stream) {
    if (in_pitch_b < 0) launch_pad<R, B, TWO, MODE, INV, false>(in, out, n_lines, tw_full, tw_real, scale, 0, 0, stream);
    else                launch_pad<R, B, TWO, MODE, INV, true>(in, out, n_lines, tw_full, tw_real, scale, in_pitch_b, in_len_b, stream);
}

template<typename R, int B, bool TWO>
void launch_b(const void *in, void *out, long long n_lines, dsc_fft_mode mode, bool inverse, const void *tw_full, const void *tw_real,
              double scale, int pb, int lb, hipStream_t stream) {
    if (mode == DSC_MODE_R2C_PACKED)      launch_one<R, B, TWO, DSC_MODE_R2C_PACKED, false>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
    else if (mode == DSC_MODE_C2R_PACKED) launch_one<R, B, TWO, DSC_MODE_C2R_PACKED, true>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
    else if (mode == DSC_MODE_R2C_CAST && !inverse) launch_one<R, B, TWO, DSC_MODE_R2C_CAST, false>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
    else if (mode == DSC_MODE_R2C_CAST)   launch_one<R, B, TWO, DSC_MODE_R2C_CAST, true>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
    else if (inverse)                     launch_one<R, B, TWO, DSC_MODE_C2C, true>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
    else                                  launch_one<R, B, TWO, DSC_MODE_C2C, false>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
}

}  // namespace

// lengths 32 .. 256: full contiguous lines only (zero-padded / cropped lines of 256 take the PAD instantiation of fft_mid_kernel)
bool dsc_fft_regs_small_supports(int L) { return L == 32 || L == 64 || L == 128 || L == 256; }   // measured: at 512 the direct-load kernel wins

bool dsc_fft_regs_mid_supports(int L, dsc_fft_mode mode, bool single_precision) {
    if (L == 32768) return single_precision && mode == DSC_MODE_C2C;   // the packed-real 65536-point f32 transforms have their own kernels
    return L == 256 || L == 512 || L == 1024 || L == 2048 || L == 4096 || L == 8192 || L == 16384;
}

template<typename R>
static void launch_len(const void *in, void *out, long long n_lines, int L, dsc_fft_mode mode, bool inverse, const void *tw_full,
                       const void *tw_real, double scale, int pb, int lb, hipStream_t stream) {
    switch (L) {
        case 256:   launch_b<R, 8, true>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, pb, lb, stream); break;
        case 512:   launch_b<R, 16, true>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, pb, lb, stream); break;
        case 1024:  launch_b<R, 32, true>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, pb, lb, stream); break;
        case 2048:  launch_b<R, 2, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, pb, lb, stream); break;
        case 4096:  launch_b<R, 4, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, pb, lb, stream); break;
        case 8192:  launch_b<R, 8, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, pb, lb, stream); break;
        default:    launch_b<R, 16, false>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, pb, lb, stream); break;
    }
}

// in_pitch / in_len: input line pitch and valid length in INPUT ELEMENTS (reals for R2C_PACKED / R2C_CAST, complex otherwise);
// in_pitch < 0 = full contiguous lines.
void dsc_launch_fft_regs_mid(const void *in, void *out, long long n_lines, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                             const void *tw_full, const void *tw_real, double scale, long long in_pitch, int in_len, hipStream_t stream) {
    if (n_lines <= 0) return;
    if (dsc_fft_regs_small_supports(L)) {                                   // 32 .. 256 points: the LDS-staged kernel
        const int eb = (single_precision ? 4 : 8) * ((mode == DSC_MODE_R2C_PACKED || mode == DSC_MODE_R2C_CAST) ? 1 : 2);
        if (in_pitch >= 0) { in_pitch *= eb; in_len *= eb; }                  // the kernel counts bytes
        if (single_precision) {
            if (L == 32)       launch_small<float, 1>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
            else if (L == 64)  launch_small<float, 2>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
            else if (L == 128) launch_small<float, 4>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
            else               launch_small<float, 8>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
        } else {
            if (L == 32)       launch_small<double, 1>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
            else if (L == 64)  launch_small<double, 2>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
            else if (L == 128) launch_small<double, 4>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
            else               launch_small<double, 8>(in, out, n_lines, mode, inverse, tw_full, tw_real, scale, in_pitch, in_len, stream);
        }
        return;
    }
    const int real_b = single_precision ? 4 : 8;
    const int elem_b = (mode == DSC_MODE_R2C_PACKED || mode == DSC_MODE_R2C_CAST) ? real_b : 2 * real_b;
    const int pb = in_pitch < 0 ? -1 : (int) (in_pitch * elem_b), lb = in_pitch < 0 ? 0 : in_len * elem_b;
    if (!single_precision) {
        launch_len<double>(in, out, n_lines, L, mode, inverse, tw_full, tw_real, scale, pb, lb, stream);
    } else if (L == 32768) {
        if (inverse) launch_one<float, 32, false, DSC_MODE_C2C, true>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
        else         launch_one<float, 32, false, DSC_MODE_C2C, false>(in, out, n_lines, tw_full, tw_real, scale, pb, lb, stream);
    } else {
        launch_len<float>(in, out, n_lines, L, mode, inverse, tw_full, tw_real, scale, pb, lb, stream);
    }
}

// y = irfft(rfft(s, 2L) * H) fused, L = 256 .. 16384: s = [n_lines][in_pitch] reals of which in_len <= 2L are used, H = [L + 1]
// bins, y = [n_lines][2L] reals
void dsc_launch_filter_regs_mid(const void *s, const void *H, void *y, long long n_lines, int L, bool single_precision, const void *tw_full,
                                const void *tw_real, long long in_pitch, int in_len, hipStream_t stream) {
    if (n_lines <= 0) return;
    const int rb = single_precision ? 4 : 8;
    if (single_precision) launch_filter_len<float>(L, s, H, y, n_lines, tw_full, tw_real, (int) (in_pitch * rb), in_len * rb, stream);
    else                  launch_filter_len<double>(L, s, H, y, n_lines, tw_full, tw_real, (int) (in_pitch * rb), in_len * rb, stream);
}
