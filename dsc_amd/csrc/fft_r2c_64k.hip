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

// Diagnostic build only (tools/probe64k.hip defines DSC_R2C64K_STAMPS): wave 0 of every
// workgroup records 100 MHz realtime stamps per phase into a side buffer.
#ifdef DSC_R2C64K_STAMPS
#define STAMP(i)                                                                              \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (stamps && threadIdx.x == 0 && it < 64)                                            \
            stamps[((size_t) blockIdx.x * 64 + it) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#else
#define STAMP(i)
#endif
#ifdef DSC_R2C64K_PROBE            // extra kernel arguments of the diagnostic builds
#define PROBE_ARGS , unsigned long long *stamps, int io_on
#define PROBE_NULL , nullptr, 1
#define IO_ON io_on                // 0: zero-record descriptors, every global access is dropped
#else
#define PROBE_ARGS
#define PROBE_NULL
#define IO_ON 1
#endif
#ifndef DSC_R2C64K_SKIP
#define DSC_R2C64K_SKIP 0
#endif
#define SKIP(bit) ((DSC_R2C64K_SKIP) & (bit))   // compile-time ablation: 1 dft32, 2 twiddles, 4 LDS traffic, 8 barriers, 16 post-pass

// Values derived from threadIdx are loop invariant; hipcc hoists every address built from
// them out of the persistent row loop (dozens of VGPRs) and then spills them.  Passing the
// thread id through an empty asm once per row keeps those computations inside the loop.
__device__ __forceinline__ int per_row(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// Thread id rebuilt from the wave number (an SGPR) and the hardware lane count, so that not
// even threadIdx.x itself has to stay in a VGPR across the row.
__device__ __forceinline__ int thread_id(int wave_sgpr) {
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));      // opaque 0: the two mbcnt below cannot be hoisted
    const int lane = __builtin_amdgcn_mbcnt_hi(-1, __builtin_amdgcn_mbcnt_lo(-1, zero));
    return (wave_sgpr << 6) | lane;
}

// ------------------------------------------------------------------------------------------
// forward: x [batch][65536] f32  ->  X [batch][32769] c32
__global__ __launch_bounds__(1024) void rfft64k_kernel(const float *__restrict__ x, f2 *__restrict__ X, int batch,
                                                       const f2 *__restrict__ aux PROBE_ARGS) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *plane = lds;
    f2 *w1024 = (f2 *) (lds + kPlaneFloats);

    {
        const int t0 = threadIdx.x;
        w1024[t0] = aux[kAuxW1024 + t0];
    }
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
            (void *) (x + (size_t) row0 * 65536), 0, row0 < batch ? 65536 * 4 * IO_ON : 0, 0x00020000);
        const int load_off = thread_id(wave_sgpr) * 8;
#pragma unroll
        for (int j1 = 0; j1 < 32; ++j1)
            v[j1] = to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r0, load_off, j1 * 8192, 0)));
    }

    int it = 0;
    for (int row = blockIdx.x; row < batch; row += gridDim.x, ++it) {
        // thread-derived indices are re-derived per phase from a laundered thread id (see
        // per_row): kept live across the row they cost ~10 VGPRs, which the compiler spills, and
        // every scratch reload drains vmcnt, i.e. waits for all outstanding global stores.
        STAMP(0);

        // row descriptors: wave-uniform base, per-lane 32-bit byte offset, SGPR/immediate steps.
        // The next row's descriptor has zero records past the end of the batch: loads return 0.
        const int next_row = row + gridDim.x;
        const __amdgpu_buffer_rsrc_t rnext = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (x + (size_t) next_row * 65536), 0, next_row < batch ? 65536 * 4 * IO_ON : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc((void *) (X + (size_t) row * (kM + 1)), 0, (kM + 1) * 8 * IO_ON, 0x00020000);

        STAMP(1);                                          // loads issued
        // ---- pass 1 (over j1) and twiddle W_1024^{j2 k1}
        if (!SKIP(1)) dft32<false>(v);
        if (!SKIP(2)) {
            const int hi = thread_id(wave_sgpr) >> 5;      // j2
#pragma unroll
            for (int k1 = 1; k1 < 32; ++k1) v[br5(k1)] = cmul(v[br5(k1)], to_cf(w1024[hi * k1]));
        }

        // ---- exchange 1: (j2, j3)[k1] -> (k1, j3)[j2];  row = k1*32 + j3 (slot k1), col = j2
        const int t = thread_id(wave_sgpr);
        const int wbase1 = (t & 31) * kRowPitch + (t >> 5);
        cf u[32];
        STAMP(2);                                          // pass 1 done (includes the wait for the loads)
        // Barriers sit AFTER each read phase (not before each write phase): a wave's LDS writes
        // then overlap the tail of its own butterflies and the other waves' arithmetic.
        if (!SKIP(4)) plane_write<0>(plane, wbase1, v);
        if (!SKIP(8)) lds_barrier();
        if (!SKIP(4)) plane_read<0>(plane, t, u); else { for (int i = 0; i < 32; ++i) u[i] = v[31 - i]; }
        if (!SKIP(8)) lds_barrier();
        if (!SKIP(4)) plane_write<1>(plane, wbase1, v);
        if (!SKIP(8)) lds_barrier();
        if (!SKIP(4)) plane_read<1>(plane, t, u);
        if (!SKIP(8)) lds_barrier();                     // plane free for exchange 2
        STAMP(3);                                          // exchange 1 done

        // ---- pass 2 (over j2) and twiddle W_32768^{j3 k1} * W_1024^{j3 k2}
        if (!SKIP(1)) dft32<false>(u);
        if (!SKIP(2)) {
            const int t2 = thread_id(wave_sgpr);
            const int hi = t2 >> 5, lo = t2 & 31;                      // (k1, j3)
            const cf tw2_base = to_cf(aux[kAuxW32768 + hi * lo]);             // W_32768^{j3 k1}
            u[0] = cmul(u[0], tw2_base);
#pragma unroll
            for (int k2 = 1; k2 < 32; ++k2) u[br5(k2)] = cmul(u[br5(k2)], cmul(tw2_base, to_cf(w1024[lo * k2])));
        }

        // ---- exchange 2: (k1, j3)[k2] -> column k' = k1 + 32 k2, [j3];  row = k' (slot k2), col = j3
        const int t3 = thread_id(wave_sgpr);
        const int wbase2 = (t3 >> 5) * kRowPitch + (t3 & 31);
        const int kp2 = column_of(t3 >> 6, t3 & 63);
        STAMP(4);                                          // pass 2 done
        if (!SKIP(4)) plane_write<0>(plane, wbase2, u);
        if (!SKIP(8)) lds_barrier();
        if (!SKIP(4)) plane_read<0>(plane, kp2, v); else { for (int i = 0; i < 32; ++i) v[i] = u[31 - i]; }
        if (!SKIP(8)) lds_barrier();
        if (!SKIP(4)) plane_write<1>(plane, wbase2, u);
        if (!SKIP(8)) lds_barrier();
        if (!SKIP(4)) plane_read<1>(plane, kp2, v);
        if (!SKIP(8)) lds_barrier();                     // plane free for the next row's exchange 1
        STAMP(5);                                          // exchange 2 done

        // ---- pass 3 (over j3): v[p] = Z[k' + 1024 br5(p)]
        if (!SKIP(1)) dft32<false>(v);
        STAMP(6);                                          // pass 3 done

        // ---- packed-real post-pass.  Rows 0..15 of this column pair with rows 31..16 of the
        // partner column (odd registers there); fetch them, finish both bins of each pair:
        // xk[k3] = X[k], xm[k3] = X[M-k], k = k' + 1024 k3.
        const int t4 = thread_id(wave_sgpr);
        const int lane = t4 & 63, wave = t4 >> 6;
        const int kp = column_of(wave, lane);
        const int partner_addr = ((wave == 0 && (lane == 63 || lane == 0)) ? lane : 63 - lane) * 4;
        cf xk[16], xm[16];
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            const int src = 31 - br5(k3);                  // = br5(31 - k3)
            xm[k3].x = bperm(partner_addr, v[src].x);
            xm[k3].y = bperm(partner_addr, v[src].y);
        }
        const cf zmid = v[br5(16)];                        // Z[M/2] in column 0
        if (wave == 0) {                                   // column 0 pairs row k3 with row 32 - k3 of itself
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) {
                const cf own = v[br5((32 - k3) & 31)];
                xm[k3] = lane == 0 ? own : xm[k3];
            }
        }
        {
            // -(i/2) W_65536^{k'}: the post-pass multiplies (a - conj b) by -i w / 2
            const cf wpost = to_cf(aux[kAuxW65536 + kp]);
            const cf post_base = cf{0.5f * wpost.y, -0.5f * wpost.x};
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) {
                const cf a = v[br5(k3)], b = xm[k3];
                const cf s = cf{a.x + b.x, a.y - b.y};                // a + conj b
                const cf d = cf{a.x - b.x, a.y + b.y};                // a - conj b
                const cf c = cf{root64_re(k3), root64_im(k3)};        // W_64^{k3} = W_65536^{1024 k3}
                const cf w = (k3 == 0 || SKIP(16)) ? post_base : cmul(post_base, c);
                const cf wd = SKIP(16) ? d : cmul(d, w);
                xk[k3] = cf{0.5f * s.x + wd.x, 0.5f * s.y + wd.y};
                xm[k3] = cf{0.5f * s.x - wd.x, wd.y - 0.5f * s.y};
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
        for (int j1 = 0; j1 < 16; ++j1)                    // xk[] is dead: first half of the next row
            v[j1] = to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rnext, load_off, j1 * 8192, 0)));
        if (!SKIP(8)) lds_barrier();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int k = 2 * (t4 + 1024 * m) - skew;       // first bin of this lane's 16-B chunk
            if (m > 0 || k >= 0) {                          // only the row's first chunk can start before bin 0
                const f2 lo2 = stage[k], hi2 = stage[k + 1];
                const f4 q = f4{lo2.x, lo2.y, hi2.x, hi2.y};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q), rout, k * 8, 0, 0);
            } else if (m == 0 && k == -1) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, stage[0]), rout, 0, 0, 0);
            }
            if (m & 1) __builtin_amdgcn_sched_barrier(0);       // at most two chunks of staging reads in flight (VGPR budget)
        }
        if (!SKIP(8)) lds_barrier();
        // half 2: bins [16384 - skew, 32768], staged at index bin - kStage2
        constexpr int kStage2 = kM / 2 - 16;
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) stage[(kM - kStage2) - kp - 1024 * k3] = to_f2(xm[k3]);
        if (kp >= 1009) stage[kp + 15 * 1024 - kStage2] = to_f2(xk15);               // bins 16369..16383
        if (t4 == 0) stage[kM / 2 - kStage2] = f2{zmid.x, -zmid.y};            // bin M/2 = conj Z[M/2] (dsc_fft.h:218)
#pragma unroll
        for (int j1 = 16; j1 < 32; ++j1)                   // xm[] is dead: second half of the next row
            v[j1] = to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rnext, load_off, j1 * 8192, 0)));
        if (!SKIP(8)) lds_barrier();
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            if (m == 8 && wave != 0) break;                // bins past 32768 + 15 do not exist
            const int k = kM / 2 + 2 * (t4 + 1024 * m) - skew;
            if (m < 8 || k + 1 <= kM) {                     // only the row's last chunks can run past bin M
                const f2 lo2 = stage[k - kStage2], hi2 = stage[k + 1 - kStage2];
                const f4 q = f4{lo2.x, lo2.y, hi2.x, hi2.y};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q), rout, k * 8, 0, 0);
            } else if (m == 8 && k == kM) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, stage[k - kStage2]), rout, k * 8, 0, 0);
            }
            if (m & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (!SKIP(8)) lds_barrier();                     // plane free for the next row's exchange 1
        STAMP(7);                                          // post-pass done, stores issued
    }
}


// ------------------------------------------------------------------------------------------
// inverse: X [batch][32769] c32  ->  x [batch][65536] f32        (dsc_irfft, dsc_fft.h:194-236)
//
// The forward pipeline run backwards.  The bins are read in the column layout of the forward
// post-pass (lane l and lane 63-l of a wave hold columns c and 1024-c, 32 rows k = c + 1024 a
// each), the packed-real pre-pass Z[k] = h1 + conj(w) h2 pairs them through ds_bpermute, and
// three conjugate-twiddle passes over (a, b, c') with k = 1024 a + 32 b + c' end with thread t
// holding z[t + 1024 r]: the time samples leave as aligned, coalesced 8-B stores.
// The 2/(2n) scale of the reference (dsc_fft.h:232) is folded into the pre-pass constants.
__global__ __launch_bounds__(1024) void irfft64k_kernel(const f2 *__restrict__ X, float *__restrict__ x, int batch,
                                                        const f2 *__restrict__ aux) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *plane = lds;
    f2 *w1024 = (f2 *) (lds + kPlaneFloats);
    w1024[threadIdx.x] = aux[kAuxW1024 + threadIdx.x];
    __syncthreads();

    const int wave_sgpr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr float kScale = 1.0f / (float) kM;

    for (int row = blockIdx.x; row < batch; row += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rin =
            __builtin_amdgcn_make_buffer_rsrc((void *) (X + (size_t) row * (kM + 1)), 0, (kM + 1) * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc((void *) (x + (size_t) row * 65536), 0, 65536 * 4, 0x00020000);

        // ---- load column c, rows a = 0..31: Y[c + 1024 a]
        cf v[32];
        cf y_last = cf{0.f, 0.f};
        {
            const int t0 = thread_id(wave_sgpr);
            const int c = column_of(t0 >> 6, t0 & 63);
#pragma unroll
            for (int a = 0; a < 32; ++a)
                v[a] = to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rin, c * 8, a * 8192, 0)));
            if (c == 0) y_last = to_cf(__builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rin, kM * 8, 0, 0)));   // bin M
        }

        // ---- packed-real pre-pass (dsc_fft.h:199-228): rows 0..15 pair with rows 31..16 of the
        // partner column.  Z[k] = s/2 + wq d,  Z[M-k] = conj(s/2 - wq d),  s = p + conj q,
        // d = p - conj q,  wq = (i/2) conj(W_65536^k); everything pre-multiplied by 1/M.
        {
            const int t1 = thread_id(wave_sgpr);
            const int lane = t1 & 63, wave = t1 >> 6;
            const int c = column_of(wave, lane);
            const int partner_addr = ((wave == 0 && (lane == 63 || lane == 0)) ? lane : 63 - lane) * 4;
            // Column 0 (lane 0 of wave 0) pairs row a with row 32 - a of itself and row 0 with bin M.
            // Shifting its rows 17..31 down by one (and putting bin M in row 31) turns that into
            // the general "row a with row 31 - a of the partner" with itself as partner.
            const cf y_mid = v[16];                        // bin M/2 pairs with itself
            if (wave == 0) {
#pragma unroll
                for (int r = 16; r < 31; ++r) v[r] = lane == 0 ? v[r + 1] : v[r];
                v[31] = lane == 0 ? y_last : v[31];
                if (lane == 0) { v[0].y = 0.f; v[31].y = 0.f; }         // dsc_fft.h:227-228 reads the real parts only
            }
            const cf wpre = to_cf(aux[kAuxW65536 + c]);
            const cf wq_base = cf{0.5f * kScale * wpre.y, 0.5f * kScale * wpre.x};      // (i/2) conj(W^c) / M
#pragma unroll
            for (int half = 0; half < 2; ++half) {         // two batches of 8 pairs: VGPR budget
                cf q[8], zm[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int a = half * 8 + i;
                    q[i].x = bperm(partner_addr, v[31 - a].x);
                    q[i].y = bperm(partner_addr, v[31 - a].y);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int a = half * 8 + i;
                    const cf p = v[a];
                    const cf s = cf{p.x + q[i].x, p.y - q[i].y};
                    const cf d = cf{p.x - q[i].x, p.y + q[i].y};
                    const cf cj = cf{root64_re(a), -root64_im(a)};       // conj(W_64^a)
                    const cf wq = a == 0 ? wq_base : cmul(wq_base, cj);
                    const cf wd = cmul(d, wq);
                    v[a] = cf{0.5f * kScale * s.x + wd.x, 0.5f * kScale * s.y + wd.y};
                    zm[i] = cf{0.5f * kScale * s.x - wd.x, wd.y - 0.5f * kScale * s.y};
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {              // the partner finished my rows 31..16
                    const int a = half * 8 + i;
                    v[31 - a].x = bperm(partner_addr, zm[i].x);
                    v[31 - a].y = bperm(partner_addr, zm[i].y);
                }
            }
            if (wave == 0) {                               // undo the shift of column 0
#pragma unroll
                for (int r = 31; r > 16; --r) v[r] = lane == 0 ? v[r - 1] : v[r];
                v[16] = lane == 0 ? cf{kScale * y_mid.x, -kScale * y_mid.y} : v[16];   // Z[M/2] = conj Y[M/2] (dsc_fft.h:218)
            }
        }

        // ---- pass 1 (over a), twiddle conj(W_1024^{b r1}), exchange 1: (b, c')[r1] -> (r1, c')[b]
        dft32<true>(v);
        cf u[32];
        {
            const int t2 = thread_id(wave_sgpr);
            const int c = column_of(t2 >> 6, t2 & 63);
            const int b = c >> 5;
#pragma unroll
            for (int r1 = 1; r1 < 32; ++r1) v[br5(r1)] = cmul_conj(v[br5(r1)], to_cf(w1024[b * r1]));
            const int wbase1 = (c & 31) * kRowPitch + b;
            plane_write<0>(plane, wbase1, v);
            lds_barrier();
            plane_read<0>(plane, t2, u);
            lds_barrier();
            plane_write<1>(plane, wbase1, v);
            lds_barrier();
            plane_read<1>(plane, t2, u);
            lds_barrier();
        }

        // ---- pass 2 (over b), twiddle conj(W_32768^{c' r1} W_1024^{c' r2}), exchange 2 -> thread t, [c']
        dft32<true>(u);
        {
            const int t3 = thread_id(wave_sgpr);
            const int hi = t3 >> 5, lo = t3 & 31;          // (r1, c')
            const cf tw2_base = to_cf(aux[kAuxW32768 + hi * lo]);
            u[0] = cmul_conj(u[0], tw2_base);
#pragma unroll
            for (int r2 = 1; r2 < 32; ++r2) u[br5(r2)] = cmul_conj(u[br5(r2)], cmul(tw2_base, to_cf(w1024[lo * r2])));
            const int wbase2 = hi * kRowPitch + lo;
            plane_write<0>(plane, wbase2, u);
            lds_barrier();
            plane_read<0>(plane, t3, v);
            lds_barrier();
            plane_write<1>(plane, wbase2, u);
            lds_barrier();
            plane_read<1>(plane, t3, v);
            lds_barrier();
        }

        // ---- pass 3 (over c'): v[p] = z[t + 1024 br5(p)] = (x[2j], x[2j+1])
        dft32<true>(v);
        {
            const int t4 = thread_id(wave_sgpr);
#pragma unroll
            for (int p = 0; p < 32; ++p)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, to_f2(v[p])), rout, t4 * 8, br5(p) * 8192, 0);
        }
    }
}

}  // namespace

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

void dsc_launch_rfft64k(const float *x, void *X, int batch, const void *aux, int n_cu, hipStream_t stream) {
    if (batch <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) rfft64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        attr_set = true;
    }
    const int grid = batch < n_cu ? batch : n_cu;
    hipLaunchKernelGGL(rfft64k_kernel, dim3(grid), dim3(1024), kLdsBytes, stream, x, (f2 *) X, batch, (const f2 *) aux PROBE_NULL);
}

void dsc_launch_irfft64k(const void *X, float *x, int batch, const void *aux, int n_cu, hipStream_t stream) {
    if (batch <= 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void) hipFuncSetAttribute((const void *) irfft64k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        attr_set = true;
    }
    const int grid = batch < n_cu ? batch : n_cu;
    hipLaunchKernelGGL(irfft64k_kernel, dim3(grid), dim3(1024), kLdsBytes, stream, (const f2 *) X, x, batch, (const f2 *) aux);
}
void dsc_launch_filter64k(const float *, const void *, float *, int, const void *, int, hipStream_t) {}
