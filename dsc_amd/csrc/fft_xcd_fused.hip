// fft_xcd_fused.hip — long complex rows (L = 256 x L2: 32768 and 131072 points in f64, 65536 in f32 and f64; real lengths twice
// that) in ONE launch, the four-step intermediate held in the L2 of the XCD that works on the row instead of making a round trip
// through HBM (fft_r2c_2pass.hip moves every row twice).  Described for L2 = 256:
//
//   j = j1 + 256 j2   (input),      k = 256 k1 + k2   (output),      j1, k1, j2, k2 < 256
//   rows  A[j1][k2] = W_L^{j1 k2} * sum_{j2} z[j1 + 256 j2] W_256^{j2 k2}       256-point FFTs, tasks of 16 adjacent j1
//   cols  Z[256 k1 + k2] = sum_{j1} A[j1][k2] W_256^{j1 k1}                     256-point FFTs, tasks of 16 columns k2
//   X from Z by the packed-real pass (dsc_fft.h:199-225), fused into the column task
//
// A task is 16 values per thread of a 256-thread (f64) or 512-thread (f32) workgroup: 4096 / 8192 complex, i.e. 16 / 32 adjacent
// lines of a row task (L2 = 256; L2 = 128, 512: twice / half as many) and 16 / 32 columns of a column task — pieces of at least
// 128 B on the external side.  256 = 16 x 16: two in-register 16-point passes and one LDS exchange.  A row is L2 / 16 (f64) or
// L2 / 32 (f32) row tasks, then as many column tasks, by a TEAM of that many persistent workgroups that run on the same XCD
// (HW_REG_XCC_ID), so that the 512 KiB .. 2 MiB of A they write and read back stay in that XCD's 4 MiB L2:
//   * teams form at kernel start from the order in which workgroups of one XCD arrive (a counter per XCD), after one
//     grid-wide arrival count; the launch is sized to be fully resident (two workgroups per CU);
//   * the team barrier is a counter in the XCD's own L2 — plain atomics, no agent-scope fence, hence no L2 write-back /
//     invalidate; A is written with ordinary stores (the L1 is write-through) and read with sc1 loads (miss the L1);
//   * teams work in PAIRS that share one scratch row and take turns on it (see below): few rows of A are live per XCD, yet
//     there is always a second team to fill a team's waits;
//   * rows are claimed from a global counter by the team's first workgroup and published at a barrier; the next row's samples
//     are requested during the second task of the current one;
//   * every spin is bounded: a barrier that does not complete writes a code to a pinned host word (the host aborts at the next synchronise).
// Measured skeleton and the L2 behaviour behind this design: tools/xcdbench.hip, profiles/r02_c5_infinity_cache.md section 3.
// Reference: dsc_rfft / dsc_irfft / dsc_fft / dsc_ifft (dsc/src/dsc.cpp:1958-2260, dsc_fft.h:57-238).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <utility>

#include "fft_regs_common.h"

namespace {

constexpr int kMaxTeams = 8;              // teams per XCD
constexpr int kPQ = 276, kPK = 17;        // row-task exchange: line pitch (values), k2' pitch
#ifndef DSC_FUSED_POLL_LOAD
#define DSC_FUSED_POLL_LOAD 0
#endif
#ifndef DSC_FUSED_EXT_LOAD
#define DSC_FUSED_EXT_LOAD kStream
#endif
#ifndef DSC_FUSED_EXT_STORE
#define DSC_FUSED_EXT_STORE kStream
#endif
#ifndef DSC_FUSED_BINS_LOAD
#define DSC_FUSED_BINS_LOAD kStream       // round 3: -3 % on the inverse of config 5 and 1 GB less traffic (profiles/r03_l2probe.md); loads have no partial-line penalty
#endif
#ifndef DSC_FUSED_BINS_STORE
#define DSC_FUSED_BINS_STORE kCached
#endif
// workgroups per CU the f64 (256-thread) forms are built for and ask for: 2 = teams in pairs, 256 VGPRs; 3 caps the kernel at 168 VGPRs
#ifndef DSC_FUSED_WG_PER_CU
#define DSC_FUSED_WG_PER_CU 2
#endif
constexpr int kCoherent = 16;             // aux bits: sc1 (device scope: the load misses the L1)
constexpr unsigned kSpinLimit = 1u << 22; // polls (each > 0.5 us) before a barrier gives up

struct fused_ctl {                        // zero before every launch; one 256-B block per writer group (no L2 line shared between XCDs)
    unsigned arrived, pad0[63];
    unsigned row_counter, pad1[63];
    unsigned xcc_count[8][64];
    unsigned team[8 * kMaxTeams][64];     // [0] barrier counter, [4] / [5] rows published by the team's first workgroup
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }   // HW_REG_XCC_ID[3:0]

// atomics that execute in the issuing XCD's L2 (no scope bits)
__device__ __forceinline__ unsigned l2_fetch_add(unsigned *p, unsigned v) {
    unsigned r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}
// polling read of a counter.  An atomic add of zero, not a load: polling with `global_load_dword sc1` was measured 25-30 % slower
// end to end (config 5: 3.66 vs 2.94 ms) — the waiters see the count later
__device__ __forceinline__ unsigned l2_peek(unsigned *p) {
#if DSC_FUSED_POLL_LOAD
    unsigned r;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
#else
    return l2_fetch_add(p, 0u);
#endif
}
__device__ __forceinline__ void l2_add(unsigned *p, unsigned v) { asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void l2_store(unsigned *p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
}

// 128-bit stores (c64) at two or more waves per SIMD: the data registers of a buffer_store_dwordx4 must not be rewritten by the very
// next VALU instruction — observed on gfx950 as the last quad of each 16-lane row storing the NEXT value (tools/stress_fused.py:
// 16 elements of one store instruction wrong in ~1 row of 100, only with f64 data at occupancy 2).  hipcc pads this hazard only for a
// literal soffset; two wait states after every such store cost nothing here.
template<int POL, typename R>
__device__ __forceinline__ void st(cpx<R> a, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    buf_store<POL>(a, r, voff, soff);                       // the c64 form carries the two wait states, tied to its data registers
}

template<typename R, bool INV, int M, int G, int... K>
__device__ __forceinline__ void x_group(cpx<R> (&v)[16], std::integer_sequence<int, K...>) {
    (([&] {
         const cpx<R> u = v[G + K] + v[G + K + M / 2];
         const cpx<R> d = v[G + K] - v[G + K + M / 2];
         v[G + K] = u;
         v[G + K + M / 2] = mul_root<R, INV, M, K>(d);
     }()),
     ...);
}
template<typename R, bool INV, int M, int... G>
__device__ __forceinline__ void x_stage(cpx<R> (&v)[16], std::integer_sequence<int, G...>) {
    (x_group<R, INV, M, G * M>(v, std::make_integer_sequence<int, M / 2>{}), ...);
}
// 16-point DFT, natural order in, v[p] = bin brev(p, 4)
template<typename R, bool INV>
__device__ __forceinline__ void dft16(cpx<R> (&v)[16]) {
    x_stage<R, INV, 16>(v, std::make_integer_sequence<int, 1>{});
    x_stage<R, INV, 8>(v, std::make_integer_sequence<int, 2>{});
    x_stage<R, INV, 4>(v, std::make_integer_sequence<int, 4>{});
    x_stage<R, INV, 2>(v, std::make_integer_sequence<int, 8>{});
}

// two 8-point DFTs, of v[0..7] and of v[8..15]: natural order in, v[8 h + p] = bin brev(p, 3) of half h
template<typename R, bool INV>
__device__ __forceinline__ void dft8x2(cpx<R> (&v)[16]) {
    x_stage<R, INV, 8>(v, std::make_integer_sequence<int, 2>{});
    x_stage<R, INV, 4>(v, std::make_integer_sequence<int, 4>{});
    x_stage<R, INV, 2>(v, std::make_integer_sequence<int, 8>{});
}

// W_L^{j1 (tau + S k)}, k = 4 a + b: W_L^{j1 tau} W_L^{4 S j1 a} W_L^{S j1 b}, three exact table values (S = 8, 16 or 32)
template<typename R, bool CONJ, bool BREV, int S>
__device__ __forceinline__ void four_step_twiddle16(cpx<R> (&v)[16], const cpx<R> *twL, int j1, int tau) {
    using C = cpx<R>;
    const C base = twL[j1 * tau];
    const C b1 = twL[S * j1], b2 = twL[2 * S * j1], b3 = twL[3 * S * j1];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const C bq = a == 0 ? base : cmul(base, twL[4 * S * j1 * a]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const C w = b == 0 ? bq : b == 1 ? cmul(bq, b1) : b == 2 ? cmul(bq, b2) : cmul(bq, b3);
            const int k = 4 * a + b;
            const int r = BREV ? brev(k, 4) : k;
            v[r] = CONJ ? cmulc(v[r], w) : cmul(v[r], w);
        }
    }
}

// REAL: dsc_rfft (forward) / dsc_irfft (INV).  !REAL: dsc_fft / dsc_ifft of complex rows.
// ext  = the time-domain side (forward input, inverse output): row pitch ext_pitch_b bytes, ext_len_b valid bytes
// bins = the frequency-domain side: row pitch bins_pitch bins, bins_len valid bins
// L2 = 128, 256 or 512: the row-task transform length (L = 256 L2).  kNT = 256 threads (tasks of 4096 complex, 16 columns) or 512
// (8192 complex, 32 columns: f32 rows of 131072 points, whose pieces would be 64 B otherwise); TS = L2 / (kNT / 16) workgroups
// per team.
// CAST (complex transforms only): the input rows are REAL and widened on the way in (dsc_fft / dsc_ifft of a real tensor,
// dsc.cpp:1984-1988).
template<typename R, bool REAL, bool INV, int L2, int kNT, bool CAST = false>
__global__ __launch_bounds__(kNT, (kNT == 512 ? 4 : DSC_FUSED_WG_PER_CU)) void fused_l2_kernel(const char *__restrict__ ext, char *__restrict__ ext_out,
                                                                               const cpx<R> *__restrict__ bins_in, cpx<R> *__restrict__ bins_out,
                                                                               cpx<R> *scratch, fused_ctl *ctl, unsigned *host_error, int rows, int teams_cap,
                                                                               const cpx<R> *__restrict__ twL, const cpx<R> *__restrict__ tw_real, R scale,
                                                                               long long ext_pitch_b, int ext_len_b, long long bins_pitch, int bins_len) {
    using C = cpx<R>;
    constexpr int CB = (int) sizeof(C), L = 256 * L2, NC = kNT / 16, H = NC / 2;
    constexpr int kTS = L2 / NC;                                    // tasks per phase = workgroups per team
    constexpr int LINES = 16 * kNT / L2, TPL = L2 / 16;             // row task: lines per task, threads per line
    constexpr int kPQ5 = 532, kPK5 = 33;                            // row-task exchange of the 512-point lines
    constexpr int BL = REAL ? DSC_FUSED_BINS_LOAD : kStream;                    // spectrum rows of the real transforms are skewed: fft_r2c_2pass.hip
    // ... and written in 64-B pieces that straddle sectors: with the default policy the pieces of the 16 column tasks of a row
    // (same XCD, same moment) meet in the L2 and leave as whole lines
    constexpr int BS = REAL ? DSC_FUSED_BINS_STORE : kStream;
    constexpr int kPQ1 = 148, kPK1 = 9;                             // ... of the 128-point lines (32 lines)
    constexpr int kPlaneRows = LINES * (L2 == 128 ? kPQ1 : L2 == 256 ? kPQ : kPQ5), kPlaneCols = 257 * NC;
    __shared__ __attribute__((aligned(16))) R plane[kPlaneRows > kPlaneCols ? kPlaneRows : kPlaneCols];      // row task: LINES x pitch; column task: [k1][ell] 257 x NC
    constexpr int TAB = L2 > 256 ? L2 : 256;                        // W_TAB^m: W_256^m = wtab[W1 m], W_L2^m = wtab[W2 m]
    __shared__ C wtab[TAB];
    __shared__ int info[8];
    const int tid = threadIdx.x;
    for (int i = tid; i < TAB; i += kNT) wtab[i] = twL[i * (L / TAB)];
    constexpr int W1 = TAB / 256, W2 = TAB / L2;

    // ---- teams
    // (Round 3, measured and NOT adopted: forming the teams per CU — the first workgroup to arrive on a CU joins an even team, the second its odd
    // partner, HW_REG_HW_ID[15:8] names the CU (tools/hwid_probe.hip) — removes the members that share a CU with a team mate and make the team
    // barrier wait every row (profiles/r03_l2probe.md section 4), but then all 32 members store and read `A` in the same 3 us and the slot
    // cycle stays at ~10 us: config 5 3.03 -> 3.05 ms, its inverse 3.08 -> 3.32, c64 ifft L = 131072 2.45 -> 2.84.  The arrival order spreads
    // a team's members over the phases; it stays.)
    if (tid == 0) {
        const unsigned x = xcc_id() & 7u;                           // MI355X: 8 XCDs (the control block has 8 slots)
        const unsigned arr = l2_fetch_add(&ctl->xcc_count[x][0], 1u);
        info[0] = (int) x; info[1] = (int) arr; info[2] = 0;
        atomicAdd(&ctl->arrived, 1u);
        unsigned spins = 0;
        while (atomicAdd(&ctl->arrived, 0u) < gridDim.x) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > kSpinLimit) { __hip_atomic_store(host_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); info[2] = 1; break; }
        }
        info[3] = (int) (l2_fetch_add(&ctl->xcc_count[x][0], 0u) / kTS);
    }
    __syncthreads();
    if (info[2]) return;
    // wave-uniform values read from LDS: readfirstlane keeps the descriptors built from them in SGPRs (no waterfall loops)
    const int arrival = __builtin_amdgcn_readfirstlane(info[1]);
    const int xcc = __builtin_amdgcn_readfirstlane(info[0]), team = arrival / kTS, rank = arrival % kTS;
    if (team >= __builtin_amdgcn_readfirstlane(info[3]) || team >= kMaxTeams || team >= teams_cap) return;               // workgroups that do not fill a team
    // Counters of a team (one 256-B block in the XCD's L2): [0] arrivals at "A is written", [4] / [5] published rows, [8] arrivals
    // at "A has been read", [12] members that have left.
    // PAIRS.  A row of A is 512 KiB .. 2 MiB and the 4 MiB L2 also carries the streams, so the L2 has room for few rows — too few
    // teams to hide a team's barrier chain if every team owned one.  Teams 2p and 2p + 1 of an XCD therefore SHARE one scratch row
    // and take turns: a team may write A only once its partner has read its own; while one team is in its write -> barrier ->
    // read window the other computes its column task and the next row task.  Two workgroups per CU.  Measured against teams with
    // a row each: c64 L = 65536 1.31 -> 1.07 ms, rfft f64 N = 131072 1.52 -> 1.20 ms, config 5 3.37 -> 3.0-3.2 ms.
    unsigned *tb = &ctl->team[xcc * kMaxTeams + team][0];
    const int n_teams = __builtin_amdgcn_readfirstlane(info[3]) < teams_cap ? __builtin_amdgcn_readfirstlane(info[3]) : teams_cap;
    const bool has_partner = (team ^ 1) < n_teams && (team ^ 1) < kMaxTeams;
    unsigned *pb = &ctl->team[xcc * kMaxTeams + (team ^ 1)][0];
    C *scr = scratch + (size_t) (xcc * teams_cap + (team >> 1)) * L;
    const __amdgpu_buffer_rsrc_t rwork = __builtin_amdgcn_make_buffer_rsrc((void *) scr, 0, L * CB, 0x00020000);
    unsigned target = 0;
    bool broken = false;

    // arrive: everything this workgroup stored is in the L2; wait: the whole team has arrived.  `publish`: the team's first
    // workgroup claims the row after next and leaves it in slot `slot` before it arrives; everyone reads it when the barrier opens.
    auto arrive = [&](bool stores_pending, int slot) {
        target += kTS;
        if (stores_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (tid == kNT - 1) {
            if (rank == 0 && slot >= 0) l2_store(tb + 4 + slot, atomicAdd(&ctl->row_counter, 1u));
            l2_add(tb, 1u);
        }
    };
    // "this workgroup has read its part of A" (counter [8]); and the wait before A is written again: until `need` such arrivals
    // at counter `c`, or until every member of that team has left (counter c[4] = [12])
    auto arrive_read = [&]() {
        lds_barrier();
        if (tid == kNT - 1) l2_add(tb + 8, 1u);
    };
    auto wait_read = [&](unsigned *c, unsigned need) {
        if (tid == kNT - 1) {
            unsigned spins = 0;
            while ((int) (l2_peek(c) - need) < 0) {
                if ((spins & 7u) == 7u && l2_peek(c + 4) >= (unsigned) kTS) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { __hip_atomic_store(host_error, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); info[2] = 1; break; }
            }
        }
        lds_barrier();
        broken = info[2] != 0;
    };
    auto leave = [&]() {
        lds_barrier();
        if (tid == kNT - 1) l2_add(tb + 12, 1u);
    };
    auto spin = [&](int slot) {
        if (tid == kNT - 1) {
            unsigned spins = 0;
            while ((int) (l2_peek(tb) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) { __hip_atomic_store(host_error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); info[2] = 1; break; }
            }
            if (slot >= 0) info[4 + slot] = (int) l2_fetch_add(tb + 4 + slot, 0u);
        }
    };
    auto join = [&]() {
        lds_barrier();
        broken = info[2] != 0;
    };

    // ---- lane roles
    // row task `rank`: lines j1 = 16 rank + q
    //   writer lanes tid = 16 s + q: hold z[j1 + 256 (s + 16 m)], m < 16   (pieces of 16 adjacent j1)
    //   reader lanes tid = 16 q + tau: hold A[j1][tau + 16 k], k < 16       (pieces of 16 adjacent k2)
    //   L2 = 512: 8 lines per task, 32 threads per line: writer tid = 8 s + q holds z[j1 + 256 (s + 32 m)]; reader tid = 32 q + tau,
    //   tau = ka + 16 c, holds A[j1][tau + 32 kb] (the 32-point second stage = a radix-2 step folded into the LDS read + 16 points)
    const int wq = tid % LINES, ws = tid / LINES;
    const int rq = tid / TPL, rtau = tid % TPL;
    constexpr int EB = (CAST && !INV) ? (int) sizeof(R) : CB;      // bytes per sample on the time-domain side
    const int zoff = ((LINES * rank + wq) + 256 * ws) * EB;    // REAL: z[j] = (x[2j], x[2j + 1])
    const int aoff = ((LINES * rank + rq) * L2 + rtau) * CB;
    const int j1r = LINES * rank + rq;
    // the seven table values of the inter-pass twiddle are re-read per row (L2 hits) instead of living in 14 / 28 registers
    auto opaque_j1 = [&]() { int j = j1r; asm volatile("" : "+v"(j)); return j; };
    constexpr int ZSTEP = TPL * 256 * EB, ASTEP = TPL * CB;
    // column task `rank`: lanes tid = 16 t + ell: column ell (REAL: 8 columns 8 b + 1 .. 8 b + 8 and their mirrors; column 0
    // takes the place of the duplicate 128 in the last block), slice t of the 256-point axis (j1 = t + 16 i; k1 = t + 16 k)
    const int ell = tid % NC, t = tid / NC;
    const bool last = rank == kTS - 1;
    const bool col0 = REAL && last && ell == H;
    const int col = !REAL ? NC * rank + ell : col0 ? 0 : ell < H ? H * rank + 1 + ell : L2 - H - H * rank + (ell - H);
    const int ellp = (last && (ell == H - 1 || ell == H)) ? ell : NC - 1 - ell;
    const int woff = (t * L2 + col) * CB, boff = (L2 * t + col) * CB;       // bin L2 (t + 16 k) + col: + k * 16 BSTEP
    constexpr int WSTEP = 16 * L2 * CB, BSTEP = L2 * CB;
    R *mine = plane + t * NC + ell;                                  // plane[k1 = t + 16 k][ell]: + k * 16 NC
    const R *theirs = plane + (15 - t) * NC + ellp + (col0 ? NC : 0);     // plane[255 - k1 (+ 1 in column 0)][ellp]: + (15 - k) * 16 NC
    const C wt0 = REAL ? cmul(tw_real[col], tw_real[L2 * t]) : C{(R) 1, (R) 0};       // W_2L^{L2 t + col}

    int next = 0;
    C cur[16];                                                      // the samples (bins) of the current row; refilled with the next row's
    C cur_last = C{(R) 0, (R) 0};                                   // while the second task of the current one runs (it is dead by then)
    auto request = [&](C (&dst)[16], C &dlast, int row) {
        if constexpr (!INV) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) (ext + (size_t) row * ext_pitch_b), 0, ext_len_b, 0x00020000);
#pragma unroll
            for (int m = 0; m < 16; ++m) dst[m] = CAST ? buf_load_real<DSC_FUSED_EXT_LOAD>(r, zoff, m * ZSTEP, R{}) : buf_load<DSC_FUSED_EXT_LOAD>(r, zoff, m * ZSTEP, R{});
        } else {
            if constexpr (CAST) {                                   // dsc_ifft of a real tensor: rows of reals
                constexpr int RB = (int) sizeof(R);
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) bins_in + (size_t) row * bins_pitch * RB), 0, bins_len * RB, 0x00020000);
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[e] = buf_load_real<kStream>(r, (L2 * t + col) * RB, 16 * e * L2 * RB, R{});
                return;
            }
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) (bins_in + (size_t) row * bins_pitch), 0, bins_len * CB, 0x00020000);
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[e] = buf_load<BL>(r, boff, 16 * e * BSTEP, R{});
            if (REAL && col0 && t == 0) dlast = buf_load<BL>(r, L * CB, 0, R{});
        }
    };

    // "this workgroup has read A": the partner may overwrite it as soon as the whole team has arrived, so wait for the reads
    // themselves.  (The next row is requested in the middle of the second task, once the registers that held A are free: at two
    // workgroups per CU there are 256 (f64) / 128 (f32, 512 threads) per lane.)
    auto release_scratch = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrive_read();
    };

    // ---- the first two rows of this team
    if (rank == 0 && tid == kNT - 1) {
        l2_store(tb + 4, atomicAdd(&ctl->row_counter, 1u));
        l2_store(tb + 5, atomicAdd(&ctl->row_counter, 1u));
    }
    arrive(false, -1);
    spin(-1);
    join();
    if (broken) return;
    if (tid == kNT - 1) { info[4] = (int) l2_fetch_add(tb + 4, 0u); info[5] = (int) l2_fetch_add(tb + 5, 0u); }
    lds_barrier();
    int row = __builtin_amdgcn_readfirstlane(info[4]);
    next = __builtin_amdgcn_readfirstlane(info[5]);
    if (row < rows) request(cur, cur_last, row);

#ifdef DSC_FUSED_PROFILE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pl = wall_clock64();
    __shared__ unsigned long long pstamp[12][8];          // absolute stamps of rows 20 .. 31 of this workgroup (a timeline of both teams of a pair)
    int pit = 0;
#define PMARK(i) do { const unsigned long long n_ = wall_clock64(); pt[i] += n_ - pl; pl = n_; if (tid == 0 && pit >= 20 && pit < 32) pstamp[pit - 20][i] = n_; } while (0)
#else
#define PMARK(i) do { } while (0)
#endif
    for (int it = 0; row < rows; ++it) {
#ifdef DSC_FUSED_PROFILE
        pit = it;
#endif
        PMARK(0);
        // the previous row's last reads of the LDS plane (packed-real partners, or the second task's exchange) carry no barrier of
        // their own: no wave may start writing the plane for this row before every wave is done with it
        lds_barrier();
        const int slot = it & 1;                                    // slot of `row`: free once everyone holds `row` and `next`
        C u[16], v[16];
        if constexpr (!INV) {
            // ================= row task: samples -> A (scratch)
            dft16<R, false>(cur);                                   // over m -> first index in cur[brev(.)]
#pragma unroll
            for (int k = 1; k < 16; ++k) cur[brev(k, 4)] = cmul(cur[brev(k, 4)], wtab[W2 * ws * k]);
            if constexpr (L2 == 256) {
                R *wr = plane + wq * kPQ + ws;
                const R *rd = plane + rq * kPQ + rtau * kPK;
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[k * kPK] = cur[brev(k, 4)].x;
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m].x = rd[m];
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[k * kPK] = cur[brev(k, 4)].y;
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m].y = rd[m];
            } else if constexpr (L2 == 128) {
                // 8-point second stage: lane tau takes the two first-stage bins ka = tau and tau + 8 (so that it ends up with
                // k2 = tau + 8 k, 8 adjacent lanes = 128 contiguous bytes of A)
                R *wr = plane + wq * kPQ1 + ws;                     // plane[q][ka][s], s < 8
                const R *rd = plane + rq * kPQ1 + rtau * kPK1;
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[k * kPK1] = cur[brev(k, 4)].x;
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 8; ++m) { v[m].x = rd[m]; v[8 + m].x = rd[8 * kPK1 + m]; }
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[k * kPK1] = cur[brev(k, 4)].y;
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 8; ++m) { v[m].y = rd[m]; v[8 + m].y = rd[8 * kPK1 + m]; }
            } else {
                // 32-point second stage over s = s' + 16 c: y_c[s'] = (x[s'] +- x[s' + 16]) (W_32^{s'} for c = 1), then 16 points
                // over s': bin kr = 2 kb + c.  The +- is taken while reading the exchange plane.
                const bool hi = rtau >= 16;
                R *wr = plane + wq * kPQ5 + ws;                     // plane[q][ka][s]
                const R *rd = plane + rq * kPQ5 + (rtau & 15) * kPK5;
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[k * kPK5] = cur[brev(k, 4)].x;
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 16; ++m) { const R a = rd[m], b = rd[m + 16]; v[m].x = hi ? a - b : a + b; }
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[k * kPK5] = cur[brev(k, 4)].y;
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 16; ++m) { const R a = rd[m], b = rd[m + 16]; v[m].y = hi ? a - b : a + b; }
                if (hi) {                                           // (a branch, not a select: the 30 constants stay out of the registers)
#pragma unroll
                    for (int m = 1; m < 16; ++m) v[m] = cmul(v[m], C{(R) root64_re(2 * m), (R) root64_im(2 * m)});      // W_32^{s'}
                }
            }
            if constexpr (L2 == 128) {
                dft8x2<R, false>(v);                                // v[8 h + p]: ka = tau + 8 h, kr = brev(p, 3): k2 = ka + 16 kr = tau + 8 (2 kr + h)
#pragma unroll
                for (int k = 0; k < 16; ++k) u[k] = v[8 * (k & 1) + brev(k >> 1, 3)];
                four_step_twiddle16<R, false, false, TPL>(u, twL, opaque_j1(), rtau);
#pragma unroll
                for (int k = 0; k < 16; ++k) v[brev(k, 4)] = u[k];  // the order the stores below expect
            } else {
                dft16<R, false>(v);                                 // over s (s') -> k in v[brev(k)]: k2 = tau + TPL k
                four_step_twiddle16<R, false, true, TPL>(v, twL, opaque_j1(), rtau);
            }
            PMARK(1);
            // A may be written again once it has been read: by this team (its previous row) and, in a pair, by the partner (whose
            // turn lay in between — unless it has run out of rows, which is why the team's own count is checked as well)
            if (it > 0) {
                wait_read(tb + 8, (unsigned) kTS * (unsigned) it);
                if (broken) return;
            }
            if (has_partner) {
                const unsigned need = (unsigned) kTS * (unsigned) ((team & 1) ? it + 1 : it);
                if (need > 0) { wait_read(pb + 8, need); if (broken) return; }
            }
            PMARK(2);
#pragma unroll
            for (int k = 0; k < 16; ++k) st<kCached>(v[brev(k, 4)], rwork, aoff, k * ASTEP);
            arrive(true, slot);
            PMARK(3);
            spin(slot);
            join();
            PMARK(4);
            if (broken) return;
            // ================= column task: A -> bins
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *) (bins_out + (size_t) row * bins_pitch), 0, bins_len * CB, 0x00020000);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = buf_load<kCoherent>(rwork, woff, i * WSTEP, R{});
            release_scratch();
            PMARK(5);
            dft16<R, false>(v);                                     // over i -> k1' in v[brev(k1')]
#pragma unroll
            for (int k = 1; k < 16; ++k) v[brev(k, 4)] = cmul(v[brev(k, 4)], wtab[W1 * t * k]);
            {
                R *xw = plane + t * NC + ell;                       // plane[k1'][t][ell]
                const R *xr = plane + t * 16 * NC + ell;            // thread (ell, t = k1') reads every slice t'
#pragma unroll
                for (int k = 0; k < 16; ++k) xw[k * 16 * NC] = v[brev(k, 4)].x;
                lds_barrier();
#pragma unroll
                for (int tp = 0; tp < 16; ++tp) u[tp].x = xr[tp * NC];
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) xw[k * 16 * NC] = v[brev(k, 4)].y;
                lds_barrier();
#pragma unroll
                for (int tp = 0; tp < 16; ++tp) u[tp].y = xr[tp * NC];
                lds_barrier();
            }
            if (next < rows) request(cur, cur_last, next);          // v is dead: its registers take the next row
            dft16<R, false>(u);                                     // over t' -> k: u[p] = Z[k1 = t + 16 brev(p)][col]
#ifdef DSC_FUSED_SKIP_REAL_PASS          // timing experiment only (wrong results): what does the packed-real pass cost?
            if constexpr (true) {
#else
            if constexpr (!REAL) {
#endif
#pragma unroll
                for (int p = 0; p < 16; ++p) st<kStream>(u[p], rb, boff, 16 * brev(p, 4) * BSTEP);
            } else {
                // packed-real pass: a = Z[k1][col] (own), b = Z[255 - k1][256 - col] (256 - k1 in column 0), through the plane
                R bx[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) mine[brev(p, 4) * 16 * NC] = u[p].x;
                lds_barrier();
#pragma unroll
                for (int p = 0; p < 16; ++p) bx[p] = theirs[(15 - brev(p, 4)) * 16 * NC];
                lds_barrier();
#pragma unroll
                for (int p = 0; p < 16; ++p) mine[brev(p, 4) * 16 * NC] = u[p].y;
                lds_barrier();
                C wt = wt0, xlast = C{(R) 0, (R) 0};
                asm volatile("" : "+v"(wt.x), "+v"(wt.y));
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int k3 = brev(p, 4);
                    const R by = theirs[(15 - k3) * 16 * NC];
                    const C w = cmul(wt, C{(R) root64_re(2 * k3), (R) root64_im(2 * k3)});      // W_2L^{256 k1 + col}
                    const R wqx = (R) 0.5 * w.y, wqy = (R) -0.5 * w.x;                            // -(i/2) W_2L^k
                    const R ax = u[p].x, ay = u[p].y;
                    const R sx = ax + bx[p], sy = ay - by, dx = ax - bx[p], dy = ay + by;
                    C xk = C{(R) 0.5 * sx + (dx * wqx - dy * wqy), (R) 0.5 * sy + (dx * wqy + dy * wqx)};
                    if (p == 0 && col0 && t == 0) {                 // k = 0: X[0], X[L] real (dsc_fft.h:221-225)
                        xk = C{ax + ay, (R) 0};
                        xlast = C{ax - ay, (R) 0};
                    }
                    u[p] = xk;
                }
                __builtin_amdgcn_sched_barrier(0);                  // all values final before the first store (see the inverse row task)
                if (col0 && t == 0) st<BS>(xlast, rb, L * CB, 0);
#pragma unroll
                for (int p = 0; p < 16; ++p) st<BS>(u[p], rb, boff, 16 * brev(p, 4) * BSTEP);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // ================= column task backwards: bins -> A (scratch)
            if constexpr (REAL) {
                if (col0 && t == 0) { cur[0].y = (R) 0; }
                R bx[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) mine[e * 16 * NC] = cur[e].x;
                lds_barrier();
#pragma unroll
                for (int e = 0; e < 16; ++e) bx[e] = theirs[(15 - e) * 16 * NC];
                if (col0 && t == 0) bx[0] = cur_last.x;             // bin 0 pairs with bin L (real parts only, dsc_fft.h:227-228)
                lds_barrier();
#pragma unroll
                for (int e = 0; e < 16; ++e) mine[e * 16 * NC] = cur[e].y;
                lds_barrier();
                C wt = wt0;
                asm volatile("" : "+v"(wt.x), "+v"(wt.y));
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    R by = theirs[(15 - e) * 16 * NC];
                    if (e == 0 && col0 && t == 0) by = (R) 0;
                    const C w = cmul(wt, C{(R) root64_re(2 * e), (R) root64_im(2 * e)});
                    const R wqx = (R) 0.5 * w.y, wqy = (R) 0.5 * w.x;                             // (i/2) conj(W_2L^k)
                    const R ax = cur[e].x, ay = cur[e].y;
                    const R sx = ax + bx[e], sy = ay - by, dx = ax - bx[e], dy = ay + by;
                    cur[e] = C{(R) 0.5 * sx + (dx * wqx - dy * wqy), (R) 0.5 * sy + (dx * wqy + dy * wqx)};
                }
                lds_barrier();
            }
            dft16<R, true>(cur);                                    // over k (k1 = t + 16 k) -> t' in cur[brev(t')]
#pragma unroll
            for (int tp = 1; tp < 16; ++tp) cur[brev(tp, 4)] = cmulc(cur[brev(tp, 4)], wtab[W1 * tp * t]);
            {
                R *xw = plane + t * 16 * NC + ell;                  // plane[k1' = t][t'][ell]
                const R *xr = plane + t * NC + ell;                 // thread (ell, t = t') reads every k1'
#pragma unroll
                for (int tp = 0; tp < 16; ++tp) xw[tp * NC] = cur[brev(tp, 4)].x;
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) u[k].x = xr[k * 16 * NC];
                lds_barrier();
#pragma unroll
                for (int tp = 0; tp < 16; ++tp) xw[tp * NC] = cur[brev(tp, 4)].y;
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) u[k].y = xr[k * 16 * NC];
            }
            dft16<R, true>(u);                                      // over k1' -> i in u[brev(i)]: A[t + 16 i][col]
            // A may be written again once it has been read: by this team (its previous row) and, in a pair, by the partner (whose
            // turn lay in between — unless it has run out of rows, which is why the team's own count is checked as well)
            if (it > 0) {
                wait_read(tb + 8, (unsigned) kTS * (unsigned) it);
                if (broken) return;
            }
            if (has_partner) {
                const unsigned need = (unsigned) kTS * (unsigned) ((team & 1) ? it + 1 : it);
                if (need > 0) { wait_read(pb + 8, need); if (broken) return; }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) st<kCached>(u[brev(i, 4)], rwork, woff, i * WSTEP);
            arrive(true, slot);
            spin(slot);
            join();
            if (broken) return;
            // ================= row task backwards: A -> samples
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *) (ext_out + (size_t) row * ext_pitch_b), 0, ext_len_b, 0x00020000);
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = buf_load<kCoherent>(rwork, aoff, k * ASTEP, R{});
            release_scratch();
            four_step_twiddle16<R, true, false, TPL>(v, twL, opaque_j1(), rtau);
            if constexpr (L2 == 128) {
                // k = 2 kr + h: two inverse 8-point DFTs over kr (ka = tau + 8 h), then back through the plane
#pragma unroll
                for (int k = 0; k < 16; ++k) u[8 * (k & 1) + (k >> 1)] = v[k];
                dft8x2<R, true>(u);                                 // u[8 h + p] = value for s = brev(p, 3)
                R *wr = plane + rq * kPQ1 + rtau * kPK1;            // plane[q][ka = tau + 8 h][s]
                const R *rd = plane + wq * kPQ1 + ws;
#pragma unroll
                for (int p = 0; p < 8; ++p) { wr[brev(p, 3)] = u[p].x; wr[8 * kPK1 + brev(p, 3)] = u[8 + p].x; }
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k].x = rd[k * kPK1];
                lds_barrier();
#pragma unroll
                for (int p = 0; p < 8; ++p) { wr[brev(p, 3)] = u[p].y; wr[8 * kPK1 + brev(p, 3)] = u[8 + p].y; }
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k].y = rd[k * kPK1];
#pragma unroll
                for (int k = 0; k < 16; ++k) u[k] = k == 0 ? v[0] : cmulc(v[k], wtab[W2 * ws * k]);      // conj W_128^{s ka}
            } else {
            dft16<R, true>(v);                                      // over k -> s (s') in v[brev(.)]
            if constexpr (L2 == 256) {
#pragma unroll
                for (int s = 1; s < 16; ++s) v[brev(s, 4)] = cmulc(v[brev(s, 4)], wtab[W2 * s * rtau]);
                R *wr = plane + rq * kPQ + rtau * kPK;
                const R *rd = plane + wq * kPQ + ws;
#pragma unroll
                for (int s = 0; s < 16; ++s) wr[s] = v[brev(s, 4)].x;
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) u[k].x = rd[k * kPK];
                lds_barrier();
#pragma unroll
                for (int s = 0; s < 16; ++s) wr[s] = v[brev(s, 4)].y;
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) u[k].y = rd[k * kPK];
            } else {
                // x[s'] = y_0[s'] + conj(W_32^{s'}) y_1[s'], x[s' + 16] = y_0[s'] - ...: this lane holds y_c, the +- is taken by the reader
                const bool hi = rtau >= 16;
                if (hi) {
#pragma unroll
                    for (int m = 1; m < 16; ++m) v[brev(m, 4)] = cmulc(v[brev(m, 4)], C{(R) root64_re(2 * m), (R) root64_im(2 * m)});
                }
                R *wr = plane + rq * kPQ5 + (rtau & 15) * kPK5 + (hi ? 16 : 0);      // plane[q][ka][s' + 16 c]
                const R *rd = plane + wq * kPQ5 + (ws & 15);
                const bool up = ws >= 16;
#pragma unroll
                for (int m = 0; m < 16; ++m) wr[m] = v[brev(m, 4)].x;
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) { const R p0 = rd[k * kPK5], p1 = rd[k * kPK5 + 16]; u[k].x = up ? p0 - p1 : p0 + p1; }
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 16; ++m) wr[m] = v[brev(m, 4)].y;
                lds_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) { const R p0 = rd[k * kPK5], p1 = rd[k * kPK5 + 16]; u[k].y = up ? p0 - p1 : p0 + p1; }
#pragma unroll
                for (int k = 1; k < 16; ++k) u[k] = cmulc(u[k], wtab[W2 * ws * k]);  // conj W_512^{s ka}
            }
            }
            dft16<R, true>(u);                                      // over tau -> m in u[brev(m)]
            // (all 16 values are final before the first store: no store's data registers are rewritten while it drains)
#pragma unroll
            for (int m = 0; m < 16; ++m) u[m] = C{u[m].x * scale, u[m].y * scale};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 16; ++m) st<DSC_FUSED_EXT_STORE>(u[brev(m, 4)], ro, zoff, m * ZSTEP);
            __builtin_amdgcn_sched_barrier(0);
            // the inverse has no registers to spare in its second task (requesting earlier spills 50 of them and costs 30 %)
            if (next < rows) request(cur, cur_last, next);
        }
        PMARK(6);
        // the row after next was published in `slot` at the first barrier of this row
        const int after = __builtin_amdgcn_readfirstlane(info[4 + slot]);
        row = next; next = after;
    }
    leave();
#ifdef DSC_FUSED_PROFILE
    if (tid == 0 && xcc == 0 && team == 0)                  // every member of one team, two rows: who is the barrier waiting for?
        for (int r = 2; r < 4; ++r)
            printf("fused_l2 members team %d rank %d row-iteration %d (10 ns ticks): top %llu | row task done %llu | slot free %llu | A stored, arrived %llu | team complete %llu | A read %llu | column task done %llu\n",
                   team, rank, 20 + r, pstamp[r][0], pstamp[r][1], pstamp[r][2], pstamp[r][3], pstamp[r][4], pstamp[r][5], pstamp[r][6]);
    if (tid == 0 && xcc == 0 && team < 2 && (rank == 0 || rank == kTS - 1))
        for (int r = 0; r < 12; ++r)
            printf("fused_l2 timeline team %d rank %d row-iteration %d (10 ns ticks): top %llu | row task done %llu | slot free %llu | A stored, arrived %llu | team complete %llu | A read %llu | column task done %llu\n",
                   team, rank, 20 + r, pstamp[r][0], pstamp[r][1], pstamp[r][2], pstamp[r][3], pstamp[r][4], pstamp[r][5], pstamp[r][6]);
    if (tid == 0 && xcc == 0 && team == 0 && (rank == 0 || rank == 7))
        printf("fused_l2 profile rank %d (10 ns ticks, sums over the rows of this team): loop %llu | phase-1 compute %llu | wait B %llu | store A + arrive %llu | wait A %llu | read A + request %llu | phase 2 %llu\n",
               rank, pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], pt[6]);
#endif
}

// workgroups per CU the launch asks for, and the rows of scratch that implies (one per possible team, + 1 per XCD of slack for an
// uneven dispatch).  f32 rows of 512 KiB: six teams per XCD (3 MiB of its 4 MiB L2); f64: two workgroups per CU, teams in pairs
// that share a row (1 MiB rows: four teams, 2 MiB: two).
// f32: 512-thread tasks (8192 complex: 32 / 16 lines, 32 columns) — with 256 threads the spectrum pieces of a column task are 64 B
// (measured at L = 65536: rfft 1.48 -> 1.34 ms, irfft 1.59 -> 1.48 ms, fft 1.18 -> 1.15 ms; at 131072 the 256-thread form loses to
// the two-kernel route).  f64: 256 threads (the same pieces in bytes).
constexpr int threads_of(int L, bool single_precision) { (void) L; return single_precision ? 512 : 256; }
constexpr int wg_per_cu(int L, bool single_precision) { (void) L; (void) single_precision; return DSC_FUSED_WG_PER_CU; }     // teams work in pairs (3: measured)
constexpr int team_size_of(int L, bool single_precision) { return (L / 256) / (threads_of(L, single_precision) / 16); }
constexpr int teams_cap_of(int L, bool single_precision) {
    return wg_per_cu(L, single_precision) * 32 / team_size_of(L, single_precision) + 1;       // workgroups per XCD (32 CUs) / team size
}

template<typename R, bool REAL, bool INV, int L2, int NT, bool CAST = false>
bool launch_one(const void *in, void *out, long long rows, void *scratch, unsigned *host_error, const void *tw_full, const void *tw_real, double scale,
                long long ext_pitch_b, int ext_len_b, long long bins_pitch, int bins_len, hipStream_t stream) {
    using C = cpx<R>;
    constexpr int L = 256 * L2, TS = team_size_of(L, sizeof(R) == 4);
    constexpr int cap = teams_cap_of(L, sizeof(R) == 4);
    static_assert(NT == threads_of(L, sizeof(R) == 4), "thread count of this length");
    static int grids[64];                                       // resident launch size per device (0 = not asked yet, -1 = does not fit)
    int dev = 0;
    DSC_KERNEL_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return false;
    if (grids[dev] == 0) {
        int per_cu = 0, cus = 0;
        DSC_KERNEL_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) fused_l2_kernel<R, REAL, INV, L2, NT, CAST>, NT, 0));
        DSC_KERNEL_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int per_cu_raw = per_cu;
        if (per_cu > wg_per_cu(L, sizeof(R) == 4)) per_cu = wg_per_cu(L, sizeof(R) == 4);
        int g = cus * per_cu;
        g -= g % (8 * TS);
        if (g > 8 * TS * (cap - 1)) g = 8 * TS * (cap - 1);
        grids[dev] = g >= 8 * TS ? g : -1;
        if (getenv("DSC_FUSED_VERBOSE")) fprintf(stderr, "fused_l2: L2 = %d, %d threads: occupancy query %d per CU, asked %d, %d CUs -> grid %d (team size %d, teams cap %d)\n", L2, NT, per_cu_raw, wg_per_cu(L, sizeof(R) == 4), cus, grids[dev], TS, cap);
    }
    if (grids[dev] < 0) return false;
    fused_ctl *ctl = (fused_ctl *) scratch;
    C *rowsbuf = (C *) ((char *) scratch + dsc_fft_fused_l2_ctl_bytes());
    DSC_KERNEL_CHECK(hipMemsetAsync(ctl, 0, sizeof(fused_ctl), stream));
    // The kernel's barriers need every workgroup resident.  A COOPERATIVE launch makes the runtime guarantee that (or refuse the launch)
    // instead of leaving it to an occupancy query and to the convention that nothing else runs on the GPU: a second context that
    // holds part of the device then costs the fused route, not the result — the caller falls back to the two-kernel route.
    // (DSC_FUSED_PLAIN_LAUNCH=1: the ordinary launch, for A/B timing.)
    const char *ext_in = INV ? (const char *) nullptr : (const char *) in;
    char *ext_out = INV ? (char *) out : (char *) nullptr;
    const C *bins_in = INV ? (const C *) in : (const C *) nullptr;
    C *bins_out = INV ? (C *) nullptr : (C *) out;
    int rows_i = (int) rows, cap_i = cap;
    static const int cap_env = getenv("DSC_FUSED_TEAMS_CAP") ? atoi(getenv("DSC_FUSED_TEAMS_CAP")) : 0;     // experiment: fewer teams = less live scratch per XCD
    if (cap_env > 0 && cap_env < cap_i) cap_i = cap_env;
    const C *twf = (const C *) tw_full, *twr = (const C *) tw_real;
    R scale_r = (R) scale;
    // Under rocprofv3 the ordinary launch is used as well: with cooperative dispatches in the trace, `rocprofv3 --kernel-trace --stats` on this
    // image (ROCm 7.2.0) writes its files and then dies with SIGSEGV at process exit (gpurun_out/r03fam/fft_c32_65536/trace.err), which
    // loses the per-kernel summaries under profiles/.  The two launches time within 0 - 4 % of each other.
    static const bool plain = getenv("DSC_FUSED_PLAIN_LAUNCH") != nullptr || getenv("ROCPROF_OUTPUT_PATH") != nullptr || getenv("ROCPROFILER_LIBRARY_CTOR") != nullptr;
    if (plain) {
        DSC_LAUNCH((fused_l2_kernel<R, REAL, INV, L2, NT, CAST>), dim3((unsigned) grids[dev]), dim3(NT), 0, stream, ext_in, ext_out, bins_in, bins_out, rowsbuf, ctl,
                   host_error, rows_i, cap_i, twf, twr, scale_r, ext_pitch_b, ext_len_b, bins_pitch, bins_len);
        return true;
    }
    void *args[] = {&ext_in, &ext_out, &bins_in, &bins_out, &rowsbuf, &ctl, &host_error, &rows_i, &cap_i, &twf, &twr, &scale_r, &ext_pitch_b, &ext_len_b, &bins_pitch, &bins_len};
    const hipError_t e = hipLaunchCooperativeKernel((const void *) fused_l2_kernel<R, REAL, INV, L2, NT, CAST>, dim3((unsigned) grids[dev]), dim3(NT), args, 0, stream);
    if (e != hipSuccess) {                                          // refused (too large for what is free, or unsupported): not fatal
        (void) hipGetLastError();
        static bool said = false;
        if (!said) { fprintf(stderr, "dsc: cooperative launch of the fused L2 transform refused (%s); using the two-kernel route\n", hipGetErrorString(e)); said = true; }
        return false;
    }
    return true;
}

}  // namespace

size_t dsc_fft_fused_l2_ctl_bytes() { return (sizeof(fused_ctl) + 4095) / 4096 * 4096; }

// complex length 65536 (256 x 256), f32 and f64; 131072 (256 x 512) in f64 = BASELINE config 5 (its 512-point row tasks read
// 8 adjacent lines: 128-B pieces in f64, only 64 B in f32 — measured 2.21 / 2.00 / 2.02 ms for rfft / irfft / fft against 1.94 /
// 1.78 / 1.51 ms on the two-kernel route, so f32 stays there)
bool dsc_fft_fused_l2_supports(int L, bool single_precision, bool real, bool inverse) {
    if (L == 65536) return true;
    if (L == 32768) return !single_precision;          // f32 has its own one-pass kernels at this length
    (void) real; (void) inverse;
    return L == 131072;                                // f64: config 5 (256-thread tasks); f32: 512-thread tasks (16 lines / 32 columns)
}

// bytes of scratch a launch needs: the control block + one row of A per possible team
size_t dsc_fft_fused_l2_scratch_bytes(int L, bool single_precision) {
    return dsc_fft_fused_l2_ctl_bytes() + (size_t) 8 * teams_cap_of(L, single_precision) * L * (single_precision ? 8 : 16);
}

template<typename R, int L2, int NT = 256>
static bool launch_any(const void *in, void *out, long long rows, bool real, bool inverse, bool cast, void *scratch, unsigned *host_error, const void *tw_full,
                       const void *tw_real, long long in_pitch, int in_len, hipStream_t stream) {
    constexpr long long CBl = 2 * sizeof(R);
    constexpr int L = 256 * L2;
    const double inv_scale = 1.0 / (double) L;                              // dsc_fft.h:232 (2 / 2n) and :168-175
    if (real) {
        if (!inverse) return launch_one<R, true, false, L2, NT>(in, out, rows, scratch, host_error, tw_full, tw_real, 1.0, in_pitch * (long long) sizeof(R), (int) (in_len * sizeof(R)),
                                                            (long long) L + 1, L + 1, stream);
        return launch_one<R, true, true, L2, NT>(in, out, rows, scratch, host_error, tw_full, tw_real, inv_scale, (long long) L * CBl, (int) (L * CBl), in_pitch, in_len, stream);
    }
    if (inverse && cast) return launch_one<R, false, true, L2, NT, true>(in, out, rows, scratch, host_error, tw_full, tw_full, inv_scale, (long long) L * CBl, (int) (L * CBl), in_pitch, in_len, stream);
    if (!inverse && cast) return launch_one<R, false, false, L2, NT, true>(in, out, rows, scratch, host_error, tw_full, tw_full, 1.0, in_pitch * (long long) sizeof(R),
                                                                       (int) (in_len * sizeof(R)), (long long) L, L, stream);
    if (!inverse) return launch_one<R, false, false, L2, NT>(in, out, rows, scratch, host_error, tw_full, tw_full, 1.0, in_pitch * CBl, (int) (in_len * CBl), (long long) L, L, stream);
    return launch_one<R, false, true, L2, NT>(in, out, rows, scratch, host_error, tw_full, tw_full, inv_scale, (long long) L * CBl, (int) (L * CBl), in_pitch, in_len, stream);
}

// Same arguments as dsc_launch_rfft_two_pass / dsc_launch_fft_two_pass (real = packed-real transform; cast = forward complex
// transform of REAL samples: in_pitch / in_len then count reals).  Returns false when the
// launch cannot be made fully resident on this device (the caller falls back to the two-kernel route).
bool dsc_launch_fft_fused_l2(const void *in, void *out, long long rows, int L, bool real, bool inverse, bool cast, bool single_precision, void *scratch,
                             unsigned *host_error, const void *tw_full, const void *tw_real, long long in_pitch, int in_len, hipStream_t stream) {
    if (rows <= 0) return true;
    if (!dsc_fft_fused_l2_supports(L, single_precision, real, inverse) || rows > 0x7fffff00) return false;
    if (L == 131072 && single_precision) return launch_any<float, 512, 512>(in, out, rows, real, inverse, cast, scratch, host_error, tw_full, tw_real, in_pitch, in_len, stream);
    if (L == 131072) return launch_any<double, 512>(in, out, rows, real, inverse, cast, scratch, host_error, tw_full, tw_real, in_pitch, in_len, stream);
    if (L == 32768) return launch_any<double, 128>(in, out, rows, real, inverse, cast, scratch, host_error, tw_full, tw_real, in_pitch, in_len, stream);
    return single_precision ? launch_any<float, 256, 512>(in, out, rows, real, inverse, cast, scratch, host_error, tw_full, tw_real, in_pitch, in_len, stream)
                            : launch_any<double, 256>(in, out, rows, real, inverse, cast, scratch, host_error, tw_full, tw_real, in_pitch, in_len, stream);
}
