// tools/l2probe.hip — round 3: WHEN does a dirty scratch line written and re-read inside one XCD's L2 get written back to HBM?
// The team kernel of fft_xcd_fused.hip moves x1.74 of its algorithmic bytes across the L2 <-> fabric boundary because the four-step
// intermediate A, although read back from the L2, is still written out once per row.  This probe is that kernel's traffic
// skeleton (persistent 256-thread workgroups, 16 x 16 B per thread and task, teams by HW_REG_XCC_ID, L2-local counters) with
// everything that could matter as a parameter:
//   * the size of a team's scratch slot (= row size: team size x 64 KiB) and the number of teams per XCD — i.e. the number of
//     bytes that pass through the L2 between two writes of the same scratch line (the "reuse distance");
//   * the cache-policy bits of the input loads, the output stores, the scratch stores and the scratch loads;
//   * dropping the input stream or the output stream altogether (which of them evicts A?).
// One launch per variant; under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` the n-th dispatch of a probe kernel is the n-th
// variant printed (L2P_PMC=1: exactly one launch per variant).  Every scratch value read is checked (row tag + position).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int T = 256, E = 16;                         // threads, 16-B elements per thread and task
constexpr int kTaskE = T * E;                          // 4096 elements = 64 KiB
constexpr long long kTotalE = (4LL << 30) / 16;        // 4 GiB in, 4 GiB out (config 5)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }
__device__ __forceinline__ void lds_only_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ unsigned l2_read(unsigned *p) {
    unsigned v;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(0u) : "memory");
    return v;
}
__device__ __forceinline__ void l2_add(unsigned *p) { asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(1u) : "memory"); }

__global__ void init_rows(u4 *in, long long total, int row_e) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) in[i] = u4{(unsigned) (i / row_e), (unsigned) (i % row_e), 0x5a5a5a5au, 0u};
}

struct params {
    const u4 *in; u4 *out; u4 *scratch; unsigned *bars, *tickets, *errs;
    int rows, ts, tpx, delay, no_in, no_out, pair, refresh, refresh_kind, frac, bar_mode;
};

// IPOL / OPOL: aux bits of the input loads / output stores; APOL / RPOL: of the scratch stores / loads (1 = sc0, 2 = nt, 16 = sc1)
template<int IPOL, int OPOL, int APOL, int RPOL>
__global__ __launch_bounds__(T, 2) void probe(params p) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x, TS = p.ts;
    if (tid == 0) {
        const unsigned x = xcc_id() & 7u;
        lds[0] = x;
        lds[1] = atomicAdd(&p.tickets[64 * x], 1u);
        lds[2] = 0;
    }
    __syncthreads();
    if ((int) lds[1] >= TS * p.tpx) { if (tid == 0) atomicAdd(p.errs + 1, 1u); return; }      // uneven dispatch: give up
    const int xcc = (int) lds[0], lteam = (int) lds[1] / TS, q = (int) lds[1] % TS;
    const int team = xcc + 8 * lteam;
    const int row_e = kTaskE * TS, row_b = row_e * 16;
    const int F = p.frac, slot_b = TS * T * F * 16;          // only F of a thread's 16 elements go through the scratch slot
    // pair: teams 2k and 2k + 1 of an XCD share one slot (they still run independently here — the probe only wants the footprint)
    u4 *slot = p.scratch + ((size_t) xcc * 64 * (2 << 20) + (size_t) (p.pair ? lteam / 2 : lteam) * slot_b) / 16;     // an XCD's slots are contiguous (2 MiB apart they would share L2 sets)
    unsigned *bar = p.bars + 64 * team;
    unsigned target = 0;
    const int l8 = tid & 7, p8 = tid >> 3, l16 = tid & 15, p16 = tid >> 4;

    u4 cur[E], nxt[E];
    auto request = [&](u4 (&dst)[E], int row) {
        if (p.no_in) {
#pragma unroll
            for (int e = 0; e < E; ++e) dst[e] = u4{(unsigned) row, (unsigned) ((q + TS * (e * (T / 8) + p8)) * 8 + l8), 0x5a5a5a5au, 0u};
            return;
        }
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) (p.in + (size_t) row * row_e), 0, row_b, 0x00020000);
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] = __builtin_amdgcn_raw_buffer_load_b128(r, ((q + TS * (e * (T / 8) + p8)) * 8 + l8) * 16, 0, IPOL);
    };
    // bar_mode 0: everybody adds to ONE counter and polls it with atomics (the team kernel's form); 1: the arrivals go to the counter, the LAST
    // one (seen in its fetch-add's return value) publishes the generation in a flag on another line and everybody polls that flag;
    // 2: as 0 but polling with sc1 loads; 3: as 1, polling with sc1 loads
    auto barrier = [&](bool drain) {
        target += TS;
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_only_barrier();
        if (tid == T - 1) {
            unsigned *flag = bar + 32;
            if (p.bar_mode == 1 || p.bar_mode == 3) {
                unsigned old;
                asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(bar), "v"(1u) : "memory");
                if (old + 1 == target) asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : : "v"(flag), "v"(target) : "memory");
            } else {
                l2_add(bar);
            }
            unsigned *watch = (p.bar_mode == 1 || p.bar_mode == 3) ? flag : bar;
            unsigned spins = 0;
            for (;;) {
                unsigned seen;
                if (p.bar_mode >= 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(watch) : "memory");
                else seen = l2_read(watch);
                if ((int) (seen - target) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 17)) { atomicAdd(p.errs + 2, 1u); lds[2] = 1; break; }
            }
        }
        lds_only_barrier();
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) slot, 0, slot_b, 0x00020000);
    // keep-warm: touch every line of the slot (one dword per 128-B line, two lines per thread; sc1 so that the L1 does not answer)
    unsigned warm = 0;
    auto refresh = [&]() {
        const int off = ((q * T + tid) * 2) * 128;          // (covers the slot only when F == 16)
        if (p.refresh_kind == 0) {
            warm += __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 16);
            warm += __builtin_amdgcn_raw_buffer_load_b32(rs, off, 128, 16);
        } else if (p.refresh_kind == 1) {                 // atomic OR of zero, no return: a read-modify-write in the L2 that changes nothing
            unsigned *a = (unsigned *) slot + off / 4;
            asm volatile("global_atomic_or %0, %1, off\n\tglobal_atomic_or %0, %1, off offset:128" : : "v"(a), "v"(0u) : "memory");
        } else {
            warm += __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 17);
            warm += __builtin_amdgcn_raw_buffer_load_b32(rs, off, 128, 17);
        }
    };
    const int stride = 8 * p.tpx;
    int row = team;
    if (row < p.rows) request(cur, row);
    unsigned bad = 0, sink = 0;
    for (; row < p.rows; row += stride) {
        const int next = row + stride;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int d = 0; d < p.delay; ++d) __builtin_amdgcn_s_sleep(16);
#pragma unroll
        for (int e = 0; e < E; ++e) if (e < F) __builtin_amdgcn_raw_buffer_store_b128(cur[e], rs, (q * T * F + e * T + tid) * 16, 0, APOL);
        barrier(true);
        if (lds[2]) return;
        u4 got[E];
#pragma unroll
        for (int e = 0; e < E; ++e) got[e] = e < F ? __builtin_amdgcn_raw_buffer_load_b128(rs, ((q + TS * (e * (T / 16) + p16)) * 16 + l16) * 16, 0, RPOL) : cur[e];
        if (next < p.rows) request(nxt, next);          // stays in flight across the second barrier
        for (int d = 0; d < p.delay; ++d) __builtin_amdgcn_s_sleep(16);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *) (p.out + (size_t) row * row_e), 0, row_b, 0x00020000);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int s = (q + TS * (e * (T / 16) + p16)) * 16 + l16;
            const int wq = s / (T * F), we = (s % (T * F)) / T, wt = s % T;
            const unsigned j = (unsigned) ((wq + TS * (we * (T / 8) + (wt >> 3))) * 8 + (wt & 7));
            if (e < F) bad += (got[e].x != (unsigned) row) | (got[e].y != j);
            if (p.no_out) sink ^= got[e].z + got[e].w;
            else __builtin_amdgcn_raw_buffer_store_b128(got[e], ro, s * 16, 0, OPOL);
        }
        // R refreshes spread over the time the output stores and the next row's loads take to drain
        if (p.refresh >= 4) { asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); refresh(); }
        if (p.refresh >= 2) { asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); refresh(); }
        if (p.refresh >= 4) { asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); refresh(); }
        if (p.refresh >= 1) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); refresh(); }
        if (p.refresh >= 8) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); refresh(); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); refresh();
                              asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); refresh(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); refresh(); }
        barrier(false);                                  // the scratch loads have landed (their values were used)
        if (lds[2]) return;
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = nxt[e];
    }
    if (bad) atomicAdd(p.errs, bad);
    if (sink == 0x12345u || warm == 0x54321u) p.errs[3] = sink + warm;
}


// K sequential sub-exchanges per row through a slot of 1/K of the row (the "sliced" variant of the team kernel): every workgroup
// writes 16/K of its elements, the team meets, every workgroup reads 16/K elements written by all, the team meets again.
template<int K>
__global__ __launch_bounds__(T, 2) void probe_split(params p) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x, TS = p.ts;
    constexpr int EK = E / K;
    if (tid == 0) {
        const unsigned x = xcc_id() & 7u;
        lds[0] = x;
        lds[1] = atomicAdd(&p.tickets[64 * x], 1u);
        lds[2] = 0;
    }
    __syncthreads();
    if ((int) lds[1] >= TS * p.tpx) { if (tid == 0) atomicAdd(p.errs + 1, 1u); return; }
    const int xcc = (int) lds[0], lteam = (int) lds[1] / TS, q = (int) lds[1] % TS;
    const int team = xcc + 8 * lteam;
    const int row_e = kTaskE * TS, row_b = row_e * 16, slot_b = TS * T * EK * 16;
    u4 *slot = p.scratch + ((size_t) xcc * 64 * (2 << 20) + (size_t) lteam * slot_b) / 16;
    unsigned *bar = p.bars + 64 * team;
    unsigned target = 0;
    const int l8 = tid & 7, p8 = tid >> 3, l16 = tid & 15, p16 = tid >> 4;
    u4 cur[E], nxt[E];
    auto request = [&](u4 (&dst)[E], int row) {
        if (p.no_in) {
#pragma unroll
            for (int e = 0; e < E; ++e) dst[e] = u4{(unsigned) row, (unsigned) ((q + TS * (e * (T / 8) + p8)) * 8 + l8), 0x5a5a5a5au, 0u};
            return;
        }
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) (p.in + (size_t) row * row_e), 0, row_b, 0x00020000);
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] = __builtin_amdgcn_raw_buffer_load_b128(r, ((q + TS * (e * (T / 8) + p8)) * 8 + l8) * 16, 0, 2);
    };
    auto barrier = [&](bool drain) {
        target += TS;
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_only_barrier();
        if (tid == T - 1) {
            l2_add(bar);
            unsigned spins = 0;
            while ((int) (l2_read(bar) - target) < 0) {
                if (p.delay) __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 17)) { atomicAdd(p.errs + 2, 1u); lds[2] = 1; break; }
            }
        }
        lds_only_barrier();
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) slot, 0, slot_b, 0x00020000);
    const int stride = 8 * p.tpx;
    int row = team;
    if (row < p.rows) request(cur, row);
    unsigned bad = 0, sink = 0;
    for (; row < p.rows; row += stride) {
        const int next = row + stride;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *) (p.out + (size_t) row * row_e), 0, row_b, 0x00020000);
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int e = 0; e < EK; ++e) __builtin_amdgcn_raw_buffer_store_b128(cur[k * EK + e], rs, (q * T * EK + e * T + tid) * 16, 0, 0);
            barrier(true);
            if (lds[2]) return;
            u4 got[EK];
#pragma unroll
            for (int e = 0; e < EK; ++e) got[e] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((q + TS * (e * (T / 16) + p16)) * 16 + l16) * 16, 0, 16);
            if (k == 0 && next < p.rows) request(nxt, next);
#pragma unroll
            for (int e = 0; e < EK; ++e) {
                const int sl = (q + TS * (e * (T / 16) + p16)) * 16 + l16;          // position inside the slot
                const int wq = sl / (T * EK), we = (sl % (T * EK)) / T, wt = sl % T;
                const unsigned j = (unsigned) ((wq + TS * ((k * EK + we) * (T / 8) + (wt >> 3))) * 8 + (wt & 7));
                bad += (got[e].x != (unsigned) row) | (got[e].y != j);
                const int so = (q + TS * ((k * EK + e) * (T / 16) + p16)) * 16 + l16;   // position in the output row
                if (p.no_out) sink ^= got[e].z + got[e].w;
                else __builtin_amdgcn_raw_buffer_store_b128(got[e], ro, so * 16, 0, 2);
            }
            barrier(false);
            if (lds[2]) return;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = nxt[e];
    }
    if (bad) atomicAdd(p.errs, bad);
    if (sink == 0x12345u) p.errs[3] = sink;
}


// The sliced exchange as a PIPELINE: a row's scratch passes in K slices through a ring of S slots per team (slot = 1/K row).
// Every workgroup writes its 16/K elements of every slice; slice s is consumed by TS/K of the workgroups (16 elements per thread),
// LAG slices after they wrote it, so that a slice settles in the L2 while its consumers already write the next one.  Counters
// per slot (cumulative): W = workgroups that have written, R = consumers that have read.
template<int K, int S, int LAG>
__global__ __launch_bounds__(T, 2) void probe_pipe(params p) {
    extern __shared__ unsigned lds[];
    static_assert(K % S == 0 && LAG < S, "ring");
    const int tid = threadIdx.x, TS = p.ts;
    constexpr int EK = E / K;
    if (tid == 0) {
        const unsigned x = xcc_id() & 7u;
        lds[0] = x;
        lds[1] = atomicAdd(&p.tickets[64 * x], 1u);
        lds[2] = 0;
    }
    __syncthreads();
    if ((int) lds[1] >= TS * p.tpx) { if (tid == 0) atomicAdd(p.errs + 1, 1u); return; }
    const int xcc = (int) lds[0], lteam = (int) lds[1] / TS, q = (int) lds[1] % TS;
    const int team = xcc + 8 * lteam;
    const int CPS = TS / K, sc = q / CPS, cidx = q % CPS;         // consumers per slice; my slice; my place among its consumers
    const int row_e = kTaskE * TS, row_b = row_e * 16, slot_b = TS * T * EK * 16;
    u4 *ring = p.scratch + ((size_t) xcc * 64 * (2 << 20) + (size_t) lteam * S * slot_b) / 16;
    unsigned *cW = p.bars + 1024 * team, *cR = cW + 512;          // counter of slot i at [32 i]: one 128-B line each
    const int l8 = tid & 7, p8 = tid >> 3, l16 = tid & 15, p16 = tid >> 4;
    u4 cur[E], nxt[E];
    auto request = [&](u4 (&dst)[E], int row) {
        if (p.no_in) {
#pragma unroll
            for (int e = 0; e < E; ++e) dst[e] = u4{(unsigned) row, (unsigned) ((q + TS * (e * (T / 8) + p8)) * 8 + l8), 0x5a5a5a5au, 0u};
            return;
        }
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) (p.in + (size_t) row * row_e), 0, row_b, 0x00020000);
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] = __builtin_amdgcn_raw_buffer_load_b128(r, ((q + TS * (e * (T / 8) + p8)) * 8 + l8) * 16, 0, 2);
    };
    auto wait_for = [&](unsigned *c, unsigned need) {
        if (tid == T - 1) {
            unsigned spins = 0;
            while ((int) (l2_read(c) - need) < 0) {
                if (p.delay) __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 17)) { atomicAdd(p.errs + 2, 1u); lds[2] = 1; break; }
            }
        }
        lds_only_barrier();
    };
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *) ring, 0, S * slot_b, 0x00020000);
    const int stride = 8 * p.tpx;
    int row = team;
    if (row < p.rows) request(cur, row);
    unsigned bad = 0, sink = 0;
    for (int it = 0; row < p.rows; row += stride, ++it) {
        const int next = row + stride;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *) (p.out + (size_t) row * row_e), 0, row_b, 0x00020000);
        auto consume = [&]() {
            const int slot = sc % S;
            const unsigned use = (unsigned) (it * (K / S) + sc / S);
            wait_for(cW + 32 * slot, (use + 1) * (unsigned) TS);
            u4 got[E];
#pragma unroll
            for (int e = 0; e < E; ++e) got[e] = __builtin_amdgcn_raw_buffer_load_b128(rr, slot * slot_b + ((cidx + CPS * (e * (T / 16) + p16)) * 16 + l16) * 16, 0, 16);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int sl = (cidx + CPS * (e * (T / 16) + p16)) * 16 + l16;
                const int wq = sl / (T * EK), we = (sl % (T * EK)) / T, wt = sl % T;
                const unsigned j = (unsigned) ((wq + TS * ((sc * EK + we) * (T / 8) + (wt >> 3))) * 8 + (wt & 7));
                bad += (got[e].x != (unsigned) row) | (got[e].y != j);
            }
            // (the loads have landed: their values were compared)
            lds_only_barrier();
            if (tid == T - 1) l2_add(cR + 32 * slot);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int so = (q + TS * (e * (T / 16) + p16)) * 16 + l16;
                if (p.no_out) sink ^= got[e].z + got[e].w;
                else __builtin_amdgcn_raw_buffer_store_b128(got[e], ro, so * 16, 0, 2);
            }
        };
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int slot = s % S;
            const unsigned use = (unsigned) (it * (K / S) + s / S);
            if (it > 0 || s >= S) wait_for(cR + 32 * slot, use * (unsigned) CPS);
            if (lds[2]) return;
#pragma unroll
            for (int e = 0; e < EK; ++e) __builtin_amdgcn_raw_buffer_store_b128(cur[s * EK + e], rr, slot * slot_b + (q * T * EK + e * T + tid) * 16, 0, 0);
            if (s == K - 1 && next < p.rows) request(nxt, next);
            // the stores of this slice are in the L2 (the next row's loads, issued after them, may still be in flight)
            if (s == K - 1 && next < p.rows && !p.no_in) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(E) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_only_barrier();
            if (tid == T - 1) l2_add(cW + 32 * slot);
            if (s == (sc + LAG < K - 1 ? sc + LAG : K - 1)) { consume(); if (lds[2]) return; }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = nxt[e];
    }
    if (bad) atomicAdd(p.errs, bad);
    if (sink == 0x12345u) p.errs[3] = sink;
}

struct state { u4 *in, *out, *scratch; unsigned *bars, *tickets, *errs; int row_e; bool pmc; };

template<int IPOL, int OPOL, int APOL, int RPOL>
void run(state &s, int ts, int tpx, const char *pol, int no_in = 0, int no_out = 0, int pair = 0, int delay = 0, int refresh = 0, int rkind = 0, int frac = 16, int bar_mode = 0) {
    const int row_e = kTaskE * ts;
    if (s.row_e != row_e) {
        hipLaunchKernelGGL(init_rows, dim3((unsigned) (kTotalE / 256)), dim3(256), 0, 0, s.in, kTotalE, row_e);
        s.row_e = row_e;
    }
    params p{s.in, s.out, s.scratch, s.bars, s.tickets, s.errs, (int) (kTotalE / row_e), ts, tpx, delay, no_in, no_out, pair, refresh, rkind, frac, bar_mode};
    const int wg_per_xcd = ts * tpx, wpc = (wg_per_xcd + 31) / 32;
    const int lds = wpc == 1 ? 96 * 1024 : 48 * 1024;
    auto k = probe<IPOL, OPOL, APOL, RPOL>;
    CK(hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&c));
    float best = 1e30f;
    for (int r = 0; r < (s.pmc ? 1 : 4); ++r) {
        CK(hipMemsetAsync(s.bars, 0, 256 * 64 * 4, 0));
        CK(hipMemsetAsync(s.tickets, 0, 8 * 64 * 4, 0));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(8 * wg_per_xcd), dim3(T), lds, 0, p);
        CK(hipGetLastError());
        CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
        float ms; CK(hipEventElapsedTime(&ms, a, c)); if (ms < best) best = ms;
    }
    unsigned errs[4];
    CK(hipMemcpy(errs, s.errs, sizeof errs, hipMemcpyDeviceToHost));
    CK(hipMemset(s.errs, 0, sizeof errs));
    static int n = 0;
    // bytes through the XCD's L2 between two writes of one scratch line: every team's in + out + scratch of one row (a pair shares)
    const double slot_kib = ts * 64.0 * frac / 16, reuse_mib = tpx * (ts * 64.0 * (2 - no_in - no_out) + slot_kib) / 1024.0;
    printf("variant %2d: frac %2d/16 slot %5.0f KiB  teams/XCD %2d%s  between rewrites %5.2f MiB  pol %-22s%s%s refresh %d%s barrier %d delay %d  %8.3f ms  %6.0f GB/s  stale %u uneven %u timeouts %u\n", n++, frac,
           slot_kib, tpx, pair ? " (pairs share)" : "", reuse_mib, pol, no_in ? " NO-IN" : "", no_out ? " NO-OUT" : "", refresh, rkind == 1 ? " (atomic or 0)" : rkind == 2 ? " (sc0 sc1 loads)" : "", bar_mode, delay, best, 2.0 * kTotalE * 16 / best / 1e6, errs[0],
           errs[1], errs[2]);
    fflush(stdout);
}

template<int K>
void run_split(state &s, int ts, int tpx, int no_in = 0, int no_out = 0, int poll_sleep = 1) {
    const int row_e = kTaskE * ts;
    if (s.row_e != row_e) {
        hipLaunchKernelGGL(init_rows, dim3((unsigned) (kTotalE / 256)), dim3(256), 0, 0, s.in, kTotalE, row_e);
        s.row_e = row_e;
    }
    params p{s.in, s.out, s.scratch, s.bars, s.tickets, s.errs, (int) (kTotalE / row_e), ts, tpx, poll_sleep, no_in, no_out, 0, 0, 0, 16};
    const int wg_per_xcd = ts * tpx, wpc = (wg_per_xcd + 31) / 32;
    const int lds = wpc == 1 ? 96 * 1024 : 48 * 1024;
    auto k = probe_split<K>;
    CK(hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&c));
    float best = 1e30f;
    for (int r = 0; r < (s.pmc ? 1 : 4); ++r) {
        CK(hipMemsetAsync(s.bars, 0, 256 * 64 * 4, 0));
        CK(hipMemsetAsync(s.tickets, 0, 8 * 64 * 4, 0));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(8 * wg_per_xcd), dim3(T), lds, 0, p);
        CK(hipGetLastError());
        CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
        float ms; CK(hipEventElapsedTime(&ms, a, c)); if (ms < best) best = ms;
    }
    unsigned errs[4];
    CK(hipMemcpy(errs, s.errs, sizeof errs, hipMemcpyDeviceToHost));
    CK(hipMemset(s.errs, 0, sizeof errs));
    static int n = 0;
    const double slot_kib = ts * 64.0 / K;
    const double cycles = (double) p.rows / (8 * tpx) * K;           // slot cycles per team
    printf("variant %2d: split K=%d slot %5.0f KiB  teams/XCD %2d  live %5.2f MiB%s%s poll-sleep %d delay 0  %8.3f ms  %6.0f GB/s  %5.2f us per slot cycle  stale %u uneven %u timeouts %u\n", n++, K,
           slot_kib, tpx, tpx * slot_kib / 1024.0, no_in ? " NO-IN" : "", no_out ? " NO-OUT" : "", poll_sleep, best, 2.0 * kTotalE * 16 / best / 1e6, best * 1000.0 / cycles, errs[0], errs[1], errs[2]);
    fflush(stdout);
}

template<int K, int S, int LAG>
void run_pipe(state &s, int ts, int tpx, int no_in = 0, int no_out = 0, int poll_sleep = 0) {
    const int row_e = kTaskE * ts;
    if (s.row_e != row_e) {
        hipLaunchKernelGGL(init_rows, dim3((unsigned) (kTotalE / 256)), dim3(256), 0, 0, s.in, kTotalE, row_e);
        s.row_e = row_e;
    }
    params p{s.in, s.out, s.scratch, s.bars, s.tickets, s.errs, (int) (kTotalE / row_e), ts, tpx, poll_sleep, no_in, no_out, 0, 0, 0, 16};
    const int wg_per_xcd = ts * tpx, wpc = (wg_per_xcd + 31) / 32;
    const int lds = wpc == 1 ? 96 * 1024 : 48 * 1024;
    auto k = probe_pipe<K, S, LAG>;
    CK(hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&c));
    float best = 1e30f;
    for (int r = 0; r < (s.pmc ? 1 : 4); ++r) {
        CK(hipMemsetAsync(s.bars, 0, 64 * 1024 * 4, 0));
        CK(hipMemsetAsync(s.tickets, 0, 8 * 64 * 4, 0));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(8 * wg_per_xcd), dim3(T), lds, 0, p);
        CK(hipGetLastError());
        CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
        float ms; CK(hipEventElapsedTime(&ms, a, c)); if (ms < best) best = ms;
    }
    unsigned errs[4];
    CK(hipMemcpy(errs, s.errs, sizeof errs, hipMemcpyDeviceToHost));
    CK(hipMemset(s.errs, 0, sizeof errs));
    static int n = 0;
    const double slot_kib = ts * 64.0 / K;
    printf("variant %2d: pipe K=%d S=%d lag %d slot %5.0f KiB  teams/XCD %2d  live %5.2f MiB%s%s poll-sleep %d delay 0  %8.3f ms  %6.0f GB/s  stale %u uneven %u timeouts %u\n", n++, K, S, LAG,
           slot_kib, tpx, tpx * S * slot_kib / 1024.0, no_in ? " NO-IN" : "", no_out ? " NO-OUT" : "", poll_sleep, best, 2.0 * kTotalE * 16 / best / 1e6, errs[0], errs[1], errs[2]);
    fflush(stdout);
}

int main() {
    state s;
    s.pmc = getenv("L2P_PMC") != nullptr;
    CK(hipMalloc(&s.in, (size_t) kTotalE * 16)); CK(hipMalloc(&s.out, (size_t) kTotalE * 16));
    CK(hipMalloc(&s.scratch, (size_t) 8 * 64 * (2 << 20)));
    CK(hipMalloc(&s.bars, 64 * 1024 * 4)); CK(hipMalloc(&s.tickets, 8 * 64 * 4)); CK(hipMalloc(&s.errs, 16));
    CK(hipMemset(s.errs, 0, 16));
    s.row_e = 0;
    const char *only = getenv("L2P_ONLY");
    auto want = [&](const char *tag) { return only == nullptr || strstr(only, tag) != nullptr; };

    if (want("E1")) {
        printf("# E1: slot size x teams per XCD (in: nt loads, out: nt stores, scratch: plain stores, sc1 loads)\n");
        for (int ts : {2, 4, 8, 16, 32})
            for (int tpx : {1, 2, 4, 8, 16, 32}) {
                if (ts * tpx > 64 || ts * tpx < 8) continue;
                run<2, 2, 0, 16>(s, ts, tpx, "in nt / out nt");
            }
    }
    if (want("E2")) {
        printf("# E2: policies, slot 2 MiB, 1 and 2 teams per XCD; then the 512 KiB slot with 4 teams\n");
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int ts = cfg == 2 ? 8 : 32, tpx = cfg == 0 ? 1 : cfg == 1 ? 2 : 4;
            run<2, 2, 0, 16>(s, ts, tpx, "in nt / out nt");
            run<0, 0, 0, 16>(s, ts, tpx, "in - / out -");
            run<2, 0, 0, 16>(s, ts, tpx, "in nt / out -");
            run<0, 2, 0, 16>(s, ts, tpx, "in - / out nt");
            run<17, 17, 0, 16>(s, ts, tpx, "in sc0sc1 / out sc0sc1");
            run<19, 19, 0, 16>(s, ts, tpx, "in,out sc0sc1nt");
            run<16, 16, 0, 16>(s, ts, tpx, "in sc1 / out sc1");
            run<18, 18, 0, 16>(s, ts, tpx, "in,out sc1nt");
            run<2, 19, 0, 16>(s, ts, tpx, "in nt / out sc0sc1nt");
            run<2, 17, 0, 16>(s, ts, tpx, "in nt / out sc0sc1");
            run<19, 2, 0, 16>(s, ts, tpx, "in sc0sc1nt / out nt");
            run<3, 3, 0, 16>(s, ts, tpx, "in,out sc0nt");
            run<2, 2, 0, 0>(s, ts, tpx, "nt, scratch loads plain");
            run<2, 2, 1, 16>(s, ts, tpx, "nt, scratch stores sc0");
            run<2, 2, 0, 16>(s, ts, tpx, "in nt / out nt", 1, 0);
            run<2, 2, 0, 16>(s, ts, tpx, "in nt / out nt", 0, 1);
            run<2, 2, 0, 16>(s, ts, tpx, "in nt / out nt", 1, 1);
            run<0, 0, 0, 16>(s, ts, tpx, "in - / out -", 1, 0);
            run<0, 0, 0, 16>(s, ts, tpx, "in - / out -", 0, 1);
        }
    }
    if (want("E3")) {
        printf("# E3: pairs sharing a slot (the fused kernel's arrangement)\n");
        run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 0, 0, 1);
        run<2, 2, 0, 16>(s, 16, 4, "in nt / out nt", 0, 0, 1);
        run<2, 2, 0, 16>(s, 8, 8, "in nt / out nt", 0, 0, 1);
        run<2, 2, 0, 16>(s, 4, 16, "in nt / out nt", 0, 0, 1);
    }
    if (want("E4")) {
        printf("# E4: keep-warm touches of the slot between its read and its next write (R rounds of one dword per line)\n");
        for (int r : {0, 1, 2, 4, 8}) {
            run<2, 2, 0, 16>(s, 32, 1, "in nt / out nt", 0, 0, 0, 0, r);
            run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 0, 0, 1, 0, r);
            run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 0, 0, 0, 0, r);
            run<2, 2, 0, 16>(s, 16, 4, "in nt / out nt", 0, 0, 1, 0, r);
            run<2, 2, 0, 16>(s, 8, 8, "in nt / out nt", 0, 0, 1, 0, r);
            run<2, 2, 0, 16>(s, 8, 4, "in nt / out nt", 0, 0, 0, 0, r);
            run<2, 2, 0, 16>(s, 8, 8, "in nt / out nt", 0, 0, 0, 0, r);
        }
    }
    if (want("E5")) {
        printf("# E5: keep-warm with L2 atomics (or 0) / sc0 sc1 loads instead of sc1 loads\n");
        for (int kind : {1, 2})
            for (int r : {1, 2, 4, 8}) {
                run<2, 2, 0, 16>(s, 32, 1, "in nt / out nt", 0, 0, 0, 0, r, kind);
                run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 0, 0, 1, 0, r, kind);
                run<2, 2, 0, 16>(s, 8, 4, "in nt / out nt", 0, 0, 0, 0, r, kind);
            }
    }
    if (want("E6")) {
        printf("# E6: only F of 16 elements per thread pass through the scratch slot (small live scratch, full streams): does a SMALL slot survive a LARGE stream?\n");
        for (int f : {1, 2, 4, 8, 16}) {
            run<2, 2, 0, 16>(s, 32, 1, "in nt / out nt", 0, 0, 0, 0, 0, 0, f);
            run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 0, 0, 0, 0, 0, 0, f);
            run<2, 2, 0, 16>(s, 16, 4, "in nt / out nt", 0, 0, 0, 0, 0, 0, f);
            run<2, 2, 0, 16>(s, 8, 8, "in nt / out nt", 0, 0, 0, 0, 0, 0, f);
            run<0, 0, 0, 16>(s, 32, 2, "in - / out -", 0, 0, 0, 0, 0, 0, f);
            run<2, 0, 0, 16>(s, 32, 2, "in nt / out -", 0, 0, 0, 0, 0, 0, f);
        }
    }
    if (want("E7")) {
        printf("# E7a: write-back vs LIVE scratch per XCD between 1 and 2 MiB (2 teams of 32, F of 16 elements through the slot)\n");
        for (int f : {4, 5, 6, 7, 8}) run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 0, 0, 0, 0, 0, 0, f);
        for (int f : {10, 12, 14}) run<2, 2, 0, 16>(s, 32, 1, "in nt / out nt", 0, 0, 0, 0, 0, 0, f);
        printf("# E7b: K sequential sub-exchanges per row through a slot of 1/K row: time per slot cycle, and what the streams add\n");
        for (int no : {1, 0}) {
            run_split<1>(s, 32, 1, no, no); run_split<2>(s, 32, 1, no, no); run_split<4>(s, 32, 1, no, no); run_split<8>(s, 32, 1, no, no); run_split<16>(s, 32, 1, no, no);
            run_split<1>(s, 32, 2, no, no); run_split<2>(s, 32, 2, no, no); run_split<4>(s, 32, 2, no, no); run_split<8>(s, 32, 2, no, no);
            run_split<1>(s, 16, 4, no, no); run_split<2>(s, 16, 4, no, no); run_split<4>(s, 16, 4, no, no); run_split<8>(s, 16, 4, no, no);
            run_split<1>(s, 8, 8, no, no); run_split<2>(s, 8, 8, no, no); run_split<4>(s, 8, 8, no, no);
        }
        run_split<2>(s, 32, 2, 1, 1, 0); run_split<4>(s, 32, 2, 1, 1, 0); run_split<2>(s, 32, 2, 0, 0, 0); run_split<4>(s, 32, 2, 0, 0, 0);
    }
    if (want("E9")) {
        printf("# E9: the team barrier itself: 0 = one counter, atomic polls; 1 = last arrival publishes a flag, atomic polls; 2 / 3 = the same with sc1-load polls\n");
        for (int bm : {0, 1, 2, 3}) {
            run<2, 2, 0, 16>(s, 32, 1, "in nt / out nt", 1, 1, 0, 0, 0, 0, 16, bm);
            run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 1, 1, 1, 0, 0, 0, 16, bm);
            run<2, 2, 0, 16>(s, 32, 2, "in nt / out nt", 0, 0, 1, 0, 0, 0, 16, bm);
            run<2, 2, 0, 16>(s, 8, 8, "in nt / out nt", 0, 0, 1, 0, 0, 0, 16, bm);
        }
    }
    if (want("E8")) {
        printf("# E8: the sliced exchange as a pipeline (K slices through a ring of S slots per team, consumers LAG slices behind)\n");
        for (int no : {1, 0}) {
            run_pipe<4, 1, 0>(s, 32, 2, no, no);
            run_pipe<4, 2, 0>(s, 32, 2, no, no); run_pipe<4, 2, 1>(s, 32, 2, no, no);
            run_pipe<8, 2, 0>(s, 32, 2, no, no); run_pipe<8, 2, 1>(s, 32, 2, no, no);
            run_pipe<8, 4, 1>(s, 32, 2, no, no); run_pipe<8, 4, 2>(s, 32, 2, no, no); run_pipe<8, 4, 3>(s, 32, 2, no, no);
            run_pipe<16, 4, 1>(s, 32, 2, no, no); run_pipe<16, 4, 3>(s, 32, 2, no, no); run_pipe<16, 8, 3>(s, 32, 2, no, no); run_pipe<16, 8, 7>(s, 32, 2, no, no);
            run_pipe<8, 2, 1>(s, 32, 1, no, no); run_pipe<8, 4, 3>(s, 32, 1, no, no); run_pipe<16, 8, 7>(s, 32, 1, no, no);
            run_pipe<4, 2, 1>(s, 16, 4, no, no); run_pipe<8, 2, 1>(s, 16, 4, no, no); run_pipe<8, 4, 3>(s, 16, 4, no, no);
            run_pipe<4, 2, 1>(s, 8, 8, no, no); run_pipe<8, 2, 1>(s, 8, 8, no, no);
        }
    }
    return 0;
}
