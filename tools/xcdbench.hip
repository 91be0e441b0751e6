// tools/xcdbench.hip — can the intermediate of a two-pass transform stay inside ONE XCD's 4 MiB L2?
// Traffic skeleton of a fused four-step kernel: persistent workgroups, one TEAM per XCD (workgroups with the same
// HW_REG_XCC_ID), a team walks one row (2 MiB in, 2 MiB intermediate, 2 MiB out) at a time:
//   phase 1  every member reads its share of the row (128-B pieces, strided like the j1 groups of a rows pass) and writes it
//            to the team's scratch row (contiguous runs)
//   team barrier (counter in the XCD's own L2: plain atomics, no agent-scope fence, hence no L2 write-back / invalidate)
//   phase 2  every member reads 256-B pieces of the scratch row written by ALL members (loads that bypass the L1) and
//            streams them out
//   team barrier (or a second scratch row instead)
// The kernel checks every intermediate value it reads (row tag + position), so stale lines show up as errors.
// Prints ms for 2048 rows of 2 MiB (config 5's shape) per variant; the no-scratch copy with the same access shapes is the
// reference.  Every spin is bounded: a team that never completes sets the error flag and leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kRowB2M = 2 << 20;             // bytes per row (in, scratch and out): config 5
constexpr int kTotalE = 2048 * (kRowB2M / 16); // 16-B elements of the whole input

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void lds_only_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned l2_counter_read(unsigned *p) {
    unsigned v;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(0u) : "memory");
    return v;
}
__device__ __forceinline__ void l2_counter_add(unsigned *p) {
    asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(1u) : "memory");
}

__global__ void init_rows(u4 *in, long long total, int row_e) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) in[i] = u4{(unsigned) (i / row_e), (unsigned) (i % row_e), 0x5a5a5a5au, 0u};
}

__global__ void xcc_map(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// T threads, E elements of 16 B per thread and task.  SCR: 0 = no scratch (copy, no barriers), 1 = one scratch row + two
// barriers per row, 2 = two scratch rows + one barrier per row.  PF: request the next row before waiting at the barrier.
// SPOL: aux bits of the scratch loads (16 = sc1: device scope, misses the L1).
template<int T, int E, int SCR, bool PF, int SPOL, int TPX, int kRowB, int WPC>
__global__ __launch_bounds__(T, (WPC * T / 256)) void team_walk(const u4 *__restrict__ in, u4 *__restrict__ out, u4 *scratch, unsigned *bars, unsigned *team_ids,
                                               unsigned *errs, int rows, int delay, unsigned long long *stamps) {
    extern __shared__ unsigned lds[];
    constexpr int kRowE = kRowB / 16;
    const int tid = threadIdx.x;
    // team = XCC the workgroup runs on; rank = order of arrival
    if (tid == 0) {
        const unsigned x = xcc_id();
        lds[0] = x;
        lds[1] = atomicAdd(&team_ids[x], 1u);
        lds[2] = 0;
    }
    __syncthreads();
    const int TS = gridDim.x / (8 * TPX);               // team size the host assumed: E * T * TS = kRowE
    if ((int) lds[1] >= TS * TPX) { if (tid == 0) atomicAdd(errs + 1, 1u); return; }          // uneven dispatch: this probe gives up
    const int team = (int) lds[0] + 8 * ((int) lds[1] / TS), q = (int) lds[1] % TS;        // TPX teams per XCD
    unsigned *bar = bars + 64 * team;
    unsigned target = 0;
    u4 *scr = scratch + (size_t) team * 2 * kRowE;

    const int l8 = tid & 7, p8 = tid >> 3;             // phase-1 read pieces: 8 lanes x 16 B
    const int l16 = tid & 15, p16 = tid >> 4;          // phase-2 pieces: 16 lanes x 16 B
    u4 cur[E], nxt[E];
    auto request = [&](u4 (&dst)[E], int row) {
        const u4 *src = in + (size_t) row * kRowE;
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] = __builtin_nontemporal_load(src + ((q + TS * (e * (T / 8) + p8)) * 8 + l8));
    };
    auto barrier = [&]() {
        target += TS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_only_barrier();
        if (tid == T - 1) {
            l2_counter_add(bar);
            unsigned spins = 0;
            while ((int) (l2_counter_read(bar) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 16)) { atomicAdd(errs + 2, 1u); lds[2] = 1; break; }
            }
        }
    };
    int row = team;
    if (tid == 0) stamps[2 * (team * 64 + q)] = wall_clock64();
    if (row < rows) request(cur, row);
    int it = 0;
    for (; row < rows; row += 8 * TPX, ++it) {
        const int next = row + 8 * TPX;
        u4 *out_row = out + (size_t) row * kRowE;
        if constexpr (SCR == 0) {
            if (PF && next < rows) request(nxt, next);
#pragma unroll
            for (int e = 0; e < E; ++e) __builtin_nontemporal_store(cur[e], out_row + ((q + TS * (e * (T / 16) + p16)) * 16 + l16));
            if (!PF && next < rows) request(nxt, next);
        } else {
            u4 *s = scr + (SCR == 2 ? (it & 1) * kRowE : 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16);      // stands in for the first transform (~0.4 us a step)
#pragma unroll
            for (int e = 0; e < E; ++e) s[q * (T * E) + e * T + tid] = cur[e];
            barrier();
            if (PF && next < rows) request(nxt, next);
            lds_only_barrier();
            if (lds[2]) return;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) s, 0, kRowB, 0x00020000);
            u4 got[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int slot = (q + TS * (e * (T / 16) + p16)) * 16 + l16;
                got[e] = __builtin_amdgcn_raw_buffer_load_b128(rs, slot * 16, 0, SPOL);
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PF ? E : 0) : "memory");
            for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(16);      // the second transform
            unsigned bad = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int slot = (q + TS * (e * (T / 16) + p16)) * 16 + l16;
                const int wq = slot / (T * E), we = (slot % (T * E)) / T, wt = slot % T;
                const unsigned j = (unsigned) ((wq + TS * (we * (T / 8) + (wt >> 3))) * 8 + (wt & 7));
                bad += (got[e].x != (unsigned) row) | (got[e].y != j);
                __builtin_nontemporal_store(got[e], out_row + slot);
            }
            if (bad) atomicAdd(errs, bad);
            if constexpr (SCR == 1) { barrier(); lds_only_barrier(); if (lds[2]) return; }
            if (!PF && next < rows) request(nxt, next);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = nxt[e];
    }
    if (tid == 0) stamps[2 * (team * 64 + q) + 1] = wall_clock64();
}

struct bufs { u4 *in, *out, *scratch; unsigned *bars, *team_ids, *errs; int rows; unsigned long long *stamps; int row_e; };

template<int T, int E, int SCR, bool PF, int SPOL, int TPX, int kRowB, int WPC>
void run_wpc(bufs &b, const char *name, int delay) {
    const int wg_per_cu = WPC;

    constexpr int kRowE = kRowB / 16;
    const int TS = kRowE / (T * E), grid = 8 * TS * TPX;
    if (b.row_e != kRowE) {
        hipLaunchKernelGGL(init_rows, dim3((unsigned) (kTotalE / 256)), dim3(256), 0, 0, b.in, (long long) kTotalE, kRowE);
        b.row_e = kRowE; b.rows = kTotalE / kRowE;
    }
    const int lds = wg_per_cu == 1 ? 96 * 1024 : wg_per_cu == 2 ? 48 * 1024 : 36 * 1024;     // keeps the residency at wg_per_cu
    auto k = team_walk<T, E, SCR, PF, SPOL, TPX, kRowB, WPC>;
    CK(hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&c));
    float best = 1e30f;
    unsigned errs[4] = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) {
        CK(hipMemsetAsync(b.bars, 0, 64 * 64 * 4, 0));
        CK(hipMemsetAsync(b.team_ids, 0, 16 * 4, 0));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(grid), dim3(T), lds, 0, b.in, b.out, b.scratch, b.bars, b.team_ids, b.errs, b.rows, delay, b.stamps);
        CK(hipGetLastError());
        CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
        float ms; CK(hipEventElapsedTime(&ms, a, c)); if (ms < best) best = ms;
    }
    CK(hipMemcpy(errs, b.errs, sizeof errs, hipMemcpyDeviceToHost));
    CK(hipMemset(b.errs, 0, sizeof errs));
    std::vector<unsigned long long> st(2 * 64 * 64);
    CK(hipMemcpy(st.data(), b.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int t = 0; t < 8 * TPX; ++t) for (int q = 0; q < TS; ++q) if (st[2 * (t * 64 + q)] < t0) t0 = st[2 * (t * 64 + q)];
    char spans[1024]; int n = 0;
    for (int t = 0; t < 8 * TPX; t += 8) {          // teams on XCD 0
        unsigned long long lo = ~0ull, hi = 0;
        for (int q = 0; q < TS; ++q) { lo = st[2 * (t * 64 + q)] < lo ? st[2 * (t * 64 + q)] : lo; hi = st[2 * (t * 64 + q) + 1] > hi ? st[2 * (t * 64 + q) + 1] : hi; }
        n += snprintf(spans + n, sizeof spans - n, " team %d: %.0f..%.0f us", t, (lo - t0) / 100.0, (hi - t0) / 100.0);
    }
    printf("%-58s delay %d grid %4d x %4d  %8.3f ms  %6.0f GB/s (in+out)  stale %u  uneven %u  timeouts %u\n", name, delay, grid, T, best,
           2.0 * kTotalE * 16 / best / 1e6, errs[0], errs[1], errs[2]);
    printf("      %s\n", spans);
    fflush(stdout);
}

template<int T, int E, int SCR, bool PF, int SPOL, int TPX = 1, int kRowB = kRowB2M>
void run(bufs &b, int wg_per_cu, const char *name, int delay = 0) {
    if (wg_per_cu == 1) run_wpc<T, E, SCR, PF, SPOL, TPX, kRowB, 1>(b, name, delay);
    else if (wg_per_cu == 2) run_wpc<T, E, SCR, PF, SPOL, TPX, kRowB, 2>(b, name, delay);
    else run_wpc<T, E, SCR, PF, SPOL, TPX, kRowB, 3>(b, name, delay);
}

int main() {
    bufs b;
    b.rows = 2048;
    CK(hipMalloc(&b.in, (size_t) kTotalE * 16)); CK(hipMalloc(&b.out, (size_t) kTotalE * 16));
    CK(hipMalloc(&b.scratch, (size_t) 64 * 2 * kRowB2M));
    CK(hipMalloc(&b.bars, 64 * 64 * 4)); CK(hipMalloc(&b.team_ids, 16 * 4)); CK(hipMalloc(&b.errs, 16));
    CK(hipMemset(b.errs, 0, 16));
    CK(hipMalloc(&b.stamps, 2 * 64 * 64 * 8)); CK(hipMemset(b.stamps, 0, 2 * 64 * 64 * 8));
    b.row_e = 0;

    unsigned *map; CK(hipMalloc(&map, 512 * 4));
    hipLaunchKernelGGL(xcc_map, dim3(512), dim3(64), 0, 0, map);
    std::vector<unsigned> h(512); CK(hipMemcpy(h.data(), map, 512 * 4, hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..31:");
    for (int i = 0; i < 32; ++i) printf(" %u", h[i]);
    int rr = 0; for (int i = 0; i < 512; ++i) rr += h[i] == (unsigned) (i % 8);
    printf("\nworkgroups with XCC_ID == blockIdx %% 8: %d of 512\n", rr);
    fflush(stdout);

    if (!getenv("XCD_ONLY_SMALL")) {
    run<256, 16, 0, true, 0>(b, 1, "copy, no scratch, 256 thr x 16, prefetch");
    run<512, 8, 0, true, 0>(b, 1, "copy, no scratch, 512 thr x 8, prefetch");
    run<1024, 4, 0, true, 0>(b, 1, "copy, no scratch, 1024 thr x 4, prefetch");
    run<512, 4, 0, true, 0>(b, 2, "copy, no scratch, 2 x (512 thr x 4) per CU, prefetch");
    run<512, 8, 1, false, 16>(b, 1, "1 scratch row, 2 barriers, 512 x 8, no prefetch, sc1");
    run<512, 8, 1, true, 16>(b, 1, "1 scratch row, 2 barriers, 512 x 8, prefetch, sc1");
    run<256, 16, 1, true, 16>(b, 1, "1 scratch row, 2 barriers, 256 x 16, prefetch, sc1");
    run<1024, 4, 1, true, 16>(b, 1, "1 scratch row, 2 barriers, 1024 x 4, prefetch, sc1");
    run<512, 8, 2, true, 16>(b, 1, "2 scratch rows, 1 barrier, 512 x 8, prefetch, sc1");
    run<1024, 4, 2, true, 16>(b, 1, "2 scratch rows, 1 barrier, 1024 x 4, prefetch, sc1");
    run<512, 4, 1, true, 16>(b, 2, "1 scratch row, 2 barriers, 2 x (512 x 4) per CU, prefetch, sc1");
    run<512, 4, 2, true, 16>(b, 2, "2 scratch rows, 1 barrier, 2 x (512 x 4) per CU, prefetch, sc1");
    run<256, 16, 1, true, 16, 2>(b, 2, "2 teams per XCD (2 wg per CU): 1 scratch row each, 256 x 16");
    run<512, 8, 1, true, 16, 2>(b, 2, "2 teams per XCD (2 wg per CU): 1 scratch row each, 512 x 8");
    run<256, 16, 0, true, 0, 2>(b, 2, "copy, no scratch, 2 teams per XCD, 256 x 16");
    run<256, 8, 1, true, 16, 1>(b, 2, "1 team of 64 per XCD (2 wg per CU), 256 x 8");
    for (int delay : {2, 4, 8}) {
        run<256, 16, 1, true, 16, 1>(b, 1, "1 team per XCD, 256 x 16", delay);
        run<512, 8, 1, true, 16, 1>(b, 1, "1 team per XCD, 512 x 8", delay);
        run<256, 16, 1, true, 16, 2>(b, 2, "2 teams per XCD, 256 x 16", delay);
        run<512, 8, 1, true, 16, 2>(b, 2, "2 teams per XCD, 512 x 8", delay);
    }
    run<512, 8, 1, true, 1>(b, 1, "1 scratch row, 2 barriers, 512 x 8, prefetch, sc0 only");
    run<512, 8, 1, true, 0>(b, 1, "1 scratch row, 2 barriers, 512 x 8, prefetch, plain loads");
    }
    // rows of 512 KiB (rfft f32 N = 131072) and 1 MiB: several teams per XCD fit the L2
    constexpr int K512 = 512 << 10, M1 = 1 << 20;
    run<256, 8, 0, true, 0, 4, K512>(b, 2, "512 KiB rows: copy, 64 wg per XCD");
    for (int delay : {0, 2, 4}) {
        run<256, 8, 1, true, 16, 4, K512>(b, 2, "512 KiB rows: 4 teams of 16 per XCD, 256 x 8", delay);
        run<256, 8, 1, true, 16, 2, K512>(b, 1, "512 KiB rows: 2 teams of 16 per XCD, 256 x 8", delay);
        run<256, 8, 1, true, 16, 6, K512>(b, 3, "512 KiB rows: 6 teams of 16 per XCD, 256 x 8", delay);
        run<256, 16, 1, true, 16, 4, K512>(b, 1, "512 KiB rows: 4 teams of 8 per XCD, 256 x 16", delay);
        run<512, 8, 1, true, 16, 4, K512>(b, 1, "512 KiB rows: 4 teams of 8 per XCD, 512 x 8", delay);
        run<256, 8, 1, true, 16, 2, M1>(b, 2, "1 MiB rows: 2 teams of 32 per XCD, 256 x 8", delay);
        run<256, 16, 1, true, 16, 2, M1>(b, 1, "1 MiB rows: 2 teams of 16 per XCD, 256 x 16", delay);
    }
    return 0;
}
