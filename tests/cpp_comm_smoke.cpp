// A C++ host drives the multi-GPU reassembly through the C ABI alone (include/dsc_mi355x.h section C): communicator bootstrap,
// dsc_rfft writing this rank's slot of the persistent destination in place, dsc_shard_exchange_rows per chunk, dsc_shard_allgather
// of the whole shard — here with ONE rank (the test box has one GPU; more ranks differ only in the id having to travel).
// argv[1] == 0: link check only.
#include "dsc_mi355x.h"

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2 || std::atoi(argv[1]) == 0) {
        std::printf("linked: %p %p\n", (void *) &dsc_shard_allgather, (void *) &dsc_comm_init_rank);
        return 0;
    }
    dsc_ctx *ctx = dsc_ctx_init((size_t) 1 << 30, (size_t) 1 << 28);
    const int rows = 96, n = 65536, bins = n / 2 + 1, chunk = 40;
    std::vector<float> x((size_t) rows * n);
    unsigned s = 12345u;
    for (auto &v : x) { s = s * 1664525u + 1013904223u; v = (float) (s >> 8) / 8388608.0f - 1.0f; }
    const int shp_x[2] = {rows, n};
    dsc_tensor *tx = dsc_new_tensor(ctx, 2, shp_x, DSC_F32, nullptr);
    dsc_copy_from_host(ctx, tx, x.data(), x.size() * 4);

    // reference result: an ordinary tensor
    dsc_tensor *want_t = dsc_rfft(ctx, tx, nullptr, -1, -1);
    std::vector<std::complex<float>> want((size_t) rows * bins), got((size_t) rows * bins);
    dsc_copy_to_host(ctx, want_t, want.data(), want.size() * 8);

    dsc_comm_id id;
    if (dsc_comm_unique_id(&id) != 0) { std::printf("dsc_comm_unique_id failed\n"); return 2; }
    dsc_comm *comm = dsc_comm_init_rank(ctx, &id, 1, 0);
    if (comm == nullptr) { std::printf("dsc_comm_init_rank failed\n"); return 2; }
    if (dsc_comm_n_ranks(comm) != 1 || dsc_comm_rank(comm) != 0) { std::printf("rank bookkeeping wrong\n"); return 2; }

    const size_t row_bytes = (size_t) bins * 8;
    char *dest = (char *) dsc_device_alloc(ctx, (size_t) dsc_comm_n_ranks(comm) * rows * row_bytes);     // dest[P][rows][row_bytes]
    if (dest == nullptr) return 2;
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        const int nr = rows - r0 < chunk ? rows - r0 : chunk;
        const int sx[2] = {nr, n}, so[2] = {nr, bins};
        dsc_tensor *xin = dsc_tensor_from_device_ptr(ctx, (char *) tx->data + (size_t) r0 * n * 4, (size_t) nr * n * 4, 2, sx, DSC_F32);
        dsc_tensor *slot = dsc_tensor_from_device_ptr(ctx, dest + (size_t) dsc_comm_rank(comm) * rows * row_bytes + (size_t) r0 * row_bytes, (size_t) nr * row_bytes, 2, so, DSC_C32);
        dsc_rfft(ctx, xin, slot, -1, -1);                                     // the shard is written where it will be gathered from
        if (dsc_shard_exchange_rows(ctx, comm, dest, rows, row_bytes, r0, nr) != 0) { std::printf("dsc_shard_exchange_rows failed\n"); return 2; }
        dsc_tensor_free(ctx, xin);
        dsc_tensor_free(ctx, slot);
    }
    if (dsc_shard_allgather(ctx, comm, dest, rows, row_bytes) != 0) { std::printf("dsc_shard_allgather failed\n"); return 2; }
    dsc_synchronize(ctx);
    const int sg[2] = {rows, bins};
    dsc_tensor *all = dsc_tensor_from_device_ptr(ctx, dest, (size_t) rows * row_bytes, 2, sg, DSC_C32);
    dsc_copy_to_host(ctx, all, got.data(), got.size() * 8);
    const bool same = std::memcmp(got.data(), want.data(), got.size() * 8) == 0;
    std::printf("gathered == transform output: %s\n", same ? "ok" : "MISMATCH");
    dsc_tensor_free(ctx, all);
    dsc_comm_free(comm);
    dsc_device_free(ctx, dest);
    dsc_ctx_free(ctx);
    return same ? 0 : 1;
}
