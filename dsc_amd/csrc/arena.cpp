// arena.cpp — HBM arena bookkeeping (host side).
//
// Policy mirrors the reference's two allocators, restated for a device buffer whose bytes
// the host cannot touch:
//   main    best fit over an address-ordered free list, split on alloc, coalesce with both
//           neighbours on free           (dsc/src/dsc_allocator.cpp:51-221)
//   scratch bump pointer, reset as a whole (dsc/src/dsc_allocator.cpp:226-304)
#include "dsc_internal.h"

void dsc_main_arena::init(char *base, size_t size) {
    base_ = base;
    size_ = size;
    clear();
}

void dsc_main_arena::clear() {
    free_.clear();
    live_.clear();
    free_[0] = size_;
    used_ = 0;
}

// from_top: carve the block from the END of the highest free block that fits.  Used for the long-lived FFT plan
// tables: placed by best fit they end up wherever a tensor happened to be freed and cut the space for large tensors in
// two (a [2048, 262144] f64 batch needs three contiguous 4 GiB blocks); kept at the top they stay out of the way.
// Would blocks of these sizes (0 = unused) fit next to each other right now?  For callers that have another way when the
// arena is tight (the reference's allocator can only exit, dsc_allocator.cpp:112-114).  Exact for up to two blocks: each
// goes into its own free block, or both into one that holds their sum.
bool dsc_main_arena::fits(size_t nb_a, size_t nb_b) const {
    const size_t a = nb_a ? DSC_ALIGN_UP(nb_a, DSC_DEVICE_ALIGN) : 0, b = nb_b ? DSC_ALIGN_UP(nb_b, DSC_DEVICE_ALIGN) : 0;
    const size_t hi = a > b ? a : b, lo = a > b ? b : a;
    int n_hi = 0, n_lo = 0;
    for (const auto &blk : free_) {
        if (blk.second >= a + b) return true;
        if (blk.second >= hi) ++n_hi;
        else if (blk.second >= lo) ++n_lo;
    }
    return n_hi >= 2 || (n_hi >= 1 && n_lo >= 1) || (lo == 0 && n_hi >= 1);
}

char *dsc_main_arena::alloc(size_t nb, bool from_top) {
    DSC_ASSERT(nb > 0);
    const size_t need = DSC_ALIGN_UP(nb, DSC_DEVICE_ALIGN);

    auto best = free_.end();
    size_t largest = 0;
    for (auto it = free_.begin(); it != free_.end(); ++it) {
        if (it->second > largest) largest = it->second;
        if (it->second < need) continue;
        if (from_top) best = it;                                            // address ordered: the last fitting block
        else if (best == free_.end() || it->second < best->second) best = it;
    }
    if (best == free_.end()) {
        DSC_LOG_FATAL("error allocating %.2fKB in the main HBM arena (%.1f of %.1f MB in use, largest free block %.1f MB)",
                      (double) need / 1024., (double) used_ / 1048576., (double) size_ / 1048576., (double) largest / 1048576.);
    }
    const size_t off = best->first;
    const size_t left = best->second - need;
    free_.erase(best);
    size_t at = off;
    if (from_top) {
        at = off + left;
        if (left > 0) free_[off] = left;
    } else if (left > 0) {
        free_[off + need] = left;
    }
    live_[at] = need;
    used_ += need;
    return base_ + at;
}

void dsc_main_arena::free(char *p) {
    if (p == nullptr) return;
    const size_t off = (size_t) (p - base_);
    auto it = live_.find(off);
    if (it == live_.end()) return;             // double free / stale pointer: ignore
    size_t size = it->second;
    live_.erase(it);
    used_ -= size;

    size_t start = off;
    auto next = free_.lower_bound(off);
    if (next != free_.end() && next->first == off + size) {       // merge with the block after
        size += next->second;
        next = free_.erase(next);
    }
    if (next != free_.begin()) {                                   // merge with the block before
        auto prev = std::prev(next);
        if (prev->first + prev->second == off) {
            start = prev->first;
            size += prev->second;
            free_.erase(prev);
        }
    }
    free_[start] = size;
}

char *dsc_scratch_arena::alloc(size_t nb) {
    const size_t need = DSC_ALIGN_UP(nb, DSC_DEVICE_ALIGN);
    if (top_ + need > size_) {
        DSC_LOG_FATAL("error allocating %.2fKB in the scratch HBM arena (%.1f of %.1f MB in use)",
                      (double) need / 1024., (double) top_ / 1048576., (double) size_ / 1048576.);
    }
    char *p = base_ + top_;
    top_ += need;
    return p;
}
