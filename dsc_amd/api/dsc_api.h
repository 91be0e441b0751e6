// dsc_api.h — header-only C++ wrapper over the C ABI in include/dsc_mi355x.h.
//
// Mirror of the reference's dsc/api/dsc_api.h for the hot-path subset: `dsc::init`, RAII
// `dsc::tensor<T>`, `operator*`, `dsc::sum`, `dsc::fft / ifft / rfft / irfft` (reference
// lines 15-21, 24-34, 36-143, 148-186, 285-302, 321-343) plus `dsc::filter_fft`.  The one semantic
// difference: tensor payloads live in HBM, so construction from host data and `to_host()`
// copy through dsc_copy_from_host / dsc_copy_to_host instead of dereferencing `data()`
// (reference: memcpy into x_->data, dsc_api.h:63-66).
#pragma once

#include "dsc_mi355x.h"

#include <cstddef>
#include <initializer_list>
#include <type_traits>
#include <vector>

// dsc_api.h:15-21
#define DSC_SLICE_ALL()                 (dsc_slice{DSC_VALUE_NONE, DSC_VALUE_NONE, 1})
#define DSC_SLICE_IDX(idx_)             (dsc_slice{(idx_), (idx_), (idx_)})      // a single element, not a slice
#define DSC_SLICE_ALL_STEP(step_)       (dsc_slice{DSC_VALUE_NONE, DSC_VALUE_NONE, (step_)})
#define DSC_SLICE_FROM(start_)          (dsc_slice{(start_), DSC_VALUE_NONE, 1})
#define DSC_SLICE_TO(stop_)             (dsc_slice{DSC_VALUE_NONE, (stop_), 1})
#define DSC_SLICE_RANGE(start_, stop_)  (dsc_slice{(start_), (stop_), 1})

namespace dsc {

static dsc_ctx *ctx = nullptr;             // dsc_api.h:26

// dsc_api.h:28-34: with scratch_mem == 0, 90 % of main_mem goes to the main arena and a
// tenth of that to scratch.
static inline void init(size_t main_mem, size_t scratch_mem = 0, int device = 0) {
    if (scratch_mem == 0) {
        main_mem = (size_t) ((double) main_mem * 0.9);
        scratch_mem = (size_t) ((double) main_mem * 0.1);
    }
    dsc_set_device(device);
    ctx = dsc_ctx_init(main_mem, scratch_mem);
}

template<typename T> struct dtype_of;
template<> struct dtype_of<float>   { static constexpr dsc_dtype value = DSC_F32; };
template<> struct dtype_of<double>  { static constexpr dsc_dtype value = DSC_F64; };
template<> struct dtype_of<dsc_c32> { static constexpr dsc_dtype value = DSC_C32; };
template<> struct dtype_of<dsc_c64> { static constexpr dsc_dtype value = DSC_C64; };

template<typename T>
class tensor {
public:
    tensor() noexcept : x_(nullptr) {}
    tensor(dsc_tensor *x) noexcept : x_(x) {}
    // host data -> new 1-D device tensor (reference: tensor(const T*, int), dsc_api.h:63-66)
    tensor(const T *host, int ne) noexcept {
        x_ = dsc_new_tensor(ctx, 1, &ne, dtype_of<T>::value, nullptr);
        dsc_copy_from_host(ctx, x_, host, (size_t) ne * sizeof(T));
    }
    // host data -> new n-D device tensor
    tensor(const T *host, std::initializer_list<int> shape) noexcept {
        int s[DSC_MAX_DIMS];
        int n = 0;
        size_t ne = 1;
        for (int d : shape) { s[n++] = d; ne *= (size_t) d; }
        x_ = dsc_new_tensor(ctx, n, s, dtype_of<T>::value, nullptr);
        dsc_copy_from_host(ctx, x_, host, ne * sizeof(T));
    }
    // {a, b, c} -> 1-D tensor of scalars; ({shape}, fill) -> constant tensor        (dsc_api.h:46-57)
    tensor(std::initializer_list<T> scalars) noexcept : tensor(std::vector<T>(scalars).data(), (int) scalars.size()) {}
    tensor(std::initializer_list<int> shape, const T fill) noexcept {
        int s[DSC_MAX_DIMS];
        int n = 0;
        size_t ne = 1;
        for (int d : shape) { s[n++] = d; ne *= (size_t) d; }
        x_ = dsc_new_tensor(ctx, n, s, dtype_of<T>::value, nullptr);
        const std::vector<T> host(ne, fill);
        dsc_copy_from_host(ctx, x_, host.data(), ne * sizeof(T));
    }
    // Copies are DEEP, as in the reference (memcpy of the payload, dsc_api.h:63-81); here the payload is copied on the
    // device: selecting everything with one full slice is dsc_tensor_get_slice's copy kernel.
    tensor(const tensor &o) noexcept : x_(o.x_ == nullptr ? nullptr : clone(o.x_)) {}
    tensor &operator=(const tensor &o) noexcept {
        if (this != &o) {
            if (x_ != nullptr) dsc_tensor_free(ctx, x_);
            x_ = o.x_ == nullptr ? nullptr : clone(o.x_);
        }
        return *this;
    }
    tensor(tensor &&o) noexcept : x_(o.x_) { o.x_ = nullptr; }
    tensor &operator=(tensor &&o) noexcept {
        if (this != &o) {
            if (x_ != nullptr) dsc_tensor_free(ctx, x_);
            x_ = o.x_;
            o.x_ = nullptr;
        }
        return *this;
    }
    ~tensor() noexcept { if (x_ != nullptr) dsc_tensor_free(ctx, x_); }

    int dim(int idx) const noexcept { return x_->shape[idx < 0 ? DSC_MAX_DIMS + idx : DSC_MAX_DIMS - x_->n_dim + idx]; }
    int size() const noexcept { return dim(0); }
    int ndim() const noexcept { return x_->n_dim; }
    int ne() const noexcept { return x_->ne; }
    dsc_dtype dtype() const noexcept { return x_->dtype; }
    dsc_tensor *raw() const noexcept { return x_; }

    // device -> host copy of the whole payload, as elements of U (e.g. dsc_c32 for an rfft result:
    // like the reference, rfft<float> returns tensor<float> whose payload is complex, dsc_api.h:333-337)
    template<typename U = T>
    std::vector<U> to_host() const {
        const size_t bytes = (size_t) x_->ne * (x_->dtype == DSC_F32 ? 4 : x_->dtype == DSC_C64 ? 16 : 8);
        std::vector<U> out(bytes / sizeof(U));
        dsc_copy_to_host(ctx, x_, out.data(), bytes);
        return out;
    }

    // dsc_api.h:148-186: element-wise operators with broadcasting; a scalar operand of the tensor's own element type is
    // wrapped into a one-element tensor (dsc_wrap_*), on either side.
    tensor operator+(const tensor &o) const noexcept { return dsc_add(ctx, x_, o.x_, nullptr); }
    tensor operator+(const T v) const noexcept { return dsc_add(ctx, x_, wrap(v).x_, nullptr); }
    friend tensor operator+(const T v, const tensor &o) noexcept { return wrap(v) + o; }
    tensor operator-(const tensor &o) const noexcept { return dsc_sub(ctx, x_, o.x_, nullptr); }
    tensor operator-(const T v) const noexcept { return dsc_sub(ctx, x_, wrap(v).x_, nullptr); }
    friend tensor operator-(const T v, const tensor &o) noexcept { return wrap(v) - o; }
    tensor operator*(const tensor &o) const noexcept { return dsc_mul(ctx, x_, o.x_, nullptr); }
    tensor operator*(const T v) const noexcept { return dsc_mul(ctx, x_, wrap(v).x_, nullptr); }
    friend tensor operator*(const T v, const tensor &o) noexcept { return wrap(v) * o; }
    tensor operator/(const tensor &o) const noexcept { return dsc_div(ctx, x_, o.x_, nullptr); }
    tensor operator/(const T v) const noexcept { return dsc_div(ctx, x_, wrap(v).x_, nullptr); }
    friend tensor operator/(const T v, const tensor &o) noexcept { return wrap(v) / o; }
    tensor &operator/=(const tensor &o) noexcept { dsc_div(ctx, x_, o.x_, x_); return *this; }     // in place: out = x
    tensor &operator*=(const tensor &o) noexcept { dsc_mul(ctx, x_, o.x_, x_); return *this; }

    // dsc_api.h:117-143: x.get(2, 3) (indexes) / x.get(DSC_SLICE_ALL(), DSC_SLICE_TO(n)) (slices) copy on the device
    template<typename... Args>
    tensor get(Args... sel) const noexcept {
        if constexpr ((std::is_same_v<Args, dsc_slice> && ...)) return dsc_tensor_get_slice(ctx, x_, (int) sizeof...(Args), sel...);
        else                                                      return dsc_tensor_get_idx(ctx, x_, (int) sizeof...(Args), ((int) sel)...);
    }
    template<typename... Args>
    tensor &set(const tensor &other, Args... sel) noexcept {
        if constexpr ((std::is_same_v<Args, dsc_slice> && ...)) dsc_tensor_set_slice(ctx, x_, other.x_, (int) sizeof...(Args), sel...);
        else                                                      dsc_tensor_set_idx(ctx, x_, other.x_, (int) sizeof...(Args), ((int) sel)...);
        return *this;
    }

    dsc_tensor *x_;

private:
    static dsc_tensor *clone(dsc_tensor *src) noexcept { return dsc_tensor_get_slice(ctx, src, 1, DSC_SLICE_ALL()); }
    static tensor wrap(const T v) noexcept {
        if constexpr (std::is_same_v<T, float>)        return dsc_wrap_f32(ctx, v);
        else if constexpr (std::is_same_v<T, double>)  return dsc_wrap_f64(ctx, v);
        else if constexpr (std::is_same_v<T, dsc_c32>) return dsc_wrap_c32(ctx, v);
        else                                           return dsc_wrap_c64(ctx, v);
    }
};

template<typename T>
static inline tensor<T> sum(const tensor<T> &x, int axis = -1, bool keep_dims = true) noexcept {            // dsc_api.h:285-290
    return dsc_sum(ctx, x.x_, nullptr, axis, keep_dims);
}
template<typename T>
static inline tensor<T> mean(const tensor<T> &x, int axis = -1, bool keep_dims = true) noexcept {
    return dsc_mean(ctx, x.x_, nullptr, axis, keep_dims);
}
template<typename T>
static inline tensor<T> max(const tensor<T> &x, int axis = -1, bool keep_dims = true) noexcept {
    return dsc_max(ctx, x.x_, nullptr, axis, keep_dims);
}
template<typename T>
static inline tensor<T> min(const tensor<T> &x, int axis = -1, bool keep_dims = true) noexcept {
    return dsc_min(ctx, x.x_, nullptr, axis, keep_dims);
}

// dsc_api.h:294-302: transpose(x) reverses the axes, transpose(x, 1, 0, ...) permutes them
template<typename T, typename... Args>
static inline tensor<T> transpose(const tensor<T> &x, Args... axes) noexcept {
    static_assert((std::is_same_v<Args, int> && ...), "axes are ints");
    if constexpr (sizeof...(Args) == 0) return dsc_transpose(ctx, x.x_, 0);
    else                                return dsc_transpose(ctx, x.x_, (int) sizeof...(Args), axes...);
}

template<typename T>
static inline tensor<T> fft(const tensor<T> &x, int n = -1, int axis = -1) noexcept { return dsc_fft(ctx, x.x_, nullptr, n, axis); }
template<typename T>
static inline tensor<T> ifft(const tensor<T> &x, int n = -1, int axis = -1) noexcept { return dsc_ifft(ctx, x.x_, nullptr, n, axis); }
template<typename T>
static inline tensor<T> rfft(const tensor<T> &x, int n = -1, int axis = -1) noexcept { return dsc_rfft(ctx, x.x_, nullptr, n, axis); }
template<typename T>
static inline tensor<T> irfft(const tensor<T> &x, int n = -1, int axis = -1) noexcept { return dsc_irfft(ctx, x.x_, nullptr, n, axis); }

// README.md:141-163 (C++ filterFFT) as one call: y = irfft(rfft(s, n) * H)
template<typename T>
static inline tensor<T> filter_fft(const tensor<T> &s, const tensor<T> &H) noexcept { return dsc_filter_fft(ctx, s.x_, H.x_, nullptr); }

static inline void synchronize() noexcept { dsc_synchronize(ctx); }

}  // namespace dsc
