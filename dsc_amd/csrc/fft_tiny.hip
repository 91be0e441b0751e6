// fft_tiny.hip — transforms of contiguous lines with complex length L = 2 .. 16 (real lengths 4 .. 32): ONE THREAD PER LINE.
//
// A line this short is a handful of registers; what costs is moving it.  A 256-thread workgroup owns 256 consecutive lines: their
// contiguous block is copied flat to LDS (coalesced, 8 / 16 B per lane), every thread picks its own line up from there, transforms it in
// registers (radix-2 DIF with compile-time twiddles, fft_regs_common.h), runs the packed-real pass on its own registers — both
// partners of a pair (k, L - k) live in the same thread — and writes the result back through LDS.  Two barriers, no exchange.
// PAD: lines with a byte pitch / valid byte length (zero padding / cropping through n=), gathered element by element.
// Before: fft_lines_kernel (fft_generic.hip) at 20-45 % of the roofline for these lengths.
//
// Reference: exec_fft / exec_rfft (dsc/src/dsc.cpp:1958-2007, 2102-2171) over dsc_complex_fft / dsc_real_fft (dsc_fft.h:57-238).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include "fft_regs_common.h"

namespace {

#ifndef DSC_TINY_NT
#define DSC_TINY_NT 256
#endif
constexpr int kTinyNT = DSC_TINY_NT;

// W_2L^k = exp(-2 pi i k / 2L), k <= L, for 2L <= 32: a 64th root
template<int L> __device__ constexpr double w2l_re(int k) { return root64_re(k * (32 / L)); }
template<int L> __device__ constexpr double w2l_im(int k) { return root64_im(k * (32 / L)); }

template<typename R, int L> constexpr size_t tiny_lds_bytes() { return (size_t) kTinyNT * (L + 2) * 2 * sizeof(R); }

// MODE: DSC_MODE_C2C, DSC_MODE_R2C_CAST (L reals in, either direction), DSC_MODE_R2C_PACKED (forward), DSC_MODE_C2R_PACKED (inverse).
// in_pitch_b / in_len_b (PAD only): byte pitch and valid bytes of an input line.
template<typename R, int L, int MODE, bool INV, bool PAD>
__global__ __launch_bounds__(kTinyNT) void fft_tiny_kernel(const void *__restrict__ in, void *__restrict__ out, long long n_lines, R scale,
                                                           int in_pitch_b, int in_len_b) {
    using C = cpx<R>;
    constexpr int NT = kTinyNT, G = NT, P = L + 2;                          // LDS pitch of a line (complex): odd-ish, holds L + 1 bins
    constexpr bool REAL_IN = MODE == DSC_MODE_R2C_CAST;
    constexpr int IN_PITCH = MODE == DSC_MODE_C2R_PACKED ? L + 1 : L, OUT_PITCH = MODE == DSC_MODE_R2C_PACKED ? L + 1 : L;
    constexpr int CB = (int) sizeof(C), EB = REAL_IN ? (int) sizeof(R) : CB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    C *stage = (C *) lds_raw;
    const int tid = threadIdx.x;
    const long long line0 = (long long) blockIdx.x * G;
    const long long left = n_lines - line0;
    const int n_valid = left < G ? (int) left : G;

    // ---- stage in: the block of lines, flat; element e = tid + NT m of the block is (line, j), e = line IN_PITCH + j
    constexpr int STEPS_IN = (G * IN_PITCH + NT - 1) / NT;                  // = IN_PITCH
    const long long gpitch_b = PAD ? (long long) in_pitch_b : (long long) IN_PITCH * EB;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *) ((const char *) in + line0 * gpitch_b), 0,
                                                                         (int) (PAD ? (n_valid - 1) * gpitch_b + in_len_b : n_valid * gpitch_b), 0x00020000);
    {
        C tmp[STEPS_IN];
        int line = tid / IN_PITCH, j = tid % IN_PITCH;
#pragma unroll
        for (int m = 0; m < STEPS_IN; ++m) {
            int voff = PAD ? (j * EB < in_len_b ? line * in_pitch_b + j * EB : 0x7f000000) : (tid + m * NT) * EB;
            if constexpr (REAL_IN) tmp[m] = buf_load_real<kCached>(rin, voff, 0, R{});
            else                   tmp[m] = buf_load<kCached>(rin, voff, 0, R{});
            if (PAD && !REAL_IN && j * EB + EB > in_len_b) tmp[m].y = (R) 0;   // a packed-real pair across the end of its line
            j += NT % IN_PITCH;
            line += NT / IN_PITCH;
            if (j >= IN_PITCH) { j -= IN_PITCH; ++line; }
        }
        line = tid / IN_PITCH; j = tid % IN_PITCH;
#pragma unroll
        for (int m = 0; m < STEPS_IN; ++m) {
            stage[line * P + j] = tmp[m];
            j += NT % IN_PITCH;
            line += NT / IN_PITCH;
            if (j >= IN_PITCH) { j -= IN_PITCH; ++line; }
        }
    }
    __syncthreads();

    // ---- this thread's line
    C v[32];
    const C *mine = stage + tid * P;
    if constexpr (MODE == DSC_MODE_C2R_PACKED) {                            // pre-pass (dsc_fft.h:194-228), both partners in this thread
#pragma unroll
        for (int k = 0; k < L; ++k) {
            C a = mine[k], b = mine[L - k];
            if (k == 0) { a.y = (R) 0; b.y = (R) 0; }
            const R wx = (R) w2l_re<L>(k), wy = (R) w2l_im<L>(k);
            const R wqx = (R) 0.5 * wy, wqy = (R) 0.5 * wx;                  // (i/2) conj(W_2L^k)
            const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            v[k] = C{(R) 0.5 * sx + (dx * wqx - dy * wqy), (R) 0.5 * sy + (dx * wqy + dy * wqx)};
        }
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) v[k] = mine[k];
    }
    __syncthreads();                                                        // every line is in registers: the area takes the results
    dft_n<R, INV, L>(v);                                                    // v[p] = bin brev(p, log2 L)
    constexpr int LOGL = ilog2(L);
    C *mine_w = stage + tid * P;
    if constexpr (MODE == DSC_MODE_R2C_PACKED) {                            // post-pass (dsc_fft.h:199-225)
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            const C a = v[brev(k == L ? 0 : k, LOGL)], b = v[brev((k == 0 || k == L) ? 0 : L - k, LOGL)];
            const R wx = (R) w2l_re<L>(k), wy = (R) w2l_im<L>(k);
            const R wqx = (R) 0.5 * wy, wqy = (R) -0.5 * wx;                 // -(i/2) W_2L^k
            const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            C x = C{((R) 0.5 * sx + (dx * wqx - dy * wqy)) * scale, ((R) 0.5 * sy + (dx * wqy + dy * wqx)) * scale};
            if (k == 0 || k == L) x.y = (R) 0;
            mine_w[k] = x;
        }
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const C r = v[brev(k, LOGL)];
            mine_w[k] = C{r.x * scale, r.y * scale};
        }
    }
    __syncthreads();

    // ---- stage out, flat (stores past the end of the batch are dropped by the descriptor range)
    constexpr int STEPS_OUT = (G * OUT_PITCH + NT - 1) / NT;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) ((char *) out + line0 * OUT_PITCH * CB), 0,
                                                                          n_valid * OUT_PITCH * CB, 0x00020000);
    int line = tid / OUT_PITCH, k = tid % OUT_PITCH;
#pragma unroll
    for (int m = 0; m < STEPS_OUT; ++m) {
        buf_store<kCached>(stage[line * P + k], rout, (tid + m * NT) * CB, 0);
        k += NT % OUT_PITCH;
        line += NT / OUT_PITCH;
        if (k >= OUT_PITCH) { k -= OUT_PITCH; ++line; }
    }
}

// The same lengths along a NON-LAST axis of [slices][axis][inner]: neighbouring lines are contiguous, so a thread per line (lane = column)
// loads and stores coalesced rows directly — no staging at all.  in_len valid elements of the input axis, the rest reads as zero; in_axis /
// out_axis: the axis lengths of the two tensors, in their own element types.
template<typename R, int L, int MODE, bool INV>
__global__ __launch_bounds__(kTinyNT) void fft_tiny_cols_kernel(const void *__restrict__ in, void *__restrict__ out, int inner, int tiles_per_slice,
                                                                int in_axis, int in_len, int out_axis, R scale) {
    using C = cpx<R>;
    constexpr int CB = (int) sizeof(C), RB = (int) sizeof(R), LOGL = ilog2(L);
    constexpr int IB = (MODE == DSC_MODE_R2C_PACKED || MODE == DSC_MODE_R2C_CAST) ? RB : CB;
    constexpr int OB = MODE == DSC_MODE_C2R_PACKED ? RB : CB;
    constexpr int kOut = 0x7f000000;
    const int slice = blockIdx.x / tiles_per_slice;
    const int col = (blockIdx.x - slice * tiles_per_slice) * kTinyNT + threadIdx.x;
    const bool live = col < inner;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void *) ((const char *) in + (size_t) slice * in_axis * inner * IB), 0, in_len * inner * IB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        (void *) ((char *) out + (size_t) slice * out_axis * inner * OB), 0, out_axis * inner * OB, 0x00020000);
    const int vin = live ? col * IB : kOut, vout = live ? col * OB : kOut;
    const int irow = inner * IB, orow = inner * OB;

    C v[32];
    if constexpr (MODE == DSC_MODE_R2C_PACKED) {                            // z[m] = (x[2m], x[2m + 1]): two rows of the axis
#pragma unroll
        for (int m = 0; m < L; ++m) v[m] = C{buf_load_real<kStream>(rin, vin, 2 * m * irow, R{}).x, buf_load_real<kStream>(rin, vin, (2 * m + 1) * irow, R{}).x};
    } else if constexpr (MODE == DSC_MODE_R2C_CAST) {
#pragma unroll
        for (int m = 0; m < L; ++m) v[m] = buf_load_real<kStream>(rin, vin, m * irow, R{});
    } else if constexpr (MODE == DSC_MODE_C2R_PACKED) {                     // bins 0 .. L, pre-pass (dsc_fft.h:194-228)
        C y[L + 1];
#pragma unroll
        for (int k = 0; k <= L; ++k) y[k] = buf_load<kStream>(rin, vin, k * irow, R{});
#pragma unroll
        for (int k = 0; k < L; ++k) {
            C a = y[k], b = y[L - k];
            if (k == 0) { a.y = (R) 0; b.y = (R) 0; }
            const R wx = (R) w2l_re<L>(k), wy = (R) w2l_im<L>(k);
            const R wqx = (R) 0.5 * wy, wqy = (R) 0.5 * wx;
            const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            v[k] = C{(R) 0.5 * sx + (dx * wqx - dy * wqy), (R) 0.5 * sy + (dx * wqy + dy * wqx)};
        }
    } else {
#pragma unroll
        for (int m = 0; m < L; ++m) v[m] = buf_load<kStream>(rin, vin, m * irow, R{});
    }
    dft_n<R, INV, L>(v);
    if constexpr (MODE == DSC_MODE_R2C_PACKED) {
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            const C a = v[brev(k == L ? 0 : k, LOGL)], b = v[brev((k == 0 || k == L) ? 0 : L - k, LOGL)];
            const R wx = (R) w2l_re<L>(k), wy = (R) w2l_im<L>(k);
            const R wqx = (R) 0.5 * wy, wqy = (R) -0.5 * wx;
            const R sx = a.x + b.x, sy = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            C x = C{((R) 0.5 * sx + (dx * wqx - dy * wqy)) * scale, ((R) 0.5 * sy + (dx * wqy + dy * wqx)) * scale};
            if (k == 0 || k == L) x.y = (R) 0;
            buf_store<kStream>(x, rout, vout, k * orow);
        }
    } else if constexpr (MODE == DSC_MODE_C2R_PACKED) {                     // sample pair (2m, 2m + 1) = (re, im) of z[m]
#pragma unroll
        for (int m = 0; m < L; ++m) {
            const C r = v[brev(m, LOGL)];
            if constexpr (sizeof(R) == 4) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.x * scale), rout, vout, 2 * m * orow, kStream);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r.y * scale), rout, vout, (2 * m + 1) * orow, kStream);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, r.x * scale), rout, vout, 2 * m * orow, kStream);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, r.y * scale), rout, vout, (2 * m + 1) * orow, kStream);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const C r = v[brev(k, LOGL)];
            buf_store<kStream>(C{r.x * scale, r.y * scale}, rout, vout, k * orow);
        }
    }
}

template<typename R, int L, int MODE, bool INV>
void launch_tiny_cols_one(const void *in, void *out, long long slices, int inner, int in_axis, int in_len, int out_axis, double scale, hipStream_t stream) {
    const int tiles = (inner + kTinyNT - 1) / kTinyNT;
    DSC_LAUNCH((fft_tiny_cols_kernel<R, L, MODE, INV>), dim3((unsigned) (slices * tiles)), dim3(kTinyNT), 0, stream, in, out, inner, tiles, in_axis, in_len,
               out_axis, (R) scale);
}
template<typename R, int L>
void launch_tiny_cols(const void *in, void *out, long long slices, int inner, int in_axis, int in_len, int out_axis, dsc_fft_mode mode, bool inverse,
                      double scale, hipStream_t stream) {
    if (mode == DSC_MODE_R2C_PACKED)      launch_tiny_cols_one<R, L, DSC_MODE_R2C_PACKED, false>(in, out, slices, inner, in_axis, in_len, out_axis, scale, stream);
    else if (mode == DSC_MODE_C2R_PACKED) launch_tiny_cols_one<R, L, DSC_MODE_C2R_PACKED, true>(in, out, slices, inner, in_axis, in_len, out_axis, scale, stream);
    else if (mode == DSC_MODE_R2C_CAST && !inverse) launch_tiny_cols_one<R, L, DSC_MODE_R2C_CAST, false>(in, out, slices, inner, in_axis, in_len, out_axis, scale, stream);
    else if (mode == DSC_MODE_R2C_CAST)   launch_tiny_cols_one<R, L, DSC_MODE_R2C_CAST, true>(in, out, slices, inner, in_axis, in_len, out_axis, scale, stream);
    else if (inverse)                     launch_tiny_cols_one<R, L, DSC_MODE_C2C, true>(in, out, slices, inner, in_axis, in_len, out_axis, scale, stream);
    else                                  launch_tiny_cols_one<R, L, DSC_MODE_C2C, false>(in, out, slices, inner, in_axis, in_len, out_axis, scale, stream);
}
template<typename R>
void launch_tiny_cols_len(int L, const void *in, void *out, long long slices, int inner, int in_axis, int in_len, int out_axis, dsc_fft_mode mode,
                          bool inverse, double scale, hipStream_t stream) {
    switch (L) {
        case 2:  launch_tiny_cols<R, 2>(in, out, slices, inner, in_axis, in_len, out_axis, mode, inverse, scale, stream); break;
        case 4:  launch_tiny_cols<R, 4>(in, out, slices, inner, in_axis, in_len, out_axis, mode, inverse, scale, stream); break;
        case 8:  launch_tiny_cols<R, 8>(in, out, slices, inner, in_axis, in_len, out_axis, mode, inverse, scale, stream); break;
        default: launch_tiny_cols<R, 16>(in, out, slices, inner, in_axis, in_len, out_axis, mode, inverse, scale, stream); break;
    }
}

template<typename R, int L, int MODE, bool INV, bool PAD>
void launch_tiny_pad(const void *in, void *out, long long n_lines, double scale, int in_pitch_b, int in_len_b, hipStream_t stream) {
    constexpr size_t lds = tiny_lds_bytes<R, L>();
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) fft_tiny_kernel<R, L, MODE, INV, PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
    const long long groups = (n_lines + kTinyNT - 1) / kTinyNT;
    DSC_LAUNCH((fft_tiny_kernel<R, L, MODE, INV, PAD>), dim3((unsigned) groups), dim3(kTinyNT), lds, stream, in, out, n_lines, (R) scale, in_pitch_b, in_len_b);
}

template<typename R, int L, int MODE, bool INV>
void launch_tiny_one(const void *in, void *out, long long n_lines, double scale, long long in_pitch_b, int in_len_b, hipStream_t stream) {
    if (in_pitch_b < 0) launch_tiny_pad<R, L, MODE, INV, false>(in, out, n_lines, scale, 0, 0, stream);
    else                launch_tiny_pad<R, L, MODE, INV, true>(in, out, n_lines, scale, (int) in_pitch_b, in_len_b, stream);
}

template<typename R, int L>
void launch_tiny(const void *in, void *out, long long n_lines, dsc_fft_mode mode, bool inverse, double scale, long long pb, int lb, hipStream_t stream) {
    if (mode == DSC_MODE_R2C_PACKED)      launch_tiny_one<R, L, DSC_MODE_R2C_PACKED, false>(in, out, n_lines, scale, pb, lb, stream);
    else if (mode == DSC_MODE_C2R_PACKED) launch_tiny_one<R, L, DSC_MODE_C2R_PACKED, true>(in, out, n_lines, scale, pb, lb, stream);
    else if (mode == DSC_MODE_R2C_CAST && !inverse) launch_tiny_one<R, L, DSC_MODE_R2C_CAST, false>(in, out, n_lines, scale, pb, lb, stream);
    else if (mode == DSC_MODE_R2C_CAST)   launch_tiny_one<R, L, DSC_MODE_R2C_CAST, true>(in, out, n_lines, scale, pb, lb, stream);
    else if (inverse)                     launch_tiny_one<R, L, DSC_MODE_C2C, true>(in, out, n_lines, scale, pb, lb, stream);
    else                                  launch_tiny_one<R, L, DSC_MODE_C2C, false>(in, out, n_lines, scale, pb, lb, stream);
}

template<typename R>
void launch_tiny_len(int L, const void *in, void *out, long long n_lines, dsc_fft_mode mode, bool inverse, double scale, long long pb, int lb,
                     hipStream_t stream) {
    switch (L) {
        case 2:  launch_tiny<R, 2>(in, out, n_lines, mode, inverse, scale, pb, lb, stream); break;
        case 4:  launch_tiny<R, 4>(in, out, n_lines, mode, inverse, scale, pb, lb, stream); break;
        case 8:  launch_tiny<R, 8>(in, out, n_lines, mode, inverse, scale, pb, lb, stream); break;
        default: launch_tiny<R, 16>(in, out, n_lines, mode, inverse, scale, pb, lb, stream); break;
    }
}

}  // namespace

bool dsc_fft_tiny_supports(int L) { return L == 2 || L == 4 || L == 8 || L == 16; }

// in_pitch / in_len: input line pitch and valid length in INPUT ELEMENTS (reals for R2C_PACKED / R2C_CAST, complex otherwise);
// in_pitch < 0 = full contiguous lines.  scale: applied to the results (1 forward; 1/L inverse complex, 2/2L inverse real).
void dsc_launch_fft_tiny(const void *in, void *out, long long n_lines, int L, dsc_fft_mode mode, bool inverse, bool single_precision, double scale,
                         long long in_pitch, int in_len, hipStream_t stream) {
    if (n_lines <= 0) return;
    const int eb = (single_precision ? 4 : 8) * ((mode == DSC_MODE_R2C_PACKED || mode == DSC_MODE_R2C_CAST) ? 1 : 2);
    const long long pb = in_pitch < 0 ? -1 : in_pitch * eb;
    const int lb = in_pitch < 0 ? 0 : in_len * eb;
    if (single_precision) launch_tiny_len<float>(L, in, out, n_lines, mode, inverse, scale, pb, lb, stream);
    else                  launch_tiny_len<double>(L, in, out, n_lines, mode, inverse, scale, pb, lb, stream);
}

// Tensor [slices][axis][inner] (contiguous), transform along `axis` (strided lines): in has in_axis elements along it of which in_len
// are used, out has out_axis; element counts in each side's own element type.  Every slice must stay below 2 GiB: the caller checks.
void dsc_launch_fft_tiny_cols(const void *in, void *out, long long slices, int inner, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                              double scale, int in_axis, int in_len, int out_axis, hipStream_t stream) {
    if (slices <= 0 || inner <= 0) return;
    if (single_precision) launch_tiny_cols_len<float>(L, in, out, slices, inner, in_axis, in_len, out_axis, mode, inverse, scale, stream);
    else                  launch_tiny_cols_len<double>(L, in, out, slices, inner, in_axis, in_len, out_axis, mode, inverse, scale, stream);
}
