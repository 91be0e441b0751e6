// kernels.h — host-callable launchers of the HIP kernels (gfx950).  Everything is enqueued
// on the given stream; nothing here synchronises or allocates.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <cstdlib>

// Error convention of the reference (dsc/include/dsc.h:14-28): message on stderr, exit.  Every kernel launch and every
// attribute call goes through these: a refused launch (grid too large, LDS not granted, ...) must never return an
// uninitialised tensor silently.
#define DSC_KERNEL_CHECK(call)                                                                                   \
    do {                                                                                                         \
        hipError_t kerr_ = (call);                                                                               \
        if (kerr_ != hipSuccess) {                                                                               \
            fprintf(stderr, "HIP error %s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(kerr_)); \
            exit(EXIT_FAILURE);                                                                                  \
        }                                                                                                        \
    } while (0)
#define DSC_LAUNCH(...)                        \
    do {                                       \
        hipLaunchKernelGGL(__VA_ARGS__);       \
        DSC_KERNEL_CHECK(hipGetLastError());   \
    } while (0)

// Dynamic-LDS opt-ins are per device: true the first time a call site runs on the calling thread's current device
// (one process per GPU is the design, but a host that drives several devices from one process stays correct).
static inline bool dsc_first_use_on_device(unsigned long long &seen) {
    int dev = 0;
    DSC_KERNEL_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (seen & bit) return false;
    seen |= bit;
    return true;
}

// A batch of 1-D lines inside a tensor.  Line q (0 <= q < n_lines) starts at element
//   (q / inner) * outer_stride + (q % inner) * inner_stride
// and advances by elem_stride per sample.  Strides are in elements of the array's own
// element type (real or complex).  For a contiguous DSC tensor transformed along `axis`:
// inner = prod(shape[axis+1:]), inner_stride = 1, elem_stride = inner,
// outer_stride = shape[axis] * inner  — the lines dsc_axis_iterator walks
// (reference: dsc/include/dsc_iter.h:11-65).
struct dsc_line_layout {
    long long outer_stride;
    long long inner_stride;
    long long elem_stride;
};

enum dsc_fft_mode {
    DSC_MODE_C2C = 0,        // complex in  -> complex out          (dsc_fft / dsc_ifft)
    DSC_MODE_R2C_CAST = 1,   // real in, cast to complex -> complex (dsc_fft on f32/f64, dsc.cpp:1984-1988)
    DSC_MODE_R2C_PACKED = 2, // 2L reals -> L+1 bins                (dsc_rfft,  dsc_fft.h:178-225)
    DSC_MODE_C2R_PACKED = 3, // L+1 bins -> 2L reals                (dsc_irfft, dsc_fft.h:194-236)
};

struct dsc_fft_lines_args {
    const void *in;
    void *out;
    long long n_lines;
    long long inner;
    dsc_line_layout lin, lout;
    int L;                 // complex transform length (power of two, <= dsc_fft_lds_max_len)
    int in_len;            // valid input samples along the axis (reals for R2C_PACKED, bins for C2R); the rest reads as zero
    int inverse;           // conjugate twiddles
    double scale;          // multiplied into every output
    const void *tw;        // W_L^k, k in [0, L)       interleaved (cos, sin), precision of the transform
    const void *tw_real;   // W_{2L}^k, k in [0, L]    (packed modes only)
    long long tw4_len;     // four-step: also multiply output k of line q by W_{tw4_len}^{(q % inner) * k}; 0 = off
    const void *tw4;       // table W_{tw4_len}^m, m in [0, tw4_len) (gathered, L2 resident); NULL = sincospi in double
};

// Largest complex length the LDS line kernel handles for the given precision.
int dsc_fft_lds_max_len(bool single_precision);

// One pass over HBM: gather -> Stockham radix-4/2 in LDS -> (real post-pass) -> scatter.
void dsc_launch_fft_lines(const dsc_fft_lines_args &a, dsc_fft_mode mode, bool single_precision, hipStream_t stream);

// Helpers of the multi-pass (four-step) path: contiguous [n_lines][L] complex work buffers holding
// tensor lines q_first .. q_first + n_lines - 1.
//   pack     tensor lines -> work (zero-pad / crop / cast / pair-up reals), per `mode`
//   prepass  C2R: bins in work-like layout -> Z = h1 + conj(w) h2           (dsc_fft.h:194-228)
//   postpass R2C: Z (work) -> L+1 bins scattered to the tensor               (dsc_fft.h:199-225)
//   unpack   work -> tensor lines (complex, or 2L reals for C2R) times scale
void dsc_launch_fft_pack(const void *in, void *work, long long q_first, long long n_lines, long long inner, dsc_line_layout lin,
                         int L, int in_len, dsc_fft_mode mode, bool single_precision, hipStream_t stream);
void dsc_launch_fft_c2r_prepass(const void *in, void *work, long long q_first, long long n_lines, long long inner, dsc_line_layout lin,
                                int L, int in_len, const void *tw_real, bool single_precision, hipStream_t stream);
void dsc_launch_fft_r2c_postpass(const void *work, void *out, long long q_first, long long n_lines, long long inner, dsc_line_layout lout,
                                 int L, const void *tw_real, bool single_precision, hipStream_t stream);
void dsc_launch_fft_unpack(const void *work, void *out, long long q_first, long long n_lines, long long inner, dsc_line_layout lout,
                           int L, double scale, dsc_fft_mode mode, bool single_precision, hipStream_t stream);

// ---- register-resident 65536-point real transforms (f32), one HBM round trip ------------
// x: [batch][65536] f32 contiguous rows; X: [batch][32769] c32 contiguous rows.
// aux: tables built by dsc_r2c64k_build_tables (device pointer).
size_t dsc_r2c64k_table_bytes();
void   dsc_r2c64k_build_tables(void *host_dst);          // fills a host staging buffer of table_bytes
// in_pitch: floats between input rows; in_len <= 65536 valid samples per row, the rest reads as zero (zero padding / crop)
void   dsc_launch_rfft64k(const float *x, void *X, int batch, int in_pitch, int in_len, const void *aux, int n_cu, hipStream_t stream);
void   dsc_launch_irfft64k(const void *X, float *x, int batch, int in_pitch, int in_len, const void *aux, int n_cu, hipStream_t stream);   // pitch / valid length in bins
// complex 32768-point transform of c32 rows in the same design (z: [batch][in_pitch], in_len <= 32768 samples used)
void   dsc_launch_fft32k_c32(const void *z, void *Z, int batch, int in_pitch, int in_len, bool inverse, bool cast, const void *aux, int n_cu,
                             hipStream_t stream);
// y = irfft(rfft(s) * H) fused; H: [32769] c32
void   dsc_launch_filter64k(const float *s, const void *H, float *y, int batch, int in_pitch, int in_len, const void *aux, int n_cu,
                            hipStream_t stream);

// ---- transforms along a non-last axis, lanes = neighbouring lines (fft_regs_cols.hip): complex length 32 .. 2048 (c32 data: 4096)
bool   dsc_fft_regs_cols_supports(int L, dsc_fft_mode mode, bool single_precision);
void   dsc_launch_fft_regs_cols(const void *in, void *out, long long slices, int inner, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                                const void *tw_full, const void *tw_real, double scale, int in_axis, int in_len, int out_axis, hipStream_t stream);
// complex transform of n = n1 n2 points along the middle axis of [slices][n][inner] in two passes of the column kernel (four-step)
bool   dsc_fft_cols_4step_split(int n, bool single_precision, int cols, int *n1, int *n2);
// real transforms of n = n1 n2 points along the middle axis of [slices][n][2 cc_n] reals: two neighbouring columns as one complex column
void   dsc_launch_rfft_cols_4step(const void *in, void *work, void *out, long long slices, int cc_n, int n1, int n2, bool single_precision,
                                  const void *tw1, const void *tw2, const void *twn, hipStream_t stream);
void   dsc_launch_irfft_cols_4step(const void *in, void *work, void *out, long long slices, int cc_n, int n1, int n2, bool single_precision,
                                   const void *tw1, const void *tw2, const void *twn, double scale, hipStream_t stream);
void   dsc_launch_fft_cols_4step(const void *in, void *work, void *out, long long slices, int inner, int n1, int n2, dsc_fft_mode mode, bool inverse,
                                 bool single_precision, const void *tw1, const void *tw2, const void *twn, double scale, hipStream_t stream);

// ---- long real transforms in two passes over HBM (fft_r2c_2pass.hip): packed complex length L = 32768 (f64 only),
// 65536 ... 1048576; f32 and f64.  forward: reals -> out = [rows][L + 1] bins; inverse: bins -> [rows][2L] reals.
// work: rows * L complex of scratch; tw_full: W_L^k, k < L; tw_real: W_{2L}^k, k <= L (the REAL plan's own tables).
bool   dsc_fft_two_pass_supports(int L, bool single_precision);
// the same lengths for complex data (dsc_fft / dsc_ifft of complex tensors): in = complex rows, out = [rows][L]
void   dsc_launch_fft_two_pass(const void *in, void *out, long long rows, int L, bool inverse, bool cast, bool single_precision, void *work,
                               const void *tw_full, long long in_pitch, int in_len, hipStream_t stream);
// in_pitch / in_len: pitch and valid length of the input rows in input elements (reals forward, bins inverse): shorter rows are zero
// padded, longer ones cropped.
void   dsc_launch_rfft_two_pass(const void *in, void *out, long long rows, int L, bool inverse, bool single_precision, void *work,
                                const void *tw_full, const void *tw_real, long long in_pitch, int in_len, hipStream_t stream);

// ---- complex rows of 65536 and 131072 points (f32, f64; the f64 131072 = config 5) and 32768 points (f64) in one launch with the four-step intermediate in the XCD-local L2
// (fft_xcd_fused.hip).  Arguments as the two-pass launchers; `scratch`: dsc_fft_fused_l2_scratch_bytes() bytes, of which the
// first dsc_fft_fused_l2_ctl_bytes() are the control block; host_error: a pinned, device-visible word that receives a non-zero
// code if a barrier of the launch did not complete.  Returns false when the launch cannot be made fully resident on this device.
bool   dsc_fft_fused_l2_supports(int L, bool single_precision, bool real, bool inverse);
size_t dsc_fft_fused_l2_ctl_bytes();
size_t dsc_fft_fused_l2_scratch_bytes(int L, bool single_precision);
bool   dsc_launch_fft_fused_l2(const void *in, void *out, long long rows, int L, bool real, bool inverse, bool cast, bool single_precision, void *scratch,
                               unsigned *host_error, const void *tw_full, const void *tw_real, long long in_pitch, int in_len, hipStream_t stream);

// ---- complex lengths 2 .. 16 (real 4 .. 32), contiguous lines, one thread per line (fft_tiny.hip).  in_pitch / in_len in input
// elements (reals for R2C_PACKED / R2C_CAST), in_pitch < 0 = full lines; scale multiplies the results.
bool dsc_fft_tiny_supports(int L);
// ... and along a non-last axis of [slices][axis][inner] (lane = column, no staging)
void dsc_launch_fft_tiny_cols(const void *in, void *out, long long slices, int inner, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                              double scale, int in_axis, int in_len, int out_axis, hipStream_t stream);
void dsc_launch_fft_tiny(const void *in, void *out, long long n_lines, int L, dsc_fft_mode mode, bool inverse, bool single_precision, double scale,
                         long long in_pitch, int in_len, hipStream_t stream);

// ---- register-resident transforms of contiguous full lines, complex length 256 .. 16384 (f32, f64; f32 C2C also 32768)
// (fft_regs_mid.hip).  in / out: [n_lines][L] complex (C2C), [n_lines][2L] reals -> [n_lines][L+1] bins
// (R2C_PACKED) or the converse (C2R_PACKED).  tw_full: W_L^k, k < L; tw_real: W_{2L}^k, k <= L.
bool dsc_fft_regs_mid_supports(int L, dsc_fft_mode mode, bool single_precision);
bool dsc_fft_regs_small_supports(int L);        // 32 .. 256: same launcher, full contiguous lines only (in_pitch < 0)
// in_pitch / in_len: input line pitch and valid length in input elements (reals for R2C_PACKED / R2C_CAST, complex otherwise):
// shorter lines are zero padded, longer ones cropped; in_pitch < 0 = full contiguous lines (the fast instantiation).
void dsc_launch_fft_regs_mid(const void *in, void *out, long long n_lines, int L, dsc_fft_mode mode, bool inverse, bool single_precision,
                             const void *tw_full, const void *tw_real, double scale, long long in_pitch, int in_len, hipStream_t stream);

// fused y = irfft(rfft(s, 2L) * H) at the same lengths (README filterFFT): s = [n_lines][in_pitch] reals of which in_len <= 2L are
// used, H = [L + 1] bins shared by all rows, y = [n_lines][2L] reals
void dsc_launch_filter_regs_mid(const void *s, const void *H, void *y, long long n_lines, int L, bool single_precision, const void *tw_full,
                                const void *tw_real, long long in_pitch, int in_len, hipStream_t stream);

// ---- element-wise ------------------------------------------------------------------------
// dtype codes are dsc_dtype values (0 f32, 1 f64, 2 c32, 3 c64)
void dsc_launch_cast(const void *in, int in_dtype, void *out, int out_dtype, long long ne, hipStream_t stream);

// op: 0 abs, 1 angle, 2 conj, 3 real, 4 imag (output real dtype of the input; conj keeps the dtype)
void dsc_launch_unary(const void *in, int in_dtype, void *out, int op, long long ne, hipStream_t stream);

struct dsc_bcast_args {
    int out_shape[4];
    int a_stride[4];       // element strides, 0 on broadcast dims (dsc_iter.h:67-95)
    int b_stride[4];
    long long ne;
    int a_scalar, b_scalar;   // reference's scalar fast paths (dsc.cpp:1194-1212)
    // index fast paths (no per-element divisions): 0 general, 1 both operands have the output's shape,
    // 2 a full / b = the trailing dims of the output (row broadcast: ib = i % b_ne), 3 the converse,
    // 4 a full / b = the leading dims of the output, ones behind (column broadcast: ib = i / small_ne), 5 the converse
    int fast;
    int small_ne;             // fast = 2, 3: element count of the broadcast operand; fast = 4, 5: output elements per element of it
};
// op: 0 add, 1 sub, 2 mul, 3 div   (only mul is exported through the C ABI this round)
void dsc_launch_binary(const void *a, const void *b, void *out, int dtype, int op, const dsc_bcast_args &g, hipStream_t stream);
// operands of different dtypes, equal shapes, contiguous: casts to the promoted dtype in registers.  false = not taken
// (odd element count or unaligned views): the caller casts into scratch tensors and calls dsc_launch_binary.
bool dsc_launch_binary_mixed(const void *a, int a_dtype, const void *b, int b_dtype, void *out, int op, long long ne, hipStream_t stream);

// ---- reductions along one axis -----------------------------------------------------------
// x viewed as [outer][axis_n][inner] contiguous; out as [outer][inner].
// op: 0 sum, 1 mean, 2 max, 3 min
// workspace: scratch for the segmented path (may be NULL)
void dsc_launch_reduce(const void *x, void *out, int dtype, int op, long long outer, int axis_n, long long inner,
                       void *workspace, size_t workspace_bytes, hipStream_t stream);

// ---- strided gather / scatter of a slice region (indexing.cpp) -----------------------------
// Region of a tensor: element (i0, i1, i2, i3), i_d < count[d], sits at base + sum_d i_d * stride[d]
// (elements; strides may be negative).  Row-major order over the region = flat index of the dense side.
struct dsc_region {
    int count[4];
    long long stride[4];
    long long base;
    long long ne;             // product of count
};
// gather: dense[i] = strided[region(i)];  scatter: strided[region(i)] = dense[i % dense_ne]
void dsc_launch_region_copy(const void *src, void *dst, int elem_bytes, const dsc_region &r, bool scatter, long long dense_ne,
                            hipStream_t stream);

// out[b][c][r] = in[b][r][c] for `batch` matrices of rows x cols elements (32 x 32 tiles through LDS: both sides coalesced)
void dsc_launch_transpose_last2(const void *in, void *out, int elem_bytes, long long batch, int rows, int cols, hipStream_t stream);
// any permutation that moves the last axis (n_dim <= 4, dense input): tiled through LDS, coalesced on both sides
bool dsc_launch_transpose_moving_last(const void *in, void *out, int elem_bytes, int n_dim, const int *shape, const int *in_stride, const int *perm,
                                      hipStream_t stream);
