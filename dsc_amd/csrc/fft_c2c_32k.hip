// fft_c2c_32k.hip — 32768-point complex transform of c32 rows (dsc_fft / dsc_ifft, dsc_fft.h:156-176): the three passes of
// fft_r2c_64k.hip without a real pass, in its own translation unit (adding a kernel to that file perturbed the register
// allocation of the 65536-point kernels, which sit exactly at 128 VGPRs).
#define DSC_R2C64K_HELPERS_ONLY 1
#include "fft_r2c_64k.hip"

namespace {

// complex: z [batch][32768] c32 -> Z [batch][32768] c32          (dsc_fft / dsc_ifft of complex rows, dsc_fft.h:156-176)
//
// The three passes without a real pass: the same persistent, software-pipelined row loop (rows are 256 KiB, line aligned
// on both sides; the results leave through the LDS staging area as 16-B stores, the next row's loads are issued ahead of
// them).  in_pitch / in_len in complex samples: shorter rows are zero padded by the descriptor range.
// CAST: the rows hold REAL samples, widened on the way in (dsc_fft / dsc_ifft of a real tensor, dsc.cpp:1984-1988); in_pitch / in_len
// then count reals.
template<bool INV, bool CAST>
__global__ __launch_bounds__(1024) void c2c32k_kernel(const f2 *__restrict__ z, f2 *__restrict__ Z, int batch, const f2 *__restrict__ aux,
                                                      int in_pitch, int in_len) {
    constexpr int EB = CAST ? 4 : 8;                                    // bytes per input sample
    auto sample = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        if constexpr (CAST) return cf{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, kStream)), 0.0f};
        else return load_c(r, voff, soff);
    };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *plane = lds;
    f2 *w1024 = (f2 *) (lds + kPlaneFloats);
    w1024[threadIdx.x] = aux[kAuxW1024 + threadIdx.x];
    __syncthreads();
    const int wave_sgpr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr float kScale = 1.0f / (float) kM;

    cf v[32];
    {
        const int row0 = blockIdx.x;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
            (void *) ((const char *) z + (size_t) row0 * in_pitch * EB), 0, row0 < batch ? in_len * EB : 0, 0x00020000);
        const int load_off = thread_id(wave_sgpr) * EB;
#pragma unroll
        for (int a = 0; a < 32; ++a) v[a] = sample(r0, load_off, a * 1024 * EB);
    }
    for (int row = blockIdx.x; row < batch; row += gridDim.x) {
        const int next_row = row + gridDim.x;
        const __amdgpu_buffer_rsrc_t rnext = __builtin_amdgcn_make_buffer_rsrc(
            (void *) ((const char *) z + (size_t) next_row * in_pitch * EB), 0, next_row < batch ? in_len * EB : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *) (Z + (size_t) row * kM), 0, kM * 8, 0x00020000);
        three_passes<INV>(v, plane, w1024, aux, wave_sgpr, false, false);      // v[p] = Z[t + 1024 br5(p)]
        if (INV) {
#pragma unroll
            for (int p = 0; p < 32; ++p) v[p] = cf{v[p].x * kScale, v[p].y * kScale};
        }
        const int load_off = thread_id(wave_sgpr) * EB;
        staged_time_store(v, plane, rout, wave_sgpr, [&](bool first_half) {
            if (first_half) {
#pragma unroll
                for (int a = 0; a < 16; ++a) v[2 * a] = sample(rnext, load_off, a * 1024 * EB);
            } else {
#pragma unroll
                for (int a = 16; a < 32; ++a) v[2 * (a - 16) + 1] = sample(rnext, load_off, a * 1024 * EB);
            }
        });
        unzip_rows(v);
    }
}

}  // namespace

// z: [batch][in_pitch] c32 (cast: f32) of which in_len <= 32768 samples are transformed; Z: [batch][32768] c32
void dsc_launch_fft32k_c32(const void *z, void *Z, int batch, int in_pitch, int in_len, bool inverse, bool cast, const void *aux, int n_cu,
                           hipStream_t stream) {
    if (batch <= 0) return;
    static unsigned long long attr_devices = 0;
    if (dsc_first_use_on_device(attr_devices)) {
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) c2c32k_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) c2c32k_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) c2c32k_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        DSC_KERNEL_CHECK(hipFuncSetAttribute((const void *) c2c32k_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    }
    const int grid = batch < n_cu ? batch : n_cu;
#define C2C32K(INV, CAST) DSC_LAUNCH((c2c32k_kernel<INV, CAST>), dim3(grid), dim3(1024), kLdsBytes, stream, (const f2 *) z, (f2 *) Z, batch, (const f2 *) aux, in_pitch, in_len)
    if (inverse) { if (cast) C2C32K(true, true); else C2C32K(true, false); }
    else         { if (cast) C2C32K(false, true); else C2C32K(false, false); }
#undef C2C32K
}
