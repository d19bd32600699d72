// Diagnostic only (not part of the C ABI): what rate can v_mfma_f32_32x32x2_f32 sustain on this chip
//   mode 0: registers only;  mode 1: + the conv kernel's ds_read_b128 operand fetches;  mode 2: + 2 barriers per k-step
#include "common.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void mfma_rate_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[128 * 36];
    __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 36; i += 256) { As[i] = (float)(i % 7) * 0.01f; Bs[i] = (float)(i % 5) * 0.02f; }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, fr = lane & 31, fh = lane >> 5;
    float4 a[2], b[2];
    a[0] = a[1] = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
    b[0] = b[1] = make_float4(1.0f, 0.5f, 0.25f, 0.125f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(&As[(wm + i * 32 + fr) * 36 + ks * 8 + fh * 4]);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[(wn + j * 32 + fr) * 36 + ks * 8 + fh * 4]);
            }
#define STEP(C)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].C, b[j].C, acc[i][j], 0, 0, 0);
            STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
        }
        if (MODE >= 2) {
            __syncthreads();
            if (it & 1) As[tid] = acc[0][0][0] * 1e-30f;
            __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

extern "C" int uemdbg_mfma_rate(float* out, int blocks, int iters, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) mfma_rate_kernel<0><<<blocks, 256, 0, st>>>(out, iters);
    else if (mode == 1) mfma_rate_kernel<1><<<blocks, 256, 0, st>>>(out, iters);
    else mfma_rate_kernel<2><<<blocks, 256, 0, st>>>(out, iters);
    return uem_check_launch("mfma_rate");
}


// Diagnostic only: where do the blocks of a persistent launch land?  Every block records its XCC id and HW_ID (compute unit, shader
// array / engine) and stays resident for `spin` clocks so that the grid fills the chip as a real persistent launch does.
// out[3*b + 0] = XCC_ID, + 1 = HW_ID, + 2 = start time (s_memtime, low 32 bits)
__global__ __launch_bounds__(256) void block_census_kernel(unsigned* out, int spin) {
    extern __shared__ float pad[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[3 * blockIdx.x + 0] = xcc;
        out[3 * blockIdx.x + 1] = hw;
        out[3 * blockIdx.x + 2] = (unsigned)t0;
        pad[0] = 0.f;
    }
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(4);
}
extern "C" int uemdbg_block_census(unsigned* out, int blocks, int lds_bytes, int spin, void* stream) {
    if (!uem_allow_lds((const void*)block_census_kernel, (size_t)lds_bytes)) return uem_check_launch("block_census");
    block_census_kernel<<<blocks, 256, lds_bytes, (hipStream_t)stream>>>(out, spin);
    return uem_check_launch("block_census");
}
