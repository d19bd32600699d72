// Shared host/device helpers for libuemda_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "uemda_hip.h"

#define UEM_WAVE 64

int uem_fail(int code, const char* fmt, ...);
int uem_check_launch(const char* what);
bool uem_allow_lds(const void* kernel, size_t dynamic_lds_bytes);    // api.cpp: once per (kernel, device); false = do not launch

#define UEM_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return uem_fail(UEM_ERR_INVALID, __VA_ARGS__); \
    } while (0)

static inline int64_t uem_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// grid for HBM-bound grid-stride kernels: enough blocks to fill 256 CUs x 8, capped.
static inline int uem_stream_grid(int64_t work_items, int block) {
    int64_t g = uem_cdiv(work_items, block);
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

// Grid of a streaming elementwise pass over a LARGE tensor: one work item (a 16-byte vector) per thread, the grid-stride loop runs once.
// From 128 MB per tensor up this moves 9-13 % more bytes per second than 4096 looping blocks (affine + residual 5.4 -> 6.1 TB/s,
// BatchNorm-backward apply 5.1 -> 5.9 on the 537 MB layer1 tensors; scripts/bench_bn.py); below, where the operands partly live in the
// 256 MB infinity cache, the looping form is as fast or faster.  UEM_FLAT_GRID=0 restores the capped grid everywhere.
static inline int uem_flat_grid(int64_t work_items, int block) {
    static const int on = getenv("UEM_FLAT_GRID") ? atoi(getenv("UEM_FLAT_GRID")) : 1;
    const int64_t g = uem_cdiv(work_items, block);
    if (on && work_items >= ((int64_t)1 << 23) && g < ((int64_t)1 << 31) - 1) return (int)g;
    return uem_stream_grid(work_items, block);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Wave reductions on the DPP path (no LDS traffic: __shfl_xor is ds_bpermute_b32, one LDS-pipe round trip per step and six dependent
// steps per reduction).  Prefix scan inside each 16-lane row (row_shr 1, 2, 4, 8, zero fill), then row_bcast:15 / row_bcast:31 carry
// the row totals up: lane 63 holds the wave's total, read back through v_readlane as a wave-uniform value.  Every lane of the wave must
// be active at the call (EXEC all ones), as for any cross-lane operation.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float old, float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take0(float v) {      // lanes without a source read 0
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_take0<0x111, 0xF>(v);                         // row_shr:1
    v += dpp_take0<0x112, 0xF>(v);                         // row_shr:2
    v += dpp_take0<0x114, 0xF>(v);                         // row_shr:4
    v += dpp_take0<0x118, 0xF>(v);                         // row_shr:8   -> lane 15 of each row = the row's sum
    v += dpp_take<0x142, 0xA>(0.f, v);                     // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xC>(0.f, v);                     // row_bcast:31 into rows 2 and 3 -> lane 63 = the wave's sum
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {   // lanes without a source keep their own value: max(v, v) = v
    v = fmaxf(v, dpp_take<0x111, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x112, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x114, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x118, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_take<0x143, 0xC>(v, v));
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Hardware transcendental forms for the HBM-bound per-pixel kernels (mining, losses), which were VALU-bound on the IEEE forms: a
// correctly rounded `a / b` is ~10 VALU instructions, expf / logf ~12-15, powf ~40; label_refine spent ~900 instructions per pixel on
// 42 divisions and 18 expf against 56 bytes of traffic.  v_rcp_f32 / v_exp_f32 / v_log_f32 are accurate to 1 ulp of their own result;
// the argument scaling of exp adds |x| * 6e-8 relative (x <= 0 everywhere here: softmax shifts), far inside the 1e-5 / 2e-5 bars the
// parity tests hold these kernels to.
__device__ __forceinline__ float fast_rcp(const float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp(const float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_log(const float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
// x^p for x in [0, 1], p > 0:  0^p = 0 (log2(0) = -inf, exp2(-inf) = 0), 1^p = 1 exactly
__device__ __forceinline__ float fast_pow01(const float x, const float p) { return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)); }

// order-preserving float <-> uint key (key(a) < key(b)  <=>  a < b); key 0 is below every float.
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// bilinear source index / lambda, PyTorch area_pixel_compute_source_index semantics.
struct Lerp {
    int i0, i1;
    float l0, l1;
};
// align_corners=True form with the scale (in-1)/(out-1) computed once by the caller: the same arithmetic as lerp_setup(.., true)
__device__ __forceinline__ float lerp_scale_ac(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}
__device__ __forceinline__ Lerp lerp_ac(int dst, int in_size, float scale) {
    const float src = scale * (float)dst;
    int i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    float l1 = src - (float)i0;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    r.l1 = l1;
    r.l0 = 1.f - l1;
    return r;
}
__device__ __forceinline__ Lerp lerp_setup(int dst, int in_size, int out_size, bool align_corners) {
    float src;
    if (align_corners) {
        float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
        src = scale * (float)dst;
    } else {
        float scale = (float)in_size / (float)out_size;
        src = scale * ((float)dst + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
    }
    int i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    float l1 = src - (float)i0;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    r.l1 = l1;
    r.l0 = 1.f - l1;
    return r;
}
