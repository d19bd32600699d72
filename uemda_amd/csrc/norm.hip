// Normalisation / pooling / resampling / layout kernels (all HBM-bound, NHWC fp32).
// Reference: BatchNorm2d at _resnets.py:96-110 & Encoder.py:20,37; MaxPool2d _resnets.py:153;
// InstanceNorm2d Encoder.py:123,147; AdaptiveAvgPool2d + bilinear(align_corners=False) Encoder.py:18,48-51;
// Dropout2d Encoder.py:39.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// thread -> (channel vector, row group) mapping shared by the column-reduction kernels:
// LPR lanes cover one row's channel segment (4 channels per lane), 256/LPR rows per pass.
// ---------------------------------------------------------------------------------------------------------
struct ColMap {
    int lpr, rpp, cv, rg, c0;
};
__device__ __forceinline__ ColMap col_map(int C, int seg) {
    ColMap m;
    int cvecs = C >> 2;
    m.lpr = cvecs < 64 ? cvecs : 64;
    m.rpp = 256 / m.lpr;
    m.cv = threadIdx.x % m.lpr;
    m.rg = threadIdx.x / m.lpr;
    m.c0 = seg * 256 + m.cv * 4;
    return m;
}
static inline bool col_shape_ok(int C) {
    // C/4 must be a power of two <= 64, or a multiple of 64 (so 256 % lpr == 0)
    if (C % 4) return false;
    int cv = C / 4;
    if (cv >= 64) return cv % 64 == 0;
    return (cv & (cv - 1)) == 0;
}

// Chan et al. parallel combination of (n, mean, M2)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    if (nb == 0.f) return;
    const float nt = n + nb;
    const float delta = meanb - mean;
    mean = mean + delta * (nb / nt);
    m2 = m2 + m2b + delta * delta * (n * nb / nt);
    n = nt;
}

// ---------------------------------------------------------------------------------------------------------
// storage types: float (default) or bf16 bits (unsigned short; BASELINE config 5's bf16-storage path).  4 channels
// per access either way; arithmetic is always fp32.
// ---------------------------------------------------------------------------------------------------------
typedef unsigned short bf16_t;
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned f2bf_bits(float f) {                   // round to nearest even; NaN stays NaN
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f2bf_bits(v.x) | (f2bf_bits(v.y) << 16), f2bf_bits(v.z) | (f2bf_bits(v.w) << 16));
}

// 8 bf16 channels per access (16 bytes): what the elementwise bf16 passes use when C % 8 == 0
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = f2bf_bits(v[2 * e]) | (f2bf_bits(v[2 * e + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void ldv8(const float* p, float (&v)[8]) {
    *reinterpret_cast<float4*>(&v[0]) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(&v[4]) = *reinterpret_cast<const float4*>(p + 4);
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm statistics (training mode)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, int M, int C, int ld,
                                                               int rows_per_chunk, float* __restrict__ ws) {
    const ColMap cm = col_map(C, blockIdx.x);
    const int chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
    float n = 0.f;
    bool first = true;
    for (int r = r0 + cm.rg; r < r1; r += cm.rpp) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ld + cm.c0);
        if (first) { K = v; first = false; }
        const float dx = v.x - K.x, dy = v.y - K.y, dz = v.z - K.z, dw = v.w - K.w;
        s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
        s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        n += 1.f;
    }
    // per-thread (n, mean, M2) for 4 channels, then merge the row groups through LDS
    __shared__ float sh[256][9];
    float mean[4], m2[4];
    const float inv = n > 0.f ? 1.f / n : 0.f;
    mean[0] = K.x + s1.x * inv; mean[1] = K.y + s1.y * inv; mean[2] = K.z + s1.z * inv; mean[3] = K.w + s1.w * inv;
    m2[0] = s2.x - s1.x * s1.x * inv; m2[1] = s2.y - s1.y * s1.y * inv;
    m2[2] = s2.z - s1.z * s1.z * inv; m2[3] = s2.w - s1.w * s1.w * inv;
    sh[threadIdx.x][0] = n;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[threadIdx.x][1 + j] = mean[j]; sh[threadIdx.x][5 + j] = m2[j]; }
    __syncthreads();
    if (cm.rg == 0) {
        float nn = n;
        for (int g = 1; g < cm.rpp; ++g) {
            const int t = g * cm.lpr + cm.cv;
            const float nb = sh[t][0];
            float ntmp;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ntmp = nn; chan_merge(ntmp, mean[j], m2[j], nb, sh[t][1 + j], sh[t][5 + j]); }
            nn += nb;
        }
        // ws layout: [chunk][3][C]  (n, mean, m2)
        float* w = ws + (size_t)chunk * 3 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[cm.c0 + j] = nn;
            w[C + cm.c0 + j] = mean[j];
            w[2 * C + cm.c0 + j] = m2[j];
        }
    }
}
// chunking shared by the column reductions: enough (segment x chunk) blocks to fill 256 CUs several times
static inline void col_chunks(int M, int C, int* chunks, int* rows_per_chunk) {
    const int segs = (int)uem_cdiv(C, 256);
    int ch = (int)uem_cdiv(2048, segs);
    int rpc = (int)uem_cdiv(M, ch);
    if (rpc < 64) rpc = 64;
    *rows_per_chunk = rpc;
    *chunks = (int)uem_cdiv(M, rpc);
}
extern "C" int64_t uem_bn_workspace_floats(int M, int C) {
    int chunks, rpc;
    col_chunks(M, C, &chunks, &rpc);
    return (int64_t)3 * C * chunks;
}

// one wave per channel: lanes merge chunks lane, lane+64, ... then a shuffle tree (fixed order => deterministic)
__global__ __launch_bounds__(64) void bn_stats_finalize_kernel(const float* __restrict__ ws, int chunks, int M, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float momentum, float* __restrict__ rmean,
                                                               float* __restrict__ rvar, float* __restrict__ smean,
                                                               float* __restrict__ sinv, float* __restrict__ scale,
                                                               float* __restrict__ shift) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int j = lane; j < chunks; j += 64) {
        const float* w = ws + (size_t)j * 3 * C;
        const float nb = w[c];
        if (n == 0.f) { n = nb; mean = w[C + c]; m2 = w[2 * C + c]; }
        else chan_merge(n, mean, m2, nb, w[C + c], w[2 * C + c]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
        // symmetric merge so that both partners end with the same value
        const float nt = n + nb;
        if (nt > 0.f) {
            const float delta = mb - mean;
            const float mnew = (n * mean + nb * mb) / nt;
            m2 = m2 + qb + delta * delta * (n * nb / nt);
            mean = mnew;
        }
        n = nt;
    }
    if (lane != 0) return;
    const float var = m2 / (float)M;                       // biased (normalisation)
    const float invstd = 1.0f / sqrtf(var + eps);
    if (smean) smean[c] = mean;
    if (sinv) sinv[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - mean * sc;
    if (rmean) {
        const float unbiased = M > 1 ? m2 / (float)(M - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
}
// statistics from the conv epilogue's per-tile column sums: tile_stats[0][c][tile] = sum y, [1][c][tile] = sum y*y
// over the tile's `rows` rows; tiles are merged with the same Chan combination as the stand-alone path
// block = one channel, 256 threads.  Every tile holds the same number of rows, so the merged moments are plain sums:
// mean = sum_t s1_t / M and M2 = sum_t [ (s2_t - s1_t^2 / rows) + rows * (s1_t / rows - mean)^2 ] (Chan's pairwise formula
// telescoped; every term is >= 0) -- two passes over the channel's 2 x tiles partials (L2-resident), no chain of dependent
// divisions as in a sequential merge (that chain made this 13 us per launch, 106 launches per step).
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                               // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ void bn_stats_finish(const int c, const float mean, const float m2, const int M, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const float eps, const float momentum,
                                                float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ smean,
                                                float* __restrict__ sinv, float* __restrict__ scale, float* __restrict__ shift) {
    const float var = m2 / (float)M;
    const float invstd = 1.0f / sqrtf(var + eps);
    if (smean) smean[c] = mean;
    if (sinv) sinv[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - mean * sc;
    if (rmean) {
        const float unbiased = M > 1 ? m2 / (float)(M - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
}
__global__ __launch_bounds__(256) void bn_stats_tiles_finalize_kernel(const float* __restrict__ ts, int tiles, int rows, int M, int C,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float eps, float momentum, float* __restrict__ rmean,
                                                                      float* __restrict__ rvar, float* __restrict__ smean,
                                                                      float* __restrict__ sinv, float* __restrict__ scale,
                                                                      float* __restrict__ shift) {
    __shared__ float red[4];
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* const t1 = ts + (size_t)c * tiles;
    const float* const t2 = ts + ((size_t)C + c) * tiles;
    const float nb = (float)rows, inb = 1.0f / nb;
    float a = 0.f, q = 0.f;
    if (tiles <= 16 * 256) {
        // both partial rows in registers after ONE round trip to memory (these launches sit between two convolutions with C blocks on
        // the whole chip: their time is latency); same sums in the same order as the two-pass form below
        float r1[16], r2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = lane + 256 * k;
            r1[k] = j < tiles ? t1[j] : 0.f;
            r2[k] = j < tiles ? t2[j] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (lane + 256 * k < tiles) a += r1[k];
        const float mean = block_sum_256(a, red) / (float)M;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (lane + 256 * k < tiles) {
                const float s1 = r1[k], s2 = r2[k];
                const float mb = s1 * inb, d = mb - mean;
                q += fmaxf(s2 - s1 * mb, 0.f) + nb * d * d;
            }
        }
        const float m2 = block_sum_256(q, red);
        if (lane != 0) return;
        bn_stats_finish(c, mean, m2, M, gamma, beta, eps, momentum, rmean, rvar, smean, sinv, scale, shift);
        return;
    }
    for (int j = lane; j < tiles; j += 256) a += t1[j];
    const float mean = block_sum_256(a, red) / (float)M;
    for (int j = lane; j < tiles; j += 256) {
        const float s1 = t1[j], s2 = t2[j];
        const float mb = s1 * inb, d = mb - mean;
        q += fmaxf(s2 - s1 * mb, 0.f) + nb * d * d;
    }
    const float m2 = block_sum_256(q, red);
    if (lane != 0) return;
    bn_stats_finish(c, mean, m2, M, gamma, beta, eps, momentum, rmean, rvar, smean, sinv, scale, shift);
}
extern "C" int uem_bn_stats_from_tiles(const float* tile_stats, int tiles, int M, int C, const float* gamma, const float* beta,
                                       float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                       float* save_invstd, float* scale, float* shift, void* stream) {
    UEM_REQUIRE(tile_stats && scale && shift && tiles > 0 && M == tiles * 128 && C > 0, "bn_stats_from_tiles: bad arguments");
    UEM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_from_tiles: running stats must come in pairs");
    bn_stats_tiles_finalize_kernel<<<C, 256, 0, (hipStream_t)stream>>>(tile_stats, tiles, 128, M, C, gamma, beta, eps, momentum,
                                                                      running_mean, running_var, save_mean, save_invstd, scale, shift);
    return uem_check_launch("bn_stats_from_tiles");
}

extern "C" int uem_bn_stats(const float* x, int M, int C, int ld, const float* gamma, const float* beta, float eps,
                            float momentum, float* running_mean, float* running_var, float* save_mean,
                            float* save_invstd, float* scale, float* shift, float* workspace, void* stream) {
    UEM_REQUIRE(x && scale && shift && workspace, "bn_stats: null pointer");
    UEM_REQUIRE(M > 0 && col_shape_ok(C) && ld >= C && (ld % 4) == 0, "bn_stats: unsupported shape M=%d C=%d ld=%d", M, C, ld);
    UEM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running stats must come in pairs");
    hipStream_t st = (hipStream_t)stream;
    int chunks, rpc;
    col_chunks(M, C, &chunks, &rpc);
    dim3 grid((unsigned)uem_cdiv(C, 256), (unsigned)chunks);
    bn_stats_partial_kernel<<<grid, 256, 0, st>>>(x, M, C, ld, rpc, workspace);
    bn_stats_finalize_kernel<<<C, 64, 0, st>>>(workspace, chunks, M, C, gamma, beta, eps, momentum, running_mean, running_var,
                                               save_mean, save_invstd, scale, shift);
    return uem_check_launch("bn_stats");
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                                      float* __restrict__ invstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
    if (mean) mean[c] = rm[c];
    if (invstd) invstd[c] = is;
}
extern "C" int uem_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  float eps, float* scale, float* shift, float* mean, float* invstd, int C, void* stream) {
    UEM_REQUIRE(running_mean && running_var && scale && shift && C > 0, "bn_eval_affine: bad arguments");
    bn_eval_affine_kernel<<<(int)uem_cdiv(C, 64), 64, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps, scale, shift,
                                                                                mean, invstd, C);
    return uem_check_launch("bn_eval_affine");
}

// ---------------------------------------------------------------------------------------------------------
// y = act(x*scale + shift (+ res))
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res,
                                                         const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                         T* __restrict__ y, int64_t nvec, int C, int relu,
                                                         uint32_t* __restrict__ bits) {
    // `bits` (optional; needs nvec % 8 == 0): one bit per element, bit (e & 31) of word e >> 5 = [y_e > 0].  The
    // backward passes read these 1/32-size words instead of the materialised output.
    const int64_t nloop = bits ? (nvec + 7) / 8 * 8 : nvec;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nloop; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        float4 v = ld4<T>(x + i * 4);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (res) {
            float4 r = ld4<T>(res + i * 4);
            if (rscale) {
                const float4 rs = *reinterpret_cast<const float4*>(rscale + c);
                const float4 rt = *reinterpret_cast<const float4*>(rshift + c);
                r.x = r.x * rs.x + rt.x; r.y = r.y * rs.y + rt.y; r.z = r.z * rs.z + rt.z; r.w = r.w * rs.w + rt.w;
            }
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        st4<T>(y + i * 4, v);
        if (bits) {
            // 8 consecutive lanes hold the 32 elements of one word (i is lane-aligned: grid stride is a multiple of 256)
            uint32_t m = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            m <<= 4 * (threadIdx.x & 7);
            m |= __shfl_xor(m, 1);
            m |= __shfl_xor(m, 2);
            m |= __shfl_xor(m, 4);
            if ((threadIdx.x & 7) == 0) bits[i >> 3] = m;
        }
    }
}
__global__ __launch_bounds__(256) void affine_act_bf16x8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const bf16_t* __restrict__ res,
                                                                const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                                bf16_t* __restrict__ y, int64_t nvec, int C, int relu,
                                                                uint32_t* __restrict__ bits) {
    // same arithmetic as affine_act_kernel<bf16_t>, 8 channels (16 bytes) per lane; with `bits` 4 consecutive lanes hold one word
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 8) % C);
        float v[8], sc[8], sh[8];
        ld8(x + i * 8, v);
        ldv8(scale + c, sc);
        ldv8(shift + c, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (res) {
            float r[8];
            ld8(res + i * 8, r);
            if (rscale) {
                ldv8(rscale + c, sc);
                ldv8(rshift + c, sh);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = r[e] * sc[e] + sh[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        st8(y + i * 8, v);
        if (bits) {
            uint32_t m = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) m |= (v[e] > 0.f ? 1u : 0u) << e;
            m <<= 8 * (threadIdx.x & 3);
            m |= __shfl_xor(m, 1);
            m |= __shfl_xor(m, 2);
            if ((threadIdx.x & 3) == 0) bits[i >> 2] = m;
        }
    }
}
// R rows per lane for the two passes above on large tensors (bn_rows below): per-channel vectors loaded once per lane, the rows' loads
// in flight together.  Same arithmetic; the bit words come out the same way (a row's vectors are lane-aligned: the row stride is a
// multiple of 256).
template <typename T, int R>
__global__ __launch_bounds__(256) void affine_act_rows_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const T* __restrict__ res,
                                                              const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                              T* __restrict__ y, int64_t nvec, int C, int relu,
                                                              uint32_t* __restrict__ bits) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)((i0 * 4) % C);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    float4 rs = make_float4(1.f, 1.f, 1.f, 1.f), rt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res && rscale) { rs = *reinterpret_cast<const float4*>(rscale + c); rt = *reinterpret_cast<const float4*>(rshift + c); }
    float4 v[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i < nvec) {
            v[r] = ld4<T>(x + i * 4);
            if (res) q[r] = ld4<T>(res + i * 4);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i >= nvec) continue;                                        // nvec % 8 == 0 with bits: whole words inside or outside
        float4 a = v[r];
        a.x = a.x * sc.x + sh.x; a.y = a.y * sc.y + sh.y; a.z = a.z * sc.z + sh.z; a.w = a.w * sc.w + sh.w;
        if (res) {
            float4 b = q[r];
            if (rscale) { b.x = b.x * rs.x + rt.x; b.y = b.y * rs.y + rt.y; b.z = b.z * rs.z + rt.z; b.w = b.w * rs.w + rt.w; }
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        st4<T>(y + i * 4, a);
        if (bits) {
            uint32_t m = (a.x > 0.f ? 1u : 0u) | (a.y > 0.f ? 2u : 0u) | (a.z > 0.f ? 4u : 0u) | (a.w > 0.f ? 8u : 0u);
            m <<= 4 * (threadIdx.x & 7);
            m |= __shfl_xor(m, 1);
            m |= __shfl_xor(m, 2);
            m |= __shfl_xor(m, 4);
            if ((threadIdx.x & 7) == 0) bits[i >> 3] = m;
        }
    }
}
template <int R>
__global__ __launch_bounds__(256) void affine_act_bf16x8_rows_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, const bf16_t* __restrict__ res,
                                                                     const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                                     bf16_t* __restrict__ y, int64_t nvec, int C, int relu,
                                                                     uint32_t* __restrict__ bits) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)((i0 * 8) % C);
    float sc[8], sh[8], rs[8], rt[8];
    ldv8(scale + c, sc);
    ldv8(shift + c, sh);
    if (res && rscale) { ldv8(rscale + c, rs); ldv8(rshift + c, rt); }
    float v[R][8], q[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i < nvec) {
            ld8(x + i * 8, v[r]);
            if (res) ld8(res + i * 8, q[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i >= nvec) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[r][e] = v[r][e] * sc[e] + sh[e];
        if (res) {
            if (rscale) {
#pragma unroll
                for (int e = 0; e < 8; ++e) q[r][e] = q[r][e] * rs[e] + rt[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[r][e] += q[r][e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[r][e] = fmaxf(v[r][e], 0.f);
        }
        st8(y + i * 8, v[r]);
        if (bits) {
            uint32_t m = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) m |= (v[r][e] > 0.f ? 1u : 0u) << e;
            m <<= 8 * (threadIdx.x & 3);
            m |= __shfl_xor(m, 1);
            m |= __shfl_xor(m, 2);
            if ((threadIdx.x & 3) == 0) bits[i >> 2] = m;
        }
    }
}
static inline int bn_rows(int64_t nvec, int C, int per_vec);
extern "C" int uem_affine_act(const float* x, const float* scale, const float* shift, const float* res,
                              const float* res_scale, const float* res_shift, float* y, int64_t M, int C, int relu,
                              uint32_t* relu_bits, void* stream) {
    UEM_REQUIRE(x && scale && shift && y && M > 0 && C > 0 && (C % 4) == 0, "affine_act: bad arguments");
    UEM_REQUIRE(!relu_bits || (C % 32) == 0, "affine_act: relu_bits needs C %% 32 == 0 (C=%d)", C);
    UEM_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (!res_scale || res), "affine_act: bad residual affine");
    const int64_t nvec = M * C / 4;
    // (fp32: one row per lane unless asked for -- the block-output affine ran at 5.9 TB/s as it was and drops to 5.7 with two rows per
    // lane, where the backward apply gains, 5.7 -> 6.1, and both bf16 passes do: UEM_BN_ROWS_AFFINE_F32)
    static const int f32_rows = getenv("UEM_BN_ROWS_AFFINE_F32") ? atoi(getenv("UEM_BN_ROWS_AFFINE_F32")) : 1;
    const int rows = (f32_rows > 1 && (!relu_bits || nvec % 8 == 0)) ? bn_rows(nvec, C, 4) : 1;
    if (rows == 4)
        affine_act_rows_kernel<float, 4><<<(unsigned)uem_cdiv(nvec, 1024), 256, 0, (hipStream_t)stream>>>(x, scale, shift, res, res_scale, res_shift, y, nvec, C, relu, relu_bits);
    else if (rows == 2)
        affine_act_rows_kernel<float, 2><<<(unsigned)uem_cdiv(nvec, 512), 256, 0, (hipStream_t)stream>>>(x, scale, shift, res, res_scale, res_shift, y, nvec, C, relu, relu_bits);
    else
        affine_act_kernel<float><<<uem_flat_grid(nvec, 256), 256, 0, (hipStream_t)stream>>>(x, scale, shift, res, res_scale, res_shift, y, nvec, C, relu, relu_bits);
    return uem_check_launch("affine_act");
}
extern "C" int uem_affine_act_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* res,
                                   const float* res_scale, const float* res_shift, uint16_t* y, int64_t M, int C, int relu,
                                   uint32_t* relu_bits, void* stream) {
    UEM_REQUIRE(x && scale && shift && y && M > 0 && C > 0 && (C % 4) == 0, "affine_act_bf16: bad arguments");
    UEM_REQUIRE(!relu_bits || (C % 32) == 0, "affine_act_bf16: relu_bits needs C %% 32 == 0 (C=%d)", C);
    UEM_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (!res_scale || res), "affine_act_bf16: bad residual affine");
    if (C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0) {       // with bits: C % 32 == 0, whole lane quads
        const int64_t nvec8 = M * C / 8;
        const int rows = (!relu_bits || nvec8 % 4 == 0) ? bn_rows(nvec8, C, 8) : 1;
        if (rows == 4)
            affine_act_bf16x8_rows_kernel<4><<<(unsigned)uem_cdiv(nvec8, 1024), 256, 0, (hipStream_t)stream>>>(x, scale, shift, res, res_scale, res_shift, y, nvec8, C, relu, relu_bits);
        else if (rows == 2)
            affine_act_bf16x8_rows_kernel<2><<<(unsigned)uem_cdiv(nvec8, 512), 256, 0, (hipStream_t)stream>>>(x, scale, shift, res, res_scale, res_shift, y, nvec8, C, relu, relu_bits);
        else
            affine_act_bf16x8_kernel<<<uem_flat_grid(nvec8, 256), 256, 0, (hipStream_t)stream>>>(x, scale, shift, res, res_scale, res_shift, y,
                                                                                                   nvec8, C, relu, relu_bits);
        return uem_check_launch("affine_act_bf16");
    }
    const int64_t nvec = M * C / 4;
    affine_act_kernel<bf16_t><<<uem_flat_grid(nvec, 256), 256, 0, (hipStream_t)stream>>>(x, scale, shift, res, res_scale, res_shift, y, nvec, C, relu, relu_bits);
    return uem_check_launch("affine_act_bf16");
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm backward (training stats).  dp = dy * [relu mask];  xhat = (x - mean) * invstd
// ---------------------------------------------------------------------------------------------------------
// relu: 1 = mask from the materialised output `ymask` (float) or, without it, from x*scale+shift recomputed;
//       2 (UEM_RELU_BITS) = `ymask` points at the packed sign bits written by uem_affine_act
// Gradient of a 3x3 / stride 2 / pad 1 max-pool read in GATHER form: input pixel (iy, ix) collects from the <= 2x2 windows that
// contain it the output gradients whose argmax tap (idx, 0..8 row-major) is this pixel.
__device__ __forceinline__ float4 pool_grad4(const float* __restrict__ dy, const uint8_t* __restrict__ idx, int n, int iy, int ix, int c, int C,
                                             int Ho, int Wo) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int oy0 = iy / 2, oy1 = (iy + 1) / 2;   // candidates: windows starting at 2*oy-1 covering iy
    const int ox0 = ix / 2, ox1 = (ix + 1) / 2;
    for (int oy = oy0; oy <= oy1; ++oy) {
        if (oy >= Ho) continue;
        const int ky = iy - (oy * 2 - 1);
        if (ky < 0 || ky > 2) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (ox >= Wo) continue;
            const int kx = ix - (ox * 2 - 1);
            if (kx < 0 || kx > 2) continue;
            const unsigned char k = (unsigned char)(ky * 3 + kx);
            const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + c;
            const uchar4 bi = *reinterpret_cast<const uchar4*>(idx + o);
            const float4 g = *reinterpret_cast<const float4*>(dy + o);
            if (bi.x == k) acc.x += g.x;
            if (bi.y == k) acc.y += g.y;
            if (bi.z == k) acc.z += g.z;
            if (bi.w == k) acc.w += g.w;
        }
    }
    return acc;
}
__device__ __forceinline__ float4 relu_mask4(float4 dy, float4 x, float4 sc, float4 sh, const float* ymask, size_t off,
                                             int relu = 1) {
    float4 pre;
    if (relu == 2) {
        const uint32_t m = reinterpret_cast<const uint32_t*>(ymask)[off >> 5] >> (off & 31);
        dy.x = (m & 1u) ? dy.x : 0.f; dy.y = (m & 2u) ? dy.y : 0.f; dy.z = (m & 4u) ? dy.z : 0.f; dy.w = (m & 8u) ? dy.w : 0.f;
        return dy;
    }
    if (ymask) pre = *reinterpret_cast<const float4*>(ymask + off);       // materialised relu output
    else { pre.x = x.x * sc.x + sh.x; pre.y = x.y * sc.y + sh.y; pre.z = x.z * sc.z + sh.z; pre.w = x.w * sc.w + sh.w; }
    dy.x = pre.x > 0.f ? dy.x : 0.f; dy.y = pre.y > 0.f ? dy.y : 0.f;
    dy.z = pre.z > 0.f ? dy.z : 0.f; dy.w = pre.w > 0.f ? dy.w : 0.f;
    return dy;
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             const float* __restrict__ res, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ smean,
                                                             const float* __restrict__ sinv, int M, int C, int relu,
                                                             int rows_per_chunk, float* __restrict__ ws) {
    const ColMap cm = col_map(C, blockIdx.x);
    const int chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const float4 sc = *reinterpret_cast<const float4*>(scale + cm.c0);
    const float4 sh = *reinterpret_cast<const float4*>(shift + cm.c0);
    const float4 mu = *reinterpret_cast<const float4*>(smean + cm.c0);
    const float4 is = *reinterpret_cast<const float4*>(sinv + cm.c0);
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
    for (int r = r0 + cm.rg; r < r1; r += cm.rpp) {
        const size_t off = (size_t)r * C + cm.c0;
        const float4 xv = ld4<T>(x + off);
        float4 d = ld4<T>(dy + off);
        if (relu) d = relu_mask4(d, xv, sc, sh, res, off, relu);
        sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
        sg.x += d.x * ((xv.x - mu.x) * is.x); sg.y += d.y * ((xv.y - mu.y) * is.y);
        sg.z += d.z * ((xv.z - mu.z) * is.z); sg.w += d.w * ((xv.w - mu.w) * is.w);
    }
    __shared__ float sh2[256][8];
    sh2[threadIdx.x][0] = sb.x; sh2[threadIdx.x][1] = sb.y; sh2[threadIdx.x][2] = sb.z; sh2[threadIdx.x][3] = sb.w;
    sh2[threadIdx.x][4] = sg.x; sh2[threadIdx.x][5] = sg.y; sh2[threadIdx.x][6] = sg.z; sh2[threadIdx.x][7] = sg.w;
    __syncthreads();
    if (cm.rg == 0) {
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = sh2[threadIdx.x][j];
        for (int g = 1; g < cm.rpp; ++g) {
            const int t = g * cm.lpr + cm.cv;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += sh2[t][j];
        }
        float* w = ws + (size_t)chunk * 2 * C;      // [chunk][2][C] : dbeta, dgamma
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[cm.c0 + j] = a[j]; w[C + cm.c0 + j] = a[4 + j]; }
    }
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ ws, int chunks, int C,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ acc_gamma, float* __restrict__ acc_beta) {
    __shared__ float red[4];
    const int c = blockIdx.x, lane = threadIdx.x;
    float b = 0.f, g = 0.f;
    for (int j = lane; j < chunks; j += 256) { b += ws[(size_t)j * 2 * C + c]; g += ws[(size_t)j * 2 * C + C + c]; }
    b = block_sum_256(b, red);
    g = block_sum_256(g, red);
    if (lane == 0) {
        dbeta[c] = b;
        dgamma[c] = g;
        // parameter-gradient accumulation (stream-ordered: the two graphs of a step run back to back)
        if (acc_gamma) acc_gamma[c] += g;
        if (acc_beta) acc_beta[c] += b;
    }
}
// dgamma / dbeta from the data-gradient epilogue's per-tile partials: tiles[0][c][tile] = sum dp, [1][c][tile] = sum dp*xhat
__global__ __launch_bounds__(256) void bn_bwd_tiles_finalize_kernel(const float* __restrict__ tp, int tiles, int C,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    float* __restrict__ acc_gamma, float* __restrict__ acc_beta) {
    __shared__ float red[4];
    const int c = blockIdx.x, lane = threadIdx.x;
    float b = 0.f, g = 0.f;
    for (int j = lane; j < tiles; j += 256) { b += tp[(size_t)c * tiles + j]; g += tp[((size_t)C + c) * tiles + j]; }
    b = block_sum_256(b, red);
    g = block_sum_256(g, red);
    if (lane == 0) {
        dbeta[c] = b;
        dgamma[c] = g;
        if (acc_gamma) acc_gamma[c] += g;
        if (acc_beta) acc_beta[c] += b;
    }
}
extern "C" int uem_bn_bwd_from_tiles(const float* tile_partials, int tiles, int C, float* dgamma, float* dbeta, float* grad_gamma,
                                     float* grad_beta, void* stream) {
    UEM_REQUIRE(tile_partials && dgamma && dbeta && tiles > 0 && C > 0, "bn_bwd_from_tiles: bad arguments");
    bn_bwd_tiles_finalize_kernel<<<C, 256, 0, (hipStream_t)stream>>>(tile_partials, tiles, C, dgamma, dbeta, grad_gamma, grad_beta);
    return uem_check_launch("bn_bwd_from_tiles");
}
extern "C" int uem_bn_bwd_reduce(const float* x, const float* dy, const void* ymask, const float* scale, const float* shift,
                                 const float* save_mean, const float* save_invstd, int M, int C, int relu, float* dgamma,
                                 float* dbeta, float* grad_gamma, float* grad_beta, float* workspace, void* stream) {
    UEM_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && dgamma && dbeta && workspace, "bn_bwd_reduce: null pointer");
    UEM_REQUIRE(M > 0 && col_shape_ok(C), "bn_bwd_reduce: unsupported shape M=%d C=%d", M, C);
    UEM_REQUIRE(relu != UEM_RELU_BITS || (ymask && C % 32 == 0), "bn_bwd_reduce: UEM_RELU_BITS needs the bit mask and C %% 32 == 0");
    hipStream_t st = (hipStream_t)stream;
    int chunks, rpc;
    col_chunks(M, C, &chunks, &rpc);
    dim3 grid((unsigned)uem_cdiv(C, 256), (unsigned)chunks);
    bn_bwd_partial_kernel<float><<<grid, 256, 0, st>>>(x, dy, (const float*)ymask, scale, shift, save_mean, save_invstd, M, C, relu, rpc, workspace);
    bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(workspace, chunks, C, dgamma, dbeta, grad_gamma, grad_beta);
    return uem_check_launch("bn_bwd_reduce");
}
extern "C" int uem_bn_bwd_reduce_bf16(const uint16_t* x, const uint16_t* dy, const uint32_t* relu_bits, const float* scale,
                                      const float* shift, const float* save_mean, const float* save_invstd, int M, int C, int relu,
                                      float* dgamma, float* dbeta, float* grad_gamma, float* grad_beta, float* workspace, void* stream) {
    UEM_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && dgamma && dbeta && workspace, "bn_bwd_reduce_bf16: null pointer");
    UEM_REQUIRE(M > 0 && col_shape_ok(C), "bn_bwd_reduce_bf16: unsupported shape M=%d C=%d", M, C);
    UEM_REQUIRE(relu == 0 || relu == 1 || (relu == UEM_RELU_BITS && relu_bits && C % 32 == 0), "bn_bwd_reduce_bf16: relu is 0, 1 (mask recomputed from x) or UEM_RELU_BITS");
    UEM_REQUIRE(relu != 1 || relu_bits == nullptr, "bn_bwd_reduce_bf16: relu = 1 recomputes the mask, pass no bits");
    hipStream_t st = (hipStream_t)stream;
    int chunks, rpc;
    col_chunks(M, C, &chunks, &rpc);
    dim3 grid((unsigned)uem_cdiv(C, 256), (unsigned)chunks);
    bn_bwd_partial_kernel<bf16_t><<<grid, 256, 0, st>>>(x, dy, (const float*)relu_bits, scale, shift, save_mean, save_invstd, M, C, relu, rpc, workspace);
    bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(workspace, chunks, C, dgamma, dbeta, grad_gamma, grad_beta);
    return uem_check_launch("bn_bwd_reduce_bf16");
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ res, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ smean,
                                                           const float* __restrict__ sinv, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, int64_t nvec, int C, float invM,
                                                           int relu, T* __restrict__ dx, T* __restrict__ dres) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const size_t off = (size_t)i * 4;
        const float4 xv = ld4<T>(x + off);
        float4 d = ld4<T>(dy + off);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        if (relu) d = relu_mask4(d, xv, sc, *reinterpret_cast<const float4*>(shift + c), res, off, relu);
        if (dres) st4<T>(dres + off, d);
        const float4 mu = *reinterpret_cast<const float4*>(smean + c);
        const float4 is = *reinterpret_cast<const float4*>(sinv + c);
        const float4 dg = *reinterpret_cast<const float4*>(dgamma + c);
        const float4 db = *reinterpret_cast<const float4*>(dbeta + c);
        float4 o;
        o.x = sc.x * (d.x - db.x * invM - ((xv.x - mu.x) * is.x) * (dg.x * invM));
        o.y = sc.y * (d.y - db.y * invM - ((xv.y - mu.y) * is.y) * (dg.y * invM));
        o.z = sc.z * (d.z - db.z * invM - ((xv.z - mu.z) * is.z) * (dg.z * invM));
        o.w = sc.w * (d.w - db.w * invM - ((xv.w - mu.w) * is.w) * (dg.w * invM));
        st4<T>(dx + off, o);
    }
}
// R rows per lane (see bn_bwd_apply_bf16x8_rows_kernel below): the lane's per-channel vectors loaded once, the loads of all R rows in
// flight together; element arithmetic identical to the kernel above
template <typename T, int R>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                const float* __restrict__ res, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ smean,
                                                                const float* __restrict__ sinv, const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, int64_t nvec, int C, float invM,
                                                                int relu, T* __restrict__ dx, T* __restrict__ dres) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)((i0 * 4) % C);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = relu == 1 ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 mu = *reinterpret_cast<const float4*>(smean + c);
    const float4 is = *reinterpret_cast<const float4*>(sinv + c);
    float4 dg = *reinterpret_cast<const float4*>(dgamma + c);
    float4 db = *reinterpret_cast<const float4*>(dbeta + c);
    dg.x *= invM; dg.y *= invM; dg.z *= invM; dg.w *= invM;
    db.x *= invM; db.y *= invM; db.z *= invM; db.w *= invM;
    float4 xv[R], d[R], pre[R];
    uint32_t m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i < nvec) {
            const size_t off = (size_t)i * 4;
            xv[r] = ld4<T>(x + off);
            d[r] = ld4<T>(dy + off);
            if (relu == 2) m[r] = reinterpret_cast<const uint32_t*>(res)[off >> 5];
            else if (relu && res) pre[r] = *reinterpret_cast<const float4*>(res + off);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i >= nvec) continue;
        const size_t off = (size_t)i * 4;
        float4 g = d[r];
        const float4 v = xv[r];
        if (relu == 2) {
            const uint32_t mm = m[r] >> (off & 31);
            g.x = (mm & 1u) ? g.x : 0.f; g.y = (mm & 2u) ? g.y : 0.f; g.z = (mm & 4u) ? g.z : 0.f; g.w = (mm & 8u) ? g.w : 0.f;
        } else if (relu) {
            float4 q;
            if (res) q = pre[r];
            else { q.x = v.x * sc.x + sh.x; q.y = v.y * sc.y + sh.y; q.z = v.z * sc.z + sh.z; q.w = v.w * sc.w + sh.w; }
            g.x = q.x > 0.f ? g.x : 0.f; g.y = q.y > 0.f ? g.y : 0.f; g.z = q.z > 0.f ? g.z : 0.f; g.w = q.w > 0.f ? g.w : 0.f;
        }
        if (dres) st4<T>(dres + off, g);
        float4 o;
        o.x = sc.x * (g.x - db.x - ((v.x - mu.x) * is.x) * dg.x);
        o.y = sc.y * (g.y - db.y - ((v.y - mu.y) * is.y) * dg.y);
        o.z = sc.z * (g.z - db.z - ((v.z - mu.z) * is.z) * dg.z);
        o.w = sc.w * (g.w - db.w - ((v.w - mu.w) * is.w) * dg.w);
        st4<T>(dx + off, o);
    }
}
// rows per lane of the elementwise BatchNorm passes on large tensors: UEM_BN_ROWS (1 = the grid-stride kernels)
// A lane keeps ONE set of per-channel vectors for all its rows: the row stride (gridDim * 256 vectors of per_vec elements) must be a
// multiple of C, which an odd block count breaks at C = 2048 with 4-element vectors (odd M = N*H*W) -> one row per lane there.
static inline int bn_rows(int64_t nvec, int C, int per_vec) {
    static const int rows = getenv("UEM_BN_ROWS") ? atoi(getenv("UEM_BN_ROWS")) : 2;
    if (rows < 2 || nvec < ((int64_t)1 << 20) || C < 64 || C > 2048 || (C & (C - 1)) != 0) return 1;
    const int R = rows >= 4 ? 4 : 2;
    if ((uem_cdiv(nvec, (int64_t)256 * R) * 256 * per_vec) % C != 0) return 1;
    return R;
}
// the paired apply kernels are instantiated for two rows per lane ONLY, whatever UEM_BN_ROWS says for the single passes: their
// admission test is the row-stride condition of THEIR grid (under UEM_BN_ROWS=4 the stride was checked for a grid of
// cdiv(nvec, 1024) blocks while cdiv(nvec, 512) were launched)
static inline bool bn_pair_rows_ok(int64_t nvec, int C, int per_vec) {
    static const int rows = getenv("UEM_BN_ROWS") ? atoi(getenv("UEM_BN_ROWS")) : 2;
    if (rows < 2 || nvec < ((int64_t)1 << 20) || C < 64 || C > 2048 || (C & (C - 1)) != 0) return false;
    return (uem_cdiv(nvec, (int64_t)512) * 256 * per_vec) % C == 0;
}
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16x8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                  const uint32_t* __restrict__ rbits, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ smean,
                                                                  const float* __restrict__ sinv, const float* __restrict__ dgamma,
                                                                  const float* __restrict__ dbeta, int64_t nvec, int C, float invM,
                                                                  int relu, bf16_t* dx, bf16_t* __restrict__ dres) {
    // same arithmetic as bn_bwd_apply_kernel<bf16_t>, 8 channels (16 bytes) per lane; dx may alias dy (element-wise in place)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 8) % C);
        const size_t off = (size_t)i * 8;
        float xv[8], d[8], sc[8], t[8];
        ld8(x + off, xv);
        ld8(dy + off, d);
        ldv8(scale + c, sc);
        if (relu == 2) {
            const uint32_t m = rbits[off >> 5] >> (off & 31);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = ((m >> e) & 1u) ? d[e] : 0.f;
        } else if (relu) {
            ldv8(shift + c, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = (xv[e] * sc[e] + t[e] > 0.f) ? d[e] : 0.f;
        }
        if (dres) st8(dres + off, d);
        float mu[8], is[8], dg[8], db[8], o[8];
        ldv8(smean + c, mu);
        ldv8(sinv + c, is);
        ldv8(dgamma + c, dg);
        ldv8(dbeta + c, db);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sc[e] * (d[e] - db[e] * invM - ((xv[e] - mu[e]) * is[e]) * (dg[e] * invM));
        st8(dx + off, o);
    }
}
// The same pass with R rows per lane: with 8 channels per 16 bytes of tensor the per-channel vectors are up to 12 load instructions beside
// 2 loads and a store of data; a lane's channels are the same in every row it visits (the row stride, gridDim * 256 vectors, is a multiple
// of C / 8 for the power-of-two channel counts of the encoder), so they are loaded once.  Element arithmetic identical to the kernel above.
template <int R>
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16x8_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                       const uint32_t* __restrict__ rbits, const float* __restrict__ scale,
                                                                       const float* __restrict__ shift, const float* __restrict__ smean,
                                                                       const float* __restrict__ sinv, const float* __restrict__ dgamma,
                                                                       const float* __restrict__ dbeta, int64_t nvec, int C, float invM,
                                                                       int relu, bf16_t* dx, bf16_t* __restrict__ dres) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)((i0 * 8) % C);
    float sc[8], t[8], mu[8], is[8], dgm[8], dbm[8];
    ldv8(scale + c, sc);
    if (relu == 1) ldv8(shift + c, t);
    ldv8(smean + c, mu);
    ldv8(sinv + c, is);
    ldv8(dgamma + c, dgm);
    ldv8(dbeta + c, dbm);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dgm[e] = dgm[e] * invM; dbm[e] = dbm[e] * invM; }
    float xv[R][8], d[R][8];
    uint32_t m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i < nvec) {
            const size_t off = (size_t)i * 8;
            ld8(x + off, xv[r]);
            ld8(dy + off, d[r]);
            m[r] = relu == 2 ? rbits[off >> 5] : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i >= nvec) continue;
        const size_t off = (size_t)i * 8;
        if (relu == 2) {
            const uint32_t mm = m[r] >> (off & 31);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[r][e] = ((mm >> e) & 1u) ? d[r][e] : 0.f;
        } else if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d[r][e] = (xv[r][e] * sc[e] + t[e] > 0.f) ? d[r][e] : 0.f;
        }
        if (dres) st8(dres + off, d[r]);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sc[e] * (d[r][e] - dbm[e] - ((xv[r][e] - mu[e]) * is[e]) * dgm[e]);
        st8(dx + off, o);
    }
}
extern "C" int uem_bn_bwd_apply(const float* x, const float* dy, const void* ymask, const float* scale, const float* shift,
                                const float* save_mean, const float* save_invstd, const float* dgamma, const float* dbeta,
                                int M, int C, int relu, float* dx, float* dres, void* stream) {
    UEM_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && dgamma && dbeta && dx, "bn_bwd_apply: null pointer");
    UEM_REQUIRE(M > 0 && C > 0 && (C % 4) == 0, "bn_bwd_apply: bad shape");
    UEM_REQUIRE(relu != UEM_RELU_BITS || (ymask && C % 32 == 0), "bn_bwd_apply: UEM_RELU_BITS needs the bit mask and C %% 32 == 0");
    const int64_t nvec = (int64_t)M * C / 4;
    const int rows = bn_rows(nvec, C, 4);
    if (rows == 4)
        bn_bwd_apply_rows_kernel<float, 4><<<(unsigned)uem_cdiv(nvec, 1024), 256, 0, (hipStream_t)stream>>>(
            x, dy, (const float*)ymask, scale, shift, save_mean, save_invstd, dgamma, dbeta, nvec, C, 1.0f / (float)M, relu, dx, dres);
    else if (rows == 2)
        bn_bwd_apply_rows_kernel<float, 2><<<(unsigned)uem_cdiv(nvec, 512), 256, 0, (hipStream_t)stream>>>(
            x, dy, (const float*)ymask, scale, shift, save_mean, save_invstd, dgamma, dbeta, nvec, C, 1.0f / (float)M, relu, dx, dres);
    else
        bn_bwd_apply_kernel<float><<<uem_flat_grid(nvec, 256), 256, 0, (hipStream_t)stream>>>(
            x, dy, (const float*)ymask, scale, shift, save_mean, save_invstd, dgamma, dbeta, nvec, C, 1.0f / (float)M, relu, dx, dres);
    return uem_check_launch("bn_bwd_apply");
}
extern "C" int uem_bn_bwd_apply_bf16(const uint16_t* x, const uint16_t* dy, const uint32_t* relu_bits, const float* scale,
                                     const float* shift, const float* save_mean, const float* save_invstd, const float* dgamma,
                                     const float* dbeta, int M, int C, int relu, uint16_t* dx, uint16_t* dres, void* stream) {
    UEM_REQUIRE(x && dy && scale && shift && save_mean && save_invstd && dgamma && dbeta && dx, "bn_bwd_apply_bf16: null pointer");
    UEM_REQUIRE(M > 0 && C > 0 && (C % 4) == 0, "bn_bwd_apply_bf16: bad shape");
    UEM_REQUIRE(relu == 0 || (relu == 1 && !relu_bits) || (relu == UEM_RELU_BITS && relu_bits && C % 32 == 0), "bn_bwd_apply_bf16: bad relu mode");
    if (C % 8 == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0) {
        const int64_t nvec8 = (int64_t)M * C / 8;
        const int rows = bn_rows(nvec8, C, 8);
        if (rows > 1) {
            if (rows >= 4)
                bn_bwd_apply_bf16x8_rows_kernel<4><<<(unsigned)uem_cdiv(nvec8, 1024), 256, 0, (hipStream_t)stream>>>(
                    x, dy, relu_bits, scale, shift, save_mean, save_invstd, dgamma, dbeta, nvec8, C, 1.0f / (float)M, relu, dx, dres);
            else
                bn_bwd_apply_bf16x8_rows_kernel<2><<<(unsigned)uem_cdiv(nvec8, 512), 256, 0, (hipStream_t)stream>>>(
                    x, dy, relu_bits, scale, shift, save_mean, save_invstd, dgamma, dbeta, nvec8, C, 1.0f / (float)M, relu, dx, dres);
            return uem_check_launch("bn_bwd_apply_bf16");
        }
        bn_bwd_apply_bf16x8_kernel<<<uem_flat_grid(nvec8, 256), 256, 0, (hipStream_t)stream>>>(
            x, dy, relu_bits, scale, shift, save_mean, save_invstd, dgamma, dbeta, nvec8, C, 1.0f / (float)M, relu, dx, dres);
        return uem_check_launch("bn_bwd_apply_bf16");
    }
    const int64_t nvec = (int64_t)M * C / 4;
    bn_bwd_apply_kernel<bf16_t><<<uem_flat_grid(nvec, 256), 256, 0, (hipStream_t)stream>>>(
        x, dy, (const float*)relu_bits, scale, shift, save_mean, save_invstd, dgamma, dbeta, nvec, C, 1.0f / (float)M, relu, dx, dres);
    return uem_check_launch("bn_bwd_apply_bf16");
}

// ---------------------------------------------------------------------------------------------------------
// Two BatchNorm backward apply passes that share their incoming gradient (round 5): the bottleneck blocks with a downsample branch end in
// y = relu(bn3(z3) + bn_ds(zd)) (_resnets.py:104-112), so bn3's and the downsample BatchNorm's backward both read dy gated by the same
// packed ReLU bits.  One pass reads dy and the bits once, z3 and zd, and writes dz3 and dzd (dzd may overwrite dy): five tensor passes
// instead of six.  Element arithmetic identical to bn_bwd_apply_rows_kernel / bn_bwd_apply_bf16x8_rows_kernel (UEM_RELU_BITS form).
// ---------------------------------------------------------------------------------------------------------
struct BnPairVec { const float *scale, *mean, *invstd, *dgamma, *dbeta; };
template <int R>
__global__ __launch_bounds__(256) void bn_bwd_apply_pair_rows_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                                     const float* dy, const uint32_t* __restrict__ bits, const BnPairVec v1,
                                                                     const BnPairVec v2, int64_t nvec, int C, float invM,
                                                                     float* __restrict__ dx1, float* dx2) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)((i0 * 4) % C);
    auto ld = [&](const float* p) { return *reinterpret_cast<const float4*>(p + c); };
    auto scaled = [&](float4 a) { a.x *= invM; a.y *= invM; a.z *= invM; a.w *= invM; return a; };
    const float4 sc1 = ld(v1.scale), mu1 = ld(v1.mean), is1 = ld(v1.invstd), dg1 = scaled(ld(v1.dgamma)), db1 = scaled(ld(v1.dbeta));
    const float4 sc2 = ld(v2.scale), mu2 = ld(v2.mean), is2 = ld(v2.invstd), dg2 = scaled(ld(v2.dgamma)), db2 = scaled(ld(v2.dbeta));
    float4 a[R], b[R], d[R];
    uint32_t m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i < nvec) {
            const size_t off = (size_t)i * 4;
            a[r] = *reinterpret_cast<const float4*>(x1 + off);
            b[r] = *reinterpret_cast<const float4*>(x2 + off);
            d[r] = *reinterpret_cast<const float4*>(dy + off);
            m[r] = bits[off >> 5];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i >= nvec) continue;
        const size_t off = (size_t)i * 4;
        const uint32_t mm = m[r] >> (off & 31);
        float4 g = d[r];
        g.x = (mm & 1u) ? g.x : 0.f; g.y = (mm & 2u) ? g.y : 0.f; g.z = (mm & 4u) ? g.z : 0.f; g.w = (mm & 8u) ? g.w : 0.f;
        float4 o;
        o.x = sc1.x * (g.x - db1.x - ((a[r].x - mu1.x) * is1.x) * dg1.x);
        o.y = sc1.y * (g.y - db1.y - ((a[r].y - mu1.y) * is1.y) * dg1.y);
        o.z = sc1.z * (g.z - db1.z - ((a[r].z - mu1.z) * is1.z) * dg1.z);
        o.w = sc1.w * (g.w - db1.w - ((a[r].w - mu1.w) * is1.w) * dg1.w);
        *reinterpret_cast<float4*>(dx1 + off) = o;
        o.x = sc2.x * (g.x - db2.x - ((b[r].x - mu2.x) * is2.x) * dg2.x);
        o.y = sc2.y * (g.y - db2.y - ((b[r].y - mu2.y) * is2.y) * dg2.y);
        o.z = sc2.z * (g.z - db2.z - ((b[r].z - mu2.z) * is2.z) * dg2.z);
        o.w = sc2.w * (g.w - db2.w - ((b[r].w - mu2.w) * is2.w) * dg2.w);
        *reinterpret_cast<float4*>(dx2 + off) = o;
    }
}
template <int R>
__global__ __launch_bounds__(256) void bn_bwd_apply_pair_bf16x8_rows_kernel(const bf16_t* __restrict__ x1, const bf16_t* __restrict__ x2,
                                                                            const bf16_t* dy, const uint32_t* __restrict__ bits,
                                                                            const BnPairVec v1, const BnPairVec v2, int64_t nvec, int C,
                                                                            float invM, bf16_t* __restrict__ dx1, bf16_t* dx2) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)((i0 * 8) % C);
    float sc1[8], mu1[8], is1[8], dg1[8], db1[8], sc2[8], mu2[8], is2[8], dg2[8], db2[8];
    ldv8(v1.scale + c, sc1); ldv8(v1.mean + c, mu1); ldv8(v1.invstd + c, is1); ldv8(v1.dgamma + c, dg1); ldv8(v1.dbeta + c, db1);
    ldv8(v2.scale + c, sc2); ldv8(v2.mean + c, mu2); ldv8(v2.invstd + c, is2); ldv8(v2.dgamma + c, dg2); ldv8(v2.dbeta + c, db2);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg1[e] = dg1[e] * invM; db1[e] = db1[e] * invM; dg2[e] = dg2[e] * invM; db2[e] = db2[e] * invM; }
    float a[R][8], b[R][8], d[R][8];
    uint32_t m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i < nvec) {
            const size_t off = (size_t)i * 8;
            ld8(x1 + off, a[r]);
            ld8(x2 + off, b[r]);
            ld8(dy + off, d[r]);
            m[r] = bits[off >> 5];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + r * stride;
        if (i >= nvec) continue;
        const size_t off = (size_t)i * 8;
        const uint32_t mm = m[r] >> (off & 31);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[r][e] = ((mm >> e) & 1u) ? d[r][e] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sc1[e] * (d[r][e] - db1[e] - ((a[r][e] - mu1[e]) * is1[e]) * dg1[e]);
        st8(dx1 + off, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sc2[e] * (d[r][e] - db2[e] - ((b[r][e] - mu2[e]) * is2[e]) * dg2[e]);
        st8(dx2 + off, o);
    }
}
// UEM_ERR_UNSUPPORTED (nothing launched) unless the tensors are large power-of-two-channel maps the rows kernels take (the shapes of the
// encoder's downsample blocks at training batch sizes); the caller then runs the two uem_bn_bwd_apply passes.
extern "C" int uem_bn_bwd_apply_pair(const float* x1, const float* x2, const float* dy, const uint32_t* relu_bits, const float* scale1,
                                     const float* mean1, const float* invstd1, const float* dgamma1, const float* dbeta1,
                                     const float* scale2, const float* mean2, const float* invstd2, const float* dgamma2,
                                     const float* dbeta2, int M, int C, float* dx1, float* dx2, void* stream) {
    UEM_REQUIRE(x1 && x2 && dy && relu_bits && dx1 && dx2 && scale1 && mean1 && invstd1 && dgamma1 && dbeta1 && scale2 && mean2 && invstd2 &&
                dgamma2 && dbeta2, "bn_bwd_apply_pair: null pointer");
    UEM_REQUIRE(M > 0 && C > 0 && C % 32 == 0, "bn_bwd_apply_pair: bad shape");
    const int64_t nvec = (int64_t)M * C / 4;
    if (!bn_pair_rows_ok(nvec, C, 4)) return uem_fail(UEM_ERR_UNSUPPORTED, "bn_bwd_apply_pair: shape not taken by the rows kernels");
    const BnPairVec v1{scale1, mean1, invstd1, dgamma1, dbeta1}, v2{scale2, mean2, invstd2, dgamma2, dbeta2};
    bn_bwd_apply_pair_rows_kernel<2><<<(unsigned)uem_cdiv(nvec, 512), 256, 0, (hipStream_t)stream>>>(x1, x2, dy, relu_bits, v1, v2, nvec, C,
                                                                                                  1.0f / (float)M, dx1, dx2);
    return uem_check_launch("bn_bwd_apply_pair");
}
extern "C" int uem_bn_bwd_apply_pair_bf16(const uint16_t* x1, const uint16_t* x2, const uint16_t* dy, const uint32_t* relu_bits,
                                          const float* scale1, const float* mean1, const float* invstd1, const float* dgamma1,
                                          const float* dbeta1, const float* scale2, const float* mean2, const float* invstd2,
                                          const float* dgamma2, const float* dbeta2, int M, int C, uint16_t* dx1, uint16_t* dx2,
                                          void* stream) {
    UEM_REQUIRE(x1 && x2 && dy && relu_bits && dx1 && dx2 && scale1 && mean1 && invstd1 && dgamma1 && dbeta1 && scale2 && mean2 && invstd2 &&
                dgamma2 && dbeta2, "bn_bwd_apply_pair_bf16: null pointer");
    UEM_REQUIRE(M > 0 && C > 0 && C % 32 == 0, "bn_bwd_apply_pair_bf16: bad shape");
    const int64_t nvec8 = (int64_t)M * C / 8;
    if ((((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dy | (uintptr_t)dx1 | (uintptr_t)dx2) & 15) != 0 || !bn_pair_rows_ok(nvec8, C, 8))
        return uem_fail(UEM_ERR_UNSUPPORTED, "bn_bwd_apply_pair_bf16: shape not taken by the rows kernels");
    const BnPairVec v1{scale1, mean1, invstd1, dgamma1, dbeta1}, v2{scale2, mean2, invstd2, dgamma2, dbeta2};
    bn_bwd_apply_pair_bf16x8_rows_kernel<2><<<(unsigned)uem_cdiv(nvec8, 512), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const bf16_t*>(x1), reinterpret_cast<const bf16_t*>(x2), reinterpret_cast<const bf16_t*>(dy), relu_bits, v1, v2, nvec8,
        C, 1.0f / (float)M, reinterpret_cast<bf16_t*>(dx1), reinterpret_cast<bf16_t*>(dx2));
    return uem_check_launch("bn_bwd_apply_pair_bf16");
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm(+ReLU) backward of the layer in front of a 3x3 / stride 2 / pad 1 max-pool (the stem), reading the POOLED gradient
// (N, Ho, Wo, C) and the pool's argmax taps instead of a materialised (N, H, W, C) gradient.  Work item = one 2x2 block of
// input pixels (2k..2k+1, 2j..2j+1): it lies in exactly the four windows (k..k+1, j..j+1), so four (tap, gradient) loads serve
// four pixels -- pixel (2k, 2j) takes window (k, j) at tap 4; (2k, 2j+1) windows (k, j) / (k, j+1) at taps 5 / 3; (2k+1, 2j)
// windows (k, j) / (k+1, j) at 7 / 1; (2k+1, 2j+1) all four at 8, 6, 2, 0 -- added in the order uem_maxpool3x3s2_bwd adds them,
// so sums and dx equal the two-kernel path bit for bit.  H and W even.
// ---------------------------------------------------------------------------------------------------------
struct Pool2x2 {
    float4 d[4];                                   // gradients of pixels (0,0), (0,1), (1,0), (1,1) of the block
};
__device__ __forceinline__ float4 tap_sel(const uchar4 bi, const float4 g, const unsigned char k) {
    return make_float4(bi.x == k ? g.x : 0.f, bi.y == k ? g.y : 0.f, bi.z == k ? g.z : 0.f, bi.w == k ? g.w : 0.f);
}
__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
template <typename TG>
__device__ __forceinline__ Pool2x2 pool_grad_2x2(const TG* __restrict__ dy, const uint8_t* __restrict__ idx, int n, int k, int j, int c,
                                                 int C, int Ho, int Wo) {
    uchar4 bi[2][2];
    float4 g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool ok = k + a < Ho && j + b < Wo;
            const size_t o = (((size_t)n * Ho + (ok ? k + a : k)) * Wo + (ok ? j + b : j)) * C + c;
            bi[a][b] = *reinterpret_cast<const uchar4*>(idx + o);
            g[a][b] = ld4<TG>(dy + o);
            if (!ok) bi[a][b] = make_uchar4(255, 255, 255, 255);       // no such window: matches no tap
        }
    Pool2x2 r;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    r.d[0] = z; add4(r.d[0], tap_sel(bi[0][0], g[0][0], 4));
    r.d[1] = z; add4(r.d[1], tap_sel(bi[0][0], g[0][0], 5)); add4(r.d[1], tap_sel(bi[0][1], g[0][1], 3));
    r.d[2] = z; add4(r.d[2], tap_sel(bi[0][0], g[0][0], 7)); add4(r.d[2], tap_sel(bi[1][0], g[1][0], 1));
    r.d[3] = z; add4(r.d[3], tap_sel(bi[0][0], g[0][0], 8)); add4(r.d[3], tap_sel(bi[0][1], g[0][1], 6));
    add4(r.d[3], tap_sel(bi[1][0], g[1][0], 2)); add4(r.d[3], tap_sel(bi[1][1], g[1][1], 0));
    return r;
}
// rows of the reduction = 2x2 blocks; the partial sums are accumulated pixel by pixel in the row-major order of the block's rows
// T: storage type of x (the stem's conv output z) and of the pooled gradient -- float, or bf16 under bf16 storage (round 5: the
// stem's 64-channel half-resolution map is the network's largest tensor; in bf16 every pass over it moves half the bytes)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_pool_kernel(const T* __restrict__ x, const T* __restrict__ dyp,
                                                                  const uint8_t* __restrict__ idx, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ smean,
                                                                  const float* __restrict__ sinv, int N, int H, int W, int C, int relu,
                                                                  int blocks_per_chunk, float* __restrict__ ws) {
    const ColMap cm = col_map(C, blockIdx.x);
    const int Hb = H >> 1, Wb = W >> 1, Ho = Hb, Wo = Wb, nb = N * Hb * Wb;          // even H, W: Ho = H/2, Wo = W/2
    const int chunk = blockIdx.y;
    const int b0 = chunk * blocks_per_chunk, b1 = min(nb, b0 + blocks_per_chunk);
    const float4 sc = *reinterpret_cast<const float4*>(scale + cm.c0);
    const float4 sh = *reinterpret_cast<const float4*>(shift + cm.c0);
    const float4 mu = *reinterpret_cast<const float4*>(smean + cm.c0);
    const float4 is = *reinterpret_cast<const float4*>(sinv + cm.c0);
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
    for (int b = b0 + cm.rg; b < b1; b += cm.rpp) {
        const int n = b / (Hb * Wb), rem = b - n * (Hb * Wb);
        const int k = rem / Wb, j = rem - k * Wb;
        const Pool2x2 pg = pool_grad_2x2<T>(dyp, idx, n, k, j, cm.c0, C, Ho, Wo);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t off = (((size_t)n * H + 2 * k + (q >> 1)) * W + 2 * j + (q & 1)) * C + cm.c0;
            const float4 xv = ld4<T>(x + off);
            float4 d = pg.d[q];
            if (relu) d = relu_mask4(d, xv, sc, sh, nullptr, off, 1);
            sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
            sg.x += d.x * ((xv.x - mu.x) * is.x); sg.y += d.y * ((xv.y - mu.y) * is.y);
            sg.z += d.z * ((xv.z - mu.z) * is.z); sg.w += d.w * ((xv.w - mu.w) * is.w);
        }
    }
    __shared__ float sh2[256][8];
    sh2[threadIdx.x][0] = sb.x; sh2[threadIdx.x][1] = sb.y; sh2[threadIdx.x][2] = sb.z; sh2[threadIdx.x][3] = sb.w;
    sh2[threadIdx.x][4] = sg.x; sh2[threadIdx.x][5] = sg.y; sh2[threadIdx.x][6] = sg.z; sh2[threadIdx.x][7] = sg.w;
    __syncthreads();
    if (cm.rg == 0) {
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = sh2[threadIdx.x][q];
        for (int g = 1; g < cm.rpp; ++g) {
            const int t = g * cm.lpr + cm.cv;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += sh2[t][q];
        }
        float* w = ws + (size_t)chunk * 2 * C;      // [chunk][2][C] : dbeta, dgamma
#pragma unroll
        for (int q = 0; q < 4; ++q) { w[cm.c0 + q] = a[q]; w[C + cm.c0 + q] = a[4 + q]; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const T* __restrict__ x, const T* __restrict__ dyp,
                                                                const uint8_t* __restrict__ idx, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ smean,
                                                                const float* __restrict__ sinv, const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, int N, int H, int W, int C, float invM,
                                                                int relu, T* __restrict__ dx) {
    const int Hb = H >> 1, Wb = W >> 1, cv = C >> 2;
    const int64_t total = (int64_t)N * Hb * Wb * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        int64_t t = i / cv;
        const int j = (int)(t % Wb); t /= Wb;
        const int k = (int)(t % Hb);
        const int n = (int)(t / Hb);
        const Pool2x2 pg = pool_grad_2x2<T>(dyp, idx, n, k, j, c, C, Hb, Wb);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(smean + c), is = *reinterpret_cast<const float4*>(sinv + c);
        const float4 dg = *reinterpret_cast<const float4*>(dgamma + c), db = *reinterpret_cast<const float4*>(dbeta + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t off = (((size_t)n * H + 2 * k + (q >> 1)) * W + 2 * j + (q & 1)) * C + c;
            const float4 xv = ld4<T>(x + off);
            float4 d = pg.d[q];
            if (relu) d = relu_mask4(d, xv, sc, sh, nullptr, off, 1);
            float4 o;
            o.x = sc.x * (d.x - db.x * invM - ((xv.x - mu.x) * is.x) * (dg.x * invM));
            o.y = sc.y * (d.y - db.y * invM - ((xv.y - mu.y) * is.y) * (dg.y * invM));
            o.z = sc.z * (d.z - db.z * invM - ((xv.z - mu.z) * is.z) * (dg.z * invM));
            o.w = sc.w * (d.w - db.w * invM - ((xv.w - mu.w) * is.w) * (dg.w * invM));
            st4<T>(dx + off, o);
        }
    }
}
template <typename T>
static int bn_bwd_reduce_pool_impl(const T* x, const T* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                                   const float* save_mean, const float* save_invstd, int N, int H, int W, int C, int relu,
                                   float* dgamma, float* dbeta, float* grad_gamma, float* grad_beta, float* workspace, void* stream) {
    UEM_REQUIRE(x && dy_pool && idx && scale && shift && save_mean && save_invstd && dgamma && dbeta && workspace, "bn_bwd_reduce_pool: null pointer");
    UEM_REQUIRE(N > 0 && H > 1 && W > 1 && col_shape_ok(C) && (int64_t)N * H * W < 2147483647LL, "bn_bwd_reduce_pool: unsupported shape");
    if ((H | W) & 1) return uem_fail(UEM_ERR_UNSUPPORTED, "bn_bwd_reduce_pool: H and W must be even");
    UEM_REQUIRE(relu == 0 || relu == 1, "bn_bwd_reduce_pool: relu is 0 or 1 (mask recomputed from x)");
    hipStream_t st = (hipStream_t)stream;
    const int nb = N * (H / 2) * (W / 2);
    int chunks, bpc;
    col_chunks(nb, C, &chunks, &bpc);                 // never more chunks than uem_bn_workspace_floats(N*H*W, C) provides for
    dim3 grid((unsigned)uem_cdiv(C, 256), (unsigned)chunks);
    bn_bwd_partial_pool_kernel<T><<<grid, 256, 0, st>>>(x, dy_pool, idx, scale, shift, save_mean, save_invstd, N, H, W, C, relu, bpc, workspace);
    bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(workspace, chunks, C, dgamma, dbeta, grad_gamma, grad_beta);
    return uem_check_launch("bn_bwd_reduce_pool");
}
template <typename T>
static int bn_bwd_apply_pool_impl(const T* x, const T* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                                  const float* save_mean, const float* save_invstd, const float* dgamma, const float* dbeta, int N,
                                  int H, int W, int C, int relu, T* dx, void* stream) {
    UEM_REQUIRE(x && dy_pool && idx && scale && shift && save_mean && save_invstd && dgamma && dbeta && dx, "bn_bwd_apply_pool: null pointer");
    UEM_REQUIRE(N > 0 && H > 1 && W > 1 && C > 0 && (C % 4) == 0 && (int64_t)N * H * W < 2147483647LL, "bn_bwd_apply_pool: bad shape");
    if ((H | W) & 1) return uem_fail(UEM_ERR_UNSUPPORTED, "bn_bwd_apply_pool: H and W must be even");
    UEM_REQUIRE(relu == 0 || relu == 1, "bn_bwd_apply_pool: relu is 0 or 1 (mask recomputed from x)");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    bn_bwd_apply_pool_kernel<T><<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(
        x, dy_pool, idx, scale, shift, save_mean, save_invstd, dgamma, dbeta, N, H, W, C, 1.0f / (float)((int64_t)N * H * W), relu, dx);
    return uem_check_launch("bn_bwd_apply_pool");
}
extern "C" int uem_bn_bwd_reduce_pool(const float* x, const float* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                                      const float* save_mean, const float* save_invstd, int N, int H, int W, int C, int relu,
                                      float* dgamma, float* dbeta, float* grad_gamma, float* grad_beta, float* workspace, void* stream) {
    return bn_bwd_reduce_pool_impl<float>(x, dy_pool, idx, scale, shift, save_mean, save_invstd, N, H, W, C, relu, dgamma, dbeta, grad_gamma,
                                          grad_beta, workspace, stream);
}
extern "C" int uem_bn_bwd_apply_pool(const float* x, const float* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                                     const float* save_mean, const float* save_invstd, const float* dgamma, const float* dbeta, int N,
                                     int H, int W, int C, int relu, float* dx, void* stream) {
    return bn_bwd_apply_pool_impl<float>(x, dy_pool, idx, scale, shift, save_mean, save_invstd, dgamma, dbeta, N, H, W, C, relu, dx, stream);
}
extern "C" int uem_bn_bwd_reduce_pool_bf16(const uint16_t* x, const uint16_t* dy_pool, const uint8_t* idx, const float* scale,
                                           const float* shift, const float* save_mean, const float* save_invstd, int N, int H, int W, int C,
                                           int relu, float* dgamma, float* dbeta, float* grad_gamma, float* grad_beta, float* workspace,
                                           void* stream) {
    return bn_bwd_reduce_pool_impl<bf16_t>(x, dy_pool, idx, scale, shift, save_mean, save_invstd, N, H, W, C, relu, dgamma, dbeta, grad_gamma,
                                           grad_beta, workspace, stream);
}
extern "C" int uem_bn_bwd_apply_pool_bf16(const uint16_t* x, const uint16_t* dy_pool, const uint8_t* idx, const float* scale,
                                          const float* shift, const float* save_mean, const float* save_invstd, const float* dgamma,
                                          const float* dbeta, int N, int H, int W, int C, int relu, uint16_t* dx, void* stream) {
    return bn_bwd_apply_pool_impl<bf16_t>(x, dy_pool, idx, scale, shift, save_mean, save_invstd, dgamma, dbeta, N, H, W, C, relu, dx, stream);
}

// ---------------------------------------------------------------------------------------------------------
// fp32 <-> bf16 (round to nearest even): the weight arena's bf16 copy, and the two ends of the bf16-storage region
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t nvec, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) st4<bf16_t>(y + i * 4, ld4<float>(x + i * 4));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const int64_t i = (n & ~(int64_t)3) + threadIdx.x; y[i] = (bf16_t)f2bf_bits(x[i]); }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t nvec, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) st4<float>(y + i * 4, ld4<bf16_t>(x + i * 4));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const int64_t i = (n & ~(int64_t)3) + threadIdx.x; y[i] = __uint_as_float((unsigned)x[i] << 16); }
}
extern "C" int uem_cast_f32_bf16(const float* x, uint16_t* y, int64_t n, void* stream) {
    UEM_REQUIRE(x && y && n > 0 && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 7) == 0), "cast_f32_bf16: bad arguments");
    cast_f32_bf16_kernel<<<uem_flat_grid(n / 4 + 1, 256), 256, 0, (hipStream_t)stream>>>(x, y, n / 4, n);
    return uem_check_launch("cast_f32_bf16");
}
extern "C" int uem_cast_bf16_f32(const uint16_t* x, float* y, int64_t n, void* stream) {
    UEM_REQUIRE(x && y && n > 0 && (((uintptr_t)x & 7) == 0) && (((uintptr_t)y & 15) == 0), "cast_bf16_f32: bad arguments");
    cast_bf16_f32_kernel<<<uem_flat_grid(n / 4 + 1, 256), 256, 0, (hipStream_t)stream>>>(x, y, n / 4, n);
    return uem_check_launch("cast_bf16_f32");
}
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ res, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int64_t nvec, int C, int relu,
                                                             float* __restrict__ dx, float* __restrict__ dres) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const size_t off = (size_t)i * 4;
        float4 d = *reinterpret_cast<const float4*>(dy + off);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        if (relu) d = relu_mask4(d, *reinterpret_cast<const float4*>(x + off), sc, *reinterpret_cast<const float4*>(shift + c), res, off, relu);
        if (dres) *reinterpret_cast<float4*>(dres + off) = d;
        d.x *= sc.x; d.y *= sc.y; d.z *= sc.z; d.w *= sc.w;
        *reinterpret_cast<float4*>(dx + off) = d;
    }
}
extern "C" int uem_affine_act_bwd(const float* x, const float* dy, const float* ymask, const float* scale, const float* shift,
                                  int64_t M, int C, int relu, float* dx, float* dres, void* stream) {
    UEM_REQUIRE(x && dy && scale && shift && dx && M > 0 && C > 0 && (C % 4) == 0, "affine_act_bwd: bad arguments");
    UEM_REQUIRE(relu != UEM_RELU_BITS || (ymask && C % 32 == 0), "affine_act_bwd: UEM_RELU_BITS needs the bit mask and C %% 32 == 0");
    const int64_t nvec = M * C / 4;
    affine_act_bwd_kernel<<<uem_flat_grid(nvec, 256), 256, 0, (hipStream_t)stream>>>(x, dy, ymask, scale, shift, nvec, C, relu, dx, dres);
    return uem_check_launch("affine_act_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// MaxPool 3x3 stride 2 pad 1 (first max in row-major window order wins, like torch CPU); idx in 0..8
// ---------------------------------------------------------------------------------------------------------
template <bool AFFINE, typename T = float>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
    // AFFINE: the pooled tensor is relu(x*scale + shift) (the stem's BatchNorm + ReLU), never written to memory
    const int cv = C >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        int64_t t = i / cv;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
        bool any = false;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float4 v = ld4<T>(x + (((size_t)n * H + iy) * W + ix) * C + c);
                if (AFFINE) {
                    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
                    v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
                    v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
                }
                const unsigned char k = (unsigned char)(ky * 3 + kx);
                if (!any) { best = v; bi = make_uchar4(k, k, k, k); any = true; }
                else {
                    if (v.x > best.x) { best.x = v.x; bi.x = k; }
                    if (v.y > best.y) { best.y = v.y; bi.y = k; }
                    if (v.z > best.z) { best.z = v.z; bi.z = k; }
                    if (v.w > best.w) { best.w = v.w; bi.w = k; }
                }
            }
        }
        const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + c;
        st4<T>(y + o, best);
        if (idx) *reinterpret_cast<uchar4*>(idx + o) = bi;
    }
}
extern "C" int uem_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    UEM_REQUIRE(x && y && N > 0 && H > 1 && W > 1 && C > 0 && (C % 4) == 0, "maxpool_fwd: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    maxpool_fwd_kernel<false><<<uem_flat_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, nullptr, nullptr, y, idx, N, H, W, C, Ho, Wo);
    return uem_check_launch("maxpool_fwd");
}
extern "C" int uem_maxpool3x3s2_affine_fwd(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx, int N, int H,
                                           int W, int C, void* stream) {
    UEM_REQUIRE(x && scale && shift && y && N > 0 && H > 1 && W > 1 && C > 0 && (C % 4) == 0, "maxpool_affine_fwd: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    maxpool_fwd_kernel<true><<<uem_flat_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, scale, shift, y, idx, N, H, W, C, Ho, Wo);
    return uem_check_launch("maxpool_affine_fwd");
}
extern "C" int uem_maxpool3x3s2_affine_fwd_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y, uint8_t* idx, int N,
                                                int H, int W, int C, void* stream) {
    UEM_REQUIRE(x && scale && shift && y && N > 0 && H > 1 && W > 1 && C > 0 && (C % 4) == 0, "maxpool_affine_fwd_bf16: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    maxpool_fwd_kernel<true, bf16_t><<<uem_flat_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, scale, shift, y, idx, N, H, W, C, Ho, Wo);
    return uem_check_launch("maxpool_affine_fwd_bf16");
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    // gather form: input (iy, ix) collects from the <= 2x2 windows that contain it
    const int cv = C >> 2;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        int64_t t = i / cv;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        const float4 acc = pool_grad4(dy, idx, n, iy, ix, c, C, Ho, Wo);
        *reinterpret_cast<float4*>(dx + (((size_t)n * H + iy) * W + ix) * C + c) = acc;
    }
}
extern "C" int uem_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, void* stream) {
    UEM_REQUIRE(dy && idx && dx && N > 0 && H > 1 && W > 1 && C > 0 && (C % 4) == 0, "maxpool_bwd: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * H * W * (C / 4);
    maxpool_bwd_kernel<<<uem_flat_grid(total, 256), 256, 0, (hipStream_t)stream>>>(dy, idx, dx, N, H, W, C, Ho, Wo);
    return uem_check_launch("maxpool_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// InstanceNorm2d (no affine, no running stats): block = (image n, 64-channel tile), 4 pixel groups
// ---------------------------------------------------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const TX* __restrict__ x, float* __restrict__ y,
                                                           float* __restrict__ smean, float* __restrict__ sinv, int HW, int C,
                                                           float eps) {
    // 16 lanes x float4 = the block's 64 channels, 16 pixel groups: 16-byte accesses, 1 KiB per wave instruction (the scalar
    // version ran at 2.1 TB/s).  One pass of shifted sums per thread (K = its first pixel), Chan merge across the 16 groups.
    const int n = blockIdx.y, l = threadIdx.x & 15, g = threadIdx.x >> 4, c = blockIdx.x * 64 + l * 4;
    const TX* xb = x + (size_t)n * HW * C + c;
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
    float cnt = 0.f;
    for (int p = g; p < HW; p += 16) {
        const float4 v = ld4<TX>(xb + (size_t)p * C);
        if (cnt == 0.f) K = v;
        const float4 d = make_float4(v.x - K.x, v.y - K.y, v.z - K.z, v.w - K.w);
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * d.x; s2.y += d.y * d.y; s2.z += d.z * d.z; s2.w += d.w * d.w;
        cnt += 1.f;
    }
    __shared__ float sh[16][64][3];
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    const float k4[4] = {K.x, K.y, K.z, K.w}, a4[4] = {s1.x, s1.y, s1.z, s1.w}, q4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[g][l * 4 + e][0] = cnt;
        sh[g][l * 4 + e][1] = k4[e] + a4[e] * inv;
        sh[g][l * 4 + e][2] = q4[e] - a4[e] * a4[e] * inv;
    }
    __syncthreads();
    float mean4[4], is4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = l * 4 + e;
        float nn = sh[0][ch][0], mean = sh[0][ch][1], m2 = sh[0][ch][2];
        for (int j = 1; j < 16; ++j) chan_merge(nn, mean, m2, sh[j][ch][0], sh[j][ch][1], sh[j][ch][2]);
        mean4[e] = mean;
        is4[e] = 1.0f / sqrtf(m2 / (float)HW + eps);
    }
    if (g == 0) {
        *reinterpret_cast<float4*>(smean + n * C + c) = make_float4(mean4[0], mean4[1], mean4[2], mean4[3]);
        *reinterpret_cast<float4*>(sinv + n * C + c) = make_float4(is4[0], is4[1], is4[2], is4[3]);
    }
    float* yb = y + (size_t)n * HW * C + c;
    for (int p = g; p < HW; p += 16) {
        const float4 v = ld4<TX>(xb + (size_t)p * C);
        *reinterpret_cast<float4*>(yb + (size_t)p * C) = make_float4((v.x - mean4[0]) * is4[0], (v.y - mean4[1]) * is4[1],
                                                                     (v.z - mean4[2]) * is4[2], (v.w - mean4[3]) * is4[3]);
    }
}
// Round 5: one pass over HBM.  A plane of InstanceNorm (one image, HW pixels) is small at the encoder's output stride (32 x 32 = 1024
// pixels of a 512 x 512 tile): a block of 512 threads takes 32 channels -- 8 lanes x float4 per pixel, 64 pixel groups -- and every
// thread KEEPS its HW / 64 pixels in registers between the statistics and the normalisation (forward: x, 16 float4; backward: y and
// dy, 32 float4), so each tensor is read once instead of twice (the second read missed: 256 KB per block against 128 KB of L2 per CU).
// All loads of a thread are independent and issued up front.  PPT = pixels per thread: 4 (HW <= 256) or 16 (HW <= 1024); larger
// planes (1024 x 1024 tiles: 4096 pixels) keep the two-pass kernels above.  Block order: xcd_order below keeps the 32-channel
// halves of a 128-byte line on one XCD's L2.
template <typename TX, int PPT>
__global__ __launch_bounds__(512) void instnorm_fwd_regs_kernel(const TX* __restrict__ x, float* __restrict__ y, float* __restrict__ smean,
                                                                float* __restrict__ sinv, int HW, int C, float eps, int ctiles) {
    const int q = blockIdx.x >> 3, xcd = blockIdx.x & 7, per = gridDim.x >> 3;                  // gridDim.x is a multiple of 8
    const int work = xcd * per + q;                                                             // consecutive work items share an XCD
    const int n = work / ctiles, ct = work - n * ctiles;
    const int l = threadIdx.x & 7, g = threadIdx.x >> 3, c = ct * 32 + l * 4;
    const TX* const xb = x + (size_t)n * HW * C + c;
    float4 v[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = g + 64 * i;
        v[i] = p < HW ? ld4<TX>(xb + (size_t)p * C) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // shifted sums around the thread's first pixel (every group has one: HW >= 64 is checked by the launcher)
    const float4 K = v[0];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float cnt = 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        if (g + 64 * i < HW) {
            const float4 d = make_float4(v[i].x - K.x, v[i].y - K.y, v[i].z - K.z, v[i].w - K.w);
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = fmaf(d.x, d.x, s2.x); s2.y = fmaf(d.y, d.y, s2.y); s2.z = fmaf(d.z, d.z, s2.z); s2.w = fmaf(d.w, d.w, s2.w);
            cnt += 1.f;
        }
    }
    __shared__ float sh[64][32][3];
    const float inv = 1.f / cnt;
    const float k4[4] = {K.x, K.y, K.z, K.w}, a4[4] = {s1.x, s1.y, s1.z, s1.w}, q4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[g][l * 4 + e][0] = cnt;
        sh[g][l * 4 + e][1] = k4[e] + a4[e] * inv;
        sh[g][l * 4 + e][2] = q4[e] - a4[e] * a4[e] * inv;
    }
    __syncthreads();
    __shared__ float fin[2][32];
    if (threadIdx.x < 32) {
        const int ch = threadIdx.x;
        float nn = sh[0][ch][0], mean = sh[0][ch][1], m2 = sh[0][ch][2];
        for (int j = 1; j < 64; ++j) chan_merge(nn, mean, m2, sh[j][ch][0], sh[j][ch][1], sh[j][ch][2]);
        const float is = 1.0f / sqrtf(m2 / (float)HW + eps);
        fin[0][ch] = mean; fin[1][ch] = is;
        smean[n * C + ct * 32 + ch] = mean;
        sinv[n * C + ct * 32 + ch] = is;
    }
    __syncthreads();
    const float4 mu = *reinterpret_cast<const float4*>(&fin[0][l * 4]), is = *reinterpret_cast<const float4*>(&fin[1][l * 4]);
    float* const yb = y + (size_t)n * HW * C + c;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = g + 64 * i;
        if (p < HW)
            *reinterpret_cast<float4*>(yb + (size_t)p * C) = make_float4((v[i].x - mu.x) * is.x, (v[i].y - mu.y) * is.y,
                                                                         (v[i].z - mu.z) * is.z, (v[i].w - mu.w) * is.w);
    }
}
static const int g_instnorm_regs = getenv("UEM_INSTNORM_REGS") ? atoi(getenv("UEM_INSTNORM_REGS")) : 1;
// the register-resident kernels take planes of 64 ... 1024 pixels, 32-channel tiles and a block count that divides over the 8 XCDs
static int instnorm_regs_ppt(int N, int HW, int C) {
    if (!g_instnorm_regs || HW < 64 || HW > 1024 || C % 32 != 0 || ((int64_t)N * (C / 32)) % 8 != 0) return 0;
    return HW <= 256 ? 4 : 16;
}
template <typename TX>
static int instnorm_fwd_launch(const TX* x, float* y, float* save_mean, float* save_invstd, int N, int HW, int C, float eps, hipStream_t st) {
    // (a bf16 input's 32-channel rows are 64-byte runs: measured 129 us against the two-pass kernel's 123 at 32 x 1024 x 2048 -- fp32
    // input 141 against 164, backward 205 / 179 against 257 / 232)
    const int ppt = sizeof(TX) == 4 ? instnorm_regs_ppt(N, HW, C) : 0;
    const int ctiles = C / 32;
    if (ppt == 4) instnorm_fwd_regs_kernel<TX, 4><<<N * ctiles, 512, 0, st>>>(x, y, save_mean, save_invstd, HW, C, eps, ctiles);
    else if (ppt == 16) instnorm_fwd_regs_kernel<TX, 16><<<N * ctiles, 512, 0, st>>>(x, y, save_mean, save_invstd, HW, C, eps, ctiles);
    else instnorm_fwd_kernel<TX><<<dim3(C / 64, N), 256, 0, st>>>(x, y, save_mean, save_invstd, HW, C, eps);
    return uem_check_launch("instnorm_fwd");
}
extern "C" int uem_instnorm_fwd(const float* x, float* y, float* save_mean, float* save_invstd, int N, int HW, int C,
                                float eps, void* stream) {
    UEM_REQUIRE(x && y && save_mean && save_invstd && N > 0 && HW > 0 && C > 0 && (C % 64) == 0, "instnorm_fwd: bad arguments (C %% 64)");
    return instnorm_fwd_launch<float>(x, y, save_mean, save_invstd, N, HW, C, eps, (hipStream_t)stream);
}
// bf16 storage: the InstanceNorm at the end of the bf16 region reads the bf16 layer4 output directly and writes the fp32 feature map the
// heads and the mining read (no separate cast pass), and its backward writes the bf16 gradient directly
extern "C" int uem_instnorm_fwd_bf16(const uint16_t* x, float* y, float* save_mean, float* save_invstd, int N, int HW, int C,
                                     float eps, void* stream) {
    UEM_REQUIRE(x && y && save_mean && save_invstd && N > 0 && HW > 0 && C > 0 && (C % 64) == 0, "instnorm_fwd_bf16: bad arguments (C %% 64)");
    return instnorm_fwd_launch<bf16_t>(x, y, save_mean, save_invstd, N, HW, C, eps, (hipStream_t)stream);
}
template <typename TO>
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                           const float* __restrict__ sinv, TO* __restrict__ dx, int HW, int C) {
    const int n = blockIdx.y, l = threadIdx.x & 15, g = threadIdx.x >> 4, c = blockIdx.x * 64 + l * 4;
    const size_t base = (size_t)n * HW * C + c;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (int p = g; p < HW; p += 16) {
        const float4 d = *reinterpret_cast<const float4*>(dy + base + (size_t)p * C);
        const float4 v = *reinterpret_cast<const float4*>(y + base + (size_t)p * C);
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * v.x; s2.y += d.y * v.y; s2.z += d.z * v.z; s2.w += d.w * v.w;
    }
    __shared__ float sh[16][64][2];
    const float a4[4] = {s1.x, s1.y, s1.z, s1.w}, b4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[g][l * 4 + e][0] = a4[e]; sh[g][l * 4 + e][1] = b4[e]; }
    __syncthreads();
    float m1[4], m2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = l * 4 + e;
        float a = 0.f, b = 0.f;
        for (int j = 0; j < 16; ++j) { a += sh[j][ch][0]; b += sh[j][ch][1]; }
        m1[e] = a / (float)HW;
        m2[e] = b / (float)HW;
    }
    const float4 is = *reinterpret_cast<const float4*>(sinv + n * C + c);
    for (int p = g; p < HW; p += 16) {
        const size_t o = base + (size_t)p * C;
        const float4 d = *reinterpret_cast<const float4*>(dy + o);
        const float4 v = *reinterpret_cast<const float4*>(y + o);
        st4<TO>(dx + o, make_float4(is.x * (d.x - m1[0] - v.x * m2[0]), is.y * (d.y - m1[1] - v.y * m2[1]),
                                    is.z * (d.z - m1[2] - v.z * m2[2]), is.w * (d.w - m1[3] - v.w * m2[3])));
    }
}
template <typename TO, int PPT>
__global__ __launch_bounds__(512) void instnorm_bwd_regs_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                const float* __restrict__ sinv, TO* __restrict__ dx, int HW, int C, int ctiles) {
    const int q = blockIdx.x >> 3, xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int work = xcd * per + q;
    const int n = work / ctiles, ct = work - n * ctiles;
    const int l = threadIdx.x & 7, g = threadIdx.x >> 3, c = ct * 32 + l * 4;
    const size_t base = (size_t)n * HW * C + c;
    float4 v[PPT], d[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = g + 64 * i;
        const bool ok = p < HW;
        d[i] = ok ? *reinterpret_cast<const float4*>(dy + base + (size_t)p * C) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[i] = ok ? *reinterpret_cast<const float4*>(y + base + (size_t)p * C) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        s1.x += d[i].x; s1.y += d[i].y; s1.z += d[i].z; s1.w += d[i].w;
        s2.x = fmaf(d[i].x, v[i].x, s2.x); s2.y = fmaf(d[i].y, v[i].y, s2.y); s2.z = fmaf(d[i].z, v[i].z, s2.z); s2.w = fmaf(d[i].w, v[i].w, s2.w);
    }
    __shared__ float sh[64][32][2];
    const float a4[4] = {s1.x, s1.y, s1.z, s1.w}, b4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[g][l * 4 + e][0] = a4[e]; sh[g][l * 4 + e][1] = b4[e]; }
    __syncthreads();
    __shared__ float fin[2][32];
    if (threadIdx.x < 64) {
        const int ch = threadIdx.x & 31, which = threadIdx.x >> 5;
        float a = 0.f;
        for (int j = 0; j < 64; ++j) a += sh[j][ch][which];
        fin[which][ch] = a / (float)HW;
    }
    __syncthreads();
    const float4 m1 = *reinterpret_cast<const float4*>(&fin[0][l * 4]), m2 = *reinterpret_cast<const float4*>(&fin[1][l * 4]);
    const float4 is = *reinterpret_cast<const float4*>(sinv + n * C + c);
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = g + 64 * i;
        if (p < HW)
            st4<TO>(dx + base + (size_t)p * C, make_float4(is.x * (d[i].x - m1.x - v[i].x * m2.x), is.y * (d[i].y - m1.y - v[i].y * m2.y),
                                                           is.z * (d[i].z - m1.z - v[i].z * m2.z), is.w * (d[i].w - m1.w - v[i].w * m2.w)));
    }
}
template <typename TO>
static int instnorm_bwd_launch(const float* y, const float* dy, const float* save_invstd, TO* dx, int N, int HW, int C, hipStream_t st) {
    const int ppt = instnorm_regs_ppt(N, HW, C);
    const int ctiles = C / 32;
    if (ppt == 4) instnorm_bwd_regs_kernel<TO, 4><<<N * ctiles, 512, 0, st>>>(y, dy, save_invstd, dx, HW, C, ctiles);
    else if (ppt == 16) instnorm_bwd_regs_kernel<TO, 16><<<N * ctiles, 512, 0, st>>>(y, dy, save_invstd, dx, HW, C, ctiles);
    else instnorm_bwd_kernel<TO><<<dim3(C / 64, N), 256, 0, st>>>(y, dy, save_invstd, dx, HW, C);
    return uem_check_launch("instnorm_bwd");
}
extern "C" int uem_instnorm_bwd(const float* y, const float* dy, const float* save_invstd, float* dx, int N, int HW, int C,
                                void* stream) {
    UEM_REQUIRE(y && dy && save_invstd && dx && N > 0 && HW > 0 && (C % 64) == 0, "instnorm_bwd: bad arguments");
    return instnorm_bwd_launch<float>(y, dy, save_invstd, dx, N, HW, C, (hipStream_t)stream);
}
extern "C" int uem_instnorm_bwd_bf16(const float* y, const float* dy, const float* save_invstd, uint16_t* dx, int N, int HW, int C,
                                     void* stream) {
    UEM_REQUIRE(y && dy && save_invstd && dx && N > 0 && HW > 0 && (C % 64) == 0, "instnorm_bwd_bf16: bad arguments");
    return instnorm_bwd_launch<bf16_t>(y, dy, save_invstd, dx, N, HW, C, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
// PPM pieces
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bin_lo(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int bin_hi(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }
__global__ void adaptive_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int S) {
    const int64_t total = (int64_t)N * S * S * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % S); t /= S;
        const int oy = (int)(t % S);
        const int n = (int)(t / S);
        const int y0 = bin_lo(oy, H, S), y1 = bin_hi(oy, H, S), x0 = bin_lo(ox, W, S), x1 = bin_hi(ox, W, S);
        float s = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) s += x[(((size_t)n * H + yy) * W + xx) * C + c];
        y[i] = s / (float)((y1 - y0) * (x1 - x0));
    }
}
extern "C" int uem_adaptive_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int S, void* stream) {
    UEM_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && S > 0 && S <= H && S <= W, "adaptive_avgpool_fwd: bad arguments");
    const int64_t total = (int64_t)N * S * S * C;
    adaptive_avgpool_fwd_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, y, N, H, W, C, S);
    return uem_check_launch("adaptive_avgpool_fwd");
}
__global__ void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C, int S) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        float s = 0.f;
        for (int oy = 0; oy < S; ++oy) {
            const int y0 = bin_lo(oy, H, S), y1 = bin_hi(oy, H, S);
            if (yy < y0 || yy >= y1) continue;
            for (int ox = 0; ox < S; ++ox) {
                const int x0 = bin_lo(ox, W, S), x1 = bin_hi(ox, W, S);
                if (xx < x0 || xx >= x1) continue;
                s += dy[(((size_t)n * S + oy) * S + ox) * C + c] / (float)((y1 - y0) * (x1 - x0));
            }
        }
        dx[i] += s;
    }
}
extern "C" int uem_adaptive_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int S, void* stream) {
    UEM_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && S > 0, "adaptive_avgpool_bwd: bad arguments");
    const int64_t total = (int64_t)N * H * W * C;
    adaptive_avgpool_bwd_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(dy, dx, N, H, W, C, S);
    return uem_check_launch("adaptive_avgpool_bwd");
}

// The feature gradient of a PPM head in ONE pass (Encoder.py:34-41 backward): dfeat = dcat[..., :C] (the concat's first C
// channels, row stride dcat_ld) + sum over the pooled branches of the adaptive-average-pool backward of dp_i -- instead of a
// slice copy plus one read-modify-write pass over dfeat per branch (5 passes over a 268 MB tensor per head).  Bins of scales
// that do not divide the map overlap, so a pixel collects from up to 2 x 2 bins per branch; the terms are added in the order
// of the per-branch kernel (branch by branch, bin rows outermost).
struct PpmGradP {
    const float* dp[4];
    int S[4];
    int nb;
};
__global__ __launch_bounds__(256) void ppm_feat_grad_kernel(const float* __restrict__ dcat, int dcat_ld, const PpmGradP g,
                                                            float* __restrict__ dfeat, int N, int H, int W, int C) {
    // per block: for every branch and every row / column of the map, the first bin that contains it, how many do (1 or 2:
    // bins overlap when the scale does not divide the map) and every bin's extent -- the per-element loop is then table
    // look-ups and at most 2 x 2 sixteen-byte loads per branch (H, W <= 128, scales <= 16: checked by the launcher)
    __shared__ unsigned char first_y[4][128], cnt_y[4][128], first_x[4][128], cnt_x[4][128];
    __shared__ float ext_y[4][16], ext_x[4][16];
    for (int t = threadIdx.x; t < 4 * 128; t += 256) {
        const int b = t >> 7, p = t & 127;
        if (b >= g.nb) continue;
        const int S = g.S[b];
        if (p < H) {
            int f = -1, k = 0;
            for (int o = 0; o < S; ++o) if (p >= bin_lo(o, H, S) && p < bin_hi(o, H, S)) { if (f < 0) f = o; ++k; }
            first_y[b][p] = (unsigned char)f; cnt_y[b][p] = (unsigned char)k;
        }
        if (p < W) {
            int f = -1, k = 0;
            for (int o = 0; o < S; ++o) if (p >= bin_lo(o, W, S) && p < bin_hi(o, W, S)) { if (f < 0) f = o; ++k; }
            first_x[b][p] = (unsigned char)f; cnt_x[b][p] = (unsigned char)k;
        }
        if (p < S) { ext_y[b][p] = (float)(bin_hi(p, H, S) - bin_lo(p, H, S)); ext_x[b][p] = (float)(bin_hi(p, W, S) - bin_lo(p, W, S)); }
    }
    __syncthreads();
    const int cv = C >> 2;
    const int64_t total = (int64_t)N * H * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        int64_t t = i / cv;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        float4 v = *reinterpret_cast<const float4*>(dcat + (((size_t)n * H + yy) * W + xx) * dcat_ld + c);
        for (int b = 0; b < g.nb; ++b) {
            const int S = g.S[b];
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int oy0 = first_y[b][yy], ny = cnt_y[b][yy], ox0 = first_x[b][xx], nx = cnt_x[b][xx];
            for (int oy = oy0; oy < oy0 + ny; ++oy)
                for (int ox = ox0; ox < ox0 + nx; ++ox) {
                    const float area = ext_y[b][oy] * ext_x[b][ox];         // small integers: exact, = (y1 - y0) * (x1 - x0)
                    const float4 d = *reinterpret_cast<const float4*>(g.dp[b] + (((size_t)n * S + oy) * S + ox) * C + c);
                    sacc.x += d.x / area; sacc.y += d.y / area; sacc.z += d.z / area; sacc.w += d.w / area;
                }
            v.x += sacc.x; v.y += sacc.y; v.z += sacc.z; v.w += sacc.w;
        }
        *reinterpret_cast<float4*>(dfeat + (((size_t)n * H + yy) * W + xx) * C + c) = v;
    }
}
extern "C" int uem_ppm_feat_grad(const float* dcat, int dcat_ld, const float* const* dp, const int* scales, int nbranch, float* dfeat,
                                 int N, int H, int W, int C, void* stream) {
    UEM_REQUIRE(dcat && dp && scales && dfeat && nbranch >= 0 && nbranch <= 4 && N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 &&
                    dcat_ld >= C && (dcat_ld % 4) == 0, "ppm_feat_grad: bad arguments");
    PpmGradP g;
    g.nb = nbranch;
    for (int i = 0; i < 4; ++i) { g.dp[i] = i < nbranch ? dp[i] : nullptr; g.S[i] = i < nbranch ? scales[i] : 1; }
    for (int i = 0; i < nbranch; ++i) UEM_REQUIRE(g.dp[i] && g.S[i] > 0 && g.S[i] <= H && g.S[i] <= W && g.S[i] <= 16, "ppm_feat_grad: bad branch %d", i);
    if (H > 128 || W > 128) return uem_fail(UEM_ERR_UNSUPPORTED, "ppm_feat_grad: feature maps up to 128 x 128");
    const int64_t total = (int64_t)N * H * W * (C / 4);
    ppm_feat_grad_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(dcat, dcat_ld, g, dfeat, N, H, W, C);
    return uem_check_launch("ppm_feat_grad");
}

__global__ void bilinear_up_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int h, int w, int C, int H,
                                       int W, int y_ld, int align, const float* __restrict__ scale,
                                       const float* __restrict__ shift, int relu) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int X = (int)(t % W); t /= W;
        const int Y = (int)(t % H);
        const int n = (int)(t / H);
        const Lerp ly = lerp_setup(Y, h, H, align != 0), lx = lerp_setup(X, w, W, align != 0);
        const float* xb = x + (size_t)n * h * w * C + c;
        float v00 = xb[((size_t)ly.i0 * w + lx.i0) * C], v01 = xb[((size_t)ly.i0 * w + lx.i1) * C];
        float v10 = xb[((size_t)ly.i1 * w + lx.i0) * C], v11 = xb[((size_t)ly.i1 * w + lx.i1) * C];
        if (scale) {
            const float sc = scale[c], sh = shift[c];
            v00 = v00 * sc + sh; v01 = v01 * sc + sh; v10 = v10 * sc + sh; v11 = v11 * sc + sh;
        }
        if (relu) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
        y[(((size_t)n * H + Y) * W + X) * y_ld + c] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    }
}
extern "C" int uem_bilinear_up_fwd(const float* x, float* y, int N, int h, int w, int C, int H, int W, int y_ld,
                                   int align_corners, const float* scale, const float* shift, int relu, void* stream) {
    UEM_REQUIRE(x && y && N > 0 && h > 0 && w > 0 && C > 0 && H > 0 && W > 0 && y_ld >= C, "bilinear_up_fwd: bad arguments");
    UEM_REQUIRE((scale == nullptr) == (shift == nullptr), "bilinear_up_fwd: scale/shift come in pairs");
    const int64_t total = (int64_t)N * H * W * C;
    bilinear_up_fwd_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, y, N, h, w, C, H, W, y_ld, align_corners, scale, shift, relu);
    return uem_check_launch("bilinear_up_fwd");
}
__global__ void bilinear_up_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int h, int w, int C, int H,
                                       int W, int dy_ld, int align) {
    // gather: each low-res cell scans all destination pixels whose lerp touches it (h, w are tiny: PPM bins)
    const int64_t total = (int64_t)N * h * w * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int cx = (int)(t % w); t /= w;
        const int cy = (int)(t % h);
        const int n = (int)(t / h);
        float s = 0.f;
        for (int Y = 0; Y < H; ++Y) {
            const Lerp ly = lerp_setup(Y, h, H, align != 0);
            const float wy = (ly.i0 == cy ? ly.l0 : 0.f) + (ly.i1 == cy ? ly.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int X = 0; X < W; ++X) {
                const Lerp lx = lerp_setup(X, w, W, align != 0);
                const float wx = (lx.i0 == cx ? lx.l0 : 0.f) + (lx.i1 == cx ? lx.l1 : 0.f);
                if (wx == 0.f) continue;
                s += wy * wx * dy[(((size_t)n * H + Y) * W + X) * dy_ld + c];
            }
        }
        dx[i] = s;
    }
}
// block = (image, low-res cell, 64 channels): the cell's row / column weights over the destination map go to LDS once (the
// gather above recomputed both lerps -- two divisions each -- for every one of the H x W pixels, per element), then 16 pixel
// groups x 16 lanes x float4 walk the pixels of the cell's support with 256-byte accesses and combine through LDS
__global__ __launch_bounds__(256) void bilinear_up_bwd_tiled_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int h,
                                                                    int w, int C, int H, int W, int dy_ld, int align) {
    __shared__ float wy[128], wx[128];
    __shared__ int ylist[128], xlist[128], ny, nx;
    __shared__ float red[16][64];
    const int cgroups = C >> 6;
    int b = blockIdx.x;
    const int cg = b % cgroups; b /= cgroups;
    const int cx = b % w; b /= w;
    const int cy = b % h;
    const int n = b / h;
    const int tid = threadIdx.x, l = tid & 15, g = tid >> 4, c = cg * 64 + l * 4;
    if (tid < H) {
        const Lerp ly = lerp_setup(tid, h, H, align != 0);
        wy[tid] = (ly.i0 == cy ? ly.l0 : 0.f) + (ly.i1 == cy ? ly.l1 : 0.f);
    }
    if (tid >= 128 && tid - 128 < W) {
        const Lerp lx = lerp_setup(tid - 128, w, W, align != 0);
        wx[tid - 128] = (lx.i0 == cx ? lx.l0 : 0.f) + (lx.i1 == cx ? lx.l1 : 0.f);
    }
    __syncthreads();
    if (tid == 0) { int k = 0; for (int Y = 0; Y < H; ++Y) if (wy[Y] != 0.f) ylist[k++] = Y; ny = k; }
    if (tid == 64) { int k = 0; for (int X = 0; X < W; ++X) if (wx[X] != 0.f) xlist[k++] = X; nx = k; }
    __syncthreads();
    const int npix = ny * nx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = g; q < npix; q += 16) {
        const int Y = ylist[q / nx], X = xlist[q % nx];
        const float wgt = wy[Y] * wx[X];
        const float4 d = *reinterpret_cast<const float4*>(dy + (((size_t)n * H + Y) * W + X) * dy_ld + c);
        acc.x += wgt * d.x; acc.y += wgt * d.y; acc.z += wgt * d.z; acc.w += wgt * d.w;
    }
    red[g][l * 4 + 0] = acc.x; red[g][l * 4 + 1] = acc.y; red[g][l * 4 + 2] = acc.z; red[g][l * 4 + 3] = acc.w;
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) a += red[j][tid];
        dx[(((size_t)n * h + cy) * w + cx) * C + cg * 64 + tid] = a;
    }
}
extern "C" int uem_bilinear_up_bwd(const float* dy, float* dx, int N, int h, int w, int C, int H, int W, int dy_ld,
                                   int align_corners, void* stream) {
    UEM_REQUIRE(dy && dx && N > 0 && h > 0 && w > 0 && C > 0 && H > 0 && W > 0 && dy_ld >= C, "bilinear_up_bwd: bad arguments");
    if (C % 64 == 0 && H <= 128 && W <= 128 && dy_ld % 4 == 0 && (((uintptr_t)dy) & 15) == 0 && (int64_t)N * h * w * (C / 64) < 2147483647LL) {
        const int blocks = N * h * w * (C / 64);
        bilinear_up_bwd_tiled_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dy, dx, N, h, w, C, H, W, dy_ld, align_corners);
        return uem_check_launch("bilinear_up_bwd");
    }
    const int64_t total = (int64_t)N * h * w * C;
    bilinear_up_bwd_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(dy, dx, N, h, w, C, H, W, dy_ld, align_corners);
    return uem_check_launch("bilinear_up_bwd");
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ void dropout2d_mask_kernel(float* __restrict__ mask, int NC, float p, uint64_t seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    const float u = (float)(splitmix64(seed ^ ((uint64_t)i * 0x2545F4914F6CDD1Dull)) >> 40) * (1.0f / 16777216.0f);
    mask[i] = (u >= p) ? 1.0f / (1.0f - p) : 0.f;
}
__global__ void dropout2d_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mask,
                                       int64_t total, int HW, int C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((int64_t)HW * C));
        y[i] = x[i] * mask[n * C + c];
    }
}
extern "C" int uem_dropout2d(const float* x, float* y, float* mask, int N, int HW, int C, float p, uint64_t seed, void* stream) {
    UEM_REQUIRE(x && y && mask && N > 0 && HW > 0 && C > 0 && p >= 0.f && p < 1.f, "dropout2d: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (seed != 0) dropout2d_mask_kernel<<<(int)uem_cdiv(N * C, 256), 256, 0, st>>>(mask, N * C, p, seed);   // seed 0: reuse mask (backward)
    const int64_t total = (int64_t)N * HW * C;
    dropout2d_apply_kernel<<<uem_stream_grid(total, 256), 256, 0, st>>>(x, y, mask, total, HW, C);
    return uem_check_launch("dropout2d");
}

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] += b[i];
}
extern "C" int uem_add_inplace(float* a, const float* b, int64_t n, void* stream) {
    UEM_REQUIRE(a && b && n > 0, "add_inplace: bad arguments");
    add_inplace_kernel<<<uem_flat_grid(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n);
    return uem_check_launch("add_inplace");
}

// a += b; b = 0 in one pass (the fold of the shadow gradient arena the step's second graph accumulates into: 4 n bytes read + written
// per operand instead of the add's 12 n + the memset's 4 n).  16-byte accesses where both are aligned and n allows.
__global__ __launch_bounds__(256) void add_clear_kernel(float* __restrict__ a, float* __restrict__ b, int64_t n4, int64_t n) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4* a4 = reinterpret_cast<float4*>(a);
    float4* b4 = reinterpret_cast<float4*>(b);
    for (int64_t i = i0; i < n4; i += step) {
        float4 x = a4[i];
        const float4 y = b4[i];
        x.x += y.x, x.y += y.y, x.z += y.z, x.w += y.w;
        a4[i] = x;
        b4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t i = n4 * 4 + i0; i < n; i += step) {
        a[i] += b[i];
        b[i] = 0.f;
    }
}
extern "C" int uem_add_clear(float* a, float* b, int64_t n, void* stream) {
    UEM_REQUIRE(a && b && n > 0, "add_clear: bad arguments");
    const bool al = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    const int64_t n4 = al ? n / 4 : 0;
    add_clear_kernel<<<uem_flat_grid(al ? (n + 3) / 4 : n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n4, n);
    return uem_check_launch("add_clear");
}

// ---------------------------------------------------------------------------------------------------------
// layout transforms at the API edge (LDS-tiled transposes)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int Cc) {
    // per image: x[R][Cc] -> y[Cc][R]; 32x32 tiles, block (32, 8)
    __shared__ float tile[32][33];
    const size_t img = (size_t)blockIdx.z * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < Cc) tile[j][tx] = x[img + (size_t)(r0 + j) * Cc + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < Cc && r0 + tx < R) y[img + (size_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}
extern "C" int uem_nhwc_to_nchw(const float* x, float* y, int N, int HW, int C, void* stream) {
    UEM_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "nhwc_to_nchw: bad arguments");
    transpose_kernel<<<dim3((unsigned)uem_cdiv(C, 32), (unsigned)uem_cdiv(HW, 32), (unsigned)N), 256, 0, (hipStream_t)stream>>>(x, y, HW, C);
    return uem_check_launch("nhwc_to_nchw");
}
extern "C" int uem_nchw_to_nhwc(const float* x, float* y, int N, int HW, int C, void* stream) {
    UEM_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "nchw_to_nhwc: bad arguments");
    transpose_kernel<<<dim3((unsigned)uem_cdiv(HW, 32), (unsigned)uem_cdiv(C, 32), (unsigned)N), 256, 0, (hipStream_t)stream>>>(x, y, C, HW);
    return uem_check_launch("nchw_to_nhwc");
}
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ x4, int64_t HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / HW, p = i % HW;
        const float* xb = x + n * 3 * HW + p;
        reinterpret_cast<float4*>(x4)[i] = make_float4(xb[0], xb[HW], xb[2 * HW], 0.f);
    }
}
extern "C" int uem_nchw3_to_nhwc4(const float* x, float* x4, int N, int H, int W, void* stream) {
    UEM_REQUIRE(x && x4 && N > 0 && H > 0 && W > 0, "nchw3_to_nhwc4: bad arguments");
    const int64_t HW = (int64_t)H * W, total = HW * N;
    nchw3_to_nhwc4_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(x, x4, HW, total);
    return uem_check_launch("nchw3_to_nhwc4");
}
