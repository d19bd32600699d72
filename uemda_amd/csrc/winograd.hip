// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the deep layers (layer3 / layer4 / the PPM head's 4096 -> 512 conv),
// exact fp32 (gfx950).  Y = A^T [ (G g G^T) (.) (B^T d B) ] A summed over input channels: 16 multiplies per 2x2 output block and
// channel pair instead of 36, i.e. 2.25x fewer MFMA flops for the same fp32 result up to summation order (measured against fp64:
// 4.2e-7 relative L2 against 1.9e-7 for the direct sum at Cin = 512 -- inside the 2-8e-7 band of the direct kernels).
//
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
//
// The 16 element-wise products over Cin are 16 independent GEMMs [tiles x Cin] x [Cin x Cout]: they run on the f32-MFMA 1x1 kernel
// of conv.hip as ONE launch (`uem_wino_gemm`: rows = (position, tile), the filter bank of a row tile chosen by its position).  The
// kernels in this file are the HBM-bound transforms around it:
//   uem_wino_input    x (N,H,W,C) [-> relu(x*scale + shift), the producer's BatchNorm, zero padding AFTER it] -> V[16][T][C]
//   uem_wino_output   M[16][T][C] -> y (N,H,W,C), with the per-128-pixel BatchNorm statistics (forward) or the first pass of the
//                     consumer's BatchNorm+ReLU backward (data gradient) in the same [2][C][M/128] format as the conv epilogues
//   uem_wino_dy       dY (N,H,W,C) -> dM[16][T][C] = A dY A^T     (weight gradient: dU[pos] = sum_tiles dM[pos]^T V[pos], a batched
//                     pixel-reduction GEMM on the wgrad kernel, then uem_wino_filter_grad: dW += G^T dU G)
//   uem_wino_filter   W (Cout,3,3,Cin) -> U[16][Cout][Cin]; flipped + transposed -> U'[16][Cin][Cout] for the data gradient
//                     (dX = conv(dY, W') with W'[ci][ky][kx][co] = W[co][2-ky][2-kx][ci], same padding and dilation)
// Dilation d: the d*d interleaved sub-images are independent dilation-1 convolutions; a tile is (image, sub-image, ty, tx).
// Replaces cuDNN behind nn.Conv2d(3x3): reference uemda/_resnets.py:100-103 (conv2 of a Bottleneck), Encoder.py:35 (PPM conv_last).
#include "common.h"

struct WinoP {
    const float* x;            // NHWC tensor read (input transform, dY transform) or written (output transform)
    float* v;                  // transform-domain tensor [16][T][C]
    const float* scale;        // input transform: optional BatchNorm affine (+ ReLU) prologue
    const float* shift;
    int relu;
    int N, H, W, C, d;
    int th, tw, T;             // tiles per sub-image (rows, cols), tiles in all
    // output transform extras
    float* tile_stats;         // [2][C][M/128]: sum y, sum y*y per 128-pixel group (32 tiles)
    const float* bn_z;         // data gradient: first pass of the consumer's BatchNorm+ReLU backward
    const float* bn_vec;       // (4, C): scale, shift, mean, invstd
    float* tile_bnbwd;         // [2][C][M/128]: sum dp, sum dp*xhat
    // dY transform extra: a buffer the launch also zeroes (the weight gradient's dU accumulator, so that no separate fill is launched)
    float* zero;
    long long zero_q;          // its size in float4
};

// every block clears its slice of p.zero (16-byte stores)
__device__ __forceinline__ void wino_zero_slice(const WinoP& p) {
    if (p.zero == nullptr) return;
    const long long nblk = (long long)gridDim.x * gridDim.y, bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    const long long per = (p.zero_q + nblk - 1) / nblk;
    for (long long i = threadIdx.x; i < per; i += 256) {
        const long long idx = bid * per + i;
        if (idx < p.zero_q) reinterpret_cast<float4*>(p.zero)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// tile -> (image, sub-image row / column, tile row / column); th / tw are tiles per sub-image for the launch's tile edge (2 or 4)
__device__ __forceinline__ void wino_tile(const WinoP& p, const int tile, int& n, int& ry, int& rx, int& ty, int& tx) {
    tx = tile % p.tw;
    int r = tile / p.tw;
    ty = r % p.th;
    r /= p.th;
    const int dd = p.d * p.d;
    const int sub = r % dd;
    n = r / dd;
    ry = sub / p.d;
    rx = sub - ry * p.d;
}

__device__ __forceinline__ float4 f4add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4neg(const float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ float4 f4half(const float4 a) { return make_float4(0.5f * a.x, 0.5f * a.y, 0.5f * a.z, 0.5f * a.w); }

// block = 16 channel quads (64 channels) x 16 tiles; grid = (T / 16, C / 64)
// PASS (0 forward, 1 data gradient, 2 weight gradient recomputing V) changes nothing but the kernel's NAME: profiles attribute the
// launch to the family it belongs to (the PPM head's forward transform has no affine prologue either)
template <bool AFFINE, int PASS>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoP p) {
    const int q = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int tile = blockIdx.x * 16 + tl;
    const int c = blockIdx.y * 64 + q * 4;
    if (tile >= p.T) return;
    int n, ry, rx, ty, tx;
    wino_tile(p, tile, n, ry, rx, ty, tx);
    const int Hs = p.H / p.d, Ws = p.W / p.d;
    float4 sc, sh;
    if (AFFINE) { sc = *reinterpret_cast<const float4*>(p.scale + c); sh = *reinterpret_cast<const float4*>(p.shift + c); }
    float4 dv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int Y = 2 * ty - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int X = 2 * tx - 1 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Y >= 0 && Y < Hs && X >= 0 && X < Ws) {
                v = *reinterpret_cast<const float4*>(p.x + (((size_t)n * p.H + (size_t)(Y * p.d + ry)) * p.W + (size_t)(X * p.d + rx)) * p.C + c);
                if (AFFINE) {           // one fused rounding, as the conv kernels' operand fetch; zero padding stays zero
                    v.x = __builtin_fmaf(v.x, sc.x, sh.x); v.y = __builtin_fmaf(v.y, sc.y, sh.y);
                    v.z = __builtin_fmaf(v.z, sc.z, sh.z); v.w = __builtin_fmaf(v.w, sc.w, sh.w);
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
            }
            dv[i][j] = v;
        }
    }
    // B^T d: rows
    float4 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = f4sub(dv[0][j], dv[2][j]);
        t[1][j] = f4add(dv[1][j], dv[2][j]);
        t[2][j] = f4sub(dv[2][j], dv[1][j]);
        t[3][j] = f4sub(dv[1][j], dv[3][j]);
    }
    // (B^T d) B: columns
    const size_t pos_stride = (size_t)p.T * p.C;
    float* const out = p.v + (size_t)tile * p.C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 0) * pos_stride) = f4sub(t[i][0], t[i][2]);
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 1) * pos_stride) = f4add(t[i][1], t[i][2]);
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 2) * pos_stride) = f4sub(t[i][2], t[i][1]);
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 3) * pos_stride) = f4sub(t[i][1], t[i][3]);
    }
}

// dM = A dY A^T with A = [1 0; 1 1; 1 -1; 0 -1]; same thread mapping as the input transform
__global__ __launch_bounds__(256) void wino_dy_kernel(const WinoP p) {
    wino_zero_slice(p);
    const int q = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int tile = blockIdx.x * 16 + tl;
    const int c = blockIdx.y * 64 + q * 4;
    if (tile >= p.T) return;
    int n, ry, rx, ty, tx;
    wino_tile(p, tile, n, ry, rx, ty, tx);
    float4 g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            g[a][b] = *reinterpret_cast<const float4*>(p.x + (((size_t)n * p.H + (size_t)((2 * ty + a) * p.d + ry)) * p.W + (size_t)((2 * tx + b) * p.d + rx)) * p.C + c);
    float4 u[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u[0][s] = g[0][s];
        u[1][s] = f4add(g[0][s], g[1][s]);
        u[2][s] = f4sub(g[0][s], g[1][s]);
        u[3][s] = f4neg(g[1][s]);
    }
    const size_t pos_stride = (size_t)p.T * p.C;
    float* const out = p.v + (size_t)tile * p.C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 0) * pos_stride) = u[i][0];
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 1) * pos_stride) = f4add(u[i][0], u[i][1]);
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 2) * pos_stride) = f4sub(u[i][0], u[i][1]);
        *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 3) * pos_stride) = f4neg(u[i][1]);
    }
}

// block = 16 channel quads (64 channels) x 16 tile lanes x 2 tiles each = 32 tiles = 128 output pixels = one statistics group;
// grid = (T / 32, C / 64).  EXTRA: 0 plain, 1 BatchNorm statistics of y (forward), 2 BatchNorm+ReLU backward partials (data gradient)
template <int EXTRA>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoP p) {
    __shared__ float red[2][16][64];
    const int q = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + q * 4;
    const size_t pos_stride = (size_t)p.T * p.C;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), bg = bb;
    float4 sc, sh, mu, is;
    if (EXTRA == 2) {
        sc = *reinterpret_cast<const float4*>(p.bn_vec + c);
        sh = *reinterpret_cast<const float4*>(p.bn_vec + p.C + c);
        mu = *reinterpret_cast<const float4*>(p.bn_vec + 2 * p.C + c);
        is = *reinterpret_cast<const float4*>(p.bn_vec + 3 * p.C + c);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int tile = blockIdx.x * 32 + it * 16 + tl;
        if (tile >= p.T) continue;
        int n, ry, rx, ty, tx;
        wino_tile(p, tile, n, ry, rx, ty, tx);
        const float* const in = p.v + (size_t)tile * p.C + c;
        float4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const float4*>(in + (size_t)(i * 4 + j) * pos_stride);
        float4 t[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
            t[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float4 y[2];
            y[0] = f4add(f4add(t[a][0], t[a][1]), t[a][2]);
            y[1] = f4sub(f4sub(t[a][1], t[a][2]), t[a][3]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const size_t off = (((size_t)n * p.H + (size_t)((2 * ty + a) * p.d + ry)) * p.W + (size_t)((2 * tx + b) * p.d + rx)) * p.C + c;
                const float4 v = y[b];
                *reinterpret_cast<float4*>(const_cast<float*>(p.x) + off) = v;
                if (EXTRA == 1) {
                    bb = f4add(bb, v);
                    bg.x = fmaf(v.x, v.x, bg.x); bg.y = fmaf(v.y, v.y, bg.y); bg.z = fmaf(v.z, v.z, bg.z); bg.w = fmaf(v.w, v.w, bg.w);
                }
                if (EXTRA == 2) {
                    const float4 z = *reinterpret_cast<const float4*>(p.bn_z + off);
                    const float dx_ = z.x * sc.x + sh.x > 0.f ? v.x : 0.f, dy_ = z.y * sc.y + sh.y > 0.f ? v.y : 0.f;
                    const float dz_ = z.z * sc.z + sh.z > 0.f ? v.z : 0.f, dw_ = z.w * sc.w + sh.w > 0.f ? v.w : 0.f;
                    bb.x += dx_; bb.y += dy_; bb.z += dz_; bb.w += dw_;
                    bg.x += dx_ * ((z.x - mu.x) * is.x); bg.y += dy_ * ((z.y - mu.y) * is.y);
                    bg.z += dz_ * ((z.z - mu.z) * is.z); bg.w += dw_ * ((z.w - mu.w) * is.w);
                }
            }
        }
    }
    if (EXTRA == 0) return;
    *reinterpret_cast<float4*>(&red[0][tl][q * 4]) = bb;
    *reinterpret_cast<float4*>(&red[1][tl][q * 4]) = bg;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += red[which][r][col];
        float* const out = EXTRA == 1 ? p.tile_stats : p.tile_bnbwd;
        out[((size_t)which * p.C + blockIdx.y * 64 + col) * (size_t)gridDim.x + blockIdx.x] = a;
    }
}

// U = G g G^T.  TRANSPOSED = false: U[pos][co][ci] from w[co][ky][kx][ci];  true: U'[pos][ci][co] from the flipped taps w[co][2-ky][2-kx][ci]
// (the data gradient's filter bank).  One thread per (co, ci quad); the transposed form goes through LDS so that both the reads
// (ci contiguous) and the writes (co contiguous) are coalesced: block = 16 co x 16 ci quads.
#define WINO_FILTER_SMEM (8 * 16 * 65)          // floats: the transposed forms stage [positions][16 co][64 ci + 1 pad] through LDS
template <bool TRANSPOSED>
__device__ __forceinline__ void wino_filter_body(const float* __restrict__ w, float* __restrict__ U, const int Cout, const int Cin,
                                                 const int bx, const int by, float* __restrict__ smem) {
    // TRANSPOSED: smem is [pos (half of them)][co][ci (64 + 1 pad)]
    const int q = threadIdx.x & 15, col = threadIdx.x >> 4;
    const int co = by * 16 + col, ci = bx * 64 + q * 4;
    float4 g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int ky = TRANSPOSED ? 2 - a : a, kx = TRANSPOSED ? 2 - b : b;
            g[a][b] = *reinterpret_cast<const float4*>(w + (((size_t)co * 3 + ky) * 3 + kx) * Cin + ci);
        }
    float4 r[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        r[0][b] = g[0][b];
        r[1][b] = f4half(f4add(f4add(g[0][b], g[1][b]), g[2][b]));
        r[2][b] = f4half(f4add(f4sub(g[0][b], g[1][b]), g[2][b]));
        r[3][b] = g[2][b];
    }
    const size_t pos_stride = (size_t)Cout * Cin;
    const int wco = threadIdx.x & 15, wci0 = threadIdx.x >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii;
            float4 u[4];
            u[0] = r[i][0];
            u[1] = f4half(f4add(f4add(r[i][0], r[i][1]), r[i][2]));
            u[2] = f4half(f4add(f4sub(r[i][0], r[i][1]), r[i][2]));
            u[3] = r[i][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!TRANSPOSED) {
                    *reinterpret_cast<float4*>(U + (size_t)(i * 4 + j) * pos_stride + (size_t)co * Cin + ci) = u[j];
                } else {
                    float* t = smem + ((ii * 4 + j) * 16 + col) * 65 + q * 4;
                    t[0] = u[j].x; t[1] = u[j].y; t[2] = u[j].z; t[3] = u[j].w;
                }
            }
        }
        if (TRANSPOSED) {
            __syncthreads();
            // write U'[pos][ci][co]: 16 consecutive co per ci row
#pragma unroll
            for (int pl = 0; pl < 8; ++pl)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cil = wci0 + 16 * k;
                    U[(size_t)(half * 8 + pl) * pos_stride + (size_t)(bx * 64 + cil) * Cout + by * 16 + wco] = smem[(pl * 16 + wco) * 65 + cil];
                }
            __syncthreads();
        }
    }
}
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, const int Cout, const int Cin) {
    __shared__ float smem[TRANSPOSED ? WINO_FILTER_SMEM : 1];
    wino_filter_body<TRANSPOSED>(w, U, Cout, Cin, blockIdx.x, blockIdx.y, smem);
}

// dW[co][ky][kx][ci] += (G^T dU G)[ky][kx], dU[pos][co][ci]; one thread per (co, ci quad)
__global__ __launch_bounds__(256) void wino_filter_grad_kernel(const float* __restrict__ dU, float* __restrict__ dw, const int Cout, const int Cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int cq = Cin / 4;
    if (idx >= Cout * cq) return;
    const int co = idx / cq, ci = (idx - co * cq) * 4;
    const size_t pos_stride = (size_t)Cout * Cin;
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const float4*>(dU + (size_t)(i * 4 + j) * pos_stride + (size_t)co * Cin + ci);
    float4 s[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 h1 = f4half(m[1][j]), h2 = f4half(m[2][j]);
        s[0][j] = f4add(f4add(m[0][j], h1), h2);
        s[1][j] = f4sub(h1, h2);
        s[2][j] = f4add(f4add(h1, h2), m[3][j]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float4 h1 = f4half(s[a][1]), h2 = f4half(s[a][2]);
        float4 o[3];
        o[0] = f4add(f4add(s[a][0], h1), h2);
        o[1] = f4sub(h1, h2);
        o[2] = f4add(f4add(h1, h2), s[a][3]);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float4* dst = reinterpret_cast<float4*>(dw + (((size_t)co * 3 + a) * 3 + b) * Cin + ci);
            *dst = f4add(*dst, o[b]);
        }
    }
}


// =====================================================================================================================
// F(4x4, 3x3): 36 multiplies per 4x4 output block and channel pair instead of 144 (4x fewer MFMA flops, 1.78x fewer than F(2x2,3x3)),
// transform-domain tensors [36][T][C] with T = N*H*W/16 -- 2.25x the activation instead of 4x.  Interpolation points
// (0, 1, -1, 1/2, -2, inf): against float64 a Cin = 512 convolution comes out at 1.4e-6 relative L2 (the textbook points 0, +-1, +-2:
// 2.1e-6; F(2x2,3x3): 3e-7; the direct fmaf chain: 2e-7), every entry of B^T and A^T an exact binary fraction.
//   B^T = [1 -3/2 -2 3/2 1 0; 0 -1 1/2 5/2 1 0; 0 1 -5/2 1/2 1 0; 0 -2 -1 2 1 0; 0 1/2 -1 -1/2 1 0; 0 1 -3/2 -2 3/2 1]
//   G   = [1 0 0; 1/3 1/3 1/3; -1/3 1/3 -1/3; -16/15 -8/15 -4/15; 1/15 -2/15 4/15; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 0; 0 1 -1 1/8 -8 1]
// Used where the gradient noise floor (2-3 % per encoder tensor) dwarfs the transform error: data and weight gradients;
// the forward takes it only when ops.WINOGRAD_F4_FWD says so.
// =====================================================================================================================
struct W4 {
    __device__ static constexpr float bt(const int r, const int c) {
        constexpr float t[6][6] = {{1.f, -1.5f, -2.f, 1.5f, 1.f, 0.f}, {0.f, -1.f, 0.5f, 2.5f, 1.f, 0.f}, {0.f, 1.f, -2.5f, 0.5f, 1.f, 0.f},
                                   {0.f, -2.f, -1.f, 2.f, 1.f, 0.f},   {0.f, 0.5f, -1.f, -0.5f, 1.f, 0.f}, {0.f, 1.f, -1.5f, -2.f, 1.5f, 1.f}};
        return t[r][c];
    }
    __device__ static constexpr float at(const int r, const int c) {
        constexpr float t[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 0.5f, -2.f, 0.f}, {0.f, 1.f, 1.f, 0.25f, 4.f, 0.f},
                                   {0.f, 1.f, -1.f, 0.125f, -8.f, 1.f}};
        return t[r][c];
    }
    __device__ static constexpr float g(const int r, const int c) {
        constexpr float t[6][3] = {{1.f, 0.f, 0.f}, {1.f / 3.f, 1.f / 3.f, 1.f / 3.f}, {-1.f / 3.f, 1.f / 3.f, -1.f / 3.f},
                                   {-16.f / 15.f, -8.f / 15.f, -4.f / 15.f}, {1.f / 15.f, -2.f / 15.f, 4.f / 15.f}, {0.f, 0.f, 1.f}};
        return t[r][c];
    }
};

// acc (+)= k * a with the compile-time constant k folded: 0 skips, +-1 adds / subtracts, anything else is one fma per component
__device__ __forceinline__ void f4mac(float4& acc, bool& first, const float k, const float4 a) {
    if (k == 0.f) return;
    if (first) {
        acc = k == 1.f ? a : (k == -1.f ? f4neg(a) : make_float4(k * a.x, k * a.y, k * a.z, k * a.w));
        first = false;
    } else if (k == 1.f) acc = f4add(acc, a);
    else if (k == -1.f) acc = f4sub(acc, a);
    else acc = make_float4(__builtin_fmaf(k, a.x, acc.x), __builtin_fmaf(k, a.y, acc.y), __builtin_fmaf(k, a.z, acc.z), __builtin_fmaf(k, a.w, acc.w));
}

// V = B^T d B; block = 16 channel quads (64 channels) x 16 tiles; grid = (T / 16, C / 64).  A window column is transformed as soon as
// it is loaded, so 36 (not 72) float4 stay live.
template <bool AFFINE, int PASS>
__global__ __launch_bounds__(256) void wino4_input_kernel(const WinoP p) {
    const int q = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int tile = blockIdx.x * 16 + tl;
    const int c = blockIdx.y * 64 + q * 4;
    if (tile >= p.T) return;
    int n, ry, rx, ty, tx;
    wino_tile(p, tile, n, ry, rx, ty, tx);
    const int Hs = p.H / p.d, Ws = p.W / p.d;
    float4 sc, sh;
    if (AFFINE) { sc = *reinterpret_cast<const float4*>(p.scale + c); sh = *reinterpret_cast<const float4*>(p.shift + c); }
    float4 t[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int X = 4 * tx - 1 + j;
        float4 dcol[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int Y = 4 * ty - 1 + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Y >= 0 && Y < Hs && X >= 0 && X < Ws) {
                v = *reinterpret_cast<const float4*>(p.x + (((size_t)n * p.H + (size_t)(Y * p.d + ry)) * p.W + (size_t)(X * p.d + rx)) * p.C + c);
                if (AFFINE) {
                    v.x = __builtin_fmaf(v.x, sc.x, sh.x); v.y = __builtin_fmaf(v.y, sc.y, sh.y);
                    v.z = __builtin_fmaf(v.z, sc.z, sh.z); v.w = __builtin_fmaf(v.w, sc.w, sh.w);
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
            }
            dcol[i] = v;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int i = 0; i < 6; ++i) f4mac(acc, first, W4::bt(a, i), dcol[i]);
            t[a][j] = acc;
        }
    }
    const size_t pos_stride = (size_t)p.T * p.C;
    float* const out = p.v + (size_t)tile * p.C + c;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int j = 0; j < 6; ++j) f4mac(acc, first, W4::bt(b, j), t[a][j]);
            *reinterpret_cast<float4*>(out + (size_t)(a * 6 + b) * pos_stride) = acc;
        }
}

// dM = A dY A^T (A = (A^T)^T, 6x4): the weight gradient's transform of a 4x4 block of dY
__global__ __launch_bounds__(256) void wino4_dy_kernel(const WinoP p) {
    wino_zero_slice(p);
    const int q = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int tile = blockIdx.x * 16 + tl;
    const int c = blockIdx.y * 64 + q * 4;
    if (tile >= p.T) return;
    int n, ry, rx, ty, tx;
    wino_tile(p, tile, n, ry, rx, ty, tx);
    float4 u[6][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float4 gcol[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            gcol[r] = *reinterpret_cast<const float4*>(p.x + (((size_t)n * p.H + (size_t)((4 * ty + r) * p.d + ry)) * p.W + (size_t)((4 * tx + s) * p.d + rx)) * p.C + c);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) f4mac(acc, first, W4::at(r, a), gcol[r]);
            u[a][s] = acc;
        }
    }
    const size_t pos_stride = (size_t)p.T * p.C;
    float* const out = p.v + (size_t)tile * p.C + c;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int s = 0; s < 4; ++s) f4mac(acc, first, W4::at(s, b), u[a][s]);
            *reinterpret_cast<float4*>(out + (size_t)(a * 6 + b) * pos_stride) = acc;
        }
}

// y = A^T M A; block = 16 channel quads x 16 tiles of 16 pixels = two 128-pixel statistics groups; grid = (T / 16, C / 64).
// EXTRA as in wino_output_kernel.
template <int EXTRA>
__global__ __launch_bounds__(256) void wino4_output_kernel(const WinoP p) {
    __shared__ float red[2][16][64];
    const int q = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + q * 4;
    const size_t pos_stride = (size_t)p.T * p.C;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), bg = bb;
    float4 sc, sh, mu, is;
    if (EXTRA == 2) {
        sc = *reinterpret_cast<const float4*>(p.bn_vec + c);
        sh = *reinterpret_cast<const float4*>(p.bn_vec + p.C + c);
        mu = *reinterpret_cast<const float4*>(p.bn_vec + 2 * p.C + c);
        is = *reinterpret_cast<const float4*>(p.bn_vec + 3 * p.C + c);
    }
    const int tile = blockIdx.x * 16 + tl;                       // T is a multiple of 16 (wino_geometry)
    int n, ry, rx, ty, tx;
    wino_tile(p, tile, n, ry, rx, ty, tx);
    const float* const in = p.v + (size_t)tile * p.C + c;
    float4 t[4][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float4 mc[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) mc[a] = *reinterpret_cast<const float4*>(in + (size_t)(a * 6 + b) * pos_stride);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int a = 0; a < 6; ++a) f4mac(acc, first, W4::at(r, a), mc[a]);
            t[r][b] = acc;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int b = 0; b < 6; ++b) f4mac(v, first, W4::at(s, b), t[r][b]);
            const size_t off = (((size_t)n * p.H + (size_t)((4 * ty + r) * p.d + ry)) * p.W + (size_t)((4 * tx + s) * p.d + rx)) * p.C + c;
            *reinterpret_cast<float4*>(const_cast<float*>(p.x) + off) = v;
            if (EXTRA == 1) {
                bb = f4add(bb, v);
                bg.x = fmaf(v.x, v.x, bg.x); bg.y = fmaf(v.y, v.y, bg.y); bg.z = fmaf(v.z, v.z, bg.z); bg.w = fmaf(v.w, v.w, bg.w);
            }
            if (EXTRA == 2) {
                const float4 z = *reinterpret_cast<const float4*>(p.bn_z + off);
                const float dx_ = z.x * sc.x + sh.x > 0.f ? v.x : 0.f, dy_ = z.y * sc.y + sh.y > 0.f ? v.y : 0.f;
                const float dz_ = z.z * sc.z + sh.z > 0.f ? v.z : 0.f, dw_ = z.w * sc.w + sh.w > 0.f ? v.w : 0.f;
                bb.x += dx_; bb.y += dy_; bb.z += dz_; bb.w += dw_;
                bg.x += dx_ * ((z.x - mu.x) * is.x); bg.y += dy_ * ((z.y - mu.y) * is.y);
                bg.z += dz_ * ((z.z - mu.z) * is.z); bg.w += dw_ * ((z.w - mu.w) * is.w);
            }
        }
    if (EXTRA == 0) return;
    *reinterpret_cast<float4*>(&red[0][tl][q * 4]) = bb;
    *reinterpret_cast<float4*>(&red[1][tl][q * 4]) = bg;
    __syncthreads();
    {
        const int which = threadIdx.x >> 7, grp = (threadIdx.x >> 6) & 1, col = threadIdx.x & 63;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) a += red[which][grp * 8 + r][col];
        float* const out = EXTRA == 1 ? p.tile_stats : p.tile_bnbwd;
        out[((size_t)which * p.C + blockIdx.y * 64 + col) * (size_t)(2 * gridDim.x) + 2 * blockIdx.x + grp] = a;
    }
}

// U = G g G^T, 36 positions; TRANSPOSED as in wino_filter_body (flipped taps, U'[pos][ci][co] through LDS, one U row at a time)
template <bool TRANSPOSED>
__device__ __forceinline__ void wino4_filter_body(const float* __restrict__ w, float* __restrict__ U, const int Cout, const int Cin,
                                                  const int bx, const int by, float* __restrict__ smem) {
    const int q = threadIdx.x & 15, col = threadIdx.x >> 4;
    const int co = by * 16 + col, ci = bx * 64 + q * 4;
    float4 r[6][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float4 gcol[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int ky = TRANSPOSED ? 2 - a : a, kx = TRANSPOSED ? 2 - b : b;
            gcol[a] = *reinterpret_cast<const float4*>(w + (((size_t)co * 3 + ky) * 3 + kx) * Cin + ci);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) f4mac(acc, first, W4::g(i, a), gcol[a]);
            r[i][b] = acc;
        }
    }
    const size_t pos_stride = (size_t)Cout * Cin;
    const int wco = threadIdx.x & 15, wci0 = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int b = 0; b < 3; ++b) f4mac(u, first, W4::g(j, b), r[i][b]);
            if (!TRANSPOSED) {
                *reinterpret_cast<float4*>(U + (size_t)(i * 6 + j) * pos_stride + (size_t)co * Cin + ci) = u;
            } else {
                float* t = smem + (j * 16 + col) * 65 + q * 4;
                t[0] = u.x; t[1] = u.y; t[2] = u.z; t[3] = u.w;
            }
        }
        if (TRANSPOSED) {
            __syncthreads();
#pragma unroll
            for (int pl = 0; pl < 6; ++pl)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cil = wci0 + 16 * k;
                    U[(size_t)(i * 6 + pl) * pos_stride + (size_t)(bx * 64 + cil) * Cout + by * 16 + wco] = smem[(pl * 16 + wco) * 65 + cil];
                }
            __syncthreads();
        }
    }
}
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ U, const int Cout, const int Cin) {
    __shared__ float smem[TRANSPOSED ? WINO_FILTER_SMEM : 1];
    wino4_filter_body<TRANSPOSED>(w, U, Cout, Cin, blockIdx.x, blockIdx.y, smem);
}

// dW[co][ky][kx][ci] += (G^T dU G)[ky][kx], dU[36][co][ci]; one thread per (co, ci quad)
__global__ __launch_bounds__(256) void wino4_filter_grad_kernel(const float* __restrict__ dU, float* __restrict__ dw, const int Cout, const int Cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int cq = Cin / 4;
    if (idx >= Cout * cq) return;
    const int co = idx / cq, ci = (idx - co * cq) * 4;
    const size_t pos_stride = (size_t)Cout * Cin;
    float4 s[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        float4 mc[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) mc[a] = *reinterpret_cast<const float4*>(dU + (size_t)(a * 6 + b) * pos_stride + (size_t)co * Cin + ci);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int a = 0; a < 6; ++a) f4mac(acc, first, W4::g(a, k), mc[a]);
            s[k][b] = acc;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            bool first = true;
#pragma unroll
            for (int b = 0; b < 6; ++b) f4mac(o, first, W4::g(b, l), s[k][b]);
            float4* dst = reinterpret_cast<float4*>(dw + (((size_t)co * 3 + k) * 3 + l) * Cin + ci);
            *dst = f4add(*dst, o);
        }
}

static int wino_geometry(WinoP& p, int N, int H, int W, int C, int d, int m, const char* what) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || d <= 0 || d > 4) return uem_fail(UEM_ERR_INVALID, "%s: bad shape", what);
    if (m != 2 && m != 4) return uem_fail(UEM_ERR_INVALID, "%s: output tile edge must be 2 (F(2x2,3x3)) or 4 (F(4x4,3x3))", what);
    if (H % (m * d) != 0 || W % (m * d) != 0 || C % 64 != 0)
        return uem_fail(UEM_ERR_UNSUPPORTED, "%s: needs H, W multiples of tile*dilation and C a multiple of 64", what);
    p.N = N; p.H = H; p.W = W; p.C = C; p.d = d;
    p.th = H / d / m; p.tw = W / d / m;
    const int64_t T = (int64_t)N * d * d * p.th * p.tw;
    const int64_t npos = (m + 2) * (m + 2);
    if (T * npos * C >= ((int64_t)1 << 30) || T % 32 != 0)               // 32-bit byte offsets into the transform-domain tensor
        return uem_fail(UEM_ERR_UNSUPPORTED, "%s: needs the tile count a multiple of 32 and positions*T*C < 2^30 elements (split the batch)", what);
    p.T = (int)T;
    p.tile_stats = nullptr; p.bn_z = nullptr; p.bn_vec = nullptr; p.tile_bnbwd = nullptr; p.scale = p.shift = nullptr; p.relu = 0;
    p.zero = nullptr; p.zero_q = 0;
    return UEM_OK;
}

template <int PASS>
static void wino_input_go(const WinoP& p, int m, bool affine, dim3 grid, hipStream_t st) {
    if (m == 2) {
        if (affine) wino_input_kernel<true, PASS><<<grid, 256, 0, st>>>(p);
        else wino_input_kernel<false, PASS><<<grid, 256, 0, st>>>(p);
    } else {
        if (affine) wino4_input_kernel<true, PASS><<<grid, 256, 0, st>>>(p);
        else wino4_input_kernel<false, PASS><<<grid, 256, 0, st>>>(p);
    }
}

extern "C" int uem_wino_input(const float* x, const float* in_scale, const float* in_shift, int relu, float* V, int N, int H, int W,
                              int C, int dil, int m, int pass, void* stream) {
    UEM_REQUIRE(x && V, "wino_input: null pointer");
    UEM_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "wino_input: scale and shift go together");
    UEM_REQUIRE(pass >= 0 && pass <= 2, "wino_input: pass is 0 (forward), 1 (data gradient) or 2 (weight gradient)");
    WinoP p;
    const int rc = wino_geometry(p, N, H, W, C, dil, m, "wino_input");
    if (rc) return rc;
    p.x = x; p.v = V; p.scale = in_scale; p.shift = in_shift; p.relu = relu;
    const dim3 grid((unsigned)(p.T / 16), (unsigned)(C / 64));
    hipStream_t st = (hipStream_t)stream;
    if (pass == 0) wino_input_go<0>(p, m, in_scale != nullptr, grid, st);
    else if (pass == 1) wino_input_go<1>(p, m, in_scale != nullptr, grid, st);
    else wino_input_go<2>(p, m, in_scale != nullptr, grid, st);
    return uem_check_launch("wino_input");
}

extern "C" int uem_wino_dy(const float* dy, float* dM, int N, int H, int W, int C, int dil, int m, float* zero, int64_t zero_floats,
                           void* stream) {
    UEM_REQUIRE(dy && dM, "wino_dy: null pointer");
    UEM_REQUIRE((zero == nullptr && zero_floats == 0) || (zero != nullptr && zero_floats > 0 && zero_floats % 4 == 0 && ((uintptr_t)zero & 15) == 0),
                "wino_dy: the buffer to clear needs a 16-byte aligned address and a multiple of 4 floats");
    WinoP p;
    const int rc = wino_geometry(p, N, H, W, C, dil, m, "wino_dy");
    if (rc) return rc;
    p.x = dy; p.v = dM; p.zero = zero; p.zero_q = zero_floats / 4;
    const dim3 grid((unsigned)(p.T / 16), (unsigned)(C / 64));
    if (m == 2) wino_dy_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
    else wino4_dy_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
    return uem_check_launch("wino_dy");
}

extern "C" int uem_wino_output(const float* Mt, float* y, int N, int H, int W, int C, int dil, int m, float* tile_stats, const float* bn_z,
                               const float* bn_vec, float* tile_bnbwd, void* stream) {
    UEM_REQUIRE(Mt && y, "wino_output: null pointer");
    UEM_REQUIRE((bn_z == nullptr) == (bn_vec == nullptr) && (bn_z == nullptr) == (tile_bnbwd == nullptr), "wino_output: bn_z, bn_vec and tile_bnbwd go together");
    UEM_REQUIRE(!(tile_stats && tile_bnbwd), "wino_output: either forward statistics or backward partials");
    WinoP p;
    const int rc = wino_geometry(p, N, H, W, C, dil, m, "wino_output");
    if (rc) return rc;
    p.x = y; p.v = const_cast<float*>(Mt); p.tile_stats = tile_stats; p.bn_z = bn_z; p.bn_vec = bn_vec; p.tile_bnbwd = tile_bnbwd;
    hipStream_t st = (hipStream_t)stream;
    if (m == 2) {
        const dim3 grid((unsigned)(p.T / 32), (unsigned)(C / 64));
        if (tile_stats) wino_output_kernel<1><<<grid, 256, 0, st>>>(p);
        else if (tile_bnbwd) wino_output_kernel<2><<<grid, 256, 0, st>>>(p);
        else wino_output_kernel<0><<<grid, 256, 0, st>>>(p);
    } else {
        const dim3 grid((unsigned)(p.T / 16), (unsigned)(C / 64));
        if (tile_stats) wino4_output_kernel<1><<<grid, 256, 0, st>>>(p);
        else if (tile_bnbwd) wino4_output_kernel<2><<<grid, 256, 0, st>>>(p);
        else wino4_output_kernel<0><<<grid, 256, 0, st>>>(p);
    }
    return uem_check_launch("wino_output");
}

extern "C" int uem_wino_filter(const float* w_ohwi, float* U, int Cout, int Cin, int transposed, int m, void* stream) {
    UEM_REQUIRE(w_ohwi && U && Cout > 0 && Cin > 0 && (m == 2 || m == 4), "wino_filter: bad arguments");
    if (Cout % 16 != 0 || Cin % 64 != 0) return uem_fail(UEM_ERR_UNSUPPORTED, "wino_filter: needs Cout %% 16 == 0 and Cin %% 64 == 0");
    const dim3 grid((unsigned)(Cin / 64), (unsigned)(Cout / 16));
    hipStream_t st = (hipStream_t)stream;
    if (m == 2) {
        if (transposed) wino_filter_kernel<true><<<grid, 256, 0, st>>>(w_ohwi, U, Cout, Cin);
        else wino_filter_kernel<false><<<grid, 256, 0, st>>>(w_ohwi, U, Cout, Cin);
    } else {
        if (transposed) wino4_filter_kernel<true><<<grid, 256, 0, st>>>(w_ohwi, U, Cout, Cin);
        else wino4_filter_kernel<false><<<grid, 256, 0, st>>>(w_ohwi, U, Cout, Cin);
    }
    return uem_check_launch("wino_filter");
}

extern "C" int uem_wino_filter_grad(const float* dU, float* dw_ohwi, int Cout, int Cin, int m, void* stream) {
    UEM_REQUIRE(dU && dw_ohwi && Cout > 0 && Cin > 0 && Cin % 4 == 0 && (m == 2 || m == 4), "wino_filter_grad: bad arguments");
    const int64_t n = (int64_t)Cout * (Cin / 4);
    if (m == 2) wino_filter_grad_kernel<<<(unsigned)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(dU, dw_ohwi, Cout, Cin);
    else wino4_filter_grad_kernel<<<(unsigned)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(dU, dw_ohwi, Cout, Cin);
    return uem_check_launch("wino_filter_grad");
}

// =====================================================================================================================
// Every per-step weight re-layout of a model in ONE launch (uem_weight_prep): the (Cin,KH,KW,Cout) banks of the direct data gradients,
// the Winograd filter banks U / U' of either tile size, the stem's padded 7x8x4 taps.  A step used to spend ~70 launches of 5-8 us
// on them (46 weight transposes + the filter transforms: profiles/r03_i_launches_per_step.txt); the job table lives in device memory
// and is rebuilt only when the set of weights changes.  Block b belongs to the job whose [starts[j], starts[j+1]) holds it.
// =====================================================================================================================
__global__ __launch_bounds__(256) void weight_prep_kernel(const uem_prep_job* __restrict__ jobs, const int* __restrict__ starts, const int njobs) {
    __shared__ float smem[WINO_FILTER_SMEM];
    int lo = 0, hi = njobs;
    const int b = (int)blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (starts[mid] <= b) lo = mid; else hi = mid;
    }
    const uem_prep_job j = jobs[lo];
    const int lb = b - starts[lo];
    const float* const w = reinterpret_cast<const float*>(j.src);
    float* const out = reinterpret_cast<float*>(j.dst);
    if (j.kind == UEM_PREP_TRANSPOSE) {
        // w[o][t][i] -> wt[i][t][o], 1024 consecutive outputs per block
        const int64_t total = (int64_t)j.cout * j.taps * j.cin;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t idx = (int64_t)lb * 1024 + k * 256 + threadIdx.x;
            if (idx < total) {
                const int o = (int)(idx % j.cout);
                const int64_t r = idx / j.cout;
                const int t = (int)(r % j.taps);
                const int i = (int)(r / j.taps);
                out[idx] = w[((size_t)o * j.taps + t) * j.cin + i];
            }
        }
    } else if (j.kind == UEM_PREP_TRANSPOSE_BF16) {
        // the same re-layout rounded to bf16 (RNE): the data-gradient banks of a bf16-storage model (52 launches of 6 us per step before)
        unsigned short* const outh = reinterpret_cast<unsigned short*>(j.dst);
        const int64_t total = (int64_t)j.cout * j.taps * j.cin;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t idx = (int64_t)lb * 1024 + k * 256 + threadIdx.x;
            if (idx < total) {
                const int o = (int)(idx % j.cout);
                const int64_t r = idx / j.cout;
                const int t = (int)(r % j.taps);
                const int i = (int)(r / j.taps);
                outh[idx] = __builtin_bit_cast(unsigned short, (__bf16)w[((size_t)o * j.taps + t) * j.cin + i]);
            }
        }
    } else if (j.kind == UEM_PREP_STEM_PACK) {
        // w[64][7][7][3] (OHWI) -> w8[64][7][8][4], zero padded
        const int idx = lb * 256 + threadIdx.x;
        if (idx < 64 * 7 * 8 * 4) {
            const int c = idx & 3, kx = (idx >> 2) & 7, ky = (idx >> 5) % 7, o = idx / (7 * 32);
            out[idx] = (c < 3 && kx < 7) ? w[((o * 7 + ky) * 7 + kx) * 3 + c] : 0.f;
        }
    } else {
        const int gx = j.cin / 64;
        const int bx = lb % gx, by = lb / gx;
        if (j.kind == UEM_PREP_WINO2) wino_filter_body<false>(w, out, j.cout, j.cin, bx, by, smem);
        else if (j.kind == UEM_PREP_WINO2_T) wino_filter_body<true>(w, out, j.cout, j.cin, bx, by, smem);
        else if (j.kind == UEM_PREP_WINO4) wino4_filter_body<false>(w, out, j.cout, j.cin, bx, by, smem);
        else wino4_filter_body<true>(w, out, j.cout, j.cin, bx, by, smem);
    }
}

extern "C" int uem_weight_prep_blocks(int kind, int cout, int cin, int taps) {
    if (cout <= 0 || cin <= 0 || taps <= 0) return -1;
    switch (kind) {
    case UEM_PREP_TRANSPOSE: case UEM_PREP_TRANSPOSE_BF16: return (int)uem_cdiv((int64_t)cout * cin * taps, 1024);
    case UEM_PREP_STEM_PACK: return (cout == 64 && cin == 3 && taps == 49) ? (64 * 7 * 8 * 4 + 255) / 256 : -1;
    case UEM_PREP_WINO2: case UEM_PREP_WINO2_T: case UEM_PREP_WINO4: case UEM_PREP_WINO4_T:
        return (taps == 9 && cout % 16 == 0 && cin % 64 == 0) ? (cin / 64) * (cout / 16) : -1;
    default: return -1;
    }
}

extern "C" int uem_weight_prep(const uem_prep_job* jobs_dev, const int* block_starts_dev, int njobs, int total_blocks, void* stream) {
    UEM_REQUIRE(jobs_dev && block_starts_dev && njobs > 0 && total_blocks > 0, "weight_prep: bad arguments");
    weight_prep_kernel<<<(unsigned)total_blocks, 256, 0, (hipStream_t)stream>>>(jobs_dev, block_starts_dev, njobs);
    return uem_check_launch("weight_prep");
}
