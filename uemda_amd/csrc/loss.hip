// Fused "bilinear upsample (align_corners=True) + per-pixel loss" kernels, forward AND backward in
// one pass, in GATHER form: one block per low-resolution logit cell walks the <=34x34 window of
// full-resolution pixels whose interpolation touches that cell, recomputes their softmax, and
// reduces the cell's gradient in registers/LDS -- no atomics, deterministic, and the (B,C,H,W)
// upsampled logits are never materialised.
// Reference: uemda/utils/tools.py:240-254 (loss_calc), uemda/gast/balance.py:81-101 (CrossEntropy),
//            uemda/gast/balance.py:356-423,437-451 (UVEMLoss, loss_calc_uvem).
#include "common.h"

template <int CMAX>
__device__ __forceinline__ void up_logits(const float* __restrict__ low, int C, int w, const Lerp& ly, const Lerp& lx,
                                          float (&v)[CMAX]) {
    const float* r0 = low + (size_t)ly.i0 * w * C;
    const float* r1 = low + (size_t)ly.i1 * w * C;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
            float v00 = r0[lx.i0 * C + c], v01 = r0[lx.i1 * C + c];
            float v10 = r1[lx.i0 * C + c], v11 = r1[lx.i1 * C + c];
            v[c] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
        }
    }
}

// softmax in place; returns log-sum-exp pieces so that ce = -(v[label] - m - log(s))
template <int CMAX>
__device__ __forceinline__ void softmax_ce(float (&v)[CMAX], int C, int label, float& ce) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) m = fmaxf(m, v[c]);
    float s = 0.f, vl = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) { if (c == label) vl = v[c]; v[c] = expf(v[c] - m); s += v[c]; }
    const float inv = 1.0f / s;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) v[c] *= inv;
    ce = (label >= 0) ? -(vl - m - logf(s)) : 0.f;
}

__device__ __forceinline__ float uvem_weight_dev(float u, float m, float t, float inv_gamma) {
    // UVEMLoss.get_weight (balance.py:396-423)
    float left = 1.0f;
    if (m > 0.f) {
        float x = (u <= m && u >= 0.f) ? u : 1.0f;
        float q = (-1.0f / (m * m)) * ((x - m) * (x - m)) + 1.0f;
        q = fminf(fmaxf(q, 0.f), 1.f);
        left = powf(q, inv_gamma);
    }
    float right = 0.f;
    if (m < t) {
        float x = (u > m && u <= t) ? u : 0.f;
        float q = (-1.0f / ((t - m) * (t - m))) * ((x - m) * (x - m)) + 1.0f;
        q = fminf(fmaxf(q, 0.f), 1.f);
        right = powf(q, inv_gamma);
    }
    float wgt = (u <= m) ? left : right;
    return (u >= t) ? 0.f : wgt;
}

// window of destination indices whose lerp can touch source cell `i`
__device__ __forceinline__ void cell_window(int i, int in_size, int out_size, int& lo, int& hi) {
    if (in_size <= 1 || out_size <= 1) { lo = 0; hi = out_size - 1; return; }
    const float inv = (float)(out_size - 1) / (float)(in_size - 1);
    lo = (int)floorf((float)(i - 1) * inv) - 1;
    hi = (int)ceilf((float)(i + 1) * inv) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out_size - 1 ? out_size - 1 : hi;
}

// MODE 0: CE (mean over all pixels), MODE 1: UVEM.  NH = number of heads handled (1 or 2).
template <int CMAX, int MODE>
__global__ __launch_bounds__(256) void loss_gather_kernel(
    const float* __restrict__ lg1, const float* __restrict__ lg2, const int64_t* __restrict__ label,
    const float* __restrict__ soft, const float* __restrict__ pixw, float* __restrict__ dl1, float* __restrict__ dl2,
    float* __restrict__ partial, int C, int h, int w, int H, int W, float um, float ut, float inv_gamma,
    int64_t ignore, float coef) {
    const int cell = blockIdx.x;
    const int cx = cell % w, cy = (cell / w) % h, b = cell / (w * h);
    const size_t plane = (size_t)H * W;
    int ylo, yhi, xlo, xhi;
    cell_window(cy, h, H, ylo, yhi);
    cell_window(cx, w, W, xlo, xhi);
    const int ww = xhi - xlo + 1, wh = yhi - ylo + 1;
    const float* l1b = lg1 + (size_t)b * h * w * C;
    const float* l2b = lg2 ? lg2 + (size_t)b * h * w * C : nullptr;
    // every pixel that touches this cell interpolates inside the 3x3 cell neighbourhood: stage it in LDS once
    __shared__ float nb[2][3][3][CMAX];
    for (int i = threadIdx.x; i < 2 * 9 * CMAX; i += 256) {
        const int c = i % CMAX, q = (i / CMAX) % 9, hd = i / (9 * CMAX);
        const int yy = cy - 1 + q / 3, xx = cx - 1 + q % 3;
        const float* src = hd ? l2b : l1b;
        float v = 0.f;
        if (src != nullptr && c < C && yy >= 0 && yy < h && xx >= 0 && xx < w) v = src[((size_t)yy * w + xx) * C + c];
        nb[hd][q / 3][q % 3][c] = v;
    }
    __syncthreads();
    float g1[CMAX], g2[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) g1[c] = g2[c] = 0.f;
    float loss1 = 0.f, loss2 = 0.f, valid = 0.f;
    for (int i = threadIdx.x; i < ww * wh; i += 256) {
        const int Y = ylo + i / ww, X = xlo + i % ww;
        const Lerp ly = lerp_setup(Y, h, H, true), lx = lerp_setup(X, w, W, true);
        const float wy = (ly.i0 == cy ? ly.l0 : 0.f) + (ly.i1 == cy ? ly.l1 : 0.f);
        const float wx = (lx.i0 == cx ? lx.l0 : 0.f) + (lx.i1 == cx ? lx.l1 : 0.f);
        const bool touches = (ly.i0 == cy || ly.i1 == cy) && (lx.i0 == cx || lx.i1 == cx);
        if (!touches) continue;
        const bool owner = (ly.i0 == cy) && (lx.i0 == cx);       // forward value counted once
        const size_t p = (size_t)Y * W + X;
        const int64_t lab64 = label[(size_t)b * plane + p];
        const bool lab_ok = (lab64 != ignore) && lab64 >= 0 && lab64 < C;
        const int lab = lab_ok ? (int)lab64 : -1;
        float pw = 1.0f;      // per-pixel coefficient on (softmax - onehot)
        if (MODE == 1) {
            float u = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) { float q = soft[((size_t)b * C + c) * plane + p]; u += -q * logf(q); }
            const bool gate = !(u > ut);                          // ce[u > t] = 0
            pw = gate ? uvem_weight_dev(u, um, ut, inv_gamma) : 0.f;
            if (owner && (u <= ut) && lab64 != ignore) valid += 1.f;
        }
        if (pixw) pw *= pixw[(size_t)b * plane + p];
        if (!lab_ok) pw = 0.f;                                    // ignore_index: zero loss and gradient
        float v[CMAX], ce;
        const int ry0 = ly.i0 - cy + 1, ry1 = ly.i1 - cy + 1, rx0 = lx.i0 - cx + 1, rx1 = lx.i1 - cx + 1;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) v[c] = ly.l0 * (lx.l0 * nb[0][ry0][rx0][c] + lx.l1 * nb[0][ry0][rx1][c]) +
                              ly.l1 * (lx.l0 * nb[0][ry1][rx0][c] + lx.l1 * nb[0][ry1][rx1][c]);
        softmax_ce<CMAX>(v, C, lab, ce);
        if (owner) loss1 += pw * ce;
        const float k1 = pw * wy * wx;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) g1[c] += k1 * (v[c] - (c == lab ? 1.f : 0.f));
        if (l2b) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) v[c] = ly.l0 * (lx.l0 * nb[1][ry0][rx0][c] + lx.l1 * nb[1][ry0][rx1][c]) +
                                  ly.l1 * (lx.l0 * nb[1][ry1][rx0][c] + lx.l1 * nb[1][ry1][rx1][c]);
            softmax_ce<CMAX>(v, C, lab, ce);
            if (owner) loss2 += pw * ce;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) g2[c] += k1 * (v[c] - (c == lab ? 1.f : 0.f));
        }
    }
    // block reduction: wave shuffles, then 4 waves through LDS
    __shared__ float red[4][2 * CMAX + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        float a = wave_sum(g1[c]), d = wave_sum(g2[c]);
        if (lane == 0) { red[wave][c] = a; red[wave][CMAX + c] = d; }
    }
    {
        float a = wave_sum(loss1), d = wave_sum(loss2), e = wave_sum(valid);
        if (lane == 0) { red[wave][2 * CMAX] = a; red[wave][2 * CMAX + 1] = d; red[wave][2 * CMAX + 2] = e; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CMAX + 3) {
        const int j = threadIdx.x;
        const float s = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
        if (j < CMAX) { if (j < C) dl1[(size_t)cell * C + j] = s * coef; }
        else if (j < 2 * CMAX) { if (dl2 && (j - CMAX) < C) dl2[(size_t)cell * C + (j - CMAX)] = s * coef; }
        else partial[(size_t)cell * 4 + (j - 2 * CMAX)] = s;
    }
}

// single block: ordered (deterministic) reduction of the per-cell partials
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partial, int cells, int mode,
                                                            int nheads, float ce_denominator, float* __restrict__ loss_out,
                                                            float* __restrict__ inv_valid) {
    double a = 0.0, d = 0.0, e = 0.0;
    for (int i = threadIdx.x; i < cells; i += 256) {
        a += partial[(size_t)i * 4];
        d += partial[(size_t)i * 4 + 1];
        e += partial[(size_t)i * 4 + 2];
    }
    __shared__ double red[3][4];
    a = wave_sum_d(a); d = wave_sum_d(d); e = wave_sum_d(e);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = d; red[2][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        d = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        e = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        if (mode == 0) {
            loss_out[0] = (float)((a + d) / ce_denominator / nheads);
        } else {
            const float den = (float)e + 1e-7f;
            loss_out[0] = ((float)a / den + (float)d / den) / (float)nheads;
            inv_valid[0] = 1.0f / den;
        }
    }
}
__global__ void scale_by_device_scalar_kernel(float* __restrict__ a, float* __restrict__ b, int64_t n,
                                              const float* __restrict__ s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float f = s[0];
        a[i] *= f;
        if (b) b[i] *= f;
    }
}

extern "C" int uem_loss_blocks(int B, int h, int w) { return B * h * w; }
extern "C" int uem_scale_by_scalar(float* a, float* b, int64_t n, const float* scalar, void* stream) {
    UEM_REQUIRE(a && scalar && n > 0, "scale_by_scalar: bad arguments");
    scale_by_device_scalar_kernel<<<(int)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n, scalar);
    return uem_check_launch("scale_by_scalar");
}

extern "C" int uem_ce_upsampled(const float* logits1, const float* logits2, const int64_t* label,
                                const float* pixel_weight, float* loss_out, float* dlogits1, float* dlogits2,
                                float* partial, int B, int C, int h, int w, int H, int W, int64_t ignore_label,
                                float loss_scale, void* stream) {
    UEM_REQUIRE(logits1 && label && loss_out && dlogits1 && partial, "ce_upsampled: null pointer");
    UEM_REQUIRE(!logits2 || dlogits2, "ce_upsampled: dlogits2 required with logits2");
    UEM_REQUIRE(B > 0 && C >= 1 && C <= UEM_MAX_CLASSES && h > 0 && w > 0 && H >= h && W >= w, "ce_upsampled: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int cells = B * h * w;
    const int nheads = logits2 ? 2 : 1;
    // mean over ALL pixels, ignored ones included in the denominator (balance.py:97-101)
    const float denom = (float)B * (float)H * (float)W;
    const float coef = loss_scale / denom / (float)nheads;
    if (C <= 8)
        loss_gather_kernel<8, 0><<<cells, 256, 0, st>>>(logits1, logits2, label, nullptr, pixel_weight, dlogits1, dlogits2,
                                                        partial, C, h, w, H, W, 0.f, 0.f, 0.f, ignore_label, coef);
    else
        loss_gather_kernel<16, 0><<<cells, 256, 0, st>>>(logits1, logits2, label, nullptr, pixel_weight, dlogits1, dlogits2,
                                                         partial, C, h, w, H, W, 0.f, 0.f, 0.f, ignore_label, coef);
    loss_finalize_kernel<<<1, 256, 0, st>>>(partial, cells, 0, nheads, denom, loss_out, nullptr);
    return uem_check_launch("ce_upsampled");
}

extern "C" int uem_uvem_upsampled(const float* logits1, const float* logits2, const int64_t* hard, const float* soft,
                                  const float* pixel_weight, float* loss_out, float* dlogits1, float* dlogits2,
                                  float* partial, int B, int C, int h, int w, int H, int W, float m, float t, float gamma,
                                  int64_t ignore_label, float loss_scale, void* stream) {
    UEM_REQUIRE(logits1 && hard && soft && loss_out && dlogits1 && partial, "uvem_upsampled: null pointer");
    UEM_REQUIRE(!logits2 || dlogits2, "uvem_upsampled: dlogits2 required with logits2");
    UEM_REQUIRE(B > 0 && C >= 1 && C <= UEM_MAX_CLASSES && h > 0 && w > 0 && H >= h && W >= w, "uvem_upsampled: bad shape");
    UEM_REQUIRE(gamma > 0.f && t > 0.f, "uvem_upsampled: bad hyper-parameters");
    hipStream_t st = (hipStream_t)stream;
    const int cells = B * h * w;
    const int nheads = logits2 ? 2 : 1;
    const float coef = loss_scale / (float)nheads;       // 1/(valid+eps) applied after the count is known
    if (C <= 8)
        loss_gather_kernel<8, 1><<<cells, 256, 0, st>>>(logits1, logits2, hard, soft, pixel_weight, dlogits1, dlogits2,
                                                        partial, C, h, w, H, W, m, t, 1.0f / gamma, ignore_label, coef);
    else
        loss_gather_kernel<16, 1><<<cells, 256, 0, st>>>(logits1, logits2, hard, soft, pixel_weight, dlogits1, dlogits2,
                                                         partial, C, h, w, H, W, m, t, 1.0f / gamma, ignore_label, coef);
    float* inv_valid = partial + (size_t)cells * 4;      // one extra float behind the partials
    loss_finalize_kernel<<<1, 256, 0, st>>>(partial, cells, 1, nheads, 1.f, loss_out, inv_valid);
    const int64_t n = (int64_t)cells * C;
    scale_by_device_scalar_kernel<<<(int)uem_cdiv(n, 256), 256, 0, st>>>(dlogits1, logits2 ? dlogits2 : nullptr, n, inv_valid);
    return uem_check_launch("uvem_upsampled");
}

// ---------------------------------------------------------------------------------------------------------
// eval-mode network output (Encoder.py:153-155): prob = (softmax(up(x1)) + softmax(up(x2))) / 2  -> NCHW
// ---------------------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void upsample_softmax_avg_kernel(const float* __restrict__ lg1, const float* __restrict__ lg2,
                                                                   float* __restrict__ prob, int C, int h, int w, int H, int W) {
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= plane) return;
    const int Y = (int)(p / W), X = (int)(p % W);
    const Lerp ly = lerp_setup(Y, h, H, true), lx = lerp_setup(X, w, W, true);
    float v[CMAX], u[CMAX], ce;
    up_logits<CMAX>(lg1 + (size_t)b * h * w * C, C, w, ly, lx, v);
    softmax_ce<CMAX>(v, C, -1, ce);
    if (lg2) {
        up_logits<CMAX>(lg2 + (size_t)b * h * w * C, C, w, ly, lx, u);
        softmax_ce<CMAX>(u, C, -1, ce);
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) v[c] = (v[c] + u[c]) / 2.0f;
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) prob[((size_t)b * C + c) * plane + p] = v[c];
}
extern "C" int uem_upsample_softmax_avg(const float* logits1, const float* logits2, float* prob, int B, int C, int h, int w,
                                        int H, int W, void* stream) {
    UEM_REQUIRE(logits1 && prob && B > 0 && C >= 1 && C <= UEM_MAX_CLASSES, "upsample_softmax_avg: bad arguments");
    dim3 grid((unsigned)uem_cdiv((int64_t)H * W, 256), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (C <= 8) upsample_softmax_avg_kernel<8><<<grid, 256, 0, st>>>(logits1, logits2, prob, C, h, w, H, W);
    else upsample_softmax_avg_kernel<16><<<grid, 256, 0, st>>>(logits1, logits2, prob, C, h, w, H, W);
    return uem_check_launch("upsample_softmax_avg");
}

// ---------------------------------------------------------------------------------------------------------
// small helpers of the loss surface
// ---------------------------------------------------------------------------------------------------------
__global__ void uvem_weight_kernel(const float* __restrict__ u, float* __restrict__ wgt, int64_t n, float m, float t, float ig) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wgt[i] = uvem_weight_dev(u[i], m, t, ig);
}
extern "C" int uem_uvem_weight(const float* u, float* w, int64_t n, float m, float t, float gamma, void* stream) {
    UEM_REQUIRE(u && w && n > 0 && gamma > 0.f, "uvem_weight: bad arguments");
    uvem_weight_kernel<<<(int)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(u, w, n, m, t, 1.0f / gamma);
    return uem_check_launch("uvem_weight");
}

__global__ __launch_bounds__(256) void class_count_kernel(const int64_t* __restrict__ label, int64_t n, int C, int64_t ignore,
                                                          float* __restrict__ counts) {
    __shared__ int hist[UEM_MAX_CLASSES + 1];
    if (threadIdx.x <= C) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t v = label[i];
        const int bin = (v == ignore || v < 0 || v >= C) ? C : (int)v;
        atomicAdd(&hist[bin], 1);
    }
    __syncthreads();
    if (threadIdx.x <= C && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (float)hist[threadIdx.x]);
}
extern "C" int uem_class_count(const int64_t* label, int64_t n, int C, int64_t ignore_label, float* counts, void* stream) {
    UEM_REQUIRE(label && counts && n > 0 && C >= 1 && C <= UEM_MAX_CLASSES, "class_count: bad arguments");
    class_count_kernel<<<uem_stream_grid(n, 256 * 8), 256, 0, (hipStream_t)stream>>>(label, n, C, ignore_label, counts);
    return uem_check_launch("class_count");
}
__global__ void class_weight_gather_kernel(const int64_t* __restrict__ label, const float* __restrict__ cw,
                                           float* __restrict__ out, int64_t n, int C, int64_t ignore) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t v = label[i];
        out[i] = (v == ignore || v < 0 || v >= C) ? 0.f : cw[v];
    }
}
extern "C" int uem_class_weight_gather(const int64_t* label, const float* class_w, float* out, int64_t n, int C,
                                       int64_t ignore_label, void* stream) {
    UEM_REQUIRE(label && class_w && out && n > 0, "class_weight_gather: bad arguments");
    class_weight_gather_kernel<<<(int)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(label, class_w, out, n, C, ignore_label);
    return uem_check_launch("class_weight_gather");
}
