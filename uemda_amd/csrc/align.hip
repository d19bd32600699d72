// Stage-2 alignment losses, forward and backward fused:
//   PrototypeContrastiveLoss  uemda/loss.py:10-47      (L2-normalise, feat . Proto^T / T, cross-entropy)
//   CoralLoss                 uemda/gast/coral.py:15-47 (Frobenius distance of the two feature covariances;
//                             the 2048x2048 Gram matrices and the two backward GEMMs run on the MFMA conv kernels)
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// PCL
// ---------------------------------------------------------------------------------------------------------
__global__ void pcl_proto_normalize_kernel(const float* __restrict__ protos, float* __restrict__ pn, int k) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* p = protos + (size_t)c * k;
    float ss = 0.f;
    for (int j = lane; j < k; j += 64) ss += p[j] * p[j];
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    for (int j = lane; j < k; j += 64) pn[(size_t)c * k + j] = p[j] * inv;
}
__global__ __launch_bounds__(256) void pcl_count_kernel(const int64_t* __restrict__ lab, int n, int C, int64_t ignore, float* __restrict__ cnt) {
    // single block: number of rows with a usable label -> cnt[0] (deterministic)
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const int64_t l = lab[i]; s += (l != ignore && l >= 0 && l < C) ? 1.f : 0.f; }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) cnt[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
template <int CMAX>
__global__ __launch_bounds__(256) void pcl_kernel(const float* __restrict__ feat, const float* __restrict__ pn_g,
                                                  const int64_t* __restrict__ lab, const float* __restrict__ cnt,
                                                  float* __restrict__ dfeat, float* __restrict__ partial, int n, int k, int C,
                                                  float inv_temp, int64_t ignore) {
    extern __shared__ __attribute__((aligned(16))) float pn[];       // [C][k] normalised prototypes
    const int tid = threadIdx.x;
    for (int i = tid; i < (C * k) >> 2; i += 256) reinterpret_cast<float4*>(pn)[i] = reinterpret_cast<const float4*>(pn_g)[i];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const float invN = 1.0f / cnt[0];
    float loss_acc = 0.f;
    for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
        const float* x = feat + (size_t)r * k;
        float* dx = dfeat + (size_t)r * k;
        const int64_t l64 = lab[r];
        const bool valid = (l64 != ignore) && l64 >= 0 && l64 < C;
        if (!valid) {                                                // masked out of the loss: zero gradient row
            for (int j = lane * 4; j < k; j += 256) *reinterpret_cast<float4*>(dx + j) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float ss = 0.f, d[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) d[c] = 0.f;
        for (int j = lane * 4; j < k; j += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x + j);
            ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) {
                    const float4 p = *reinterpret_cast<const float4*>(pn + (size_t)c * k + j);
                    d[c] += a.x * p.x + a.y * p.y + a.z * p.z + a.w * p.w;
                }
        }
        ss = wave_sum(ss);
        const float inv_norm = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        float logit[CMAX], m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) { logit[c] = wave_sum(d[c]) * inv_norm * inv_temp; m = fmaxf(m, logit[c]); }
        float se = 0.f, ly = 0.f;
        const int y = (int)l64;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) { if (c == y) ly = logit[c]; se += expf(logit[c] - m); }
        if (lane == 0) loss_acc += -(ly - m - logf(se));
        // g_c = (softmax_c - onehot_c)/N ; v = sum_c g_c p^_c / T ; s = f^.v = sum_c g_c logit_c ; df = (v - f^ s)/||f||
        float g[CMAX], s = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) { g[c] = (expf(logit[c] - m) / se - (c == y ? 1.f : 0.f)) * invN; s += g[c] * logit[c]; }
        for (int j = lane * 4; j < k; j += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x + j);         // L1/L2 hit
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) {
                    const float4 p = *reinterpret_cast<const float4*>(pn + (size_t)c * k + j);
                    const float gc = g[c] * inv_temp;
                    v.x += gc * p.x; v.y += gc * p.y; v.z += gc * p.z; v.w += gc * p.w;
                }
            const float fs = inv_norm * s;
            v.x = (v.x - a.x * fs) * inv_norm; v.y = (v.y - a.y * fs) * inv_norm;
            v.z = (v.z - a.z * fs) * inv_norm; v.w = (v.w - a.w * fs) * inv_norm;
            *reinterpret_cast<float4*>(dx + j) = v;
        }
    }
    __shared__ float red[4];
    if (lane == 0) red[wave] = loss_acc;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void pcl_finalize_kernel(const float* __restrict__ partial, int nb, const float* __restrict__ cnt,
                                                           float* __restrict__ loss) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)cnt[0]);
}
#define UEM_PCL_BLOCKS 768
extern "C" int64_t uem_pcl_workspace_floats(int k, int C) { return (int64_t)C * k + UEM_PCL_BLOCKS + 8; }
extern "C" int uem_pcl_loss(const float* protos, const float* feat, const int64_t* labels, float* loss_out, float* dfeat,
                            float* workspace, int n, int k, int C, float temperature, int64_t ignore_label, void* stream) {
    UEM_REQUIRE(protos && feat && labels && loss_out && dfeat && workspace, "pcl_loss: null pointer");
    UEM_REQUIRE(n > 0 && k >= 8 && (k % 4) == 0 && C >= 1 && C <= UEM_MAX_CLASSES && temperature > 0.f, "pcl_loss: bad arguments");
    UEM_REQUIRE((size_t)C * k * 4 <= 150 * 1024, "pcl_loss: C*k too large for LDS");
    hipStream_t st = (hipStream_t)stream;
    float* pn = workspace;
    float* partial = workspace + (size_t)C * k;
    float* cnt = partial + UEM_PCL_BLOCKS;
    pcl_proto_normalize_kernel<<<C, 64, 0, st>>>(protos, pn, k);
    pcl_count_kernel<<<1, 256, 0, st>>>(labels, n, C, ignore_label, cnt);
    int grid = (int)uem_cdiv(n, 4);
    if (grid > UEM_PCL_BLOCKS) grid = UEM_PCL_BLOCKS;
    const size_t lds = (size_t)C * k * sizeof(float);
    if (C <= 8) {
        if (uem_allow_lds((const void*)pcl_kernel<8>, lds)) pcl_kernel<8><<<grid, 256, lds, st>>>(feat, pn, labels, cnt, dfeat, partial, n, k, C, 1.0f / temperature, ignore_label);
    } else {
        if (uem_allow_lds((const void*)pcl_kernel<16>, lds)) pcl_kernel<16><<<grid, 256, lds, st>>>(feat, pn, labels, cnt, dfeat, partial, n, k, C, 1.0f / temperature, ignore_label);
    }
    pcl_finalize_kernel<<<1, 256, 0, st>>>(partial, grid, cnt, loss_out);
    return uem_check_launch("pcl_loss");
}

// ---------------------------------------------------------------------------------------------------------
// CORAL: from the raw Gram matrices S = X^T X (computed by uem_conv2d_wgrad with x = dy = features) and the column
// means, form the covariances, the loss and the two (pre-scaled) matrices of the backward GEMMs
//   xc = (S_s - ns mu_s mu_s^T)/(ns-1), loss = sum (xc - xct)^2 / (4 d^2),  G = 2 (xc - xct)/(4 d^2)
//   d source = (source - mu_s) . [ 2 G/(ns-1)],   d target = (target - mu_t) . [-2 G/(nt-1)]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coral_finish_kernel(const float* __restrict__ Ss, const float* __restrict__ St,
                                                           const float* __restrict__ mus, const float* __restrict__ mut,
                                                           float ns, float nt, int d, float* __restrict__ Gs,
                                                           float* __restrict__ Gt, float* __restrict__ partial) {
    const int64_t total = (int64_t)d * d;
    float acc = 0.f;
    const float k4 = 1.0f / (4.0f * (float)d * (float)d);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int a = (int)(i / d), b = (int)(i % d);
        const float xc = (Ss[i] - ns * mus[a] * mus[b]) / (ns - 1.f);
        const float xt = (St[i] - nt * mut[a] * mut[b]) / (nt - 1.f);
        const float df = xc - xt;
        acc += df * df;
        const float g = 2.f * df * k4;
        Gs[i] = g * (2.f / (ns - 1.f));
        Gt[i] = -g * (2.f / (nt - 1.f));
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void coral_loss_kernel(const float* __restrict__ partial, int nb, int d, float* __restrict__ loss) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (4.0 * (double)d * (double)d));
}
#define UEM_CORAL_BLOCKS 1024
extern "C" int uem_coral_finish(const float* gram_s, const float* gram_t, const float* mean_s, const float* mean_t, int ns,
                                int nt, int d, float* g_s, float* g_t, float* loss_out, float* partial /* >= 1024 floats */,
                                void* stream) {
    UEM_REQUIRE(gram_s && gram_t && mean_s && mean_t && g_s && g_t && loss_out && partial, "coral_finish: null pointer");
    UEM_REQUIRE(ns > 1 && nt > 1 && d > 0, "coral_finish: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int grid = (int)uem_cdiv((int64_t)d * d, 256 * 8);
    if (grid > UEM_CORAL_BLOCKS) grid = UEM_CORAL_BLOCKS;
    coral_finish_kernel<<<grid, 256, 0, st>>>(gram_s, gram_t, mean_s, mean_t, (float)ns, (float)nt, d, g_s, g_t, partial);
    coral_loss_kernel<<<1, 256, 0, st>>>(partial, grid, d, loss_out);
    return uem_check_launch("coral_finish");
}
__global__ void neg_copy_kernel(const float* __restrict__ a, float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = -a[i];
}
extern "C" int uem_negate(const float* a, float* b, int n, void* stream) {
    UEM_REQUIRE(a && b && n > 0, "negate: bad arguments");
    neg_copy_kernel<<<(int)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n);
    return uem_check_launch("negate");
}
