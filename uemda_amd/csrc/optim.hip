// clip_grad_norm_(max_norm, L2) + SGD(momentum, weight_decay) fused over the flat parameter arena.
// Reference: tools/train_ssl_uem.py:169-170,228-232 (torch.optim.SGD + clip_grad.clip_grad_norm_).
#include "common.h"

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    double s = 0.0;
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (int64_t i = (nvec << 2) + threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
}
extern "C" int uem_grad_sqnorm(const float* grad, int64_t n, float* partial, float* norm_out, void* stream) {
    UEM_REQUIRE(grad && partial && norm_out && n > 0, "grad_sqnorm: bad arguments");
    UEM_REQUIRE(((uintptr_t)grad & 15) == 0, "grad_sqnorm: grad must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)uem_cdiv(n / 4 + 1, 256 * 4);
    if (nb > UEM_NORM_BLOCKS) nb = UEM_NORM_BLOCKS;
    sqnorm_partial_kernel<<<nb, 256, 0, st>>>(grad, n, partial);
    sqnorm_final_kernel<<<1, 256, 0, st>>>(partial, nb, norm_out);
    return uem_check_launch("grad_sqnorm");
}

__global__ __launch_bounds__(256) void sgd_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                       int64_t n, const float* __restrict__ norm, float max_norm, float lr,
                                                       float momentum, float wd, int first, float prescale,
                                                       const float* __restrict__ lr_dev) {
    // lr_dev (optional): the learning rate as a DEVICE scalar -- a step captured in a hipGraph bakes by-value arguments in, the
    // schedule then writes this word before every replay
    if (lr_dev) lr = lr_dev[0];
    // clip coefficient exactly as torch.nn.utils.clip_grad_norm_: min(max_norm / (total_norm + 1e-6), 1)
    float coef = 1.0f;
    if (norm) { coef = max_norm / (norm[0] * prescale + 1e-6f); coef = coef > 1.0f ? 1.0f : coef; }
    coef *= prescale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        g[i] = gi;                                        // clip_grad_norm_ scales .grad in place
        const float d = gi + wd * p[i];
        const float b = first ? d : momentum * buf[i] + d;
        buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}
extern "C" int uem_sgd_clip_step(float* param, float* grad, float* momentum_buf, int64_t n, const float* norm, float max_norm,
                                 float lr, float momentum, float weight_decay, int first_step, float grad_prescale,
                                 const float* lr_dev, void* stream) {
    UEM_REQUIRE(param && grad && momentum_buf && n > 0, "sgd_clip_step: bad arguments");
    sgd_clip_kernel<<<uem_stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(param, grad, momentum_buf, n, norm, max_norm, lr,
                                                                             momentum, weight_decay, first_step, grad_prescale, lr_dev);
    return uem_check_launch("sgd_clip_step");
}
