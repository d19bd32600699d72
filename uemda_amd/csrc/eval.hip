// Kernels for the rows either side of the training step: sliding-window / TTA accumulation
// (uemda/utils/tools.py:61-97,132-152), argmax + confusion matrix for evaluation (uemda/utils/eval.py:41-50),
// prototype initialisation (uemda/gast/alignment.py:121-122).
#include "common.h"

__global__ void window_accumulate_kernel(float* __restrict__ dst, float* __restrict__ cnt, const float* __restrict__ src,
                                         int B, int C, int H, int W, int y1, int x1, int th, int tw, int sh, int sw) {
    // dst[b,c,y1+i,x1+j] += src[b,c,i,j] for i<th, j<tw (src has row stride sw and plane sh*sw: it may be padded)
    const int64_t total = (int64_t)B * C * th * tw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % tw);
        int64_t t = i / tw;
        const int y = (int)(t % th); t /= th;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        dst[(((size_t)b * C + c) * H + y1 + y) * W + x1 + x] += src[(((size_t)b * C + c) * sh + y) * sw + x];
        if (c == 0) cnt[((size_t)b * H + y1 + y) * W + x1 + x] += 1.0f;
    }
}
extern "C" int uem_window_accumulate(float* dst, float* cnt, const float* src, int B, int C, int H, int W, int y1, int x1,
                                     int th, int tw, int src_h, int src_w, void* stream) {
    UEM_REQUIRE(dst && cnt && src, "window_accumulate: null pointer");
    UEM_REQUIRE(B > 0 && C > 0 && y1 >= 0 && x1 >= 0 && th > 0 && tw > 0 && y1 + th <= H && x1 + tw <= W && th <= src_h && tw <= src_w,
                "window_accumulate: window outside the image");
    const int64_t total = (int64_t)B * C * th * tw;
    window_accumulate_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(dst, cnt, src, B, C, H, W, y1, x1, th, tw, src_h, src_w);
    return uem_check_launch("window_accumulate");
}
__global__ void window_normalize_kernel(float* __restrict__ dst, const float* __restrict__ cnt, int C, int64_t HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % HW, b = i / (HW * C);
        dst[i] = dst[i] / cnt[b * HW + p];
    }
}
extern "C" int uem_window_normalize(float* dst, const float* cnt, int B, int C, int H, int W, void* stream) {
    UEM_REQUIRE(dst && cnt && B > 0 && C > 0 && H > 0 && W > 0, "window_normalize: bad arguments");
    const int64_t HW = (int64_t)H * W, total = HW * C * B;
    window_normalize_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(dst, cnt, C, HW, total);
    return uem_check_launch("window_normalize");
}
__global__ void scale_kernel(float* __restrict__ a, int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] *= s;
}
extern "C" int uem_scale(float* a, int64_t n, float s, void* stream) {
    UEM_REQUIRE(a && n > 0, "scale: bad arguments");
    scale_kernel<<<uem_stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(a, n, s);
    return uem_check_launch("scale");
}

// argmax over the class planes of an NCHW map (first maximum wins, like torch.argmax on CPU) and, when gt is
// given, the confusion matrix cm[gt][pred] over pixels with 0 <= gt < C
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float* __restrict__ prob, const int64_t* __restrict__ gt,
                                                               int64_t* __restrict__ pred_out, unsigned long long* __restrict__ cm,
                                                               int C, int64_t HW) {
    extern __shared__ unsigned int hist[];     // C*C block-local counts
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256) {
        const float* pl = prob + (size_t)b * C * HW + p;
        float best = pl[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = pl[(size_t)c * HW];
            if (v > best) { best = v; arg = c; }
        }
        if (pred_out) pred_out[(size_t)b * HW + p] = arg;
        if (gt) {
            const int64_t g = gt[(size_t)b * HW + p];
            if (g >= 0 && g < C) atomicAdd(&hist[(int)g * C + arg], 1u);
        }
    }
    __syncthreads();
    if (cm)
        for (int i = threadIdx.x; i < C * C; i += 256)
            if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}
extern "C" int uem_argmax_confusion(const float* prob, const int64_t* gt, int64_t* pred, int64_t* cm, int B, int C, int64_t HW,
                                    void* stream) {
    UEM_REQUIRE(prob && (pred || (gt && cm)), "argmax_confusion: nothing to compute");
    UEM_REQUIRE(B > 0 && C >= 1 && C <= 64 && HW > 0, "argmax_confusion: bad shape");
    int chunks = (int)uem_cdiv(HW, 256 * 8);
    if (chunks > 256) chunks = 256;
    argmax_confusion_kernel<<<dim3(chunks, B), 256, sizeof(unsigned int) * C * C, (hipStream_t)stream>>>(
        prob, gt, pred, (unsigned long long*)cm, C, HW);
    return uem_check_launch("argmax_confusion");
}

__global__ void proto_mean_kernel(const float* __restrict__ sums, const float* __restrict__ counts, float* __restrict__ protos, int k, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * k) protos[i] = sums[i] / (counts[i / k] + 1e-7f);
}
extern "C" int uem_proto_mean(const float* sums, const float* counts, float* protos, int k, int C, void* stream) {
    UEM_REQUIRE(sums && counts && protos && k > 0 && C > 0, "proto_mean: bad arguments");
    proto_mean_kernel<<<(int)uem_cdiv((int64_t)C * k, 256), 256, 0, (hipStream_t)stream>>>(sums, counts, protos, k, C);
    return uem_check_launch("proto_mean");
}
