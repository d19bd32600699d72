// Fused "bilinear upsample (align_corners=True) + per-pixel loss" kernels, forward AND backward in
// one pass: one block per band of full-resolution rows that interpolate between the same two
// low-resolution logit rows; every pixel is evaluated once, the gradient of the band's two logit rows
// is reduced in registers + LDS, and the (B,C,H,W) upsampled logits are never materialised.
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
    for (int c = 0; c < CMAX; ++c) if (c < C) { if (c == label) vl = v[c]; v[c] = fast_exp(v[c] - m); s += v[c]; }
    const float inv = fast_rcp(s);
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) v[c] *= inv;
    ce = (label >= 0) ? -(vl - m - fast_log(s)) : 0.f;
}

__device__ __forceinline__ float uvem_weight_dev(float u, float m, float t, float inv_gamma) {
    // UVEMLoss.get_weight (balance.py:396-423)
    float left = 1.0f;
    if (m > 0.f) {
        float x = (u <= m && u >= 0.f) ? u : 1.0f;
        float q = (-1.0f / (m * m)) * ((x - m) * (x - m)) + 1.0f;
        q = fminf(fmaxf(q, 0.f), 1.f);
        left = fast_pow01(q, inv_gamma);
    }
    float right = 0.f;
    if (m < t) {
        float x = (u > m && u <= t) ? u : 0.f;
        float q = (-1.0f / ((t - m) * (t - m))) * ((x - m) * (x - m)) + 1.0f;
        q = fminf(fmaxf(q, 0.f), 1.f);
        right = fast_pow01(q, inv_gamma);
    }
    float wgt = (u <= m) ? left : right;
    return (u >= t) ? 0.f : wgt;
}

// MODE 0: CE (mean over all pixels), MODE 1: UVEM.
// One block per (image, low-resolution row `cy`): it owns the BAND of full-resolution rows whose upper
// interpolation row is cy (~(H-1)/(h-1) rows), so every pixel is evaluated exactly once, reading label / soft-label
// rows coalesced.  A thread walks one pixel column down the band; its pixels all interpolate between the same four
// logit cells (rows cy, cy+1 x columns i0x, i1x), so the gradient is summed in registers over the column (y weights
// applied per pixel), then added with the two x weights into a per-wave LDS image of the band's two logit rows
// (LDS atomics; per-wave images summed in fixed order).  Band cy's second row and band cy+1's first row are the
// same logit row: the bands' images go to the workspace and loss_band_combine_kernel adds the pair.
// Round 5.  (a) ONE HEAD PER THREAD: a block is 256 threads per head (512 for the two-head calls of the training step), thread
// group hd = tid / 256 owns head hd of every pixel column; the live state halves (~64 VGPRs, eight waves per SIMD instead of four).
// The label / soft-label fetch and the UVEM weight are computed by both groups (the second read is an L2 hit).
// (b) NO LDS ATOMICS.  Round 4 added each column's 24 gradient sums into per-wave LDS images of the two logit rows with ds_add_f32:
// the ~16 columns of a logit cell sit in adjacent lanes, every such instruction carried four addresses with sixteen lanes each, and
// the LDS serialised them -- an ablation with the atomics compiled out ran the band in 71 us (CE) / 101 us (UVEM) against 196 / 206
// with them: more than half of the kernel.  Now a pass's column sums go to LDS column-major, [value][column], and after one barrier
// each (row, cell, class) output is summed by ONE thread over the ~33 columns that touch its cell, in ascending column order (still
// deterministic), straight into the band image.
#define LOSS_COLS 256                    // pixel columns per pass (threads per head)
#define LOSS_WAVES (LOSS_COLS / 64)      // waves per head
#define LOSS_CPAD (LOSS_COLS + 1)        // row pitch of the column-major sums (value j, column x -> bank (j + x) mod 32)
static inline size_t loss_lds_floats(int w, int cmax, int nheads) {
    // low [2 heads][2 rows][w][cmax] + column sums [nheads][2 cmax][LOSS_CPAD] + per-column l1 [LOSS_COLS] + band image [nheads][2][w][cmax]
    // + first column of each cell [w + 1]
    return (size_t)4 * w * cmax + (size_t)nheads * 2 * cmax * LOSS_CPAD + LOSS_COLS + (size_t)nheads * 2 * w * cmax + (size_t)w + 1;
}
// first column X in [lo, hi) whose upper-left logit cell min((int)(sx * X), w - 1) is >= x (hi if none): the cells are monotone in X
__device__ __forceinline__ int loss_first_col(int x, int w, float sx, int lo, int hi) {
    if (x <= 0) return lo;
    if (!(sx > 0.f)) return hi;
    int g = (int)((float)x / sx);
    g = g < lo ? lo : (g > hi ? hi : g);
    while (g > lo && lerp_ac(g - 1, w, sx).i0 >= x) --g;
    while (g < hi && lerp_ac(g, w, sx).i0 < x) ++g;
    return g;
}
// EXACT: the class count IS CMAX (6 and 7, the reference's two datasets, have their own CMAX): the `c < C` guards of the unrolled
// per-class loops fold at compile time instead of costing a compare + select each
template <int CMAX, int MODE, bool EXACT>
__global__ __launch_bounds__(2 * LOSS_COLS, (CMAX <= 6 ? 8 : 1)) void loss_band_kernel(
    const float* __restrict__ lg1, const float* __restrict__ lg2, const int64_t* __restrict__ label,
    const float* __restrict__ soft, const float* __restrict__ pixw, float* __restrict__ band_grad,
    float* __restrict__ partial, int C_, int h, int w, int H, int W, float um, float ut, float inv_gamma, int64_t ignore) {
    const int C = EXACT ? CMAX : C_;
    extern __shared__ __attribute__((aligned(16))) float loss_sm[];
    const int nheads = lg2 ? 2 : 1;                    // blockDim.x == LOSS_COLS * nheads
    const int nthr = LOSS_COLS * nheads;
    const int nout = 2 * w * CMAX;                     // outputs per head: [2 rows][w][CMAX]
    float* low = loss_sm;                              // [2 heads][2 rows][w][CMAX]
    float* colA = low + 4 * w * CMAX;                  // [nheads][2 * CMAX][LOSS_CPAD]: this pass's column sums (band row 0 classes, then row 1)
    float* colL1 = colA + (size_t)nheads * 2 * CMAX * LOSS_CPAD;   // [LOSS_COLS]: this pass's x weights l1
    float* img = colL1 + LOSS_COLS;                    // [nheads][2 rows][w][CMAX]: the band's gradient image
    int* cstart = reinterpret_cast<int*>(img + (size_t)nheads * nout);     // [w + 1]: first column of each cell in the current pass
    const int cy = blockIdx.x % h, b = blockIdx.x / h;
    const int cy1 = cy + (cy < h - 1 ? 1 : 0);
    const size_t plane = (size_t)H * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int hd = tid / LOSS_COLS, ctid = tid % LOSS_COLS, wave = ctid >> 6;
    for (int i = tid; i < 4 * w * CMAX; i += nthr) {
        const int c = i % CMAX, x = (i / CMAX) % w, r = (i / (CMAX * w)) & 1, hh = i / (2 * CMAX * w);
        const float* src = hh ? lg2 : lg1;
        float v = 0.f;
        if (src != nullptr && c < C) v = src[(((size_t)b * h + (r ? cy1 : cy)) * w + x) * C + c];
        low[i] = v;
    }
    for (int i = tid; i < nheads * nout; i += nthr) img[i] = 0.f;
    __syncthreads();
    // candidate rows of the band (a superset; rows whose i0 differs are skipped)
    int ya = 0, yb = H - 1;
    if (h > 1 && H > 1) {
        const float inv = (float)(H - 1) / (float)(h - 1);
        ya = (int)floorf((float)cy * inv) - 1;
        yb = (int)ceilf((float)(cy + 1) * inv) + 1;
        ya = ya < 0 ? 0 : ya;
        yb = yb > H - 1 ? H - 1 : yb;
    }
    // exact row range of the band inside the candidate window
    while (ya <= yb && lerp_setup(ya, h, H, true).i0 != cy) ++ya;
    while (yb >= ya && lerp_setup(yb, h, H, true).i0 != cy) --yb;
    const float* lowh = low + (size_t)hd * 2 * w * CMAX;
    float* colAh = colA + (size_t)hd * 2 * CMAX * LOSS_CPAD;
    float* imgh = img + (size_t)hd * nout;
    const float sy = lerp_scale_ac(h, H), sx = lerp_scale_ac(w, W);     // hoisted: one division per block instead of one per pixel
    float loss = 0.f, valid = 0.f;
    struct Px { int64_t lab; float q[CMAX]; float pw; };
    for (int Xb = 0; Xb < W; Xb += LOSS_COLS) {                         // block-uniform: every thread takes every pass (barriers inside)
        const int X = Xb + ctid;
        const bool on = X < W;
        const Lerp lx = lerp_ac(on ? X : W - 1, w, sx);
        float a0[CMAX], a1[CMAX];                                       // the column's gradient sums for the band's two rows
#pragma unroll
        for (int c = 0; c < CMAX; ++c) a0[c] = a1[c] = 0.f;
        if (on) {
            const float* L0 = lowh + (size_t)lx.i0 * CMAX;
            const float* L1 = lowh + (size_t)lx.i1 * CMAX;
            // x-interpolated logits of the band's two rows (constant down the column)
            float top[CMAX], bot[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                top[c] = lx.l0 * L0[c] + lx.l1 * L1[c];
                bot[c] = lx.l0 * L0[w * CMAX + c] + lx.l1 * L1[w * CMAX + c];
            }
            auto fetch = [&](int Y, Px& px) {
                const size_t p = (size_t)b * plane + (size_t)Y * W + X;
                px.lab = label[p];
                if (MODE == 1) {
#pragma unroll
                    for (int c = 0; c < CMAX; ++c)
                        if (c < C) px.q[c] = soft[((size_t)b * C + c) * plane + (size_t)Y * W + X];
                }
                px.pw = pixw ? pixw[p] : 1.0f;
            };
            Px cur, nxt;
            if (ya <= yb) fetch(ya, cur);
            for (int Y = ya; Y <= yb; ++Y) {
                if (Y < yb) fetch(Y + 1, nxt);                            // next row's loads fly during this row's math
                const Lerp ly = lerp_ac(Y, h, sy);
                const int64_t lab64 = cur.lab;
                const bool lab_ok = (lab64 != ignore) && lab64 >= 0 && lab64 < C;
                const int lab = lab_ok ? (int)lab64 : -1;
                float pw = 1.0f;      // per-pixel coefficient on (softmax - onehot)
                if (MODE == 1) {
                    float u = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c)
                        if (c < C) { const float q = cur.q[c]; u += -q * fast_log(q); }
                    const bool gate = !(u > ut);                          // ce[u > t] = 0
                    pw = gate ? uvem_weight_dev(u, um, ut, inv_gamma) : 0.f;
                    if ((u <= ut) && lab64 != ignore) valid += 1.f;
                }
                pw *= cur.pw;
                if (!lab_ok) pw = 0.f;                                    // ignore_index: zero loss and gradient
                float v[CMAX], ce;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) v[c] = ly.l0 * top[c] + ly.l1 * bot[c];
                softmax_ce<CMAX>(v, C, lab, ce);
                loss += pw * ce;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) { const float g = pw * (v[c] - (c == lab ? 1.f : 0.f)); a0[c] += ly.l0 * g; a1[c] += ly.l1 * g; }
                cur = nxt;
            }
        }
        // ---- this pass's columns -> the band image, without atomics ----
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            colAh[(size_t)c * LOSS_CPAD + ctid] = a0[c];
            colAh[(size_t)(CMAX + c) * LOSS_CPAD + ctid] = a1[c];
        }
        if (hd == 0) colL1[ctid] = lx.l1;
        // first column of every logit cell inside this pass (cstart[w] = the pass's end): one thread per cell
        for (int i = tid; i <= w; i += nthr) cstart[i] = loss_first_col(i, w, sx, Xb, min(Xb + LOSS_COLS, W));
        __syncthreads();
        for (int o = ctid; o < nout; o += LOSS_COLS) {
            const int c = o % CMAX, x = (o / CMAX) % w, r = o / (CMAX * w);
            if (c < C) {
                const float* A = colAh + (size_t)(r * CMAX + c) * LOSS_CPAD - Xb;
                const float* L = colL1 - Xb;
                const int lo = cstart[x], hi = cstart[x + 1], lp = x > 0 ? cstart[x - 1] : lo;
                const bool last = x == w - 1;                             // last cell: i1 = i0, both x weights land here (l0 + l1)
                float t = 0.f;
                // columns of cell x - 1 (their right neighbour is x, weight l1), then the cell's own columns (weight l0): ascending
                // column order in chunks of 8 with the tail predicated off, so that a chunk's 16 LDS reads are in flight together
                for (int X0 = lp; X0 < lo; X0 += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int X2 = min(X0 + j, lo - 1);
                        const float wgt = (X0 + j < lo) ? L[X2] : 0.f;
                        t = fmaf(wgt, A[X2], t);
                    }
                }
                for (int X0 = lo; X0 < hi; X0 += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int X2 = min(X0 + j, hi - 1);
                        const float l1 = L[X2];
                        const float wgt = (X0 + j < hi) ? (last ? (1.f - l1) + l1 : 1.f - l1) : 0.f;
                        t = fmaf(wgt, A[X2], t);
                    }
                }
                imgh[o] += t;
            }
        }
        __syncthreads();
    }
    __shared__ float red[2][LOSS_WAVES][2];
    {
        const float a = wave_sum(loss), e = wave_sum(valid);
        if (lane == 0) { red[hd][wave][0] = a; red[hd][wave][1] = e; }
    }
    __syncthreads();
    float* out = band_grad + (size_t)blockIdx.x * 4 * w * CMAX;       // [2 rows][w][2 heads][CMAX]
    for (int i = tid; i < 4 * w * CMAX; i += nthr) {
        const int c = i % CMAX, hh = (i / CMAX) & 1, x = (i / (2 * CMAX)) % w, r = i / (2 * CMAX * w);
        out[i] = hh < nheads ? img[(size_t)hh * nout + ((size_t)r * w + x) * CMAX + c] : 0.f;
    }
    if (tid < 3) {                                                    // loss of head 0, loss of head 1, valid count (head 0's threads)
        const int hh = tid == 1 ? 1 : 0, f = tid == 2 ? 1 : 0;
        float t = 0.f;
        if (hh < nheads) {
#pragma unroll
            for (int k = 0; k < LOSS_WAVES; ++k) t += red[hh][k][f];
        }
        partial[(size_t)blockIdx.x * 4 + tid] = t;
    }
}

// dlogits[b][r][x][c] = coef * (*scale) * (band r's first row + band r-1's second row [+ the last band's own second row])
template <int CMAX>
__global__ void loss_band_combine_kernel(const float* __restrict__ band_grad, float* __restrict__ dl1, float* __restrict__ dl2,
                                         int B, int C, int h, int w, float coef, const float* __restrict__ scale) {
    const int64_t n = (int64_t)B * h * w * C;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C), x = (int)((i / C) % w), r = (int)((i / ((int64_t)C * w)) % h), b = (int)(i / ((int64_t)C * w * h));
    const float f = scale ? coef * scale[0] : coef;
    const size_t band = (size_t)4 * w * CMAX;
    const float* g = band_grad + ((size_t)b * h + r) * band;
    const size_t o0 = ((size_t)(0 * w + x) * 2) * CMAX + c, o1 = ((size_t)(1 * w + x) * 2) * CMAX + c;
    float s1 = g[o0], s2 = g[o0 + CMAX];
    if (r > 0) { s1 += (g - band)[o1]; s2 += (g - band)[o1 + CMAX]; }
    if (r == h - 1) { s1 += g[o1]; s2 += g[o1 + CMAX]; }            // the last band's rows cy and cy+1 coincide
    dl1[i] = s1 * f;
    if (dl2) dl2[i] = s2 * f;
}

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

static inline int loss_cmax(int C) { return C <= 4 ? 4 : (C <= 6 ? 6 : (C == 7 ? 7 : (C <= 8 ? 8 : 16))); }
extern "C" int64_t uem_loss_workspace_floats(int B, int C, int h, int w) {
    const int cmax = loss_cmax(C);
    return (int64_t)B * h * 4 * w * cmax + (int64_t)B * h * 4 + 4;
}
extern "C" int uem_scale_by_scalar(float* a, float* b, int64_t n, const float* scalar, void* stream) {
    UEM_REQUIRE(a && scalar && n > 0, "scale_by_scalar: bad arguments");
    scale_by_device_scalar_kernel<<<(int)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n, scalar);
    return uem_check_launch("scale_by_scalar");
}

template <int CMAX, int MODE>
static void loss_launch(const float* l1, const float* l2, const int64_t* label, const float* soft, const float* pixw,
                        float* loss_out, float* d1, float* d2, float* ws, int B, int C, int h, int w, int H, int W, float m,
                        float t, float inv_gamma, int64_t ignore, float coef, float ce_denominator, hipStream_t st) {
    const int bands = B * h, nheads = l2 ? 2 : 1;
    float* band_grad = ws;
    float* partial = ws + (size_t)bands * 4 * w * CMAX;
    float* inv_valid = partial + (size_t)bands * 4;
    const size_t lds = loss_lds_floats(w, CMAX, nheads) * sizeof(float);
    if (C == CMAX) {
        auto k = loss_band_kernel<CMAX, MODE, true>;
        if (!uem_allow_lds((const void*)k, lds)) return;
        k<<<bands, LOSS_COLS * nheads, lds, st>>>(l1, l2, label, soft, pixw, band_grad, partial, C, h, w, H, W, m, t, inv_gamma, ignore);
    } else {
        auto k = loss_band_kernel<CMAX, MODE, false>;
        if (!uem_allow_lds((const void*)k, lds)) return;
        k<<<bands, LOSS_COLS * nheads, lds, st>>>(l1, l2, label, soft, pixw, band_grad, partial, C, h, w, H, W, m, t, inv_gamma, ignore);
    }
    loss_finalize_kernel<<<1, 256, 0, st>>>(partial, bands, MODE, nheads, ce_denominator, loss_out, MODE ? inv_valid : nullptr);
    const int64_t n = (int64_t)B * h * w * C;
    loss_band_combine_kernel<CMAX><<<(int)uem_cdiv(n, 256), 256, 0, st>>>(band_grad, d1, d2, B, C, h, w, coef,
                                                                          MODE ? inv_valid : nullptr);
}

extern "C" int uem_ce_upsampled(const float* logits1, const float* logits2, const int64_t* label,
                                const float* pixel_weight, float* loss_out, float* dlogits1, float* dlogits2,
                                float* workspace, int B, int C, int h, int w, int H, int W, int64_t ignore_label,
                                float loss_scale, void* stream) {
    UEM_REQUIRE(logits1 && label && loss_out && dlogits1 && workspace, "ce_upsampled: null pointer");
    UEM_REQUIRE(!logits2 || dlogits2, "ce_upsampled: dlogits2 required with logits2");
    UEM_REQUIRE(B > 0 && C >= 1 && C <= UEM_MAX_CLASSES && h > 0 && w > 0 && H >= h && W >= w, "ce_upsampled: bad shape");
    UEM_REQUIRE(loss_lds_floats(w, loss_cmax(C), logits2 ? 2 : 1) * sizeof(float) <= 150 * 1024, "ce_upsampled: low-resolution width %d too large", w);
    hipStream_t st = (hipStream_t)stream;
    const int nheads = logits2 ? 2 : 1;
    // mean over ALL pixels, ignored ones included in the denominator (balance.py:97-101)
    const float denom = (float)B * (float)H * (float)W;
    const float coef = loss_scale / denom / (float)nheads;
#define CE_GO(CM) loss_launch<CM, 0>(logits1, logits2, label, nullptr, pixel_weight, loss_out, dlogits1, dlogits2, workspace, B, C, \
                                     h, w, H, W, 0.f, 0.f, 0.f, ignore_label, coef, denom, st)
    switch (loss_cmax(C)) { case 4: CE_GO(4); break; case 6: CE_GO(6); break; case 7: CE_GO(7); break; case 8: CE_GO(8); break; default: CE_GO(16); }
#undef CE_GO
    return uem_check_launch("ce_upsampled");
}

extern "C" int uem_uvem_upsampled(const float* logits1, const float* logits2, const int64_t* hard, const float* soft,
                                  const float* pixel_weight, float* loss_out, float* dlogits1, float* dlogits2,
                                  float* workspace, int B, int C, int h, int w, int H, int W, float m, float t, float gamma,
                                  int64_t ignore_label, float loss_scale, void* stream) {
    UEM_REQUIRE(logits1 && hard && soft && loss_out && dlogits1 && workspace, "uvem_upsampled: null pointer");
    UEM_REQUIRE(!logits2 || dlogits2, "uvem_upsampled: dlogits2 required with logits2");
    UEM_REQUIRE(B > 0 && C >= 1 && C <= UEM_MAX_CLASSES && h > 0 && w > 0 && H >= h && W >= w, "uvem_upsampled: bad shape");
    UEM_REQUIRE(gamma > 0.f && t > 0.f, "uvem_upsampled: bad hyper-parameters");
    UEM_REQUIRE(loss_lds_floats(w, loss_cmax(C), logits2 ? 2 : 1) * sizeof(float) <= 150 * 1024, "uvem_upsampled: low-resolution width %d too large", w);
    hipStream_t st = (hipStream_t)stream;
    const int nheads = logits2 ? 2 : 1;
    const float coef = loss_scale / (float)nheads;       // 1/(valid+eps) is applied by the combine pass, once the count is known
#define UV_GO(CM) loss_launch<CM, 1>(logits1, logits2, hard, soft, pixel_weight, loss_out, dlogits1, dlogits2, workspace, B, C, h, w, \
                                     H, W, m, t, 1.0f / gamma, ignore_label, coef, 1.f, st)
    switch (loss_cmax(C)) { case 4: UV_GO(4); break; case 6: UV_GO(6); break; case 7: UV_GO(7); break; case 8: UV_GO(8); break; default: UV_GO(16); }
#undef UV_GO
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
