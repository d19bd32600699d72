// Pseudo-label mining kernels (HBM-bound): Pearson similarity, superpixel segment-max, fused
// three-view label refinement, threshold selection, label downscale, prototype sums / EMA.
// Reference call sites: uemda/gast/alignment.py:194-293,328-355,424-451,484-509,
// uemda/gast/pseudo_generation.py:59-93.  See include/uemda_hip.h for the per-entry citations.
#include "common.h"

// ================================================================================================
// Pearson similarity  sim[n][c] = 1 / (0.5 * (1 - cov/(k-1+eps) / (std_x*std_p + eps)))
// One wave owns TWO feature rows at a time (k floats each, NHWC => contiguous); centred
// prototypes live in LDS (C*k floats) and each LDS read feeds both rows.
// ================================================================================================
__global__ void proto_center_kernel(const float* __restrict__ protos, float* __restrict__ pc,
                                    float* __restrict__ pstd, int k) {
    // one wave per class: pc = p - mean(p); pstd = unbiased std
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    const float* p = protos + (size_t)c * k;
    float s = 0.f;
    for (int j = lane; j < k; j += 64) s += p[j];
    const float mean = wave_sum(s) / (float)k;
    float ss = 0.f;
    for (int j = lane; j < k; j += 64) {
        float d = p[j] - mean;
        pc[(size_t)c * k + j] = d;
        ss += d * d;
    }
    ss = wave_sum(ss);
    if (lane == 0) pstd[c] = sqrtf(ss / (float)(k - 1));
}

template <int CMAX, bool INVERT>
__global__ __launch_bounds__(256) void pearson_kernel(const float* __restrict__ feat,
                                                      const float* __restrict__ pc_g,
                                                      const float* __restrict__ pstd,
                                                      float* __restrict__ out, int n, int k, int C) {
    extern __shared__ __attribute__((aligned(16))) float pc[];   // [C][k]
    const int tid = threadIdx.x;
    const int nvec = (C * k) >> 2;
    for (int i = tid; i < nvec; i += 256)
        reinterpret_cast<float4*>(pc)[i] = reinterpret_cast<const float4*>(pc_g)[i];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const float eps = 1e-7f;
    for (int r0 = (blockIdx.x * 4 + wave) * 2; r0 < n; r0 += gridDim.x * 8) {
        const bool has1 = (r0 + 1) < n;
        const float* x0 = feat + (size_t)r0 * k;
        const float* x1 = feat + (size_t)(has1 ? r0 + 1 : r0) * k;
        float s0 = 0.f, s1 = 0.f;
        for (int j = lane * 4; j < k; j += 256) {
            float4 a = *reinterpret_cast<const float4*>(x0 + j);
            float4 b = *reinterpret_cast<const float4*>(x1 + j);
            s0 += (a.x + a.y) + (a.z + a.w);
            s1 += (b.x + b.y) + (b.z + b.w);
        }
        const float m0 = wave_sum(s0) / (float)k, m1 = wave_sum(s1) / (float)k;
        float ss0 = 0.f, ss1 = 0.f;
        float d0[CMAX], d1[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) d0[c] = d1[c] = 0.f;
        for (int j = lane * 4; j < k; j += 256) {
            float4 a = *reinterpret_cast<const float4*>(x0 + j);   // second pass: L1/L2 hit
            float4 b = *reinterpret_cast<const float4*>(x1 + j);
            a.x -= m0; a.y -= m0; a.z -= m0; a.w -= m0;
            b.x -= m1; b.y -= m1; b.z -= m1; b.w -= m1;
            ss0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
            ss1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < C) {
                    float4 p = *reinterpret_cast<const float4*>(pc + (size_t)c * k + j);
                    d0[c] += a.x * p.x + a.y * p.y + a.z * p.z + a.w * p.w;
                    d1[c] += b.x * p.x + b.y * p.y + b.z * p.z + b.w * p.w;
                }
            }
        }
        ss0 = wave_sum(ss0);
        ss1 = wave_sum(ss1);
        const float std0 = sqrtf(ss0 / (float)(k - 1)), std1 = sqrtf(ss1 / (float)(k - 1));
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                float c0 = wave_sum(d0[c]), c1 = wave_sum(d1[c]);
                if (lane == 0) {
                    float ps = pstd[c];
                    float dist0 = (-1.0f * (c0 / ((float)(k - 1) + eps)) / (std0 * ps + eps) + 1.0f) * 0.5f;
                    out[(size_t)r0 * C + c] = INVERT ? 1.0f / dist0 : dist0;
                    if (has1) {
                        float dist1 = (-1.0f * (c1 / ((float)(k - 1) + eps)) / (std1 * ps + eps) + 1.0f) * 0.5f;
                        out[(size_t)(r0 + 1) * C + c] = INVERT ? 1.0f / dist1 : dist1;
                    }
                }
            }
        }
    }
}

// Round 5: the row stays in REGISTERS, and the arithmetic is fused.  pearson_kernel above reads every feature row twice (the mean,
// then the centred sums) and, built like the rest of the library with -ffp-contract=off, spends ~2000 VALU instructions per pair of
// rows: every a*b + c is a multiply and an add, the C run-time guard keeps 8 classes' worth of code alive, 24 IEEE divisions close
// each pair -- 33 M wave-instructions per launch at B = 32 = 128 us of issue on 1024 SIMDs: it was VALU-bound (99-110 us), not
// bandwidth-bound, which is why keeping the rows in registers alone changed nothing (101 us).  Here: a lane keeps its KV float4 of
// each of the wave's two rows (k = 256 * KV floats: 2048 -> 32 VGPRs per row; the feature map is fetched once), the dots and the
// centred squares are explicit fmaf chains, the class count is a template argument (6, 7; 0 = run time), the wave sums run on the DPP
// path, and the two divisions per output are one hoisted reciprocal and one v_rcp_f32.  ~800 instructions per pair of rows.
// The prototype centring (proto_center_kernel: a six-block launch of 19 us in front of every call) moves into the block's prologue:
// wave w centres classes w, w + 4, ... straight into the LDS image.
template <int CMAX, int CEX, bool INVERT, int KV>
__global__ __launch_bounds__(256) void pearson_rows_kernel(const float* __restrict__ feat, const float* __restrict__ protos,
                                                           float* __restrict__ out, int n, int C_) {
    constexpr int k = 256 * KV;
    const int C = CEX > 0 ? CEX : C_;
    extern __shared__ __attribute__((aligned(16))) float pc[];   // [C][k]
    __shared__ float pstd_s[CMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float eps = 1e-7f;
    const float inv_k = 1.0f / (float)k, inv_km1 = 1.0f / (float)(k - 1), inv_km1e = 1.0f / ((float)(k - 1) + eps);
    // Two register sets: the wave's NEXT pair of rows is requested before the current pair is computed on, so each wave always has
    // 16 KB in flight (the one-set form left a wave with nothing outstanding for the ~1100 instructions of its compute phase; at
    // three waves per SIMD the HBM stream was ~3.8 TB/s).  The first pair is requested before the prologue.
    float4 a[KV], b[KV], a2[KV], b2[KV];
    const int stride = gridDim.x * 8;
    int r0 = (blockIdx.x * 4 + wave) * 2;
    auto request = [&](int r, float4 (&ra)[KV], float4 (&rb)[KV]) {
        if (r < n) {
            const float4* x0 = reinterpret_cast<const float4*>(feat + (size_t)r * k) + lane;
            const float4* x1 = reinterpret_cast<const float4*>(feat + (size_t)((r + 1) < n ? r + 1 : r) * k) + lane;
#pragma unroll
            for (int i = 0; i < KV; ++i) { ra[i] = x0[i * 64]; rb[i] = x1[i * 64]; }
        }
    };
    request(r0, a, b);
    // prologue: wave w centres classes w, w + 4, ... into the LDS image (16-byte loads, the whole prototype in flight at once)
    for (int c = wave; c < C; c += 4) {
        const float4* p = reinterpret_cast<const float4*>(protos + (size_t)c * k) + lane;
        float4 v[KV];
#pragma unroll
        for (int i = 0; i < KV; ++i) v[i] = p[i * 64];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < KV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum_dpp(s) * inv_k;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            ss = fmaf(v[i].x, v[i].x, fmaf(v[i].y, v[i].y, fmaf(v[i].z, v[i].z, fmaf(v[i].w, v[i].w, ss))));
            *reinterpret_cast<float4*>(pc + (size_t)c * k + i * 256 + lane * 4) = v[i];
        }
        ss = wave_sum_dpp(ss);
        if (lane == 0) pstd_s[c] = sqrtf(ss * inv_km1);
    }
    __syncthreads();
    auto compute = [&](const int r, float4 (&ra)[KV], float4 (&rb)[KV]) {
        const bool has1 = (r + 1) < n;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            s0 += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
            s1 += (rb[i].x + rb[i].y) + (rb[i].z + rb[i].w);
        }
        const float m0 = wave_sum_dpp(s0) * inv_k, m1 = wave_sum_dpp(s1) * inv_k;
        float ss0 = 0.f, ss1 = 0.f;
#pragma unroll
        for (int i = 0; i < KV; ++i) {                             // centred in place
            ra[i].x -= m0; ra[i].y -= m0; ra[i].z -= m0; ra[i].w -= m0;
            rb[i].x -= m1; rb[i].y -= m1; rb[i].z -= m1; rb[i].w -= m1;
            ss0 = fmaf(ra[i].x, ra[i].x, fmaf(ra[i].y, ra[i].y, fmaf(ra[i].z, ra[i].z, fmaf(ra[i].w, ra[i].w, ss0))));
            ss1 = fmaf(rb[i].x, rb[i].x, fmaf(rb[i].y, rb[i].y, fmaf(rb[i].z, rb[i].z, fmaf(rb[i].w, rb[i].w, ss1))));
        }
        ss0 = wave_sum_dpp(ss0);
        ss1 = wave_sum_dpp(ss1);
        const float std0 = sqrtf(ss0 * inv_km1), std1 = sqrtf(ss1 * inv_km1);
        float o0 = 0.f, o1 = 0.f;                                  // lane c keeps class c's two results: one store per row
        // class by class, as a REAL loop: one prototype's KV fragments against both rows.  Unrolled (either loop order) the scheduler
        // hoists all 6 x KV prototype reads to the top -- nothing orders an LDS read -- and the kernel comes out at 243-284 registers.
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int i = 0; i < KV; ++i) {
                const float4 q = *reinterpret_cast<const float4*>(pc + (size_t)c * k + i * 256 + lane * 4);
                d0 = fmaf(ra[i].x, q.x, fmaf(ra[i].y, q.y, fmaf(ra[i].z, q.z, fmaf(ra[i].w, q.w, d0))));
                d1 = fmaf(rb[i].x, q.x, fmaf(rb[i].y, q.y, fmaf(rb[i].z, q.z, fmaf(rb[i].w, q.w, d1))));
            }
            const float c0 = wave_sum_dpp(d0), c1 = wave_sum_dpp(d1);
            const float ps = pstd_s[c];
            const float dist0 = (1.0f - (c0 * inv_km1e) * fast_rcp(std0 * ps + eps)) * 0.5f;
            const float dist1 = (1.0f - (c1 * inv_km1e) * fast_rcp(std1 * ps + eps)) * 0.5f;
            if (lane == c) { o0 = INVERT ? fast_rcp(dist0) : dist0; o1 = INVERT ? fast_rcp(dist1) : dist1; }
        }
        if (lane < C) {
            out[(size_t)r * C + lane] = o0;
            if (has1) out[(size_t)(r + 1) * C + lane] = o1;
        }
    };
    for (; r0 < n; r0 += 2 * stride) {                              // two pairs per trip: set A, then set B
        request(r0 + stride, a2, b2);
        compute(r0, a, b);
        if (r0 + stride < n) {
            request(r0 + 2 * stride, a, b);
            compute(r0 + stride, a2, b2);
        }
    }
}

template <int KV>
static bool pearson_rows_launch(const float* a, const float* b, float* out, int n, int m, bool invert, hipStream_t st) {
    const size_t lds = (size_t)m * 256 * KV * sizeof(float);
    int grid = (int)uem_cdiv(n, 8);
    if (grid > 256 * 3) grid = 256 * 3;
#define LAUNCH_PR(CE, INV)                                                                               \
    do {                                                                                                 \
        if (!uem_allow_lds((const void*)pearson_rows_kernel<8, CE, INV, KV>, lds)) return false;         \
        pearson_rows_kernel<8, CE, INV, KV><<<grid, 256, lds, st>>>(a, b, out, n, m);                    \
    } while (0)
    if (m == 6) { if (invert) LAUNCH_PR(6, true); else LAUNCH_PR(6, false); }
    else if (m == 7) { if (invert) LAUNCH_PR(7, true); else LAUNCH_PR(7, false); }
    else { if (invert) LAUNCH_PR(0, true); else LAUNCH_PR(0, false); }
#undef LAUNCH_PR
    return true;
}

static int pearson_launch(const float* a, const float* b, float* out, float* scratch, int n, int m, int k,
                          bool invert, hipStream_t st) {
    // feature widths of the encoder stages (k = 256 * KV) with up to 8 classes: one pass over the features, rows in registers
    static const bool rows_on = !(getenv("UEM_PEARSON_ROWS") && atoi(getenv("UEM_PEARSON_ROWS")) == 0);
    if (rows_on && m <= 8 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0) {
        bool done = false;
        if (k == 2048) done = pearson_rows_launch<8>(a, b, out, n, m, invert, st);
        else if (k == 1024) done = pearson_rows_launch<4>(a, b, out, n, m, invert, st);
        else if (k == 512) done = pearson_rows_launch<2>(a, b, out, n, m, invert, st);
        else if (k == 256) done = pearson_rows_launch<1>(a, b, out, n, m, invert, st);
        if (done) return uem_check_launch("pearson");
    }
    // scratch: [m*k] centred b + [m] std
    float* pc = scratch;
    float* pstd = scratch + (size_t)m * k;
    proto_center_kernel<<<m, 64, 0, st>>>(b, pc, pstd, k);
    size_t lds = (size_t)m * k * sizeof(float);
    int grid = (int)uem_cdiv(n, 8);
    if (grid > 256 * 3) grid = 256 * 3;
    if (grid < 1) grid = 1;
#define LAUNCH_P(CM, INV)                                                                                \
    do {                                                                                                 \
        if (uem_allow_lds((const void*)pearson_kernel<CM, INV>, lds))                                    \
            pearson_kernel<CM, INV><<<grid, 256, lds, st>>>(a, pc, pstd, out, n, k, m);                   \
    } while (0)
    if (m <= 8) {
        if (invert) LAUNCH_P(8, true); else LAUNCH_P(8, false);
    } else {
        if (invert) LAUNCH_P(16, true); else LAUNCH_P(16, false);
    }
#undef LAUNCH_P
    return uem_check_launch("pearson");
}

// workspace: (C*k + C) floats for the centred prototypes and their std (library keeps no state).
extern "C" int uem_pearson_sim(const float* feat, const float* protos, float* sim, float* workspace,
                                  int n, int k, int C, void* stream) {
    UEM_REQUIRE(feat && protos && sim && workspace, "pearson_sim: null pointer");
    UEM_REQUIRE(n > 0 && k >= 8 && (k % 4) == 0 && C >= 1 && C <= UEM_MAX_CLASSES, "pearson_sim: bad shape n=%d k=%d C=%d", n, k, C);
    UEM_REQUIRE((size_t)C * k * 4 <= 150 * 1024, "pearson_sim: C*k too large for LDS");
    return pearson_launch(feat, protos, sim, workspace, n, C, k, true, (hipStream_t)stream);
}
extern "C" int uem_pearson_dist(const float* a, const float* b, float* dist, float* workspace, int n, int m,
                                   int k, void* stream) {
    UEM_REQUIRE(a && b && dist && workspace, "pearson_dist: null pointer");
    UEM_REQUIRE(n > 0 && k >= 8 && (k % 4) == 0 && m >= 1 && m <= UEM_MAX_CLASSES, "pearson_dist: bad shape");
    UEM_REQUIRE((size_t)m * k * 4 <= 150 * 1024, "pearson_dist: m*k too large for LDS");
    return pearson_launch(a, b, dist, workspace, n, m, k, false, (hipStream_t)stream);
}

// ================================================================================================
// index max
// ================================================================================================
__global__ void index_max_kernel(const int64_t* __restrict__ idx, int64_t count, unsigned long long* out) {
    long long m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        long long v = idx[i];
        m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        long long t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)m);
}
extern "C" int uem_index_max(const int64_t* idx, int64_t count, int64_t* out, void* stream) {
    UEM_REQUIRE(idx && out && count > 0, "index_max: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(out, 0, sizeof(int64_t), st);
    index_max_kernel<<<uem_stream_grid(count, 256 * 4), 256, 0, st>>>(idx, count, (unsigned long long*)out);
    return uem_check_launch("index_max");
}

// ================================================================================================
// superpixel segment max over an NCHW-planar soft label.
// Block = 64x16 pixel tile; ids are spatially compact, so a tile touches a handful of segments:
// an LDS open-addressing table (key = id) absorbs the per-pixel atomics, then one global
// atomicMax per (segment, class) per tile.  Overflowing ids fall back to direct global atomics.
// ================================================================================================
#define SEG_SLOTS 256
#define SEG_ROWS 16
// Block = 256 columns x SEG_ROWS rows.  A thread walks ONE column down the tile and keeps the running maximum of the segment it is
// in (superpixels are spatially compact: a column of 16 pixels crosses one or two of them), so the LDS table sees one update per
// (column, segment run) instead of one per pixel: the round-3 kernel issued 6 LDS atomics per pixel, 16 lanes of every wave on the
// same address (148 us for 268 MB at B = 32: LDS-atomic bound at 1.8 TB/s).  Rows of a plane are read coalesced (lanes = columns).
template <int CMAX, int CEX>
__global__ __launch_bounds__(256) void segment_max_kernel(const float* __restrict__ soft,
                                                          const int64_t* __restrict__ sup,
                                                          uint32_t* __restrict__ seg, int C_, int H, int W, int S,
                                                          int* __restrict__ oor) {
    const int C = CEX > 0 ? CEX : C_;
    __shared__ int keys[SEG_SLOTS];
    __shared__ uint32_t vals[SEG_SLOTS][CMAX];
    const int tid = threadIdx.x;
    for (int i = tid; i < SEG_SLOTS; i += 256) {
        keys[i] = -1;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) vals[i][c] = 0u;
    }
    __syncthreads();
    const int b = blockIdx.z;
    const int x = blockIdx.x * 256 + tid;
    const int y0 = blockIdx.y * SEG_ROWS;
    const size_t plane = (size_t)H * W;
    const float* sb = soft + (size_t)b * C * plane;
    const int64_t* ib = sup + (size_t)b * plane;
    uint32_t* segb = seg + (size_t)b * S * C;
    int bad = 0;                                        // largest id outside [0, S) seen by this thread (negative ids count as INT_MAX)
    int cur_id = -1;                                    // segment of the current run (-1: none / out of range)
    uint32_t cur[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) cur[c] = 0u;
    auto flush = [&]() {
        if (cur_id < 0) return;
        uint32_t hsh = ((uint32_t)cur_id * 2654435761u) >> 24;       // 8 bits
        int slot = -1;
        for (int probe = 0; probe < SEG_SLOTS; ++probe) {
            const int prev = atomicCAS(&keys[hsh], -1, cur_id);
            if (prev == -1 || prev == cur_id) { slot = (int)hsh; break; }
            hsh = (hsh + 1) & (SEG_SLOTS - 1);
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C && cur[c]) {
                if (slot >= 0) atomicMax(&vals[slot][c], cur[c]);
                else atomicMax(&segb[(size_t)cur_id * C + c], cur[c]);
            }
        }
    };
    if (x < W) {
        const int rows = min(SEG_ROWS, H - y0);
        for (int r = 0; r < rows; ++r) {
            const size_t p = (size_t)(y0 + r) * W + x;
            const int64_t id64 = ib[p];
            uint32_t kv[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) kv[c] = c < C ? f2key(sb[(size_t)c * plane + p]) : 0u;
            int id;
            if (id64 < 0 || id64 >= (int64_t)S) {
                bad = max(bad, id64 < 0 || id64 > 0x7fffffffLL ? 0x7fffffff : (int)id64);
                id = -1;
            } else id = (int)id64;
            if (id != cur_id) {
                flush();
                cur_id = id;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) cur[c] = kv[c];
            } else {
#pragma unroll
                for (int c = 0; c < CMAX; ++c) cur[c] = max(cur[c], kv[c]);
            }
        }
        flush();
    }
    __syncthreads();
    for (int i = tid; i < SEG_SLOTS * CMAX; i += 256) {
        const int slot = i / CMAX, c = i % CMAX;
        const int id = keys[slot];
        if (id >= 0 && c < C) {
            uint32_t kv = vals[slot][c];
            if (kv) atomicMax(&segb[(size_t)id * C + c], kv);
        }
    }
    // ids the table cannot hold are never folded into another segment: they are reported (the host raises) and the
    // refinement kernel leaves those pixels' weights untouched
    if (oor != nullptr && __any(bad != 0)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bad = max(bad, __shfl_xor(bad, o, 64));
        if ((tid & 63) == 0) atomicMax(oor, bad);
    }
}
extern "C" int uem_segment_max_planar(const float* soft, const int64_t* sup, uint32_t* seg_keys, int B, int C,
                                      int H, int W, int S, int* out_of_range, void* stream) {
    UEM_REQUIRE(soft && sup && seg_keys, "segment_max: null pointer");
    UEM_REQUIRE(B > 0 && C >= 1 && C <= UEM_MAX_CLASSES && H > 0 && W > 0 && S > 0, "segment_max: bad shape");
    dim3 grid((unsigned)uem_cdiv(W, 256), (unsigned)uem_cdiv(H, SEG_ROWS), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (C == 6) segment_max_kernel<8, 6><<<grid, 256, 0, st>>>(soft, sup, seg_keys, C, H, W, S, out_of_range);
    else if (C == 7) segment_max_kernel<8, 7><<<grid, 256, 0, st>>>(soft, sup, seg_keys, C, H, W, S, out_of_range);
    else if (C <= 8) segment_max_kernel<8, 0><<<grid, 256, 0, st>>>(soft, sup, seg_keys, C, H, W, S, out_of_range);
    else segment_max_kernel<16, 0><<<grid, 256, 0, st>>>(soft, sup, seg_keys, C, H, W, S, out_of_range);
    return uem_check_launch("segment_max");
}

// ================================================================================================
// fused label refinement (one thread per full-resolution pixel)
// ================================================================================================
template <int CMAX>
__device__ __forceinline__ void softmax_maxnorm(float (&v)[CMAX], int C, float inv_temp) {
    // v <- softmax(v * inv_temp) / (max + 1e-7)     (alignment.py:221-222, 230-235, 252-253).  Round 5, fewer instructions (the
    // refinement kernel is VALU-bound: 131 K wave-pixels x ~700 instructions x 4 cycles on 1024 SIMDs IS its 140 us): temperature and
    // log2(e) are one factor (the exponent is exp2(t - max t), t = v * inv_temp * log2 e); the largest exponential is exp2(0) = 1
    // exactly, so the softmax's maximum is 1 / sum and "softmax / (max + 1e-7)" is e * (rs * rcp(rs + 1e-7)), rs = 1 / sum: no second
    // maximum, one multiply per class.
    const float sc = inv_temp * 1.44269504088896340736f;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) { v[c] *= sc; m = fmaxf(m, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) { v[c] = __builtin_amdgcn_exp2f(v[c] - m); s += v[c]; }
    const float rs = fast_rcp(s);
    const float kf = rs * fast_rcp(rs + 1e-7f);
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) v[c] = v[c] * kf;
}

// One block = 256 consecutive pixels of ONE image row, so the vertical lerp is block-uniform and the strip only touches
// <= 256*(w-1)/(W-1) + 3 low-resolution cells: the two source rows of sim / logits1 / logits2 for that cell range are interpolated
// in y ONCE per block while they are staged into LDS ([map][cell][CMAX]) and every per-pixel tap is a 16-byte LDS read (72 scattered
// loads per pixel in the first version: 3.0 ms per B=32 step; round 3 staged both rows and interpolated 4 taps per pixel).  Round 4:
// the pixel's own operands (superpixel id first, then the soft label) are requested BEFORE the staging so that their latency, the
// staging's and the segment gather's overlap instead of queueing behind one another; divisions and libm calls are gone (common.h).
template <int CMAX>
__device__ __forceinline__ void xlerp_lds(const float* __restrict__ row, int C, int c0, const Lerp& lx, float (&v)[CMAX]) {
    const float4* a0 = reinterpret_cast<const float4*>(row + (lx.i0 - c0) * CMAX);
    const float4* a1 = reinterpret_cast<const float4*>(row + (lx.i1 - c0) * CMAX);
#pragma unroll
    for (int q = 0; q < CMAX / 4; ++q) {
        const float4 x0 = a0[q], x1 = a1[q];
        v[4 * q + 0] = fmaf(lx.l1, x1.x, lx.l0 * x0.x); v[4 * q + 1] = fmaf(lx.l1, x1.y, lx.l0 * x0.y);
        v[4 * q + 2] = fmaf(lx.l1, x1.z, lx.l0 * x0.z); v[4 * q + 3] = fmaf(lx.l1, x1.w, lx.l0 * x0.w);
    }
}

// CEX: the class count when it is one of the two the reference's datasets have (6: ISPRS, 7: LoveDA) -- every `c < C` guard of the
// unrolled per-class loops then folds at compile time (with a run-time C they were 266 v_cndmask + their compares per pixel, a
// quarter of the kernel's instructions); 0 = any C <= CMAX.
//
// Round 5.  (a) The superpixel view softmax(seg_max / T) / (max + 1e-7) is a function of the SEGMENT, not of the pixel: it is computed
// once per table entry by segment_weight_kernel (B * S entries, ~4 us) and the pixel gathers 8 finished floats (two 16-byte loads)
// instead of 6 keys + 6 exp + 2 rcp + ~50 other instructions -- same arithmetic, same values.  (b) A block owns LR_ROWS image rows of
// its 256-pixel strip instead of one: the kernel was latency-bound, not bandwidth- or VALU-bound (counters of round 4: VALU active
// 17 % of the wave cycles, 41 % of them parked on memory; 32768 blocks of three dependent round trips -- id, then the segment gather,
// then the stores -- around ~700 instructions), so every thread now has its LR_ROWS pixels' ids and soft labels in flight at once,
// then all their segment gathers, and the block's launch, barrier and per-(image, class) maximum are paid once per LR_ROWS rows.
#define LR_ROWS 4
template <int CMAX, int CEX>
__global__ __launch_bounds__(256) void segment_weight_kernel(const uint32_t* __restrict__ seg, float* __restrict__ segw, int64_t nseg,
                                                             int C_, float inv_temp) {
    const int C = CEX > 0 ? CEX : C_;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nseg) return;
    const uint32_t* sg = seg + (size_t)i * C;
    float v[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        const uint32_t kv = c < C ? sg[c] : 0u;
        v[c] = kv ? key2f(kv) : 0.f;
    }
    softmax_maxnorm<CMAX>(v, C, inv_temp);
    float4* dst = reinterpret_cast<float4*>(segw + (size_t)i * CMAX);
#pragma unroll
    for (int q = 0; q < CMAX / 4; ++q)
        dst[q] = make_float4(4 * q + 0 < C ? v[4 * q + 0] : 0.f, 4 * q + 1 < C ? v[4 * q + 1] : 0.f, 4 * q + 2 < C ? v[4 * q + 2] : 0.f,
                             4 * q + 3 < C ? v[4 * q + 3] : 0.f);
}

// MODE_T: the refinement mode when it is `all` with two prediction heads (the training step's call: the mode tests, the absent views'
// zero fills and the single-head branch fold away); -1 = mode / head count read at run time
// FULLT: W is a multiple of 256 and H of LR_ROWS -- every thread has a pixel in every row, no validity tests
template <int CMAX, int CEX, int MODE_T = -1, bool FULLT = false>
__global__ __launch_bounds__(256) void label_refine_kernel(
    const float* __restrict__ soft, const int64_t* __restrict__ sup, const float* __restrict__ sim,
    const float* __restrict__ lg1, const float* __restrict__ lg2_, const float* __restrict__ segw,
    const int64_t* __restrict__ ignore_id, float* __restrict__ out, float* __restrict__ blockmax, int C_, int h,
    int w, int H, int W, int S, float inv_temp, int mode_, int ncell, float* __restrict__ cand_val,
    uint8_t* __restrict__ cand_code, float cand_low) {
    const int C = CEX > 0 ? CEX : C_;
    const int mode = MODE_T >= 0 ? MODE_T : mode_;
    const float* const lg2 = lg2_;
    const bool two_heads = MODE_T >= 0 ? true : lg2_ != nullptr;
    extern __shared__ __attribute__((aligned(16))) float lowres[];       // [LR_ROWS][3 maps][ncell][CMAX], interpolated in y
    const int b = blockIdx.z, Y0 = blockIdx.y * LR_ROWS, X0 = blockIdx.x * 256;
    const size_t plane = (size_t)H * W;
    const int X = X0 + threadIdx.x;
    const bool active = FULLT || X < W;
    const bool use_sup = mode == UEM_REFINE_ALL || mode == UEM_REFINE_S;
    // ---- this thread's pixels (one per row of the block): operands requested first, consumed last ---------------------------
    int64_t id[LR_ROWS];
    float sv[LR_ROWS][CMAX];
    bool rowok[LR_ROWS];
#pragma unroll
    for (int r = 0; r < LR_ROWS; ++r) {
        rowok[r] = FULLT || (active && (Y0 + r) < H);
        id[r] = 0;
        if (rowok[r] && use_sup) id[r] = sup[(size_t)b * plane + (size_t)(Y0 + r) * W + X];
    }
#pragma unroll
    for (int r = 0; r < LR_ROWS; ++r) {
        const size_t p = (size_t)(Y0 + r) * W + X;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) sv[r][c] = (rowok[r] && c < C) ? soft[((size_t)b * C + c) * plane + p] : 0.f;
    }
    // ---- the strip's low-resolution cells, interpolated between the two source rows of each image row ------------------------
    const float sy = lerp_scale_ac(h, H), sx = lerp_scale_ac(w, W);
    const int c0 = lerp_ac(X0, w, sx).i0;                                // first cell of the strip
    const int Xl = min(X0 + 255, W - 1);
    const int c1 = lerp_ac(Xl, w, sx).i1;                                // last cell of the strip
    const int nc = c1 - c0 + 1;                                          // <= ncell by construction
    const int ncc = ncell * CMAX;                                        // CMAX is 8 or 16: cell / class by shift and mask
    {
        const float* maps[3] = {sim, lg1, lg2};
#pragma unroll
        for (int r = 0; r < LR_ROWS; ++r) {
            const Lerp ly = lerp_ac(min(Y0 + r, H - 1), h, sy);
            for (int i = threadIdx.x; i < ncc; i += 256) {
                const int cell = i / CMAX, c = i % CMAX;
                const bool in = cell < nc && c < C;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    float v = 0.f;
                    if (maps[m] != nullptr && in) {
                        const float r0 = maps[m][(((size_t)b * h + ly.i0) * w + (c0 + cell)) * C + c];
                        const float r1 = maps[m][(((size_t)b * h + ly.i1) * w + (c0 + cell)) * C + c];
                        v = fmaf(ly.l1, r1, ly.l0 * r0);
                    }
                    lowres[(r * 3 + m) * ncc + i] = v;
                }
            }
        }
    }
    // ---- the superpixels' finished class weights: the gathers can leave as soon as the ids are here -------------------------
    bool ignored[LR_ROWS];
    float sw[LR_ROWS][CMAX];
#pragma unroll
    for (int r = 0; r < LR_ROWS; ++r) {
        ignored[r] = true;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) sw[r][c] = 1.0f;
        if (rowok[r] && use_sup) {
            const bool inrange = id[r] >= 0 && id[r] < (int64_t)S;
            // an id outside the table (reported by uem_segment_max_planar, the host raises) never borrows another
            // segment's maxima: the pixel keeps its weight, as an ignored one does
            ignored[r] = (id[r] == *ignore_id) || !inrange;
            if (inrange) {
                const float4* sg = reinterpret_cast<const float4*>(segw + ((size_t)b * S + (size_t)id[r]) * CMAX);
#pragma unroll
                for (int q = 0; q < CMAX / 4; ++q) {
                    const float4 t = sg[q];
                    sw[r][4 * q + 0] = t.x; sw[r][4 * q + 1] = t.y; sw[r][4 * q + 2] = t.z; sw[r][4 * q + 3] = t.w;
                }
            }
        }
    }
    __syncthreads();
    float omax[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) omax[c] = 0.f;
    const Lerp lx = lerp_ac(active ? X : 0, w, sx);
#pragma unroll
    for (int r = 0; r < LR_ROWS; ++r) {
        if (!rowok[r]) continue;
        const float* s_sim = lowres + (r * 3 + 0) * ncc;
        const float* s_l1 = lowres + (r * 3 + 1) * ncc;
        const float* s_l2 = lowres + (r * 3 + 2) * ncc;
        float wgt[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) wgt[c] = 0.f;
        if (mode == UEM_REFINE_ALL || mode == UEM_REFINE_P) {          // prototype view
            float v[CMAX];
            xlerp_lds<CMAX>(s_sim, C, c0, lx, v);
            softmax_maxnorm<CMAX>(v, C, 1.0f);
#pragma unroll
            for (int c = 0; c < CMAX; ++c) wgt[c] += v[c];
        }
        if (mode == UEM_REFINE_ALL || mode == UEM_REFINE_L) {          // prediction view
            float v[CMAX];
            xlerp_lds<CMAX>(s_l1, C, c0, lx, v);
            if (two_heads) {
                float u[CMAX];
                xlerp_lds<CMAX>(s_l2, C, c0, lx, u);
                // 0.5 * (softmax(x1/T) + softmax(x2/T)), then max-normalise
                const float sc = inv_temp * 1.44269504088896340736f;
                float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) { v[c] *= sc; u[c] *= sc; m1 = fmaxf(m1, v[c]); m2 = fmaxf(m2, u[c]); }
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) { v[c] = __builtin_amdgcn_exp2f(v[c] - m1); u[c] = __builtin_amdgcn_exp2f(u[c] - m2); s1 += v[c]; s2 += u[c]; }
                const float r1 = fast_rcp(s1) * 0.5f, r2 = fast_rcp(s2) * 0.5f;
                float pm = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) { v[c] = fmaf(v[c], r1, u[c] * r2); pm = fmaxf(pm, v[c]); }
                const float rd = fast_rcp(pm + 1e-7f);
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) v[c] = v[c] * rd;
            } else {
                softmax_maxnorm<CMAX>(v, C, inv_temp);
            }
#pragma unroll
            for (int c = 0; c < CMAX; ++c) wgt[c] += v[c];
        }
        if (use_sup) {                                                 // superpixel view (weights finished per segment)
            if (mode == UEM_REFINE_ALL) {
#pragma unroll
                for (int c = 0; c < CMAX; ++c) wgt[c] = ignored[r] ? wgt[c] : wgt[c] * sw[r][c];
            } else {
#pragma unroll
                for (int c = 0; c < CMAX; ++c) wgt[c] = ignored[r] ? 1.0f : sw[r][c];
            }
        }
        float o[CMAX];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { o[c] = 0.f; if (c < C) { o[c] = wgt[c] * sv[r][c]; sum += o[c]; } }
        const float rd = fast_rcp(sum + 1e-7f);
        const size_t p = (size_t)(Y0 + r) * W + X;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) { o[c] = o[c] * rd; out[((size_t)b * C + c) * plane + p] = o[c]; omax[c] = fmaxf(omax[c], o[c]); }
        if (cand_val != nullptr) {
            // The selection pass (pseudo_generation.py:62-84) keeps a pixel when exactly ONE class is above its threshold
            // max(top * plane maximum, low) >= low.  Which classes are above `low` is known here: none -> the pixel is ignored whatever
            // the maxima turn out to be (code 254); one -> only that class can pass, its value decides (code = class); several -> the
            // pass reads the pixel's classes again (code 255; impossible for low >= 0.5 up to rounding: the classes sum to 1).  The
            // pass then reads 5 bytes per pixel instead of 4 * C.
            int nlow = 0, code = 254;
            float val = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C && o[c] > cand_low) { if (nlow == 0) { code = c; val = o[c]; } ++nlow; }
            if (nlow > 1) code = 255;
            cand_val[(size_t)b * plane + p] = val;
            cand_code[(size_t)b * plane + p] = (uint8_t)code;
        }
    }
    // per-(b,c) maximum for the selection pass: block maximum -> blockmax[b][block][c]; a second tiny kernel
    // reduces the blocks.  (One atomicMax per wave on the B*C result words serialised 0.8 M atomics on 192
    // addresses and cost 3 ms: 95 % of this kernel.)
    __shared__ float wmax[4][CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        const float m = c < C ? wave_max_dpp(omax[c]) : 0.f;
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6][c] = m;
    }
    __syncthreads();
    if (threadIdx.x < CMAX) {
        const int c = threadIdx.x;
        const float m = fmaxf(fmaxf(wmax[0][c], wmax[1][c]), fmaxf(wmax[2][c], wmax[3][c]));
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        blockmax[((size_t)b * gridDim.x * gridDim.y + blk) * CMAX + c] = fmaxf(m, 0.f);
    }
}
__global__ __launch_bounds__(256) void blockmax_reduce_kernel(const float* __restrict__ blockmax, uint32_t* __restrict__ plane_max,
                                                              int nblk, int C, int cmax) {
    // grid = (C, B): one block per result word
    const int c = blockIdx.x, b = blockIdx.y;
    float m = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) m = fmaxf(m, blockmax[((size_t)b * nblk + i) * cmax + c]);
    __shared__ float red[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) plane_max[b * C + c] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

extern "C" int64_t uem_label_refine_workspace_floats(int B, int C, int H, int W, int S) {
    // per-block maxima [B][strips][row groups][cmax] + the per-segment class weights [B][S][cmax]
    const int cmax = C <= 8 ? 8 : 16;
    return (int64_t)B * uem_cdiv(W, 256) * uem_cdiv(H, LR_ROWS) * cmax + (int64_t)B * (S > 0 ? S : 0) * cmax;
}
struct LrSelect { int64_t* hard; float top, low; int64_t ignore; float* cand_val; uint8_t* cand_code; };
template <int CMAX, int CEX>
__global__ __launch_bounds__(256) void select_cand_kernel(const float* __restrict__ soft, const float* __restrict__ cand_val,
                                                          const uint8_t* __restrict__ cand_code, const float* __restrict__ blockmax,
                                                          int nblk, uint32_t* __restrict__ plane_max, int64_t* __restrict__ hard,
                                                          int C_, int64_t HW, float top, float low, int64_t ignore);
static int label_refine_impl(const float* soft, const int64_t* sup, const float* sim, const float* logits1, const float* logits2,
                             const uint32_t* seg_keys, const int64_t* ignore_id, float* soft_out, uint32_t* plane_max, float* workspace,
                             int B, int C, int h, int w, int H, int W, int S, float temp, int mode, const LrSelect* sel, void* stream);
extern "C" int uem_label_refine(const float* soft, const int64_t* sup, const float* sim, const float* logits1,
                                const float* logits2, const uint32_t* seg_keys, const int64_t* ignore_id,
                                float* soft_out, uint32_t* plane_max, float* workspace, int B, int C, int h, int w,
                                int H, int W, int S, float temp, int mode, void* stream) {
    return label_refine_impl(soft, sup, sim, logits1, logits2, seg_keys, ignore_id, soft_out, plane_max, workspace, B, C, h, w, H, W, S,
                             temp, mode, nullptr, stream);
}
// label_refine followed by pseudo_selection (train_ssl_uem.py:209-214 calls them back to back on the same map) in three launches:
// the refinement kernel leaves, beside the refined map, each pixel's candidate -- the one class above cutoff_low and its value -- and
// the selection pass reduces the blocks' maxima itself and reads the candidates (5 bytes per pixel instead of 4 * C; pixels with
// several classes above cutoff_low, if any, re-read their classes).  Results equal uem_label_refine + uem_pseudo_select bit for bit.
// cand_workspace: B*H*W floats followed by B*H*W bytes (uem_label_refine_select_workspace_bytes); needs H*W % 4 == 0
// (UEM_ERR_UNSUPPORTED otherwise: call the two entries in sequence).
extern "C" int64_t uem_label_refine_select_workspace_bytes(int B, int H, int W) { return (int64_t)B * H * W * 5; }
extern "C" int uem_label_refine_select(const float* soft, const int64_t* sup, const float* sim, const float* logits1,
                                       const float* logits2, const uint32_t* seg_keys, const int64_t* ignore_id, float* soft_out,
                                       uint32_t* plane_max, float* workspace, void* cand_workspace, int64_t* hard, int B, int C, int h,
                                       int w, int H, int W, int S, float temp, int mode, float cutoff_top, float cutoff_low,
                                       int64_t ignore_label, void* stream) {
    UEM_REQUIRE(hard && cand_workspace, "label_refine_select: null pointer");
    if (((int64_t)H * W) % 4 != 0 || (((uintptr_t)cand_workspace | (uintptr_t)hard) & 15) != 0)
        return uem_fail(UEM_ERR_UNSUPPORTED, "label_refine_select: needs H*W %% 4 == 0 and 16-byte aligned buffers");
    LrSelect sel{hard, cutoff_top, cutoff_low, ignore_label, reinterpret_cast<float*>(cand_workspace),
                 reinterpret_cast<uint8_t*>(cand_workspace) + (size_t)B * H * W * 4};
    return label_refine_impl(soft, sup, sim, logits1, logits2, seg_keys, ignore_id, soft_out, plane_max, workspace, B, C, h, w, H, W, S,
                             temp, mode, &sel, stream);
}
static int label_refine_impl(const float* soft, const int64_t* sup, const float* sim, const float* logits1, const float* logits2,
                             const uint32_t* seg_keys, const int64_t* ignore_id, float* soft_out, uint32_t* plane_max, float* workspace,
                             int B, int C, int h, int w, int H, int W, int S, float temp, int mode, const LrSelect* sel, void* stream) {
    UEM_REQUIRE(soft && soft_out && plane_max && workspace, "label_refine: null pointer");
    UEM_REQUIRE(mode >= 0 && mode <= 3, "label_refine: bad mode %d", mode);
    UEM_REQUIRE(B > 0 && C >= 1 && C <= UEM_MAX_CLASSES && h > 0 && w > 0 && H >= h && W >= w && H <= 65535 * LR_ROWS, "label_refine: bad shape");
    UEM_REQUIRE(temp > 0.f, "label_refine: temp must be > 0");
    const bool use_sup = mode == UEM_REFINE_ALL || mode == UEM_REFINE_S;
    if (mode == UEM_REFINE_ALL || mode == UEM_REFINE_P) UEM_REQUIRE(sim, "label_refine: sim required");
    if (mode == UEM_REFINE_ALL || mode == UEM_REFINE_L) UEM_REQUIRE(logits1, "label_refine: logits required");
    if (use_sup) UEM_REQUIRE(sup && seg_keys && ignore_id && S > 0, "label_refine: superpixel inputs required");
    // cells touched by a 256-pixel strip: floor(255*(w-1)/(W-1)) + 3 covers every alignment
    const int ncell = (W > 1 ? (int)((255LL * (w - 1)) / (W - 1)) : 0) + 4;      // +1 slack for float rounding
    const int cmax = C <= 8 ? 8 : 16;
    const size_t lds = (size_t)LR_ROWS * 3 * ncell * cmax * sizeof(float);
    UEM_REQUIRE(lds <= 150 * 1024, "label_refine: low-resolution strip does not fit LDS");
    dim3 grid((unsigned)uem_cdiv(W, 256), (unsigned)uem_cdiv(H, LR_ROWS), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    float* blockmax = workspace;
    float* segw = workspace + (size_t)B * grid.x * grid.y * cmax;
    const int64_t nseg = (int64_t)B * S;
#define LAUNCH_LR(CM, CE, MT, FT)                                                                                                \
    do {                                                                                                                         \
        if (use_sup)                                                                                                             \
            segment_weight_kernel<CM, CE><<<(unsigned)uem_cdiv(nseg, 256), 256, 0, st>>>(seg_keys, segw, nseg, C, 1.0f / temp);    \
        if (uem_allow_lds((const void*)label_refine_kernel<CM, CE, MT, FT>, lds))                                                 \
            label_refine_kernel<CM, CE, MT, FT><<<grid, 256, lds, st>>>(soft, sup, sim, logits1, logits2, segw, ignore_id,       \
                                                                        soft_out, blockmax, C, h, w, H, W, S, 1.0f / temp, mode, \
                                                                        ncell, sel ? sel->cand_val : nullptr,                   \
                                                                        sel ? sel->cand_code : nullptr, sel ? sel->low : 0.f);  \
    } while (0)
    // train_ssl_uem.py:209-214: all three views, two heads, on whole 256 x LR_ROWS pixel blocks
    const bool train_call = mode == UEM_REFINE_ALL && logits2 != nullptr && W % 256 == 0 && H % LR_ROWS == 0;
#define LAUNCH_LR3(CM, CE, MT) LAUNCH_LR(CM, CE, MT, false)
    if (C == 6) { if (train_call) LAUNCH_LR(8, 6, UEM_REFINE_ALL, true); else LAUNCH_LR3(8, 6, -1); }
    else if (C == 7) { if (train_call) LAUNCH_LR(8, 7, UEM_REFINE_ALL, true); else LAUNCH_LR3(8, 7, -1); }
    else if (C <= 8) LAUNCH_LR3(8, 0, -1);
    else LAUNCH_LR3(16, 0, -1);
#undef LAUNCH_LR3
#undef LAUNCH_LR
    if (sel == nullptr) {
        blockmax_reduce_kernel<<<dim3(C, B), 256, 0, st>>>(blockmax, plane_max, (int)(grid.x * grid.y), C, cmax);
        return uem_check_launch("label_refine");
    }
    const int64_t HW = (int64_t)H * W;
    const dim3 sgrid((unsigned)uem_cdiv(HW, 4096), (unsigned)B);
    const int nblk = (int)(grid.x * grid.y);
#define LAUNCH_SEL(CM, CE)                                                                                                          \
    select_cand_kernel<CM, CE><<<sgrid, 256, 0, st>>>(soft_out, sel->cand_val, sel->cand_code, blockmax, nblk, plane_max, sel->hard, C, \
                                                      HW, sel->top, sel->low, sel->ignore)
    if (C == 6) LAUNCH_SEL(8, 6);
    else if (C == 7) LAUNCH_SEL(8, 7);
    else if (C <= 8) LAUNCH_SEL(8, 0);
    else LAUNCH_SEL(16, 0);
#undef LAUNCH_SEL
    return uem_check_launch("label_refine_select");
}

// grid (chunks of 4096 pixels, B); the block first reduces its image's block maxima (label_refine_kernel left one entry per block:
// [b][block][cmax]) to the thresholds -- the same maxima and the same products as blockmax_reduce_kernel + pseudo_select_kernel
template <int CMAX, int CEX>
__global__ __launch_bounds__(256) void select_cand_kernel(const float* __restrict__ soft, const float* __restrict__ cand_val,
                                                          const uint8_t* __restrict__ cand_code, const float* __restrict__ blockmax,
                                                          int nblk, uint32_t* __restrict__ plane_max, int64_t* __restrict__ hard,
                                                          int C_, int64_t HW, float top, float low, int64_t ignore) {
    const int C = CEX > 0 ? CEX : C_;
    const int b = blockIdx.y, tid = threadIdx.x;
    float m[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) m[c] = 0.f;
    for (int i = tid; i < nblk; i += 256) {
        const float4* src = reinterpret_cast<const float4*>(blockmax + ((size_t)b * nblk + i) * CMAX);
#pragma unroll
        for (int q = 0; q < CMAX / 4; ++q) {
            const float4 t = src[q];
            m[4 * q] = fmaxf(m[4 * q], t.x); m[4 * q + 1] = fmaxf(m[4 * q + 1], t.y);
            m[4 * q + 2] = fmaxf(m[4 * q + 2], t.z); m[4 * q + 3] = fmaxf(m[4 * q + 3], t.w);
        }
    }
    __shared__ float wm[4][CMAX];
    __shared__ float thr[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        const float v = wave_max_dpp(m[c]);
        if ((tid & 63) == 0) wm[tid >> 6][c] = v;
    }
    __syncthreads();
    if (tid < CMAX) {
        const float pm = fmaxf(fmaxf(wm[0][tid], wm[1][tid]), fmaxf(wm[2][tid], wm[3][tid]));
        thr[tid] = tid < C ? fmaxf(__fmul_rn(pm, top), low) : INFINITY;
        if (blockIdx.x == 0 && tid < C) plane_max[b * C + tid] = __float_as_uint(pm);
    }
    __syncthreads();
    const float* const vb = cand_val + (size_t)b * HW;
    const uint8_t* const cb = cand_code + (size_t)b * HW;
    int64_t* const hb = hard + (size_t)b * HW;
    const int64_t p0 = (int64_t)blockIdx.x * 4096 + tid * 4;
    float4 v4[4];
    uchar4 c4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t p = p0 + j * 1024;
        if (p < HW) { v4[j] = *reinterpret_cast<const float4*>(vb + p); c4[j] = *reinterpret_cast<const uchar4*>(cb + p); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t p = p0 + j * 1024;
        if (p >= HW) continue;
        const float vv[4] = {v4[j].x, v4[j].y, v4[j].z, v4[j].w};
        const unsigned cc[4] = {c4[j].x, c4[j].y, c4[j].z, c4[j].w};
        int64_t out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int64_t lab = ignore;
            if (cc[e] < (unsigned)C) {
                if (vv[e] > thr[cc[e]]) lab = (int64_t)cc[e];
            } else if (cc[e] == 255u) {                      // several classes above cutoff_low: the selection rule in full
                int cnt = 0, first = 0;
                for (int c = 0; c < C; ++c) {
                    const float x = soft[((size_t)b * C + c) * HW + p + e];
                    if (x > thr[c]) { if (cnt == 0) first = c; ++cnt; }
                }
                if (cnt == 1) lab = first;
            }
            out[e] = lab;
        }
        *reinterpret_cast<longlong2*>(hb + p) = make_longlong2(out[0], out[1]);
        *reinterpret_cast<longlong2*>(hb + p + 2) = make_longlong2(out[2], out[3]);
    }
}

// ================================================================================================
// pseudo_selection
// ================================================================================================
__global__ __launch_bounds__(256) void plane_max_kernel(const float* __restrict__ mask, uint32_t* __restrict__ pm,
                                                        int64_t HW) {
    // grid: (chunks, B*C); monotone key so negative inputs still order correctly
    const float* pl = mask + (size_t)blockIdx.y * HW;
    float m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) m = fmaxf(m, pl[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(&pm[blockIdx.y], f2key(m));
}
__global__ void plane_max_decode_kernel(uint32_t* pm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pm[i] = __float_as_uint(key2f(pm[i]));
}
extern "C" int uem_plane_max(const float* mask, uint32_t* plane_max, int B, int C, int64_t HW, void* stream) {
    UEM_REQUIRE(mask && plane_max && B > 0 && C > 0 && HW > 0, "plane_max: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(plane_max, 0, sizeof(uint32_t) * B * C, st);
    int chunks = (int)uem_cdiv(HW, 256 * 8);
    if (chunks > 64) chunks = 64;
    plane_max_kernel<<<dim3(chunks, B * C), 256, 0, st>>>(mask, plane_max, HW);
    plane_max_decode_kernel<<<(int)uem_cdiv(B * C, 64), 64, 0, st>>>(plane_max, B * C);
    return uem_check_launch("plane_max");
}

template <int CMAX, int CEX>
__global__ __launch_bounds__(256) void pseudo_select_kernel(const float* __restrict__ mask,
                                                            const uint32_t* __restrict__ plane_max,
                                                            int64_t* __restrict__ hard, int* __restrict__ range_flag,
                                                            int C_, int64_t HW, float top, float low, int64_t ignore) {
    const int C = CEX > 0 ? CEX : C_;
    const int b = blockIdx.y;
    float thr[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        thr[c] = c < C ? fmaxf(__fmul_rn(__uint_as_float(plane_max[b * C + c]), top), low) : INFINITY;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    int cnt = 0, lab = 0;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
            float v = mask[((size_t)b * C + c) * HW + p];
            bad |= !(v >= 0.f && v <= 1.f);
            if (v > thr[c]) { if (cnt == 0) lab = c; ++cnt; }
        }
    }
    hard[(size_t)b * HW + p] = (cnt == 1) ? (int64_t)lab : ignore;
    if (bad) atomicOr(range_flag, 1);
}
extern "C" int uem_pseudo_select(const float* mask, const uint32_t* plane_max, int64_t* hard, int* range_flag, int B,
                                 int C, int64_t HW, float cutoff_top, float cutoff_low, int64_t ignore_label,
                                 void* stream) {
    UEM_REQUIRE(mask && plane_max && hard && range_flag, "pseudo_select: null pointer");
    UEM_REQUIRE(B > 0 && C >= 1 && C <= UEM_MAX_CLASSES && HW > 0, "pseudo_select: bad shape");
    dim3 grid((unsigned)uem_cdiv(HW, 256), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (C == 6) pseudo_select_kernel<8, 6><<<grid, 256, 0, st>>>(mask, plane_max, hard, range_flag, C, HW, cutoff_top, cutoff_low, ignore_label);
    else if (C == 7) pseudo_select_kernel<8, 7><<<grid, 256, 0, st>>>(mask, plane_max, hard, range_flag, C, HW, cutoff_top, cutoff_low, ignore_label);
    else if (C <= 8) pseudo_select_kernel<8, 0><<<grid, 256, 0, st>>>(mask, plane_max, hard, range_flag, C, HW, cutoff_top, cutoff_low, ignore_label);
    else pseudo_select_kernel<16, 0><<<grid, 256, 0, st>>>(mask, plane_max, hard, range_flag, C, HW, cutoff_top, cutoff_low, ignore_label);
    return uem_check_launch("pseudo_select");
}

// ================================================================================================
// DownscaleLabel: one block per output cell, LDS histogram over (n_classes + 1) bins
// ================================================================================================
__global__ __launch_bounds__(256) void downscale_label_kernel(const int64_t* __restrict__ label,
                                                              int64_t* __restrict__ out, int H, int W, int scale,
                                                              int n_classes, int64_t ignore, float min_ratio) {
    __shared__ int hist[UEM_MAX_CLASSES + 2];
    const int tid = threadIdx.x;
    if (tid <= n_classes) hist[tid] = 0;
    __syncthreads();
    const int wo = W / scale, ho = H / scale;
    const int cx = blockIdx.x % wo, cy = (blockIdx.x / wo) % ho, b = blockIdx.x / (wo * ho);
    const int64_t* base = label + ((size_t)b * H + (size_t)cy * scale) * W + (size_t)cx * scale;
    for (int i = tid; i < scale * scale; i += 256) {
        int64_t v = base[(size_t)(i / scale) * W + (i % scale)];
        int bin = (v == ignore || v < 0 || v >= n_classes) ? n_classes : (int)v;
        atomicAdd(&hist[bin], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0, bc = hist[0];
        for (int c = 1; c <= n_classes; ++c) if (hist[c] > bc) { bc = hist[c]; best = c; }   // first max wins
        float ratio = (float)bc / (float)(scale * scale);
        int64_t r = best;
        if (best == n_classes) r = ignore;
        if (ratio < min_ratio) r = ignore;
        out[blockIdx.x] = r;
    }
}
extern "C" int uem_downscale_label(const int64_t* label, int64_t* out, int B, int H, int W, int scale, int n_classes,
                                   int64_t ignore_label, float min_ratio, void* stream) {
    UEM_REQUIRE(label && out, "downscale_label: null pointer");
    UEM_REQUIRE(B > 0 && scale > 1 && H % scale == 0 && W % scale == 0, "downscale_label: H, W must be multiples of scale");
    UEM_REQUIRE(n_classes >= 1 && n_classes <= UEM_MAX_CLASSES, "downscale_label: bad n_classes");
    int cells = B * (H / scale) * (W / scale);
    downscale_label_kernel<<<cells, 256, 0, (hipStream_t)stream>>>(label, out, H, W, scale, n_classes, ignore_label, min_ratio);
    return uem_check_launch("downscale_label");
}

// ================================================================================================
// prototype sums: deterministic two-stage (row chunks -> partial slabs -> ordered reduce)
// ================================================================================================
template <int CMAX>
__global__ __launch_bounds__(256) void proto_partial_kernel(const float* __restrict__ feat,
                                                            const int64_t* __restrict__ lab, float* __restrict__ part,
                                                            float* __restrict__ part_cnt, int n, int k, int C,
                                                            int rows_per_chunk, int64_t ignore) {
    const int chunk = blockIdx.y;
    const int kk = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(n, r0 + rows_per_chunk);
    float4 acc[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kk < k) {
        for (int r = r0; r < r1; ++r) {
            const int64_t l = lab[r];                      // block-uniform
            if (l == ignore || l < 0 || l >= C) continue;
            const float4 v = *reinterpret_cast<const float4*>(feat + (size_t)r * k + kk);
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c == (int)l) { acc[c].x += v.x; acc[c].y += v.y; acc[c].z += v.z; acc[c].w += v.w; }
            }
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) *reinterpret_cast<float4*>(part + ((size_t)chunk * C + c) * k + kk) = acc[c];
    }
    if (blockIdx.x == 0 && threadIdx.x < C) {
        float cnt = 0.f;
        for (int r = r0; r < r1; ++r) cnt += (lab[r] == (int64_t)threadIdx.x) ? 1.f : 0.f;
        part_cnt[chunk * C + threadIdx.x] = cnt;
    }
}
__global__ void proto_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_cnt,
                                    float* __restrict__ sums, float* __restrict__ counts, int chunks, int Ck, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Ck) {
        float s = 0.f;
        for (int j = 0; j < chunks; ++j) s += part[(size_t)j * Ck + i];
        sums[i] = s;
    }
    if (i < C) {
        float s = 0.f;
        for (int j = 0; j < chunks; ++j) s += part_cnt[j * C + i];
        counts[i] = s;
    }
}
extern "C" int uem_proto_sums(const float* feat, const int64_t* label_ds, float* sums, float* counts, float* workspace,
                              int n, int k, int C, int64_t ignore_label, void* stream) {
    UEM_REQUIRE(feat && label_ds && sums && counts && workspace, "proto_sums: null pointer");
    UEM_REQUIRE(n > 0 && k > 0 && (k % 4) == 0 && C >= 1 && C <= UEM_MAX_CLASSES, "proto_sums: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = UEM_PROTO_SPLIT;
    const int rpc = (int)uem_cdiv(n, chunks);
    float* part = workspace;
    float* part_cnt = workspace + (size_t)chunks * C * k;
    dim3 grid((unsigned)uem_cdiv(k, 1024), (unsigned)chunks);
    if (C <= 8) proto_partial_kernel<8><<<grid, 256, 0, st>>>(feat, label_ds, part, part_cnt, n, k, C, rpc, ignore_label);
    else proto_partial_kernel<16><<<grid, 256, 0, st>>>(feat, label_ds, part, part_cnt, n, k, C, rpc, ignore_label);
    proto_reduce_kernel<<<(int)uem_cdiv((int64_t)C * k, 256), 256, 0, st>>>(part, part_cnt, sums, counts, chunks, C * k, C);
    return uem_check_launch("proto_sums");
}
__global__ void proto_ema_kernel(const float* __restrict__ sums, const float* __restrict__ counts,
                                 float* __restrict__ protos, int k, int C, float decay) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * k) return;
    const float n = counts[i / k];
    const float old = protos[i];
    const float local = (n < 1.f) ? old : sums[i] / (n + 1e-7f);
    protos[i] = (1.0f - decay) * local + decay * old;
}
extern "C" int uem_proto_ema(const float* sums, const float* counts, float* protos, int k, int C, float decay, void* stream) {
    UEM_REQUIRE(sums && counts && protos && k > 0 && C > 0, "proto_ema: bad arguments");
    UEM_REQUIRE(decay > 0.f && decay < 1.f, "proto_ema: decay must be in (0,1)");
    proto_ema_kernel<<<(int)uem_cdiv((int64_t)C * k, 256), 256, 0, (hipStream_t)stream>>>(sums, counts, protos, k, C, decay);
    return uem_check_launch("proto_ema");
}

// ================================================================================================
// generic torch_scatter-style scatter (API completeness; not on the fused hot path)
// ================================================================================================
__global__ void scatter_kernel(const float* __restrict__ src, const int64_t* __restrict__ index, float* __restrict__ out,
                               float* __restrict__ cnt, int N, int C, int S, int reduce) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int64_t id = index[(size_t)b * N + p];
    if (id < 0 || id >= S) return;
    const float* s = src + ((size_t)b * N + p) * C;
    float* o = out + ((size_t)b * S + id) * C;
    if (reduce == 0) {
        for (int c = 0; c < C; ++c) atomicMax(reinterpret_cast<uint32_t*>(o) + c, f2key(s[c]));
    } else {
        for (int c = 0; c < C; ++c) atomicAdd(o + c, s[c]);
        if (reduce == 2) atomicAdd(cnt + (size_t)b * S + id, 1.0f);
    }
}
__global__ void scatter_finish_kernel(float* __restrict__ out, const float* __restrict__ cnt, int64_t total, int C, int reduce) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (reduce == 0) {
        uint32_t kv = reinterpret_cast<uint32_t*>(out)[i];
        out[i] = kv ? key2f(kv) : 0.f;
    } else if (reduce == 2) {
        float n = cnt[i / C];
        out[i] = out[i] / fmaxf(n, 1.f);
    }
}
extern "C" int uem_scatter(const float* src, const int64_t* index, float* out, float* workspace, int B, int N, int C,
                           int S, int reduce, void* stream) {
    UEM_REQUIRE(src && index && out, "scatter: null pointer");
    UEM_REQUIRE(B > 0 && N > 0 && C > 0 && S > 0 && reduce >= 0 && reduce <= 2, "scatter: bad arguments");
    UEM_REQUIRE(reduce != 2 || workspace, "scatter: mean needs a B*S workspace");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * S * C, st);
    if (reduce == 2) hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)B * S, st);
    scatter_kernel<<<dim3((unsigned)uem_cdiv(N, 256), (unsigned)B), 256, 0, st>>>(src, index, out, workspace, N, C, S, reduce);
    const int64_t total = (int64_t)B * S * C;
    if (reduce != 1) scatter_finish_kernel<<<(int)uem_cdiv(total, 256), 256, 0, st>>>(out, workspace, total, C, reduce);
    return uem_check_launch("scatter");
}

// ---------------------------------------------------------------------------------------------------------
// superpixel edge shrinking (gast/superpixels.py:129-152): a pixel keeps its superpixel id only if every pixel of
// the (2*win+1)^2 window around it (clipped at the image border) carries the same id; the others get `ignore_id`
// (= H/16 * W/16, the id label_refine treats as "no superpixel").  Offline preprocessing in the reference (three
// nested Python loops per image); one thread per pixel here, the window comes out of L1/L2.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void superpixel_shrink_kernel(const int32_t* __restrict__ label, int32_t* __restrict__ out,
                                                                int B, int H, int W, int win, int32_t ignore_id) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int32_t* img = label + (i - (int64_t)y * W - x);
    const int32_t v = img[(size_t)y * W + x];
    bool keep = true;
    const int y0 = max(0, y - win), y1 = min(H - 1, y + win), x0 = max(0, x - win), x1 = min(W - 1, x + win);
    for (int yy = y0; yy <= y1 && keep; ++yy)
        for (int xx = x0; xx <= x1; ++xx)
            if (img[(size_t)yy * W + xx] != v) { keep = false; break; }
    out[i] = keep ? v : ignore_id;
}
extern "C" int uem_superpixel_shrink(const int32_t* label, int32_t* out, int B, int H, int W, int win_size, int32_t ignore_id,
                                     void* stream) {
    UEM_REQUIRE(label && out && label != out, "superpixel_shrink: needs distinct input and output buffers");
    UEM_REQUIRE(B > 0 && H > 0 && W > 0 && win_size >= 0 && win_size <= 16, "superpixel_shrink: bad arguments");
    const int64_t n = (int64_t)B * H * W;
    superpixel_shrink_kernel<<<(unsigned)uem_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(label, out, B, H, W, win_size, ignore_id);
    return uem_check_launch("superpixel_shrink");
}
