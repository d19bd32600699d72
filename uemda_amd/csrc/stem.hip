// The 7x7 stride-2 stem convolution (uemda/_resnets.py:149-153, 205-212: conv1 = Conv2d(3, 64, 7, stride 2, padding 3, bias False)) as
// its own pair of kernels (round 5).  Rounds 1-4 ran it through the generic register-staged implicit GEMM with the taps padded to
// 7 x 8 x 4 = 224 reduction elements for 147 real ones (1.52x the multiplies) and an operand gather of four global loads per k-step.
// Here a block owns an 8 x 32 tile of output pixels, stages its 21 x 69 input patch ONCE in LDS -- split by channel and by column
// parity, so that the 32 lanes of an MFMA operand (32 output pixels along x, input columns two apart) read 32 consecutive words --
// and walks the 147 (+1 zero) reduction elements in the order of the OHWI filter bank, (ky, kx, c): v_mfma_f32_32x32x2_f32 takes one
// word per lane and operand, so every element's patch offset is an immediate of the fully unrolled loop.
//   forward : rows = pixels, cols = output channels; the filter bank lives in registers (74 per lane: one 32-channel half, all k);
//             epilogue = plain 128-byte row segments + the BatchNorm tile statistics (per wave: 32 channels x 128 pixels, no LDS).
//   wgrad   : rows = reduction elements (5 x 32 >= 148), cols = output channels, reduction over pixels (2 per MFMA); a wave keeps
//             all ten 32 x 32 accumulators for its quarter of the tile's pixels; blocks are persistent and leave ONE partial bank
//             each, summed in a fixed order by two small kernels (deterministic, no atomics).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ST_ROWS 8
#define ST_COLS 32
#define ST_PR (2 * ST_ROWS + 5)          // 21 patch rows
#define ST_PC (2 * ST_COLS + 5)          // 69 patch columns
#define ST_HW 36                         // words per row of a parity plane (35 even / 34 odd columns used)
#define ST_PLANE (ST_PR * ST_HW)
#define ST_PATCH (6 * ST_PLANE)          // [channel 3][parity 2][21][36] floats = 18 144 bytes
#define ST_NPF ((ST_PR * ST_PC + 255) / 256)
#define ST_K 147
#define ST_KP 74                         // k pairs walked (148 elements, the last one zero)

// patch offset (words) of reduction element e = (ky * 7 + kx) * 3 + c for the pixel at patch origin
__host__ __device__ constexpr int st_off(int e) {
    return e >= ST_K ? 0 : (((e % 3) * 2 + (((e % 21) / 3) & 1)) * ST_PR + e / 21) * ST_HW + (((e % 21) / 3) >> 1);
}

// Forward: the two lane halves of an MFMA step hold two different reduction elements.  They are paired so that the second one's patch
// offset is the first one's plus a constant of the pair's group -- the lane's base pointer carries half * D and every read keeps an
// immediate offset (with arbitrary pairs the compiler kept 74 loop-invariant addresses in registers and spilled):
//   j <  49: (ky, kx) = (j / 7, j % 7), channels 0 | 1                 D = two planes
//   j <  70: channel 2, (kx, kx + 2) for kx = 0, 1, 4                   D = one word
//   j <  73: channel 2, kx = 5, rows (ky, ky + 1) for ky = 0, 2, 4      D = one row
//   j == 73: channel 2, kx = 5, ky = 6 | the zero element               D = 0
__host__ __device__ constexpr int st_pair_e(int j, int h) {
    return j < 49 ? ((j / 7) * 7 + j % 7) * 3 + h
         : j < 70 ? (((j - 49) / 3) * 7 + ((j - 49) % 3 == 0 ? 0 : (j - 49) % 3 == 1 ? 1 : 4) + 2 * h) * 3 + 2
         : j < 73 ? ((2 * (j - 70) + h) * 7 + 5) * 3 + 2
         : (h == 0 ? (6 * 7 + 5) * 3 + 2 : ST_K);
}
#define ST_DA (2 * ST_PLANE)
#define ST_DB 1
#define ST_DC ST_HW

struct StemP {
    const float* x4;          // (N, H, W, 4) fp32, channel 3 zero
    int N, H, W, Ho, Wo;
    int tiles_x, tiles_y, ntiles;
};

// this thread's share of a tile's patch: global -> registers (next tile, under the MFMAs of the current one) -> LDS
struct StemFetch {
    int pp[ST_NPF];           // patch row | patch column << 8 of this thread's i-th element, -1 = none
    __device__ __forceinline__ void init(int tid) {
#pragma unroll
        for (int i = 0; i < ST_NPF; ++i) {
            const int p = tid + 256 * i, pr = p / ST_PC, pc = p - pr * ST_PC;
            pp[i] = p < ST_PR * ST_PC ? (pr | (pc << 8)) : -1;
        }
    }
    __device__ __forceinline__ void load(const StemP& p, int n, int oy0, int ox0, float (&v)[ST_NPF][3]) const {
#pragma unroll
        for (int i = 0; i < ST_NPF; ++i) {
            const int iy = 2 * oy0 - 3 + (pp[i] & 255), ix = 2 * ox0 - 3 + (pp[i] >> 8);
            v[i][0] = v[i][1] = v[i][2] = 0.f;
            if (pp[i] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const float4 t = *reinterpret_cast<const float4*>(p.x4 + ((size_t)(n * p.H + iy) * p.W + ix) * 4);
                v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z;
            }
        }
    }
    __device__ __forceinline__ void stash(float* patch, const float (&v)[ST_NPF][3]) const {
#pragma unroll
        for (int i = 0; i < ST_NPF; ++i)
            if (pp[i] >= 0) {
                const int pr = pp[i] & 255, pc = pp[i] >> 8;
                float* const dst = patch + ((pc & 1) * ST_PR + pr) * ST_HW + (pc >> 1);     // channel c: + 2 * c * ST_PLANE
#pragma unroll
                for (int c = 0; c < 3; ++c) dst[2 * c * ST_PLANE] = v[i][c];
            }
    }
};

__device__ __forceinline__ void stem_tile(const StemP& p, int tile, int& n, int& oy0, int& ox0) {
    const int per = p.tiles_x * p.tiles_y;
    n = tile / per;
    const int rem = tile - n * per;
    const int ty = rem / p.tiles_x;
    oy0 = ty * ST_ROWS; ox0 = (rem - ty * p.tiles_x) * ST_COLS;
}

// ---------------------------------------------------------------------------------------------------------
// forward.  wave = (channel half, row group of 4 output rows); acc[t] = output row t of the group x 32 pixels x 32 channels
// ---------------------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(const StemP p, const float* __restrict__ w, float* __restrict__ y,
                                                          float* __restrict__ tile_stats) {
    // the filter bank passes through LDS once per block (coalesced 16-byte reads; lane = channel then reads words 147 apart: 147 = 19
    // mod 32, every bank once) -- read straight from the OHWI bank each of the 74 loads of a wave touched 64 cache lines; the patch
    // then lives in the same words
    __shared__ __attribute__((aligned(16))) float smem[64 * ST_K];
    static_assert(64 * ST_K >= ST_PATCH && (64 * ST_K) % 4 == 0, "the patch reuses the filter bank's staging area");
    float* const patch = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31, chtile = wave & 1, pg = wave >> 1;
    const int ch = chtile * 32 + l32;
    for (int i = tid; i < 64 * ST_K / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(w)[i];
    __syncthreads();
    float wb[ST_KP];
#pragma unroll
    for (int j = 0; j < ST_KP; ++j) {
        const int e = half ? st_pair_e(j, 1) : st_pair_e(j, 0);
        wb[j] = e < ST_K ? smem[ch * ST_K + e] : 0.f;
    }
    __syncthreads();
    StemFetch f;
    f.init(tid);
    float pf[ST_NPF][3];
    int n, oy0, ox0;
    int tile = blockIdx.x;
    if (tile < p.ntiles) { stem_tile(p, tile, n, oy0, ox0); f.load(p, n, oy0, ox0, pf); }
    const float* const pbase = patch + (2 * 4 * pg) * ST_HW + l32;
    const float* const pbA = pbase + half * ST_DA;
    const float* const pbB = pbase + half * ST_DB;
    const float* const pbC = pbase + half * ST_DC;
    const size_t tiles_m = (size_t)2 * p.ntiles;
    if (tile < p.ntiles) {
        f.stash(patch, pf);
        const int next = tile + gridDim.x;
        if (next < p.ntiles) { stem_tile(p, next, n, oy0, ox0); f.load(p, n, oy0, ox0, pf); }
    }
    int cn = 0, coy = 0, cox = 0;
    if (tile < p.ntiles) stem_tile(p, tile, cn, coy, cox);
    __syncthreads();
    for (; tile < p.ntiles; tile += gridDim.x) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // the four accumulators of a k pair stay together (four independent MFMAs back to back; left alone the scheduler strings a dozen
        // dependent ones on one accumulator to share patch words between taps) and the next pair's patch words are read under them
        float a[2][4];
        auto rd = [&](const int j, float (&dst)[4]) {
            const float* const src = (j < 49 ? pbA : j < 70 ? pbB : j < 73 ? pbC : pbase) + st_off(st_pair_e(j, 0));
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = src[2 * t * ST_HW];
        };
        rd(0, a[0]);
#pragma unroll
        for (int j = 0; j < ST_KP; ++j) {
            if (j + 1 < ST_KP) rd(j + 1, a[(j + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][t], wb[j], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next tile's patch goes into LDS and the tile after that is requested BEFORE this tile's stores are issued: the wait for
        // the patch words then has a whole MFMA phase behind it, and the stores drain under the next one
        __syncthreads();                                     // every wave past its reads of this patch
        const int next = tile + gridDim.x;
        const int on = cn, ooy = coy, oox = cox;
        if (next < p.ntiles) {
            f.stash(patch, pf);
            cn = n; coy = oy0; cox = ox0;
            const int after = next + gridDim.x;
            if (after < p.ntiles) { stem_tile(p, after, n, oy0, ox0); f.load(p, n, oy0, ox0, pf); }
        }
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float* const yr = y + ((size_t)(on * p.Ho + ooy + 4 * pg + t) * p.Wo + oox) * 64 + ch;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = acc[t][r];
                yr[(size_t)px * 64] = v;
                if (STATS) { s += v; q = fmaf(v, v, q); }
            }
        }
        if (STATS) {
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (half == 0) {
                const size_t entry = (size_t)2 * tile + pg;
                tile_stats[(size_t)ch * tiles_m + entry] = s;
                tile_stats[(size_t)(64 + ch) * tiles_m + entry] = q;
            }
        }
        __syncthreads();                                     // the next patch is complete
    }
}

static int stem_shape(StemP& p, const float* x4, int N, int H, int W) {
    p.x4 = x4; p.N = N; p.H = H; p.W = W;
    p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
    if (p.Ho % ST_ROWS != 0 || p.Wo % ST_COLS != 0) return 0;
    p.tiles_x = p.Wo / ST_COLS; p.tiles_y = p.Ho / ST_ROWS;
    const int64_t nt = (int64_t)N * p.tiles_x * p.tiles_y;
    if (nt >= (1 << 30) || (int64_t)N * H * W * 4 >= ((int64_t)1 << 31)) return 0;
    p.ntiles = (int)nt;
    return 1;
}

// Returns UEM_ERR_UNSUPPORTED (nothing launched) unless the output is whole 8 x 32 pixel tiles; the caller then takes
// uem_conv2d_stem_fwd[_stats].  tile_stats (optional): [2][64][N*Ho*Wo/128] partial sums of z and z*z, one entry per 128 pixels.
extern "C" int uem_stem_conv_fwd(const float* x4, const float* w_ohwi, float* z, int N, int H, int W, float* tile_stats, void* stream) {
    UEM_REQUIRE(x4 && w_ohwi && z && N > 0 && H >= 7 && W >= 7, "stem_conv_fwd: bad arguments");
    UEM_REQUIRE((((uintptr_t)x4) & 15) == 0, "stem_conv_fwd: x4 must be 16-byte aligned");
    StemP p;
    if (!stem_shape(p, x4, N, H, W)) return uem_fail(UEM_ERR_UNSUPPORTED, "stem_conv_fwd: needs Ho %% 8 == 0 and Wo %% 32 == 0");
    const int grid = p.ntiles < 512 ? p.ntiles : 512;
    hipStream_t st = (hipStream_t)stream;
    if (tile_stats) stem_fwd_kernel<true><<<grid, 256, 0, st>>>(p, w_ohwi, z, tile_stats);
    else stem_fwd_kernel<false><<<grid, 256, 0, st>>>(p, w_ohwi, z, nullptr);
    return uem_check_launch("stem_conv_fwd");
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient.  dW[o][e] += sum_px dz[px][o] * patch(px, e).  MFMA rows = e (five 32-row tiles), cols = o (two), k = pixel pair.
// A wave owns two output rows of the block's tile (64 pixels = 32 MFMA steps) and all ten accumulators.
// ---------------------------------------------------------------------------------------------------------
#define ST_WG_E 160
template <typename TD>
__device__ __forceinline__ float st_widen(TD v);
template <>
__device__ __forceinline__ float st_widen<float>(float v) { return v; }
template <>
__device__ __forceinline__ float st_widen<unsigned short>(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

template <typename TD>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const StemP p, const TD* __restrict__ dz, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float smem[ST_WG_E * 64];            // the patch while tiles are walked, the block's bank at the end
    float* const patch = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    int aoff[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) aoff[m] = st_off(m * 32 + l32) + (2 * 2 * wave) * ST_HW + half;
    // (rows 147 ... 159 of the bank read a real patch word and accumulate a finite value nobody reads)
    StemFetch f;
    f.init(tid);
    float pf[ST_NPF][3];
    int n, oy0, ox0;
    int tile = blockIdx.x;
    if (tile < p.ntiles) { stem_tile(p, tile, n, oy0, ox0); f.load(p, n, oy0, ox0, pf); }
    f32x16 acc[5][2];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
    for (; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();
        f.stash(patch, pf);
        __syncthreads();
        // dz of this wave's two rows: pixel pair q -> row q >> 4, column 2 * (q & 15) + half
        const TD* const d0 = dz + ((size_t)(n * p.Ho + oy0 + 2 * wave) * p.Wo + ox0 + half) * 64 + l32;
        const TD* const d1 = d0 + (size_t)p.Wo * 64;
        const int next = tile + gridDim.x;
        if (next < p.ntiles) { stem_tile(p, next, n, oy0, ox0); f.load(p, n, oy0, ox0, pf); }
        constexpr int CH = 8;                                                    // steps per dz prefetch group
        TD b[2][CH][2];
        auto fetch_b = [&](const int g, TD (&dst)[CH][2]) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int q = g * CH + u;
                const TD* const src = ((q >> 4) ? d1 : d0) + 2 * (q & 15) * 64;
                dst[u][0] = src[0]; dst[u][1] = src[32];
            }
        };
        fetch_b(0, b[0]);
#pragma unroll
        for (int g = 0; g < 32 / CH; ++g) {
            if (g + 1 < 32 / CH) fetch_b(g + 1, b[(g + 1) & 1]);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int q = g * CH + u;
                const int imm = (2 * (q >> 4)) * ST_HW + 2 * (q & 15);
                float a[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) a[m] = patch[aoff[m] + imm];
                const float b0 = st_widen<TD>(b[g & 1][u][0]), b1 = st_widen<TD>(b[g & 1][u][1]);
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b0, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b1, acc[m][1], 0, 0, 0);
                }
            }
        }
    }
    // the four waves' banks, added in wave order (fixed summation order), then the block's bank to its slot
    float* const bank = smem;                                                    // [160][64]
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float* const dst = &bank[e * 64 + j * 32 + l32];
                        *dst = wv == 0 ? acc[m][j][r] : *dst + acc[m][j][r];
                    }
        }
    }
    __syncthreads();
    float* const out = part + (size_t)blockIdx.x * (ST_WG_E * 64);
    for (int i = tid; i < ST_WG_E * 64 / 4; i += 256) reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(bank)[i];
}

#define ST_WG_FOLD 16
// part[nblk][160][64] -> part2[ST_WG_FOLD][160*64]: slice s sums blocks s, s + FOLD, ...
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ part2, int nblk) {
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i >= ST_WG_E * 64) return;
    float a = 0.f;
    for (int b = s; b < nblk; b += ST_WG_FOLD) a += part[(size_t)b * (ST_WG_E * 64) + i];
    part2[(size_t)s * (ST_WG_E * 64) + i] = a;
}
// dw_ohwi[o][e] += sum_s part2[s][e][o]
__global__ __launch_bounds__(256) void stem_wgrad_final_kernel(const float* __restrict__ part2, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * ST_K) return;
    const int o = i / ST_K, e = i - o * ST_K;
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < ST_WG_FOLD; ++s) a += part2[(size_t)s * (ST_WG_E * 64) + e * 64 + o];
    dw[i] += a;
}

extern "C" int64_t uem_stem_conv_wgrad_workspace_floats(void) { return (int64_t)(512 + ST_WG_FOLD) * ST_WG_E * 64; }

static int stem_wgrad_launch(const float* x4, const void* dz, int dz_bf16, float* dw_ohwi, float* workspace, int N, int H, int W,
                             void* stream) {
    UEM_REQUIRE(x4 && dz && dw_ohwi && workspace && N > 0 && H >= 7 && W >= 7, "stem_conv_wgrad: bad arguments");
    UEM_REQUIRE((((uintptr_t)x4 | (uintptr_t)workspace) & 15) == 0, "stem_conv_wgrad: x4 / workspace must be 16-byte aligned");
    StemP p;
    if (!stem_shape(p, x4, N, H, W)) return uem_fail(UEM_ERR_UNSUPPORTED, "stem_conv_wgrad: needs Ho %% 8 == 0 and Wo %% 32 == 0");
    const int grid = p.ntiles < 512 ? p.ntiles : 512;
    hipStream_t st = (hipStream_t)stream;
    float* const part = workspace;
    float* const part2 = workspace + (size_t)512 * ST_WG_E * 64;
    if (dz_bf16) stem_wgrad_kernel<unsigned short><<<grid, 256, 0, st>>>(p, reinterpret_cast<const unsigned short*>(dz), part);
    else stem_wgrad_kernel<float><<<grid, 256, 0, st>>>(p, reinterpret_cast<const float*>(dz), part);
    stem_wgrad_fold_kernel<<<dim3((ST_WG_E * 64 + 255) / 256, ST_WG_FOLD), 256, 0, st>>>(part, part2, grid);
    stem_wgrad_final_kernel<<<(64 * ST_K + 255) / 256, 256, 0, st>>>(part2, dw_ohwi);
    return uem_check_launch("stem_conv_wgrad");
}
// dw_ohwi (64, 7, 7, 3) += the stem's weight gradient; workspace: uem_stem_conv_wgrad_workspace_floats() floats.  Same shape rule as
// uem_stem_conv_fwd (UEM_ERR_UNSUPPORTED otherwise: the caller takes uem_conv2d_stem_wgrad).
extern "C" int uem_stem_conv_wgrad(const float* x4, const float* dz, float* dw_ohwi, float* workspace, int N, int H, int W, void* stream) {
    return stem_wgrad_launch(x4, dz, 0, dw_ohwi, workspace, N, H, W, stream);
}
extern "C" int uem_stem_conv_wgrad_bf16(const float* x4, const uint16_t* dz, float* dw_ohwi, float* workspace, int N, int H, int W,
                                        void* stream) {
    return stem_wgrad_launch(x4, dz, 1, dw_ohwi, workspace, N, H, W, stream);
}
