// Weight gradient of a convolution, LDS-DMA form (gfx950):   dW[o][tap][i] += sum_m dY[m][o] * A[m'(m,tap)][i]
//
// Measured on the register-staged kernel in conv.hip (profiles/r02_a_wgrad_ablation.txt): with the global loads and the
// ds_write staging removed its MFMA loop runs at 142 TFLOP/s, with them at 107 -- the exact-f32 matrix pipe was
// starved by operand staging (8 x global_load_dwordx4 + 8 x ds_write_b128 per thread and 32-pixel step), not by its own
// issue stream.  Here both operand tiles travel global -> LDS by `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write, the
// buffer descriptor's bounds check supplies the zeros of padding and of the ragged tail), double-buffered with ONE
// barrier per step, and everything the staging used to do to the data moves to the operand fetch:
//   * the producer's BatchNorm affine + ReLU (the conv consumed relu(bn(z)), only z is in memory): a lane's input channel
//     is fixed for the whole kernel, so scale/shift sit in registers and cost 2 VALU per fetched operand;
//   * zero padding AFTER that transform: a per-row validity word beside each staged tile.
// 3x3 layers: one block owns the three taps of a filter row (ky) for 32 consecutive pixels of one output row.  The
// three taps read the SAME staged input pixels shifted by kx*dilation, so a step stages dY (32 x TM) once and
// 32*stride + 2*dilation input pixels once for 3x the MFMAs: a third of the operand traffic per flop.
// k-major LDS tiles ([pixel][channel], as the rows sit in NHWC memory) are MFMA operands by plain ds_read (conv.hip).
// Replaces cuDNN's backward-filter behind nn.Conv2d (reference uemda/_resnets.py:95-110, Encoder.py:35,74).
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned int* lds_u32p;
// buffer_load_dwordx4 ... lds: per-lane global offset, LDS destination = wave-uniform base (M0) + lane * 16
extern "C" __device__ void uem_raw_buffer_load_lds(i32x4 rsrc, lds_u32p lds, int size, int voffset, int soffset, int offset,
                                                   int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

#define WG_OOB 0xFFFFFFF0u     // beyond every descriptor's num_records: the load returns zeros

struct WgP {
    const float* x;
    const float* dy;
    const float* in_scale;
    const float* in_shift;
    float* dw;
    int M, N, H, W, Cin, Ho, Wo, Cout, KH, KW, pad, x_ld, dy_ld;
    int steps_total, steps_per_split, tiles_co, tiles_ci;
    unsigned x_bytes, dy_bytes;
    // batched pixel-reduction GEMM (Winograd weight gradient, uem_wino_wgrad_gemm; 1x1 LINEAR form only): item b reduces the rows
    // [b*M, (b+1)*M) of x / dy into dw + b*Cout*Cin; nbatch == 1 everywhere else
    int nbatch;
};

__device__ __forceinline__ i32x4 wg_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));     // stride 0: raw byte offsets
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;                                                     // gfx9 raw buffer, dword elements
    return r;
}

// TM x TN: output channels x input channels of the tile; NTAP: 1 (1x1 conv) or 3 (the kx taps of one filter row);
// S: conv stride, D: dilation (NTAP == 3 only); AFFINE: BatchNorm affine + ReLU on the input operand;
// LINEAR: 1x1 stride-1 conv, the 32-pixel steps are plain consecutive rows of x (any M); otherwise a step is 32
// consecutive pixels of ONE output row (Wo % 32 == 0).
template <int BK, int TM, int TN, int NTAP, int S, int D, bool AFFINE, bool LINEAR>
struct WgCfg {
    static constexpr int MT = TM / 64, NT = TN / 64;                       // 2x2 waves, 32x32 MFMA tiles per wave
    static constexpr int XR = NTAP == 1 ? BK : (BK - 1) * S + 2 * D + 1;      // staged input pixels per step
    static constexpr int RPI_X = 256 / TN;                                 // rows per 1-KiB wave instruction
    static constexpr int XROWS = (XR + 4 * RPI_X - 1) / (4 * RPI_X) * (4 * RPI_X);
    static constexpr int D_IPW = BK * TM / 1024;                        // DMA instructions per wave and step
    static constexpr int X_IPW = XROWS / (4 * RPI_X);
    static constexpr int D_FLOATS = BK * TM, X_FLOATS = XROWS * TN, V_FLOATS = (XROWS + 4 + 3) / 4 * 4;
    static constexpr int STAGE_FLOATS = D_FLOATS + X_FLOATS + V_FLOATS;
    static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
    static constexpr int ACC = NTAP * MT * NT * 16;
    static constexpr int BPC_LDS = 160 * 1024 / LDS_BYTES;                 // resident blocks per CU: LDS, then registers
    static constexpr int BPC = BPC_LDS < 1 ? 1 : (BPC_LDS > (ACC <= 64 ? 3 : 2) ? (ACC <= 64 ? 3 : 2) : BPC_LDS);
};

template <int BK, int TM, int TN, int NTAP, int S, int D, bool AFFINE, bool LINEAR>
__global__ __launch_bounds__(256, (WgCfg<BK, TM, TN, NTAP, S, D, AFFINE, LINEAR>::BPC)) void wgrad_dma_kernel(const WgP p) {
    using C = WgCfg<BK, TM, TN, NTAP, S, D, AFFINE, LINEAR>;
    constexpr int MT = C::MT, NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- which (tile, filter row, pixel slice) ------------------------------------------------------------
    const int rows_k = NTAP == 3 ? p.KH : 1;                               // filter rows = tap groups
    const int ntiles = p.tiles_co * p.tiles_ci * rows_k;
    const int nwg = gridDim.x;
    int lin;                                                               // XCD-aware: blocks of one XCD walk consecutive ids
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int split = lin / (ntiles * p.nbatch);
    int t_ = lin - split * (ntiles * p.nbatch);
    const int bi = t_ / ntiles;                                            // batch item (0 unless batched)
    t_ -= bi * ntiles;
    const int ci_t = t_ % p.tiles_ci; t_ /= p.tiles_ci;
    const int ky = t_ % rows_k;
    const int co_t = t_ / rows_k;
    const int co0 = co_t * TM, ci0 = ci_t * TN;
    const int q_beg = bi * p.steps_total + split * p.steps_per_split;
    const int q_end = bi * p.steps_total + min(p.steps_total, (split + 1) * p.steps_per_split);
    const int T = q_end - q_beg;

    const i32x4 rs_d = wg_rsrc(p.dy, p.dy_bytes), rs_x = wg_rsrc(p.x, p.x_bytes);

    // ---- per-lane constants of the DMA pattern --------------------------------------------------------------
    unsigned dconst[C::D_IPW], xconst[C::X_IPW];
    int xrow[C::X_IPW];
#pragma unroll
    for (int j = 0; j < C::D_IPW; ++j) {
        const int e = (j * 4 + wave) * 256 + lane * 4;
        dconst[j] = (unsigned)(((e / TM) * p.dy_ld + co0 + (e % TM)) * 4);
    }
    constexpr int SX = NTAP == 3 ? 1 : S;                                  // input-pixel distance of consecutive staged rows
#pragma unroll
    for (int j = 0; j < C::X_IPW; ++j) {
        const int e = (j * 4 + wave) * 256 + lane * 4;
        xrow[j] = e / TN;
        xconst[j] = (unsigned)((xrow[j] * SX * p.x_ld + ci0 + (e % TN)) * 4);
    }
    // position of the next step to issue (ROW mode): image, output row, 32-pixel chunk of that row
    const int chunks = LINEAR ? 1 : p.Wo / BK;
    int qi = q_beg, in_ = 0, ioy = 0, ixc = 0;
    if (!LINEAR) {
        const int rowid = q_beg / chunks;
        ixc = q_beg - rowid * chunks;
        in_ = rowid / p.Ho;
        ioy = rowid - in_ * p.Ho;
    }
    // ONE function takes the stage being refilled and the stage being read as two __restrict__ pointers: after inlining
    // every DMA carries alias scope "fill" and every ds_read scope "use", which is what lets the compiler's wait-count
    // pass see that the operand reads of step t do not depend on the DMA of step t+1 issued just before them (without
    // the scopes it drains vmcnt(0) in front of the first ds_read: every wave then waits out its own prefetch).
    f32x16 acc[NTAP][MT][NT];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    const int fr = lane & 31, fh = lane >> 5;
    float sc[NT], sh[NT];
    if (AFFINE) {
#pragma unroll
        for (int j = 0; j < NT; ++j) { sc[j] = p.in_scale[ci0 + wn + j * 32 + fr]; sh[j] = p.in_shift[ci0 + wn + j * 32 + fr]; }
    }

    auto step = [&](float* __restrict__ fill, const float* __restrict__ use, const bool do_issue, const bool do_phase) {
        if (do_issue) {
            float* const Ds = fill;
            float* const Xs = fill + C::D_FLOATS;
            float* const Vs = Xs + C::X_FLOATS;
            const unsigned dbase = (unsigned)qi * (unsigned)(BK * 4) * (unsigned)p.dy_ld;   // dY rows are always linear in m
#pragma unroll
            for (int j = 0; j < C::D_IPW; ++j)
                uem_raw_buffer_load_lds(rs_d, (lds_u32p)(Ds + (j * 4 + wave) * 256), 16, (int)(dconst[j] + dbase), 0, 0, 0);
            if (LINEAR) {
                const unsigned xbase = (unsigned)qi * (unsigned)(BK * 4) * (unsigned)p.x_ld;
#pragma unroll
                for (int j = 0; j < C::X_IPW; ++j)
                    uem_raw_buffer_load_lds(rs_x, (lds_u32p)(Xs + (j * 4 + wave) * 256), 16, (int)(xconst[j] + xbase), 0, 0, 0);
            } else {
                const int iy = ioy * S - p.pad + ky * D;
                const int ix0 = ixc * (BK * S) - p.pad;
                const bool rowok = iy >= 0 && iy < p.H && in_ < p.N;
                const unsigned xbase = (unsigned)(((in_ * p.H + iy) * p.W + ix0) * p.x_ld) * 4u;
#pragma unroll
                for (int j = 0; j < C::X_IPW; ++j) {
                    const int ix = ix0 + xrow[j] * SX;
                    const bool ok = rowok && ix >= 0 && ix < p.W && xrow[j] < C::XR;
                    uem_raw_buffer_load_lds(rs_x, (lds_u32p)(Xs + (j * 4 + wave) * 256), 16, (int)(ok ? xconst[j] + xbase : WG_OOB), 0, 0, 0);
                    if (AFFINE && NTAP == 3 && (lane % (TN / 4)) == 0) Vs[xrow[j]] = ok ? 1.f : 0.f;
                }
                if (NTAP == 3 && tid == 0) Vs[C::XROWS] = rowok ? 1.f : 0.f;
                if (++ixc == chunks) { ixc = 0; if (++ioy == p.Ho) { ioy = 0; ++in_; } }
            }
            ++qi;
        }
        if (!do_phase) return;
        const float* const Ds = use;
        const float* const Xs = use + C::D_FLOATS;
        const float* const Vs = Xs + C::X_FLOATS;
        if (NTAP == 3) {
            // a filter row that falls into the padding for this output row contributes nothing: skip its MFMAs
            if (__builtin_amdgcn_readfirstlane(__float_as_int(Vs[C::XROWS])) == 0) return;
        }
        // lane-half fh takes pixel k = 2*kp + fh.  With MT == 2 a lane fetches the channel pair (2*fr, 2*fr + 1) of its
        // wave's 64-channel strip in one ds_read_b64: MFMA tile i then holds output channels wm + 2*row + i.
        constexpr int RS = NTAP == 3 ? S : 1;                              // staged rows per output pixel
        const float* const dsl = Ds + fh * TM + wm + (MT == 2 ? 2 * fr : fr);
        const float* const xsl = Xs + fh * RS * TN + wn + fr;
        const float* const vsl = Vs + fh * RS;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float a[MT], b[NTAP][NT];
            if (MT == 2) {
                const float2 v = *reinterpret_cast<const float2*>(dsl + 2 * kp * TM);
                a[0] = v.x; a[MT - 1] = v.y;
            } else {
                a[0] = dsl[2 * kp * TM];
            }
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int row = 2 * kp * RS + t * D;                      // compile-time: folded into the ds_read offset
#pragma unroll
                for (int j = 0; j < NT; ++j) b[t][j] = xsl[row * TN + j * 32];
                if (AFFINE) {
                    float v = 1.f;
                    if (NTAP == 3) v = vsl[row];
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float u = fmaxf(__builtin_fmaf(b[t][j], sc[j], sh[j]), 0.f);    // one fused rounding, as the forward's fetch
                        b[t][j] = NTAP == 3 ? u * v : u;                  // zero padding comes after the transform
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[t][j], acc[t][i][j], 0, 0, 0);
        }
    };
    // own DMA of the step about to be read (and the validity words written beside it) complete, then everybody's; past the
    // barrier every wave has also finished reading the other stage, so it is refilled right away
#define WG_SYNC()                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                   \
    asm volatile("" ::: "memory")
    if (T > 0) {
        float* const st0 = smem;
        float* const st1 = smem + C::STAGE_FLOATS;
        step(st0, st1, true, false);
        for (int t = 0; t < T; t += 2) {
            WG_SYNC();
            step(st1, st0, t + 1 < T, true);
            if (t + 1 >= T) break;
            WG_SYNC();
            step(st0, st1, t + 2 < T, true);
        }
    }
#undef WG_SYNC

    // ---- split-K: fp32 atomics into dW[o][tap][i] (accumulator column = lane => 128 contiguous bytes per half wave) ------
    const size_t row_ld = (size_t)p.KH * p.KW * p.Cin;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int tap = NTAP == 3 ? ky * p.KW + t : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float* base = p.dw + (size_t)bi * p.Cout * row_ld + (size_t)tap * p.Cin + (ci0 + wn + j * 32 + fr);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const int co = co0 + wm + (MT == 2 ? 2 * row + i : row);
                    atomicAdd(base + (size_t)co * row_ld, acc[t][i][j][r]);
                }
            }
    }
}

// =========================================================================================================
// bf16-STORAGE weight gradient (BASELINE config 5): x and dY are bf16 in HBM, dW accumulates in fp32 (split-K atomics into
// the fp32 gradient arena, as above).  Same step structure (LDS-DMA, two stages, one barrier per 32-pixel step, three taps
// of a filter row per block on 3x3 layers); there is no operand prologue on the bf16 path (activations are materialised).
// The reduction index is the pixel while a staged row is a pixel's channels, so the MFMA operands -- 8 consecutive pixels
// of one channel per lane -- come out of the k-major image by the hardware transposing read ds_read_b64_tr_b16.  Rows are
// unpadded (the DMA writes linearly): the 16-byte chunks of a row are XOR-swizzled on the SOURCE address and on the read
// so that the 32 eight-byte pieces of one read instruction (4 pixel rows x 2 channel halves x 4 pieces) fall on 32
// different 8-byte slots of the 256-byte bank row.
// =========================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int T>                                                            // T = channels per staged row
__device__ __forceinline__ int wgb_swz(const int row) {
    constexpr int RPB = 128 / T > 1 ? 128 / T : 1;                          // rows per 256-byte bank row (T=128: 1, T=64: 2)
    return ((row / RPB) & (4 / RPB - 1)) << 2;                              // XOR on the 16-byte chunk index
}
// operand fragment of v_mfma_f32_32x32x16_bf16 for the 32 channels at ch0 and the 8 pixels whose staged rows are
// row0, row0 + rs, ..., row0 + 7*rs (row0 already carries the lane half's +8 pixels)
template <int T>
__device__ __forceinline__ bf16x8 wgb_frag(const unsigned short* img, const int row0, const int rs, const int ch0, const int lane) {
    const int q = (lane & 15) >> 2, pp = lane & 3, g16 = (lane >> 4) & 1;
    const int chunk = (ch0 >> 3) + 2 * g16 + (pp >> 1);
    const int r0 = row0 + q * rs, r1 = r0 + 4 * rs;
    const unsigned short* a0 = img + r0 * T + ((chunk ^ wgb_swz<T>(r0)) << 3) + ((pp & 1) << 2);
    const unsigned short* a1 = img + r1 * T + ((chunk ^ wgb_swz<T>(r1)) << 3) + ((pp & 1) << 2);
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int TM, int TN, int NTAP, int S, int D, int BK_ = 32>
struct WgbCfg {
    static constexpr int BK = BK_;                                         // pixels per step (64 on the linear 1x1 layers)
    static constexpr int MT = TM / 64, NT = TN / 64;
    static constexpr int XR = NTAP == 1 ? BK : (BK - 1) * S + 2 * D + 1;
    static constexpr int RPI_D = 512 / TM, RPI_X = 512 / TN;               // rows per 1-KiB wave instruction
    static constexpr int XROWS = (XR + 4 * RPI_X - 1) / (4 * RPI_X) * (4 * RPI_X);
    static constexpr int D_IPW = BK / (4 * RPI_D) > 0 ? BK / (4 * RPI_D) : 1;
    static constexpr int X_IPW = XROWS / (4 * RPI_X);
    static constexpr int D_ELEMS = (BK > 4 * RPI_D ? BK : 4 * RPI_D) * TM, X_ELEMS = XROWS * TN;
    static constexpr int STAGE_ELEMS = D_ELEMS + X_ELEMS + 8;              // + the row flag
    static constexpr int LDS_BYTES = 2 * STAGE_ELEMS * 2;
    static constexpr int ACC = NTAP * MT * NT * 16;
    static constexpr int BPC = ACC <= 64 ? 3 : 2;
};

template <int TM, int TN, int NTAP, int S, int D, bool LINEAR, int BK_ = 32>
__global__ __launch_bounds__(256, (WgbCfg<TM, TN, NTAP, S, D, BK_>::BPC)) void wgrad_bf16_kernel(const WgP p) {
    using C = WgbCfg<TM, TN, NTAP, S, D, BK_>;
    constexpr int MT = C::MT, NT = C::NT, BK = C::BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* const lds16 = reinterpret_cast<unsigned short*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows_k = NTAP == 3 ? p.KH : 1;
    const int ntiles = p.tiles_co * p.tiles_ci * rows_k;
    const int nwg = gridDim.x;
    int lin;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int split = lin / ntiles;
    int t_ = lin - split * ntiles;
    const int ci_t = t_ % p.tiles_ci; t_ /= p.tiles_ci;
    const int ky = t_ % rows_k;
    const int co_t = t_ / rows_k;
    const int co0 = co_t * TM, ci0 = ci_t * TN;
    const int q_beg = split * p.steps_per_split;
    const int q_end = min(p.steps_total, q_beg + p.steps_per_split);
    const int T = q_end - q_beg;
    const i32x4 rs_d = wg_rsrc(p.dy, p.dy_bytes), rs_x = wg_rsrc(p.x, p.x_bytes);

    // per-lane constants of the DMA pattern: lane -> (staged row, 16-byte chunk); the SOURCE chunk is the swizzled one
    unsigned dconst[C::D_IPW], xconst[C::X_IPW];
    int drow[C::D_IPW], xrow[C::X_IPW];
#pragma unroll
    for (int j = 0; j < C::D_IPW; ++j) {
        const int e = ((j * 4 + wave) * 64 + lane) * 8;
        drow[j] = e / TM;
        const int chunk = (e % TM) >> 3;
        dconst[j] = (unsigned)((drow[j] * p.dy_ld + co0 + ((chunk ^ wgb_swz<TM>(drow[j])) << 3)) * 2);
    }
    constexpr int SX = NTAP == 3 ? 1 : S;
#pragma unroll
    for (int j = 0; j < C::X_IPW; ++j) {
        const int e = ((j * 4 + wave) * 64 + lane) * 8;
        xrow[j] = e / TN;
        const int chunk = (e % TN) >> 3;
        xconst[j] = (unsigned)((xrow[j] * SX * p.x_ld + ci0 + ((chunk ^ wgb_swz<TN>(xrow[j])) << 3)) * 2);
    }
    const int chunks = LINEAR ? 1 : p.Wo / BK;
    int qi = q_beg, in_ = 0, ioy = 0, ixc = 0;
    if (!LINEAR) {
        const int rowid = q_beg / chunks;
        ixc = q_beg - rowid * chunks;
        in_ = rowid / p.Ho;
        ioy = rowid - in_ * p.Ho;
    }
    f32x16 acc[NTAP][MT][NT];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    const int fr = lane & 31, fh = lane >> 5;

    auto step = [&](unsigned short* __restrict__ fill, const unsigned short* __restrict__ use, const bool do_issue, const bool do_phase) {
        if (do_issue) {
            unsigned short* const Ds = fill;
            unsigned short* const Xs = fill + C::D_ELEMS;
            unsigned short* const Fl = Xs + C::X_ELEMS;
            const unsigned dbase = (unsigned)qi * (unsigned)(BK * 2) * (unsigned)p.dy_ld;
#pragma unroll
            for (int j = 0; j < C::D_IPW; ++j)
                uem_raw_buffer_load_lds(rs_d, (lds_u32p)(Ds + (j * 4 + wave) * 512), 16,
                                        (int)(drow[j] < BK ? dconst[j] + dbase : WG_OOB), 0, 0, 0);
            if (LINEAR) {
                const unsigned xbase = (unsigned)qi * (unsigned)(BK * 2) * (unsigned)p.x_ld;
#pragma unroll
                for (int j = 0; j < C::X_IPW; ++j)
                    uem_raw_buffer_load_lds(rs_x, (lds_u32p)(Xs + (j * 4 + wave) * 512), 16,
                                            (int)(xrow[j] < C::XR ? xconst[j] + xbase : WG_OOB), 0, 0, 0);
            } else {
                const int iy = ioy * S - p.pad + ky * D;
                const int ix0 = ixc * (BK * S) - p.pad;
                const bool rowok = iy >= 0 && iy < p.H && in_ < p.N;
                const unsigned xbase = (unsigned)(((in_ * p.H + iy) * p.W + ix0) * p.x_ld) * 2u;
#pragma unroll
                for (int j = 0; j < C::X_IPW; ++j) {
                    const int ix = ix0 + xrow[j] * SX;
                    const bool ok = rowok && ix >= 0 && ix < p.W && xrow[j] < C::XR;
                    uem_raw_buffer_load_lds(rs_x, (lds_u32p)(Xs + (j * 4 + wave) * 512), 16, (int)(ok ? xconst[j] + xbase : WG_OOB), 0, 0, 0);
                }
                if (NTAP == 3 && tid == 0) Fl[0] = rowok ? 1 : 0;
                if (++ixc == chunks) { ixc = 0; if (++ioy == p.Ho) { ioy = 0; ++in_; } }
            }
            ++qi;
        }
        if (!do_phase) return;
        const unsigned short* const Ds = use;
        const unsigned short* const Xs = use + C::D_ELEMS;
        const unsigned short* const Fl = Xs + C::X_ELEMS;
        if (NTAP == 3) {
            if (__builtin_amdgcn_readfirstlane((int)Fl[0]) == 0) return;  // filter row in the padding for this output row
        }
        constexpr int RS = NTAP == 3 ? S : 1;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int k0 = ks * 16 + 8 * fh;                               // this lane half's 8 pixels of the MFMA step
            bf16x8 a[MT], b[NTAP][NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = wgb_frag<TM>(Ds, k0, 1, wm + i * 32, lane);
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int j = 0; j < NT; ++j) b[t][j] = wgb_frag<TN>(Xs, k0 * RS + t * D, RS, wn + j * 32, lane);
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[t][j], acc[t][i][j], 0, 0, 0);
        }
    };
#define WG_SYNC()                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                   \
    asm volatile("" ::: "memory")
    if (T > 0) {
        unsigned short* const st0 = lds16;
        unsigned short* const st1 = lds16 + C::STAGE_ELEMS;
        step(st0, st1, true, false);
        for (int t = 0; t < T; t += 2) {
            WG_SYNC();
            step(st1, st0, t + 1 < T, true);
            if (t + 1 >= T) break;
            WG_SYNC();
            step(st0, st1, t + 2 < T, true);
        }
    }
#undef WG_SYNC
    const size_t row_ld = (size_t)p.KH * p.KW * p.Cin;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int tap = NTAP == 3 ? ky * p.KW + t : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float* base = p.dw + (size_t)tap * p.Cin + (ci0 + wn + j * 32 + fr);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    atomicAdd(base + (size_t)co * row_ld, acc[t][i][j][r]);
                }
            }
    }
}

// tuning overrides (scripts/sweep_wgrad.py): tile TM x TN (0 = rule below), split-K rounds (0 = rule)
static int g_tm = 0, g_tn = 0, g_rounds = 0, g_bk = 0;
extern "C" void uemdbg_wgrad_config(int tm, int tn, int rounds, int bk) { g_tm = tm; g_tn = tn; g_rounds = rounds; g_bk = bk; }

template <int BK, int TM, int TN, int NTAP, int S, int D, bool LINEAR>
static void wg_go(WgP p, bool affine, hipStream_t st) {
    using C = WgCfg<BK, TM, TN, NTAP, S, D, false, LINEAR>;
    p.tiles_co = p.Cout / TM;
    p.tiles_ci = p.Cin / TN;
    const int tiles = p.tiles_co * p.tiles_ci * (NTAP == 3 ? p.KH : 1) * p.nbatch;
    p.steps_total = (int)uem_cdiv(p.M, BK);                                // per batch item
    // split-K sizing.  Blocks of one launch run in lock step, so the grid is sized to whole ROUNDS of the chip's resident
    // block slots; every block ends with one fp32-atomic pass over its tile (chip-wide 1.3 TB/s), so the fewest rounds that
    // fill the slots to >= 97 % win (profiles/r02_b_wgrad_sweep.txt: 1 round beats 2 and 3 wherever it fills the chip).
    static const int forced = getenv("UEM_WGRAD_SPLITS") ? atoi(getenv("UEM_WGRAD_SPLITS")) : 0;
    static const int forced_rounds = getenv("UEM_WGRAD_ROUNDS") ? atoi(getenv("UEM_WGRAD_ROUNDS")) : 0;
    const int slots = 256 * C::BPC;
    const int max_splits = (int)uem_cdiv(p.steps_total, 8);                // >= 8 steps per block
    int splits = 1;
    {
        const int fixed = g_rounds > 0 ? g_rounds : forced_rounds;
        double best_fill = -1.0;
        for (int r = fixed > 0 ? fixed : 1; r <= (fixed > 0 ? fixed : 3); ++r) {
            int sp = slots * r / tiles;
            if (sp > max_splits) sp = max_splits;
            if (sp < 1) sp = 1;
            const double fill = (double)tiles * sp / ((double)slots * uem_cdiv((int64_t)tiles * sp, slots));
            if (fill > best_fill + 1e-9) { best_fill = fill; splits = sp; }
            if (fill >= 0.97) break;
        }
    }
    if (forced > 0) splits = forced;
    p.steps_per_split = (int)uem_cdiv(p.steps_total, splits);
    splits = (int)uem_cdiv(p.steps_total, p.steps_per_split);
    const unsigned grid = (unsigned)tiles * (unsigned)splits;
    if (affine) {
        auto k = wgrad_dma_kernel<BK, TM, TN, NTAP, S, D, true, LINEAR>;
        if (uem_allow_lds((const void*)k, C::LDS_BYTES)) k<<<grid, 256, C::LDS_BYTES, st>>>(p);
    } else {
        auto k = wgrad_dma_kernel<BK, TM, TN, NTAP, S, D, false, LINEAR>;
        if (uem_allow_lds((const void*)k, C::LDS_BYTES)) k<<<grid, 256, C::LDS_BYTES, st>>>(p);
    }
}

template <int BK, int TM, int TN>
static bool wg_dispatch(const WgP& p, const uem_conv_shape* s, bool affine, hipStream_t st) {
    if (s->KH == 1 && s->KW == 1 && s->pad == 0) {
        if (s->stride == 1) { wg_go<BK, TM, TN, 1, 1, 0, true>(p, affine, st); return true; }
        if (s->stride == 2 && s->Wo % BK == 0) { wg_go<BK, TM, TN, 1, 2, 0, false>(p, affine, st); return true; }
        return false;
    }
    if (s->KH == 3 && s->KW == 3 && s->Wo % BK == 0) {
        if (s->stride == 1 && s->dil == 1) { wg_go<BK, TM, TN, 3, 1, 1, false>(p, affine, st); return true; }
        if (s->stride == 1 && s->dil == 2) { wg_go<BK, TM, TN, 3, 1, 2, false>(p, affine, st); return true; }
        if (s->stride == 2 && s->dil == 1) { wg_go<BK, TM, TN, 3, 2, 1, false>(p, affine, st); return true; }
    }
    return false;
}

// Returns 1 when the LDS-DMA kernel took the launch, 0 when the shape is left to the register-staged kernel (conv.hip).
int uem_wgrad_dma_try(const float* x, const float* dy, const float* in_scale, const float* in_shift, float* dw,
                      const uem_conv_shape* s, int flags, hipStream_t st) {
    static const int off = getenv("UEM_WGRAD_DMA") ? !atoi(getenv("UEM_WGRAD_DMA")) : 0;
    if (off) return 0;
    const bool affine = (flags & UEM_CONV_IN_AFFINE) != 0;
    if (flags & UEM_CONV_PREC_BF16) return 0;
    if (affine && !(flags & UEM_CONV_IN_RELU)) return 0;
    if (s->Cout % 64 != 0 || s->Cin % 64 != 0 || s->x_ld % 4 != 0 || s->y_ld % 4 != 0) return 0;
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return 0;
    const double xb = (double)s->N * s->H * s->W * s->x_ld * 4.0, db = (double)s->N * s->Ho * s->Wo * s->y_ld * 4.0;
    if (xb >= 4294967280.0 || db >= 4294967280.0) return 0;               // 32-bit buffer offsets
    WgP p;
    p.x = x; p.dy = dy; p.in_scale = in_scale; p.in_shift = in_shift; p.dw = dw;
    p.M = s->N * s->Ho * s->Wo; p.N = s->N; p.H = s->H; p.W = s->W; p.Cin = s->Cin; p.Ho = s->Ho; p.Wo = s->Wo; p.Cout = s->Cout;
    p.KH = s->KH; p.KW = s->KW; p.pad = s->pad; p.x_ld = s->x_ld; p.dy_ld = s->y_ld;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)db; p.nbatch = 1;
    // tile rule (profiles/r02_b_wgrad_sweep.txt): 128 output channels x 64 input channels everywhere -- the narrower input
    // tile halves the LDS footprint (3 resident blocks per CU instead of 2 on the 1x1 layers) and the per-block atomic pass
    const bool m128 = s->Cout % 128 == 0 && g_tm != 64, n128 = s->Cin % 128 == 0 && g_tn == 128;
    bool ok;
    // 32-pixel steps (profiles/r02_b_wgrad_sweep.txt: 16-pixel steps with twice the resident blocks are 2-4 % slower); rows
    // that are a multiple of 16 but not of 32 pixels (256x256 tiles at output stride 16) take the 16-pixel instantiation
    const bool pointwise_s1 = s->KH == 1 && s->KW == 1 && s->stride == 1;
    if (g_bk == 16 || (g_bk == 0 && !pointwise_s1 && s->Wo % 32 != 0)) {
        if (m128 && n128) ok = wg_dispatch<16, 128, 128>(p, s, affine, st);
        else if (m128) ok = wg_dispatch<16, 128, 64>(p, s, affine, st);
        else if (n128) ok = wg_dispatch<16, 64, 128>(p, s, affine, st);
        else ok = wg_dispatch<16, 64, 64>(p, s, affine, st);
        return ok ? 1 : 0;
    }
    if (m128 && n128) ok = wg_dispatch<32, 128, 128>(p, s, affine, st);
    else if (m128) ok = wg_dispatch<32, 128, 64>(p, s, affine, st);
    else if (n128) ok = wg_dispatch<32, 64, 128>(p, s, affine, st);
    else ok = wg_dispatch<32, 64, 64>(p, s, affine, st);
    return ok ? 1 : 0;
}

// Winograd weight gradient (winograd.hip): dU[pos][n][k] += sum over the T tiles of dM[pos][tile][n] * V[pos][tile][k], the npos (16 / 36)
// positions as ONE batched launch of the linear (1x1) kernel; dU (npos, N, K) must be zeroed by the caller (split-K atomics land in it).
extern "C" int uem_wino_wgrad_gemm(const float* V, const float* dM, float* dU, int T, int K, int N, int npos, void* stream) {
    UEM_REQUIRE(V && dM && dU && T > 0 && K > 0 && N > 0 && (npos == 16 || npos == 36), "wino_wgrad_gemm: bad arguments");
    if (T % 32 != 0 || K % 64 != 0 || N % 64 != 0 || (((uintptr_t)V | (uintptr_t)dM) & 15))
        return uem_fail(UEM_ERR_UNSUPPORTED, "wino_wgrad_gemm: needs T %% 32 == 0, K %% 64 == 0, N %% 64 == 0, 16-byte aligned operands");
    const double xb = (double)npos * T * K * 4.0, db = (double)npos * T * N * 4.0;
    if (xb >= 4294967280.0 || db >= 4294967280.0) return uem_fail(UEM_ERR_UNSUPPORTED, "wino_wgrad_gemm: tensor beyond 32-bit byte offsets");
    WgP p;
    p.x = V; p.dy = dM; p.in_scale = p.in_shift = nullptr; p.dw = dU;
    p.M = T; p.N = 1; p.H = 1; p.W = T; p.Cin = K; p.Ho = 1; p.Wo = T; p.Cout = N;
    p.KH = p.KW = 1; p.pad = 0; p.x_ld = K; p.dy_ld = N;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)db; p.nbatch = npos;
    hipStream_t st = (hipStream_t)stream;
    if (N % 128 == 0) wg_go<32, 128, 64, 1, 1, 0, true>(p, false, st);
    else wg_go<32, 64, 64, 1, 1, 0, true>(p, false, st);
    return uem_check_launch("wino_wgrad_gemm");
}

// ---- bf16-storage weight gradient: launcher ------------------------------------------------------------------
template <int TM, int TN, int NTAP, int S, int D, bool LINEAR, int BK_ = 32>
static void wgb_go(WgP p, hipStream_t st) {
    using C = WgbCfg<TM, TN, NTAP, S, D, BK_>;
    p.tiles_co = p.Cout / TM;
    p.tiles_ci = p.Cin / TN;
    const int tiles = p.tiles_co * p.tiles_ci * (NTAP == 3 ? p.KH : 1);
    p.steps_total = (int)uem_cdiv(p.M, C::BK);
    const int slots = 256 * C::BPC;
    const int max_splits = (int)uem_cdiv(p.steps_total, 8);
    int splits = 1;
    double best_fill = -1.0;
    for (int r = 1; r <= 3; ++r) {
        int sp = slots * r / tiles;
        if (sp > max_splits) sp = max_splits;
        if (sp < 1) sp = 1;
        const double fill = (double)tiles * sp / ((double)slots * uem_cdiv((int64_t)tiles * sp, slots));
        if (fill > best_fill + 1e-9) { best_fill = fill; splits = sp; }
        if (fill >= 0.97) break;
    }
    p.steps_per_split = (int)uem_cdiv(p.steps_total, splits);
    splits = (int)uem_cdiv(p.steps_total, p.steps_per_split);
    const unsigned grid = (unsigned)tiles * (unsigned)splits;
    auto k = wgrad_bf16_kernel<TM, TN, NTAP, S, D, LINEAR, BK_>;
    if (uem_allow_lds((const void*)k, C::LDS_BYTES)) k<<<grid, 256, C::LDS_BYTES, st>>>(p);
}
template <int TM, int TN>
static bool wgb_dispatch(const WgP& p, const uem_conv_shape* s, hipStream_t st) {
    if (s->KH == 1 && s->KW == 1 && s->pad == 0) {
        if (s->stride == 1) {
            // 64-pixel steps on the linear 1x1 layers: a 32-pixel step is two MFMA k-steps per barrier (4 MFMAs per wave),
            // too little work per step for layers that only have to stream their operands
            static const int bk64 = getenv("UEM_WGRAD_BF16_BK64") ? atoi(getenv("UEM_WGRAD_BF16_BK64")) : 1;
            if (bk64) wgb_go<TM, TN, 1, 1, 0, true, 64>(p, st);
            else wgb_go<TM, TN, 1, 1, 0, true>(p, st);
            return true;
        }
        if (s->stride == 2 && s->Wo % 32 == 0) { wgb_go<TM, TN, 1, 2, 0, false>(p, st); return true; }
        return false;
    }
    if (s->KH == 3 && s->KW == 3 && s->Wo % 32 == 0) {
        if (s->stride == 1 && s->dil == 1) { wgb_go<TM, TN, 3, 1, 1, false>(p, st); return true; }
        if (s->stride == 1 && s->dil == 2) { wgb_go<TM, TN, 3, 1, 2, false>(p, st); return true; }
        if (s->stride == 2 && s->dil == 1) { wgb_go<TM, TN, 3, 2, 1, false>(p, st); return true; }
    }
    return false;
}
extern "C" int uem_conv2d_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dw, const uem_conv_shape* s, void* stream) {
    UEM_REQUIRE(x && dy && dw && s, "conv2d_wgrad_bf16: null pointer");
    if (s->Cout % 64 != 0 || s->Cin % 64 != 0 || s->x_ld % 8 != 0 || s->y_ld % 8 != 0 || (((uintptr_t)x | (uintptr_t)dy) & 15))
        return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_wgrad_bf16: needs channel counts %% 64 == 0 and 16-byte rows");
    const double xb = (double)s->N * s->H * s->W * s->x_ld * 2.0, db = (double)s->N * s->Ho * s->Wo * s->y_ld * 2.0;
    if (xb >= 4294967280.0 || db >= 4294967280.0) return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_wgrad_bf16: tensor beyond 32-bit buffer offsets");
    WgP p;
    p.x = (const float*)x; p.dy = (const float*)dy; p.in_scale = p.in_shift = nullptr; p.dw = dw;
    p.M = s->N * s->Ho * s->Wo; p.N = s->N; p.H = s->H; p.W = s->W; p.Cin = s->Cin; p.Ho = s->Ho; p.Wo = s->Wo; p.Cout = s->Cout;
    p.KH = s->KH; p.KW = s->KW; p.pad = s->pad; p.x_ld = s->x_ld; p.dy_ld = s->y_ld;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)db; p.nbatch = 1;
    // 128 x 128 tiles on the largest pointwise filter banks (Cin * Cout >= 2^20: layer4's 2048 <-> 512 and its 1024 -> 2048 downsample,
    // -4 ... -15 % in the step; the smaller banks lose 15-30 % on them: fewer tiles, longer split-K slices).  UEM_WGRAD_BF16_TN128 = 0
    // off, 1 every pointwise layer, else the threshold on Cin * Cout.
    // Round 6: with 131072 pixels and more per bank (the 1024 x 1024 configuration's layer3) the split-K slices are long enough for the
    // mid-sized banks too (Cin * Cout >= 2^18: 1024 <-> 256): R101-1024^2 step 193.0 -> 190.5 ms, A-B-A on one box; 2^16 / 2^14 add
    // nothing measurable (190.6 / 191.1 against 191.1-191.3).  At 512 x 512 those banks see 32768 pixels and keep 128 x 64.
    static const int tn128_env = getenv("UEM_WGRAD_BF16_TN128") ? atoi(getenv("UEM_WGRAD_BF16_TN128")) : -1;
    const int tn128 = tn128_env >= 0 ? tn128_env : (p.M >= 131072 ? (1 << 18) : (1 << 20));
    const bool pw = s->KH == 1 && s->KW == 1 && s->pad == 0 && s->stride == 1;
    if (tn128 && pw && s->Cout % 128 == 0 && s->Cin % 128 == 0 && (tn128 == 1 || s->Cin * s->Cout >= tn128)) {
        wgb_go<128, 128, 1, 1, 0, true, 64>(p, (hipStream_t)stream);     // the only 128 x 128 instantiation: linear 1x1, 64-pixel steps
        return uem_check_launch("conv2d_wgrad_bf16");
    }
    const bool ok = s->Cout % 128 == 0 ? wgb_dispatch<128, 64>(p, s, (hipStream_t)stream) : wgb_dispatch<64, 64>(p, s, (hipStream_t)stream);
    if (!ok) return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_wgrad_bf16: 1x1 (stride 1, or 2 with rows of 32 pixels) and 3x3 on rows of 32 pixels only");
    return uem_check_launch("conv2d_wgrad_bf16");
}
