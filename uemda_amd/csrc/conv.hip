// Convolution as implicit GEMM on the exact-fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain).
//
//   forward / data-gradient:  Y[m][n] = sum_k A[m][k] * W[n][k]
//        m = output pixel (NHWC => dense row index), n = output channel, k = (tap, input channel);
//        A is gathered on the fly from the NHWC input (im2col never materialised), optionally
//        pushed through the producer's BatchNorm affine + ReLU while it is staged (operand prologue).
//   weight-gradient:          dW[o][tap][i] += sum_m dY[m][o] * A[m][tap, i]      (split-K, fp32 atomics)
//
// Block = 256 threads = 4 waves; BM = 128 output pixels x BN in {128, 64, 32} channels x BK = 32.
// Global -> register prefetch of tile k+1 overlaps the MFMA phase of tile k; LDS tiles are [row][36]
// floats (k contiguous, 16-B pad) read with ds_read_b128 (conflict-free).
// Replaces cuDNN behind nn.Conv2d: reference _resnets.py:95-110,149,209; Encoder.py:19,35-36,40,74-75.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BK 32
#define LDS_LD 36   // floats per LDS row: 32 + 4 pad => 144-B stride, conflict-free b128 reads

struct ConvP {
    const float* x;
    const float* w;
    const float* bias;
    const float* in_scale;
    const float* in_shift;
    float* y;
    int M;                 // rows of the GEMM = output pixels
    int N, H, W, Cin;      // "input" tensor the gather reads (for TRANSPOSED: dY dims Ho,Wo,Cout)
    int Ho, Wo, Cout;      // "output" tensor (for TRANSPOSED: dX dims H,W,Cin)
    int KH, KW, stride, pad, dil;
    int x_ld, y_ld;
    int accumulate;
    int relu;
    float* tile_stats;     // optional [2][Cout][tiles_m]: per-128-row-tile column sums of y and y*y (fused BN statistics)
    // data-gradient only: fused first pass of the BatchNorm(+ReLU) backward of the layer this gradient feeds.
    // y here is dA (grad wrt the post-ReLU activation); with z = that layer's raw conv output the epilogue also
    // leaves per-tile column sums of dp = dA*[z*sc+sh > 0] and dp*xhat in tile_bnbwd[2][Cout][tiles_m].
    const float* bn_z;
    const float* bn_vec;   // (4, Cout): scale, shift, mean, invstd
    float* tile_bnbwd;
    // residual tail of a bottleneck block's backward (uem_conv2d_dgrad_tail): the tensor accumulated into is acc_src gated
    // by the packed ReLU bits acc_bits (the identity gradient dy*[y>0], never materialised), and the fused BatchNorm reduction
    // takes its ReLU mask from the packed bits bn_bits (the PREVIOUS block's output mask) instead of recomputing it from z
    const float* acc_src;
    const uint32_t* acc_bits;
    const uint32_t* bn_bits;
    // strided data-gradient only: one launch per output-parity class (py, px); rows enumerate the pixels
    // (sub*yy + py, sub*xx + px) and only the taps that can reach that class are walked.
    int sub, py, px, Hs, Ws;          // sub == 1: dense rows (every other use)
    int ntaps;                        // number of taps walked
    // grouped pointwise GEMM (Winograd, uem_wino_gemm): GEMM rows [g*wg_rows, (g+1)*wg_rows) use the filter bank at w + g*wg_stride
    // bytes; wg_rows == 0: one filter bank
    int wg_rows;
    unsigned wg_stride;
    int dbg;                          // diagnostic builds of the schedule (uemdbg_conv_dbg); 0 in production
    unsigned long long tapmask;       // 4 bits per walked tap: tap id = ky*KW + kx (3x3 at most)
    int y_bf16;                       // stem only (MODE 2, full tiles): y is a bf16 tensor -- values rounded (RNE) at the store, the
                                      // BatchNorm tile statistics taken over the ROUNDED values (what the max-pool will read)
    int lazy;                         // conv_bf16_kernel, persistent blocks: counted waits around the epilogue (see there); read by that kernel only
};

// Schedule-ablation hooks (diagnostic builds that skip loads / stores and give WRONG results) exist only under
// -DUEM_DEBUG_HOOKS (`make DEBUG_HOOKS=1`): the shipped library cannot be switched into a wrong-result mode.
#ifdef UEM_DEBUG_HOOKS
#define UEM_DBG(v) (v)
static int g_conv_dbg = 0;
extern "C" void uemdbg_conv_dbg(int v) { g_conv_dbg = v; }
#else
#define UEM_DBG(v) 0
static constexpr int g_conv_dbg = 0;
#endif

// bijective XCD-aware remap (cdna guide T1): blocks sharing an XCD get consecutive tile ids
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Accumulators -> global memory (shared by the register-staged and the LDS-DMA main loops): C/D layout
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  `smem` must be free (every wave past its last operand read).
// EPI fixes the epilogue's options at compile time where the launcher knows them (conv_dma_go): 0 = full dense tiles, plain
// store; 1 = full dense tiles with BatchNorm tile statistics (forward) / the residual-tail and BatchNorm-backward options
// (data gradient); -1 = everything tested at run time (bias, ragged tiles, strided rows, forward accumulate).
#ifndef EPI_RB
#define EPI_RB 8
#endif
template <int BN, int WM, int WN, int MODE, int EPI = -1>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                              const int m0, const int n0) {
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    constexpr bool FULL = EPI >= 0;
    const bool acc_on = (EPI == 0 || (EPI == 1 && MODE != 1)) ? false : p.accumulate != 0;
    const bool stats_on = MODE != 1 && (EPI == 1 || (EPI < 0 && p.tile_stats != nullptr));
    const bool bnbwd_on = MODE == 1 && EPI != 0 && p.tile_bnbwd != nullptr;
    const float* const bias = FULL ? nullptr : p.bias;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WN) * (BM / WM), wn = (wave % WN) * (BN / WN);
    const int fr = lane & 31, fh = lane >> 5;
    const bool dense_rows = FULL || !(MODE == 1 && p.sub > 1);
    // strided data gradient (one launch per output-parity class): GEMM row m is pixel (sub*yy + py, sub*xx + px) of its image,
    // so a staged row still leaves as 16-byte stores, only its base offset is computed per row
    auto row_off = [&](const int m) -> size_t {
        if (dense_rows) return (size_t)m * p.y_ld;
        const int hw = p.Hs * p.Ws;
        const int ni = m / hw, rem = m - ni * hw;
        const int yy = rem / p.Ws, xx = rem - yy * p.Ws;
        return (((size_t)ni * p.Ho + (size_t)(yy * p.sub + p.py)) * p.Wo + (size_t)(xx * p.sub + p.px)) * p.y_ld;
    };
    if (FULL || ((dense_rows || (p.tile_stats == nullptr && p.tile_bnbwd == nullptr)) && m0 + BM <= p.M && n0 + BN <= p.Cout)) {
        // Full tile: the accumulators go through LDS (the operand stages are dead now) in two 64-row halves and
        // leave as 16-byte stores, 32 lanes per 512-B row segment.  (64 dword stores per lane made the epilogue
        // store-issue bound on the small-K layers.)  The staged half is also where the fused BatchNorm statistics
        // (column sums of y and y*y over the tile) are taken: consecutive threads on consecutive banks.
        constexpr int LDW = BN + 4;                     // staged row stride (floats), keeps 16-B alignment
        constexpr int TPR = BN / 4;                     // threads per staged row
        constexpr int RPP = 256 / TPR;                  // rows per store pass
        float* const stg = smem;                        // 64 x LDW floats <= the A+B stages
        const int srow = tid / TPR, sc4 = (tid % TPR) * 4;
        float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv4 = *reinterpret_cast<const float4*>(bias + n0 + sc4);
        // per-thread column partials over this thread's rows (4 columns): BatchNorm statistics sum y / sum y*y (forward) or
        // the BatchNorm-backward partials sum dp / sum dp*xhat (data gradient); combined across thread rows at the end
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), bg = bb;
#pragma unroll
        for (int hm = 0; hm < 2; ++hm) {
            if (wm / 64 == hm) {
                const int rbase = wm % 64;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stg[(rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDW + wn + j * 32 + fr] = acc[i][j][r];
            }
            __syncthreads();
            // global loads of the epilogue (the tensor being accumulated into, its gate bits, the z of the fused BatchNorm backward
            // and its mask bits) are issued together, in batches of EPI_RB rows, before their consumers: written row by row they
            // compiled to load - s_waitcnt vmcnt(0) - use, 8 to 16 serialised memory latencies per block; with the z loads in a
            // second loop behind the store loop (round 2) a half still waited four times, and on the residual tails of layer1-3 --
            // three 4C-wide streams around a K <= 256 product -- the kernel moved bytes at 1.7-3.0 TB/s
            constexpr int NRP = 64 / RPP, RBW = (MODE == 1 && EPI == 1) ? EPI_RB : 4, RB = NRP < RBW ? NRP : RBW;
            float4 sc, sh, mu, is;
            if (bnbwd_on) {
                sc = *reinterpret_cast<const float4*>(p.bn_vec + n0 + sc4);
                sh = *reinterpret_cast<const float4*>(p.bn_vec + p.Cout + n0 + sc4);
                mu = *reinterpret_cast<const float4*>(p.bn_vec + 2 * p.Cout + n0 + sc4);
                is = *reinterpret_cast<const float4*>(p.bn_vec + 3 * p.Cout + n0 + sc4);
            }
#pragma unroll
            for (int rp0 = 0; rp0 < NRP; rp0 += RB) {
                // Round 4: the packed gate / mask words are kept RAW and shifted where they are used.  `bits[off >> 5] >> (off & 31)`
                // at the load is a use of the word just requested: the compiler put an s_waitcnt vmcnt(0) behind every row's pair of
                // loads -- eight serialised memory round trips per half on the residual tails (layer1 0.424 -> 0.381 ms, layer2 0.268
                // -> 0.244, layer3 0.198 -> 0.193 in the step).  The word loads are unconditional too (a null bit pointer reads word 0
                // of the filter bank and ignores it): a uniform branch per row split the batch into basic blocks the wait-count pass
                // drained one by one.
                float4 o[RB], zz[RB];
                uint32_t ob[RB], zb[RB];
                const bool has_ab = p.acc_bits != nullptr, has_zb = p.bn_bits != nullptr;
                const uint32_t* const abp = has_ab ? p.acc_bits : reinterpret_cast<const uint32_t*>(p.w);
                const uint32_t* const zbp = has_zb ? p.bn_bits : reinterpret_cast<const uint32_t*>(p.w);
                if (acc_on) {
                    const float* const asrc = p.acc_src ? p.acc_src : p.y;
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const size_t off = row_off(m0 + hm * 64 + srow + (rp0 + u) * RPP) + n0 + sc4;
                        o[u] = *reinterpret_cast<const float4*>(asrc + off);
                        ob[u] = abp[has_ab ? off >> 5 : (size_t)0];
                    }
                }
                if (bnbwd_on) {                         // dense rows whenever bn_z is given
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const size_t off = (size_t)(m0 + hm * 64 + srow + (rp0 + u) * RPP) * p.y_ld + n0 + sc4;
                        zz[u] = *reinterpret_cast<const float4*>(p.bn_z + off);
                        zb[u] = zbp[has_zb ? off >> 5 : (size_t)0];
                    }
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int row = srow + (rp0 + u) * RPP;
                    float4 v = *reinterpret_cast<const float4*>(&stg[row * LDW + sc4]);
                    v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
                    if (acc_on) {
                        if (has_ab) {
                            const uint32_t m = ob[u] >> ((unsigned)(row_off(m0 + hm * 64 + row) + n0 + sc4) & 31u);
                            o[u].x = (m & 1u) ? o[u].x : 0.f; o[u].y = (m & 2u) ? o[u].y : 0.f;
                            o[u].z = (m & 4u) ? o[u].z : 0.f; o[u].w = (m & 8u) ? o[u].w : 0.f;
                        }
                        v.x += o[u].x; v.y += o[u].y; v.z += o[u].z; v.w += o[u].w;
                    }
                    if (MODE == 2 && p.y_bf16) {
                        const __bf16 b0 = (__bf16)v.x, b1 = (__bf16)v.y, b2 = (__bf16)v.z, b3 = (__bf16)v.w;
                        const unsigned u0 = __builtin_bit_cast(unsigned short, b0), u1 = __builtin_bit_cast(unsigned short, b1);
                        const unsigned u2 = __builtin_bit_cast(unsigned short, b2), u3 = __builtin_bit_cast(unsigned short, b3);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + row_off(m0 + hm * 64 + row) + n0 + sc4) =
                            make_uint2(u0 | (u1 << 16), u2 | (u3 << 16));
                        v = make_float4(__uint_as_float(u0 << 16), __uint_as_float(u1 << 16), __uint_as_float(u2 << 16), __uint_as_float(u3 << 16));
                    } else {
                        *reinterpret_cast<float4*>(p.y + row_off(m0 + hm * 64 + row) + n0 + sc4) = v;
                    }
                    if (stats_on) {
                        bb.x += v.x; bb.y += v.y; bb.z += v.z; bb.w += v.w;
                        bg.x = fmaf(v.x, v.x, bg.x); bg.y = fmaf(v.y, v.y, bg.y); bg.z = fmaf(v.z, v.z, bg.z); bg.w = fmaf(v.w, v.w, bg.w);
                    }
                    if (bnbwd_on) {
                        // dbeta / dgamma partials of the BatchNorm(+ReLU) behind this tensor, over the FINAL gradient v: each
                        // thread owns 4 columns x (64/RPP) rows of this half
                        const float4 z = zz[u];
                        bool kx, ky, kz, kw;
                        if (has_zb) {
                            const uint32_t m = zb[u] >> ((unsigned)((size_t)(m0 + hm * 64 + row) * p.y_ld + n0 + sc4) & 31u);
                            kx = m & 1u; ky = m & 2u; kz = m & 4u; kw = m & 8u;
                        }
                        else { kx = z.x * sc.x + sh.x > 0.f; ky = z.y * sc.y + sh.y > 0.f; kz = z.z * sc.z + sh.z > 0.f; kw = z.w * sc.w + sh.w > 0.f; }
                        const float dx_ = kx ? v.x : 0.f, dy_ = ky ? v.y : 0.f, dz_ = kz ? v.z : 0.f, dw_ = kw ? v.w : 0.f;
                        bb.x += dx_; bb.y += dy_; bb.z += dz_; bb.w += dw_;
                        bg.x += dx_ * ((z.x - mu.x) * is.x); bg.y += dy_ * ((z.y - mu.y) * is.y);
                        bg.z += dz_ * ((z.z - mu.z) * is.z); bg.w += dw_ * ((z.w - mu.w) * is.w);
                    }
                }
            }
            __syncthreads();
        }
        float* const tile_out = bnbwd_on ? p.tile_bnbwd : (stats_on ? p.tile_stats : nullptr);
        if (tile_out != nullptr) {
            // RPP thread rows hold partials of the same 4 columns: combine through LDS (stg is free again); channel-major
            // [2][Cout][tiles] output, so the per-channel finalize streams contiguous rows
            float* red = stg;                           // [2][RPP][BN]
            *reinterpret_cast<float4*>(&red[(0 * RPP + srow) * BN + sc4]) = bb;
            *reinterpret_cast<float4*>(&red[(1 * RPP + srow) * BN + sc4]) = bg;
            __syncthreads();
            if (tid < 2 * BN) {
                const int which = tid / BN, col = tid % BN;
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < RPP; ++q) a += red[(which * RPP + q) * BN + col];
                const size_t tiles_m = (size_t)((p.M + BM - 1) / BM);
                tile_out[((size_t)which * p.Cout + n0 + col) * tiles_m + (size_t)(m0 / BM)] = a;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn + j * 32 + fr;
        if (n >= p.Cout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m < p.M) {
                    size_t opix = (size_t)m;
                    if (!dense_rows) {
                        const int hw = p.Hs * p.Ws;
                        const int ni = m / hw, rem = m - ni * hw;
                        const int yy = rem / p.Ws, xx = rem - yy * p.Ws;
                        opix = ((size_t)ni * p.Ho + (size_t)(yy * p.sub + p.py)) * p.Wo + (size_t)(xx * p.sub + p.px);
                    }
                    float* dst = p.y + opix * p.y_ld + n;
                    float v = acc[i][j][r] + bv;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// MODE: 0 = forward gather, 1 = transposed gather (data gradient), 2 = stem (NHWC4, 8 px x 4 ch per tap row)
// NBUF: 1 = one LDS stage (two barriers per k-step, 3 blocks/CU); 2 = two LDS stages (one barrier per k-step,
//       74 KB => 2 blocks/CU, so power-of-two grids fill the 512 block slots in whole rounds)
// PREC: 0 = exact fp32 (v_mfma_f32_32x32x2_f32); 1 = 3xbf16 split: x = hi + lo (two bf16), the product keeps
//       hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-16 relative per product, 5.3x
//       fewer matrix-pipe cycles); 2 = plain bf16 operands (hi*hi only).  Opt-in (UEM_CONV_PREC_* flags).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LDS_LDH 40   // bf16 elements per LDS row in the split modes: 32 + 8 pad => 80-B stride, conflict-free b128 reads

template <int BN, int WM, int WN, int MODE, bool AFFINE, int NBUF, int PREC>
__global__ __launch_bounds__(256, NBUF == 1 ? 3 : 2) void conv_fwd_kernel(const ConvP p) {
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    constexpr int BROWS = BN / 32;                      // B-tile rows per thread
    // stage sizes in floats; split modes hold a hi and a lo bf16 image of each tile (same bytes + padding)
    constexpr int A_SZ = PREC == 0 ? BM * LDS_LD : BM * LDS_LDH, B_SZ = PREC == 0 ? BN * LDS_LD : BN * LDS_LDH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As0 = smem;                            // [NBUF][BM][LDS_LD]
    float* const Bs0 = smem + NBUF * A_SZ;              // [NBUF][BN][LDS_LD]
    float* const Ssc = Bs0 + NBUF * B_SZ;               // AFFINE only: [Cin] scale, then [Cin] shift
    if (AFFINE) {
        for (int i = threadIdx.x; i < p.Cin; i += 256) { Ssc[i] = p.in_scale[i]; Ssc[p.Cin + i] = p.in_shift[i]; }
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;    // loader: 8 threads per 128-B row

    // ---- per-thread gather geometry for its 4 A rows ------------------------------------------------
    int gy[4], gx[4], gpix[4];                          // gpix < 0 => row beyond M
    {
        const int HoWo = (MODE == 1) ? p.Hs * p.Ws : p.Ho * p.Wo;
        const int Wrow = (MODE == 1) ? p.Ws : p.Wo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + lrow + 32 * j;
            if (m < p.M) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int oy = rem / Wrow, ox = rem - oy * Wrow;
                if (MODE == 1) { gy[j] = oy * p.sub + p.py + p.pad; gx[j] = ox * p.sub + p.px + p.pad; }
                else { gy[j] = oy * p.stride - p.pad; gx[j] = ox * p.stride - p.pad; }
                gpix[j] = n * p.H * p.W;
            } else { gy[j] = gx[j] = 0; gpix[j] = -1; }
        }
    }
    const int cpb = p.Cin / BK;                         // channel blocks per tap
    const int KT = p.ntaps * cpb;
    const int Ktot = p.KH * p.KW * p.Cin;               // row length of the filter bank

    float4 ra[4], rb[BROWS];
    int pci0 = 0;                                       // first channel of the tile in flight (prologue operands)
    unsigned okmask = 0;                                // bit j: row j of the tile in flight is in bounds
    // Loader state.  Everything that depends on the tap only (bounds, row base pointers, filter column) is
    // set up ONCE per tap; the Cin/32 channel blocks of that tap then advance the pointers by 32 floats.
    int lt = 0, lci0 = 0;                               // tap slot, first channel of the next tile
    unsigned tapok = 0;                                 // bit j: row j is in bounds for the current tap
    unsigned aoff[4], boff[BROWS];                      // element offsets from p.x / p.w (every tensor here < 2^32 floats)
#pragma unroll
    for (int j = 0; j < 4; ++j) aoff[j] = 0;
    unsigned bvalid = 0;                                // bit j: filter row j of this thread exists (ragged Cout)
#pragma unroll
    for (int j = 0; j < BROWS; ++j) { boff[j] = 0; bvalid |= (n0 + lrow + 32 * j < p.Cout ? 1u : 0u) << j; }
    auto setup_tap = [&](int t) {
        const int tap = (MODE == 1) ? (int)((p.tapmask >> (4 * t)) & 0xF) : t;
        const int ky = (p.KW == 1) ? tap : ((p.KW == 3) ? (tap * 11) >> 5 : tap / p.KW);   // tap/3 for tap < 16
        const int kx = tap - ky * p.KW;
        tapok = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = gpix[j] >= 0;
            int iy, ix;
            if (MODE == 1) {
                const int ty = gy[j] - ky * p.dil, tx = gx[j] - kx * p.dil;
                if (p.stride == 1) { iy = ty; ix = tx; }
                else { iy = ty / p.stride; ix = tx / p.stride; ok = ok && (iy * p.stride == ty) && (ix * p.stride == tx); }
                ok = ok && ty >= 0 && tx >= 0 && iy < p.H && ix < p.W;
            } else if (MODE == 2) {
                iy = gy[j] + ky;                        // stem: tap = ky, 8 pixels along x in the row
                ix = gx[j] + (lc4 >> 2);
                ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            } else {
                iy = gy[j] + ky * p.dil; ix = gx[j] + kx * p.dil;
                ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            }
            if (ok) {
                const unsigned pix = (unsigned)(gpix[j] + iy * p.W + ix);
                aoff[j] = (MODE == 2) ? pix * 4u : pix * (unsigned)p.x_ld + (unsigned)lc4;
                tapok |= 1u << j;
            }
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const int n = n0 + lrow + 32 * j;
            boff[j] = (unsigned)(n < p.Cout ? n : 0) * (unsigned)Ktot + (unsigned)(tap * p.Cin + lc4);
        }
    };
    auto load_tiles = [&](int kt) {
        (void)kt;                                       // tiles are walked strictly in order
        if (lci0 == 0) setup_tap(lt);                   // block-uniform branch
        pci0 = lci0;
        okmask = tapok;
        // Branch-free: a row that is out of bounds loads from offset 0 (always mapped) and is zeroed when the tile
        // goes to LDS (okmask / bvalid).  A predicated load (`ok ? load : 0`) costs an exec-mask region per row, and
        // the compiler likes to sink the consumer of the value into that region -- an s_waitcnt vmcnt(0) behind
        // every single load.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const float4*>(p.x + (((tapok >> j) & 1u) ? aoff[j] : 0u));
            aoff[j] += BK;
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            rb[j] = *reinterpret_cast<const float4*>(p.w + (((bvalid >> j) & 1u) ? boff[j] : 0u));
            boff[j] += BK;
        }
        lci0 += BK;
        if (lci0 >= p.Cin) { lci0 = 0; ++lt; }
    };
    // the BatchNorm-affine + ReLU prologue is applied here, AFTER the MFMA phase the loads overlapped with
    // (touching the loaded registers earlier would force the s_waitcnt vmcnt in front of the MFMAs);
    // zero padding stays zero: rows that were out of bounds are not transformed.
    auto store_tiles = [&](int buf) {
        float* const As = As0 + buf * A_SZ;
        float* const Bs = Bs0 + buf * B_SZ;
        float4 psc, psh;
        if (AFFINE) {                                   // per-channel operands live in LDS, not in registers
            psc = *reinterpret_cast<const float4*>(&Ssc[pci0 + lc4]);
            psh = *reinterpret_cast<const float4*>(&Ssc[p.Cin + pci0 + lc4]);
        }
        auto put = [&](float* stage, int row, float4 v) {
            if constexpr (PREC == 0) {
                *reinterpret_cast<float4*>(&stage[row * LDS_LD + lc4]) = v;
            } else {
                // one bf16 image per stage ([rows][LDS_LDH])
                __bf16* hi = reinterpret_cast<__bf16*>(stage);
                bf16x4 h;
                h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
                *reinterpret_cast<bf16x4*>(&hi[row * LDS_LDH + lc4]) = h;
            }
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = ra[j];
            const bool ok = (okmask >> j) & 1u;
            if (AFFINE) {
                v.x = v.x * psc.x + psh.x; v.y = v.y * psc.y + psh.y; v.z = v.z * psc.z + psh.z; v.w = v.w * psc.w + psh.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            // zero padding comes after the transform (out-of-bounds rows stay 0)
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            put(As, lrow + 32 * j, v);
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            float4 v = rb[j];
            const bool ok = (bvalid >> j) & 1u;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            put(Bs, lrow + 32 * j, v);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wave / WN) * (BM / WM), wn = (wave % WN) * (BN / WN);
    const int fr = lane & 31, fh = lane >> 5;

    if (KT > 0) load_tiles(0);
    // the prologue operands in Ssc were written by other threads: they must be visible before the first store_tiles
    // reads them (without this barrier a late wave 0 -- cold instruction cache on a kernel's first launch -- let the
    // other waves transform the first k-tile with stale LDS contents)
    if (AFFINE) __syncthreads();
    if (KT > 0) store_tiles(0);
    __syncthreads();
    // MFMA phase over the k-tile staged in LDS stage `cur`
    auto mfma_phase = [&](int cur) {
        const float* const As = As0 + cur * A_SZ;
        const float* const Bs = Bs0 + cur * B_SZ;
        if constexpr (PREC == 0) {
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                float4 a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(&As[(wm + i * 32 + fr) * LDS_LD + ks * 8 + fh * 4]);
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[(wn + j * 32 + fr) * LDS_LD + ks * 8 + fh * 4]);
                // k outer, accumulators inner: consecutive MFMAs never depend on each other
#define MFMA_STEP(C)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                        \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].C, b[j].C, acc[i][j], 0, 0, 0);
                MFMA_STEP(x) MFMA_STEP(y) MFMA_STEP(z) MFMA_STEP(w)
#undef MFMA_STEP
            }
        } else {
            // bf16 operand fragment of v_mfma_f32_32x32x16_bf16: lane (r = l&31, h = l>>5) holds row r, k = 8h..8h+7
            const __bf16* Ah = reinterpret_cast<const __bf16*>(As);
            const __bf16* Bh = reinterpret_cast<const __bf16*>(Bs);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 ah[MT], bh[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int o = (wm + i * 32 + fr) * LDS_LDH + ks * 16 + fh * 8;
                    ah[i] = *reinterpret_cast<const bf16x8*>(&Ah[o]);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int o = (wn + j * 32 + fr) * LDS_LDH + ks * 16 + fh * 8;
                    bh[j] = *reinterpret_cast<const bf16x8*>(&Bh[o]);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    // The last k-tile is peeled off so that the loop body loads tile kt+1 UNCONDITIONALLY: with `if (kt + 1 < KT)`
    // around the load the prefetch registers became loop-carried, and the compiler copied them (after an
    // s_waitcnt vmcnt(0)) in front of the MFMA phase -- every wave waited out its own global-load latency before
    // its MFMAs instead of underneath them.
    for (int kt = 0; kt + 1 < KT; ++kt) {
        const int cur = (NBUF == 2) ? (kt & 1) : 0;
        load_tiles(kt + 1);                             // in flight during the MFMA phase
        mfma_phase(cur);
        if (NBUF == 2) {
            // the other stage was last read in iteration kt-1, which every wave left through the barrier below
            store_tiles(cur ^ 1);
            __syncthreads();
        } else {
            __syncthreads();
            store_tiles(0);
            __syncthreads();
        }
    }
    if (KT > 0) {
        mfma_phase((NBUF == 2) ? ((KT - 1) & 1) : 0);
        __syncthreads();
    }

    conv_epilogue<BN, WM, WN, MODE>(p, acc, smem, m0, n0);
}

// =========================================================================================================
// LDS-DMA main loop (exact fp32 only).  Same tiles, same epilogue; the operand tiles travel global -> LDS by
// `buffer_load_dwordx4 ... lds` instead of through VGPRs + ds_write (what that staging cost the matrix pipe:
// profiles/r02_a_wgrad_ablation.txt), two stages, ONE barrier per k-step.  What the staging did to the data moves to
// the operand fetch: BatchNorm affine + ReLU on the A fragment (scale/shift of the fragment's 4 channels come from LDS,
// two broadcast ds_read_b128 per 8-channel step), zero padding through a per-row validity word; rows outside the image
// are fetched with an out-of-range buffer offset (zeros).  LDS rows are 128 B (32 floats, no padding -- the DMA writes
// 1 KiB linearly), so the 16-B chunks of a row are XOR-swizzled by (row>>1)&7 on the SOURCE address and on the read:
// every 16-lane group of a ds_read_b128 then covers 16 distinct 16-B slots of the 256-B bank row (conflict-free).
// =========================================================================================================
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned int* lds_u32p;
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
extern "C" __device__ void uem_raw_buffer_load_lds(i32x4 rsrc, lds_u32p lds, int size, int voffset, int soffset, int offset,
                                                   int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
#define CONV_OOB 0xFFFFFFF0u
__device__ __forceinline__ i32x4 conv_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// KB: channels per k-step (32, or 16: half the stage size, so a third block fits a CU when the block is not persistent)
// PERSIST: the block walks tiles bid, bid + grid, ... and the first k-step of tile t+1 is in flight under the last MFMA phase and the
// epilogue of tile t (round 3: on the short-k layers a block spent most of its life waiting for its first operand tile and writing
// out its last one).  The epilogue then stages in the stage tile t consumed last, so a stage must hold the 64 x (BN + 4) floats of
// one epilogue half (KB = 32: 528 floats of padding per stage; KB = 16: a dedicated epilogue area, two blocks per CU instead of three).
template <int BN, int KB, bool PERSIST>
struct ConvDmaCfg {
    static constexpr int CPR = KB / 4;                                    // 16-B chunks per staged row = DMA lanes per row
    static constexpr int RPT = 256 / CPR;                                 // rows one pass of the 256 threads covers
    static constexpr int AR = BM / RPT, BR = BN / RPT;                    // A / B rows per thread
    static constexpr int A_FLOATS = BM * KB, B_FLOATS = BN * KB, V_FLOATS = BM;
    static constexpr int RAW_STAGE = A_FLOATS + B_FLOATS + V_FLOATS;
    static constexpr int EPI_FLOATS = 64 * (BN + 4);                      // the epilogue's staging area
    static constexpr bool EPI_IN_STAGE = PERSIST && KB == 32;             // epilogue staged in ONE (the just-consumed) stage
    static constexpr bool EPI_SEPARATE = PERSIST && KB != 32;             // epilogue staged behind the two stages
    static constexpr int STAGE_FLOATS = EPI_IN_STAGE && RAW_STAGE < EPI_FLOATS ? EPI_FLOATS : RAW_STAGE;
    static constexpr int BASE_FLOATS = EPI_SEPARATE ? 2 * STAGE_FLOATS + EPI_FLOATS
                                                    : (2 * STAGE_FLOATS > EPI_FLOATS ? 2 * STAGE_FLOATS : EPI_FLOATS);
    // resident blocks per CU (LDS: 65-68 KB / 49 KB / 34 KB)
    static constexpr int BPC = PERSIST ? (BN == 128 ? 2 : (KB == 16 ? 2 : 3)) : (KB == 16 ? 3 : (BN == 128 ? 2 : 3));
};

// MODE 0 forward / 1 data gradient; AFFINE: BatchNorm affine + ReLU on the input operand; PADDED: the filter has taps
// that can fall outside the image (only then does the affine path need the validity words)
template <int BN, int KB, int MODE, bool AFFINE, bool PADDED, int EPI, bool PERSIST>
__global__ __launch_bounds__(256, (ConvDmaCfg<BN, KB, PERSIST>::BPC)) void conv_dma_kernel(const ConvP p, const unsigned x_bytes, const unsigned w_bytes,
                                                                                          const int ntiles) {
    using C = ConvDmaCfg<BN, KB, PERSIST>;
    constexpr int CPR = C::CPR, RPT = C::RPT, AR = C::AR, BR = C::BR;
    constexpr int SWS = KB == 32 ? 1 : 2;                                 // swizzle = (row >> SWS) & (CPR - 1): rows per 256-B bank row
    // 4 x 1 waves: each wave owns 32 of the tile's 128 rows and ALL its columns, so every A element is fetched -- and pushed
    // through the prologue -- by exactly one wave (2 x 2 waves transformed each element twice: the prologue's VALU work cost
    // the forward 12 %, profiles/r02_d_conv_ablation.txt)
    constexpr int WM = 4, WN = 1, MT = BM / WM / 32, NT = BN / WN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ssc = smem + C::BASE_FLOATS;                            // AFFINE only: [Cin] scale, then [Cin] shift
    if (AFFINE) {
        for (int i = threadIdx.x; i < p.Cin; i += 256) { Ssc[i] = p.in_scale[i]; Ssc[p.Cin + i] = p.in_shift[i]; }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = p.Cout / BN;
    const int lrow = tid / CPR;                                          // DMA: CPR lanes per staged row, rows lrow + RPT*j
    const int lc4 = ((tid % CPR) ^ ((lrow >> SWS) & (CPR - 1))) * 4;     // swizzled source chunk (floats) of this lane
    const i32x4 rs_x = conv_rsrc(p.x, x_bytes), rs_w = conv_rsrc(p.w, w_bytes);

    // pointwise layers (1x1, stride 1: output pixel m reads input pixel m) skip the per-row divisions and bounds tests
    // (the AFFINE && !PADDED instantiation is only ever launched pointwise -- conv_dma_go -- so there it is a compile-time fact
    // and the generic gather, with its registers, drops out of the 16-channel forward kernels that run three blocks per CU)
    const bool pointwise = (AFFINE && !PADDED) ||
                           (!PADDED && p.KH * p.KW == 1 && p.pad == 0 && ((MODE == 1) ? (p.sub == 1 && p.stride == 1) : p.stride == 1));
    const int cpb = p.Cin / KB, KT = p.ntaps * cpb, Ktot = p.KH * p.KW * p.Cin;

    // ---- issue side: the tile whose operand tiles are being requested (one k-step ahead of the MFMAs; under PERSIST it moves
    // on to the block's next tile while the current one still computes) -----------------------------------------------------
    int vi = blockIdx.x;                                                 // virtual block id of the issue-side tile
    bool issue_live = vi < ntiles;
    int in0 = 0;
    unsigned iwoff = 0;                                                  // byte offset of the issue tile's filter bank (grouped GEMM)
    int gy[AR], gx[AR], gpix[AR];                                        // gather geometry of this lane's A rows
    auto issue_tile_setup = [&]() {
        const int tile = xcd_remap(vi, ntiles);
        const int im0 = (tile / tiles_n) * BM;
        in0 = (tile % tiles_n) * BN;
        iwoff = p.wg_rows > 0 ? (unsigned)(im0 / p.wg_rows) * p.wg_stride : 0u;
        if (pointwise) {
#pragma unroll
            for (int j = 0; j < AR; ++j) { gy[j] = gx[j] = 0; gpix[j] = (im0 + lrow + RPT * j < p.M) ? im0 + lrow + RPT * j : -1; }
        } else {
            const int HoWo = (MODE == 1) ? p.Hs * p.Ws : p.Ho * p.Wo;
            const int Wrow = (MODE == 1) ? p.Ws : p.Wo;
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const int m = im0 + lrow + RPT * j;
                if (m < p.M) {
                    const int n = m / HoWo, rem = m - n * HoWo;
                    const int oy = rem / Wrow, ox = rem - oy * Wrow;
                    if (MODE == 1) { gy[j] = oy * p.sub + p.py + p.pad; gx[j] = ox * p.sub + p.px + p.pad; }
                    else { gy[j] = oy * p.stride - p.pad; gx[j] = ox * p.stride - p.pad; }
                    gpix[j] = n * p.H * p.W;
                } else { gy[j] = gx[j] = 0; gpix[j] = -1; }
            }
        }
    };
    issue_tile_setup();
    // walk over (tap, channel block); byte offsets from p.x / p.w
    int lt = 0, lci0 = 0;
    unsigned tapok = 0, aoff[AR], boff[BR];
    auto setup_tap = [&](int t) {
        if (pointwise) {
            tapok = 0;
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                aoff[j] = ((unsigned)gpix[j] * (unsigned)p.x_ld + (unsigned)lc4) * 4u;
                tapok |= (gpix[j] >= 0 ? 1u : 0u) << j;
            }
#pragma unroll
            for (int j = 0; j < BR; ++j) boff[j] = ((unsigned)(in0 + lrow + RPT * j) * (unsigned)Ktot + (unsigned)lc4) * 4u + iwoff;
            return;
        }
        const int tap = (MODE == 1) ? (int)((p.tapmask >> (4 * t)) & 0xF) : t;
        const int ky = (p.KW == 1) ? tap : ((p.KW == 3) ? (tap * 11) >> 5 : tap / p.KW);
        const int kx = tap - ky * p.KW;
        tapok = 0;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            bool ok = gpix[j] >= 0;
            int iy, ix;
            if (MODE == 1) {
                const int ty = gy[j] - ky * p.dil, tx = gx[j] - kx * p.dil;
                if (p.stride == 1) { iy = ty; ix = tx; }
                else { iy = ty / p.stride; ix = tx / p.stride; ok = ok && (iy * p.stride == ty) && (ix * p.stride == tx); }
                ok = ok && ty >= 0 && tx >= 0 && iy < p.H && ix < p.W;
            } else {
                iy = gy[j] + ky * p.dil; ix = gx[j] + kx * p.dil;
                ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            }
            aoff[j] = ((unsigned)(gpix[j] + iy * p.W + ix) * (unsigned)p.x_ld + (unsigned)lc4) * 4u;
            tapok |= (ok ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j)
            boff[j] = ((unsigned)(in0 + lrow + RPT * j) * (unsigned)Ktot + (unsigned)(tap * p.Cin + lc4)) * 4u + iwoff;
    };

    f32x16 acc[MT][NT];
    const int wm = (wave / WN) * (BM / WM), wn = (wave % WN) * (BN / WN);
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> SWS) & (CPR - 1);                              // swizzle of every fragment row of this lane
    int rci0 = 0;                                                        // read-side channel base of the k-tile being consumed

    // one function, two __restrict__ stages (+ the prologue operands): alias scopes tell the wait-count pass that the
    // operand reads of tile k do not depend on the DMA of tile k+1 issued in between them (see wgrad.hip).
    // Round 3: the DMA pieces of the next operand tile are no longer requested in one burst in front of the MFMA phase (8 x
    // 60-185 issue cycles during which the wave's matrix pipe had nothing queued) but ONE piece at a time between MFMAs, where
    // a piece's issue hides under the 64 cycles of the MFMA just issued.  Straight-line code: when nothing is left to request
    // (last k-step of the block's last tile) the pieces go out with an out-of-range offset (no memory access, zeros into the
    // stage nobody reads any more).
    constexpr int NKS = KB / 8, NPC = AR + BR;                            // 8-channel MFMA steps per k-step, DMA pieces per k-step
    auto step = [&](float* __restrict__ fill, const float* __restrict__ use, const float* __restrict__ ssc, const bool live,
                    const bool do_phase) {
        float* const fAs = fill;
        float* const fBs = fill + C::A_FLOATS;
        float* const fVs = fBs + C::B_FLOATS;
        if (live && lci0 == 0) setup_tap(lt);                            // block-uniform
        auto piece = [&](const int j) {
            if (j < AR) {
                const bool ok = live && ((tapok >> j) & 1u);
                uem_raw_buffer_load_lds(rs_x, (lds_u32p)(fAs + (j * 4 + wave) * 256), 16, (int)(ok ? aoff[j] : CONV_OOB), 0, 0, 0);
                aoff[j] += KB * 4;
                if (AFFINE && PADDED && (tid % CPR) == 0) fVs[lrow + RPT * j] = ok ? 1.f : 0.f;
            } else if (j < NPC) {
                const int jb = j - AR;
                uem_raw_buffer_load_lds(rs_w, (lds_u32p)(fBs + (jb * 4 + wave) * 256), 16, (int)(live ? boff[jb] : CONV_OOB), 0, 0, 0);
                boff[jb] += KB * 4;
            }
        };
        auto advance = [&]() {
            if (!live) return;
            lci0 += KB;
            if (lci0 >= p.Cin) {
                lci0 = 0;
                if (++lt == p.ntaps) {                                   // this tile's operands are all requested: on to the block's next tile
                    lt = 0;
                    vi += gridDim.x;
                    issue_live = PERSIST && vi < ntiles;
                    if (issue_live) issue_tile_setup();
                }
            }
        };
        if (!do_phase) {
#pragma unroll
            for (int j = 0; j < NPC; ++j) piece(j);
            advance();
            return;
        }
        const float* const As = use;
        const float* const Bs = use + C::A_FLOATS;
        const float* const Vs = Bs + C::B_FLOATS;
        float vld[MT];
        if (AFFINE && PADDED) {
#pragma unroll
            for (int i = 0; i < MT; ++i) vld[i] = Vs[wm + i * 32 + fr];
        }
        // Software pipeline over the four 8-channel steps of the tile: the fragments of step ks+1 are fetched, and pushed
        // through the prologue, underneath the MFMAs of step ks (fetch + transform in front of their own MFMAs left the matrix
        // pipe idle for the read latency and the VALU chain of every step: -8 % on the affine forward).
        float4 fa[2][MT], fb[2][NT], fs[2], fh4[2];
        auto fetch = [&](const int ks, const int buf) {
            const int q = ks * 2 + fh;                                   // 16-B chunk (4 channels) of this lane half
            const int qs = (q ^ sw) * 4;
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[buf][i] = *reinterpret_cast<const float4*>(&As[(wm + i * 32 + fr) * KB + qs]);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[buf][j] = *reinterpret_cast<const float4*>(&Bs[(wn + j * 32 + fr) * KB + qs]);
            if (AFFINE) {
                fs[buf] = *reinterpret_cast<const float4*>(&ssc[rci0 + q * 4]);
                fh4[buf] = *reinterpret_cast<const float4*>(&ssc[p.Cin + rci0 + q * 4]);
            }
        };
        auto xform = [&](const int buf) {
            if (!AFFINE) return;
            const float4 s4 = fs[buf], h4 = fh4[buf];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float4 v = fa[buf][i];
                v.x = fmaxf(__builtin_fmaf(v.x, s4.x, h4.x), 0.f); v.y = fmaxf(__builtin_fmaf(v.y, s4.y, h4.y), 0.f);
                v.z = fmaxf(__builtin_fmaf(v.z, s4.z, h4.z), 0.f); v.w = fmaxf(__builtin_fmaf(v.w, s4.w, h4.w), 0.f);
                if (PADDED) { v.x *= vld[i]; v.y *= vld[i]; v.z *= vld[i]; v.w *= vld[i]; }   // zero padding after the transform
                fa[buf][i] = v;
            }
        };
#define MFMA_STEP(Cc, Bf)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                        \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[Bf][i].Cc, fb[Bf][j].Cc, acc[i][j], 0, 0, 0);
        fetch(0, 0);
        xform(0);
        constexpr int PPS = (NPC + NKS - 1) / NKS;                       // DMA pieces per 8-channel step (the last step takes the rest)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            const bool more = ks + 1 < NKS;
            if (more) fetch(ks + 1, nb);
            MFMA_STEP(x, cb)
            if (ks * PPS < NPC) piece(ks * PPS);
            MFMA_STEP(y, cb)
            if (more) xform(nb);
            MFMA_STEP(z, cb)
            if (PPS > 1 && ks * PPS + 1 < NPC) piece(ks * PPS + 1);
            MFMA_STEP(w, cb)
            // machine order: the reads of the next step, a quarter of this step's MFMAs, one DMA piece, a quarter, the next step's
            // transform, a quarter, one DMA piece, the last quarter
            if (more) __builtin_amdgcn_sched_group_barrier(0x100, MT + NT + (AFFINE ? 2 : 0), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            if (ks * PPS < NPC) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            if (more && AFFINE) __builtin_amdgcn_sched_group_barrier(0x002, MT * (PADDED ? 12 : 8), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
            if (PPS > 1 && ks * PPS + 1 < NPC) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
        }
#undef MFMA_STEP
        static_assert(PPS <= 2 && PPS * NKS >= NPC, "DMA pieces do not fit the MFMA steps");
        advance();
        rci0 += KB;
        if (rci0 >= p.Cin) rci0 = 0;
    };
#define CONV_SYNC()                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                   \
    asm volatile("" ::: "memory")
    float* const st0 = smem;
    float* const epi_area = C::EPI_SEPARATE ? smem + 2 * C::STAGE_FLOATS : smem;
    if (KT > 0) step(st0, st0 + C::STAGE_FLOATS, Ssc, true, false);      // first operand tile of the block's first tile
    int par = 0;                                                         // stage the next k-step consumes
    for (int vc = blockIdx.x; vc < ntiles; vc += gridDim.x) {
        const int tile = xcd_remap(vc, ntiles);
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            CONV_SYNC();
            // every wave has its part of the stage `par` in LDS and is past its reads of the other stage: request the next
            // operand tile (of this tile, or the first one of the block's next tile) into that other stage, under the MFMA phase.
            // ONE call site (the stages are chosen by address, not by a second copy of the loop body, which made the compiler
            // shuffle the 64 accumulator registers between the copies)
            float* const use = st0 + par * C::STAGE_FLOATS;
            float* const fill = st0 + (par ^ 1) * C::STAGE_FLOATS;
            step(fill, use, Ssc, issue_live, true);
            par ^= 1;
        }
        // every wave past its last operand read and its own DMA pieces landed (the stage consumed last becomes the epilogue's
        // staging area; under PERSIST the other one holds the next tile's first operand tile)
        CONV_SYNC();
        conv_epilogue<BN, WM, WN, MODE, EPI>(p, acc, C::EPI_IN_STAGE ? st0 + (par ^ 1) * C::STAGE_FLOATS : epi_area, m0, n0);
        if (!PERSIST) break;
    }
#undef CONV_SYNC
}


// =========================================================================================================
// bf16-STORAGE convolution (BASELINE config 5: "bf16 weights, CDNA4 bf16 MFMA"): activations and weights are bf16 in
// HBM, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, bf16 out.  Same LDS-DMA skeleton as conv_dma_kernel: a staged
// row is 64 channels = 128 bytes, its eight 16-byte chunks (8 bf16 = exactly one MFMA operand fragment) XOR-swizzled by
// (row>>1)&7, two stages, one barrier per 64-channel k-step.  No operand prologue: with matrix instructions 16x faster
// than the f32 ones a BatchNorm transform at the fetch would be the bottleneck, so this path materialises relu(bn(z))
// (uem_bn_apply_bf16) -- the same HBM bytes per activation as fp32 + prologue.  Epilogue: fp32 accumulators through LDS,
// rounded to bf16 (RNE), 16-byte stores; optional accumulate (y += ...) and per-tile BatchNorm statistics of the ROUNDED
// values (what the next layer will read).
// =========================================================================================================
#define KBH 64                                                             // bf16 channels per k-step
// PERSIST (round 4): the block walks tiles bid, bid + grid, ... and the first k-step of tile t+1 is requested under the last MFMA phase
// of tile t -- on the short-k layers (K = 64 ... 256: one to four k-steps) a block used to spend most of its life waiting for its first
// operand tile.  The epilogue then stages in the stage tile t consumed last (the other one is being filled), so a stage holds at least
// the 64 x (BN + 4) floats of an epilogue half.
// BMT (round 5): rows of the block's tile, 128 (four waves, two blocks per CU) or 256 (eight waves as 4 x 2, ONE block per CU).  The
// pointwise layers of a ResNet run a handful of 64-channel k-steps per tile and every k-step exposes one L2 / HBM round trip (the
// next operand tile is requested under ~0.2 us of MFMAs and waited for at the next barrier: ~1.3 us per k-step, measured as tile time
// = k-steps x 1.3 us + epilogue): a 256-row tile does twice (BN = 128) or four times (BN = 256) the MFMA work per round trip and
// moves half the operand bytes per flop from L2 into LDS; its wave tile 64 x 128 (BN = 256) reads 0.75 KB of LDS per MFMA where
// 64 x 64 reads 1 KB.
// NST (round 6): stages of the operand ring.  2 = rounds 2-5 (the next k-step requested under the current one's MFMAs and waited for at
// the next barrier).  3 (256-row persistent blocks only): TWO k-steps in flight -- the per-shape table of the step said every k-step
// exposed one round trip (tile time = k-steps x 1.3 us + epilogue: 0.2-0.4 us of MFMAs per k-step and CU against 1.3 us of latency),
// and with one 256-row block per CU nobody else covers it.  The barrier in front of k-step g waits with a COUNTED vmcnt for the
// pieces of step g only (the pieces of step g + 1, requested during step g - 1, stay in flight: loads, stores and LDS-DMA retire in
// issue order), the ring keeps running across the block's tiles -- the next tile's first two k-steps fly under this tile's last MFMA
// phases and its whole epilogue -- and the epilogue's barriers are LDS-only (a __syncthreads() there drains the DMA queue).
template <int BN, bool PERSIST = false, int BMT = 128, int NST = 2>
struct ConvBf16Cfg {
    static constexpr int A_ELEMS = BMT * KBH, B_ELEMS = BN * KBH;         // bf16 elements per stage
    static constexpr int RAW_STAGE_BYTES = (A_ELEMS + B_ELEMS) * 2;
    static constexpr int EPI_BYTES = 64 * (BN + 4) * 4;
    static constexpr int STAGE_BYTES = PERSIST && RAW_STAGE_BYTES < EPI_BYTES ? EPI_BYTES : RAW_STAGE_BYTES;
    static constexpr int LDS_BYTES = NST * STAGE_BYTES > EPI_BYTES ? NST * STAGE_BYTES : EPI_BYTES;
    static_assert(NST == 2 || (PERSIST && BMT == 256), "the deeper ring is built for the persistent 256-row blocks");
    static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit the CU's LDS");
};
__device__ __forceinline__ unsigned short f2bf(float f) {                  // RNE, NaN stays NaN (hipcc: v_cvt_pk_bf16_f32)
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// EPI: the epilogue the launch needs, fixed at compile time -- on the small-K layers a block's time is instruction issue, and an
// epilogue that tests its options per row keeps the compiler from batching the LDS reads, conversions and stores:
//   0  full dense tiles, plain store;  1  full dense tiles + BatchNorm tile statistics (forward) / the residual tail and
//   BatchNorm-backward options (data gradient);  -1  anything (ragged last tile, strided output rows, accumulate in forward)
// SPLIT (round 6): the vector-memory counter of a wave retires loads, stores and LDS-DMA pieces in issue order, so a wave that stores its
// share of a tile and then waits for operand pieces waits for the stores' ACKNOWLEDGEMENTS first -- microseconds under load, once per tile,
// with the matrix pipe idle (ablation, profiles/r06_b_conv_bf16_ring_ablation.txt: 256 -> 1024 on 64 x 64 maps, 144 us with the stores, 100
// without; streaming the stores between the next tile's MFMAs made it WORSE, 196 us: every k-step then waited for store
// acknowledgements, profiles/r06_c_*).  The waves therefore split the roles: the first half of the block (LOADERS) issues every operand piece
// and never stores -- its counted wait sees pieces only; the second half (STORERS) writes the tile out and never waits on the
// vector-memory counter at all -- its stores drain behind the next tile's MFMAs.  All waves run the MFMAs.
template <int BN, int MODE, int EPI, bool PERSIST, int BMT = 128, int NST = 2, bool SPLIT = false>
__global__ __launch_bounds__(2 * BMT, BMT == 128 ? 2 : 1) void conv_bf16_kernel(const ConvP p, const unsigned x_bytes, const unsigned w_bytes, const int ntiles) {
    using C = ConvBf16Cfg<BN, PERSIST, BMT, NST>;
    constexpr bool RING = NST > 2;
    constexpr int NTH = 2 * BMT, NW = NTH / 64;                           // threads, waves
    constexpr int NWL = SPLIT ? NW / 2 : NW, RPS = NWL * 8, NA = BMT / RPS;   // loader waves, rows per DMA pass (8 lanes per 128-B row), A pieces per loader thread
    constexpr int WM = BMT / 64, WN = 2, MT = BMT / WM / 32, NT = BN / WN / 32, BR = BN / RPS;
    static_assert(MT == 2 && (NA == 4 || NA == 8), "a wave owns 64 rows; four (eight: split roles) A pieces per loader thread and k-step");
    static_assert(!SPLIT || (PERSIST && EPI >= 0 && !(MODE == 1 && EPI == 1)), "split roles: persistent blocks on full tiles without epilogue loads");
    constexpr bool RAWB = RING || SPLIT;                                  // epilogue with LDS-only barriers and hand-written staging reads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* const lds16 = reinterpret_cast<unsigned short*>(smem);
    const unsigned short* const xh = reinterpret_cast<const unsigned short*>(p.x);
    unsigned short* const yh = reinterpret_cast<unsigned short*>(p.y);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = !SPLIT || wave < NWL, is_storer = !SPLIT || wave >= NWL;      // wave-uniform
    const int tiles_n = p.Cout / BN;
    int m0 = 0, n0 = 0;                                                  // compute side: the tile whose accumulators the block holds
    const int lrow = tid >> 3;                                           // DMA: 8 lanes per 128-B row, rows lrow + RPS*j
    const int lc8 = ((tid & 7) ^ ((lrow >> 1) & 7)) * 8;                 // swizzled source chunk (elements) of this lane
    const i32x4 rs_x = conv_rsrc(p.x, x_bytes), rs_w = conv_rsrc(p.w, w_bytes);
    (void)xh;

    // pointwise layers (1x1, stride 1, no padding: two thirds of a ResNet's launches): output pixel m reads input pixel m, no
    // per-row divisions and no bounds tests besides m < M -- the index arithmetic below is most of a small-K block's instructions
    const bool pointwise = p.KH * p.KW == 1 && p.pad == 0 && ((MODE == 1) ? (p.sub == 1 && p.stride == 1) : p.stride == 1);
    // issue side: the tile whose operand tiles are being requested (one k-step ahead of the MFMAs; under PERSIST it moves on to the
    // block's next tile while the current one still computes)
    int vi = blockIdx.x, in0 = 0;
    bool issue_live = vi < ntiles;
    int gy[NA], gx[NA], gpix[NA];
    auto issue_tile_setup = [&]() {
        const int tile = xcd_remap(vi, ntiles);
        const int im0 = (tile / tiles_n) * BMT;
        in0 = (tile % tiles_n) * BN;
        if (pointwise) {
#pragma unroll
            for (int j = 0; j < NA; ++j) { gy[j] = gx[j] = 0; gpix[j] = (im0 + lrow + RPS * j < p.M) ? im0 + lrow + RPS * j : -1; }
        } else {
            const int HoWo = (MODE == 1) ? p.Hs * p.Ws : p.Ho * p.Wo;
            const int Wrow = (MODE == 1) ? p.Ws : p.Wo;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int m = im0 + lrow + RPS * j;
                if (m < p.M) {
                    const int n = m / HoWo, rem = m - n * HoWo;
                    const int oy = rem / Wrow, ox = rem - oy * Wrow;
                    if (MODE == 1) { gy[j] = oy * p.sub + p.py + p.pad; gx[j] = ox * p.sub + p.px + p.pad; }
                    else { gy[j] = oy * p.stride - p.pad; gx[j] = ox * p.stride - p.pad; }
                    gpix[j] = n * p.H * p.W;
                } else { gy[j] = gx[j] = 0; gpix[j] = -1; }
            }
        }
    };
    issue_tile_setup();
    const int cpb = p.Cin / KBH, KT = p.ntaps * cpb, Ktot = p.KH * p.KW * p.Cin;
    int lt = 0, lci0 = 0;
    unsigned tapok = 0, aoff[NA], boff[BR];
    auto setup_tap = [&](int t) {
        if (pointwise) {
            tapok = 0;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                aoff[j] = ((unsigned)gpix[j] * (unsigned)p.x_ld + (unsigned)lc8) * 2u;
                tapok |= (gpix[j] >= 0 ? 1u : 0u) << j;
            }
#pragma unroll
            for (int j = 0; j < BR; ++j) boff[j] = ((unsigned)(in0 + lrow + RPS * j) * (unsigned)Ktot + (unsigned)lc8) * 2u;
            return;
        }
        const int tap = (MODE == 1) ? (int)((p.tapmask >> (4 * t)) & 0xF) : t;
        const int ky = (p.KW == 1) ? tap : ((p.KW == 3) ? (tap * 11) >> 5 : tap / p.KW);
        const int kx = tap - ky * p.KW;
        tapok = 0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            bool ok = gpix[j] >= 0;
            int iy, ix;
            if (MODE == 1) {
                const int ty = gy[j] - ky * p.dil, tx = gx[j] - kx * p.dil;
                if (p.stride == 1) { iy = ty; ix = tx; }
                else { iy = ty / p.stride; ix = tx / p.stride; ok = ok && (iy * p.stride == ty) && (ix * p.stride == tx); }
                ok = ok && ty >= 0 && tx >= 0 && iy < p.H && ix < p.W;
            } else {
                iy = gy[j] + ky * p.dil; ix = gx[j] + kx * p.dil;
                ok = ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            }
            aoff[j] = ((unsigned)(gpix[j] + iy * p.W + ix) * (unsigned)p.x_ld + (unsigned)lc8) * 2u;
            tapok |= (ok ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j)
            boff[j] = ((unsigned)(in0 + lrow + RPS * j) * (unsigned)Ktot + (unsigned)(tap * p.Cin + lc8)) * 2u;
    };

    f32x16 acc[MT][NT];
    const int wm = (wave / WN) * (BMT / WM), wn = (wave % WN) * (BN / WN);
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;

    // Round 3 (as conv_dma_kernel): the DMA pieces of the next operand tile go out one at a time between the MFMAs of the current
    // one -- 4 + BR pieces per 16 MFMAs here, where a burst of them in front of the MFMA phase cost more issue time than the MFMAs
    // themselves take -- and the step has ONE call site (stages chosen by address).  A step with nothing left to request sends its
    // pieces with an out-of-range offset (no memory access).
    constexpr int NPC = NA + BR, NKS = KBH / 16;
    auto step = [&](unsigned short* __restrict__ fill, const unsigned short* __restrict__ use, const bool live_, const bool do_phase) {
        const bool live = live_ && !(UEM_DBG(p.dbg) & 4);
        unsigned short* const fAs = fill;
        unsigned short* const fBs = fill + C::A_ELEMS;
        if (live && lci0 == 0) setup_tap(lt);
        auto piece = [&](const int j) {
            if (SPLIT && !is_loader) return;                             // split roles: the storer waves issue no operand pieces
            if (j < NA) {
                const bool ok = live && ((tapok >> j) & 1u);
                uem_raw_buffer_load_lds(rs_x, (lds_u32p)(fAs + (j * NWL + wave) * 512), 16, (int)(ok ? aoff[j] : CONV_OOB), 0, 0, 0);
                aoff[j] += KBH * 2;
            } else if (j < NPC) {
                const int jb = j - NA;
                uem_raw_buffer_load_lds(rs_w, (lds_u32p)(fBs + (jb * NWL + wave) * 512), 16, (int)(live ? boff[jb] : CONV_OOB), 0, 0, 0);
                boff[jb] += KBH * 2;
            }
        };
        auto advance = [&]() {
            if (!live) return;
            lci0 += KBH;
            if (lci0 >= p.Cin) {
                lci0 = 0;
                if (++lt == p.ntaps) {                                   // this tile's operands are all requested: on to the block's next tile
                    lt = 0;
                    vi += gridDim.x;
                    issue_live = PERSIST && vi < ntiles;
                    if (issue_live) issue_tile_setup();
                }
            }
        };
        if (!do_phase) {
#pragma unroll
            for (int j = 0; j < NPC; ++j) piece(j);
            advance();
            return;
        }
        const unsigned short* const As = use;
        const unsigned short* const Bs = use + C::A_ELEMS;
        constexpr int PPS = (NPC + NKS - 1) / NKS;                       // DMA pieces per 16-channel MFMA step
        static_assert(PPS <= MT * NT, "more DMA pieces than MFMAs in a step");
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int qs = (((ks * 2 + fh) ^ sw)) * 8;                   // this lane half's 16-B chunk = its 8 k of the MFMA step
            bf16x8 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm + i * 32 + fr) * KBH + qs]);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn + j * 32 + fr) * KBH + qs]);
            int q = 0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (q < PPS && ks * PPS + q < NPC) piece(ks * PPS + q);
                    ++q;
                }
        }
        advance();
    };
    constexpr bool FULL = EPI >= 0;
    const bool dense_rows = FULL || !(MODE == 1 && p.sub > 1);
    auto row_off = [&](const int m) -> size_t {
        if (dense_rows) return (size_t)m * p.y_ld;
        const int hw = p.Hs * p.Ws;
        const int ni = m / hw, rem = m - ni * hw;
        const int yy = rem / p.Ws, xx = rem - yy * p.Ws;
        return (((size_t)ni * p.Ho + (size_t)(yy * p.sub + p.py)) * p.Wo + (size_t)(xx * p.sub + p.px)) * p.y_ld;
    };
    constexpr int NTS = SPLIT ? NTH / 2 : NTH;                            // threads that write the tile out (split roles: the storer waves)
    constexpr int LDW = BN + 4, TPR = BN / 8, RPP = NTS / TPR, NRP = 64 / RPP, NCH = BMT / 64;     // the epilogue walks NCH chunks of 64 rows
    // LAZY (round 5, persistent blocks on full tiles without epilogue loads): the barrier in front of the epilogue does not wait for the
    // next tile's operand pieces and the first barrier of the next tile does not wait for the epilogue's stores (EPI_STORES = the
    // stores every thread issues per tile: a lower bound is all the counted wait needs)
    constexpr bool LAZY = PERSIST && EPI >= 0 && !(MODE == 1 && EPI == 1);
    constexpr int EPI_STORES = NCH * NRP;
    static_assert(!RING || EPI >= 0, "the ring serves full dense tiles");
    // LDS-only barrier (ring): a __syncthreads() with LDS-DMA pieces outstanding makes the compiler drain them (vmcnt(0))
#define EPI_SYNC()                                                              \
    do {                                                                        \
        if constexpr (RAWB) {                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  \
            __builtin_amdgcn_s_barrier();                                       \
            asm volatile("" ::: "memory");                                      \
        } else {                                                                \
            __syncthreads();                                                    \
        }                                                                       \
    } while (0)
    static_assert(EPI_STORES >= 1 && EPI_STORES < 48, "counted wait out of the counter's range");
    float* stg = smem;                                                   // PERSIST: the stage the tile consumed last
    const int etid = SPLIT ? (tid - NTH / 2) & (NTS - 1) : tid;          // storer-relative thread index (loader waves never use theirs)
    const int srow = etid / TPR, sc8 = (etid % TPR) * 8;
    // data gradient only: the residual tail (identity gradient acc_src*[acc_bits]) and the first pass of a BatchNorm(+ReLU)
    // backward over the rounded dx (same contract as the fp32 kernel's epilogue, bf16 tensors)
    const bool fuse_bn = MODE == 1 && EPI != 0 && p.tile_bnbwd != nullptr;
    const bool acc_on = (EPI == 0 || (EPI == 1 && MODE == 0)) ? false : p.accumulate != 0;
    const bool stats_on = MODE == 0 && (EPI == 1 || (EPI < 0 && p.tile_stats != nullptr));
    const bool ablate_store = (EPI < 0 || RING) && (UEM_DBG(p.dbg) & 1), ablate_stage = (EPI < 0 || RING) && (UEM_DBG(p.dbg) & 2);
    const bool ablate_epi = RING && (UEM_DBG(p.dbg) & 8);                 // DEBUG_HOOKS builds only: the tile's epilogue skipped altogether
    const unsigned short* const zh = reinterpret_cast<const unsigned short*>(p.bn_z);
    const unsigned short* const ah = p.acc_src != nullptr ? reinterpret_cast<const unsigned short*>(p.acc_src) : yh;
    float pb[8], pg[8], bsc[8], bsh[8], bmu[8], bis[8];
    // what the epilogue reads besides the accumulators -- the tensor accumulated into, its gate bits, the BatchNorm input and
    // its mask bits -- is fetched one 64-row chunk ahead: chunk 0 before the main loop, chunk c + 1 while chunk c is written out
    // (two register sets, indexed by the chunk's parity)
    uint4 eo[2][NRP], ez[2][NRP];
    unsigned eab[2][NRP], ebb[2][NRP];
    const bool has_ab = MODE == 1 && p.acc_bits != nullptr, has_zb = MODE == 1 && p.bn_bits != nullptr;
    auto epi_fetch = [&](const int hm) {
        const int hb = hm & 1;
#pragma unroll
        for (int u = 0; u < NRP; ++u) {
            const int m = m0 + hm * 64 + srow + u * RPP;
            eab[hb][u] = 0xffffffffu; ebb[hb][u] = 0u;                   // RAW words: shifted where they are used (see conv_epilogue)
            if (!FULL && m >= p.M) continue;
            if (acc_on) eo[hb][u] = *reinterpret_cast<const uint4*>(ah + row_off(m) + n0 + sc8);
            if (MODE == 1) {
                const size_t e0 = (size_t)m * p.Cout + n0 + sc8;           // dense rows whenever bits / bn_z are given
                if (has_ab) eab[hb][u] = p.acc_bits[e0 >> 5];
                if (fuse_bn) {
                    ez[hb][u] = *reinterpret_cast<const uint4*>(zh + e0);
                    if (has_zb) ebb[hb][u] = p.bn_bits[e0 >> 5];
                }
            }
        }
    };
#define CONV_SYNC()                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                   \
    asm volatile("" ::: "memory")
    unsigned short* const st0 = lds16;
    constexpr int SE = C::STAGE_BYTES / 2;                               // bf16 elements per stage
    if (RING && (UEM_DBG(p.dbg) >> 8) && (blockIdx.x & 1)) {             // DEBUG_HOOKS: odd blocks start (dbg >> 8) x 3.4 us late (chip-wide phase lock?)
        for (int i = 0; i < (UEM_DBG(p.dbg) >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    }
    if (KT > 0) step(st0, st0 + SE, issue_live, false);                  // first operand tile of the block's first tile
    if (RING && KT > 0) step(st0 + SE, st0, issue_live, false);          // ... and the second k-step of the issue stream (maybe the next tile's)
    int par = 0;                                                         // stage the next k-step consumes
    for (int vc = blockIdx.x; vc < ntiles; vc += gridDim.x) {
        {
            const int tile = xcd_remap(vc, ntiles);
            m0 = (tile / tiles_n) * BMT; n0 = (tile % tiles_n) * BN;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pb[e] = pg[e] = 0.f; bsc[e] = bsh[e] = bmu[e] = bis[e] = 0.f; }
        if (fuse_bn) {
            const float* v4 = p.bn_vec + n0 + sc8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { bsc[e] = v4[e]; bsh[e] = v4[p.Cout + e]; bmu[e] = v4[2 * p.Cout + e]; bis[e] = v4[3 * p.Cout + e]; }
        }
        if (!RING && (EPI < 0 || (MODE == 1 && EPI == 1))) epi_fetch(0);
        if (KT > 0) {
            for (int kt = 0; kt < KT; ++kt) {
                if constexpr (RING) {
                    // the epilogue's own operands (fused data gradients) are requested in front of the tile's LAST k-step: younger than
                    // every operand piece the tile still waits for, so no k-step barrier waits for an HBM stream (rounds 4-5: requested
                    // before the main loop they made persistence a loss for these epilogues)
                    // step g needs the pieces requested during step g - 2; younger than those are the NPC pieces of step g - 1 and, on
                    // the first k-step of a later tile, the epilogue's stores in between (a lower bound is all a counted wait needs)
                    // ... which holds for the SECOND k-step of a later tile as well (its pieces were requested during the previous tile's
                    // last k-step, before that tile's epilogue): only from the third k-step on does a wait for operand pieces imply a
                    // wait for the previous tile's stores to be acknowledged (one counter, retired in order)
                    if constexpr (SPLIT) {
                        // loader waves: nothing but operand pieces in their queue, NPC per k-step; storer waves: no wait on the
                        // vector-memory counter at all (their stores drain behind the MFMAs)
                        if (is_loader) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPC) : "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    } else if (kt <= 1 && vc != (int)blockIdx.x && !(UEM_DBG(p.dbg) & 16)) {
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPC + EPI_STORES) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPC) : "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if ((MODE == 1 && EPI == 1) && kt == KT - 1) epi_fetch(0);
                    const int fs = par == 0 ? NST - 1 : par - 1;
                    step(st0 + fs * SE, st0 + par * SE, issue_live, true);
                    par = par + 1 == NST ? 0 : par + 1;
                    continue;
                }
                if constexpr (SPLIT) {                                   // two stages, split roles: the loaders' queue holds operand pieces only
                    if (is_loader) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                } else if (LAZY && p.lazy && kt == 0 && vc != (int)blockIdx.x) {
                    // first k-step of a later tile of the block: its operand pieces were requested under the PREVIOUS tile's last
                    // MFMA phase and every thread has issued at least EPI_STORES stores since (the epilogue's rows) -- the memory
                    // counter retires in order, so "at most EPI_STORES outstanding" means the pieces have landed while the stores
                    // are still on their way (vmcnt(0) here waited for the write acknowledgements of a whole tile, once per tile)
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(EPI_STORES) : "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                } else {
                    CONV_SYNC();
                }
                step(st0 + (par ^ 1) * SE, st0 + par * SE, issue_live, true);
                par ^= 1;
            }
            // every wave past its last operand read, every DMA piece landed: the last step's out-of-range pieces (not persistent: the
            // staging area spans both stages) or the next tile's first operand tile (persistent: in the other stage)
            if (RAWB || (LAZY && p.lazy)) {
                // the pieces in flight fill the OTHER stage(s): nothing the epilogue touches -- it only needs every wave past its LDS reads
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                CONV_SYNC();
            }
        } else {
            __syncthreads();
        }
        if (PERSIST) stg = reinterpret_cast<float*>(st0 + (RING ? (par == 0 ? NST - 1 : par - 1) : (par ^ 1)) * SE);   // the stage consumed last

        // ---- epilogue (operands prefetched above) ---------------------------------------------------------------
        if (ablate_epi) {
            if (acc[0][0][0] == 123456.f) yh[0] = 1;                         // keep the accumulators alive
            continue;
        }
#pragma unroll
        for (int hm = 0; hm < NCH; ++hm) {
            if (wm / 64 == hm && !ablate_stage) {
                const int rbase = wm % 64;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stg[(rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDW + wn + j * 32 + fr] = acc[i][j][r];
            }
            EPI_SYNC();
            if (hm + 1 < NCH && (EPI < 0 || (MODE == 1 && EPI == 1))) epi_fetch(hm + 1);
            if (is_storer) {
            bool rok[NRP];
#pragma unroll
            for (int u = 0; u < NRP; ++u) rok[u] = FULL || m0 + hm * 64 + srow + u * RPP < p.M;
            const uint4 (&o)[NRP] = eo[hm & 1];
            const uint4 (&zq)[NRP] = ez[hm & 1];
            unsigned abyte[NRP], bbyte[NRP];
#pragma unroll
            for (int u = 0; u < NRP; ++u) {
                const unsigned bshift = (unsigned)(((size_t)(m0 + hm * 64 + srow + u * RPP) * p.Cout + n0 + sc8) & 31);
                abyte[u] = has_ab ? (eab[hm & 1][u] >> bshift) & 0xffu : 0xffu;
                bbyte[u] = has_zb ? (ebb[hm & 1][u] >> bshift) & 0xffu : 0u;
            }
            // ring: the staged accumulators are read by hand-written ds_read_b128 -- the compiler orders every LDS read it can see behind
            // every outstanding LDS-DMA piece AND, with them, behind the previous chunk's global stores (an `s_waitcnt vmcnt(0)` in front
            // of each chunk's first read, found in the ISA: the next tile's operand prefetch and the write acknowledgements of the rows
            // just stored were waited for four times per tile); the reads are ordered by the barrier above and waited for right here
            f32x4v svlo[NRP], svhi[NRP];
            if constexpr (RAWB) {
#pragma unroll
                for (int u = 0; u < NRP; ++u) {
                    const unsigned la = (unsigned)(unsigned long long)(lds_u32p)(&stg[(srow + u * RPP) * LDW + sc8]);
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(svlo[u]), "=&v"(svhi[u]) : "v"(la));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < NRP; ++u) {
                const int row = srow + u * RPP;
                if (!rok[u]) continue;
                float v[8];
                if constexpr (RAWB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = svlo[u][e]; v[4 + e] = svhi[u][e]; }
                } else {
                    *reinterpret_cast<float4*>(&v[0]) = *reinterpret_cast<const float4*>(&stg[row * LDW + sc8]);
                    *reinterpret_cast<float4*>(&v[4]) = *reinterpret_cast<const float4*>(&stg[row * LDW + sc8 + 4]);
                }
                if (acc_on) {
                    const unsigned w4[4] = {o[u].x, o[u].y, o[u].z, o[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(w4[e] << 16), hi = __uint_as_float(w4[e] & 0xffff0000u);
                        v[2 * e] += ((abyte[u] >> (2 * e)) & 1u) ? lo : 0.f;
                        v[2 * e + 1] += ((abyte[u] >> (2 * e + 1)) & 1u) ? hi : 0.f;
                    }
                }
                unsigned pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned lo = f2bf(v[2 * e]), hi = f2bf(v[2 * e + 1]);
                    pk[e] = lo | (hi << 16);
                    v[2 * e] = bf2f((unsigned short)lo); v[2 * e + 1] = bf2f((unsigned short)hi);
                }
                if (!ablate_store) *reinterpret_cast<uint4*>(yh + row_off(m0 + hm * 64 + row) + n0 + sc8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                if (stats_on) {                                              // BatchNorm statistics of what was stored (rounded)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pb[e] += v[e]; pg[e] = fmaf(v[e], v[e], pg[e]); }
                }
                if (fuse_bn) {
                    const unsigned z4[4] = {zq[u].x, zq[u].y, zq[u].z, zq[u].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float z = (e & 1) ? __uint_as_float(z4[e >> 1] & 0xffff0000u) : __uint_as_float(z4[e >> 1] << 16);
                        const bool on = p.bn_bits != nullptr ? ((bbyte[u] >> e) & 1u) != 0u : (z * bsc[e] + bsh[e] > 0.f);
                        const float dp = on ? v[e] : 0.f;
                        pb[e] += dp;
                        pg[e] = fmaf(dp, (z - bmu[e]) * bis[e], pg[e]);
                    }
                }
            }
            }
            EPI_SYNC();
            // column sums over 128 rows = two chunks: the partial-sum tensors keep one entry per 128 rows whatever the block's tile
            // (the finalize kernels merge 128-row groups)
            float* const tile_out = fuse_bn ? p.tile_bnbwd : (stats_on ? p.tile_stats : nullptr);
            if ((hm & 1) && tile_out != nullptr) {
                // every thread summed its 8 columns over its rows in registers; the RPP thread rows combine through LDS
                float* const red = stg;                                      // [2][RPP][BN]
                if (is_storer) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { red[srow * BN + sc8 + e] = pb[e]; red[(RPP + srow) * BN + sc8 + e] = pg[e]; pb[e] = pg[e] = 0.f; }
                }
                EPI_SYNC();
                for (int i = is_storer ? etid : 2 * BN; i < 2 * BN; i += NTS) {
                    const int which = i / BN, col = i % BN;
                    float a = 0.f;
#pragma unroll 8
                    for (int r = 0; r < RPP; ++r) a += red[(which * RPP + r) * BN + col];
                    const size_t tiles_m = (size_t)(p.M / 128);
                    tile_out[((size_t)which * p.Cout + n0 + col) * tiles_m + (size_t)(m0 / 128 + (hm >> 1))] = a;
                }
                if (hm + 1 < NCH) EPI_SYNC();                           // red is the next chunk's staging area
            }
        }
        if (!PERSIST) break;
    }                                                                    // tiles of a persistent block
#undef CONV_SYNC
#undef EPI_SYNC
}

static int conv_check(const uem_conv_shape* s) {
    if (!s) return uem_fail(UEM_ERR_INVALID, "conv: null shape");
    if (s->N <= 0 || s->H <= 0 || s->W <= 0 || s->Cin <= 0 || s->Cout <= 0 || s->KH <= 0 || s->KW <= 0 || s->stride <= 0 || s->dil <= 0 || s->pad < 0)
        return uem_fail(UEM_ERR_INVALID, "conv: bad shape");
    const int ho = (s->H + 2 * s->pad - s->dil * (s->KH - 1) - 1) / s->stride + 1;
    const int wo = (s->W + 2 * s->pad - s->dil * (s->KW - 1) - 1) / s->stride + 1;
    if (ho != s->Ho || wo != s->Wo) return uem_fail(UEM_ERR_INVALID, "conv: Ho/Wo (%d,%d) inconsistent with input (expected %d,%d)", s->Ho, s->Wo, ho, wo);
    if (s->x_ld < s->Cin || s->y_ld < s->Cout || (s->x_ld % 4) != 0) return uem_fail(UEM_ERR_INVALID, "conv: bad x_ld/y_ld");
    // the loaders index with 32-bit element offsets
    const double lim = 4294967296.0;
    if ((double)s->N * s->H * s->W * s->x_ld >= lim || (double)s->N * s->Ho * s->Wo * s->y_ld >= lim ||
        (double)s->Cout * s->KH * s->KW * s->Cin >= lim)
        return uem_fail(UEM_ERR_UNSUPPORTED, "conv: tensor has >= 2^32 elements (split the batch)");
    return UEM_OK;
}

template <int BN_, int WM_, int WN_, int MODE, int NBUF, int PREC>
static void conv_go(const ConvP& p, bool affine, int grid, hipStream_t st) {
    const size_t a_sz = PREC == 0 ? (size_t)BM * LDS_LD : (size_t)BM * LDS_LDH;
    const size_t b_sz = PREC == 0 ? (size_t)BN_ * LDS_LD : (size_t)BN_ * LDS_LDH;
    static const int lds_pad = getenv("UEM_CONV_LDS_PAD") ? atoi(getenv("UEM_CONV_LDS_PAD")) : 0;   // occupancy experiments
    size_t lds = (size_t)NBUF * (a_sz + b_sz) * sizeof(float) + (size_t)lds_pad;
    if (affine) lds += (size_t)2 * p.Cin * sizeof(float);
    // 769..1024 blocks of the 128x128 tile are 4 per CU: at 3 resident blocks they run as 3 + 1 (the last one alone,
    // nothing hiding its load/store phases), at 2 resident blocks as 2 + 2.  Unused LDS caps the residency at 2
    // (layer2 shapes +4..15 %, layer4 shapes +-2 %).
    static const int occ2 = getenv("UEM_CONV_OCC2") ? atoi(getenv("UEM_CONV_OCC2")) : 1;
    if (occ2 && BN_ == 128 && PREC == 0 && NBUF == 1 && grid > 768 && grid <= 1024 && lds < 56 * 1024) lds = 56 * 1024;
    if (affine) {
        auto k = conv_fwd_kernel<BN_, WM_, WN_, MODE, true, NBUF, PREC>;
        if (uem_allow_lds((const void*)k, lds)) k<<<grid, 256, lds, st>>>(p);
    } else {
        auto k = conv_fwd_kernel<BN_, WM_, WN_, MODE, false, NBUF, PREC>;
        if (uem_allow_lds((const void*)k, lds)) k<<<grid, 256, lds, st>>>(p);
    }
}

// tuning overrides (scripts/sweep_conv.py): LDS-DMA main loop on/off (-1 = rule), its N tile (0 = rule)
static int g_conv_dma = -1, g_conv_dma_bn = 0, g_conv_dma_kb = 0;
extern "C" void uemdbg_conv_config(int dma, int bn) { g_conv_dma = dma; g_conv_dma_bn = bn % 1000; g_conv_dma_kb = bn / 1000; }

static int g_conv_persist = -1;  // tuning override: 1 / 0 = persistent blocks on / off, -1 = rule
extern "C" void uemdbg_conv_persist(int v) { g_conv_persist = v; }

template <int BN_, int KB_, int MODE, bool PERSIST>
static void conv_dma_go(const ConvP& p, bool affine, unsigned xb, unsigned wb, hipStream_t st) {
    using C = ConvDmaCfg<BN_, KB_, PERSIST>;
    const int ntiles = (int)uem_cdiv(p.M, BM) * (p.Cout / BN_);
    const size_t lds = ((size_t)C::BASE_FLOATS + (affine ? 2 * (size_t)p.Cin : 0)) * sizeof(float);
    // persistent blocks: one per resident-block slot of the chip (a multiple of 8, so that a block's tiles stay on its XCD's
    // share of the tile order); a launch that does not fill the slots keeps one tile per block
    const int slots = 256 * C::BPC;
    const int grid = PERSIST && ntiles > slots ? slots : ntiles;
    // "padded" variant = the general gather with validity words; besides filters with taps outside the image it takes the
    // strided 1x1 layers, so that the unpadded affine variant can assume output pixel m reads input pixel m
    const bool padded = p.KH * p.KW > 1 || p.stride != 1 || p.pad != 0;
    auto go = [&](auto k) {
        if (uem_allow_lds((const void*)k, lds)) k<<<grid, 256, lds, st>>>(p, xb, wb, ntiles);
    };
    const bool full = p.M % BM == 0 && (MODE == 0 || p.sub == 1) && p.bias == nullptr && !(MODE == 0 && p.accumulate);
    const bool extras = MODE == 0 ? p.tile_stats != nullptr : (p.accumulate || p.acc_src || p.tile_bnbwd);
    auto pick = [&](auto epi) {
        constexpr int E = decltype(epi)::value;
        if (MODE == 0 && affine && padded) go(conv_dma_kernel<BN_, KB_, MODE, MODE == 0, MODE == 0, E, PERSIST>);
        else if (MODE == 0 && affine) go(conv_dma_kernel<BN_, KB_, MODE, MODE == 0, false, E, PERSIST>);
        else go(conv_dma_kernel<BN_, KB_, MODE, false, false, E, PERSIST>);
    };
    if (!full) pick(std::integral_constant<int, -1>{});
    else if (extras) pick(std::integral_constant<int, 1>{});
    else pick(std::integral_constant<int, 0>{});
}

// The LDS-DMA main loop takes exact-fp32 forward / data-gradient launches with 32-multiple input and 64-multiple output
// channel counts whose tensors fit 32-bit buffer offsets; 1 = launched.
template <int MODE>
static int conv_dma_try(const ConvP& p, bool affine, hipStream_t st) {
    if constexpr (MODE == 2) return 0;
    static const int env = getenv("UEM_CONV_DMA") ? atoi(getenv("UEM_CONV_DMA")) : 1;
    if (!(g_conv_dma >= 0 ? g_conv_dma : env)) return 0;
    if (p.Cin % BK != 0 || p.Cout % 64 != 0 || p.x_ld % 4 != 0 || p.ntaps <= 0) return 0;
    if (affine && (!p.relu || p.Cin > 1024)) return 0;
    if (((uintptr_t)p.x | (uintptr_t)p.w) & 15) return 0;
    const double groups = p.wg_rows > 0 ? (double)uem_cdiv(p.M, p.wg_rows) : 1.0;
    const double xb = (double)p.N * p.H * p.W * p.x_ld * 4.0, wb = groups * (double)p.Cout * p.KH * p.KW * p.Cin * 4.0;
    if (xb >= 4294967280.0 || wb >= 4294967280.0) return 0;
    // the residual tails (conv1 of a bottleneck backwards: K = C/4 channels in, C out, with the identity gradient and the previous
    // bn3's reduction streaming through the epilogue) take 64-wide tiles: three resident blocks per CU instead of two cover each
    // other's epilogues (layer1 0.51 -> 0.41 ms, layer2 0.33 -> 0.27, layer3 0.22 -> 0.19, layer4 0.65 -> 0.63:
    // scripts/bench_dgrad_tail.py), and with K this small the second read of dy costs little
    static const int wt_env = getenv("UEM_WIDE_TAIL") ? atoi(getenv("UEM_WIDE_TAIL")) : 1;     // 0 off, 1 rule, 2 rule up to 256 channels in
    const bool wide_tail = wt_env != 0 && MODE == 1 && (p.accumulate != 0 || p.tile_bnbwd != nullptr) && 2 * p.ntaps * p.Cin <= p.Cout &&
                           (wt_env != 2 || p.Cin <= 256);
    bool bn128 = p.Cout % 128 == 0 && g_conv_dma_bn != 64 && !(wide_tail && g_conv_dma_bn == 0);
    if constexpr (MODE != 2) {
        static const int penv = getenv("UEM_CONV_PERSIST") ? atoi(getenv("UEM_CONV_PERSIST")) : -1;
        const int pset = g_conv_persist >= 0 ? g_conv_persist : penv;
        const int ntiles128 = (int)uem_cdiv(p.M, BM) * (p.Cout / 128), ntiles64 = (int)uem_cdiv(p.M, BM) * (p.Cout / 64);
        // Winograd GEMMs (grouped pointwise launches, row counts 16 / 36 x the tile count): blocks of a launch run in rounds of the
        // chip's resident-block slots, so the tile width is the one whose grid fills whole rounds -- 64-wide tiles cost ~8 % per
        // flop (the A tile is read twice as often) and win when the 128-wide grid leaves a quarter of its last round empty
        // (F(4x4,3x3) on layer3 at B = 32: 1152 tiles of 128 on 768 / 512 slots, 93 / 103 TFLOP/s; 2304 tiles of 64 on 768)
        if (p.wg_rows > 0 && bn128 && g_conv_dma_bn == 0 && g_conv_dma_kb == 0 && pset < 0) {
            auto fill = [](int tiles, int slots) { return (double)tiles / ((double)uem_cdiv(tiles, slots) * slots); };
            const int slots128 = MODE == 0 ? 768 : 512;                  // forward: 16-channel k-steps, 3 blocks per CU; else 2
            if (0.92 * fill(ntiles64, 768) > fill(ntiles128, slots128) + 0.02) bn128 = false;
        }
        // persistent blocks (profiles/r03_a_conv_persist_sweep.txt): the data gradient wherever a block gets at least two tiles;
        // the forward on the 64-wide tiles and the 64-channel pointwise layers (elsewhere its 16-channel k-steps with three
        // resident blocks per CU do as well or better)
        const bool many = bn128 ? ntiles128 > 512 : ntiles64 > 768;
        const bool persist = pset >= 0 ? pset != 0 : (many && (MODE == 1 || !bn128 || (p.ntaps == 1 && p.Cin <= 64)));
        // 16-channel k-steps (34 KB of LDS: a third resident block) pay on the forward 1x1 layers, whose short k-loops leave
        // the most prologue / epilogue time to cover (+2-4 %); with padding words or the data gradient's epilogues they lose
        // (profiles/r02_h_conv_sweep_k16.txt)
        const bool k16 = g_conv_dma_kb == 16 || (g_conv_dma_kb == 0 && MODE == 0 && p.ntaps == 1 && !persist);
        const unsigned xbu = (unsigned)xb, wbu = (unsigned)wb;
        if (persist) {
            if (k16 && bn128 && MODE == 0) conv_dma_go<128, 16, MODE == 0 ? 0 : 0, true>(p, affine, xbu, wbu, st);
            else if (bn128) conv_dma_go<128, 32, MODE, true>(p, affine, xbu, wbu, st);
            else conv_dma_go<64, 32, MODE, true>(p, affine, xbu, wbu, st);
        } else {
            if (k16 && bn128) conv_dma_go<128, 16, MODE, false>(p, affine, xbu, wbu, st);
            else if (bn128) conv_dma_go<128, 32, MODE, false>(p, affine, xbu, wbu, st);
            else conv_dma_go<64, 32, MODE, false>(p, affine, xbu, wbu, st);
        }
    }
    return 1;
}

template <int MODE>
static int conv_launch(const ConvP& p, bool affine, hipStream_t st, int prec = 0) {
    if (prec == 0 && conv_dma_try<MODE>(p, affine, st)) return uem_check_launch("conv2d (dma)");
    static const int trace_legacy = getenv("UEM_CONV_TRACE_LEGACY") ? atoi(getenv("UEM_CONV_TRACE_LEGACY")) : 0;
    if (trace_legacy)                                   // diagnostic: which launches stay on the register-staged kernels, and why
        fprintf(stderr, "[uemda] register-staged conv: mode %d prec %d M=%d Cin=%d Cout=%d k=%dx%d stride=%d pad=%d dil=%d x_ld=%d y_ld=%d affine=%d x%%16=%d w%%16=%d\n",
                MODE, prec, p.M, p.Cin, p.Cout, p.KH, p.KW, p.stride, p.pad, p.dil, p.x_ld, p.y_ld, (int)affine,
                (int)((uintptr_t)p.x & 15), (int)((uintptr_t)p.w & 15));
    static const int nbuf = getenv("UEM_CONV_NBUF") ? atoi(getenv("UEM_CONV_NBUF")) : 1;
    static const int smallk_bn64 = getenv("UEM_CONV_SMALLK_BN64") ? atoi(getenv("UEM_CONV_SMALLK_BN64")) : 0;
    const int tiles_m = (int)uem_cdiv(p.M, BM);
    const bool prefer64 = smallk_bn64 > 0 && p.ntaps * p.Cin <= smallk_bn64 && p.Cout % 64 == 0;
    if (p.Cout % 128 == 0 && !prefer64) {
        const int grid = tiles_m * (p.Cout / 128);
        if (prec == 2) conv_go<128, 2, 2, MODE, 1, 2>(p, affine, grid, st);
        else if (nbuf == 2) conv_go<128, 2, 2, MODE, 2, 0>(p, affine, grid, st);
        else conv_go<128, 2, 2, MODE, 1, 0>(p, affine, grid, st);
    } else if (p.Cout % 64 == 0) {
        const int grid = tiles_m * (p.Cout / 64);
        if (prec == 2) conv_go<64, 2, 2, MODE, 1, 2>(p, affine, grid, st);
        else if (nbuf == 2) conv_go<64, 2, 2, MODE, 2, 0>(p, affine, grid, st);
        else conv_go<64, 2, 2, MODE, 1, 0>(p, affine, grid, st);
    } else {
        const int grid = tiles_m * (int)uem_cdiv(p.Cout, 32);
        if (prec == 2) conv_go<32, 4, 1, MODE, 1, 2>(p, affine, grid, st);
        else conv_go<32, 4, 1, MODE, 1, 0>(p, affine, grid, st);
    }
    return uem_check_launch("conv2d");
}

struct BnBwdFuse { const float* z; const float* vec; float* tiles; const float* acc_src; const uint32_t* acc_bits; const uint32_t* bn_bits; };
static int conv2d_fwd_impl(const float* x, const float* w, const float* bias, const float* in_scale,
                           const float* in_shift, float* y, const uem_conv_shape* s, int flags, float* tile_stats,
                           const BnBwdFuse* bnbwd, void* stream);

extern "C" int uem_conv2d_fwd(const float* x, const float* w, const float* bias, const float* in_scale,
                              const float* in_shift, float* y, const uem_conv_shape* s, int flags, void* stream) {
    return conv2d_fwd_impl(x, w, bias, in_scale, in_shift, y, s, flags, nullptr, nullptr, stream);
}
extern "C" int uem_conv2d_dgrad_bnbwd(const float* dy, const float* w_t, float* dx, const uem_conv_shape* s, const float* bn_z,
                                      const float* bn_vec, float* tile_partials, int flags, void* stream) {
    UEM_REQUIRE(bn_z && bn_vec && tile_partials && s, "conv2d_dgrad_bnbwd: null pointer");
    if (s->stride != 1 || ((int64_t)s->N * s->H * s->W) % 128 != 0 || s->Cin % 64 != 0 || s->x_ld != s->Cin)
        return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_dgrad_bnbwd: needs stride 1, M %% 128 == 0, Cin %% 64 == 0");
    BnBwdFuse f{bn_z, bn_vec, tile_partials, nullptr, nullptr, nullptr};
    UEM_REQUIRE((flags & ~UEM_CONV_PREC_BF16) == 0, "conv2d_dgrad_bnbwd: only precision flags are accepted");
    return conv2d_fwd_impl(dy, w_t, nullptr, nullptr, nullptr, dx, s, UEM_CONV_TRANSPOSED | flags, nullptr, &f, stream);
}
extern "C" int uem_conv2d_dgrad_tail(const float* dy, const float* w_t, float* dx, const uem_conv_shape* s, const float* acc_src,
                                     const uint32_t* acc_bits, const float* bn_z, const float* bn_vec, const uint32_t* bn_bits,
                                     float* tile_partials, int flags, void* stream) {
    UEM_REQUIRE(s && dx, "conv2d_dgrad_tail: null pointer");
    UEM_REQUIRE((acc_bits == nullptr) == (acc_src == nullptr), "conv2d_dgrad_tail: acc_src and acc_bits go together");
    UEM_REQUIRE((bn_z == nullptr) == (tile_partials == nullptr) && (bn_z == nullptr) == (bn_vec == nullptr) && (bn_z || !bn_bits),
                "conv2d_dgrad_tail: bn_z, bn_vec and tile_partials go together");
    if (s->stride != 1 || ((int64_t)s->N * s->H * s->W) % 128 != 0 || s->Cin % 64 != 0 || s->x_ld != s->Cin || s->Cin % 32 != 0)
        return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_dgrad_tail: needs stride 1, M %% 128 == 0, Cin %% 64 == 0, dense rows");
    UEM_REQUIRE((flags & ~(UEM_CONV_PREC_BF16 | UEM_CONV_ACCUMULATE)) == 0, "conv2d_dgrad_tail: bad flags");
    BnBwdFuse f{bn_z, bn_vec, tile_partials, acc_src, acc_bits, bn_bits};
    return conv2d_fwd_impl(dy, w_t, nullptr, nullptr, nullptr, dx, s, UEM_CONV_TRANSPOSED | flags | (acc_src ? UEM_CONV_ACCUMULATE : 0),
                           nullptr, &f, stream);
}
extern "C" int uem_conv2d_fwd_stats(const float* x, const float* w, const float* in_scale, const float* in_shift,
                                    float* y, const uem_conv_shape* s, int flags, float* tile_stats, void* stream) {
    UEM_REQUIRE(tile_stats && s, "conv2d_fwd_stats: null pointer");
    if ((flags & (UEM_CONV_TRANSPOSED | UEM_CONV_ACCUMULATE)) || ((int64_t)s->N * s->Ho * s->Wo) % 128 != 0 || s->Cout % 64 != 0)
        return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_fwd_stats: needs a forward conv with M %% 128 == 0 and Cout %% 64 == 0");
    return conv2d_fwd_impl(x, w, nullptr, in_scale, in_shift, y, s, flags, tile_stats, nullptr, stream);
}

static int conv2d_fwd_impl(const float* x, const float* w, const float* bias, const float* in_scale,
                           const float* in_shift, float* y, const uem_conv_shape* s, int flags, float* tile_stats,
                           const BnBwdFuse* bnbwd, void* stream) {
    UEM_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
    int rc = conv_check(s);
    if (rc) return rc;
    UEM_REQUIRE(!(flags & 16), "conv2d: flag 16 (the split-bf16 operand mode of rounds 1-3) is retired");
    const bool affine = (flags & UEM_CONV_IN_AFFINE) != 0;
    const bool transposed = (flags & UEM_CONV_TRANSPOSED) != 0;
    const int prec = (flags & UEM_CONV_PREC_BF16) ? 2 : 0;
    UEM_REQUIRE(!affine || (in_scale && in_shift), "conv2d_fwd: affine prologue needs scale/shift");
    UEM_REQUIRE(!(affine && transposed), "conv2d_fwd: prologue not supported on the transposed gather");
    ConvP p;
    p.x = x; p.w = w; p.bias = bias; p.in_scale = in_scale; p.in_shift = in_shift; p.y = y;
    p.KH = s->KH; p.KW = s->KW; p.stride = s->stride; p.pad = s->pad; p.dil = s->dil;
    p.accumulate = (flags & UEM_CONV_ACCUMULATE) ? 1 : 0;
    p.relu = (flags & UEM_CONV_IN_RELU) ? 1 : 0;
    p.sub = 1; p.py = p.px = 0; p.Hs = p.Ws = 0; p.ntaps = s->KH * s->KW; p.tapmask = 0; p.dbg = 0; p.y_bf16 = 0; p.wg_rows = 0; p.wg_stride = 0;
    p.tile_stats = tile_stats;
    p.bn_z = bnbwd ? bnbwd->z : nullptr; p.bn_vec = bnbwd ? bnbwd->vec : nullptr; p.tile_bnbwd = bnbwd ? bnbwd->tiles : nullptr;
    p.acc_src = bnbwd ? bnbwd->acc_src : nullptr; p.acc_bits = bnbwd ? bnbwd->acc_bits : nullptr; p.bn_bits = bnbwd ? bnbwd->bn_bits : nullptr;
    if (!transposed) {
        UEM_REQUIRE(s->Cin % BK == 0, "conv2d_fwd: Cin=%d must be a multiple of 32", s->Cin);
        p.N = s->N; p.H = s->H; p.W = s->W; p.Cin = s->Cin; p.Ho = s->Ho; p.Wo = s->Wo; p.Cout = s->Cout;
        p.x_ld = s->x_ld; p.y_ld = s->y_ld;
        p.M = s->N * s->Ho * s->Wo;
        return conv_launch<0>(p, affine, (hipStream_t)stream, prec);
    }
    // data gradient: rows = input pixels (N,H,W), reduction over (tap, Cout), gather from dY (N,Ho,Wo,Cout).
    // dX[y,x] += dY[(y+pad-ky*d)/s, (x+pad-kx*d)/s] * W[ky,kx] only where the division is exact, so the
    // pixels split into s*s parity classes, each reached by its own subset of taps: one launch per class
    // walks exactly those taps (a 3x3 stride-2 conv: 1 + 2 + 2 + 4 taps instead of 4 x 9).
    UEM_REQUIRE(s->Cout % BK == 0, "conv2d dgrad: Cout=%d must be a multiple of 32", s->Cout);
    UEM_REQUIRE(s->KH * s->KW <= 16, "conv2d dgrad: at most 16 taps (4-bit tap ids in a 64-bit list)");
    p.N = s->N; p.H = s->Ho; p.W = s->Wo; p.Cin = s->Cout;       // what the gather reads
    p.Ho = s->H; p.Wo = s->W; p.Cout = s->Cin;                   // what the kernel writes
    p.x_ld = s->y_ld; p.y_ld = s->x_ld;
    p.sub = s->stride;
    for (int py = 0; py < s->stride; ++py) {
        for (int px = 0; px < s->stride; ++px) {
            p.py = py; p.px = px;
            p.Hs = (s->H - py + s->stride - 1) / s->stride;
            p.Ws = (s->W - px + s->stride - 1) / s->stride;
            if (p.Hs <= 0 || p.Ws <= 0) continue;
            p.ntaps = 0; p.tapmask = 0;
            for (int ky = 0; ky < s->KH; ++ky) {
                if ((py + s->pad - ky * s->dil) % s->stride != 0) continue;      // C % keeps the sign: 0 is still exact
                for (int kx = 0; kx < s->KW; ++kx) {
                    if ((px + s->pad - kx * s->dil) % s->stride != 0) continue;
                    p.tapmask |= (unsigned long long)(ky * s->KW + kx) << (4 * p.ntaps);
                    ++p.ntaps;
                }
            }
            if (p.ntaps == 0 && p.accumulate) continue;       // class reached by no tap: adds zero (1x1 stride 2: 3 of 4)
            p.M = s->N * p.Hs * p.Ws;
            rc = conv_launch<1>(p, false, (hipStream_t)stream, prec);
            if (rc) return rc;
        }
    }
    return UEM_OK;
}

// The npos (16: F(2x2,3x3), 36: F(4x4,3x3)) element-wise products of Winograd (winograd.hip) as ONE pointwise launch:
// V [npos][T][K] x U[npos][N][K]^T -> M [npos][T][N]; GEMM row r = (position, tile) takes the filter bank of position r / T.
extern "C" int uem_wino_gemm(const float* V, const float* U, float* Mt, int T, int K, int N, int npos, int data_gradient, void* stream) {
    UEM_REQUIRE(V && U && Mt && T > 0 && K > 0 && N > 0 && (npos == 16 || npos == 36), "wino_gemm: bad arguments");
    if (T % BM != 0 || K % BK != 0 || N % 64 != 0 || (((uintptr_t)V | (uintptr_t)U) & 15))
        return uem_fail(UEM_ERR_UNSUPPORTED, "wino_gemm: needs T %% 128 == 0, K %% 32 == 0, N %% 64 == 0, 16-byte aligned operands");
    if ((double)npos * T * K * 4.0 >= 4294967280.0 || (double)npos * T * N * 4.0 >= 4294967280.0 || (double)npos * N * K * 4.0 >= 4294967280.0)
        return uem_fail(UEM_ERR_UNSUPPORTED, "wino_gemm: tensor beyond 32-bit byte offsets (split the batch)");
    ConvP p;
    p.x = V; p.w = U; p.bias = nullptr; p.in_scale = p.in_shift = nullptr; p.y = Mt;
    p.N = 1; p.H = 1; p.W = npos * T; p.Cin = K; p.Ho = 1; p.Wo = npos * T; p.Cout = N;
    p.KH = p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1; p.x_ld = K; p.y_ld = N;
    p.accumulate = 0; p.relu = 0; p.tile_stats = nullptr; p.bn_z = nullptr; p.bn_vec = nullptr; p.tile_bnbwd = nullptr;
    p.acc_src = nullptr; p.acc_bits = nullptr; p.bn_bits = nullptr;
    p.sub = 1; p.py = p.px = 0; p.Hs = p.Ws = 0; p.ntaps = 1; p.tapmask = 0; p.dbg = 0; p.y_bf16 = 0;
    p.wg_rows = T; p.wg_stride = (unsigned)((size_t)N * K * 4);
    p.M = npos * T;
    // data_gradient: the same pointwise product through the data-gradient instantiation (MODE 1: its tile rules, and a kernel name
    // that profiles attribute to the data-gradient family)
    const int took = data_gradient ? conv_dma_try<1>(p, false, (hipStream_t)stream) : conv_dma_try<0>(p, false, (hipStream_t)stream);
    if (!took) return uem_fail(UEM_ERR_UNSUPPORTED, "wino_gemm: shape not taken by the LDS-DMA kernel");
    return uem_check_launch("wino_gemm");
}

static int stem_fwd_impl(const float* x4, const float* w8, float* y, int N, int H, int W, float* tile_stats, int flags, void* stream,
                         int y_bf16);
extern "C" int uem_conv2d_stem_fwd(const float* x4, const float* w8, float* y, int N, int H, int W, void* stream) {
    return stem_fwd_impl(x4, w8, y, N, H, W, nullptr, 0, stream, 0);
}
extern "C" int uem_conv2d_stem_fwd_stats(const float* x4, const float* w8, float* y, int N, int H, int W, float* tile_stats, int flags,
                                         void* stream) {
    UEM_REQUIRE(tile_stats, "conv2d_stem_fwd_stats: null pointer");
    UEM_REQUIRE((flags & ~UEM_CONV_PREC_BF16) == 0, "conv2d_stem_fwd_stats: only precision flags are accepted");
    const int64_t M = (int64_t)N * ((H + 6 - 7) / 2 + 1) * ((W + 6 - 7) / 2 + 1);
    if (M % 128 != 0) return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_stem_fwd_stats: needs N*Ho*Wo %% 128 == 0");
    return stem_fwd_impl(x4, w8, y, N, H, W, tile_stats, flags, stream, 0);
}
static int stem_fwd_impl(const float* x4, const float* w8, float* y, int N, int H, int W, float* tile_stats, int flags, void* stream,
                         int y_bf16 = 0);
// bf16 storage: z leaves as a bf16 tensor (rounded at the store, statistics of the rounded values); bf16 operands
extern "C" int uem_conv2d_stem_fwd_stats_bf16(const float* x4, const float* w8, uint16_t* y, int N, int H, int W, float* tile_stats,
                                              void* stream) {
    UEM_REQUIRE(tile_stats, "conv2d_stem_fwd_stats_bf16: null pointer");
    const int64_t M = (int64_t)N * ((H + 6 - 7) / 2 + 1) * ((W + 6 - 7) / 2 + 1);
    if (M % 128 != 0) return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_stem_fwd_stats_bf16: needs N*Ho*Wo %% 128 == 0");
    return stem_fwd_impl(x4, w8, reinterpret_cast<float*>(y), N, H, W, tile_stats, UEM_CONV_PREC_BF16, stream, 1);
}
static int stem_fwd_impl(const float* x4, const float* w8, float* y, int N, int H, int W, float* tile_stats, int flags, void* stream,
                         int y_bf16) {
    UEM_REQUIRE(x4 && w8 && y && N > 0 && H >= 7 && W >= 7, "conv2d_stem_fwd: bad arguments");
    ConvP p;
    p.x = x4; p.w = w8; p.bias = nullptr; p.in_scale = p.in_shift = nullptr; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = 32;               // one tap row = 8 px x 4 ch
    p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1; p.Cout = 64;
    p.KH = 7; p.KW = 1; p.stride = 2; p.pad = 3; p.dil = 1; p.x_ld = 4; p.y_ld = 64;
    p.accumulate = 0; p.relu = 0; p.tile_stats = tile_stats; p.bn_z = nullptr; p.bn_vec = nullptr; p.tile_bnbwd = nullptr; p.acc_src = nullptr; p.acc_bits = nullptr; p.bn_bits = nullptr;
    p.sub = 1; p.py = p.px = 0; p.Hs = p.Ws = 0; p.ntaps = 7; p.tapmask = 0; p.dbg = 0; p.y_bf16 = y_bf16; p.wg_rows = 0; p.wg_stride = 0;
    p.M = N * p.Ho * p.Wo;
    return conv_launch<2>(p, false, (hipStream_t)stream, (flags & UEM_CONV_PREC_BF16) ? 2 : 0);
}

// =========================================================================================================
// weight gradient: dW[o][tap][i] += sum_m dY[m][o] * A[m][tap, i]
//   GEMM rows = output channels (TM), cols = input channels of ONE tap (TN), reduction over pixels.
//   LDS tiles are k-major ([pixel][channel]) exactly as they sit in NHWC memory; the MFMA operands
//   A[i][k], B[k][j] are then plain ds_read_b32 with consecutive lanes on consecutive channels.
//   grid = (tiles, splits); each block reduces a slice of the pixels and adds its tile with fp32 atomics.
// =========================================================================================================
struct WgradP {
    const float* x;
    const float* dy;
    const float* in_scale;
    const float* in_shift;
    float* dw;
    int M, N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil, x_ld, dy_ld, relu;
    int rows_per_split;
    int dy_bf16;           // stem only (MODE 2): dy is a bf16 tensor (bf16 storage: the stem's dz), widened at the load
    int dbg;               // UEM_WGRAD_DBG (diagnostic builds of the schedule only; results are wrong when non-zero):
                           // 1 = no global loads in the loop, 2 = no LDS stores in the loop, 4 = no atomics
};

// PREC as in conv_fwd_kernel.  The reduction runs over pixels, so both MFMA operands are needed k-major while
// global memory (and the LDS image filled from it by coalesced channel-contiguous loads) is channel-major: the
// bf16 modes keep a [pixel][channel] bf16 image and read it with ds_read_b64_tr_b16, the hardware transposing
// read (per 16 lanes: a 4-pixel x 16-channel block, delivered channel-per-lane).  Row stride T*2 + 64 bytes puts
// the 4 pixel rows of a 32-lane read into the 4 bank quarters (conflict-free); the 8-byte stores of one pixel
// row are contiguous.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 lds_tr_frag(const unsigned char* img, int ld, int k0, int ch0, int lane) {
    // operand fragment of v_mfma_f32_32x32x16_bf16 for the 32 channels at ch0 and the 8 pixels k0 .. k0+7
    const int q = (lane & 15) >> 2, pp = lane & 3, g16 = (lane >> 4) & 1;
    const unsigned char* a = img + (k0 + q) * ld + 2 * (ch0 + 16 * g16 + 4 * pp);
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * ld));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int TM, int TN, int WM, int WN, int WK, int MODE, bool AFFINE, int PREC>
__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(const WgradP p) {
    constexpr int MT = TM / WM / 32, NT = TN / WN / 32;
    constexpr int DTPR = TM / 4, DRPP = 256 / DTPR, DPASS = (BK + DRPP - 1) / DRPP;   // dY loader
    constexpr int XTPR = TN / 4, XRPP = 256 / XTPR, XPASS = (BK + XRPP - 1) / XRPP;   // X loader
    constexpr int NIMG = 1;                                                            // one bf16 image per operand
    constexpr int LDD = TM * 2 + (TM >= 64 ? 64 : 0), LDX = TN * 2 + (TN >= 64 ? 64 : 0);   // bf16 image row bytes
    constexpr int D_BYTES = PREC == 0 ? BK * TM * 4 : NIMG * BK * LDD;
    constexpr int X_BYTES = PREC == 0 ? BK * TN * 4 : NIMG * BK * LDX;
    static_assert(PREC == 0 || (BK / WK) % 16 == 0, "bf16 modes need 16 pixel rows per wave and step");
    __shared__ __attribute__((aligned(16))) unsigned char Dsm[D_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Xsm[X_BYTES];
    float* Ds = reinterpret_cast<float*>(Dsm);
    float* Xs = reinterpret_cast<float*>(Xsm);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci_tiles = (p.Cin + TN - 1) / TN;
    const int taps = p.KH * p.KW;
    // 1-D grid of tiles x splits, remapped so that one XCD walks consecutive ids = all tiles of one pixel
    // chunk: the chunk's dY / X rows are then pulled into ONE L2 instead of eight
    const int ntiles = ci_tiles * taps * ((p.Cout + TM - 1) / TM);
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / ntiles;
    int t = lin - split * ntiles;
    const int ci_t = t % ci_tiles; t /= ci_tiles;
    const int tap = t % taps;
    const int co_t = t / taps;
    const int co0 = co_t * TM, ci0 = ci_t * TN;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int HoWo = p.Ho * p.Wo;

    const int drow = tid / DTPR, dc4 = (tid % DTPR) * 4;
    const int xrow = tid / XTPR, xc4 = (tid % XTPR) * 4;
    float4 sc, sh;
    if (AFFINE) {
        sc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + xc4);
        sh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + xc4);
    }
    float4 rd[DPASS], rx[XPASS];
    unsigned xok = 0;
    // (image, oy, ox) of this thread's X rows, advanced incrementally by BK rows per step (no divisions in the loop)
    int xn[XPASS], xoy[XPASS], xox[XPASS];
#pragma unroll
    for (int j = 0; j < XPASS; ++j) {
        const int m = mbeg + xrow + j * XRPP;
        xn[j] = m / HoWo;
        const int rem = m - xn[j] * HoWo;
        xoy[j] = rem / p.Wo;
        xox[j] = rem - xoy[j] * p.Wo;
    }
    unsigned dok = 0;
    auto load_tiles = [&](int mb) {
        // branch-free: rows / columns outside the problem load from offset 0 and are zeroed at the LDS store (dok / xok)
        xok = 0;
        dok = 0;
#pragma unroll
        for (int j = 0; j < DPASS; ++j) {
            const int r = drow + j * DRPP;
            const int m = mb + r;
            const bool ok = r < BK && m < mend && (co0 + dc4) < p.Cout;
            const size_t doff = ok ? (size_t)m * p.dy_ld + co0 + dc4 : (size_t)0;
            if (MODE == 2 && p.dy_bf16) {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.dy) + doff);
                rd[j] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                    __uint_as_float(u.y & 0xffff0000u));
            } else {
                rd[j] = *reinterpret_cast<const float4*>(p.dy + doff);
            }
            dok |= (ok ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < XPASS; ++j) {
            const int r = xrow + j * XRPP;
            const int m = mb + r;
            size_t off = 0;
            if (r < BK && m < mend) {
                const int n = xn[j], oy = xoy[j], ox = xox[j];
                int iy, ix;
                if (MODE == 2) { iy = oy * 2 - 3 + ky; ix = ox * 2 - 3 + (xc4 >> 2); }
                else { iy = oy * p.stride - p.pad + ky * p.dil; ix = ox * p.stride - p.pad + kx * p.dil; }
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    const size_t pix = ((size_t)n * p.H + iy) * p.W + ix;
                    off = (MODE == 2) ? pix * 4 : pix * p.x_ld + ci0 + xc4;
                    xok |= 1u << j;
                }
            }
            rx[j] = *reinterpret_cast<const float4*>(p.x + off);
            // advance this row slot by BK pixels for the next step
            xox[j] += BK;
            while (xox[j] >= p.Wo) { xox[j] -= p.Wo; ++xoy[j]; }
            while (xoy[j] >= p.Ho) { xoy[j] -= p.Ho; ++xn[j]; }
        }
    };
    auto put_bf16 = [&](unsigned char* img, int ld, int r, int c4, float4 v) {
        bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(img + r * ld + 2 * c4) = h;
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < DPASS; ++j) {
            const int r = drow + j * DRPP;
            float4 v = rd[j];
            const bool ok = (dok >> j) & 1u;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            if (r < BK) {
                if constexpr (PREC == 0) *reinterpret_cast<float4*>(&Ds[r * TM + dc4]) = v;
                else put_bf16(Dsm, LDD, r, dc4, v);
            }
        }
#pragma unroll
        for (int j = 0; j < XPASS; ++j) {
            const int r = xrow + j * XRPP;
            float4 v = rx[j];
            const bool ok = (xok >> j) & 1u;
            if (AFFINE) {                               // prologue applied after the MFMA phase (see conv_fwd_kernel)
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            if (r < BK) {
                if constexpr (PREC == 0) *reinterpret_cast<float4*>(&Xs[r * TN + xc4]) = v;
                else put_bf16(Xsm, LDX, r, xc4, v);
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wk = wave / (WM * WN), wmn = wave % (WM * WN);
    const int wm = (wmn / WN) * (TM / WM), wn = (wmn % WN) * (TN / WN);
    const int fr = lane & 31, fh = lane >> 5;
    constexpr int KPW = BK / WK;                       // pixel rows of each step owned by this wave

    if (mbeg < mend) {
        load_tiles(mbeg);
        store_tiles();
        __syncthreads();
        // MFMA phase over the pixel block staged in LDS; the last block is peeled off so that the loop body prefetches
        // UNCONDITIONALLY (a conditional prefetch makes the registers loop-carried and the compiler then waits for the
        // loads and copies them in front of the MFMAs -- see conv_fwd_kernel)
        auto mfma_phase = [&]() {
            if constexpr (PREC == 0) {
                // operand fragments of pixel pair kp+1 are fetched before the MFMAs of pair kp are issued
                float a[2][MT], b[2][NT];
                {
                    const int k = wk * KPW + fh;
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[0][i] = Ds[k * TM + wm + i * 32 + fr];
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[0][j] = Xs[k * TN + wn + j * 32 + fr];
                }
#pragma unroll
                for (int kp = 0; kp < KPW / 2; ++kp) {
                    if (kp + 1 < KPW / 2) {
                        const int k = wk * KPW + (kp + 1) * 2 + fh;
#pragma unroll
                        for (int i = 0; i < MT; ++i) a[(kp + 1) & 1][i] = Ds[k * TM + wm + i * 32 + fr];
#pragma unroll
                        for (int j = 0; j < NT; ++j) b[(kp + 1) & 1][j] = Xs[k * TN + wn + j * 32 + fr];
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp & 1][i], b[kp & 1][j], acc[i][j], 0, 0, 0);
                    // keep that order in the machine schedule: the LDS reads of the next pair, then this pair's MFMAs
                    // (left alone, the scheduler sinks each read group behind the MFMAs and waits for it at once)
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KPW / 16; ++ks) {
                    const int k0 = wk * KPW + ks * 16 + 8 * fh;
                    bf16x8 ah[MT], bh[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) ah[i] = lds_tr_frag(Dsm, LDD, k0, wm + i * 32, lane);
#pragma unroll
                    for (int j = 0; j < NT; ++j) bh[j] = lds_tr_frag(Xsm, LDX, k0, wn + j * 32, lane);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
        };
        int mb = mbeg;
        for (; mb + BK < mend; mb += BK) {
            if (!(UEM_DBG(p.dbg) & 1)) load_tiles(mb + BK);
            mfma_phase();
            __syncthreads();
            if (!(UEM_DBG(p.dbg) & 2)) store_tiles();
            __syncthreads();
        }
        mfma_phase();
    }
    if (UEM_DBG(p.dbg) & 4) {
        if (acc[0][0][0] == 1.2345f) p.dw[0] = 1.f;     // keep the accumulators alive
        return;
    }
    const size_t row_ld = (size_t)taps * p.Cin;         // dW[o][tap][i]
    float* wbase = p.dw + (size_t)tap * p.Cin;
    if (co0 + TM <= p.Cout && ci0 + TN <= p.Cin) {      // full tile: unconditional atomics, constant strides
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float* base = wbase + (size_t)(co0 + wm + i * 32 + 4 * fh) * row_ld + (ci0 + wn + j * 32 + fr);
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicAdd(base + (size_t)((r & 3) + 8 * (r >> 2)) * row_ld, acc[i][j][r]);
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ci = ci0 + wn + j * 32 + fr;
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (co < p.Cout) atomicAdd(wbase + (size_t)co * row_ld + ci, acc[i][j][r]);
            }
        }
}

template <int TM, int TN, int WM, int WN, int WK, int MODE, int PREC = 0>
static void wgrad_go(const WgradP& p0, bool affine, hipStream_t st) {
    WgradP p = p0;
#ifdef UEM_DEBUG_HOOKS
    static const int dbg = getenv("UEM_WGRAD_DBG") ? atoi(getenv("UEM_WGRAD_DBG")) : 0;
    p.dbg = dbg;
#else
    p.dbg = 0;
#endif
    const int tiles = (int)(uem_cdiv(p.Cout, TM) * p.KH * p.KW * uem_cdiv(p.Cin, TN));
    // split-K sizing.  The grid is sized to whole ROUNDS of the chip's resident-block slots (256 CUs x blocks per CU
    // at this tile's register footprint): equal-work blocks run in lock step, so 2048 blocks on 768 slots take 3
    // rounds at 89 % fill while 2304 take the same 3 rounds with 12 % less work each (+8 % on the 3x3 layers).
    // Every split adds one fp32-atomic pass over the whole filter bank (chip-wide atomic rate 1.3 TB/s); with few
    // output tiles that pass dominates (measured: 60 % of the kernel at 16 tiles x 128 splits), so there the number
    // of rounds is limited to what keeps >= ~1024 pixel rows per split.
    static const int forced = getenv("UEM_WGRAD_SPLITS") ? atoi(getenv("UEM_WGRAD_SPLITS")) : 0;
    static const int forced_rounds = getenv("UEM_WGRAD_ROUNDS") ? atoi(getenv("UEM_WGRAD_ROUNDS")) : 0;
    constexpr int BLOCKS_PER_CU = TM * TN >= 128 * 128 ? 3 : (TM * TN >= 32 * 128 ? 4 : 6);
    static const int bpc_env = getenv("UEM_WGRAD_BLOCKS_PER_CU") ? atoi(getenv("UEM_WGRAD_BLOCKS_PER_CU")) : 0;
    const int slots = 256 * (bpc_env > 0 && TM * TN >= 128 * 128 ? bpc_env : BLOCKS_PER_CU);
    const int max_splits = (int)uem_cdiv(p.M, 4 * BK);
    int rounds = 3;
    if (tiles <= 32) {
        rounds = (int)((int64_t)p.M * tiles / 1024 / slots);
        rounds = rounds < 1 ? 1 : (rounds > 3 ? 3 : rounds);
    }
    if (forced_rounds > 0) rounds = forced_rounds;
    int splits = slots * rounds / tiles;
    if (forced > 0) splits = forced;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rps = (int)uem_cdiv(p.M, splits);
    rps = (int)uem_cdiv(rps, BK) * BK;
    splits = (int)uem_cdiv(p.M, rps);
    p.rows_per_split = rps;
    const unsigned grid = (unsigned)tiles * (unsigned)splits;
    static const int lds_pad = getenv("UEM_WGRAD_LDS_PAD") ? atoi(getenv("UEM_WGRAD_LDS_PAD")) : 0;   // residency experiments
    if (affine) conv_wgrad_kernel<TM, TN, WM, WN, WK, MODE, true, PREC><<<grid, 256, lds_pad, st>>>(p);
    else conv_wgrad_kernel<TM, TN, WM, WN, WK, MODE, false, PREC><<<grid, 256, lds_pad, st>>>(p);
}

int uem_wgrad_dma_try(const float* x, const float* dy, const float* in_scale, const float* in_shift, float* dw,
                      const uem_conv_shape* s, int flags, hipStream_t st);      // wgrad.hip

extern "C" int uem_conv2d_wgrad(const float* x, const float* dy, const float* in_scale, const float* in_shift, float* dw,
                                const uem_conv_shape* s, int flags, void* stream) {
    UEM_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
    int rc = conv_check(s);
    if (rc) return rc;
    const bool affine = (flags & UEM_CONV_IN_AFFINE) != 0;
    UEM_REQUIRE(!affine || (in_scale && in_shift), "conv2d_wgrad: affine prologue needs scale/shift");
    UEM_REQUIRE(s->Cin % 32 == 0 && s->Cout % 4 == 0 && s->y_ld % 4 == 0, "conv2d_wgrad: Cin %% 32, Cout %% 4 required");
    // 1x1 and row-aligned 3x3 layers with 64-multiple channel counts: the LDS-DMA kernel (wgrad.hip); everything else
    // (ragged channel counts, 3x3 on rows that are not a multiple of 32 pixels, reduced operand precisions) stays here
    if (uem_wgrad_dma_try(x, dy, in_scale, in_shift, dw, s, flags, (hipStream_t)stream)) return uem_check_launch("conv2d_wgrad (dma)");
    WgradP p;
    p.x = x; p.dy = dy; p.in_scale = in_scale; p.in_shift = in_shift; p.dw = dw;
    p.M = s->N * s->Ho * s->Wo; p.N = s->N; p.H = s->H; p.W = s->W; p.Cin = s->Cin; p.Ho = s->Ho; p.Wo = s->Wo;
    p.Cout = s->Cout; p.KH = s->KH; p.KW = s->KW; p.stride = s->stride; p.pad = s->pad; p.dil = s->dil;
    p.x_ld = s->x_ld; p.dy_ld = s->y_ld; p.relu = (flags & UEM_CONV_IN_RELU) ? 1 : 0; p.rows_per_split = 0; p.dy_bf16 = 0;
    hipStream_t st = (hipStream_t)stream;
    UEM_REQUIRE(!(flags & 16), "conv2d_wgrad: flag 16 (the split-bf16 operand mode of rounds 1-3) is retired");
    const int prec = (flags & UEM_CONV_PREC_BF16) ? 2 : 0;
    if (s->Cout % 128 == 0 && s->Cin % 128 == 0) {
        if (prec == 2) wgrad_go<128, 128, 2, 2, 1, 0, 2>(p, affine, st);
        else wgrad_go<128, 128, 2, 2, 1, 0>(p, affine, st);
    } else if (s->Cout % 64 == 0 && s->Cin % 64 == 0) {
        if (prec == 2) wgrad_go<64, 64, 2, 2, 1, 0, 2>(p, affine, st);
        else wgrad_go<64, 64, 2, 2, 1, 0>(p, affine, st);
    } else if (s->Cin % 128 == 0) {
        if (prec == 2) wgrad_go<32, 128, 1, 4, 1, 0, 2>(p, affine, st);
        else wgrad_go<32, 128, 1, 4, 1, 0>(p, affine, st);
    } else wgrad_go<32, 32, 1, 1, 4, 0>(p, affine, st);      // tiny filters: always exact fp32
    return uem_check_launch("conv2d_wgrad");
}

static int stem_wgrad_impl(const float* x4, const float* dy, float* dw8, int N, int H, int W, int flags, void* stream, int dy_bf16 = 0);
extern "C" int uem_conv2d_stem_wgrad_bf16(const float* x4, const uint16_t* dy, float* dw8, int N, int H, int W, void* stream) {
    return stem_wgrad_impl(x4, reinterpret_cast<const float*>(dy), dw8, N, H, W, UEM_CONV_PREC_BF16, stream, 1);
}
extern "C" int uem_conv2d_stem_wgrad(const float* x4, const float* dy, float* dw8, int N, int H, int W, void* stream) {
    return stem_wgrad_impl(x4, dy, dw8, N, H, W, 0, stream);
}
extern "C" int uem_conv2d_stem_wgrad_prec(const float* x4, const float* dy, float* dw8, int N, int H, int W, int flags, void* stream) {
    UEM_REQUIRE((flags & ~UEM_CONV_PREC_BF16) == 0, "conv2d_stem_wgrad_prec: only precision flags are accepted");
    return stem_wgrad_impl(x4, dy, dw8, N, H, W, flags, stream);
}
static int stem_wgrad_impl(const float* x4, const float* dy, float* dw8, int N, int H, int W, int flags, void* stream, int dy_bf16) {
    UEM_REQUIRE(x4 && dy && dw8 && N > 0 && H >= 7 && W >= 7, "conv2d_stem_wgrad: bad arguments");
    WgradP p;
    p.dy_bf16 = dy_bf16;
    p.x = x4; p.dy = dy; p.in_scale = p.in_shift = nullptr; p.dw = dw8;
    p.N = N; p.H = H; p.W = W; p.Cin = 32; p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1; p.Cout = 64;
    p.KH = 7; p.KW = 1; p.stride = 2; p.pad = 3; p.dil = 1; p.x_ld = 4; p.dy_ld = 64; p.relu = 0; p.rows_per_split = 0;
    p.M = N * p.Ho * p.Wo;
    if (flags & UEM_CONV_PREC_BF16) wgrad_go<64, 32, 2, 1, 2, 2, 2>(p, false, (hipStream_t)stream);
    else wgrad_go<64, 32, 2, 1, 2, 2>(p, false, (hipStream_t)stream);
    return uem_check_launch("conv2d_stem_wgrad");
}

// =========================================================================================================
// small weight re-layout helpers
// =========================================================================================================
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int T, int Cin) {
    // w[o][t][i] -> wt[i][t][o]
    const int64_t total = (int64_t)Cout * T * Cin;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % Cout);
        int64_t r = idx / Cout;
        const int t = (int)(r % T);
        const int i = (int)(r / T);
        wt[idx] = w[((size_t)o * T + t) * Cin + i];
    }
}
__global__ void weight_transpose_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt, int Cout, int T, int Cin) {
    // w[o][t][i] (fp32 master) -> wt[i][t][o] (bf16, round to nearest even): the data gradient's filter bank on the bf16 path
    const int64_t total = (int64_t)Cout * T * Cin;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % Cout);
        int64_t r = idx / Cout;
        const int t = (int)(r % T);
        const int i = (int)(r / T);
        wt[idx] = f2bf(w[((size_t)o * T + t) * Cin + i]);
    }
}
extern "C" int uem_weight_transpose_bf16(const float* w, uint16_t* wt, int Cout, int KH, int KW, int Cin, void* stream) {
    UEM_REQUIRE(w && wt && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "weight_transpose_bf16: bad arguments");
    const int64_t total = (int64_t)Cout * KH * KW * Cin;
    weight_transpose_bf16_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(w, wt, Cout, KH * KW, Cin);
    return uem_check_launch("weight_transpose_bf16");
}
extern "C" int uem_weight_transpose(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream) {
    UEM_REQUIRE(w && wt && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "weight_transpose: bad arguments");
    const int64_t total = (int64_t)Cout * KH * KW * Cin;
    weight_transpose_kernel<<<uem_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>(w, wt, Cout, KH * KW, Cin);
    return uem_check_launch("weight_transpose");
}
__global__ void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ w8) {
    // w[64][7][7][3] (OHWI) -> w8[64][7][8][4], zero padded
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * 7 * 8 * 4) return;
    const int c = idx & 3, kx = (idx >> 2) & 7, ky = (idx >> 5) % 7, o = idx / (7 * 32);
    w8[idx] = (c < 3 && kx < 7) ? w[((o * 7 + ky) * 7 + kx) * 3 + c] : 0.f;
}
extern "C" int uem_stem_pack_weight(const float* w_ohwi, float* w8, void* stream) {
    UEM_REQUIRE(w_ohwi && w8, "stem_pack_weight: null pointer");
    stem_pack_kernel<<<(64 * 7 * 32 + 255) / 256, 256, 0, (hipStream_t)stream>>>(w_ohwi, w8);
    return uem_check_launch("stem_pack_weight");
}
__global__ void stem_unpack_kernel(const float* __restrict__ dw8, float* __restrict__ dw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * 7 * 7 * 3) return;
    const int c = idx % 3, kx = (idx / 3) % 7, ky = (idx / 21) % 7, o = idx / 147;
    dw[idx] += dw8[((o * 7 + ky) * 8 + kx) * 4 + c];
}
extern "C" int uem_stem_unpack_grad(const float* dw8, float* dw_ohwi, void* stream) {
    UEM_REQUIRE(dw8 && dw_ohwi, "stem_unpack_grad: null pointer");
    stem_unpack_kernel<<<(64 * 147 + 255) / 256, 256, 0, (hipStream_t)stream>>>(dw8, dw_ohwi);
    return uem_check_launch("stem_unpack_grad");
}
__global__ void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int C, int ld) {
    // one block per channel group of 64... C is tiny (<= 32): block = 256 threads over rows, loop channels
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int m = threadIdx.x; m < M; m += 256) s += dy[(size_t)m * ld + c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) db[c] += (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" int uem_bias_grad(const float* dy, float* db, int M, int C, int ld, void* stream) {
    UEM_REQUIRE(dy && db && M > 0 && C > 0 && ld >= C, "bias_grad: bad arguments");
    bias_grad_kernel<<<C, 256, 0, (hipStream_t)stream>>>(dy, db, M, C, ld);
    return uem_check_launch("bias_grad");
}

// =========================================================================================================
// ASPP heads as ONE dense GEMM + a gather (Encoder.py:68-84).
//   sum_d conv3x3_dil_d(feat; W_d)[p] = sum_d sum_tap (feat . W_d,tap)[p + off_d,tap]
// so G = feat x Wall (a 1x1 conv with R = n_dil*9*K2 output columns, K2 = heads*classes) is computed once
// by the MFMA kernel (feat read ONCE instead of 36 times, no 12->32 column padding), and the dilated
// 3x3 structure becomes a 36-term gather of K2-float rows of G.  Backward: dG is the transposed gather
// of dOut, then dfeat / dWall are plain 1x1 dgrad / wgrad GEMMs.
// G column r = (d*9 + ky*3 + kx)*K2 + j,  j = head*C + class.
// =========================================================================================================
struct AsppP {
    int N, h, w, K2, R, nd;
    int dil[8];
};
__global__ __launch_bounds__(256) void aspp_gather_fwd_kernel(const float* __restrict__ G, const float* __restrict__ bias,
                                                              float* __restrict__ out, float* __restrict__ out2, const AsppP p) {
    // one thread per (pixel, j); out is (N,h,w,K2), or with out2 two (N,h,w,K2/2) tensors: head 0 -> out, head 1 -> out2
    const int64_t total = (int64_t)p.N * p.h * p.w * p.K2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % p.K2);
        int64_t t = i / p.K2;
        const int x = (int)(t % p.w); t /= p.w;
        const int y = (int)(t % p.h);
        const int n = (int)(t / p.h);
        float acc = 0.f;
        for (int d = 0; d < p.nd; ++d) {
            acc += bias[d * p.K2 + j];
            const int dl = p.dil[d];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + (ky - 1) * dl;
                if (yy < 0 || yy >= p.h) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + (kx - 1) * dl;
                    if (xx < 0 || xx >= p.w) continue;
                    acc += G[(((size_t)n * p.h + yy) * p.w + xx) * p.R + (size_t)((d * 9 + ky * 3 + kx) * p.K2 + j)];
                }
            }
        }
        if (out2 == nullptr) out[i] = acc;
        else {
            const int Ch = p.K2 >> 1;
            const size_t pix = (size_t)(i / p.K2);
            if (j < Ch) out[pix * Ch + j] = acc;
            else out2[pix * Ch + (j - Ch)] = acc;
        }
    }
}
__global__ __launch_bounds__(256) void aspp_gather_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ dout2,
                                                              float* __restrict__ dG, const AsppP p) {
    // one thread per (pixel q, column r): dG[q][r] = dOut[q - off][j] when that pixel exists (its tap lands on q)
    const int64_t total = (int64_t)p.N * p.h * p.w * p.R;
    const int used = p.nd * 9 * p.K2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i % p.R);
        float v = 0.f;
        if (r < used) {
            int64_t t = i / p.R;
            const int x = (int)(t % p.w); t /= p.w;
            const int y = (int)(t % p.h);
            const int n = (int)(t / p.h);
            const int j = r % p.K2, tap = (r / p.K2) % 9, d = r / (9 * p.K2);
            const int yo = y - (tap / 3 - 1) * p.dil[d], xo = x - (tap % 3 - 1) * p.dil[d];
            if (yo >= 0 && yo < p.h && xo >= 0 && xo < p.w) {
                const size_t pix = ((size_t)n * p.h + yo) * p.w + xo;
                const int Ch = p.K2 >> 1;
                v = dout2 == nullptr ? dout[pix * p.K2 + j] : (j < Ch ? dout[pix * Ch + j] : dout2[pix * Ch + (j - Ch)]);
            }
        }
        dG[i] = v;
    }
}
static int aspp_params(AsppP* p, int N, int h, int w, int K2, int R, int nd, const int* dil) {
    if (N <= 0 || h <= 0 || w <= 0 || K2 <= 0 || nd <= 0 || nd > 8 || !dil || R < nd * 9 * K2)
        return uem_fail(UEM_ERR_INVALID, "aspp_gather: bad shape (R=%d must be >= nd*9*K2=%d)", R, nd * 9 * K2);
    p->N = N; p->h = h; p->w = w; p->K2 = K2; p->R = R; p->nd = nd;
    for (int i = 0; i < 8; ++i) p->dil[i] = i < nd ? dil[i] : 0;
    return UEM_OK;
}
extern "C" int uem_aspp_gather_fwd(const float* G, const float* bias, float* out, float* out2, int N, int h, int w, int K2, int R,
                                   int nd, const int* dil, void* stream) {
    UEM_REQUIRE(G && bias && out && (out2 == nullptr || K2 % 2 == 0), "aspp_gather_fwd: null pointer / odd K2 with two outputs");
    AsppP p;
    int rc = aspp_params(&p, N, h, w, K2, R, nd, dil);
    if (rc) return rc;
    aspp_gather_fwd_kernel<<<uem_stream_grid((int64_t)N * h * w * K2, 256), 256, 0, (hipStream_t)stream>>>(G, bias, out, out2, p);
    return uem_check_launch("aspp_gather_fwd");
}
extern "C" int uem_aspp_gather_bwd(const float* dout, const float* dout2, float* dG, int N, int h, int w, int K2, int R, int nd,
                                   const int* dil, void* stream) {
    UEM_REQUIRE(dout && dG && (dout2 == nullptr || K2 % 2 == 0), "aspp_gather_bwd: null pointer / odd K2 with two inputs");
    AsppP p;
    int rc = aspp_params(&p, N, h, w, K2, R, nd, dil);
    if (rc) return rc;
    aspp_gather_bwd_kernel<<<uem_stream_grid((int64_t)N * h * w * R, 256), 256, 0, (hipStream_t)stream>>>(dout, dout2, dG, p);
    return uem_check_launch("aspp_gather_bwd");
}

// The 2 heads x nd dilations' filters <-> the GEMM's filter bank, one launch each way (round 2 did it with 16 + 24 torch copy / add
// launches per forward / backward).  Wall row (d*9 + tap)*2C + head*C + c = W[head][d][c][tap][:]; bias [nd][2][C].
struct AsppPtrs {
    float* w[8];        // [head*nd + d]: (C,3,3,cin) OHWI filters (pack: read; unpack: gradient, accumulated into)
    float* b[8];        // (C,) biases / their gradients
};
__global__ __launch_bounds__(256) void aspp_pack_kernel(const AsppPtrs ptrs, float* __restrict__ wall, float* __restrict__ bias,
                                                        const int C, const int cin, const int nd, const int R, const int nh) {
    // nh heads (1: Deeplabv2's single-head default and its cascade branch; 2: the layer5 / layer6 pair): row (d*9 + tap)*nh*C + head*C + c
    const int cq = cin / 4;
    const int64_t total = (int64_t)R * cq;
    const int used = nd * 9 * nh * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cq), ci = (int)(i - (int64_t)r * cq) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                        // rows beyond the used columns: zero filters
        if (r < used) {
            const int c = r % C, head = (r / C) % nh, tap = (r / (nh * C)) % 9, d = r / (9 * nh * C);
            v = *reinterpret_cast<const float4*>(ptrs.w[head * nd + d] + ((size_t)c * 9 + tap) * cin + ci);
        }
        *reinterpret_cast<float4*>(wall + (size_t)r * cin + ci) = v;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < nd * nh * C) {
        const int t = threadIdx.x, c = t % C, head = (t / C) % nh, d = t / (nh * C);
        bias[t] = ptrs.b[head * nd + d][c];
    }
}
__global__ __launch_bounds__(256) void aspp_unpack_grad_kernel(const float* __restrict__ dwall, const float* __restrict__ db,
                                                               const AsppPtrs ptrs, const int C, const int cin, const int nd, const int nh) {
    const int cq = cin / 4;
    const int used = nd * 9 * nh * C;
    const int64_t total = (int64_t)used * cq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cq), ci = (int)(i - (int64_t)r * cq) * 4;
        const int c = r % C, head = (r / C) % nh, tap = (r / (nh * C)) % 9, d = r / (9 * nh * C);
        // atomics: two heads may be ONE module (a caller passing the same Classifier_Module twice), their rows then add into
        // the same gradient buffer from different threads
        float* dst = ptrs.w[head * nd + d] + ((size_t)c * 9 + tap) * cin + ci;
        const float4 g = *reinterpret_cast<const float4*>(dwall + (size_t)r * cin + ci);
        atomicAdd(dst + 0, g.x); atomicAdd(dst + 1, g.y); atomicAdd(dst + 2, g.z); atomicAdd(dst + 3, g.w);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < nd * nh * C) {                // every dilation's bias sees the same output gradient
        const int t = threadIdx.x, c = t % C, head = (t / C) % nh, d = t / (nh * C);
        atomicAdd(&ptrs.b[head * nd + d][c], db[head * C + c]);
    }
}
static int aspp_ptrs(AsppPtrs* q, void* const* w, void* const* b, int C, int cin, int nd, int nh, const char* what) {
    if (!w || !b || C <= 0 || cin <= 0 || cin % 4 != 0 || nd <= 0 || nd > 4 || nh < 1 || nh > 2 || nd * nh * C > 256)
        return uem_fail(UEM_ERR_INVALID, "%s: bad arguments (nd <= 4, 1 <= heads <= 2, nd*heads*C <= 256, cin %% 4 == 0)", what);
    for (int i = 0; i < 8; ++i) {
        q->w[i] = i < nh * nd ? (float*)w[i] : nullptr;
        q->b[i] = i < nh * nd ? (float*)b[i] : nullptr;
        if (i < nh * nd && (!q->w[i] || !q->b[i] || ((uintptr_t)q->w[i] & 15))) return uem_fail(UEM_ERR_INVALID, "%s: null or misaligned filter pointer", what);
    }
    return UEM_OK;
}
extern "C" int uem_aspp_pack(void* const* w, void* const* b, float* wall, float* bias, int C, int cin, int nd, int R, int nheads,
                             void* stream) {
    UEM_REQUIRE(wall && bias && nheads >= 1 && nheads <= 2 && R >= nd * 9 * nheads * C, "aspp_pack: bad arguments");
    AsppPtrs q;
    const int rc = aspp_ptrs(&q, w, b, C, cin, nd, nheads, "aspp_pack");
    if (rc) return rc;
    aspp_pack_kernel<<<uem_stream_grid((int64_t)R * (cin / 4), 256), 256, 0, (hipStream_t)stream>>>(q, wall, bias, C, cin, nd, R, nheads);
    return uem_check_launch("aspp_pack");
}
extern "C" int uem_aspp_unpack_grad(const float* dwall, const float* db, void* const* gw, void* const* gb, int C, int cin, int nd,
                                    int nheads, void* stream) {
    UEM_REQUIRE(dwall && db, "aspp_unpack_grad: null pointer");
    AsppPtrs q;
    const int rc = aspp_ptrs(&q, gw, gb, C, cin, nd, nheads, "aspp_unpack_grad");
    if (rc) return rc;
    aspp_unpack_grad_kernel<<<uem_stream_grid((int64_t)nd * 9 * nheads * C * (cin / 4), 256), 256, 0, (hipStream_t)stream>>>(dwall, db, q, C, cin, nd,
                                                                                                                       nheads);
    return uem_check_launch("aspp_unpack_grad");
}

// =========================================================================================================
// bf16-storage entry points (BASELINE config 5)
// =========================================================================================================
static int g_bf16_persist = -1;  // tuning override: 1 / 0 = persistent blocks on / off, -1 = rule
extern "C" void uemdbg_conv_bf16_persist(int v) { g_bf16_persist = v; }
template <int BN_, int MODE, int EPI, bool PERSIST, int BMT = 128, int NST = 2, bool SPLIT = false>
static void conv_bf16_launch(const ConvP& p, unsigned xb, unsigned wb, hipStream_t st) {
    using C = ConvBf16Cfg<BN_, PERSIST, BMT, NST>;
    const int ntiles = (int)uem_cdiv(p.M, BMT) * (p.Cout / BN_);
    // persistent: one block per resident-block slot (LDS: one 256-row, two 128-wide or three 64-wide blocks per CU; a multiple of 8,
    // so a block's tiles stay on its XCD's share)
    const int per_cu = BMT == 256 ? 1 : (160 * 1024 / C::LDS_BYTES > 3 ? 3 : 160 * 1024 / C::LDS_BYTES);
    const int slots = 256 * per_cu;
    const int grid = PERSIST && ntiles > slots ? slots : ntiles;
    auto k = conv_bf16_kernel<BN_, MODE, EPI, PERSIST, BMT, NST, SPLIT>;
    if (uem_allow_lds((const void*)k, C::LDS_BYTES)) k<<<grid, 2 * BMT, C::LDS_BYTES, st>>>(p, xb, wb, ntiles);
}
// =========================================================================================================
// Pointwise bf16 GEMM with a four-stage operand ring, loader / storer waves and streamed stores (round 6).
//   C[M x N] (bf16) = A[M x K] (bf16, row stride lda) x B[N x K]^T (bf16, dense rows), fp32 accumulation -- the 1x1 / stride-1
//   convolutions of the bf16-storage path, forward (optionally with the BatchNorm tile statistics of the rounded output) and plain
//   data gradient (A = dy, B = the transposed bank).
// Why: on these layers a tile is a handful of k-steps around an epilogue that costs more than they do; the ablation
// and four variants of conv_bf16_kernel said what a block needs AT ONCE -- operand pieces several k-steps ahead (one block of eight waves
// per CU has nobody else to cover a round trip), no wave that waits for operand pieces behind its own stores (one in-order vector-memory
// counter per wave), and stores that leave under the next tile's MFMAs instead of as a burst the whole chip drains while the matrix pipe
// idles -- and that 160 KB of LDS hold a ring AND a parked tile only at 32-channel k-steps:
//   * tile 256 x 128, k-step 32 channels: a stage is A 256 x 64 B + B 128 x 64 B = 24 KB, four stages (three k-steps in flight) = 96 KB;
//     the tile's output, rounded to bf16, is parked in a 64 KB out-buffer: 160 KB, one block per CU;
//   * rows of 64 bytes, their four 16-byte chunks XOR-swizzled by (row >> 2) & 3 on the DMA source address and on the read: the 16 lanes of
//     a ds_read_b128 group cover 16 distinct slots of the 256-byte bank row;
//   * waves 0-3 (LOADERS) issue all 24 pieces of a k-step, six each, and wait with vmcnt(12): the pieces of the step about to be consumed
//     have landed, the next two steps' stay in flight, and nothing else is ever in their queue; waves 4-7 (STORERS) read the parked tile
//     back 16 bytes per lane (hand-written ds_read: the compiler would order an LDS read behind every DMA piece in flight) and store it,
//     a few rows per k-step of the NEXT tile, and never wait on the vector-memory counter; all eight run the MFMAs (wave tile 64 x 64);
//   * the ring runs across the block's tiles (persistent blocks, XCD-aware tile order, column tiles of one row tile adjacent).
// Same accumulation order over k as conv_bf16_kernel (16-channel MFMA steps in ascending order): outputs are bit-equal to it.
// =========================================================================================================
struct PwP {
    const unsigned short* A; const unsigned short* B; unsigned short* C;
    float* tile_stats;                    // [2][N][M / 128] or null (forward: BatchNorm statistics; tail: the BatchNorm-backward partial sums)
    int M, N, K, lda, ldc;
    unsigned a_bytes, b_bytes;
    // fused data-gradient tail (pw_bf16_areg_kernel<..., TAIL>): the tensor accumulated into + its gate bits, the BatchNorm input + its mask bits
    const unsigned short* acc; const unsigned* acc_bits; const unsigned short* bn_z; const float* bn_vec; const unsigned* bn_bits;
};
template <bool STATS>
__global__ __launch_bounds__(512, 1) void pw_bf16_stream_kernel(const PwP p, const int ntiles) {
    constexpr int PBM = 256, PBN = 128, PBK = 32, PNS = 4;
    constexpr int A_ELEMS = PBM * PBK, B_ELEMS = PBN * PBK, SE = A_ELEMS + B_ELEMS;          // bf16 elements per stage (24 KB)
    constexpr int NPC = 6;                                                                // pieces per loader wave and k-step: 4 A + 2 B
    constexpr int NSL = (PBM * PBN / 8) / 256;                                             // 16-byte slices per storer thread and tile: 16
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* const lds16 = reinterpret_cast<unsigned short*>(smem);
    unsigned short* const ob = lds16 + PNS * SE;                                          // [256][128] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave < 4;
    const int tiles_n = p.N / PBN;
    const int KT = p.K / PBK;
    const i32x4 rs_a = conv_rsrc(p.A, p.a_bytes), rs_b = conv_rsrc(p.B, p.b_bytes);
    // ---- issue side (loader waves): three k-steps ahead of the MFMAs, moving on to the block's next tile on its own
    const int lrow = lane >> 2;                                                           // row inside a 16-row piece
    int vi = blockIdx.x, ikt = 0;
    bool issue_live = vi < ntiles;
    unsigned aoff[4], boff[2];
    auto issue_setup = [&]() {
        const int tile = xcd_remap(vi, ntiles);
        const int im0 = (tile / tiles_n) * PBM, in0 = (tile % tiles_n) * PBN;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (j * 4 + wave) * 16 + lrow;                                     // row of the tile's A block = LDS row
            aoff[j] = ((unsigned)(im0 + r) * (unsigned)p.lda + (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) * 8)) * 2u;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (j * 4 + wave) * 16 + lrow;
            boff[j] = ((unsigned)(in0 + r) * (unsigned)p.K + (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) * 8)) * 2u;
        }
    };
    if (is_loader && issue_live) issue_setup();
    auto issue_step = [&](unsigned short* __restrict__ fill) {                            // six pieces, always (out of range when nothing is left)
        unsigned short* const fA = fill;
        unsigned short* const fB = fill + A_ELEMS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uem_raw_buffer_load_lds(rs_a, (lds_u32p)(fA + (j * 4 + wave) * 512), 16, (int)(issue_live ? aoff[j] : CONV_OOB), 0, 0, 0);
            aoff[j] += PBK * 2;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uem_raw_buffer_load_lds(rs_b, (lds_u32p)(fB + (j * 4 + wave) * 512), 16, (int)(issue_live ? boff[j] : CONV_OOB), 0, 0, 0);
            boff[j] += PBK * 2;
        }
        if (issue_live && ++ikt == KT) {
            ikt = 0;
            vi += gridDim.x;
            issue_live = vi < ntiles;
            if (issue_live) issue_setup();
        }
    };
    // ---- compute side
    f32x16 acc[2][2];
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fr = lane & 31, fh = lane >> 5;
    // ---- storer side: the parked tile leaves `per_step` slices per k-step
    const int stid = (tid - 256) & 255;
    int pm0 = 0, pn0 = 0, tr_left = 0;
    const int per_step = (NSL + KT - 1) / KT;
    auto trickle = [&](int nsend) {
        while (nsend > 0 && tr_left > 0) {
            f32x4v v[4];
            unsigned go[4];
            int nb = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < nsend && tr_left > 0) {
                    const int c = (NSL - tr_left) * 256 + stid, row = c >> 4, ch = c & 15;
                    const unsigned la = (unsigned)(unsigned long long)(lds_u32p)(ob + row * PBN + ch * 8);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(la));
                    go[u] = (unsigned)(pm0 + row) * (unsigned)p.ldc + (unsigned)(pn0 + ch * 8);
                    --tr_left;
                    nb = u + 1;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < nb) *reinterpret_cast<f32x4v*>(p.C + (size_t)go[u]) = v[u];
            nsend -= 4;
        }
    };
    // ---- prologue: three k-steps in flight
    if (is_loader) {
        issue_step(lds16 + 0 * SE);
        issue_step(lds16 + 1 * SE);
        issue_step(lds16 + 2 * SE);
    }
    int cs = 0;                                                                           // stage the next k-step consumes
    for (int vc = blockIdx.x; vc < ntiles; vc += gridDim.x) {
        const int tile = xcd_remap(vc, ntiles);
        const int m0 = (tile / tiles_n) * PBM, n0 = (tile % tiles_n) * PBN;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
            if (is_loader) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NPC) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned short* __restrict__ const As = lds16 + cs * SE;
            const unsigned short* __restrict__ const Bs = As + A_ELEMS;
            if (is_loader) issue_step(lds16 + ((cs + PNS - 1) & (PNS - 1)) * SE);          // the stage consumed last: every wave is past it
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = wm + i * 32 + fr;
                    a[i] = *reinterpret_cast<const bf16x8*>(&As[r * PBK + (((ks * 2 + fh) ^ ((r >> 2) & 3)) * 8)]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = wn + j * 32 + fr;
                    b[j] = *reinterpret_cast<const bf16x8*>(&Bs[r * PBK + (((ks * 2 + fh) ^ ((r >> 2) & 3)) * 8)]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                if (!is_loader && tr_left > 0) trickle(ks == 0 ? (per_step + 1) / 2 : per_step / 2);
            }
            cs = (cs + 1) & (PNS - 1);
        }
        // ---- tile end: no epilogue phase.  Whatever of the previous tile is still parked leaves now (nothing, when KT x per_step >= 16)
        if (!is_loader) trickle(NSL);
        float s1[2], s2[2];
        if constexpr (STATS) {
            // BatchNorm tile statistics of the ROUNDED values: the lane's two columns over its 32 rows, then the lane halves
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s1[j] = s2[j] = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float vr = bf2f(f2bf(acc[i][j][r])); s1[j] += vr; s2[j] = fmaf(vr, vr, s2[j]); }
                s1[j] += __shfl_xor(s1[j], 32);
                s2[j] += __shfl_xor(s2[j], 32);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                                     // the out-buffer is free: every storer past its reads
        asm volatile("" ::: "memory");
        if constexpr (STATS) {
            float* const sc = reinterpret_cast<float*>(ob);                               // [8 waves][2][64], in the still empty out-buffer
            if (fh == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { sc[(wave * 2 + 0) * 64 + j * 32 + fr] = s1[j]; sc[(wave * 2 + 1) * 64 + j * 32 + fr] = s2[j]; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!is_loader) {                                                             // entry e (128 rows) x 128 columns, both sums, by the storers
                const int e = stid >> 7, col = stid & 127, wni = col >> 6, cc = col & 63;
                const unsigned l0 = (unsigned)(unsigned long long)(lds_u32p)(sc + (((2 * e) * 2 + wni) * 2) * 64 + cc);
                float a0, a1, b0, b1;                     // row groups 2e and 2e + 1 are two waves = 256 floats apart; sum / sum of squares 64 floats apart
                asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:1024\n\tds_read_b32 %3, %4 offset:1280\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1) : "v"(l0) : "memory");
                const size_t tiles_m = (size_t)(p.M / 128);
                p.tile_stats[((size_t)0 * p.N + n0 + col) * tiles_m + (size_t)(m0 / 128 + e)] = a0 + a1;
                p.tile_stats[((size_t)1 * p.N + n0 + col) * tiles_m + (size_t)(m0 / 128 + e)] = b0 + b1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ob[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * PBN + wn + j * 32 + fr] = f2bf(acc[i][j][r]);
        pm0 = m0; pn0 = n0; tr_left = NSL;                // the next k-step's barrier orders the dump before the storers' reads
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!is_loader) trickle(NSL);                          // the block's last tile
}
// -1 = rule, 0 = never, 1 = wherever legal
static int g_bf16_pw = -1;
extern "C" void uemdbg_conv_bf16_pw(int v) { g_bf16_pw = v; }
template <int MODE>
static bool conv_bf16_pw_try(const ConvP& p, unsigned xb, unsigned wb, hipStream_t st) {
    static const int env = getenv("UEM_CONV_BF16_PW") ? atoi(getenv("UEM_CONV_BF16_PW")) : -1;
    const int set = g_bf16_pw >= 0 ? g_bf16_pw : env;
    const bool pointwise = p.KH * p.KW == 1 && p.pad == 0 && ((MODE == 1) ? (p.sub == 1 && p.stride == 1) : p.stride == 1);
    if (set == 0 || !pointwise || p.M % 256 != 0 || p.Cout % 128 != 0 || p.Cin % 32 != 0 || p.accumulate) return false;
    if (MODE == 1 && (p.acc_src || p.tile_bnbwd)) return false;
    if ((double)p.M * p.y_ld * 2.0 >= 4294967280.0) return false;
    const int ntiles = (p.M / 256) * (p.Cout / 128);
    // Rule: OFF (UEM_CONV_BF16_PW=1 / uemdbg_conv_bf16_pw(1): wherever legal).  Bit-equal to conv_bf16_kernel and within +-5 % of the round-5
    // dispatch on the 1024 x 1024 pointwise layers, +10 ... +20 % on the 512 x 512 ones (scripts/sweep_conv_bf16_ring.py SWEEP=pw,
    // profiles/r06_i_*): the sixth block structure to land on the same time -- what bounds these layers is the operand bytes a CU pulls
    // from L2 per flop, which no pipeline inside the block changes; the tile shape does (conv_bf16_big_bn).
    if (set != 1) return false;
    PwP q;
    q.A = reinterpret_cast<const unsigned short*>(p.x); q.B = reinterpret_cast<const unsigned short*>(p.w);
    q.C = reinterpret_cast<unsigned short*>(p.y); q.tile_stats = MODE == 0 ? p.tile_stats : nullptr;
    q.M = p.M; q.N = p.Cout; q.K = p.Cin; q.lda = p.x_ld; q.ldc = p.y_ld; q.a_bytes = xb; q.b_bytes = wb;
    constexpr int LDS = 4 * (256 * 32 + 128 * 32) * 2 + 256 * 128 * 2;                    // 160 KB
    const int grid = ntiles > 256 ? 256 : ntiles;
    if (q.tile_stats != nullptr) {
        auto k = pw_bf16_stream_kernel<true>;
        if (!uem_allow_lds((const void*)k, LDS)) return false;
        k<<<grid, 512, LDS, st>>>(q, ntiles);
    } else {
        auto k = pw_bf16_stream_kernel<false>;
        if (!uem_allow_lds((const void*)k, LDS)) return false;
        k<<<grid, 512, LDS, st>>>(q, ntiles);
    }
    return true;
}
// =========================================================================================================
// A-STATIONARY pointwise bf16 block (round 6, late): pointwise layers with a SHORT reduction and MANY output columns -- conv3 of a
// bottleneck forward (C -> 4C) and the plain data gradient of the same shape.  The six block structures above all fetch, per 256 x 128
// output tile, a 256 x K slab of A and a 128 x K slab of B from L2: (A + B) bytes per flop is what bounds these layers (DESIGN 3.3,
// round 6), and no pipeline inside the block changes it.  Here a block of eight waves owns 512 ROWS and ALL columns:
//   * each wave loads its 64 x K rows of A ONCE, straight into registers in the MFMA operand layout (K = 256: 128 VGPRs), and keeps
//     them for the N / 64 column tiles of the panel;
//   * B travels as whole 64-column tiles (64 x K bf16 = 32 KB at K = 256), double-buffered, by LDS-DMA, every wave issuing its share
//     and waiting for it by count (the stores of the previous tile are younger than the pieces it waits for); ONE barrier per tile;
//   * a wave's 64 x 64 results go through a wave-private 8 KB staging area (no block barrier) and leave as 16-byte row segments;
//   * BatchNorm tile statistics (sums of the rounded values per 128 rows): per-wave column sums, paired through a small scratch
//     behind the next tile's barrier.
// Operand bytes per 512 x N panel: 512 K (A) + N K (B) against 4 (N / 128) (256 K + 128 K) for the 256 x 128 tiles: 3-4x fewer.
// Same accumulation order over k as conv_bf16_kernel: outputs bit-equal; the statistics' summation order differs (fp32, 1e-6).
// =========================================================================================================
// TAIL (the fused data gradient of conv1 taken backwards: K = C, N = 4C): 0 none; bit 0: dx += gate(acc) -- the identity gradient through the
// packed ReLU mask (or the tensor itself where no mask is given); bit 1: the BatchNorm-backward partial sums of the layer dx feeds, over the
// ROUNDED dx, its mask from packed bits.  The wave's results are then staged in fp32, 32 rows at a time, and every 16-byte row segment is
// combined with the 16-byte segments of acc / z it needs -- each pass's loads issued one pass ahead of its arithmetic and stores, so that no
// wait for a load ever implies a wait for a store.  dx is bit-equal to conv_bf16_kernel's, the partial sums equal to their summation order.
template <int KC, bool STATS, int TAIL = 0>
__global__ __launch_bounds__(512, 1) void pw_bf16_areg_kernel(const PwP p, const int npanels, const int ngroups) {
    constexpr int QN = 64, NKS = KC / 32, B_ELEMS = QN * KC, NPW = (NKS * 4) / 8, NSTR = 8;
    constexpr bool T_ACC = (TAIL & 1) != 0, T_BN = (TAIL & 2) != 0, SUMS = STATS || T_BN;
    constexpr int SLD = 68;                                                               // floats per staged row (tail)
    static_assert(NPW >= 1, "at least one DMA piece per wave and B tile");
    static_assert(!(STATS && TAIL), "forward statistics or a data-gradient tail");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* const lds16 = reinterpret_cast<unsigned short*>(smem);
    unsigned short* const ob = lds16 + 2 * B_ELEMS;                                       // [8 waves][64][64] bf16 | tail: [8 waves][32][SLD] fp32
    float* const sc = reinterpret_cast<float*>(ob + (TAIL ? 8 * 32 * SLD * 2 : 8 * 4096)); // [2][8 waves][2][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5, lrow = lane >> 2;
    const int nt = p.N / QN;
    const i32x4 rs_b = conv_rsrc(p.B, p.b_bytes);
    unsigned short* const ow = ob + wave * 4096;
    // tail: buffer resources of the epilogue's streams, and the table [2][N] of the BatchNorm's mean / inverse deviation in LDS
    float* const vt = sc + 2 * 8 * 2 * 64;
    const bool has_ab = T_ACC && p.acc_bits != nullptr;
    __amdgpu_buffer_rsrc_t rs_acc, rs_ab, rs_z, rs_zb, rs_c;
    if constexpr (TAIL != 0) {
        const unsigned cbytes = (unsigned)(((size_t)p.M - 1) * p.ldc * 2 + (size_t)p.N * 2), zbytes = (unsigned)((size_t)p.M * p.N * 2), bbytes = (unsigned)((size_t)p.M * p.N / 8);
        rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, cbytes, 0x00020000);
        rs_acc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(T_ACC ? p.acc : p.C), 0, cbytes, 0x00020000);
        rs_ab = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(has_ab ? p.acc_bits : reinterpret_cast<const unsigned*>(p.C)), 0, has_ab ? bbytes : 0u, 0x00020000);
        rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(T_BN ? p.bn_z : p.C), 0, T_BN ? zbytes : 0u, 0x00020000);
        rs_zb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(T_BN ? p.bn_bits : reinterpret_cast<const unsigned*>(p.C)), 0, T_BN ? bbytes : 0u, 0x00020000);
        if constexpr (T_BN) {
            for (int i = tid; i < 2 * p.N; i += 512) vt[i] = p.bn_vec[2 * p.N + i];        // rows 2 (mean) and 3 (inverse deviation) of bn_vec [4][N]
        }
    }
    // ---- B tile `t` into stage `st`: this wave's NPW pieces (16 columns x 32 channels each)
    auto issue_b = [&](int t, int st) {
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int q = wave * NPW + u, ksl = q >> 2, c = (q & 3) * 16 + lrow;
            const unsigned off = ((unsigned)(t * QN + c) * (unsigned)KC + (unsigned)(ksl * 32 + (((lane & 3) ^ ((c >> 2) & 3)) * 8))) * 2u;
            uem_raw_buffer_load_lds(rs_b, (lds_u32p)(lds16 + st * B_ELEMS + ksl * 2048 + (q & 3) * 512), 16, (int)off, 0, 0, 0);
        }
    };
    auto write_stats = [&](int buf, int m0, int n0) {                                      // 4 entries of 128 rows x 64 columns, by threads 0..255
        if (tid < 256) {
            const int e = tid >> 6, col = tid & 63;
            // waves 2e and 2e + 1 hold the two 64-row halves of entry e: [wave][sum | sum of squares][64 columns]; hand-written reads (the
            // compiler would put an LDS read behind every DMA piece and store in flight)
            const unsigned l0 = (unsigned)(unsigned long long)(lds_u32p)(sc + ((buf * 8 + 2 * e) * 2) * 64 + col);
            float a0, b0, a1, b1;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1) : "v"(l0) : "memory");
            const size_t tiles_m = (size_t)(p.M / 128);
            p.tile_stats[((size_t)0 * p.N + n0 + col) * tiles_m + (size_t)(m0 / 128 + e)] = a0 + a1;
            p.tile_stats[((size_t)1 * p.N + n0 + col) * tiles_m + (size_t)(m0 / 128 + e)] = b0 + b1;
        }
    };
    // Work items: (row panel, column group).  With fewer panels than CUs the N / 64 column tiles of a panel are shared out over `ngroups`
    // blocks (each loads the panel's A rows itself: A traffic x ngroups, still one pass over B per block); ngroups = 1 otherwise.
    const int nitems = npanels * ngroups, ntg = nt / ngroups;
    int it = 0, pm0 = 0, pn0 = 0;
    bool have_prev = false;
    if ((int)blockIdx.x < nitems) issue_b(((int)blockIdx.x % ngroups) * ntg, 0);
    for (int vi = blockIdx.x; vi < nitems; vi += gridDim.x) {
        const int m0 = (vi / ngroups) * 512, wr0 = m0 + wave * 64, t_lo = (vi % ngroups) * ntg, t_hi = t_lo + ntg;
        bf16x8 a[2][KC / 16];
        bool a_loaded = false;
        for (int t = t_lo; t < t_hi; ++t, ++it) {
            const int st = it & 1;
            if (it == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTR) : "memory");   // this tile's pieces are older than the last tile's stores
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (SUMS) {
                if (have_prev) write_stats((it - 1) & 1, pm0, pn0);
            }
            {                                                                             // the next tile of this block, if any, into the other stage
                const bool more_t = t + 1 < t_hi;
                const int vn = vi + (int)gridDim.x;
                if (more_t || vn < nitems) issue_b(more_t ? t + 1 : (vn % ngroups) * ntg, st ^ 1);
            }
            if (!a_loaded) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ss = 0; ss < KC / 16; ++ss)
                        a[i][ss] = *reinterpret_cast<const bf16x8*>(p.A + (size_t)(wr0 + i * 32 + fr) * p.lda + ss * 16 + fh * 8);
                a_loaded = true;
            }
            f32x16 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            const unsigned short* __restrict__ const Bs = lds16 + st * B_ELEMS;
#pragma unroll
            for (int ss = 0; ss < KC / 16; ++ss) {
                bf16x8 b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = j * 32 + fr;
                    b[j] = *reinterpret_cast<const bf16x8*>(&Bs[(ss >> 1) * 2048 + c * 32 + ((((ss & 1) * 2 + fh) ^ ((c >> 2) & 3)) * 8)]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ss], b[j], acc[i][j], 0, 0, 0);
            }
            const int n0 = t * QN;
            if constexpr (STATS) {
                float* const sw = sc + ((st * 8 + wave) * 2) * 64;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { const float vr = bf2f(f2bf(acc[i][j][r])); s1 += vr; s2 = fmaf(vr, vr, s2); }
                    s1 += __shfl_xor(s1, 32);
                    s2 += __shfl_xor(s2, 32);
                    if (fh == 0) { sw[j * 32 + fr] = s1; sw[64 + j * 32 + fr] = s2; }
                }
            }
            if constexpr (TAIL != 0) {
                // per lane: a 16-byte segment (8 columns c8..c8+7) of row prow of each 8-row pass; everything else about an address is
                // uniform and travels as the buffer instructions' scalar offset
                float* const sg = reinterpret_cast<float*>(ob) + wave * (32 * SLD);
                const int prow = lane >> 3, c8 = (lane & 7) * 8;
                const unsigned v_row = ((unsigned)prow * (unsigned)p.ldc + (unsigned)c8) * 2u;                 // acc / dx
                const unsigned v_z = ((unsigned)prow * (unsigned)p.N + (unsigned)c8) * 2u;                     // bn_z (dense rows)
                const unsigned v_bit = (((unsigned)prow * (unsigned)p.N) >> 5) * 4u + (unsigned)(c8 >> 5) * 4u; // the 32-bit word of the lane's 8 bits
                const unsigned bshift = (unsigned)(c8 & 31);
                float pb[8], pg[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = pg[e] = 0.f;
                u32x4v eo[2], ez[2];
                unsigned eaw[2], ebw[2];
                auto fetch = [&](const int k) {                                            // pass k = (half k >> 2, rows (k & 3) * 8 + prow of it)
                    const unsigned mrow = (unsigned)(wr0 + (k >> 2) * 32 + (k & 3) * 8);   // uniform
                    if constexpr (T_ACC) {
                        eo[k & 1] = __builtin_amdgcn_raw_buffer_load_b128(rs_acc, v_row, (mrow * (unsigned)p.ldc + (unsigned)n0) * 2u, 0);
                        eaw[k & 1] = has_ab ? __builtin_amdgcn_raw_buffer_load_b32(rs_ab, v_bit, ((mrow * (unsigned)p.N + (unsigned)n0) >> 5) * 4u, 0) : 0xffffffffu;
                    }
                    if constexpr (T_BN) {
                        ez[k & 1] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, v_z, (mrow * (unsigned)p.N + (unsigned)n0) * 2u, 0);
                        ebw[k & 1] = __builtin_amdgcn_raw_buffer_load_b32(rs_zb, v_bit, ((mrow * (unsigned)p.N + (unsigned)n0) >> 5) * 4u, 0);
                    }
                };
                fetch(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((k & 3) == 0) {                                                    // the half's 32 rows into the staging area
                        const int i = k >> 2;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the previous half's reads are done
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sg[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLD + j * 32 + fr] = acc[i][j][r];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (k + 1 < 8) fetch(k + 1);
                    const int row = (k & 3) * 8 + prow;
                    const unsigned mrow = (unsigned)(wr0 + (k >> 2) * 32 + (k & 3) * 8);
                    f32x4v lo, hi;
                    {
                        const unsigned la = (unsigned)(unsigned long long)(lds_u32p)(sg + row * SLD + c8);
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(la) : "memory");
                    }
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                    if constexpr (T_ACC) {
                        const unsigned abyte = (eaw[k & 1] >> bshift) & 0xffu;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned w = eo[k & 1][e];
                            const float flo = __uint_as_float(w << 16), fhi = __uint_as_float(w & 0xffff0000u);
                            v[2 * e] += ((abyte >> (2 * e)) & 1u) ? flo : 0.f;
                            v[2 * e + 1] += ((abyte >> (2 * e + 1)) & 1u) ? fhi : 0.f;
                        }
                    }
                    u32x4v pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned l16 = f2bf(v[2 * e]), h16 = f2bf(v[2 * e + 1]);
                        pk[e] = l16 | (h16 << 16);
                        v[2 * e] = bf2f((unsigned short)l16); v[2 * e + 1] = bf2f((unsigned short)h16);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rs_c, v_row, (mrow * (unsigned)p.ldc + (unsigned)n0) * 2u, 0);
                    if constexpr (T_BN) {
                        const unsigned bbyte = (ebw[k & 1] >> bshift) & 0xffu;
                        // the lane's columns' mean / inverse deviation from the table every block loaded at its start
                        f32x4v mu0, mu1, is0, is1;
                        {
                            const unsigned lm = (unsigned)(unsigned long long)(lds_u32p)(vt + n0 + c8);
                            const unsigned li = (unsigned)(unsigned long long)(lds_u32p)(vt + p.N + n0 + c8);
                            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                         : "=&v"(mu0), "=&v"(mu1), "=&v"(is0), "=&v"(is1) : "v"(lm), "v"(li) : "memory");
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned w = ez[k & 1][e >> 1];
                            const float z = (e & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
                            const float mu = e < 4 ? mu0[e] : mu1[e - 4], is = e < 4 ? is0[e] : is1[e - 4];
                            const float dp = ((bbyte >> e) & 1u) ? v[e] : 0.f;
                            pb[e] += dp;
                            pg[e] = fmaf(dp, (z - mu) * is, pg[e]);
                        }
                    }
                }
                if constexpr (T_BN) {                                                      // the eight row-lanes of a column group -> the wave's 64-row sums
                    float* const sw = sc + ((st * 8 + wave) * 2) * 64;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float a = pb[e], b = pg[e];
                        a += __shfl_xor(a, 8);  b += __shfl_xor(b, 8);
                        a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                        if (prow == 0) { sw[c8 + e] = a; sw[64 + c8 + e] = b; }
                    }
                }
                pm0 = m0; pn0 = n0; have_prev = true;
                continue;
            }
            // ---- the wave's 64 x 64 results: bf16 into its private staging area, back as 16-byte row segments, out
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ow[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * 64 + j * 32 + fr] = f2bf(acc[i][j][r]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            {
                f32x4v v[NSTR];
#pragma unroll
                for (int u = 0; u < NSTR; ++u) {
                    const unsigned la = (unsigned)(unsigned long long)(lds_u32p)(ow + (u * 8 + (lane >> 3)) * 64 + (lane & 7) * 8);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(la));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < NSTR; ++u)
                    *reinterpret_cast<f32x4v*>(p.C + (size_t)(wr0 + u * 8 + (lane >> 3)) * p.ldc + n0 + (lane & 7) * 8) = v[u];
            }
            pm0 = m0; pn0 = n0; have_prev = true;
        }
    }
    if constexpr (SUMS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (have_prev) write_stats((it - 1) & 1, pm0, pn0);
    }
}
// -1 = rule, 0 = never, 1 = wherever legal
static int g_bf16_areg = -1;
extern "C" void uemdbg_conv_bf16_areg(int v) { g_bf16_areg = v; }
template <int KC, bool STATS, int TAIL>
static bool pw_areg_launch(const PwP& q, int npanels, int ngroups, hipStream_t st) {
    const int LDS = 2 * 64 * KC * 2 + (TAIL ? 8 * 32 * 68 * 4 : 8 * 4096 * 2) + 2 * 8 * 2 * 64 * 4 + ((TAIL & 2) ? 2 * q.N * 4 : 0);
    if (LDS > 160 * 1024) return false;
    const int grid = npanels * ngroups > 256 ? 256 : npanels * ngroups;
    auto k = pw_bf16_areg_kernel<KC, STATS, TAIL>;
    if (!uem_allow_lds((const void*)k, LDS)) return false;
    k<<<grid, 512, LDS, st>>>(q, npanels, ngroups);
    return true;
}
template <int KC>
static bool pw_areg_go(const PwP& q, int npanels, int ngroups, int tail, hipStream_t st) {
    switch (tail) {
        case 1: return pw_areg_launch<KC, false, 1>(q, npanels, ngroups, st);
        case 2: return pw_areg_launch<KC, false, 2>(q, npanels, ngroups, st);
        case 3: return pw_areg_launch<KC, false, 3>(q, npanels, ngroups, st);
        default: break;
    }
    return q.tile_stats != nullptr ? pw_areg_launch<KC, true, 0>(q, npanels, ngroups, st) : pw_areg_launch<KC, false, 0>(q, npanels, ngroups, st);
}
template <int MODE>
static bool conv_bf16_areg_try(const ConvP& p, unsigned xb, unsigned wb, hipStream_t st) {
    static const int env = getenv("UEM_CONV_BF16_AREG") ? atoi(getenv("UEM_CONV_BF16_AREG")) : -1;
    static const int env_tail = getenv("UEM_CONV_BF16_AREG_TAIL") ? atoi(getenv("UEM_CONV_BF16_AREG_TAIL")) : -1;
    const int set = g_bf16_areg >= 0 ? g_bf16_areg : env;
    const bool pointwise = p.KH * p.KW == 1 && p.pad == 0 && ((MODE == 1) ? (p.sub == 1 && p.stride == 1) : p.stride == 1);
    if (set == 0 || !pointwise || p.M % 512 != 0 || p.Cout % 64 != 0) return false;
    if (p.Cin != 64 && p.Cin != 128 && p.Cin != 256) return false;
    if (p.x_ld % 8 != 0 || p.y_ld % 8 != 0 || (((uintptr_t)p.x | (uintptr_t)p.y) & 15) != 0) return false;
    // the fused data-gradient tail: the identity gradient (through packed gate bits, or the tensor itself) and / or the BatchNorm-backward
    // partial sums with the mask from packed bits; everything else those epilogues can do stays with conv_bf16_kernel
    int tail = 0;
    if (MODE == 1) {
        if (p.accumulate) tail |= 1;
        if (p.tile_bnbwd != nullptr) {
            if (p.bn_bits == nullptr || p.bn_z == nullptr || p.bn_vec == nullptr || p.y_ld != p.Cout) return false;
            tail |= 2;
        }
        if (p.acc_bits != nullptr && p.y_ld != p.Cout) return false;
        // The tail forms are built, tested (tests/test_gpu_bf16.py, fixture "areg": fp64 references) and MEASURED SLOWER than
        // conv_bf16_kernel's epilogue on every shape (profiles/r06_m_*: +20 ... +64 %): a tail is bound by its three 4C-wide streams, and a
        // wave fetching one 8-row pass ahead keeps ~20 KB per CU in flight where the tiled kernel's block fetches a 64-row chunk ahead;
        // at 256 reduction channels the A registers (128) + accumulators (64) + the epilogue's state no longer fit 256 VGPRs (the
        // compiler moves 108 A registers to scratch and re-reads them per tile).  Never by rule: forced on only (set == 1).
        if (tail != 0 && (set != 1 || env_tail == 0)) return false;
        if (tail != 0 && (double)p.M * p.y_ld * 2.0 >= 4294967280.0) return false;           // the tail's streams use 32-bit buffer offsets
    } else if (p.accumulate) {
        return false;
    }
    const int npanels = p.M / 512;
    // Rule (scripts/sweep_conv_bf16_ring.py SWEEP=areg, profiles/r06_m_conv_bf16_areg_sweep_*.txt; forced on = wherever legal): a reduction
    // of 128 or 256 channels, at least twice as many output columns, and a panel for every CU.  There it is 10-30 % faster than the tiled
    // kernels (256 -> 1024 on 1024 x 1024 tiles: 0.144 -> 0.101 ms; 128 -> 512: -10 % / -27 % at 1024 / 512; plain data gradients of those
    // shapes -12 ... -34 %).  It loses with 64 channels (+24 %: the A registers buy nothing, the 64-column tiles cost), with no more
    // columns than channels (256 -> 64 / 128: +27 %) and with fewer panels than CUs (256 -> 1024 at 512 x 512: 64 panels, +140 %).
    if (set != 1 && !(set < 0 && npanels >= 256 && p.Cout >= 2 * p.Cin && p.Cin >= 128)) return false;
    // Fewer panels than CUs (forced on only): the panel's column tiles shared out over 2 / 4 / 8 blocks while each keeps at least four
    // tiles.  Measured: 256 -> 1024 at 512 x 512 (64 panels x 4 groups of 4 tiles) 0.032 -> 0.036 ms instead of 0.072 without the
    // groups -- a block that lives for four tiles spends as long loading its A rows and waiting for its first B tile: not by rule.
    int ngroups = 1;
    const int ntiles_n = p.Cout / 64;
    while (npanels * ngroups < 256 && ngroups < 8 && ntiles_n % (2 * ngroups) == 0 && ntiles_n / (2 * ngroups) >= 4) ngroups *= 2;
    PwP q;
    q.A = reinterpret_cast<const unsigned short*>(p.x); q.B = reinterpret_cast<const unsigned short*>(p.w);
    q.C = reinterpret_cast<unsigned short*>(p.y);
    q.tile_stats = MODE == 0 ? p.tile_stats : ((tail & 2) ? p.tile_bnbwd : nullptr);
    q.M = p.M; q.N = p.Cout; q.K = p.Cin; q.lda = p.x_ld; q.ldc = p.y_ld; q.a_bytes = xb; q.b_bytes = wb;
    q.acc = (tail & 1) ? (p.acc_src != nullptr ? reinterpret_cast<const unsigned short*>(p.acc_src) : q.C) : nullptr;
    q.acc_bits = (tail & 1) ? p.acc_bits : nullptr;
    q.bn_z = reinterpret_cast<const unsigned short*>(p.bn_z); q.bn_vec = p.bn_vec; q.bn_bits = p.bn_bits;
    switch (p.Cin) {
        case 64: return pw_areg_go<64>(q, npanels, ngroups, tail, st);
        case 128: return pw_areg_go<128>(q, npanels, ngroups, tail, st);
        default: return pw_areg_go<256>(q, npanels, ngroups, tail, st);
    }
}
// Ring of three operand stages on persistent 256-row blocks (conv_bf16_kernel<..., NST = 3>): full dense tiles, forward (plain / with the
// BatchNorm tile statistics) and data gradient (plain / residual tail / BatchNorm-backward partial sums).  -1 = rule, 0 = never,
// 1 = wherever legal (tests, sweeps).
static int g_bf16_ring = -1;
extern "C" void uemdbg_conv_bf16_ring(int v) { g_bf16_ring = v; }
template <int MODE>
static int conv_bf16_ring_bn(const ConvP& p) {
    static const int env = getenv("UEM_CONV_BF16_RING") ? atoi(getenv("UEM_CONV_BF16_RING")) : -1;
    const int set = g_bf16_ring >= 0 ? g_bf16_ring : env;
    if (set == 0 || p.M % 256 != 0 || (MODE == 1 && p.sub != 1) || p.ntaps <= 0) return 0;
    if (MODE == 0 && p.accumulate) return 0;
    const int bn = p.Cout % 128 == 0 ? 128 : 64;
    if (set == 1) return bn;
    if (set == 2) return (bn != 128 || (MODE == 1 && (p.accumulate || p.acc_src || p.tile_bnbwd))) ? 0 : 1128;   // split roles (128 columns, no epilogue loads)
    // Rule, from the per-shape sweeps against the round-5 dispatch (scripts/sweep_conv_bf16_ring.py; profiles/r06_a ... r06_f):
    //  * the ring with the classic epilogue ties or loses (forward +3 ... +30 %, fused data gradients +15 ... +25 %): never by rule;
    //  * with the roles split (loader / storer waves) it wins -5 ... -18 % where a block walks many short tiles -- at least 16 tiles of
    //    256 x 128 per CU and 2 ... 18 k-steps per tile (the pointwise layers of the 1024 x 1024 configuration) -- and loses on the long
    //    reductions (3x3 at 256 / 512 channels: +7 ... +15 %), on single-k-step tiles (64 -> 256: +14 %) and wherever a CU gets fewer
    //    than 16 tiles (every shape of the 512 x 512 configuration: +4 ... +25 %).  No epilogue loads (forward, plain data gradient).
    //    On 128-row two-stage blocks (two per CU) the split loses everywhere (+4 ... +30 %): not instantiated.
    if (bn != 128 || (MODE == 1 && (p.accumulate || p.acc_src || p.tile_bnbwd))) return 0;
    const int KT = p.ntaps * (p.Cin / KBH);
    const int ntiles = (p.M / 256) * (p.Cout / 128);
    static const int min_tiles = getenv("UEM_CONV_BF16_SPLIT_MIN_TILES") ? atoi(getenv("UEM_CONV_BF16_SPLIT_MIN_TILES")) : 4096;
    return (ntiles >= min_tiles && KT >= 2 && KT <= 18) ? 1128 : 0;
}
template <int BN_, int MODE>
static void conv_bf16_ring_go(const ConvP& p, unsigned xb, unsigned wb, hipStream_t st) {
    const bool extras = MODE == 0 ? p.tile_stats != nullptr : (p.accumulate || p.acc_src || p.tile_bnbwd);
    if (extras) conv_bf16_launch<BN_, MODE, 1, true, 256, 3>(p, xb, wb, st);
    else conv_bf16_launch<BN_, MODE, 0, true, 256, 3>(p, xb, wb, st);
}
template <int BN_, int MODE, int BMT = 256, int NST = 3>
static void conv_bf16_split_go(const ConvP& p, unsigned xb, unsigned wb, hipStream_t st) {
    if constexpr (MODE == 0) {
        if (p.tile_stats != nullptr) { conv_bf16_launch<BN_, 0, 1, true, BMT, NST, true>(p, xb, wb, st); return; }
    }
    conv_bf16_launch<BN_, MODE, 0, true, BMT, NST, true>(p, xb, wb, st);
}
static int g_bf16_lazy = -1;     // tuning override: counted waits around the persistent blocks' epilogue on / off, -1 = environment / default
extern "C" void uemdbg_conv_bf16_lazy(int v) { g_bf16_lazy = v; }
static int conv_bf16_lazy() {
    static const int env = getenv("UEM_CONV_BF16_LAZY") ? atoi(getenv("UEM_CONV_BF16_LAZY")) : 1;
    return g_bf16_lazy >= 0 ? g_bf16_lazy : env;
}
static int g_bf16_big = -1;      // tuning override: 0 = never the 256-row tiles, 1 = wherever they are legal, -1 = rule
extern "C" void uemdbg_conv_bf16_big(int v) { g_bf16_big = v; }
template <int BN_, int MODE, int BMT = 128>
static void conv_bf16_go(const ConvP& p, unsigned xb, unsigned wb, hipStream_t st) {
    const bool full = p.M % BMT == 0 && (MODE == 0 || p.sub == 1) && UEM_DBG(p.dbg) == 0;
    const bool extras = MODE == 0 ? p.tile_stats != nullptr : (p.accumulate || p.acc_src || p.tile_bnbwd);
    static const int penv = getenv("UEM_CONV_BF16_PERSIST") ? atoi(getenv("UEM_CONV_BF16_PERSIST")) : -1;
    const int pset = g_bf16_persist >= 0 ? g_bf16_persist : penv;
    const int ntiles = (int)uem_cdiv(p.M, BMT) * (p.Cout / BN_);
    // persistent blocks: full tiles, at least two tiles per block
    // Rule (profiles/r04_q_conv_bf16_persist.txt): the short k-loops (at most 12 k-steps: the pointwise layers up to K = 768 and the
    // 64-channel 3x3) with at least two rounds of tiles, except the fused data-gradient epilogues -- their prefetched operands (the
    // tensor accumulated into, z, the packed bits) are older than the operand DMA of the next tile, retire before it, and the block
    // loses what persistence hides (residual tails +15 ... 50 %).  pset: 1 = every full-tile launch, 2 = all but those epilogues.
    constexpr int per_cu = BMT == 256 ? 1 : (160 * 1024 / ConvBf16Cfg<BN_, true, BMT>::LDS_BYTES > 3 ? 3 : 160 * 1024 / ConvBf16Cfg<BN_, true, BMT>::LDS_BYTES);
    constexpr int slots = 256 * per_cu;
    const int KT = p.ntaps * (p.Cin / KBH);
    const bool persist = full && !(MODE == 0 && p.accumulate) && p.ntaps > 0 &&
                         (pset == 1 || (pset == 2 && !(MODE == 1 && extras)) ||
                          (pset < 0 && !(MODE == 1 && extras) && KT <= 12 && ntiles >= 2 * slots));
    if constexpr (BMT == 256) {
        // the 256-row tiles are dispatched for full tiles only (conv_bf16_big_bn); 256 x 256 keeps 128 accumulator registers per lane:
        // no room for the persistent form's cross-tile state or the fused data-gradient epilogues' prefetch (they spill) -> never
        // persistent, and the fused data-gradient epilogues stay at BN = 128
        if constexpr (BN_ == 256) {
            if (extras) { if constexpr (MODE == 0) conv_bf16_launch<BN_, MODE, 1, false, BMT>(p, xb, wb, st); }
            else conv_bf16_launch<BN_, MODE, 0, false, BMT>(p, xb, wb, st);
        } else {
            if (extras) { if (persist) conv_bf16_launch<BN_, MODE, 1, true, BMT>(p, xb, wb, st); else conv_bf16_launch<BN_, MODE, 1, false, BMT>(p, xb, wb, st); }
            else { if (persist) conv_bf16_launch<BN_, MODE, 0, true, BMT>(p, xb, wb, st); else conv_bf16_launch<BN_, MODE, 0, false, BMT>(p, xb, wb, st); }
        }
    } else {
        if (!full || (MODE == 0 && p.accumulate)) conv_bf16_launch<BN_, MODE, -1, false, BMT>(p, xb, wb, st);
        else if (extras) { if (persist) conv_bf16_launch<BN_, MODE, 1, true, BMT>(p, xb, wb, st); else conv_bf16_launch<BN_, MODE, 1, false, BMT>(p, xb, wb, st); }
        else { if (persist) conv_bf16_launch<BN_, MODE, 0, true, BMT>(p, xb, wb, st); else conv_bf16_launch<BN_, MODE, 0, false, BMT>(p, xb, wb, st); }
    }
}
// 256-row tiles (eight waves, one block per CU): when do they pay?  Measured per shape at B = 32 (scripts/sweep_conv_bf16_big.py,
// profiles/r05_*_conv_bf16_big_sweep.txt): the MFMA-heavy launches gain -- layer4's 3x3 convs -9 ... -11 %, its 1024 -> 2048
// downsample -5 ... -8 %, the other layer4 pointwise layers 0 ... -6 % -- and everything with a short reduction loses (one block per
// CU has nobody to overlap a tile's prologue and epilogue with: layer3's 256 -> 1024 +4 ... +52 %, its 3x3 at 256 tiles +46 %).  Rule:
// full tiles, at least 16 k-steps of 64 channels (K >= 1024), at least one tile per CU; 256 columns where that still holds, else 128;
// the fused data-gradient epilogues at 128 columns only (their prefetch registers beside 128 accumulators spill).  Returns 0 (the
// 128-row kernel), 128 or 256.  UEM_CONV_BF16_BIG: 0 never, 1 wherever legal (tests), default the rule.
template <int MODE>
static int conv_bf16_big_bn(const ConvP& p) {
    static const int env = getenv("UEM_CONV_BF16_BIG") ? atoi(getenv("UEM_CONV_BF16_BIG")) : -1;
    const int set = g_bf16_big >= 0 ? g_bf16_big : env;
    if (set == 0 || p.M % 256 != 0 || (MODE == 1 && p.sub != 1) || p.Cout % 128 != 0 || UEM_DBG(p.dbg) != 0) return 0;
    if (MODE == 0 && p.accumulate) return 0;
    const int rows = p.M / 256, KT = p.ntaps * (p.Cin / KBH);
    static const int min_tiles = getenv("UEM_CONV_BF16_BIG_MIN_TILES") ? atoi(getenv("UEM_CONV_BF16_BIG_MIN_TILES")) : 256;
    static const int min_kt = getenv("UEM_CONV_BF16_BIG_MIN_KT") ? atoi(getenv("UEM_CONV_BF16_BIG_MIN_KT")) : 16;
    const bool extras = MODE == 1 && (p.accumulate || p.acc_src || p.tile_bnbwd);
    if (set == 1) return (!extras && p.Cout % 256 == 0) ? 256 : 128;
    // the fused data-gradient epilogues (residual tail, BatchNorm-backward partial sums) LOSE on 256-row tiles: with one block per CU
    // nothing covers their three extra streams (R101 1024^2 step: forward family 46.9 -> 44.9 ms, data gradient 54.4 -> 56.4)
    // Round 6: 256 x 256 tiles ALSO on short reductions when the output is wide and the grid is deep (the 1024 x 1024 configuration: never
    // swept in round 5).  What bounds the short-k pointwise layers is not HBM and not the block's pipeline -- five structures of the block
    // land on the same time -- but the operand bytes a CU pulls from L2 per flop (~70 GB/s per CU from L2,
    // MI355X_MICROARCH.md "Indexed rows"): a 256 x 256 tile fetches 2/3 of what 256 x 128 fetches.  Measured at 1024 x 1024
    // (profiles/r06_j_conv_bf16_big_sweep_1024.txt): 128 -> 512 -13 %, 256 -> 1024 -10 %, 512 -> 2048 -18 % forward; the plain data gradients
    // with 512 ... 2048 output columns -11 ... -17 %; 512 output columns behind a reduction of 1024 and more: +2 ... +4 % (not taken).
    static const int wide_min_tiles = getenv("UEM_CONV_BF16_BIG_WIDE_MIN_TILES") ? atoi(getenv("UEM_CONV_BF16_BIG_WIDE_MIN_TILES")) : 1024;
    if (!extras && KT < min_kt && p.Cout % 256 == 0 && (p.Cout >= 1024 || KT <= 4) && p.Cout >= 512 && KT >= 2 &&
        rows * (p.Cout / 256) >= wide_min_tiles)
        return 256;
    if (KT < min_kt || extras) return 0;
    if (!extras && p.Cout % 256 == 0 && rows * (p.Cout / 256) >= min_tiles) return 256;
    return rows * (p.Cout / 128) >= min_tiles ? 128 : 0;
}
struct BnBwdFuseH { const uint16_t* z; const float* vec; float* tiles; const uint16_t* acc_src; const uint32_t* acc_bits; const uint32_t* bn_bits; };
static int conv2d_bf16_impl(const uint16_t* x, const uint16_t* w, uint16_t* y, const uem_conv_shape* s, int flags, float* tile_stats,
                            const BnBwdFuseH* fuse, void* stream);
extern "C" int uem_conv2d_bf16(const uint16_t* x, const uint16_t* w, uint16_t* y, const uem_conv_shape* s, int flags,
                               float* tile_stats, void* stream) {
    return conv2d_bf16_impl(x, w, y, s, flags, tile_stats, nullptr, stream);
}
// bf16 twin of uem_conv2d_dgrad_bnbwd / uem_conv2d_dgrad_tail (same argument meaning; tensors are bf16, vectors and partial
// sums fp32; the partial sums are taken over the ROUNDED dx, the values the apply pass reads back)
extern "C" int uem_conv2d_dgrad_tail_bf16(const uint16_t* dy, const uint16_t* w_t, uint16_t* dx, const uem_conv_shape* s,
                                          const uint16_t* acc_src, const uint32_t* acc_bits, const uint16_t* bn_z, const float* bn_vec,
                                          const uint32_t* bn_bits, float* tile_partials, int flags, void* stream) {
    UEM_REQUIRE(s && dx, "conv2d_dgrad_tail_bf16: null pointer");
    UEM_REQUIRE((acc_bits == nullptr) == (acc_src == nullptr), "conv2d_dgrad_tail_bf16: acc_src and acc_bits go together");
    UEM_REQUIRE((bn_z == nullptr) == (tile_partials == nullptr) && (bn_z == nullptr) == (bn_vec == nullptr) && (bn_z || !bn_bits),
                "conv2d_dgrad_tail_bf16: bn_z, bn_vec and tile_partials go together");
    if (s->stride != 1 || ((int64_t)s->N * s->H * s->W) % 128 != 0 || s->Cin % 64 != 0 || s->x_ld != s->Cin)
        return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_dgrad_tail_bf16: needs stride 1, M %% 128 == 0, Cin %% 64 == 0, dense rows");
    UEM_REQUIRE((flags & ~UEM_CONV_ACCUMULATE) == 0, "conv2d_dgrad_tail_bf16: bad flags");
    UEM_REQUIRE(!(acc_src && (flags & UEM_CONV_ACCUMULATE)), "conv2d_dgrad_tail_bf16: acc_src already names the tensor accumulated into");
    BnBwdFuseH f{bn_z, bn_vec, tile_partials, acc_src, acc_bits, bn_bits};
    return conv2d_bf16_impl(dy, w_t, dx, s, UEM_CONV_TRANSPOSED | flags | (acc_src ? UEM_CONV_ACCUMULATE : 0), nullptr, &f, stream);
}
static int conv2d_bf16_impl(const uint16_t* x, const uint16_t* w, uint16_t* y, const uem_conv_shape* s, int flags, float* tile_stats,
                            const BnBwdFuseH* fuse, void* stream) {
    UEM_REQUIRE(x && w && y, "conv2d_bf16: null pointer");
    int rc = conv_check(s);
    if (rc) return rc;
    const bool transposed = (flags & UEM_CONV_TRANSPOSED) != 0;
    UEM_REQUIRE((flags & ~(UEM_CONV_TRANSPOSED | UEM_CONV_ACCUMULATE)) == 0, "conv2d_bf16: only TRANSPOSED / ACCUMULATE flags");
    const int kin = transposed ? s->Cout : s->Cin, kout = transposed ? s->Cin : s->Cout;
    if (kin % KBH != 0 || kout % 64 != 0 || s->x_ld % 8 != 0 || s->y_ld % 8 != 0 || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15))
        return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_bf16: needs reduction channels %% 64 == 0, output channels %% 64 == 0, 16-byte rows");
    ConvP p;
    p.x = (const float*)x; p.w = (const float*)w; p.bias = nullptr; p.in_scale = p.in_shift = nullptr; p.y = (float*)y;
    p.KH = s->KH; p.KW = s->KW; p.stride = s->stride; p.pad = s->pad; p.dil = s->dil;
    p.accumulate = (flags & UEM_CONV_ACCUMULATE) ? 1 : 0; p.relu = 0;
    p.sub = 1; p.py = p.px = 0; p.Hs = p.Ws = 0; p.ntaps = s->KH * s->KW; p.tapmask = 0; p.dbg = g_conv_dbg; p.y_bf16 = 0; p.wg_rows = 0; p.wg_stride = 0; p.lazy = conv_bf16_lazy();
    p.tile_stats = tile_stats; p.bn_z = nullptr; p.bn_vec = nullptr; p.tile_bnbwd = nullptr; p.acc_src = nullptr; p.acc_bits = nullptr; p.bn_bits = nullptr;
    if (fuse != nullptr) {
        p.bn_z = (const float*)fuse->z; p.bn_vec = fuse->vec; p.tile_bnbwd = fuse->tiles;
        p.acc_src = (const float*)fuse->acc_src; p.acc_bits = fuse->acc_bits; p.bn_bits = fuse->bn_bits;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!transposed) {
        UEM_REQUIRE(!tile_stats || ((int64_t)s->N * s->Ho * s->Wo) % 128 == 0, "conv2d_bf16: tile statistics need M %% 128 == 0");
        p.N = s->N; p.H = s->H; p.W = s->W; p.Cin = s->Cin; p.Ho = s->Ho; p.Wo = s->Wo; p.Cout = s->Cout;
        p.x_ld = s->x_ld; p.y_ld = s->y_ld; p.M = s->N * s->Ho * s->Wo;
        const double xb = (double)p.N * p.H * p.W * p.x_ld * 2.0, wb = (double)p.Cout * p.KH * p.KW * p.Cin * 2.0;
        if (xb >= 4294967280.0 || wb >= 4294967280.0) return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_bf16: tensor beyond 32-bit buffer offsets");
        if (conv_bf16_areg_try<0>(p, (unsigned)xb, (unsigned)wb, st)) return uem_check_launch("conv2d_bf16 (A-stationary pointwise)");
        if (conv_bf16_pw_try<0>(p, (unsigned)xb, (unsigned)wb, st)) return uem_check_launch("conv2d_bf16 (pointwise stream)");
        const int ring = conv_bf16_ring_bn<0>(p);
        const int big = ring ? 0 : conv_bf16_big_bn<0>(p);
        if (ring == 1128) conv_bf16_split_go<128, 0>(p, (unsigned)xb, (unsigned)wb, st);
        else if (ring == 128) conv_bf16_ring_go<128, 0>(p, (unsigned)xb, (unsigned)wb, st);
        else if (ring == 64) conv_bf16_ring_go<64, 0>(p, (unsigned)xb, (unsigned)wb, st);
        else if (big == 256) conv_bf16_go<256, 0, 256>(p, (unsigned)xb, (unsigned)wb, st);
        else if (big == 128) conv_bf16_go<128, 0, 256>(p, (unsigned)xb, (unsigned)wb, st);
        else if (p.Cout % 128 == 0) conv_bf16_go<128, 0>(p, (unsigned)xb, (unsigned)wb, st);
        else conv_bf16_go<64, 0>(p, (unsigned)xb, (unsigned)wb, st);
        return uem_check_launch("conv2d_bf16");
    }
    UEM_REQUIRE(!tile_stats, "conv2d_bf16: no statistics on the data gradient");
    UEM_REQUIRE(s->KH * s->KW <= 16, "conv2d_bf16 dgrad: at most 16 taps");
    p.N = s->N; p.H = s->Ho; p.W = s->Wo; p.Cin = s->Cout;
    p.Ho = s->H; p.Wo = s->W; p.Cout = s->Cin;
    p.x_ld = s->y_ld; p.y_ld = s->x_ld;
    p.sub = s->stride;
    const double xb = (double)p.N * p.H * p.W * p.x_ld * 2.0, wb = (double)p.Cout * p.KH * p.KW * p.Cin * 2.0;
    if (xb >= 4294967280.0 || wb >= 4294967280.0) return uem_fail(UEM_ERR_UNSUPPORTED, "conv2d_bf16: tensor beyond 32-bit buffer offsets");
    for (int py = 0; py < s->stride; ++py) {
        for (int px = 0; px < s->stride; ++px) {
            p.py = py; p.px = px;
            p.Hs = (s->H - py + s->stride - 1) / s->stride;
            p.Ws = (s->W - px + s->stride - 1) / s->stride;
            if (p.Hs <= 0 || p.Ws <= 0) continue;
            p.ntaps = 0; p.tapmask = 0;
            for (int ky = 0; ky < s->KH; ++ky) {
                if ((py + s->pad - ky * s->dil) % s->stride != 0) continue;
                for (int kx = 0; kx < s->KW; ++kx) {
                    if ((px + s->pad - kx * s->dil) % s->stride != 0) continue;
                    p.tapmask |= (unsigned long long)(ky * s->KW + kx) << (4 * p.ntaps);
                    ++p.ntaps;
                }
            }
            if (p.ntaps == 0 && p.accumulate) continue;
            p.M = s->N * p.Hs * p.Ws;
            // the residual tails on 64-wide tiles, as in the fp32 kernel (conv_dma_try): dgrad family 10.71 -> 10.38 ms per step; the
            // forward's wide, small-K layers lose on them (8.48 -> 8.87 ms)
            // (round 4: with the packed-bit words no longer serialising the epilogue's loads the 64-wide tiles stopped paying --
            // layer3 / layer4 tails 10-13 % faster on 128-wide tiles, layer1 / layer2 equal: off unless UEM_BF16_WIDE_TAIL=1)
            static const int wt_env = getenv("UEM_BF16_WIDE_TAIL") ? atoi(getenv("UEM_BF16_WIDE_TAIL")) : 0;
            const bool wide_tail = wt_env != 0 && (p.accumulate != 0 || p.tile_bnbwd != nullptr) && 2 * p.ntaps * p.Cin <= p.Cout &&
                                   (wt_env != 2 || p.Cin <= 256);
            if (!wide_tail && conv_bf16_areg_try<1>(p, (unsigned)xb, (unsigned)wb, st)) continue;
            if (!wide_tail && conv_bf16_pw_try<1>(p, (unsigned)xb, (unsigned)wb, st)) continue;
            const int ring = wide_tail ? 0 : conv_bf16_ring_bn<1>(p);
            const int big = (wide_tail || ring) ? 0 : conv_bf16_big_bn<1>(p);
            if (ring == 1128) conv_bf16_split_go<128, 1>(p, (unsigned)xb, (unsigned)wb, st);
            else if (ring == 128) conv_bf16_ring_go<128, 1>(p, (unsigned)xb, (unsigned)wb, st);
            else if (ring == 64) conv_bf16_ring_go<64, 1>(p, (unsigned)xb, (unsigned)wb, st);
            else if (big == 256) conv_bf16_go<256, 1, 256>(p, (unsigned)xb, (unsigned)wb, st);
            else if (big == 128) conv_bf16_go<128, 1, 256>(p, (unsigned)xb, (unsigned)wb, st);
            else if (p.Cout % 128 == 0 && !wide_tail) conv_bf16_go<128, 1>(p, (unsigned)xb, (unsigned)wb, st);
            else conv_bf16_go<64, 1>(p, (unsigned)xb, (unsigned)wb, st);
        }
    }
    return uem_check_launch("conv2d_bf16 (dgrad)");
}
