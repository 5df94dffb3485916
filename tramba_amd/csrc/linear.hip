// 1x1 convolution == row-major GEMM in channels-last:  Y[M,N] = epilogue(X[M,K] W[N,K]^T).
//
// Replaces Linear2d.forward (Models/modules.py:10-13: F.conv2d with a (out,in,1,1) weight) and
// the grouped x_proj conv1d (vmamba.py:233-234, evaluated once in spatial order, see
// ss2d_fused.hip).  Both operands are K-contiguous, which is exactly the MFMA A/B fragment
// order on gfx950: lane l of v_mfma_f32_32x32x16_{bf16,f16} holds A[row l&31][k = 8(l>>5)+j]
// and B[k = 8(l>>5)+j][col l&31], j = 0..7 -- one 16-byte load per fragment, no transpose.
// fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
//
// v1 structure: 256 threads = 2x2 waves, each wave a 64x64 output tile (2x2 MFMA tiles, 64
// accumulator registers), fragments loaded straight from global/L2 (the activations of this
// network are short-K: K = 128..4096, most GEMMs are HBM/L2-bound on the M x N output, so the
// fused epilogue matters more than LDS staging; an LDS-staged variant is the next step).
// Epilogue fused: + bias, activation, + residual, cast.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace tramba {

typedef __attribute__((ext_vector_type(8))) short frag8_t;     // 8 x 16-bit
typedef __attribute__((ext_vector_type(8))) _Float16 frag8h_t; // 8 x f16
typedef __attribute__((ext_vector_type(16))) float acc16_t;

template <typename T> struct Mfma;
template <> struct Mfma<__hip_bfloat16> {
    static constexpr int KSTEP = 16;
    using frag = frag8_t;
    static __device__ __forceinline__ acc16_t run(frag a, frag b, acc16_t c)
    {
        typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<__half> {
    static constexpr int KSTEP = 16;
    using frag = frag8_t;
    static __device__ __forceinline__ acc16_t run(frag a, frag b, acc16_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(frag8h_t, a), __builtin_bit_cast(frag8h_t, b), c, 0, 0, 0);
    }
};

// 16-bit operands: fragment = 8 consecutive k of one row
template <typename T>
__device__ __forceinline__ frag8_t load_frag16(const T *__restrict__ base, long row, long nrows, int k0, int K,
                                               bool vec)
{
    frag8_t f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nrows) {
        const T *p = base + row * K + k0;
        if (vec && k0 + 8 <= K) {
            f = *reinterpret_cast<const frag8_t *>(p);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < K) f[j] = *reinterpret_cast<const short *>(p + j);
        }
    }
    return f;
}

template <typename T, typename TO>
__device__ __forceinline__ void store_tile(const acc16_t &acc, TO *__restrict__ y, const float *__restrict__ bias,
                                           const T *__restrict__ res, long m0, int n0, long M, int N, int act,
                                           int lane)
{
    const int col = n0 + (lane & 31);
    if (col >= N) return;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) {
            float v = apply_act(acc[r] + bv, act);
            if (res) v = combine_residual(v, Cvt<T>::to_f(res[row * N + col]), act);
            y[row * N + col] = Cvt<TO>::from_f(v);
        }
    }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void linear16_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                      const float *__restrict__ bias, const T *__restrict__ res,
                                                      TO *__restrict__ y, long M, int N, int K, int act, int vec)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.y * 128 + (wave >> 1) * 64;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (m0 >= M || n0 >= N) return;
    acc16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r32 = lane & 31, kh = (lane >> 5) * 8;
    for (int k0 = 0; k0 < K; k0 += 16) {
        frag8_t a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = load_frag16<T>(x, m0 + i * 32 + r32, M, k0 + kh, K, vec);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = load_frag16<T>(w, n0 + j * 32 + r32, N, k0 + kh, K, vec);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Mfma<T>::run(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<T, TO>(acc[i][j], y, bias, res, m0 + i * 32, n0 + j * 32, M, N, act, lane);
}


// ------------------------------------------------------------------------------------------------
// v2: LDS-staged, double-buffered tile kernel for 16-bit operands.
//   block = 256 threads = 2x2 waves; block tile BM x BN, K step 64; wave tile (BM/2) x (BN/2) made of
//   32x32 MFMA tiles.  Global -> registers -> LDS (16-byte chunks, XOR-swizzled by row so the
//   ds_read_b128 fragment reads of 128-byte rows are conflict-free), tile k+1 is in flight while tile
//   k is multiplied; one barrier per K step.  The accumulators leave through LDS so that bias,
//   activation, residual and the store all run on whole 128-byte row segments (the MFMA C layout
//   has one COLUMN per lane, which would store 2-byte scalars at a row stride).
constexpr int kBK = 64;

// 16-byte chunk swizzle of a 128-byte LDS row.  ds_read_b128 is served in 16-lane groups
// {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32): rows of equal parity in a group must land on
// different 16-byte slots -> key on (row >> 1), not on row.
__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Implicit-GEMM geometry of a 3x3 / stride-2 / pad-1 convolution on a channels-last map: row m of the
// GEMM is output pixel (b, ho, wo), column k = (ky*3 + kx)*Cin + ci.  Cin % 64 == 0, so every 64-deep
// K step lies inside ONE tap: the A tile of a step is 64 contiguous channels of one (shifted) input
// pixel per row -- the same 16-byte chunk loads as the plain GEMM, with a per-row base address.
struct ConvGeom {
    int hin, win, cin, hout, wout;
};

// The tile kernels issue their MFMAs with the operands SWAPPED (weight fragment first): the 32x32 accumulator of a
// wave then holds, per lane, ONE output row m (= lane & 31) and 16 output columns in four groups of 4 consecutive ones
// (n = 8g + 4(lane >> 5) + 0..3).  Dropping it into the fp32 staging tile is four 16-byte LDS writes per lane instead of
// sixteen 4-byte ones, and with a row stride of BN + 4 floats the 16 lanes of a write pass hit 16 different bank groups
// (the plain orientation wrote 2-way conflicting 4-byte columns: a quarter of the tall GEMMs' LDS time).
// Wave layouts of a 256-thread block: NWM = 2 -> 2 x 2 waves (each BM/2 x BN/2); NWM = 3 -> 3 x 1 compute waves (each
// BM/3 x BN), the fourth wave only stages operands and streams the epilogue (the 96x64 tile: M = 2304 is 24 x 96, so the
// 15-block stage's 512-wide GEMMs make 192 workgroups = ONE round on 256 CUs, where 64x64 tiles make 288 = two).
template <int BM, int BN, int NWM>
struct WaveLayout {
    static constexpr int NWN = NWM == 2 ? 2 : 1;
    static constexpr int WM = BM / NWM, WN = BN / NWN, TM = WM / 32, TN = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tiles are whole 32x32 MFMA tiles");
};

template <int BM, int BN, int NWM = 2, typename ACC>
__device__ __forceinline__ void acc_to_lds(ACC &acc, float *ep)   // ACC = acc16_t[TM][TN] of the layout
{
    using WL = WaveLayout<BM, BN, NWM>;
    constexpr int TM = WL::TM, TN = WL::TN, WM = WL::WM, WN = WL::WN, LDE = BN + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= NWM * WL::NWN) return;       // loader waves (the 4th of the 3 x 1 layout, waves 4-7 of a 512-thread block) hold no accumulators
    const int wm = wave / WL::NWN, wn = wave % WL::NWN;
    const int r32 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float *dst = ep + (wm * WM + i * 32 + r32) * LDE + wn * WN + j * 32 + 8 * g + 4 * hi;
                *reinterpret_cast<float4 *>(dst) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            }
}

// Epilogue through LDS, block-wide (shared by the tile kernels): see the comment inside.
template <typename T, typename TO, int BM, int BN, int NWM = 2, int NT = 256, typename ACC>
__device__ __forceinline__ void tile_epilogue(ACC &acc, unsigned char *lds,
                                              const float *__restrict__ bias, const T *__restrict__ res,
                                              TO *__restrict__ y, long M, int N, int act, long m0, int n0,
                                              const float *__restrict__ colsum = nullptr,
                                              const float2 *__restrict__ rowstat = nullptr,
                                              TO *__restrict__ y_pre = nullptr)
{
    // y_pre (training, linear_dma_kernel<.., DUAL>): the value BEFORE the activation (acc + bias) also leaves, to a second
    // tensor -- the pre-activation the backward of a GELU needs, without a separate activation launch re-reading it.
    // colsum / rowstat: LayerNorm of the INPUT rows folded into the GEMM (linear_lean_kernel<.., LNIN>): the accumulator
    // holds x . W' with W' = W * gamma; the normalised product is rstd_m * (acc - mean_m * colsum_n), and beta . W rides
    // in `bias`.  rowstat[row] = (mean, rstd) of the block's rows, in LDS behind the staging tile.
    const int tid = threadIdx.x;
    // ---- epilogue through LDS, block-wide: the four waves drop their accumulators into one BM x BN fp32
    // tile, then all 256 threads stream it out as 16-byte stores covering whole rows (a wave-private
    // 32-column epilogue wrote 64-byte half lines with 8-byte stores: ~1 TB/s on the output-bound layers)
    constexpr int LDE = BN + 4;
    float *ep = reinterpret_cast<float *>(lds);
    acc_to_lds<BM, BN, NWM>(acc, ep);
    __syncthreads();
    constexpr int CPL = 8;                 // columns per lane: 16 bytes of a 16-bit output row
    constexpr int LPRW = BN / CPL;         // lanes per row
    constexpr int RPI = NT / LPRW;         // rows per pass of the block (NT threads: 256, or 512 with loader waves)
    const int cl = (tid % LPRW) * CPL, rl = tid / LPRW;
    const int gcol = n0 + cl;
#pragma unroll
    for (int it = 0; it < (BM + RPI - 1) / RPI; ++it) {
        const int row = it * RPI + rl;
        const long grow = m0 + row;
        if (grow >= M || gcol >= N) continue;
        float o[CPL];
        {
            const float4 v0 = *reinterpret_cast<const float4 *>(ep + row * LDE + cl);
            const float4 v1 = *reinterpret_cast<const float4 *>(ep + row * LDE + cl + 4);
            o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
        }
        if (gcol + CPL <= N && (N & 7) == 0) {
            if (colsum) {
                const float2 st = rowstat[row];
                float cs[CPL];
                load_pack<float, CPL>(colsum + gcol, cs);
#pragma unroll
                for (int q = 0; q < CPL; ++q) o[q] = st.y * fmaf(-st.x, cs[q], o[q]);
            }
            if (bias) {
                float bv[CPL];
                load_pack<float, CPL>(bias + gcol, bv);
#pragma unroll
                for (int q = 0; q < CPL; ++q) o[q] += bv[q];
            }
            if (y_pre) store_pack<TO, CPL>(y_pre + grow * N + gcol, o);
#pragma unroll
            for (int q = 0; q < CPL; ++q) o[q] = apply_act(o[q], act);
            if (res) {
                float rv[CPL];
                load_pack<T, CPL>(res + grow * N + gcol, rv);
#pragma unroll
                for (int q = 0; q < CPL; ++q) o[q] = combine_residual(o[q], rv[q], act);
            }
            store_pack<TO, CPL>(y + grow * N + gcol, o);
        } else {
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                if (gcol + q < N) {
                    if (colsum) o[q] = rowstat[row].y * fmaf(-rowstat[row].x, colsum[gcol + q], o[q]);
                    const float pre = o[q] + (bias ? bias[gcol + q] : 0.f);
                    if (y_pre) y_pre[grow * N + gcol + q] = Cvt<TO>::from_f(pre);
                    float t = apply_act(pre, act);
                    if (res) {
                        const float rv = Cvt<T>::to_f(res[grow * N + gcol + q]);
                        t = combine_residual(t, rv, act);
                    }
                    y[grow * N + gcol + q] = Cvt<TO>::from_f(t);
                }
            }
        }
    }
}

template <typename T, typename TO, int BM, int BN, int PF, bool CONV>
__global__ __launch_bounds__(256) void linear_tiled_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                          const float *__restrict__ bias,
                                                          const T *__restrict__ res, TO *__restrict__ y, long M,
                                                          int N, int K, int act, ConvGeom cg)
{
    constexpr int TM = BM / 64, TN = BN / 64;           // 32x32 tiles per wave in m / n
    constexpr int A_CHUNKS = BM * (kBK / 8), B_CHUNKS = BN * (kBK / 8);
    constexpr int A_PER_T = A_CHUNKS / 256, B_PER_T = B_CHUNKS / 256;
    constexpr int TILE_BYTES = (BM + BN) * kBK * 2;
    constexpr int EPI_BYTES = BM * (BN + 4) * 4;
    constexpr int LDS_BYTES = 2 * TILE_BYTES > EPI_BYTES ? 2 * TILE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so with
    // the plain (x = n-tile, y = m-tile) order the blocks sharing an A panel sit on 8 different XCDs and every
    // L2 streams the whole of A.  Give XCD j a CONTIGUOUS run of tiles (n fastest) instead: an A panel is
    // fetched by one L2, only the (small) weight matrix is replicated.
    long m0;
    int n0;
    {
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned q = nblk >> 3, r = nblk & 7, xcd = lin & 7, slot = lin >> 3;
        const unsigned t = xcd * q + (xcd < r ? xcd : r) + slot;
        m0 = (long)(t / gridDim.x) * BM;
        n0 = (int)(t % gridDim.x) * BN;
    }
    const int nk = (K + kBK - 1) / kBK;

    acc16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // register pipeline: stage i holds the operands of tile kt+1+i (PF tiles in flight: short-M GEMMs
    // run ~1 block per CU, so the K loop itself must cover the HBM/L2 latency)
    struct Stage { frag8_t a[A_PER_T], b[B_PER_T]; };
    // Loads are UNCONDITIONAL from clamped (always valid) addresses; out-of-range rows / K chunks are
    // zeroed when the stage is written to LDS.  (Predicated loads become exec-masked branches, after
    // which hipcc's wait insertion falls back to vmcnt(0) and the register pipeline collapses.)
    const int kmax = ((K - 1) / 8) * 8;  // start of the last whole 16-byte chunk of a row
    // conv: per-row output pixel decoded once (rows of a thread do not change over the K loop)
    int cb[A_PER_T], chi[A_PER_T], cwi[A_PER_T];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int row = (tid + i * 256) >> 3;
            const long gr = m0 + row < M ? m0 + row : M - 1;
            const int wo = (int)(gr % cg.wout);
            const long t = gr / cg.wout;
            chi[i] = 2 * (int)(t % cg.hout) - 1;
            cwi[i] = 2 * wo - 1;
            cb[i] = (int)(t / cg.hout);
        }
    }
    auto conv_src = [&](int kt, int i, bool &inside) -> long {  // element offset of the 64-channel run
        const int k0 = kt * kBK;
        const int tap = k0 / cg.cin, ci = k0 - tap * cg.cin;  // block-uniform
        const int hy = chi[i] + tap / 3, wx = cwi[i] + tap % 3;
        inside = hy >= 0 && hy < cg.hin && wx >= 0 && wx < cg.win;
        const int hc = hy < 0 ? 0 : (hy >= cg.hin ? cg.hin - 1 : hy), wc = wx < 0 ? 0 : (wx >= cg.win ? cg.win - 1 : wx);
        return (((long)cb[i] * cg.hin + hc) * cg.win + wc) * cg.cin + ci;
    };
    auto gload = [&](int kt, Stage &st) {
        const int ktc = kt < nk ? kt : nk - 1;
        const int k0 = ktc * kBK;
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int q = tid + i * 256, row = q >> 3, c = q & 7;
            if (CONV) {
                bool inside;
                st.a[i] = *reinterpret_cast<const frag8_t *>(x + conv_src(ktc, i, inside) + c * 8);
            } else {
                const long gr = m0 + row < M ? m0 + row : M - 1;
                const int kc = k0 + c * 8 <= kmax ? k0 + c * 8 : kmax;
                st.a[i] = *reinterpret_cast<const frag8_t *>(x + gr * K + kc);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int q = tid + i * 256, row = q >> 3, c = q & 7;
            const int gn = n0 + row < N ? n0 + row : N - 1;
            const int kc = k0 + c * 8 <= kmax ? k0 + c * 8 : kmax;
            st.b[i] = *reinterpret_cast<const frag8_t *>(w + (long)gn * K + kc);
        }
    };
    auto lstore = [&](int kt, const Stage &st) {
        const int buf = kt & 1, k0 = kt * kBK;
        unsigned char *As = lds + buf * TILE_BYTES, *Bs = As + BM * kBK * 2;
        const frag8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int q = tid + i * 256, row = q >> 3, c = q & 7;
            bool ok = m0 + row < M && k0 + c * 8 < K;
            if (CONV) {
                bool inside;
                conv_src(kt, i, inside);
                ok = ok && inside;  // zero padding of the convolution
            }
            *reinterpret_cast<frag8_t *>(As + row * 128 + swz_chunk(row, c) * 16) = ok ? st.a[i] : zero;
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int q = tid + i * 256, row = q >> 3, c = q & 7;
            const bool ok = n0 + row < N && k0 + c * 8 < K;
            *reinterpret_cast<frag8_t *>(Bs + row * 128 + swz_chunk(row, c) * 16) = ok ? st.b[i] : zero;
        }
    };

    // Static ring of NS = PF+1 register stages: tile t lives in pipe[t % NS].  The K loop is unrolled
    // by NS so every stage index is a compile-time constant -- no register rotation (moving a register
    // with a load in flight forces s_waitcnt vmcnt(0) and drains the pipeline) and hipcc can emit
    // counted vmcnt waits.
    constexpr int NS = PF + 1;
    Stage pipe[NS];
    gload(0, pipe[0]);
    lstore(0, pipe[0]);
#pragma unroll
    for (int i = 1; i < NS; ++i) gload(i, pipe[i]);
    __syncthreads();
    const int r32 = lane & 31, hi = lane >> 5;
    // One K step, no control flow inside (a per-step `if` makes hipcc merge wait-count states
    // conservatively: vmcnt(7) instead of vmcnt(15), i.e. half the tiles in flight).  Steps past the
    // real K range run on zero tiles: loads are clamped, lstore masks k >= K to zero.
    auto kstep = [&](int kt, Stage &load_into, const Stage &store_from) {
        gload(kt + NS, load_into);  // this stage held tile kt, already in LDS
        const unsigned char *As = lds + (kt & 1) * TILE_BYTES, *Bs = As + BM * kBK * 2;
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            frag8_t a[TM], b[TN];
            const int c = kk * 2 + hi;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / 2) + i * 32 + r32;
                a[i] = *reinterpret_cast<const frag8_t *>(As + row * 128 + swz_chunk(row, c) * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / 2) + j * 32 + r32;
                b[j] = *reinterpret_cast<const frag8_t *>(Bs + row * 128 + swz_chunk(row, c) * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mfma<T>::run(b[j], a[i], acc[i][j]);   // swapped: see acc_to_lds
        }
        lstore(kt + 1, store_from);
        __syncthreads();
    };
    // hipcc drains vmcnt(0) at the top of every loop trip (conservative merge over the back edge): full
    // trips cover 4*NS steps, the remainder runs in groups of NS steps.
    const int nkp = (nk + NS - 1) / NS * NS;
    constexpr int UNR = 4 * NS;
    int kt0 = 0;
    for (; kt0 + UNR <= nkp; kt0 += UNR) {
#pragma unroll
        for (int ui = 0; ui < UNR; ++ui) kstep(kt0 + ui, pipe[ui % NS], pipe[(ui + 1) % NS]);
    }
    for (; kt0 < nkp; kt0 += NS) {
#pragma unroll
        for (int ui = 0; ui < NS; ++ui) kstep(kt0 + ui, pipe[ui % NS], pipe[(ui + 1) % NS]);
    }

    tile_epilogue<T, TO, BM, BN>(acc, lds, bias, res, y, M, N, act, m0, n0);
}

// Epilogue of the LAST decoder stage fused into its expand GEMM (Trambav6.py:132-137): the GEMM's N axis is
// P*P groups of C = 128 channels, group g of input pixel (h, w) being output pixel (h*P + g/P, w*P + g%P).  With
// BN = 128 a block tile holds whole groups, so LayerNorm over the group and the 1x1 head (C -> 1) run on the
// accumulators: y (B, H*P, W*P) f32 = <LN(acc), head_w> + head_b.  Neither the (M, 2048) expand output nor the
// (B, 384, 384, 128) normalised map is ever written (2 x 151 MB at batch 4) or read back.
// LayerNorm of the INPUT rows folded into a GEMM (LayerNorm2d -> Linear2d pairs: VSSBlock norm -> in_proj, norm2 -> fc1,
// vmamba.py:384-396): y = act(LN(x) W^T + b) = act(rstd * (x W'^T - mean * colsum) + t), W' = W * gamma (folded once per
// weight version), colsum_n = sum_k W'_nk, t = W beta + b.  The block derives mean / rstd of its own 64 rows from the A tiles it stages anyway (sums taken from
// the staging registers, see lstore); the normalised map is never written or re-read, and the LayerNorm launch disappears.
struct LnIn {
    const float *colsum;
    float eps;
};

struct LnHead {
    const float *ln_w, *ln_b, *head_w;
    float head_b, eps;
    int H, W, P;       // input map H x W, pixel-shuffle factor
    float *out;        // (B, H*P, W*P)
};

template <typename T, int BM>
__device__ __forceinline__ void tile_epilogue_ln_head(acc16_t (&acc)[BM / 64][2], unsigned char *lds, const LnHead hd,
                                                      long M, long m0, int n0)
{
    constexpr int BN = 128, TM = BM / 64, TN = 2, WM = BM / 2, WN = BN / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;
    constexpr int LDE = BN + 4;
    float *ep = reinterpret_cast<float *>(lds);
    acc_to_lds<BM, BN>(acc, ep);
    __syncthreads();
    // 4 lanes per row; lane `sub` owns the float4 chunks sub, sub + 4, ... (channels 16 j + 4 sub + e)
    const int sub = tid & 3;
    float gam[32], bet[32], hw[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 16 * j + 4 * sub;
        const float4 a = *reinterpret_cast<const float4 *>(hd.ln_w + c);
        const float4 b = *reinterpret_cast<const float4 *>(hd.ln_b + c);
        const float4 h = *reinterpret_cast<const float4 *>(hd.head_w + c);
        gam[4 * j] = a.x; gam[4 * j + 1] = a.y; gam[4 * j + 2] = a.z; gam[4 * j + 3] = a.w;
        bet[4 * j] = b.x; bet[4 * j + 1] = b.y; bet[4 * j + 2] = b.z; bet[4 * j + 3] = b.w;
        hw[4 * j] = h.x; hw[4 * j + 1] = h.y; hw[4 * j + 2] = h.z; hw[4 * j + 3] = h.w;
    }
    const int g = n0 / BN;                         // group of this tile
#pragma unroll
    for (int it = 0; it < BM / 64; ++it) {
        const int row = it * 64 + (tid >> 2);
        const long grow = m0 + row;
        float v[32];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 t = *reinterpret_cast<const float4 *>(ep + row * LDE + 16 * j + 4 * sub);
            v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
        }
        float s1 = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) s1 += v[q];
        s1 += __shfl_xor(s1, 1, 64);
        s1 += __shfl_xor(s1, 2, 64);
        const float mean = s1 * (1.f / 128.f);
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float d = v[q] - mean;
            s2 = fmaf(d, d, s2);
        }
        s2 += __shfl_xor(s2, 1, 64);
        s2 += __shfl_xor(s2, 2, 64);
        const float rstd = rsqrtf(s2 * (1.f / 128.f) + hd.eps);
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float yn = (v[q] - mean) * rstd * gam[q] + bet[q];
            dot = fmaf(yn, hw[q], dot);   // fp32 throughout: no 16-bit rounding between GEMM, norm and head
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        if (sub == 0 && grow < M) {
            const unsigned pix = (unsigned)grow;                     // (b*H + h)*W + w, M < 2^31 host-checked
            const unsigned wi = pix % (unsigned)hd.W, bh = pix / (unsigned)hd.W;
            const unsigned hh = bh % (unsigned)hd.H, bb = bh / (unsigned)hd.H;
            const long o = ((long)bb * (hd.H * hd.P) + (hh * hd.P + g / hd.P)) * (long)(hd.W * hd.P) + (wi * hd.P + g % hd.P);
            hd.out[o] = dot + hd.head_b;
        }
    }
}

// Lean form of the tile kernel for the plain GEMM with K % 64 == 0 (every 1x1 convolution of the model).
// The generic kernel spends ~37 VALU instructions per K step on addresses, clamps and masks -- more issue
// time than its 4 MFMAs -- and these GEMMs run ~1 wave per SIMD, so that time is exposed.  Here:
//   * operands come through buffer descriptors based at the block's first row: the per-thread byte offset
//     is loop-invariant, the K step rides in the scalar offset, rows past M / N read as zero through the
//     hardware range check -> no clamps, no masks, no 64-bit address arithmetic in the loop;
//   * LDS write addresses are loop-invariant; fragment read addresses differ per 16-deep slice only by an
//     XOR of bits 5-6 (chunk swizzle), and the buffer half is a compile-time immediate;
//   * exactly nk steps run (ring trips + a statically indexed tail), so no zero-padded dummy steps.
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

// acc + lo + hi and acc + lo^2 + hi^2 of one packed pair of 16-bit elements
template <typename T> __device__ __forceinline__ float dot2_ones(unsigned v, float acc);
template <typename T> __device__ __forceinline__ float dot2_self(unsigned v, float acc);
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef _Float16 hf2_t __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ float dot2_ones<__hip_bfloat16>(unsigned v, float acc)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, v), __builtin_bit_cast(bf2_t, 0x3f803f80u), acc, false);
}
template <> __device__ __forceinline__ float dot2_self<__hip_bfloat16>(unsigned v, float acc)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, v), __builtin_bit_cast(bf2_t, v), acc, false);
}
template <> __device__ __forceinline__ float dot2_ones<__half>(unsigned v, float acc)
{
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(hf2_t, v), __builtin_bit_cast(hf2_t, 0x3c003c00u), acc, false);
}
template <> __device__ __forceinline__ float dot2_self<__half>(unsigned v, float acc)
{
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(hf2_t, v), __builtin_bit_cast(hf2_t, v), acc, false);
}
template <> __device__ __forceinline__ float dot2_ones<float>(unsigned, float acc) { return acc; }
template <> __device__ __forceinline__ float dot2_self<float>(unsigned, float acc) { return acc; }

template <typename T, typename TO, int BM, int BN, int PF, bool LNHEAD = false, int NWM = 2, bool LNIN = false>
__global__ __launch_bounds__(256) void linear_lean_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                         const float *__restrict__ bias,
                                                         const T *__restrict__ res, TO *__restrict__ y, long M,
                                                         int N, int K, int act, const T *__restrict__ x2, int K1,
                                                         LnHead hd = LnHead{}, LnIn li = LnIn{nullptr, 0.f})
{
    // x2 != nullptr: A = [x (M, K1) | x2 (M, K - K1)], both halves whole K steps (no concatenation in memory)
    using WL = WaveLayout<BM, BN, NWM>;
    constexpr int TM = WL::TM, TN = WL::TN;
    static_assert(!LNHEAD || NWM == 2, "the fused head epilogue is written for the 2 x 2 layout");
    constexpr int A_PER_T = BM * (kBK / 8) / 256, B_PER_T = BN * (kBK / 8) / 256;
    constexpr int TILE_BYTES = (BM + BN) * kBK * 2;
    constexpr int EPI_BYTES = BM * (BN + 4) * 4;
    static_assert((BM * (kBK / 8)) % 256 == 0 && (BN * (kBK / 8)) % 256 == 0, "whole 16-byte chunks per thread");
    constexpr int LDS_BYTES = 2 * TILE_BYTES > EPI_BYTES ? 2 * TILE_BYTES : EPI_BYTES;
    constexpr int NS = PF + 1;
    static_assert(NS % 2 == 0, "the LDS buffer parity must be a compile-time constant");
    static_assert(!LNIN || (BM == 64 && NWM == 2 && !LNHEAD), "the input-LayerNorm form is written for 64-row tiles");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES + (LNIN ? BM * 8 : 0)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool compute = wave < NWM * WL::NWN;                     // (3 x 1 layout: the fourth wave only moves data)
    const int wm = compute ? wave / WL::NWN : 0, wn = wave % WL::NWN;
    const int r32 = lane & 31, hi = lane >> 5;
    long m0;
    int n0;
    {   // XCD-aware tile order (see linear_tiled_kernel)
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned q = nblk >> 3, r = nblk & 7, xcd = lin & 7, slot = lin >> 3;
        const unsigned t = xcd * q + (xcd < r ? xcd : r) + slot;
        m0 = (long)(t / gridDim.x) * BM;
        n0 = (int)(t % gridDim.x) * BN;
    }
    const int nk = K / kBK;
    const unsigned rowb = (unsigned)K * 2u;   // bytes per operand row
    const long mrows = M - m0 < BM ? M - m0 : BM;
    const int nrows = N - n0 < BN ? N - n0 : BN;
    const int ka = x2 ? K1 : K;                       // row length of the first A source
    const unsigned rowa = (unsigned)ka * 2u, rowa2 = (unsigned)(K - ka) * 2u;
    const int nk1 = ka / kBK;                         // K steps served by the first source
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(x + m0 * ka, (unsigned)mrows * rowa);
    const __amdgpu_buffer_rsrc_t ra2 = make_rsrc(x2 ? x2 + m0 * (K - ka) : x, x2 ? (unsigned)mrows * rowa2 : 0u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(w + (long)n0 * K, (unsigned)nrows * rowb);

    // loop-invariant per-thread offsets: global (row*K + 8c elements) and LDS (swizzled chunk)
    unsigned ag[A_PER_T], ag2[A_PER_T], bg[B_PER_T], al[A_PER_T], bl[B_PER_T];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
        const int q = tid + i * 256, row = q >> 3, c = q & 7;
        ag[i] = (unsigned)row * rowa + (unsigned)c * 16u;
        ag2[i] = (unsigned)row * rowa2 + (unsigned)c * 16u;
        al[i] = (unsigned)(row * 128 + swz_chunk(row, c) * 16);
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
        const int q = tid + i * 256, row = q >> 3, c = q & 7;
        bg[i] = (unsigned)row * rowb + (unsigned)c * 16u;
        bl[i] = (unsigned)(BM * kBK * 2 + row * 128 + swz_chunk(row, c) * 16);
    }
    // fragment read offsets of the 16-deep slice kk = 0; slice kk is the same offset ^ (kk * 32)
    unsigned ard[TM], brd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WL::WM + i * 32 + r32;
        ard[i] = (unsigned)(row * 128 + swz_chunk(row, hi) * 16);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * WL::WN + j * 32 + r32;
        brd[j] = (unsigned)(BM * kBK * 2 + row * 128 + swz_chunk(row, hi) * 16);
    }

    acc16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Stage { v4u_t a[A_PER_T], b[B_PER_T]; };
    auto gload = [&](int kt, Stage &st) {   // a K step past the end reads the following rows / zeros: never used
        const unsigned so = (unsigned)kt * (kBK * 2);
        // A comes from the first source for kt < nk1, from the second after it: a block-uniform SELECT of
        // descriptor, scalar offset and per-thread offset (not a branch around the loads)
        const bool first = kt < nk1;
        const __amdgpu_buffer_rsrc_t rsel = first ? ra : ra2;
        const unsigned soa = first ? so : so - (unsigned)nk1 * (kBK * 2);
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i)
            st.a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsel, first ? ag[i] : ag2[i], soa, 0);
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) st.b[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, bg[i], so, 0);
    };
    // LNIN: sum and sum of squares of the A rows, taken from the staging registers on their way to LDS -- thread t stages
    // chunk t & 7 of rows (t >> 3) + 32 i at every K step, so the row statistics cost no load of their own (a prologue
    // that re-read the block's rows moved 1.5x the operand bytes and two dependent round trips: +4..6 us per launch).
    // v_dot2c_f32_bf16 / _f16: two elements per instruction straight from the packed dword, exact products, fp32 sums.
    float rs1[A_PER_T], rs2[A_PER_T];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) rs1[i] = rs2[i] = 0.f;
    auto lstore = [&](int par, const Stage &st, bool real = true) {
        unsigned char *base = lds + par * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) *reinterpret_cast<v4u_t *>(base + al[i]) = st.a[i];
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) *reinterpret_cast<v4u_t *>(base + bl[i]) = st.b[i];
        if constexpr (LNIN) {
            if (real) {   // (block-uniform) a tile past the end of K holds the next rows' elements: not part of the statistics
#pragma unroll
                for (int i = 0; i < A_PER_T; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rs1[i] = dot2_ones<T>(st.a[i][e], rs1[i]);
                        rs2[i] = dot2_self<T>(st.a[i][e], rs2[i]);
                    }
            }
        }
    };

    Stage pipe[NS];
    gload(0, pipe[0]);
    lstore(0, pipe[0]);
#pragma unroll
    for (int i = 1; i < NS; ++i) gload(i, pipe[i]);
    __syncthreads();
    // one K step; `par` (LDS half holding tile kt) is a literal at every call site
    auto kstep = [&](int kt, int par, Stage &load_into, const Stage &store_from) {
        gload(kt + NS, load_into);  // this stage held tile kt, already in LDS
        const unsigned char *base = lds + par * TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            if (NWM != 2 && !compute) break;   // wave-uniform
            frag8_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const frag8_t *>(base + (ard[i] ^ (unsigned)(kk * 32)));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const frag8_t *>(base + (brd[j] ^ (unsigned)(kk * 32)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mfma<T>::run(b[j], a[i], acc[i][j]);   // swapped: see acc_to_lds
        }
        lstore(par ^ 1, store_from, kt + 1 < nk);
        __syncthreads();
    };
    constexpr int UNR = 4 * NS;   // hipcc drains vmcnt at the top of every loop trip: long trips
    int kt0 = 0;
    for (; kt0 + UNR <= nk; kt0 += UNR) {
#pragma unroll
        for (int ui = 0; ui < UNR; ++ui) kstep(kt0 + ui, ui & 1, pipe[ui % NS], pipe[(ui + 1) % NS]);
    }
    for (; kt0 + NS <= nk; kt0 += NS) {
#pragma unroll
        for (int ui = 0; ui < NS; ++ui) kstep(kt0 + ui, ui & 1, pipe[ui % NS], pipe[(ui + 1) % NS]);
    }
    // the 0 .. NS-1 remaining steps, ring indices still static (kt0 is a multiple of NS)
#pragma unroll
    for (int ui = 0; ui < NS - 1; ++ui)
        if (kt0 + ui < nk) kstep(kt0 + ui, ui & 1, pipe[ui % NS], pipe[(ui + 1) % NS]);

    if constexpr (LNHEAD) {
        static_assert(BN == 128, "one tile = whole 128-channel groups");
        tile_epilogue_ln_head<T, BM>(acc, lds, hd, M, m0, n0);
    } else if constexpr (LNIN) {
        float2 *rowstat = reinterpret_cast<float2 *>(lds + LDS_BYTES);   // read by the epilogue behind its own barrier
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            float s1 = rs1[i], s2 = rs2[i];
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {      // the 8 lanes that stage one row
                s1 += __shfl_xor(s1, m, 64);
                s2 += __shfl_xor(s2, m, 64);
            }
            const float mean = s1 / (float)K;
            const float var = fmaxf(s2 / (float)K - mean * mean, 0.f);
            if ((tid & 7) == 0) rowstat[(tid >> 3) + 32 * i] = make_float2(mean, rsqrtf(var + li.eps));
        }
        tile_epilogue<T, TO, BM, BN, NWM>(acc, lds, bias, res, y, M, N, act, m0, n0, li.colsum,
                                          reinterpret_cast<const float2 *>(lds + LDS_BYTES));
    } else {
        tile_epilogue<T, TO, BM, BN, NWM>(acc, lds, bias, res, y, M, N, act, m0, n0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Plain GEMM with the operand tiles staged by LDS-DMA (buffer_load ... lds): 64x64 block tile, K step 64, 2 x 2 waves.
// In the register-staged kernel above the K step's largest single cost is the VGPR -> LDS write of the staged tiles
// (ablation on the model's shapes, r02: dropping the ds_write_b128s takes 15-32 % off the whole launch, more than dropping
// the global loads or three quarters of the fragment reads; a ds_write_b128 occupies the LDS path for 13 cycles per KB).
// Here a tile travels L2 -> LDS without touching a VGPR:
//   * NSTG LDS stages (3, or 4 for long K on a one-block-per-CU grid) of (A 64 x 64, B 64 x 64) 16-bit = 16 KB; waves 0-1 fetch A, waves 2-3 fetch B, four 1 KB pieces per
//     wave and K step.  A piece is 8 rows x 128 B; lane i lands at piece + 16 i, so the XOR chunk swizzle of the fragment
//     reads is applied on the GLOBAL side: lane i fetches chunk (i & 7) ^ ((row >> 1) & 7) of row piece*8 + (i >> 3)
//     (the 8 lanes of a row still cover one 128-byte line);
//   * step kt:  s_waitcnt vmcnt(4)  my pieces of tile kt have landed (only the four pieces of each later tile already
//                                   requested may be outstanding: 4 (NSTG - 2), less on the last steps -- no tile past the
//                                   end of K is ever fetched)
//               s_barrier           everyone's have; everyone has read tile kt-1 out of the stage about to be refilled
//               DMA tile kt + NSTG - 1 -> the stage tile kt - 1 was read out of
//               8 ds_read_b128 (hand-written: hipcc orders an LDS read it can see behind EVERY pending LDS-DMA, i.e.
//               vmcnt(0) per step), counted lgkmcnt waits, 4 MFMAs
//   * ~70 VGPRs, no staging registers; rows past M / N read as zero through the descriptor's range check.
// Same epilogue (tile_epilogue) as the other forms.
#define TRAMBA_DSR128_(OUT, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(OUT) : "v"(ADDR) : "memory")

template <typename T, typename TO, int NSTG, bool LNIN = false, bool DUAL = false>
__global__ __launch_bounds__(256) void linear_dma_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                        const float *__restrict__ bias, const T *__restrict__ res,
                                                        TO *__restrict__ y, long M, int N, int K, int act,
                                                        LnIn li = LnIn{nullptr, 0.f}, TO *__restrict__ y_pre = nullptr)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (vector-register asm in a kernel template: see ss2d_scan_dma_kernel)
    constexpr int BM = 64, BN = 64;
    constexpr int TILE_BYTES = (BM + BN) * kBK * 2;          // 16 KB
    constexpr int EPI_BYTES = BM * (BN + 4) * 4;
    constexpr int LDS_BYTES = NSTG * TILE_BYTES > EPI_BYTES ? NSTG * TILE_BYTES : EPI_BYTES;
    static_assert(NSTG == 2 || NSTG == 3 || NSTG == 4, "stage index = kt % NSTG with the loop unrolled by NSTG (up to 8 stages are coded)");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES + (LNIN ? BM * 8 : 0)];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;
    long m0;
    int n0;
    {   // XCD-aware tile order (see linear_tiled_kernel)
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned q = nblk >> 3, r = nblk & 7, xcd = lin & 7, slot = lin >> 3;
        const unsigned t = xcd * q + (xcd < r ? xcd : r) + slot;
        m0 = (long)(t / gridDim.x) * BM;
        n0 = (int)(t % gridDim.x) * BN;
    }
    const int nk = K / kBK;
    const unsigned rowb = (unsigned)K * 2u;
    const long mrows = M - m0 < BM ? M - m0 : BM;
    const int nrows = N - n0 < BN ? N - n0 : BN;
    // waves 0-1 stage A (rows of x), waves 2-3 stage B (rows of w): one descriptor per wave, wave-uniform
    const bool stage_b = wave >= 2;
    const __amdgpu_buffer_rsrc_t rs = stage_b ? make_rsrc(w + (long)n0 * K, (unsigned)nrows * rowb)
                                              : make_rsrc(x + m0 * K, (unsigned)mrows * rowb);
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = ((wave & 1) * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[j] = (unsigned)row * rowb + (unsigned)c * 16u;
    }
    unsigned char *mine = lds + (stage_b ? BM * kBK * 2 : 0) + (wave & 1) * 4096;   // my four pieces inside a stage
    auto issue = [&](int kt, int stg) {
        const unsigned so = (unsigned)kt * (kBK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(mine + stg * TILE_BYTES + j * 1024), 16, voff[j], so, 0,
                                                     0);
    };
    // fragment read addresses (absolute LDS bytes) of the four 16-deep slices: chunk (2 kk + hi) ^ ((row >> 1) & 7)
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    unsigned aad[4], bad[4];
    {
        const int ra_ = wm * 32 + r32, rb_ = wn * 32 + r32;
        const unsigned a0 = lbase + (unsigned)(ra_ * 128 + ((hi ^ ((ra_ >> 1) & 7)) * 16));
        const unsigned b0 = lbase + (unsigned)(BM * kBK * 2 + rb_ * 128 + ((hi ^ ((rb_ >> 1) & 7)) * 16));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            aad[kk] = a0 ^ (unsigned)(kk * 32);      // (lds is 1 KB aligned: the XOR stays inside the row)
            bad[kk] = b0 ^ (unsigned)(kk * 32);
        }
    }
    acc16_t acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    // LNIN (LayerNorm of the input rows folded in, see LnIn): sum and sum of squares of the A rows, read back from the
    // staged tile -- thread t adds the 16-byte pieces at chunk POSITION t & 7 of rows t >> 3 and (t >> 3) + 32 at every K
    // step (whatever chunk the swizzle put there: a row's statistics run over all of its chunks), two more ds_read_b128
    // beside the eight fragment reads
    float rs1[2] = {0.f, 0.f}, rs2[2] = {0.f, 0.f};
    const unsigned sad = lbase + (unsigned)((tid >> 3) * 128 + (tid & 7) * 16);

    constexpr int DEPTH = NSTG - 1;                  // tiles in flight ahead of the one being multiplied
#pragma unroll
    for (int t = 0; t < DEPTH; ++t)
        if (t < nk) issue(t, t);
    // stages 4..7 lie past the 16-bit offset field of ds_read: a second set of addresses, 64 KB up
    unsigned aad_hi[4], bad_hi[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        aad_hi[kk] = aad[kk] + 65536u;
        bad_hi[kk] = bad[kk] + 65536u;
    }
    const unsigned sad_hi = sad + 65536u;
    auto kstep = [&](int kt, auto stg_c) {
        constexpr int STG = decltype(stg_c)::value;
        // my pieces of tile kt have landed: only the four pieces of each LATER tile already requested may be in flight
        const int later = nk - 1 - kt;
        if (later >= DEPTH - 1) {
            if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        } else {
            switch (later) {   // (block-uniform) the last DEPTH - 1 steps
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + DEPTH < nk) issue(kt + DEPTH, (STG + DEPTH) % NSTG);
        frag8_t a[4], b[4];
        v4u_t sv[2];
        const unsigned *aa = STG < 4 ? aad : aad_hi, *ba = STG < 4 ? bad : bad_hi;
        const unsigned sa = STG < 4 ? sad : sad_hi;
#define TRAMBA_RD_STG_(S, S2)                                                                       \
    if constexpr (LNIN) { TRAMBA_DSR128_(sv[0], sa, S); TRAMBA_DSR128_(sv[1], sa, S2); }             \
    TRAMBA_DSR128_(a[0], aa[0], S); TRAMBA_DSR128_(b[0], ba[0], S); TRAMBA_DSR128_(a[1], aa[1], S);  \
    TRAMBA_DSR128_(b[1], ba[1], S); TRAMBA_DSR128_(a[2], aa[2], S); TRAMBA_DSR128_(b[2], ba[2], S);  \
    TRAMBA_DSR128_(a[3], aa[3], S); TRAMBA_DSR128_(b[3], ba[3], S)
        if constexpr ((STG & 3) == 0) { TRAMBA_RD_STG_(0, 4096); }
        else if constexpr ((STG & 3) == 1) { TRAMBA_RD_STG_(16384, 20480); }
        else if constexpr ((STG & 3) == 2) { TRAMBA_RD_STG_(32768, 36864); }
        else { TRAMBA_RD_STG_(49152, 53248); }
#undef TRAMBA_RD_STG_
        if constexpr (LNIN) {   // (the two statistics reads were issued first: they are done once eight reads remain)
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(sv[0]), "+v"(sv[1]) : : "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rs1[i] = dot2_ones<T>(sv[i][e], rs1[i]);
                    rs2[i] = dot2_self<T>(sv[i][e], rs2[i]);
                }
        }
        // counted waits: slice kk needs the first 2 (kk + 1) reads; every destination is named so that no MFMA moves above
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0]), "+v"(b[0]) : : "memory");
        acc[0][0] = Mfma<T>::run(b[0], a[0], acc[0][0]);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[1]), "+v"(b[1]) : : "memory");
        acc[0][0] = Mfma<T>::run(b[1], a[1], acc[0][0]);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[2]), "+v"(b[2]) : : "memory");
        acc[0][0] = Mfma<T>::run(b[2], a[2], acc[0][0]);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[3]), "+v"(b[3]) : : "memory");
        acc[0][0] = Mfma<T>::run(b[3], a[3], acc[0][0]);
    };
    int kt0 = 0;
    for (; kt0 + NSTG <= nk; kt0 += NSTG) {
        kstep(kt0, std::integral_constant<int, 0>{});
        kstep(kt0 + 1, std::integral_constant<int, 1>{});
        if constexpr (NSTG > 2) kstep(kt0 + 2, std::integral_constant<int, 2>{});
        if constexpr (NSTG > 3) kstep(kt0 + 3, std::integral_constant<int, 3>{});
        if constexpr (NSTG > 4) {
            kstep(kt0 + 4, std::integral_constant<int, 4>{});
            kstep(kt0 + 5, std::integral_constant<int, 5>{});
            kstep(kt0 + 6, std::integral_constant<int, 6>{});
            kstep(kt0 + 7, std::integral_constant<int, 7>{});
        }
    }
    if (kt0 < nk) kstep(kt0, std::integral_constant<int, 0>{});
    if constexpr (NSTG > 2) {
        if (kt0 + 1 < nk) kstep(kt0 + 1, std::integral_constant<int, 1>{});
    }
    if constexpr (NSTG > 3) {
        if (kt0 + 2 < nk) kstep(kt0 + 2, std::integral_constant<int, 2>{});
    }
    if constexpr (NSTG > 4) {
        if (kt0 + 3 < nk) kstep(kt0 + 3, std::integral_constant<int, 3>{});
        if (kt0 + 4 < nk) kstep(kt0 + 4, std::integral_constant<int, 4>{});
        if (kt0 + 5 < nk) kstep(kt0 + 5, std::integral_constant<int, 5>{});
        if (kt0 + 6 < nk) kstep(kt0 + 6, std::integral_constant<int, 6>{});
    }
    __syncthreads();                                       // every wave has read the last tile: the epilogue reuses the LDS
    if constexpr (LNIN) {
        float2 *rowstat = reinterpret_cast<float2 *>(lds + LDS_BYTES);   // read by the epilogue behind its own barrier
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s1 = rs1[i], s2 = rs2[i];
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {      // the 8 lanes that cover one row
                s1 += __shfl_xor(s1, m, 64);
                s2 += __shfl_xor(s2, m, 64);
            }
            const float mean = s1 / (float)K;
            const float var = fmaxf(s2 / (float)K - mean * mean, 0.f);
            if ((tid & 7) == 0) rowstat[(tid >> 3) + 32 * i] = make_float2(mean, rsqrtf(var + li.eps));
        }
        tile_epilogue<T, TO, BM, BN, 2>(acc, lds, bias, res, y, M, N, act, m0, n0, li.colsum,
                                        reinterpret_cast<const float2 *>(lds + LDS_BYTES));
    } else {
        tile_epilogue<T, TO, BM, BN, 2>(acc, lds, bias, res, y, M, N, act, m0, n0, nullptr, nullptr, DUAL ? y_pre : nullptr);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// r04: the same 64 x 64 tile with PRODUCER and CONSUMER waves -- 512 threads, waves 0-3 multiply (2 x 2 of 32 x 32, as above),
// waves 4-7 only fetch (4-5 the A pieces, 6-7 the B pieces, four 1 KB pieces per wave and K step, the same global-side swizzle).
// Why: in linear_dma_kernel every wave issues its four LDS-DMA pieces and THEN reads its fragments and multiplies; a
// `buffer_load ... lds` costs its wave 60-185 issue cycles (MI355X guide, "LDS-DMA piece issue cost"), so with one workgroup on a
// CU -- the 288-tile launches of the 24 x 24 stage, 73 per forward -- a K step is a serial chain of ~4 x 100 (DMA issue) + ~130
// (eight ds_read_b128 round trip) + 128 (four MFMAs) cycles that nothing overlaps: the measured 0.25 us per 64-deep step.  Split
// by role, the DMA issue of tile kt + DEPTH runs in the loader waves WHILE the multiplier waves read and multiply tile kt; one
// s_barrier per step joins them (the loaders wait for their own pieces of tile kt + 1 with a counted vmcnt in front of it).
// The r03 96 x 64 form had ONE loader wave issuing twenty pieces per step -- the loader was the bottleneck; here every SIMD
// hosts one loader wave and one multiplier wave (a workgroup's waves go to the SIMDs cyclically).
// LNIN: the row statistics are read back from the staged tile by the loader waves (they have nothing else to do between
// their DMA issue and the next barrier), same arithmetic and summation order as linear_dma_kernel: bit-identical results.
#define TRAMBA_DSR128_(OUT, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(OUT) : "v"(ADDR) : "memory")
template <typename T, typename TO, int NSTG, bool LNIN = false, bool DUAL = false>
__global__ __launch_bounds__(512) void linear_pc_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                       const float *__restrict__ bias, const T *__restrict__ res,
                                                       TO *__restrict__ y, long M, int N, int K, int act,
                                                       LnIn li = LnIn{nullptr, 0.f}, TO *__restrict__ y_pre = nullptr)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (vector-register asm in a kernel template: see ss2d_scan_dma_kernel)
    constexpr int BM = 64, BN = 64;
    constexpr int TILE_BYTES = (BM + BN) * kBK * 2;          // 16 KB
    constexpr int EPI_BYTES = BM * (BN + 4) * 4;
    constexpr int LDS_BYTES = NSTG * TILE_BYTES > EPI_BYTES ? NSTG * TILE_BYTES : EPI_BYTES;
    static_assert(NSTG == 3 || NSTG == 4, "stage index = kt % NSTG with the loop unrolled by NSTG");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES + (LNIN ? BM * 8 : 0)];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;                           // wave-uniform
    const int cw = wave & 3;
    const int wm = cw >> 1, wn = cw & 1;
    const int r32 = lane & 31, hi = lane >> 5;
    long m0;
    int n0;
    {   // XCD-aware tile order (see linear_tiled_kernel)
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned q = nblk >> 3, r = nblk & 7, xcd = lin & 7, slot = lin >> 3;
        const unsigned t = xcd * q + (xcd < r ? xcd : r) + slot;
        m0 = (long)(t / gridDim.x) * BM;
        n0 = (int)(t % gridDim.x) * BN;
    }
    const int nk = K / kBK;
    const unsigned rowb = (unsigned)K * 2u;
    const long mrows = M - m0 < BM ? M - m0 : BM;
    const int nrows = N - n0 < BN ? N - n0 : BN;
    // loader waves 4-5 stage A (rows of x), 6-7 stage B (rows of w): one descriptor per wave, wave-uniform
    const bool stage_b = cw >= 2;
    const __amdgpu_buffer_rsrc_t rs = stage_b ? make_rsrc(w + (long)n0 * K, (unsigned)nrows * rowb)
                                              : make_rsrc(x + m0 * K, (unsigned)mrows * rowb);
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = ((cw & 1) * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[j] = (unsigned)row * rowb + (unsigned)c * 16u;
    }
    unsigned char *mine = lds + (stage_b ? BM * kBK * 2 : 0) + (cw & 1) * 4096;   // my four pieces inside a stage
    auto issue = [&](int kt, int stg) {
        const unsigned so = (unsigned)kt * (kBK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(mine + stg * TILE_BYTES + j * 1024), 16, voff[j], so, 0,
                                                     0);
    };
    // fragment read addresses (absolute LDS bytes) of the four 16-deep slices: chunk (2 kk + hi) ^ ((row >> 1) & 7)
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    unsigned aad[4], bad[4];
    {
        const int ra_ = wm * 32 + r32, rb_ = wn * 32 + r32;
        const unsigned a0 = lbase + (unsigned)(ra_ * 128 + ((hi ^ ((ra_ >> 1) & 7)) * 16));
        const unsigned b0 = lbase + (unsigned)(BM * kBK * 2 + rb_ * 128 + ((hi ^ ((rb_ >> 1) & 7)) * 16));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            aad[kk] = a0 ^ (unsigned)(kk * 32);      // (lds is 1 KB aligned: the XOR stays inside the row)
            bad[kk] = b0 ^ (unsigned)(kk * 32);
        }
    }
    acc16_t acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    // LNIN: multiplier thread lt adds the 16-byte pieces at chunk POSITION lt & 7 of rows lt >> 3 and (lt >> 3) + 32 at every K
    // step, two more ds_read_b128 issued in front of its eight fragment reads (as in linear_dma_kernel; read by the loader
    // waves instead, the statistics sat between their DMA issue and the next barrier: 10-20 % slower on the many-tile shapes)
    const int lt = tid & 255;
    float rs1[2] = {0.f, 0.f}, rs2[2] = {0.f, 0.f};
    const unsigned sad = lbase + (unsigned)((lt >> 3) * 128 + (lt & 7) * 16);

    constexpr int DEPTH = NSTG - 1;                  // tiles in flight ahead of the one being multiplied
    if (loader) {
#pragma unroll
        for (int t = 0; t < DEPTH; ++t)
            if (t < nk) issue(t, t);
    }
    auto kstep = [&](int kt, auto stg_c) {
        constexpr int STG = decltype(stg_c)::value;
        if (loader) {
            // my pieces of tile kt have landed: only the four pieces of each LATER tile already requested may be in flight
            const int later = nk - 1 - kt;
            if (later >= DEPTH - 1) {
                if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                switch (later) {   // (block-uniform) the last DEPTH - 1 steps
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                }
            }
        }
        __builtin_amdgcn_s_barrier();    // tile kt is in LDS for everyone; everyone has read tile kt - 1 out of the stage refilled next
        asm volatile("" ::: "memory");
        if (loader) {
            if (kt + DEPTH < nk) issue(kt + DEPTH, (STG + DEPTH) % NSTG);
        } else {
            frag8_t a[4], b[4];
            v4u_t sv[2];
            if constexpr (LNIN) {
                if constexpr (STG == 0) { TRAMBA_DSR128_(sv[0], sad, 0); TRAMBA_DSR128_(sv[1], sad, 4096); }
                else if constexpr (STG == 1) { TRAMBA_DSR128_(sv[0], sad, 16384); TRAMBA_DSR128_(sv[1], sad, 20480); }
                else if constexpr (STG == 2) { TRAMBA_DSR128_(sv[0], sad, 32768); TRAMBA_DSR128_(sv[1], sad, 36864); }
                else { TRAMBA_DSR128_(sv[0], sad, 49152); TRAMBA_DSR128_(sv[1], sad, 53248); }
            }
#define TRAMBA_RD_STG_(S)                                                                            \
    TRAMBA_DSR128_(a[0], aad[0], S); TRAMBA_DSR128_(b[0], bad[0], S); TRAMBA_DSR128_(a[1], aad[1], S); \
    TRAMBA_DSR128_(b[1], bad[1], S); TRAMBA_DSR128_(a[2], aad[2], S); TRAMBA_DSR128_(b[2], bad[2], S); \
    TRAMBA_DSR128_(a[3], aad[3], S); TRAMBA_DSR128_(b[3], bad[3], S)
            if constexpr (STG == 0) { TRAMBA_RD_STG_(0); }
            else if constexpr (STG == 1) { TRAMBA_RD_STG_(16384); }
            else if constexpr (STG == 2) { TRAMBA_RD_STG_(32768); }
            else { TRAMBA_RD_STG_(49152); }
#undef TRAMBA_RD_STG_
            // counted waits: slice kk needs the first 2 (kk + 1) reads; every destination is named so that no MFMA moves above
            if constexpr (LNIN) {   // (the two statistics reads were issued first: they are done once eight reads remain)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(sv[0]), "+v"(sv[1]) : : "memory");
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rs1[i] = dot2_ones<T>(sv[i][e], rs1[i]);
                        rs2[i] = dot2_self<T>(sv[i][e], rs2[i]);
                    }
            }
            // (sched_barrier: without it hipcc sinks the first MFMA below the NEXT wait -- legal, the waits only grow stricter --
            //  and the four MFMAs start two reads later than they could)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0]), "+v"(b[0]) : : "memory");
            acc[0][0] = Mfma<T>::run(b[0], a[0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[1]), "+v"(b[1]) : : "memory");
            acc[0][0] = Mfma<T>::run(b[1], a[1], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[2]), "+v"(b[2]) : : "memory");
            acc[0][0] = Mfma<T>::run(b[2], a[2], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[3]), "+v"(b[3]) : : "memory");
            acc[0][0] = Mfma<T>::run(b[3], a[3], acc[0][0]);
        }
    };
    int kt0 = 0;
    for (; kt0 + NSTG <= nk; kt0 += NSTG) {
        kstep(kt0, std::integral_constant<int, 0>{});
        kstep(kt0 + 1, std::integral_constant<int, 1>{});
        kstep(kt0 + 2, std::integral_constant<int, 2>{});
        if constexpr (NSTG > 3) kstep(kt0 + 3, std::integral_constant<int, 3>{});
    }
    if (kt0 < nk) kstep(kt0, std::integral_constant<int, 0>{});
    if (kt0 + 1 < nk) kstep(kt0 + 1, std::integral_constant<int, 1>{});
    if constexpr (NSTG > 3) {
        if (kt0 + 2 < nk) kstep(kt0 + 2, std::integral_constant<int, 2>{});
    }
    __syncthreads();                                       // every wave has read the last tile: the epilogue reuses the LDS
    if constexpr (LNIN) {
        float2 *rowstat = reinterpret_cast<float2 *>(lds + LDS_BYTES);   // read by the epilogue behind its own barrier
        if (!loader) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float s1 = rs1[i], s2 = rs2[i];
#pragma unroll
                for (int m = 1; m < 8; m <<= 1) {      // the 8 lanes that cover one row
                    s1 += __shfl_xor(s1, m, 64);
                    s2 += __shfl_xor(s2, m, 64);
                }
                const float mean = s1 / (float)K;
                const float var = fmaxf(s2 / (float)K - mean * mean, 0.f);
                if ((lt & 7) == 0) rowstat[(lt >> 3) + 32 * i] = make_float2(mean, rsqrtf(var + li.eps));
            }
        }
        tile_epilogue<T, TO, BM, BN, 2, 512>(acc, lds, bias, res, y, M, N, act, m0, n0, li.colsum,
                                             reinterpret_cast<const float2 *>(lds + LDS_BYTES));
    } else {
        tile_epilogue<T, TO, BM, BN, 2, 512>(acc, lds, bias, res, y, M, N, act, m0, n0, nullptr, nullptr, DUAL ? y_pre : nullptr);
    }
#endif
}
#undef TRAMBA_DSR128_

// The same staging on a 96 x 64 tile (r03) -- A MEASUREMENT FORM (TRAMBA_TUNE_GEMM_TILE 15), never the library's choice: 3 compute
// waves of 32 rows x 64 columns each, the fourth wave only fetches and streams the epilogue (WaveLayout NWM = 3).  The idea: if
// the K loop ran at the CU's L2 -> LDS fill rate (as the weight-gradient GEMM's does), what would count is bytes filled per
// output row and K step -- 20 KB for 96 rows against 16 KB for 64 -- and the rounds of the chip: M = 2304 is 24 x 96, so the
// 512-wide GEMMs of the 15-block stage make 192 workgroups = ONE round on 256 CUs where 64 x 64 tiles make 288 = two.
// Measured (scripts/bench_gemm_tile96.py, profiles/r03h_gemm_tile96.txt): 17.6 against 13.2 us at M = 2304, N = 512, K = 2048,
// slower on 17 of 21 shapes -- with one workgroup per CU and one wave per SIMD a K step is a serial chain (wait, barrier,
// 12 LDS reads, 8 MFMAs) that nothing overlaps; the 64 x 64 form keeps 3 workgroups per CU in flight and they hide each other.
//   * a stage = A 96 x 64 + B 64 x 64 16-bit = 20 KB = twenty 1 KB pieces (8 rows x 128 B); every wave fetches five: wave w the
//     pieces 5w .. 5w + 4 (A pieces first, then B), so the vmcnt waits count in fives;
//   * compute waves: 12 ds_read_b128 (A slice, two B slices per 16-deep k slice) and 8 MFMAs per K step.
#define TRAMBA_DSR128I_(OUT, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(OUT) : "v"(ADDR), "i"(OFF) : "memory")

template <typename T, typename TO, int NSTG, bool DUAL = false>
__global__ __launch_bounds__(256) void linear_dma96_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                          const float *__restrict__ bias, const T *__restrict__ res,
                                                          TO *__restrict__ y, long M, int N, int K, int act,
                                                          TO *__restrict__ y_pre = nullptr)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (vector-register asm in a kernel template: see ss2d_scan_dma_kernel)
    constexpr int BM = 96, BN = 64;
    constexpr int A_BYTES = BM * kBK * 2;                    // 12 KB
    constexpr int TILE_BYTES = (BM + BN) * kBK * 2;          // 20 KB
    constexpr int EPI_BYTES = BM * (BN + 4) * 4;
    constexpr int LDS_BYTES = NSTG * TILE_BYTES > EPI_BYTES ? NSTG * TILE_BYTES : EPI_BYTES;
    static_assert(NSTG == 2 || NSTG == 3, "stage offsets must fit the 16-bit offset field of ds_read");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hi = lane >> 5;
    long m0;
    int n0;
    {   // XCD-aware tile order (see linear_tiled_kernel)
        const unsigned nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned q = nblk >> 3, r = nblk & 7, xcd = lin & 7, slot = lin >> 3;
        const unsigned t = xcd * q + (xcd < r ? xcd : r) + slot;
        m0 = (long)(t / gridDim.x) * BM;
        n0 = (int)(t % gridDim.x) * BN;
    }
    const int nk = K / kBK;
    const unsigned rowb = (unsigned)K * 2u;
    const long mrows = M - m0 < BM ? M - m0 : BM;
    const int nrows = N - n0 < BN ? N - n0 : BN;
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(x + m0 * K, (unsigned)mrows * rowb);
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(w + (long)n0 * K, (unsigned)nrows * rowb);
    // piece p = 5 wave + j: A rows 8p .. 8p + 7 for p < 12, B rows 8 (p - 12) .. after that
    unsigned voff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int p = 5 * wave + j;
        const int row = (p < 12 ? p : p - 12) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[j] = (unsigned)row * rowb + (unsigned)c * 16u;
    }
    unsigned char *mine = lds + 5 * wave * 1024;             // my five pieces inside a stage (the B region follows A's 12 KB)
    auto issue = [&](int kt, int stg) {
        const unsigned so = (unsigned)kt * (kBK * 2);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (5 * wave + j < 12)                           // (wave-uniform)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_void *)(mine + stg * TILE_BYTES + j * 1024), 16, voff[j], so,
                                                         0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_void *)(mine + stg * TILE_BYTES + j * 1024), 16, voff[j], so,
                                                         0, 0);
        }
    };
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const bool computes = wave < 3;
    unsigned aad[4], bad0[4], bad1[4];
    {
        const int ra_ = (computes ? wave : 0) * 32 + r32, rb0_ = r32, rb1_ = 32 + r32;
        const unsigned a0 = lbase + (unsigned)(ra_ * 128 + ((hi ^ ((ra_ >> 1) & 7)) * 16));
        const unsigned b0 = lbase + (unsigned)(A_BYTES + rb0_ * 128 + ((hi ^ ((rb0_ >> 1) & 7)) * 16));
        const unsigned b1 = lbase + (unsigned)(A_BYTES + rb1_ * 128 + ((hi ^ ((rb1_ >> 1) & 7)) * 16));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            aad[kk] = a0 ^ (unsigned)(kk * 32);      // (rows are 128-byte aligned: the XOR stays inside the row)
            bad0[kk] = b0 ^ (unsigned)(kk * 32);
            bad1[kk] = b1 ^ (unsigned)(kk * 32);
        }
    }
    acc16_t acc[1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = acc[0][1][r] = 0.f;

    constexpr int DEPTH = NSTG - 1;                  // tiles in flight ahead of the one being multiplied
#pragma unroll
    for (int t = 0; t < DEPTH; ++t)
        if (t < nk) issue(t, t);
    auto kstep = [&](int kt, auto stg_c) {
        constexpr int STG = decltype(stg_c)::value;
        constexpr int S = STG * TILE_BYTES;
        const int later = nk - 1 - kt;
        if (later >= DEPTH - 1) {
            if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (DEPTH 2: the last step)
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + DEPTH < nk) issue(kt + DEPTH, (STG + DEPTH) % NSTG);
        if (computes) {
            frag8_t a[4], b0[4], b1[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                TRAMBA_DSR128I_(a[kk], aad[kk], S);
                TRAMBA_DSR128I_(b0[kk], bad0[kk], S);
                TRAMBA_DSR128I_(b1[kk], bad1[kk], S);
            }
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(a[0]), "+v"(b0[0]), "+v"(b1[0]) : : "memory");
            acc[0][0] = Mfma<T>::run(b0[0], a[0], acc[0][0]);
            acc[0][1] = Mfma<T>::run(b1[0], a[0], acc[0][1]);
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[1]), "+v"(b0[1]), "+v"(b1[1]) : : "memory");
            acc[0][0] = Mfma<T>::run(b0[1], a[1], acc[0][0]);
            acc[0][1] = Mfma<T>::run(b1[1], a[1], acc[0][1]);
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[2]), "+v"(b0[2]), "+v"(b1[2]) : : "memory");
            acc[0][0] = Mfma<T>::run(b0[2], a[2], acc[0][0]);
            acc[0][1] = Mfma<T>::run(b1[2], a[2], acc[0][1]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[3]), "+v"(b0[3]), "+v"(b1[3]) : : "memory");
            acc[0][0] = Mfma<T>::run(b0[3], a[3], acc[0][0]);
            acc[0][1] = Mfma<T>::run(b1[3], a[3], acc[0][1]);
        }
    };
    int kt0 = 0;
    for (; kt0 + NSTG <= nk; kt0 += NSTG) {
        kstep(kt0, std::integral_constant<int, 0>{});
        kstep(kt0 + 1, std::integral_constant<int, 1>{});
        if constexpr (NSTG > 2) kstep(kt0 + 2, std::integral_constant<int, 2>{});
    }
    if (kt0 < nk) kstep(kt0, std::integral_constant<int, 0>{});
    if constexpr (NSTG > 2) {
        if (kt0 + 1 < nk) kstep(kt0 + 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();                                       // every wave has read the last tile: the epilogue reuses the LDS
    tile_epilogue<T, TO, BM, BN, 3>(acc, lds, bias, res, y, M, N, act, m0, n0, nullptr, nullptr, DUAL ? y_pre : nullptr);
#endif
}
#undef TRAMBA_DSR128I_

// Tile / staging choice: see launch_tiled.  tile96(): where a 96x64 tile (3 compute waves + 1 loader wave) lands at or
// under a whole number of rounds of the 256 CUs and 64x64 tiles land just above it (M = 2304, N = 512: 288 -> 192
// workgroups); a measurement form since r02 (tramba_tune_set(TRAMBA_TUNE_GEMM_TILE, 4)).
static bool tile96(long m, int n, int k)
{
    if (m < 96) return false;
    const long t64 = ((m + 63) / 64) * ((n + 63) / 64), t96 = ((m + 95) / 96) * ((n + 63) / 64);
    // rounds of the chip: time ~ rounds x tile height
    const long r64 = (t64 + 255) / 256, r96 = (t96 + 255) / 256;
    return k >= 512 && t64 <= 1024 && r96 * 96 < r64 * 64;
}

// The 96 x 64 LDS-DMA tile (linear_dma96_kernel) is a measurement form: TRAMBA_TUNE_GEMM_TILE 15 runs it wherever it can.
static bool tile96_dma(long m, int n, int k, int tile_tune)
{
    return tile_tune == 15 && m >= 96 && (m + 95) / 96 <= 65535;
}

// ---------------------------------------------------------------------------------------------------------------------
// r04: WEIGHT-STATIONARY form for the tall short-K layers (the 96 x 96 / 48 x 48 stages: M = 9216 .. 73728 rows, K = 128 / 256).
// The tile kernels above treat such a layer as thousands of independent 64 x 64 tiles: every tile pays a prologue round trip,
// re-fetches its 64 x K slice of W, drops its accumulators into an fp32 LDS tile and re-reads them for the store (r03: 51 VALU
// instructions per MFMA, 25 us for M = 36864, N = 512, K = 128 against a 9 us HBM floor).  Here a workgroup is PERSISTENT and
// the weights never move again:
//   * 4 waves own a PANEL of 128 NCW columns (NCW = 4 / 2 / 1 column blocks of 32 per wave, NCW * K <= 512): every wave loads
//     its NCW x K/16 MFMA B-fragments ONCE into registers (<= 128 VGPRs) and keeps them;
//   * the workgroup then walks its list of 32-row tiles (tile = g + i * G: neighbouring workgroups stream neighbouring rows);
//     the 32 x K activation tiles travel L2 -> LDS by `buffer_load ... lds` on a 6- / 4-stage ring (K = 128 / 256), every wave
//     fetching K/64 1 KB pieces per tile, the 16-byte chunk c of row r stored at position c ^ (r & 15) (applied on the
//     global side, undone in the ds_read_b128 addresses: the 16 rows of a read group land on 16 different bank groups);
//   * per tile: K/16 fragment reads (shared by the wave's NCW column blocks), NCW K/16 MFMAs with swapped operands (a lane
//     then owns ONE output row and 4-column groups), and the epilogue STRAIGHT FROM THE ACCUMULATORS: bias / folded-LayerNorm
//     terms from a panel-wide LDS table, activation, optional residual, two packed conversions per 4 columns,
//     v_permlane32_swap pairs so that each lane holds 8 consecutive columns, one 16-byte store per lane -- no fp32 staging
//     tile, no workgroup barrier inside the epilogue;
//   * one s_barrier per tile (tile t is in LDS for everyone; everyone has read tile t - 1), vector-memory waits counted by
//     hand: the stores of the epilogue, the residual loads and the DMA pieces retire in order on ONE counter, so the wait in
//     front of the barrier is "everything but the ops issued after the pieces of tile t" ((NSTG - 1) S + (NSTG - 2)(RL + PPW)
//     in steady state) -- a plain vmcnt(0) would wait for the stores just issued, every tile.  Because the counter retires
//     IN ORDER the pieces of tile t also wait for the stores of tile t - NSTG: the ring is as deep as LDS and the 6-bit
//     counter allow, so that a store has NSTG tile-times to be acknowledged before anything waits behind it.
// LayerNorm folded in: the statistics of a row are summed from the A fragments as they pass (a lane holds half of row
// lane & 31: the two halves meet through one v_permlane32_swap), exact products, fp32 sums.
template <typename T> __device__ __forceinline__ unsigned pack2_(float a, float b);
template <> __device__ __forceinline__ unsigned pack2_<__hip_bfloat16>(float a, float b)
{
    unsigned pk = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
#endif
    return pk;
}
template <> __device__ __forceinline__ unsigned pack2_<__half>(float a, float b)
{
    unsigned pk = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
#endif
    return pk;
}
template <typename T> __device__ __forceinline__ float unpack_lo_(unsigned v);
template <typename T> __device__ __forceinline__ float unpack_hi_(unsigned v);
template <> __device__ __forceinline__ float unpack_lo_<__hip_bfloat16>(unsigned v) { return __builtin_bit_cast(float, v << 16); }
template <> __device__ __forceinline__ float unpack_hi_<__hip_bfloat16>(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
template <> __device__ __forceinline__ float unpack_lo_<__half>(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)); }
template <> __device__ __forceinline__ float unpack_hi_<__half>(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); }

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

#define TRAMBA_DSR128I_(OUT, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(OUT) : "v"(ADDR), "i"(OFF) : "memory")
#define TRAMBA_VMCNT_(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <typename T, int K, int NCW, bool LNIN, bool DUAL, bool RES>
__global__ __launch_bounds__(256, 2) void linear_ws_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                          const float *__restrict__ bias, const T *__restrict__ res,
                                                          T *__restrict__ y, long M, int N, int act, LnIn li,
                                                          T *__restrict__ y_pre, int npanel, int gt)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (vector-register asm in a kernel template: see ss2d_scan_dma_kernel)
    static_assert(K == 128 || K == 256, "row = one or two 256-byte bank rows");
    static_assert(NCW * K <= 512, "the wave's B fragments stay in <= 128 registers");
    static_assert((((K == 128 && NCW <= 2) ? 6 : 4) - 1) * 32 * K * 2 + 512 <= 65535, "the last ring stage inside ds_read's 16-bit offset");
    static_assert((((K == 128 && NCW <= 2) ? 6 : 4) - 1) * (NCW * 2 * (DUAL ? 2 : 1)) +
                          (((K == 128 && NCW <= 2) ? 6 : 4) - 2) * ((RES ? NCW * 4 : 0) + K / 64) <= 63,
                  "the counted vmcnt fits its 6-bit field");
    static_assert(!(DUAL && RES), "a dual-output launch has no residual");
    static_assert(!RES || NCW <= 2, "residual rows are prefetched into registers");
    constexpr int KB = K / 16;                 // MFMA k blocks
    constexpr int RB = K * 2;                  // bytes per activation row
    constexpr int TILE_BYTES = 32 * RB;
    constexpr int NSTG = (K == 128 && NCW <= 2) ? 6 : 4;     // ring depth: the pieces of tile t retire behind the stores of tile t - NSTG (see header)
    constexpr int PPW = K / 64;                // 1 KB DMA pieces per wave and tile
    constexpr int PANEL = 128 * NCW;
    constexpr int S = NCW * 2 * (DUAL ? 2 : 1);   // 16-byte stores per wave and tile
    constexpr int RL = RES ? NCW * 4 : 0;         // 8-byte residual loads per wave and tile
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTG * TILE_BYTES + PANEL * 8];
    typedef __attribute__((address_space(3))) void lds_void;
    float *lbias = reinterpret_cast<float *>(lds + NSTG * TILE_BYTES);
    float *lcs = lbias + PANEL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hi = lane >> 5;
    const int pn = blockIdx.x % npanel, g0 = blockIdx.x / npanel;
    const int n0 = pn * PANEL, n0w = n0 + wv * (NCW * 32);
    const long T32 = (M + 31) / 32;
    const int nt = g0 < T32 ? (int)((T32 - g0 + gt - 1) / gt) : 0;

    // ---- this wave's weights: NCW x KB fragments, loaded once
    frag8_t bf[NCW][KB];
#pragma unroll
    for (int cb = 0; cb < NCW; ++cb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            bf[cb][kb] = *reinterpret_cast<const frag8_t *>(w + (long)(n0w + cb * 32 + r32) * K + kb * 16 + hi * 8);
    for (int i = tid; i < PANEL; i += 256) {
        lbias[i] = bias ? bias[n0 + i] : 0.f;
        lcs[i] = LNIN ? li.colsum[n0 + i] : 0.f;
    }
    __syncthreads();

    // ---- activation tiles by LDS-DMA
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(x, (unsigned)(M * RB));
    unsigned voff[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int lo = (wv * PPW + j) * 1024 + lane * 16;        // my 16 bytes inside the (linear) stage
        const int row = lo / RB, pos = (lo % RB) / 16;
        const int c = (pos & ~15) | ((pos ^ row) & 15);
        voff[j] = (unsigned)(row * RB + c * 16);
    }
    auto issue = [&](int i, int stg) {   // tile number i of this workgroup's list (past the end: rows >= M, zero-filled)
        const long tile = (long)g0 + (long)i * gt;
        const unsigned so = (unsigned)((tile < T32 ? tile : T32) * TILE_BYTES);
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_void *)(lds + stg * TILE_BYTES + (wv * PPW + j) * 1024), 16, voff[j],
                                                     so, 0, 0);
    };
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    unsigned xa[8];
    {
        const unsigned l0 = lbase + (unsigned)(r32 * RB) + (unsigned)((((hi ^ r32) & 15)) << 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[j] = l0 ^ (unsigned)(j << 5);
    }
    const unsigned tba = lbase + (unsigned)(NSTG * TILE_BYTES) + (unsigned)((wv * NCW * 32 + 4 * hi) * 4);   // my bias / colsum floats
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(y, (unsigned)(M * (long)N * 2));
    const __amdgpu_buffer_rsrc_t rsp = make_rsrc(DUAL ? y_pre : y, (unsigned)(M * (long)N * 2));
    const __amdgpu_buffer_rsrc_t rsr = make_rsrc(RES ? res : x, (unsigned)(RES ? M * (long)N * 2 : 0));
    const unsigned yoff = (unsigned)(r32 * N * 2 + (n0w + hi * 8) * 2);    // 16-byte stores: 8 consecutive columns per lane
    const unsigned roff = (unsigned)(r32 * N * 2 + (n0w + hi * 4) * 2);    // 8-byte residual loads: my 4-column groups
    const float invk = 1.f / (float)K;

#pragma unroll
    for (int i = 0; i < NSTG - 1; ++i) issue(i, i);

    auto tile_step = [&](int t, auto stg_c) {
        constexpr int STG = decltype(stg_c)::value;
        // my pieces of tile t have landed; ops issued after them may still be in flight.  Order of a step's vector-memory ops:
        // [RL residual loads][PPW pieces of tile t + NSTG - 1][S stores].  The pieces of tile t were issued NSTG - 1 steps ago:
        // behind them came that step's stores and NSTG - 2 whole steps; in the first NSTG - 1 steps (t == STG) the pieces come
        // from the prologue: the prologue pieces of the later tiles and t whole steps are younger
        if (t >= NSTG - 1) TRAMBA_VMCNT_((NSTG - 1) * S + (NSTG - 2) * (RL + PPW));
        else TRAMBA_VMCNT_((NSTG - 2 - STG) * PPW + STG * (RL + PPW + S));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const long tile = (long)g0 + (long)t * gt;
        const unsigned so_y = (unsigned)(tile * 32 * N * 2);
        v2u_t rv[RES ? NCW * 4 : 1];
        if constexpr (RES) {
#pragma unroll
            for (int cb = 0; cb < NCW; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4"
                                 : "=v"(rv[cb * 4 + g])
                                 : "v"(roff), "s"(rsr), "s"(so_y), "i"((cb * 32 + 8 * g) * 2)
                                 : "memory");
        }
        issue(t + NSTG - 1, (STG + NSTG - 1) % NSTG);

        // ---- fragments + MFMAs, groups of 4 k blocks double-buffered
        acc16_t acc[NCW];
#pragma unroll
        for (int cb = 0; cb < NCW; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
        float s1 = 0.f, s2 = 0.f;
        frag8_t a[2][4];
#define TRAMBA_RDG_(G, BUF)                                                                                        \
    TRAMBA_DSR128I_(a[BUF][0], xa[((G) * 4 + 0) & 7], (((G) * 4 + 0) >> 3) * 256 + STG * TILE_BYTES);               \
    TRAMBA_DSR128I_(a[BUF][1], xa[((G) * 4 + 1) & 7], (((G) * 4 + 1) >> 3) * 256 + STG * TILE_BYTES);               \
    TRAMBA_DSR128I_(a[BUF][2], xa[((G) * 4 + 2) & 7], (((G) * 4 + 2) >> 3) * 256 + STG * TILE_BYTES);               \
    TRAMBA_DSR128I_(a[BUF][3], xa[((G) * 4 + 3) & 7], (((G) * 4 + 3) >> 3) * 256 + STG * TILE_BYTES)
        TRAMBA_RDG_(0, 0);
#pragma unroll
        for (int grp = 0; grp < KB / 4; ++grp) {
            if (grp + 1 < KB / 4) {
                if ((grp & 1) == 0) { TRAMBA_RDG_(grp + 1, 1); } else { TRAMBA_RDG_(grp + 1, 0); }
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[grp & 1][0]), "+v"(a[grp & 1][1]), "+v"(a[grp & 1][2]), "+v"(a[grp & 1][3]) : : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[grp & 1][0]), "+v"(a[grp & 1][1]), "+v"(a[grp & 1][2]), "+v"(a[grp & 1][3]) : : "memory");
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (LNIN) {
                    const v4u_t av = __builtin_bit_cast(v4u_t, a[grp & 1][q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1 = dot2_ones<T>(av[e], s1);
                        s2 = dot2_self<T>(av[e], s2);
                    }
                }
#pragma unroll
                for (int cb = 0; cb < NCW; ++cb) acc[cb] = Mfma<T>::run(bf[cb][grp * 4 + q], a[grp & 1][q], acc[cb]);
            }
            __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise hoists the NEXT group's wait above these MFMAs)
        }
#undef TRAMBA_RDG_
        float mean = 0.f, rstd = 1.f;
        if constexpr (LNIN) {   // the two half rows of row r32 meet: lower half-wave + upper half-wave
            const unsigned b1 = __builtin_bit_cast(unsigned, s1), b2 = __builtin_bit_cast(unsigned, s2);
            const auto p1 = __builtin_amdgcn_permlane32_swap(b1, b1, false, false);
            const auto p2 = __builtin_amdgcn_permlane32_swap(b2, b2, false, false);
            const unsigned p1a = p1[0], p1b = p1[1], p2a = p2[0], p2b = p2[1];
            const float t1 = __builtin_bit_cast(float, p1a) + __builtin_bit_cast(float, p1b);
            const float t2 = __builtin_bit_cast(float, p2a) + __builtin_bit_cast(float, p2b);
            mean = t1 * invk;
            rstd = rsqrtf(fmaxf(t2 * invk - mean * mean, 0.f) + li.eps);
        }
        if constexpr (RES) {   // the residual rows were requested before the DMA pieces of this step
            static_assert(NCW <= 2, "operand count of the wait below");
            if constexpr (NCW == 1)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]) : "n"(PPW) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%8)"
                             : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4 % (RES ? NCW * 4 : 1)]),
                               "+v"(rv[5 % (RES ? NCW * 4 : 1)]), "+v"(rv[6 % (RES ? NCW * 4 : 1)]), "+v"(rv[7 % (RES ? NCW * 4 : 1)])
                             : "n"(PPW)
                             : "memory");
        }
        // ---- epilogue from the accumulators
#pragma unroll
        for (int cb = 0; cb < NCW; ++cb) {
            v4u_t bq[4], cq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // (hand-written: an LDS read hipcc can see is ordered behind every pending LDS-DMA)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[g]) : "v"(tba), "i"((cb * 32 + 8 * g) * 4) : "memory");
                if constexpr (LNIN)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cq[g]) : "v"(tba), "i"(PANEL * 4 + (cb * 32 + 8 * g) * 4) : "memory");
            }
            if constexpr (LNIN)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3]) : : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) : : "memory");
            unsigned pk[4][2], pp[4][2];
            float o[16];
            typedef float v4f_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // (bit-cast the WHOLE vector: hipcc compiles __builtin_bit_cast(float, vec[q]) to element 0)
                const v4f_t bf4 = __builtin_bit_cast(v4f_t, bq[g]);
                v4f_t cf4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (LNIN) cf4 = __builtin_bit_cast(v4f_t, cq[g]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[cb][4 * g + q];
                    if constexpr (LNIN) v = rstd * fmaf(-mean, cf4[q], v);
                    o[4 * g + q] = v + bf4[q];
                }
            }
            if constexpr (DUAL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pp[g][0] = pack2_<T>(o[4 * g], o[4 * g + 1]);
                    pp[g][1] = pack2_<T>(o[4 * g + 2], o[4 * g + 3]);
                }
            }
            // ONE (wave-uniform) branch per 16 values: apply_act's per-element chain of comparisons does not get unswitched
            if (act == TRAMBA_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = geluf_(o[e]);
            } else if (act == TRAMBA_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = siluf_(o[e]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (RES) {
                    const v2u_t r2 = rv[cb * 4 + g];
                    o[4 * g + 0] += unpack_lo_<T>(r2[0]);
                    o[4 * g + 1] += unpack_hi_<T>(r2[0]);
                    o[4 * g + 2] += unpack_lo_<T>(r2[1]);
                    o[4 * g + 3] += unpack_hi_<T>(r2[1]);
                }
                pk[g][0] = pack2_<T>(o[4 * g], o[4 * g + 1]);
                pk[g][1] = pack2_<T>(o[4 * g + 2], o[4 * g + 3]);
            }
            // lanes l and l + 32 hold columns 8g + 0..3 / 8g + 4..7 of the same row: after the swaps the lower lane holds 8
            // consecutive columns of group pair (0, 1) or (2, 3) first half, the upper lane the second half
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const auto w0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
                const auto w1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
                const unsigned a0 = w0[0], b0 = w0[1], a1 = w1[0], b1 = w1[1];
                const v4u_t out = {a0, a1, b0, b1};
                __builtin_amdgcn_raw_buffer_store_b128(out, rsy, yoff + (unsigned)((cb * 32 + pr * 16) * 2), so_y, 0);
                if constexpr (DUAL) {
                    const auto u0 = __builtin_amdgcn_permlane32_swap(pp[2 * pr][0], pp[2 * pr + 1][0], false, false);
                    const auto u1 = __builtin_amdgcn_permlane32_swap(pp[2 * pr][1], pp[2 * pr + 1][1], false, false);
                    const unsigned c0 = u0[0], d0 = u0[1], c1 = u1[0], d1 = u1[1];
                    const v4u_t outp = {c0, c1, d0, d1};
                    __builtin_amdgcn_raw_buffer_store_b128(outp, rsp, yoff + (unsigned)((cb * 32 + pr * 16) * 2), so_y, 0);
                }
            }
        }
    };
    for (int t0 = 0; t0 < nt; t0 += NSTG) {
        tile_step(t0, std::integral_constant<int, 0>{});
        if (t0 + 1 < nt) tile_step(t0 + 1, std::integral_constant<int, 1>{});
        if (t0 + 2 < nt) tile_step(t0 + 2, std::integral_constant<int, 2>{});
        if (t0 + 3 < nt) tile_step(t0 + 3, std::integral_constant<int, 3>{});
        if constexpr (NSTG > 4) {
            if (t0 + 4 < nt) tile_step(t0 + 4, std::integral_constant<int, 4>{});
            if (t0 + 5 < nt) tile_step(t0 + 5, std::integral_constant<int, 5>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pieces requested past the end of the list land before the LDS is released
#endif
}
#undef TRAMBA_DSR128I_
#undef TRAMBA_VMCNT_

// Where the weight-stationary kernel is the library's choice (scripts/bench_gemm_pc.py on MI355X, profiles/r04_gemm_forms.txt; us per
// launch, r03 kernel / producer-consumer / weight-stationary): the K = 128 layers of the 96 x 96 stage with the LayerNorm folded
// in -- M = 36864, N = 512 + GELU 29.1 / 36.0 / 22.1, N = 256 16.8 / 20.1 / 15.2; M = 73728, N = 512 53.2 / 66.7 / 36.8 -- and, from
// batch 8 on, their plain and dual-output forms (M = 73728, N = 512: 49.2 / 45.5 / 40.3 plain, 47.0 / 49.5 / 42.4 dual).  At K = 256
// and on the 48 x 48 stage (M = 9216) the three forms are within 10 % of each other and the tile kernels stay.
static bool ws_rule(long m, int n, int k, int kind /* 0 plain, 1 LayerNorm folded in, 2 dual output */)
{
    if (k != 128 || n < 256) return false;
    return kind == 1 ? m >= 16384 : m >= 65536;
}

// Where the producer / consumer kernel is the library's choice (same measurements; kind as in ws_rule):
//   plain: everywhere (-3 .. -20 %: M = 2304, N = 136, K = 1024 8.0 -> 6.3 us, M = 9216, N = 1024, K = 256 19.6 -> 16.5) except the
//          288-tile K = 2048 launches (12.7 -> 12.9: the r03 kernel stays); 4 stages for K >= 2048 on <= 256 tiles (M = 576,
//          N = 1024, K = 4096: 19.3 -> 16.1);
//   LayerNorm folded in: up to 320 tiles (M = 2304, N = 512, K = 2048 16.3 -> 13.6; M = 576, N = 2048, K = 1024 11.8 -> 10.7); on
//          more tiles the statistics reads cost the multiplier waves more than the loader waves save (10.3 -> 11.3);
//   dual output: everywhere except the >= 4096-tile K <= 256 launches (25.2 -> 26.2).
static int pc_rule(long tiles64, int k, int kind)
{
    if (kind == 1) return tiles64 <= 320 ? 3 : 0;
    if (kind == 2) return (k <= 256 && tiles64 >= 4096) ? 0 : 3;
    if (k >= 2048 && tiles64 <= 256) return 4;
    if (k >= 2048 && tiles64 <= 320) return 0;
    return 3;
}

// Weight-stationary form (linear_ws_kernel, r04): column blocks per wave (4 / 2 / 1), or 0 = not used.  16-bit in and out, K = 128 /
// 256, N a whole number of 128 NCW-column panels, tall M.  TRAMBA_TUNE_GEMM_TILE 19 forces it wherever it can run, 0 = the
// library's rule (scripts/bench_gemm_pc.py, profiles/r04_gemm_ws.txt); every other value keeps it off.
static int ws_ncw(long m, int n, int k, int tile_tune, bool has_res, int kind)
{
    if (tile_tune != 19 && tile_tune != 0) return 0;
    if (k != 128 && k != 256) return 0;
    if ((double)m * n * 2.0 >= 2147483648.0 || (double)(m + 32) * k * 2.0 >= 2147483648.0) return 0;
    int ncw = 256 / k;                         // NCW * K <= 256: ~150 VGPRs = 3 workgroups per CU, half the weight prologue per
                                               // workgroup of the NCW * K = 512 form (measured, profiles/r04_gemm_ws.txt)
    if (tile_tune == 19 && tramba_tune_get(TRAMBA_TUNE_SCAN_W) == 99) ncw = 512 / k;   // (A/B of the wide panel: scripts only)
    if (has_res && ncw > 2) ncw = 2;
    while (ncw >= 1 && n % (128 * ncw)) ncw >>= 1;
    if (ncw < 1) return 0;
    if (tile_tune == 19) return m >= 64 ? ncw : 0;
    return ws_rule(m, n, k, kind) ? ncw : 0;
}

template <typename T, int K, bool LNIN, bool DUAL, bool RES>
static void launch_ws(int ncw, const void *x, const void *w, const float *bias, const void *res, void *y, long m, int n, int act,
                      LnIn li, void *y_pre, hipStream_t s)
{
    const int npanel = n / (128 * ncw);
    const long t32 = (m + 31) / 32;
    // persistent workgroups, as many as are resident at once (256 CUs x 3 at K = 128 with NCW <= 2, x 2 otherwise), the tiles of a panel dealt
    // round-robin: every workgroup walks tiles g, g + G, ... -- an even split, whereas a grid sized by whole rounds of tiles
    // (384 workgroups for 1152 tiles) left half of the CUs with one workgroup and the other half with two
    const long resident = 256L * ((K == 128 && ncw <= 2) ? 3 : 2);     // (by registers and LDS: 51 KB / 67 KB per workgroup)
    long gt = resident / npanel;
    if (gt > t32) gt = t32;
    if (gt < 1) gt = 1;
    dim3 grid((unsigned)(gt * npanel)), block(256);
#define WS_(NCW_)                                                                                                      \
    hipLaunchKernelGGL((linear_ws_kernel<T, K, NCW_, LNIN, DUAL, RES>), grid, block, 0, s, (const T *)x, (const T *)w, bias, \
                       (const T *)res, (T *)y, m, n, act, li, (T *)y_pre, npanel, (int)gt)
    if constexpr (K == 128) {
        if (ncw == 4) { if constexpr (!RES) { WS_(4); } }
        else if (ncw == 2) { WS_(2); }
        else { WS_(1); }
    } else {
        if (ncw == 2) { WS_(2); } else { WS_(1); }
    }
#undef WS_
}

template <typename T, bool LNIN, bool DUAL>
static void launch_ws_k(int ncw, const void *x, const void *w, const float *bias, const void *res, void *y, long m, int n, int k,
                        int act, LnIn li, void *y_pre, hipStream_t s)
{
    if (k == 128) {
        if (res) { if constexpr (!DUAL) launch_ws<T, 128, LNIN, false, true>(ncw, x, w, bias, res, y, m, n, act, li, y_pre, s); }
        else launch_ws<T, 128, LNIN, DUAL, false>(ncw, x, w, bias, res, y, m, n, act, li, y_pre, s);
    } else {
        if (res) { if constexpr (!DUAL) launch_ws<T, 256, LNIN, false, true>(ncw, x, w, bias, res, y, m, n, act, li, y_pre, s); }
        else launch_ws<T, 256, LNIN, DUAL, false>(ncw, x, w, bias, res, y, m, n, act, li, y_pre, s);
    }
}

// Producer / consumer form (linear_pc_kernel, r04): 0 = not used, else its LDS stage count.  TRAMBA_TUNE_GEMM_TILE 16 / 17 force it
// on 3 / 4 stages wherever the LDS-DMA kernel could run, 18 forbids it; the library's own rule (tune 0) is fitted to
// scripts/bench_gemm_pc.py (profiles/r04_gemm_pc.txt).
static int pc_stages(long m, int n, int k, int tile_tune, int kind)
{
    if (tile_tune == 16) return 3;
    if (tile_tune == 17) return 4;
    if (tile_tune != 0 && tile_tune != 19) return 0;
    const long tiles64 = ((m + 63) / 64) * ((n + 63) / 64);
    return pc_rule(tiles64, k, kind);
}

// K up to which a grid of >= 1024 tiles runs linear_dma_kernel on 2 LDS stages (5 workgroups per CU instead of 3):
// scripts/bench_gemm_stages.py, r03 -- -12 % at M = 36864 / 73728, N = 512, K = 128, -3..7 % on the other K <= 256 shapes,
// +5 % at K = 512
constexpr int DMA_SHORT_K = 256;

template <typename T, typename TO, bool CONV = false>
static void launch_tiled(const void *x, const void *w, const float *bias, const void *res, void *y, long m, int n,
                         int k, int act, hipStream_t s, ConvGeom cg = ConvGeom{0, 0, 0, 0, 0},
                         const void *x2 = nullptr, int k1 = 0)
{
    const long big = ((m + 127) / 128) * ((n + 127) / 128);
    // lean kernel: whole 64-deep K steps, and a 64-row operand panel within 32-bit byte offsets
    const bool lean_ok = k % 64 == 0 && (double)k * 2.0 * 128.0 < 2147483648.0;
    const long tiles64 = ((m + 63) / 64) * ((n + 63) / 64);
    // Plain GEMMs with K % 64 == 0 (every 1x1 convolution of the model): the LDS-DMA staged 64x64 kernel.  Measured back to
    // back in one process on the model's shapes at batch 4 and 8 (scripts/bench_gemm_tiles.py): 14 % / 9 % less time per
    // pass than the register-staged 64x64 kernel (2-stage ring, 82 VGPRs), faster on every shape (19.4 -> 13.0 us at
    // M = 2304, N = 512, K = 2048).  Register-staged forms for comparison, through TRAMBA_TUNE_GEMM_TILE: 1 = 64x64 (the
    // default until r02; still used for the two-source A operand), 2 = 128x128 (1-2 resident blocks: 1.3-2x slower),
    // 3 = 128x64, 4 = 96x64 where it saves a round of the chip, 5 = 64x64 on a 4-stage ring.
    const int tile_tune = tramba_tune_get(TRAMBA_TUNE_GEMM_TILE);
    // (4 stages = 3 tiles in flight where at most one block lands on a CU and K is long: nothing else hides the load latency
    //  there -- M = 576, N = 1024, K = 4096: 27 -> 22 us with operands that are not cache-resident; 8 stages = one block per
    //  CU by LDS, was measured slower on every shape and is not built)
    const bool dma_deep = tile_tune == 7 || ((tile_tune == 0 || tile_tune == 18 || tile_tune == 19) && tiles64 <= 256 && k >= 1024);
    if constexpr (!CONV && std::is_same<T, TO>::value) {
        const int ncw = (lean_ok && !x2) ? ws_ncw(m, n, k, tile_tune, res != nullptr, 0) : 0;
        if (ncw && (act == TRAMBA_ACT_NONE || act == TRAMBA_ACT_GELU || act == TRAMBA_ACT_SILU)) {
            launch_ws_k<T, false, false>(ncw, x, w, bias, res, y, m, n, k, act, LnIn{nullptr, 0.f}, nullptr, s);
            return;
        }
    }
    if (!CONV && lean_ok && !x2 && tile96_dma(m, n, k, tile_tune)) {
        dim3 grid((n + 63) / 64, (unsigned)((m + 95) / 96)), block(256);
        hipLaunchKernelGGL((linear_dma96_kernel<T, TO, 3>), grid, block, 0, s, (const T *)x, (const T *)w, bias, (const T *)res,
                           (TO *)y, m, n, k, act);
    } else if (!CONV && lean_ok && !x2 && (tile_tune == 0 || tile_tune == 6 || tile_tune == 7 || tile_tune == 13 || tile_tune == 14 ||
                                           tile_tune == 15 || tile_tune == 16 || tile_tune == 17 || tile_tune == 18 || tile_tune == 19)) {
        dim3 grid((n + 63) / 64, (unsigned)((m + 63) / 64)), block(256);
        // short K (<= 4 steps) on a grid of many tiles: the launch is prologue + epilogue, and what it needs is workgroups in
        // flight -- 2 stages = 32 KB of LDS = 5 per CU instead of 3 (TRAMBA_TUNE_GEMM_TILE 13 forces it, 14 forbids it)
        const bool dma_short = tile_tune == 13 || ((tile_tune == 0 || tile_tune == 18 || tile_tune == 19) && DMA_SHORT_K > 0 && k <= DMA_SHORT_K && tiles64 >= 1024);
        const int pc = pc_stages(m, n, k, tile_tune, 0);
        if (pc == 3)
            hipLaunchKernelGGL((linear_pc_kernel<T, TO, 3>), grid, dim3(512), 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)res, (TO *)y, m, n, k, act);
        else if (pc == 4)
            hipLaunchKernelGGL((linear_pc_kernel<T, TO, 4>), grid, dim3(512), 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)res, (TO *)y, m, n, k, act);
        else if (dma_short)
            hipLaunchKernelGGL((linear_dma_kernel<T, TO, 2>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)res, (TO *)y, m, n, k, act);
        else if (dma_deep)
            hipLaunchKernelGGL((linear_dma_kernel<T, TO, 4>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)res, (TO *)y, m, n, k, act);
        else
            hipLaunchKernelGGL((linear_dma_kernel<T, TO, 3>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)res, (TO *)y, m, n, k, act);
    } else if (!CONV && lean_ok && tile_tune == 2) {
        dim3 grid((n + 127) / 128, (unsigned)((m + 127) / 128)), block(256);
        hipLaunchKernelGGL((linear_lean_kernel<T, TO, 128, 128, 1>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                           (const T *)res, (TO *)y, m, n, k, act, (const T *)x2, k1);
    } else if (!CONV && lean_ok && tile_tune == 3) {
        dim3 grid((n + 63) / 64, (unsigned)((m + 127) / 128)), block(256);
        hipLaunchKernelGGL((linear_lean_kernel<T, TO, 128, 64, 1>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                           (const T *)res, (TO *)y, m, n, k, act, (const T *)x2, k1);
    } else if (!CONV && lean_ok && tile_tune == 4 && tile96(m, n, k)) {
        // 96x64 tiles, 3 compute waves + 1 loader wave: the tile count drops under one round of the chip
        dim3 grid((n + 63) / 64, (unsigned)((m + 95) / 96)), block(256);
        hipLaunchKernelGGL((linear_lean_kernel<T, TO, 96, 64, 1, false, 3>), grid, block, 0, s, (const T *)x, (const T *)w,
                           bias, (const T *)res, (TO *)y, m, n, k, act, (const T *)x2, k1);
    } else if (!CONV && lean_ok && tile_tune == 5) {
        // 4 stages (168 VGPRs): long K on a grid of about one block per CU
        dim3 grid((n + 63) / 64, (unsigned)((m + 63) / 64)), block(256);
        hipLaunchKernelGGL((linear_lean_kernel<T, TO, 64, 64, 3>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                           (const T *)res, (TO *)y, m, n, k, act, (const T *)x2, k1);
    } else if (!CONV && lean_ok) {
        dim3 grid((n + 63) / 64, (unsigned)((m + 63) / 64)), block(256);
        hipLaunchKernelGGL((linear_lean_kernel<T, TO, 64, 64, 1>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                           (const T *)res, (TO *)y, m, n, k, act, (const T *)x2, k1);
    } else if (!x2 && big >= 2048 && k >= 1024) {
        dim3 grid((n + 127) / 128, (unsigned)((m + 127) / 128)), block(256);
        hipLaunchKernelGGL((linear_tiled_kernel<T, TO, 128, 128, 2, CONV>), grid, block, 0, s, (const T *)x,
                           (const T *)w, bias, (const T *)res, (TO *)y, m, n, k, act, cg);
    } else if (k <= 128) {  // 1-2 K steps: a 2-stage ring, no padded dummy steps
        dim3 grid((n + 63) / 64, (unsigned)((m + 63) / 64)), block(256);
        hipLaunchKernelGGL((linear_tiled_kernel<T, TO, 64, 64, 1, CONV>), grid, block, 0, s, (const T *)x,
                           (const T *)w, bias, (const T *)res, (TO *)y, m, n, k, act, cg);
    } else {
        dim3 grid((n + 63) / 64, (unsigned)((m + 63) / 64)), block(256);
        hipLaunchKernelGGL((linear_tiled_kernel<T, TO, 64, 64, 3, CONV>), grid, block, 0, s, (const T *)x,
                           (const T *)w, bias, (const T *)res, (TO *)y, m, n, k, act, cg);
    }
}

// fp32: v_mfma_f32_32x32x2_f32, A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
template <typename TO>
__global__ __launch_bounds__(256) void linear32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, const float *__restrict__ res,
                                                      TO *__restrict__ y, long M, int N, int K, int act)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.y * 128 + (wave >> 1) * 64;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (m0 >= M || n0 >= N) return;
    acc16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r32 = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += 2) {
        const int k = k0 + kh;
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long row = m0 + i * 32 + r32;
            a[i] = (row < M && k < K) ? x[row * K + k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + j * 32 + r32;
            b[j] = (col < N && k < K) ? w[(long)col * K + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<float, TO>(acc[i][j], y, bias, res, m0 + i * 32, n0 + j * 32, M, N, act, lane);
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_linear_cl(const void *x, const void *w, const float *bias, const void *residual,
                                void *y, int64_t m, int n, int k, int act, int dtype, int out_dtype,
                                void *stream)
{
    TRAMBA_CHECK(x && w && y, "linear_cl: null tensor");
    TRAMBA_CHECK(m > 0 && n > 0 && k > 0, "linear_cl: empty shape");
    TRAMBA_CHECK(out_dtype == dtype || out_dtype == TRAMBA_F32, "linear_cl: out dtype must be dtype or f32");
    TRAMBA_CHECK(act >= TRAMBA_ACT_NONE && act <= TRAMBA_ACT_GELU_GRAD_MUL, "linear_cl: unknown activation %d", act);
    TRAMBA_CHECK(residual || (act != TRAMBA_ACT_SIGMOID_GATE && act != TRAMBA_ACT_GELU_GRAD_MUL),
                 "linear_cl: activation %d combines with a `residual` operand, none given", act);
    const long gy = (m + 127) / 128, gx = (n + 127) / 128;
    TRAMBA_CHECK(gy <= 65535, "linear_cl: M=%ld exceeds grid limits", (long)m);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_GEMM, s, 2.0 * (double)m * n * k);
    dim3 grid((unsigned)gx, (unsigned)gy), block(256);
    const int vec = (k % 8 == 0) && aligned16(x) && aligned16(w);
    const bool tiled = vec && dtype != TRAMBA_F32 && aligned16(y) && (residual == nullptr || aligned16(residual)) &&
                       (m + 63) / 64 <= 65535;
    if (tiled) {
        if (dtype == TRAMBA_BF16) {
            if (out_dtype == TRAMBA_F32) launch_tiled<__hip_bfloat16, float>(x, w, bias, residual, y, m, n, k, act, s);
            else launch_tiled<__hip_bfloat16, __hip_bfloat16>(x, w, bias, residual, y, m, n, k, act, s);
        } else {
            if (out_dtype == TRAMBA_F32) launch_tiled<__half, float>(x, w, bias, residual, y, m, n, k, act, s);
            else launch_tiled<__half, __half>(x, w, bias, residual, y, m, n, k, act, s);
        }
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    if (dtype == TRAMBA_F32) {
        hipLaunchKernelGGL(linear32_kernel<float>, grid, block, 0, s, (const float *)x, (const float *)w, bias,
                           (const float *)residual, (float *)y, (long)m, n, k, act);
    } else if (dtype == TRAMBA_BF16) {
        using T = __hip_bfloat16;
        if (out_dtype == TRAMBA_F32)
            hipLaunchKernelGGL((linear16_kernel<T, float>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (float *)y, (long)m, n, k, act, vec);
        else
            hipLaunchKernelGGL((linear16_kernel<T, T>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (T *)y, (long)m, n, k, act, vec);
    } else if (dtype == TRAMBA_F16) {
        using T = __half;
        if (out_dtype == TRAMBA_F32)
            hipLaunchKernelGGL((linear16_kernel<T, float>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (float *)y, (long)m, n, k, act, vec);
        else
            hipLaunchKernelGGL((linear16_kernel<T, T>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (T *)y, (long)m, n, k, act, vec);
    } else {
        set_error("linear_cl: bad dtype %d", dtype);
        return TRAMBA_ERR_ARG;
    }
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_linear_ln_cl(const void *x, const void *w_folded, const float *colsum, const float *bias,
                                   const void *residual, void *y, int64_t m, int n, int k, float eps, int act, int dtype,
                                   int out_dtype, void *stream)
{
    TRAMBA_CHECK(x && w_folded && colsum && y, "linear_ln_cl: null tensor");
    TRAMBA_CHECK(m > 0 && n > 0 && k > 0, "linear_ln_cl: empty shape");
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "linear_ln_cl: 16-bit activations only");
    TRAMBA_CHECK(out_dtype == dtype || out_dtype == TRAMBA_F32, "linear_ln_cl: output must be the input dtype or f32");
    TRAMBA_CHECK(k % 64 == 0 && k <= 2048 && n % 8 == 0, "linear_ln_cl: needs K %% 64 == 0, K <= 2048, N %% 8 == 0");
    TRAMBA_CHECK(aligned16(x) && aligned16(w_folded) && aligned16(y) && aligned16(colsum) && (!bias || aligned16(bias)) &&
                     (!residual || aligned16(residual)),
                 "linear_ln_cl: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_GEMM, s, 2.0 * (double)m * n * k);
    const long tiles64 = ((m + 63) / 64) * ((n + 63) / 64);
    const int tile_tune = tramba_tune_get(TRAMBA_TUNE_GEMM_TILE);
    const bool deep = tile_tune == 5;                    // 4-stage register ring: measurement only (see launch_tiled)
    const bool dma = tile_tune == 0 || tile_tune == 6 || tile_tune == 15 || tile_tune >= 16;   // the LDS-DMA staged kernel (default)
    const int pc = pc_stages(m, n, k, tile_tune, 1);
    dim3 grid((n + 63) / 64, (unsigned)((m + 63) / 64)), block(256);
    const LnIn li{colsum, eps};
    const bool dma_short = dma && (tile_tune == 0 || tile_tune >= 18) && k <= DMA_SHORT_K && ((m + 63) / 64) * ((n + 63) / 64) >= 1024;   // (launch_tiled)
#define LNIN_(T, TO)                                                                                                     \
    if (pc == 3)                                                                                                         \
        hipLaunchKernelGGL((linear_pc_kernel<T, TO, 3, true>), grid, dim3(512), 0, s, (const T *)x, (const T *)w_folded, bias, \
                           (const T *)residual, (TO *)y, m, n, k, act, li);                                              \
    else if (pc == 4)                                                                                                    \
        hipLaunchKernelGGL((linear_pc_kernel<T, TO, 4, true>), grid, dim3(512), 0, s, (const T *)x, (const T *)w_folded, bias, \
                           (const T *)residual, (TO *)y, m, n, k, act, li);                                              \
    else if (dma_short)                                                                                                       \
        hipLaunchKernelGGL((linear_dma_kernel<T, TO, 2, true>), grid, block, 0, s, (const T *)x, (const T *)w_folded, bias, \
                           (const T *)residual, (TO *)y, m, n, k, act, li);                                              \
    else if (dma)                                                                                                        \
        hipLaunchKernelGGL((linear_dma_kernel<T, TO, 3, true>), grid, block, 0, s, (const T *)x, (const T *)w_folded, bias, \
                           (const T *)residual, (TO *)y, m, n, k, act, li);                                              \
    else if (deep)                                                                                                            \
        hipLaunchKernelGGL((linear_lean_kernel<T, TO, 64, 64, 3, false, 2, true>), grid, block, 0, s, (const T *)x,       \
                           (const T *)w_folded, bias, (const T *)residual, (TO *)y, m, n, k, act, (const T *)nullptr, 0,  \
                           LnHead{}, li);                                                                                \
    else                                                                                                                 \
        hipLaunchKernelGGL((linear_lean_kernel<T, TO, 64, 64, 1, false, 2, true>), grid, block, 0, s, (const T *)x,       \
                           (const T *)w_folded, bias, (const T *)residual, (TO *)y, m, n, k, act, (const T *)nullptr, 0,  \
                           LnHead{}, li)
    const int ncw = out_dtype == dtype && (act == TRAMBA_ACT_NONE || act == TRAMBA_ACT_GELU || act == TRAMBA_ACT_SILU)
                        ? ws_ncw(m, n, k, tile_tune, residual != nullptr, 1) : 0;
    if (ncw) {
        if (dtype == TRAMBA_BF16) launch_ws_k<__hip_bfloat16, true, false>(ncw, x, w_folded, bias, residual, y, m, n, k, act, li, nullptr, s);
        else launch_ws_k<__half, true, false>(ncw, x, w_folded, bias, residual, y, m, n, k, act, li, nullptr, s);
    } else if (dtype == TRAMBA_BF16) {
        if (out_dtype == TRAMBA_F32) { LNIN_(__hip_bfloat16, float); } else { LNIN_(__hip_bfloat16, __hip_bfloat16); }
    } else {
        if (out_dtype == TRAMBA_F32) { LNIN_(__half, float); } else { LNIN_(__half, __half); }
    }
#undef LNIN_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_linear2_cl(const void *x1, const void *x2, int k1, const void *w, const float *bias,
                                 const void *residual, void *y, int64_t m, int n, int k, int act, int dtype,
                                 int out_dtype, void *stream)
{
    TRAMBA_CHECK(x1 && x2 && w && y, "linear2_cl: null tensor");
    TRAMBA_CHECK(m > 0 && n > 0 && k > 0 && k1 > 0 && k1 < k, "linear2_cl: empty shape");
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "linear2_cl: 16-bit dtypes only");
    TRAMBA_CHECK(out_dtype == dtype || out_dtype == TRAMBA_F32, "linear2_cl: out dtype must be dtype or f32");
    TRAMBA_CHECK(k1 % 64 == 0 && (k - k1) % 64 == 0, "linear2_cl: both K parts must be multiples of 64 (%d + %d)", k1, k - k1);
    TRAMBA_CHECK(aligned16(x1) && aligned16(x2) && aligned16(w) && aligned16(y) && (residual == nullptr || aligned16(residual)),
                 "linear2_cl: tensors must be 16-byte aligned");
    TRAMBA_CHECK((m + 63) / 64 <= 65535 && (double)k * 2.0 * 128.0 < 2147483648.0, "linear2_cl: shape exceeds this build's limits");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_GEMM, s, 2.0 * (double)m * n * k);
    const ConvGeom nocg{0, 0, 0, 0, 0};
    if (dtype == TRAMBA_BF16) {
        if (out_dtype == TRAMBA_F32) launch_tiled<__hip_bfloat16, float>(x1, w, bias, residual, y, m, n, k, act, s, nocg, x2, k1);
        else launch_tiled<__hip_bfloat16, __hip_bfloat16>(x1, w, bias, residual, y, m, n, k, act, s, nocg, x2, k1);
    } else {
        if (out_dtype == TRAMBA_F32) launch_tiled<__half, float>(x1, w, bias, residual, y, m, n, k, act, s, nocg, x2, k1);
        else launch_tiled<__half, __half>(x1, w, bias, residual, y, m, n, k, act, s, nocg, x2, k1);
    }
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_expand_norm_head_cl(const void *x, const void *w, const float *ln_w, const float *ln_b,
                                          const float *head_w, float head_b, float *y, int batch, int h, int wd, int cin,
                                          int p, float eps, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && ln_w && ln_b && head_w && y, "expand_norm_head_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && cin > 0 && p > 0, "expand_norm_head_cl: empty shape");
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "expand_norm_head_cl: 16-bit dtypes only");
    TRAMBA_CHECK(cin % 64 == 0, "expand_norm_head_cl: Cin=%d must be a multiple of 64", cin);
    TRAMBA_CHECK(aligned16(x) && aligned16(w), "expand_norm_head_cl: tensors must be 16-byte aligned");
    const long m = (long)batch * h * wd;
    const int n = p * p * 128;   // the fused epilogue is written for 128-channel groups (Tramba-V / -R final stage)
    TRAMBA_CHECK(m < 2147483647L && (m + 63) / 64 <= 65535, "expand_norm_head_cl: too many pixels");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_GEMM, s, 2.0 * (double)m * n * cin);
    const LnHead hd{ln_w, ln_b, head_w, head_b, eps, h, wd, p, y};
    dim3 grid((unsigned)(n / 128), (unsigned)((m + 63) / 64)), block(256);
    if (dtype == TRAMBA_BF16) {
        using T = __hip_bfloat16;
        hipLaunchKernelGGL((linear_lean_kernel<T, T, 64, 128, 1, true>), grid, block, 0, s, (const T *)x, (const T *)w,
                           (const float *)nullptr, (const T *)nullptr, (T *)nullptr, m, n, cin, 0, (const T *)nullptr, 0, hd);
    } else {
        using T = __half;
        hipLaunchKernelGGL((linear_lean_kernel<T, T, 64, 128, 1, true>), grid, block, 0, s, (const T *)x, (const T *)w,
                           (const float *)nullptr, (const T *)nullptr, (T *)nullptr, m, n, cin, 0, (const T *)nullptr, 0, hd);
    }
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_conv3x3s2_cl(const void *x, const void *w, const float *bias, void *y, int batch, int hin,
                                   int win, int cin, int cout, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && y, "conv3x3s2_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && hin > 0 && win > 0 && cin > 0 && cout > 0, "conv3x3s2_cl: empty shape");
    TRAMBA_CHECK(cin % 64 == 0, "conv3x3s2_cl: Cin=%d must be a multiple of 64", cin);
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "conv3x3s2_cl: bf16/f16 only (fp32 convs run on MIOpen)");
    TRAMBA_CHECK(aligned16(x) && aligned16(w) && aligned16(y), "conv3x3s2_cl: tensors must be 16-byte aligned");
    ConvGeom cg{hin, win, cin, (hin + 1) / 2, (win + 1) / 2};
    const long m = (long)batch * cg.hout * cg.wout;
    TRAMBA_CHECK((m + 63) / 64 <= 65535, "conv3x3s2_cl: too many output pixels");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TRAMBA_BF16)
        launch_tiled<__hip_bfloat16, __hip_bfloat16, true>(x, w, bias, nullptr, y, m, cout, 9 * cin, TRAMBA_ACT_NONE, s, cg);
    else
        launch_tiled<__half, __half, true>(x, w, bias, nullptr, y, m, cout, 9 * cin, TRAMBA_ACT_NONE, s, cg);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_linear_dual_cl(const void *x, const void *w, const float *bias, void *y_pre, void *y_act, int64_t m,
                                     int n, int k, int act, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && y_pre && y_act, "linear_dual_cl: null tensor");
    TRAMBA_CHECK(m > 0 && n > 0 && k > 0, "linear_dual_cl: empty shape");
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "linear_dual_cl: 16-bit activations only");
    TRAMBA_CHECK(act == TRAMBA_ACT_GELU || act == TRAMBA_ACT_SILU, "linear_dual_cl: act must be GELU or SiLU");
    TRAMBA_CHECK(k % 64 == 0 && n % 8 == 0 && (double)k * 2.0 * 128.0 < 2147483648.0 && (m + 63) / 64 <= 65535,
                 "linear_dual_cl: needs K %% 64 == 0, N %% 8 == 0");
    TRAMBA_CHECK(aligned16(x) && aligned16(w) && aligned16(y_pre) && aligned16(y_act), "linear_dual_cl: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_GEMM, s, 2.0 * (double)m * n * k);
    const bool dma96 = tile96_dma(m, n, k, tramba_tune_get(TRAMBA_TUNE_GEMM_TILE));
    dim3 grid((n + 63) / 64, (unsigned)(dma96 ? (m + 95) / 96 : (m + 63) / 64)), block(256);
    const int tune_ = tramba_tune_get(TRAMBA_TUNE_GEMM_TILE);
    const bool dma_short = (tune_ == 0 || tune_ >= 18) && k <= DMA_SHORT_K && ((m + 63) / 64) * ((n + 63) / 64) >= 1024;   // (launch_tiled)
    const int pc = pc_stages(m, n, k, tune_, 2);
#define DUAL_(T, S_)                                                                                                    \
    hipLaunchKernelGGL((linear_dma_kernel<T, T, S_, false, true>), grid, block, 0, s, (const T *)x, (const T *)w, bias,   \
                       (const T *)nullptr, (T *)y_act, m, n, k, act, LnIn{nullptr, 0.f}, (T *)y_pre)
#define DUAL96_(T)                                                                                                      \
    hipLaunchKernelGGL((linear_dma96_kernel<T, T, 3, true>), grid, block, 0, s, (const T *)x, (const T *)w, bias,         \
                       (const T *)nullptr, (T *)y_act, m, n, k, act, (T *)y_pre)
#define DUALPC_(T, S_)                                                                                                  \
    hipLaunchKernelGGL((linear_pc_kernel<T, T, S_, false, true>), grid, dim3(512), 0, s, (const T *)x, (const T *)w, bias, \
                       (const T *)nullptr, (T *)y_act, m, n, k, act, LnIn{nullptr, 0.f}, (T *)y_pre)
    const int ncw = ws_ncw(m, n, k, tune_, false, 2);
    if (ncw) {
        if (dtype == TRAMBA_BF16) launch_ws_k<__hip_bfloat16, false, true>(ncw, x, w, bias, nullptr, y_act, m, n, k, act, LnIn{nullptr, 0.f}, y_pre, s);
        else launch_ws_k<__half, false, true>(ncw, x, w, bias, nullptr, y_act, m, n, k, act, LnIn{nullptr, 0.f}, y_pre, s);
    } else if (dtype == TRAMBA_BF16) {
        if (dma96) DUAL96_(__hip_bfloat16); else if (pc == 3) DUALPC_(__hip_bfloat16, 3); else if (pc == 4) DUALPC_(__hip_bfloat16, 4);
        else if (dma_short) DUAL_(__hip_bfloat16, 2); else DUAL_(__hip_bfloat16, 3);
    } else {
        if (dma96) DUAL96_(__half); else if (pc == 3) DUALPC_(__half, 3); else if (pc == 4) DUALPC_(__half, 4);
        else if (dma_short) DUAL_(__half, 2); else DUAL_(__half, 3);
    }
#undef DUALPC_
#undef DUAL_
#undef DUAL96_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
