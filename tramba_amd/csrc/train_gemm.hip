// Training-path GEMMs that the forward kernels (linear.hip) do not cover.
//
// 1. Weight gradient of a 1x1 convolution / Linear2d (Models/modules.py:10-19 under autograd):
//        gw[n][k] = sum over tokens t of gy[t][n] * x[t][k]            ("TN": both operands are token-major)
//    The contraction runs over the SLOW index of both operands (M = B*H*W tokens, up to 73 728 at batch 8) while the
//    output is at most 1024 x 4096: a plain GEMM has a handful of output tiles for 256 CUs.  So the token range is split
//    over grid.z, every block accumulates a 128 x 128 tile of gw over its token chunk on the matrix cores and writes an
//    fp32 partial slab; a second small kernel sums the slabs in a fixed order (deterministic, no atomics).
//    An MFMA fragment wants 8 consecutive k (= tokens) of one channel, the tensors are channel-contiguous: the 32-token
//    tiles are staged in LDS as they lie in memory ([token][channel], coalesced 16-byte rows) and read back TRANSPOSED
//    with ds_read_b64_tr_b16 (4 tokens x 16 channels per 16-lane group) -- no transposed copy of an activation is ever
//    written.  Row stride 320 B (= 16 banks past a multiple of 64): the four token rows and the two channel halves a
//    32-lane group touches fall on disjoint banks.
//    The bias gradient (column sums of gy) rides along: the threads that stage gy add what they load.
//    `groups` / `nbatch` + strides let one launch serve the per-direction contractions of the SS2D backward, whose
//    operands are (B, K, L, C) tensors: group g, batch b, token t sits at base + b*bs + g*gs + t*ld.
// 2. rows_gemm: y[z][t][0..N) = x[z][t][:] . w[z % groups][n][:], N <= 64, fp32 rows written with a row stride -- the
//    dt_rank projection of the SS2D backward (graw (B,K,L,D) x dt_w (K,D,R) -> the first R floats of every x_dbl-gradient
//    row), one launch for all (b, k).
// 3. shadow_cast_multi: the 16-bit copies of the fp32 master weights that the next step's GEMMs read -- W (N,K) for the
//    forward and W^T (K,N) for the input gradient (dx = gy @ W runs on the forward kernel with the transposed weight) --
//    written for EVERY 2-D weight of a model by one launch after the optimizer step, from a device-resident table of
//    tensors (64 x 64 tiles, tile -> tensor by binary search).  Replaces ~230 cast and ~230 transpose launches per step.
// 4. slab_sum: out[i] = sum_s part[s][i] in a fixed order -- the partial sums of the LayerNorm / depth-wise / scan
//    parameter gradients.
#include <type_traits>

#include "common.h"

namespace tramba {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short frag8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float acc16 __attribute__((ext_vector_type(16)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ acc16 mfma16(frag8s a, frag8s b, acc16 c);
template <> __device__ __forceinline__ acc16 mfma16<__hip_bfloat16>(frag8s a, frag8s b, acc16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ acc16 mfma16<__half>(frag8s a, frag8s b, acc16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}

constexpr int kWgTok = 32;            // tokens per step
constexpr int kWgTile = 128;          // output tile edge (channels of gy / channels of x)
constexpr int kWgRow = 320;           // LDS bytes per staged token row (256 data + 64 pad)

struct WgradArgs {
    const void *gy, *x;
    float *part;                      // (groups, nsplit*nbatch, N*K + N) fp32 slabs; the last N floats = column sums of gy
    long gy_bs, gy_gs, x_bs, x_gs;    // element strides of batch / group
    int gy_ld, x_ld;                  // elements between consecutive tokens
    int M, N, K;                      // tokens per (group, batch), channels of gy, channels of x
    int nbatch, nsplit, mchunk;       // grid.z = groups * nbatch * nsplit; mchunk = tokens per split (multiple of 32)
    int want_bias;
};

template <typename T>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(WgradArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kWgTok * kWgRow];   // [buffer][gy | x]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv >> 1, wk = wv & 1;                      // wave tile: 64 (n) x 64 (k)
    // XCD-aware work order (xcd_work_item, common.h): the tiles of ONE token chunk re-read the same gy / x rows -- K / 128
    // and N / 128 times.  In launch order consecutive workgroups land on different XCDs, so every chunk was fetched by all
    // eight L2s and re-read through the fabric (the big layers moved ~150 MB per launch at ~5.4 TB/s: that, not the matrix
    // cores, set their 350 TFLOP/s).  Remapped, an XCD walks k tile -> n tile -> chunk: a chunk's tiles run side by side on
    // one XCD and its operand rows (<= 2.5 MB) are served by that XCD's L2.
    unsigned bx_, by_, bz_;
    xcd_work_item(bx_, by_, bz_);
    const int bx = (int)bx_, by = (int)by_, bz = (int)bz_;
    const int n0 = by * kWgTile, k0 = bx * kWgTile;
    const int per = a.nbatch * a.nsplit;
    const int g = bz / per, zz = bz % per;
    const int b = zz / a.nsplit, sp = zz % a.nsplit;
    const int t0 = sp * a.mchunk;
    const int t1 = t0 + a.mchunk < a.M ? t0 + a.mchunk : a.M;
    const T *gyb = (const T *)a.gy + (long)b * a.gy_bs + (long)g * a.gy_gs;
    const T *xb = (const T *)a.x + (long)b * a.x_bs + (long)g * a.x_gs;
    // descriptors over the whole (group, batch) operand: row t at t*ld elements
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gyb, (unsigned)(((long)(a.M - 1) * a.gy_ld + a.N) * (long)sizeof(T)));
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)(((long)(a.M - 1) * a.x_ld + a.K) * (long)sizeof(T)));

    // staging: a 32 x 128 tile = 512 chunks of 16 bytes; thread t moves chunks t and t + 256: token tid/16 (+16),
    // channels 8*(tid%16) .. +7
    const int stok = tid >> 4, sch = (tid & 15) * 8;
    const bool gcol = n0 + sch < a.N, xcol = k0 + sch < a.K;     // N, K are multiples of 8: a chunk is in or out whole
    auto goff = [&](int tok, bool col, int c0, int ld) -> unsigned {
        return (col && tok < t1) ? (unsigned)(((long)tok * ld + c0) * (long)sizeof(T)) : kOutOfRange;
    };
    v4u32 rgy[2], rxx[2];
    auto fetch = [&](int ts) {   // ts = first token of the step
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int tok = ts + stok + 16 * i;
            rgy[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, goff(tok, gcol, n0 + sch, a.gy_ld), 0, 0);
            rxx[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, goff(tok, xcol, k0 + sch, a.x_ld), 0, 0);
        }
    };
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool do_bias = (a.want_bias & 1) && bx == 0;
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = (stok + 16 * i) * kWgRow + sch * 2;
            *reinterpret_cast<v4u32 *>(&lds[buf][0][off]) = rgy[i];
            *reinterpret_cast<v4u32 *>(&lds[buf][1][off]) = rxx[i];
            if (do_bias) {
                const Pack<T, 8> pk = __builtin_bit_cast(Pack<T, 8>, rgy[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[j] += Cvt<T>::to_f(pk.v[j]);
            }
        }
    };

    // transposed fragment reads.  16-lane group gq = lane>>4: channels 16*(gq&1) .. +15 of a 32-channel block, tokens
    // 8*(gq>>1) + 4u .. +3; lane 4q+p of the group supplies the address of token row q, channels 4p .. 4p+3 and receives
    // channel (lane&15), the 4 tokens
    const int li = lane & 15, q = li >> 2, p = li & 3, gq = lane >> 4;
    const unsigned trbase = (unsigned)((8 * (gq >> 1) + q) * kWgRow + (16 * (gq & 1) + 4 * p) * 2);
    auto frag = [&](const unsigned char *tile, int ks, int cblk) -> frag8s {   // cblk: first channel of the 32-block
        const unsigned char *pa = tile + trbase + ks * 16 * kWgRow + cblk * 2;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3))) *)(pa));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3))) *)(pa + 4 * kWgRow));
        frag8s f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };

    acc16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nstep = (t1 - t0 + kWgTok - 1) / kWgTok;
    if (nstep > 0) {
        fetch(t0);
        stash(0);
    }
    __syncthreads();
    for (int s = 0; s < nstep; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstep) fetch(t0 + (s + 1) * kWgTok);      // block-uniform
#pragma unroll
        for (int ks = 0; ks < kWgTok / 16; ++ks) {
            frag8s fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = frag(&lds[buf][0][0], ks, wn * 64 + i * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = frag(&lds[buf][1][0], ks, wk * 64 + j * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<T>(fa[i], fb[j], acc[i][j]);
        }
        if (s + 1 < nstep) stash(buf ^ 1);
        __syncthreads();
    }

    // ---- partial slab of this (group, batch, split)
    float *slab = a.part + ((long)g * per + zz) * ((long)a.N * a.K + a.N);
    const int col_l = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kc = k0 + wk * 64 + j * 32 + col_l;
            if (kc < a.K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rh;
                    if (n < a.N) slab[(long)n * a.K + kc] = acc[i][j][r];
                }
            }
        }
    if (do_bias) {   // column sums of gy: 16 token rows of threads per channel chunk, through LDS
        float *red = reinterpret_cast<float *>(&lds[0][0][0]);   // (16, 128) floats = 8 KB; the tile loop has ended
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[stok * kWgTile + sch + j] = bsum[j];
        __syncthreads();
        if (tid < kWgTile && n0 + tid < a.N) {
            float sacc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += red[r * kWgTile + tid];
            slab[(long)a.N * a.K + n0 + tid] = sacc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same TN GEMM with its token tiles staged by LDS-DMA (r03).  wgrad_tn_kernel above keeps ONE 32-token tile in flight
// (registers -> ds_write -> barrier): every shape of a training step, 1 152 to 73 728 tokens, 0.5 to 9.7 GFLOP, took
// 22-37 us = 16 K steps x ~1.5 us of memory latency, with 8 MFMAs per wave and step in between (scripts/bench_wgrad.py; an
// XCD-aware tile order changed nothing: the launches wait on latency, not on bandwidth).  Here NSTG - 1 tiles are in flight
// per workgroup and no VGPR is spent on staging:
//   * a stage = (gy 32 tokens x 128 channels | x 32 x 128) 16-bit = 16 KB, rows of 256 B WITHOUT padding; a DMA piece is
//     4 token rows (lane i -> row i / 16, 16-byte chunk position i % 16); waves 0-1 fetch gy, waves 2-3 fetch x, four
//     pieces per wave and step;
//   * the transposed fragment reads (ds_read_b64_tr_b16: a 16-lane group reads 4 token rows x 32 B) want those 4 rows on
//     different banks, which the 320-byte rows of the register-staged kernel gave: here the 16-byte chunk c of token row t
//     sits at chunk position c ^ 4 (t & 3), applied on the GLOBAL side of the DMA (lane i fetches chunk (i % 16) ^ 4 (i / 16))
//     and undone in the per-lane read address -- the 8 (row, channel-half) pieces of a 32-lane group cover all 64 banks;
//   * step:  s_waitcnt vmcnt(4 (NSTG - 2)) (my pieces of this tile have landed; hand-counted: hipcc does not see what an
//            LDS-DMA writes), s_barrier, DMA of tile + NSTG - 1 into the stage just vacated, 16 hand-written
//            ds_read_b64_tr_b16 behind counted lgkmcnt waits, 8 MFMAs;
//   * token rows past the chunk and channel chunks past N / K arrive as zeros through the descriptor's range check.
// Same slabs, same bias column sums (read back from the staged gy tile by the k-tile-0 workgroups), same fixed-order sum.
#define TRAMBA_TR64_(OUT, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(OUT) : "v"(ADDR) : "memory")

typedef unsigned v2u32 __attribute__((ext_vector_type(2)));

template <typename T, int NSTG>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(WgradArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (vector-register asm in a kernel template: the host pass only needs the stub)
    static_assert(NSTG == 3 || NSTG == 4, "stage index = step % NSTG with the loop unrolled by NSTG");
    constexpr int kTileB = kWgTok * kWgTile * 2;     // 8 KB: one operand's 32-token tile
    constexpr int kStageB = 2 * kTileB;              // 16 KB
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTG * kStageB];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv >> 1, wk = wv & 1;                      // wave tile: 64 (n) x 64 (k)
    unsigned bx_, by_, bz_;
    xcd_work_item(bx_, by_, bz_);                              // a token chunk's tiles side by side on one XCD
    const int bx = (int)bx_, by = (int)by_, bz = (int)bz_;
    const int n0 = by * kWgTile, k0 = bx * kWgTile;
    const int per = a.nbatch * a.nsplit;
    const int g = bz / per, zz = bz % per;
    const int b = zz / a.nsplit, sp = zz % a.nsplit;
    const int t0 = sp * a.mchunk;
    const int t1 = t0 + a.mchunk < a.M ? t0 + a.mchunk : a.M;
    const T *gyb = (const T *)a.gy + (long)b * a.gy_bs + (long)g * a.gy_gs;
    const T *xb = (const T *)a.x + (long)b * a.x_bs + (long)g * a.x_gs;
    // waves 0-1 stage gy, waves 2-3 stage x; the descriptor ends with the last token of MY chunk (rows past it read as zero)
    const bool stage_x = wv >= 2;
    const int ld = stage_x ? a.x_ld : a.gy_ld, ncol = stage_x ? a.K : a.N, col0 = stage_x ? k0 : n0;
    const __amdgpu_buffer_rsrc_t rs =
        make_rsrc(stage_x ? (const void *)xb : (const void *)gyb, (unsigned)(((long)(t1 - 1) * ld + ncol) * 2l));
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = ((wv & 1) * 4 + j) * 4 + (lane >> 4);            // token row inside the tile
        const int c = (lane & 15) ^ (4 * (row & 3));                     // the chunk that belongs at my position
        voff[j] = col0 + 8 * c < ncol ? (unsigned)(((long)(t0 + row) * ld + col0 + 8 * c) * 2l) : kOutOfRange;
    }
    unsigned char *mine = lds + (stage_x ? kTileB : 0) + (wv & 1) * 4096;
    const unsigned stepb = (unsigned)(kWgTok * ld) * 2u;                 // bytes between consecutive tiles of my operand
    // (the tile's token offset rides in the VECTOR offset: the range check that zero-fills the rows past my chunk does not
    //  look at the scalar offset; an out-of-range channel chunk stays out of range, 2^31 + less than 2^31)
    auto issue = [&](int st, int stg) {
        const unsigned so = (unsigned)st * stepb;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(mine + stg * kStageB + j * 1024), 16, voff[j] + so, 0, 0, 0);
    };
    // transposed fragment reads: 16-lane group gq = lane >> 4 -> channels 16 (gq & 1) .. +15 of a 32-channel block, tokens
    // 8 (gq >> 1) .. +3 (the second read: +4); lane 4 q + p of the group addresses token row q, channels 4 p .. 4 p + 3
    const int li = lane & 15, q = li >> 2, pp = li & 3, gq = lane >> 4;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned fr0 = lbase + (unsigned)((8 * (gq >> 1) + q) * 256 + (((2 * (gq & 1) + (pp >> 1)) ^ (4 * q)) * 16) + (pp & 1) * 8);
    // channel block cblk (a multiple of 32) moves the chunk index by cblk / 8, a multiple of 4: an XOR of the address
    unsigned ga[2], xa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ga[i] = fr0 ^ (unsigned)((wn * 64 + i * 32) * 2);
        xa[i] = (fr0 ^ (unsigned)((wk * 64 + i * 32) * 2)) + (unsigned)kTileB;
    }
    // bias column sums (k-tile 0 only): thread t re-reads chunk positions t % 16 of rows t / 16 and t / 16 + 16 of the gy
    // tile -- both rows hold the same channel chunk there, (t % 16) ^ 4 ((t / 16) & 3)
    const bool do_bias = (a.want_bias & 1) && bx == 0;
    const bool counted = (a.want_bias & 256) != 0;    // (TRAMBA_TUNE_WGRAD_FORM 2: the r03 counted waits, for measurements only)
    const unsigned ba = lbase + (unsigned)((tid >> 4) * 256 + (tid & 15) * 16);
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

    acc16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nstep = (t1 - t0 + kWgTok - 1) / kWgTok;
    constexpr int DEPTH = NSTG - 1;
#pragma unroll
    for (int t = 0; t < DEPTH; ++t)
        if (t < nstep) issue(t, t);
    auto kstep = [&](int st, auto stg_c) {
        constexpr int STG = decltype(stg_c)::value;
        const int later = nstep - 1 - st;           // tiles after this one; DEPTH - 1 of them (4 pieces each) may be in flight
        if (later >= DEPTH - 1) {
            if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (later == 1) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + DEPTH < nstep) issue(st + DEPTH, (STG + DEPTH) % NSTG);
        v2u32 fg[2][2][2], fx[2][2][2];             // [channel block][16-token slice][tokens 0-3 | 4-7]
#define TRAMBA_RD_STAGE_(S)                                                                                               \
    TRAMBA_TR64_(fg[0][0][0], ga[0], S); TRAMBA_TR64_(fg[0][0][1], ga[0], S + 1024);                                      \
    TRAMBA_TR64_(fx[0][0][0], xa[0], S); TRAMBA_TR64_(fx[0][0][1], xa[0], S + 1024);                                      \
    TRAMBA_TR64_(fg[1][0][0], ga[1], S); TRAMBA_TR64_(fg[1][0][1], ga[1], S + 1024);                                      \
    TRAMBA_TR64_(fx[1][0][0], xa[1], S); TRAMBA_TR64_(fx[1][0][1], xa[1], S + 1024);                                      \
    TRAMBA_TR64_(fg[0][1][0], ga[0], S + 4096); TRAMBA_TR64_(fg[0][1][1], ga[0], S + 5120);                               \
    TRAMBA_TR64_(fx[0][1][0], xa[0], S + 4096); TRAMBA_TR64_(fx[0][1][1], xa[0], S + 5120);                               \
    TRAMBA_TR64_(fg[1][1][0], ga[1], S + 4096); TRAMBA_TR64_(fg[1][1][1], ga[1], S + 5120);                               \
    TRAMBA_TR64_(fx[1][1][0], xa[1], S + 4096); TRAMBA_TR64_(fx[1][1][1], xa[1], S + 5120)
        if constexpr (STG == 0) { TRAMBA_RD_STAGE_(0); }
        else if constexpr (STG == 1) { TRAMBA_RD_STAGE_(16384); }
        else if constexpr (STG == 2) { TRAMBA_RD_STAGE_(32768); }
        else { TRAMBA_RD_STAGE_(49152); }
#undef TRAMBA_RD_STAGE_
        v4u32 bv[2];
        if (do_bias) {   // (block-uniform)
            if constexpr (STG == 0) {
                asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(bv[0]) : "v"(ba) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bv[1]) : "v"(ba) : "memory");
            } else if constexpr (STG == 1) {
                asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(bv[0]) : "v"(ba) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(bv[1]) : "v"(ba) : "memory");
            } else if constexpr (STG == 2) {
                asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(bv[0]) : "v"(ba) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:36864" : "=v"(bv[1]) : "v"(ba) : "memory");
            } else {
                asm volatile("ds_read_b128 %0, %1 offset:49152" : "=v"(bv[0]) : "v"(ba) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:53248" : "=v"(bv[1]) : "v"(ba) : "memory");
            }
        }
        // r04: ALL of the step's transposed reads have returned before the first MFMA.  r03 waited for them four at a time with
        // counted lgkmcnt values ("the reads return in order") -- true while only workgroups of THIS kernel share the CU, but with
        // a workgroup of another LDS-heavy kernel beside it (the projections' ds_read_b128 traffic, the fused scans') the MFMAs
        // now and then consumed a ds_read_b64_tr_b16 destination that had not been written yet: weight gradients that differ from
        // run to run in a few elements, NaN about one training step in a hundred -- found when the guide branches' backward first ran on
        // a second stream (scripts/dev/debug_wgrad_concurrent.py: 16-28 of 180 launches differ beside linear_pc / linear_ws /
        // scan_dma, 0 with this wait; the counted waits below then never wait).  Costs nothing measurable (scripts/bench_wgrad.py).
        if (!counted) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto mk = [](v2u32 lo, v2u32 hi) -> frag8s {
            const v4u32 v = {lo.x, lo.y, hi.x, hi.y};
            return __builtin_bit_cast(frag8s, v);
        };
        // counted waits (the reads return in order): slice 0 of block 0 needs the first 4 reads, block 1 four more, ...
        if (do_bias) {
            asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(fg[0][0][0]), "+v"(fg[0][0][1]), "+v"(fx[0][0][0]), "+v"(fx[0][0][1]) : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(fg[0][0][0]), "+v"(fg[0][0][1]), "+v"(fx[0][0][0]), "+v"(fx[0][0][1]) : : "memory");
        }
        acc[0][0] = mfma16<T>(mk(fg[0][0][0], fg[0][0][1]), mk(fx[0][0][0], fx[0][0][1]), acc[0][0]);
        if (do_bias) {
            asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(fg[1][0][0]), "+v"(fg[1][0][1]), "+v"(fx[1][0][0]), "+v"(fx[1][0][1]) : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fg[1][0][0]), "+v"(fg[1][0][1]), "+v"(fx[1][0][0]), "+v"(fx[1][0][1]) : : "memory");
        }
        acc[1][0] = mfma16<T>(mk(fg[1][0][0], fg[1][0][1]), mk(fx[0][0][0], fx[0][0][1]), acc[1][0]);
        acc[0][1] = mfma16<T>(mk(fg[0][0][0], fg[0][0][1]), mk(fx[1][0][0], fx[1][0][1]), acc[0][1]);
        acc[1][1] = mfma16<T>(mk(fg[1][0][0], fg[1][0][1]), mk(fx[1][0][0], fx[1][0][1]), acc[1][1]);
        if (do_bias) {
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fg[0][1][0]), "+v"(fg[0][1][1]), "+v"(fx[0][1][0]), "+v"(fx[0][1][1]) : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fg[0][1][0]), "+v"(fg[0][1][1]), "+v"(fx[0][1][0]), "+v"(fx[0][1][1]) : : "memory");
        }
        acc[0][0] = mfma16<T>(mk(fg[0][1][0], fg[0][1][1]), mk(fx[0][1][0], fx[0][1][1]), acc[0][0]);
        if (do_bias) {
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fg[1][1][0]), "+v"(fg[1][1][1]), "+v"(fx[1][1][0]), "+v"(fx[1][1][1]) : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fg[1][1][0]), "+v"(fg[1][1][1]), "+v"(fx[1][1][0]), "+v"(fx[1][1][1]) : : "memory");
        }
        acc[1][0] = mfma16<T>(mk(fg[1][1][0], fg[1][1][1]), mk(fx[0][1][0], fx[0][1][1]), acc[1][0]);
        acc[0][1] = mfma16<T>(mk(fg[0][1][0], fg[0][1][1]), mk(fx[1][1][0], fx[1][1][1]), acc[0][1]);
        acc[1][1] = mfma16<T>(mk(fg[1][1][0], fg[1][1][1]), mk(fx[1][1][0], fx[1][1][1]), acc[1][1]);
        if (do_bias) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]) : : "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Pack<T, 8> pk = __builtin_bit_cast(Pack<T, 8>, bv[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[j] += Cvt<T>::to_f(pk.v[j]);
            }
        }
    };
    int st0 = 0;
    for (; st0 + NSTG <= nstep; st0 += NSTG) {
        kstep(st0, std::integral_constant<int, 0>{});
        kstep(st0 + 1, std::integral_constant<int, 1>{});
        kstep(st0 + 2, std::integral_constant<int, 2>{});
        if constexpr (NSTG > 3) kstep(st0 + 3, std::integral_constant<int, 3>{});
    }
    if (st0 < nstep) kstep(st0, std::integral_constant<int, 0>{});
    if (st0 + 1 < nstep) kstep(st0 + 1, std::integral_constant<int, 1>{});
    if constexpr (NSTG > 3) {
        if (st0 + 2 < nstep) kstep(st0 + 2, std::integral_constant<int, 2>{});
    }

    // ---- partial slab of this (group, batch, split)
    float *slab = a.part + ((long)g * per + zz) * ((long)a.N * a.K + a.N);
    const int col_l = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kc = k0 + wk * 64 + j * 32 + col_l;
            if (kc < a.K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rh;
                    if (n < a.N) slab[(long)n * a.K + kc] = acc[i][j][r];
                }
            }
        }
    if (do_bias) {   // column sums of gy: 16 row lanes per channel chunk, through LDS (the tile loop has ended)
        float *red = reinterpret_cast<float *>(&lds[0]);   // (16, 128) floats = 8 KB
        __syncthreads();
        const int sch = 8 * ((tid & 15) ^ (4 * ((tid >> 4) & 3)));      // the channel chunk my two rows hold at my position
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(tid >> 4) * kWgTile + sch + j] = bsum[j];
        __syncthreads();
        if (tid < kWgTile && n0 + tid < a.N) {
            float sacc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += red[r * kWgTile + tid];
            slab[(long)a.N * a.K + n0 + tid] = sacc;
        }
    }
#endif
}
#undef TRAMBA_TR64_

// out[i] = sum over slabs s of part[s][i], i < n (fixed order); 4 floats per thread, EIGHT slab rows requested before
// the first is added (r03: four were not enough to cover the memory latency -- 29 MB in 18.6 us = 1.6 TB/s on the widest
// layers of a training step, as much as the TN kernel in front of it).  `blk`: index of the 1024-float block.
__device__ __forceinline__ void slab_sum_body(const float *__restrict__ part, float *__restrict__ out, long n, int nslab, long blk,
                                              long stride)   // stride: floats between consecutive slabs (= n for a dense table)
{
    const long i4 = (blk * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    const float *p = part + i4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 + 4 <= n) {
        int s = 0;
        for (; s + 8 <= nslab; s += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(p + (long)(s + j) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
        }
        if (s < nslab) {   // the last 1..7 rows: clamped re-reads weighted 0 keep the loads unconditional and in flight together
            float4 v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int sj = s + j < nslab ? s + j : nslab - 1;
                v[j] = *reinterpret_cast<const float4 *>(p + (long)sj * stride);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if (s + j < nslab) {   // (uniform)
                    acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
                }
            }
        }
        *reinterpret_cast<float4 *>(out + i4) = acc;
    } else {
        for (long i = i4; i < n; ++i) {
            float sacc = 0.f;
            for (int s = 0; s < nslab; ++s) sacc += part[(long)s * stride + i];
            out[i] = sacc;
        }
    }
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ part, float *__restrict__ out, long n,
                                                      int nslab)
{
    slab_sum_body(part + (long)blockIdx.y * nslab * n, out + (long)blockIdx.y * n, n, nslab, blockIdx.x, n);
}

// ------------------------------------------------------------------------------------------------
// rows_gemm: y[z][t][n] = sum_k x[z][t][k] * w[z % groups][n][k], n < N <= 64, written as fp32 rows of stride ldy.
// One wave owns 32 token rows: A = x rows (K-contiguous, 16-byte fragment loads), B = the group's weight rows (L2).
// KSPLIT (r03): long rows (K >= 512: the 24x24 / 48x48 cores, D = 1024 / 512) gave 4.5 workgroups per (b, k) plane -- 640
// waves for the chip, every one streaming 32 rows of 2 KB with one fragment load in flight per step (19.9 us for 37.7 MB).
// There the four waves of a workgroup share ONE 32-row tile and split K in quarters (4x the workgroups, 4x the loads in
// flight per row); the partial tiles meet in LDS and are added in wave order (fixed).
template <typename T, bool KSPLIT>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                       float *__restrict__ y, int M, int N, int K, int groups, int ldy)
{
    __shared__ float part[KSPLIT ? 4 : 1][KSPLIT ? 32 : 1][KSPLIT ? 65 : 1];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int z = blockIdx.y, gidx = z % groups;
    const long row0 = KSPLIT ? (long)blockIdx.x * 32 : ((long)blockIdx.x * 4 + wv) * 32;
    if (!KSPLIT && row0 >= M) return;            // wave-uniform, no barriers on this path
    const int r32 = lane & 31, kh = (lane >> 5) * 8;
    const T *xz = x + (long)z * M * K;
    const T *wz = w + (long)gidx * N * K;
    const long row = row0 + r32 < M ? row0 + r32 : M - 1;   // clamped: rows past the end are computed and dropped
    const int c0 = r32 < N ? r32 : N - 1, c1 = 32 + r32 < N ? 32 + r32 : N - 1;
    acc16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const bool two = N > 32;                     // uniform
    const frag8s zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const int kq = KSPLIT ? K / 4 : K;           // (host: K % 64 == 0 when split)
    const int kbeg = KSPLIT ? wv * kq : 0, kend = kbeg + kq;
#pragma unroll 4
    for (int k0 = kbeg; k0 < kend; k0 += 16) {   // K is a multiple of 8 (host-checked): an 8-element run is in or out whole
        const bool in = k0 + kh < kend;
        const int kk = in ? k0 + kh : kbeg;
        frag8s fa = *reinterpret_cast<const frag8s *>(xz + row * K + kk);
        const frag8s f0 = *reinterpret_cast<const frag8s *>(wz + (long)c0 * K + kk);
        if (!in) fa = zero;
        acc[0] = mfma16<T>(fa, f0, acc[0]);
        if (two) {
            const frag8s f1 = *reinterpret_cast<const frag8s *>(wz + (long)c1 * K + kk);
            acc[1] = mfma16<T>(fa, f1, acc[1]);
        }
    }
    float *yz = y + (long)z * M * ldy;
    if constexpr (KSPLIT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + r32;
            if (col < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) part[wv][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][col] = acc[j][r];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * N; i += 256) {
            const int t = i / N, col = i - t * N;
            if (row0 + t < M)
                yz[(row0 + t) * ldy + col] = ((part[0][t][col] + part[1][t][col]) + part[2][t][col]) + part[3][t][col];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + r32;
            if (col < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long t = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (t < M) yz[t * ldy + col] = acc[j][r];
                }
            }
        }
    }
}

// out[i] = sum_s part[s][i] for the many small partial-sum tables of the backward pass (LayerNorm / depth-wise / scan
// parameter gradients: 10..1000 slabs of 256..12288 floats).  A block owns 32 consecutive floats (one 128-byte line per
// slab) and spreads the slabs over 32 row lanes, eight loads in flight per thread; the row lanes are folded through LDS in a
// fixed order.  ~200 launches per training step of Tramba-V: about half the time of a general-purpose reduction each.
__device__ __forceinline__ void col_sum_body(const float *__restrict__ part, float *__restrict__ out, long n, int nslab,
                                             long blk, float4 (*red)[8], long stride)
{
    const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const long i4 = (blk * 8 + q) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n) {
        const float *p = part + i4;
        int s = rl;
        for (; s + 7 * 32 < nslab; s += 8 * 32) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(p + (long)(s + 32 * j) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
        }
        for (; s < nslab; s += 32) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (long)s * stride);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[rl][q] = acc;
    __syncthreads();
    if (rl == 0 && i4 < n) {
        float4 t = red[0][q];
#pragma unroll
        for (int r = 1; r < 32; ++r) {
            const float4 v = red[r][q];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + i4) = t;
    }
}

__global__ __launch_bounds__(256) void col_sum_kernel(const float *__restrict__ part, float *__restrict__ out, long n,
                                                     int nslab)
{
    __shared__ float4 red[32][8];
    col_sum_body(part, out, n, nslab, blockIdx.x, red, n);
}

// Up to kMultiSum partial-sum tables reduced by ONE launch (r03: the deferred sums of a training step -- the slab sums
// of the weight-gradient GEMMs and the LayerNorm parameter gradients were ~300 launches of 4-7 us each, every one on its
// launch floor; tramba_multi_sum).  Descriptors travel by value in the kernel arguments (nothing to copy to the device,
// hipGraph-capture safe); a workgroup finds its table by its block index; tables of >= 32 slabs take the row-lane form
// (32 floats per workgroup), the others the slab-order form (1024 floats per workgroup) -- the same bodies, hence the same
// summation order, as the single-table launches.
constexpr int kMultiSum = 64;
struct MultiSumArgs {
    const float *part[kMultiSum];
    float *out[kMultiSum];
    long n[kMultiSum];
    long stride[kMultiSum];     // floats between consecutive slabs: n for a dense table, more for a row range of a wider one
    int nslab[kMultiSum];
    int first[kMultiSum + 1];   // first workgroup of every table (ascending), first[count] = grid size
    int count;
};

static_assert(sizeof(MultiSumArgs) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(256) void multi_sum_kernel(MultiSumArgs a)
{
    __shared__ float4 red[32][8];
    const int blk = blockIdx.x;
    int t = 0;
#pragma unroll 1
    for (int i = 1; i < a.count; ++i)
        if (a.first[i] <= blk) t = i;          // (block-uniform; first[] is ascending)
    const long local = blk - a.first[t];
    if (a.nslab[t] >= 32) col_sum_body(a.part[t], a.out[t], a.n[t], a.nslab[t], local, red, a.stride[t]);
    else slab_sum_body(a.part[t], a.out[t], a.n[t], a.nslab[t], local, a.stride[t]);
}

// ---- multi-tensor cast (+ transpose) of fp32 matrices: table[t] = {src, dst, dst_t, rows, cols, first_tile, dst_ld,
//      dst_t_ld} (int64 each); src is dense (rows, cols), dst rows are dst_ld apart, dst_t rows dst_t_ld apart
constexpr int kShadowTile = 64;

template <typename T>
__global__ __launch_bounds__(256) void shadow_cast_multi_kernel(const long *__restrict__ table, int ntensors)
{
    __shared__ T tile[kShadowTile][kShadowTile + 2];
    const long tileid = blockIdx.x;
    int lo = 0, hi = ntensors - 1;                  // last tensor whose first tile is <= tileid
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[(long)mid * 8 + 5] <= tileid) lo = mid; else hi = mid - 1;
    }
    const long *e = table + (long)lo * 8;
    const float *src = reinterpret_cast<const float *>(e[0]);
    T *dst = reinterpret_cast<T *>(e[1]);
    T *dst_t = reinterpret_cast<T *>(e[2]);
    const long rows = e[3], cols = e[4], ld = e[6], ld_t = e[7];
    const long local = tileid - e[5];
    const long tiles_c = (cols + kShadowTile - 1) / kShadowTile;
    const long r0 = (local / tiles_c) * kShadowTile, c0 = (local % tiles_c) * kShadowTile;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const bool vec = (cols & 3) == 0;               // 16-byte loads of whole quads
    const bool vec_d = vec && (ld & 3) == 0;        // 8-byte stores
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long r = r0 + ty + 16 * i, c = c0 + 4 * tx;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (vec && c + 4 <= cols) {
                const float4 q = *reinterpret_cast<const float4 *>(src + r * cols + c);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < cols) v[j] = src[r * cols + c + j];
            }
        }
        T o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = Cvt<T>::from_f(v[j]);
            tile[ty + 16 * i][4 * tx + j] = o[j];
        }
        if (dst && r < rows) {
            if (vec_d && c + 4 <= cols) {
                *reinterpret_cast<uint2 *>(dst + r * ld + c) = *reinterpret_cast<const uint2 *>(o);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < cols) dst[r * ld + c + j] = o[j];
            }
        }
    }
    if (!dst_t) return;                              // (uniform over the block)
    __syncthreads();
    const bool vec_t = (rows & 3) == 0 && (ld_t & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long c = c0 + ty + 16 * i, r = r0 + 4 * tx;   // output row = source column
        if (c >= cols) continue;
        T o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = tile[4 * tx + j][ty + 16 * i];
        if (vec_t && r + 4 <= rows) {
            *reinterpret_cast<uint2 *>(dst_t + c * ld_t + r) = *reinterpret_cast<const uint2 *>(o);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r + j < rows) dst_t[c * ld_t + r + j] = o[j];
        }
    }
}

}  // namespace tramba

using namespace tramba;

static void wgrad_plan(long m, int n, int k, int groups, int nbatch, int &nsplit, int &mchunk)
{
    // Token splits: ONE workgroup per CU and never chunks below 512 tokens.  A workgroup's K loop runs at the rate its CU
    // fills LDS from L2 (~30 B per cycle: a 16 KB step every ~560 cycles, whatever the staging); co-resident workgroups
    // only share that rate, while every extra split costs another 4 N K bytes of partial sums written and read back
    // (scripts/bench_wgrad.py, r03: 256 wanted workgroups beat 384 / 512 / 768 on 14 of 16 shapes, -8 % over the step's set).
    const long tiles = (long)((n + kWgTile - 1) / kWgTile) * ((k + kWgTile - 1) / kWgTile) * groups * nbatch;
    const int tune = tramba_tune_get(TRAMBA_TUNE_GEMM_TILE);   // 10 / 11 / 12: 384 / 512 / 768 workgroups wanted (measurements)
    const long target = tune == 10 ? 384 : (tune == 11 ? 512 : (tune == 12 ? 768 : 256));
    long want = (target + tiles - 1) / tiles;
    const long maxsplit = m / 512 > 1 ? m / 512 : 1;
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    long chunk = (m + want - 1) / want;
    chunk = (chunk + kWgTok - 1) / kWgTok * kWgTok;
    nsplit = (int)((m + chunk - 1) / chunk);
    mchunk = (int)chunk;
}

extern "C" size_t tramba_wgrad_workspace(int64_t m, int n, int k, int groups, int nbatch)
{
    if (m <= 0 || n <= 0 || k <= 0 || groups <= 0 || nbatch <= 0) return 0;
    int nsplit, mchunk;
    wgrad_plan(m, n, k, groups, nbatch, nsplit, mchunk);
    return (size_t)groups * nbatch * nsplit * ((size_t)n * k + n) * sizeof(float);
}

// nslab_out: null = the slabs are summed here (tramba_wgrad_cl); else the launch stops at the slabs and reports how many
// there are per group (0: a single slab, written straight to `out`) -- tramba_wgrad_parts_cl, for callers that sum later
static int wgrad_launch(const void *gy, const void *x, float *out, void *workspace, size_t workspace_bytes,
                        int64_t m, int n, int k, int groups, int nbatch, int64_t gy_bs, int64_t gy_gs, int gy_ld,
                        int64_t x_bs, int64_t x_gs, int x_ld, int want_bias, int dtype, void *stream, int *nslab_out)
{
    TRAMBA_CHECK(gy && x && out && workspace, "wgrad_cl: null tensor");
    TRAMBA_CHECK(m > 0 && n > 0 && k > 0 && groups > 0 && nbatch > 0, "wgrad_cl: empty shape");
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "wgrad_cl: 16-bit operands only");
    TRAMBA_CHECK(n % 8 == 0 && k % 8 == 0 && gy_ld % 8 == 0 && x_ld % 8 == 0 && gy_bs % 8 == 0 && gy_gs % 8 == 0 &&
                     x_bs % 8 == 0 && x_gs % 8 == 0,
                 "wgrad_cl: channel counts and strides must be multiples of 8 elements (16-byte rows)");
    TRAMBA_CHECK(aligned16(gy) && aligned16(x) && aligned16(out) && aligned16(workspace), "wgrad_cl: 16-byte alignment");
    TRAMBA_CHECK(((double)(m - 1) * gy_ld + n) * 2.0 < 2147483648.0 && ((double)(m - 1) * x_ld + k) * 2.0 < 2147483648.0,
                 "wgrad_cl: one (group, batch) operand must stay below 2 GiB");
    TRAMBA_CHECK(workspace_bytes >= tramba_wgrad_workspace(m, n, k, groups, nbatch), "wgrad_cl: workspace too small");
    int nsplit, mchunk;
    wgrad_plan(m, n, k, groups, nbatch, nsplit, mchunk);
    TRAMBA_CHECK((long)groups * nbatch * nsplit <= 65535, "wgrad_cl: too many token chunks");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_WGRAD, s, 2.0 * (double)m * n * k * groups * nbatch);
    WgradArgs a;
    const bool direct = nsplit * nbatch == 1;          // one slab per group: it IS the result
    a.gy = gy; a.x = x; a.part = direct ? out : (float *)workspace;
    a.gy_bs = gy_bs; a.gy_gs = gy_gs; a.x_bs = x_bs; a.x_gs = x_gs; a.gy_ld = gy_ld; a.x_ld = x_ld;
    a.M = (int)m; a.N = n; a.K = k; a.nbatch = nbatch; a.nsplit = nsplit; a.mchunk = mchunk; a.want_bias = want_bias ? 1 : 0;
    if (tramba_tune_get(TRAMBA_TUNE_WGRAD_FORM) == 2) a.want_bias |= 256;
    dim3 grid((k + kWgTile - 1) / kWgTile, (n + kWgTile - 1) / kWgTile, groups * nbatch * nsplit), block(256);
    // LDS-DMA staged tiles (the default) need 32-bit byte offsets inside one (group, batch) operand -- checked above -- and
    // rows of whole 16-byte chunks; TRAMBA_TUNE_GEMM_TILE 8 / 9 select the register-staged form / 4 stages for measurements
    const int tune = tramba_tune_get(TRAMBA_TUNE_GEMM_TILE);
    if (tune == 8 || tramba_tune_get(TRAMBA_TUNE_WGRAD_FORM) == 1) {
        if (dtype == TRAMBA_BF16) hipLaunchKernelGGL((wgrad_tn_kernel<__hip_bfloat16>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wgrad_tn_kernel<__half>), grid, block, 0, s, a);
    } else if (tune == 9) {
        if (dtype == TRAMBA_BF16) hipLaunchKernelGGL((wgrad_dma_kernel<__hip_bfloat16, 4>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wgrad_dma_kernel<__half, 4>), grid, block, 0, s, a);
    } else {   // 3 stages of 16 KB: three workgroups per CU (measured 3-5 % ahead of 4 stages / two workgroups)
        if (dtype == TRAMBA_BF16) hipLaunchKernelGGL((wgrad_dma_kernel<__hip_bfloat16, 3>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wgrad_dma_kernel<__half, 3>), grid, block, 0, s, a);
    }
    TRAMBA_LAUNCH_CHECK();
    if (nslab_out) {
        *nslab_out = direct ? 0 : nbatch * nsplit;
        return TRAMBA_OK;
    }
    if (direct) return TRAMBA_OK;
    const long slab = (long)n * k + n;
    const int nslab = nbatch * nsplit;
    if (groups == 1 && nslab >= 32 && slab % 4 == 0) {
        // many slabs of a small output (the 96x96 / 48x48 layers: up to 144 slabs of 33 K floats): one thread per float4
        // would walk them as a serial chain of 18 batches from 33 workgroups (13 us); col_sum_kernel spreads the slabs over
        // 32 row lanes per 128-byte column group and folds them through LDS in a fixed order
        dim3 g3((unsigned)((slab / 4 + 7) / 8));
        hipLaunchKernelGGL(col_sum_kernel, g3, dim3(256), 0, s, (const float *)workspace, out, slab, nslab);
    } else {
        dim3 g2((unsigned)((slab + 1023) / 1024), groups);
        hipLaunchKernelGGL(slab_sum_kernel, g2, dim3(256), 0, s, (const float *)workspace, out, slab, nslab);
    }
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_wgrad_cl(const void *gy, const void *x, float *out, void *workspace, size_t workspace_bytes,
                               int64_t m, int n, int k, int groups, int nbatch, int64_t gy_bs, int64_t gy_gs, int gy_ld,
                               int64_t x_bs, int64_t x_gs, int x_ld, int want_bias, int dtype, void *stream)
{
    return wgrad_launch(gy, x, out, workspace, workspace_bytes, m, n, k, groups, nbatch, gy_bs, gy_gs, gy_ld, x_bs, x_gs, x_ld,
                        want_bias, dtype, stream, nullptr);
}

extern "C" int tramba_wgrad_parts_cl(const void *gy, const void *x, float *out, void *workspace, size_t workspace_bytes,
                                     int64_t m, int n, int k, int groups, int nbatch, int64_t gy_bs, int64_t gy_gs,
                                     int gy_ld, int64_t x_bs, int64_t x_gs, int x_ld, int want_bias, int dtype,
                                     void *stream, int *nslab)
{
    TRAMBA_CHECK(nslab, "wgrad_parts_cl: null nslab");
    return wgrad_launch(gy, x, out, workspace, workspace_bytes, m, n, k, groups, nbatch, gy_bs, gy_gs, gy_ld, x_bs, x_gs, x_ld,
                        want_bias, dtype, stream, nslab);
}

extern "C" int tramba_rows_gemm_cl(const void *x, const void *w, float *y, int nz, int64_t m, int n, int k, int groups,
                                   int ldy, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && y, "rows_gemm_cl: null tensor");
    TRAMBA_CHECK(nz > 0 && m > 0 && n > 0 && k > 0 && groups > 0, "rows_gemm_cl: empty shape");
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "rows_gemm_cl: 16-bit operands only");
    TRAMBA_CHECK(n <= 64 && k % 8 == 0 && ldy >= n && nz <= 65535, "rows_gemm_cl: needs N <= 64, K %% 8 == 0, ldy >= N");
    TRAMBA_CHECK(aligned16(x) && aligned16(w), "rows_gemm_cl: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const bool split = k >= 512 && k % 64 == 0;   // long rows: the workgroup's waves share a 32-row tile and split K
    dim3 grid((unsigned)(split ? (m + 31) / 32 : (m + 127) / 128), nz), block(256);
#define ROWS_(T, S_) \
    hipLaunchKernelGGL((rows_gemm_kernel<T, S_>), grid, block, 0, s, (const T *)x, (const T *)w, y, (int)m, n, k, groups, ldy)
    if (dtype == TRAMBA_BF16) {
        if (split) ROWS_(__hip_bfloat16, true); else ROWS_(__hip_bfloat16, false);
    } else {
        if (split) ROWS_(__half, true); else ROWS_(__half, false);
    }
#undef ROWS_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_shadow_cast_multi(const void *table, int ntensors, int64_t total_tiles, int dtype, void *stream)
{
    TRAMBA_CHECK(table && ntensors > 0 && total_tiles > 0, "shadow_cast_multi: empty table");
    TRAMBA_CHECK(dtype == TRAMBA_BF16 || dtype == TRAMBA_F16, "shadow_cast_multi: 16-bit shadows only");
    TRAMBA_CHECK(total_tiles < (1ll << 31), "shadow_cast_multi: too many tiles");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)total_tiles), block(256);
    if (dtype == TRAMBA_BF16)
        hipLaunchKernelGGL((shadow_cast_multi_kernel<__hip_bfloat16>), grid, block, 0, s, (const long *)table, ntensors);
    else
        hipLaunchKernelGGL((shadow_cast_multi_kernel<__half>), grid, block, 0, s, (const long *)table, ntensors);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_slab_sum(const float *part, float *out, int64_t n, int nslab, void *stream)
{
    TRAMBA_CHECK(part && out && n > 0 && nslab > 0, "slab_sum: empty input");
    TRAMBA_CHECK(aligned16(part) && aligned16(out) && n % 4 == 0, "slab_sum: rows of whole, 16-byte aligned float4s");
    dim3 grid((unsigned)((n / 4 + 7) / 8));
    hipLaunchKernelGGL(col_sum_kernel, grid, dim3(256), 0, (hipStream_t)stream, part, out, (long)n, nslab);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_multi_sum_strided(const float *const *parts, float *const *outs, const int64_t *n, const int64_t *stride,
                                        const int *nslab, int count, void *stream)
{
    TRAMBA_CHECK(parts && outs && n && nslab && count > 0, "multi_sum: empty input");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < count; base += kMultiSum) {
        MultiSumArgs a;
        a.count = count - base < kMultiSum ? count - base : kMultiSum;
        long blocks = 0;
        for (int i = 0; i < a.count; ++i) {
            const int j = base + i;
            TRAMBA_CHECK(parts[j] && outs[j] && n[j] > 0 && nslab[j] > 0 && n[j] % 4 == 0 && aligned16(parts[j]) && aligned16(outs[j]),
                         "multi_sum: table %d must hold whole, 16-byte aligned float4 rows", j);
            TRAMBA_CHECK(!stride || (stride[j] >= n[j] && stride[j] % 4 == 0), "multi_sum: table %d: slab stride %ld under the row length or not a multiple of 4", j, (long)(stride ? stride[j] : 0));
            a.part[i] = parts[j];
            a.out[i] = outs[j];
            a.n[i] = n[j];
            a.stride[i] = stride ? stride[j] : n[j];
            a.nslab[i] = nslab[j];
            a.first[i] = (int)blocks;
            blocks += nslab[j] >= 32 ? (n[j] / 4 + 7) / 8 : (n[j] + 1023) / 1024;
            TRAMBA_CHECK(blocks < 2147483647L, "multi_sum: too many workgroups");
        }
        for (int i = a.count; i <= kMultiSum; ++i) a.first[i] = (int)blocks;
        for (int i = a.count; i < kMultiSum; ++i) {
            a.part[i] = nullptr; a.out[i] = nullptr; a.n[i] = 0; a.stride[i] = 0; a.nslab[i] = 0;
        }
        hipLaunchKernelGGL(multi_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        TRAMBA_LAUNCH_CHECK();
    }
    return TRAMBA_OK;
}

extern "C" int tramba_multi_sum(const float *const *parts, float *const *outs, const int64_t *n, const int *nslab, int count,
                                void *stream)
{
    return tramba_multi_sum_strided(parts, outs, n, nullptr, nslab, count, stream);
}
