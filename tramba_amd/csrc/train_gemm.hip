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
    const int n0 = blockIdx.y * kWgTile, k0 = blockIdx.x * kWgTile;
    const int per = a.nbatch * a.nsplit;
    const int g = (int)blockIdx.z / per, zz = (int)blockIdx.z % per;
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
    const bool do_bias = a.want_bias && blockIdx.x == 0;
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

// out[g][i] = sum over slabs s of part[g][s][i], i < n (fixed order); 4 floats per thread
__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ part, float *__restrict__ out, long n,
                                                      int nslab)
{
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    const float *p = part + (long)blockIdx.y * nslab * n + i4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 + 4 <= n) {
        int s = 0;
        for (; s + 4 <= nslab; s += 4) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4 *>(p + (long)(s + j) * n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
        }
        for (; s < nslab; ++s) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (long)s * n);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + (long)blockIdx.y * n + i4) = acc;
    } else {
        for (long i = i4; i < n; ++i) {
            float sacc = 0.f;
            for (int s = 0; s < nslab; ++s) sacc += part[((long)blockIdx.y * nslab + s) * n + i];
            out[(long)blockIdx.y * n + i] = sacc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rows_gemm: y[z][t][n] = sum_k x[z][t][k] * w[z % groups][n][k], n < N <= 64, written as fp32 rows of stride ldy.
// One wave owns 32 token rows: A = x rows (K-contiguous, 16-byte fragment loads), B = the group's weight rows (L2).
template <typename T>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                       float *__restrict__ y, int M, int N, int K, int groups, int ldy)
{
    const int lane = threadIdx.x & 63;
    const int z = blockIdx.y, gidx = z % groups;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= M) return;                       // wave-uniform, no barriers
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
    for (int k0 = 0; k0 < K; k0 += 16) {         // K is a multiple of 8 (host-checked): an 8-element run is in or out whole
        const bool in = k0 + kh < K;
        const int kk = in ? k0 + kh : 0;
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

// out[i] = sum_s part[s][i] for the many small partial-sum tables of the backward pass (LayerNorm / depth-wise / scan
// parameter gradients: 10..1000 slabs of 256..12288 floats).  A block owns 32 consecutive floats (one 128-byte line per
// slab) and spreads the slabs over 32 row lanes, eight loads in flight per thread; the row lanes are folded through LDS in a
// fixed order.  ~200 launches per training step of Tramba-V: about half the time of a general-purpose reduction each.
__global__ __launch_bounds__(256) void col_sum_kernel(const float *__restrict__ part, float *__restrict__ out, long n,
                                                     int nslab)
{
    __shared__ float4 red[32][8];
    const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const long i4 = ((long)blockIdx.x * 8 + q) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n) {
        const float *p = part + i4;
        int s = rl;
        for (; s + 7 * 32 < nslab; s += 8 * 32) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(p + (long)(s + 32 * j) * n);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
        }
        for (; s < nslab; s += 32) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (long)s * n);
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
    // Token splits: enough workgroups to fill the chip (~1.5 per CU) but never chunks below 512 tokens -- a block reads
    // 512 B per token and writes a 64 KB fp32 slab, so shorter chunks spend more bytes on partial sums than on operands
    // (r02e trace: at ~1000 blocks for every shape the slab sums cost 60 % of the GEMM time).
    const long tiles = (long)((n + kWgTile - 1) / kWgTile) * ((k + kWgTile - 1) / kWgTile) * groups * nbatch;
    long want = (384 + tiles - 1) / tiles;
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

extern "C" int tramba_wgrad_cl(const void *gy, const void *x, float *out, void *workspace, size_t workspace_bytes,
                               int64_t m, int n, int k, int groups, int nbatch, int64_t gy_bs, int64_t gy_gs, int gy_ld,
                               int64_t x_bs, int64_t x_gs, int x_ld, int want_bias, int dtype, void *stream)
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
    a.M = (int)m; a.N = n; a.K = k; a.nbatch = nbatch; a.nsplit = nsplit; a.mchunk = mchunk; a.want_bias = want_bias;
    dim3 grid((k + kWgTile - 1) / kWgTile, (n + kWgTile - 1) / kWgTile, groups * nbatch * nsplit), block(256);
    if (dtype == TRAMBA_BF16) hipLaunchKernelGGL((wgrad_tn_kernel<__hip_bfloat16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wgrad_tn_kernel<__half>), grid, block, 0, s, a);
    TRAMBA_LAUNCH_CHECK();
    if (direct) return TRAMBA_OK;
    const long slab = (long)n * k + n;
    dim3 g2((unsigned)((slab + 1023) / 1024), groups);
    hipLaunchKernelGGL(slab_sum_kernel, g2, dim3(256), 0, s, (const float *)workspace, out, slab, nbatch * nsplit);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
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
    dim3 grid((unsigned)((m + 127) / 128), nz), block(256);
    if (dtype == TRAMBA_BF16)
        hipLaunchKernelGGL((rows_gemm_kernel<__hip_bfloat16>), grid, block, 0, s, (const __hip_bfloat16 *)x,
                           (const __hip_bfloat16 *)w, y, (int)m, n, k, groups, ldy);
    else
        hipLaunchKernelGGL((rows_gemm_kernel<__half>), grid, block, 0, s, (const __half *)x, (const __half *)w, y, (int)m, n,
                           k, groups, ldy);
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
