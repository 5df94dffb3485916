// Fused SS2D core, channels-last: the model's own hot path.
//
// Restates SS2Dv2.forward_corev2 (Models/vmamba.py:230-273) for d_state N = 1:
//   xs = CrossScan(x); x_dbl = x_proj(xs); dts = dt_proj(x_dbl[:R]); ys = selective_scan(...)
//   y = CrossMerge(ys); y = out_norm(y)
// without ever materialising the K-fold (B,K,D,L) gather, `dts`/`delta`, or an NCHW merge.
//
// Why channels-last.  In (B, L, D) layout a scan order is a permutation of ROWS: sequence
// position l of direction k reads the contiguous D-vector of pixel table[k][l].  Any order
// (raster, Bresenham helix, window, dilated) becomes a coalesced whole-row gather, lanes
// map to channels, the recurrence needs no cross-lane traffic, and the per-position
// operands shared by all channels (dt low-rank vector, B, C) are wave-uniform scalars.
// Because x_proj is linear and a scan order only permutes positions, x_proj is evaluated
// ONCE in spatial order (a plain GEMM, K*(R+2) outputs) and gathered here.
//
// ss2d_scan_cl:   grid (D/32, K, B); a workgroup of W waves owns 32 channels of one direction for
//   the whole sequence.  The unit of work is a TILE of 32 positions x 32 channels:
//   * dt_proj runs on the matrix cores: dt_raw[pos][ch] = <x_dbl[pos, :R], dt_w[ch, :R]> is one
//     v_mfma_f32_32x32x16_bf16 per 16 ranks (A = the gathered x_dbl rows, one position per lane;
//     B = this wave's dt_w slice, loop-invariant registers).  fp32 activations use the split
//     hi+lo bf16 form (3 MFMAs, ~2^-16 relative) so the fp32 path keeps fp32-level accuracy.
//   * the MFMA accumulator leaves each lane with 16 positions of ONE channel (4 runs of 4
//     consecutive positions, interleaved with the partner lane l^32), so softplus / exp / the
//     recurrence are pure per-lane VALU work on all 64 lanes; the two half-waves exchange their 4
//     run aggregates once per tile and fold them in sequence order.
//   * per step of W tiles: every wave reduces its tile to a (decay, state) pair, takes the state
//     entering its tile from its predecessor's LDS mailbox and publishes the state leaving it
//     (CarryLink: no workgroup barrier, the waves run decoupled), replays its 16 steps from
//     registers and streams y rows out.
//   * operands are prefetched two tiles deep (index vector -> row / u gathers -> compute), all
//     as vector loads with per-lane addresses; x_dbl groups are padded to 16-byte multiples.
// ss2d_merge_norm_cl: one wave per pixel sums the rows listed by the inverse table
//   (deterministic, atomic-free even for the many-to-one Helix lines), applies out_norm
//   (LayerNorm over D, two-pass fp32) and the following GELU, writes the activation dtype.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "norm.h"

namespace tramba {

constexpr int kMaxW = 8;     // waves per workgroup

typedef __attribute__((ext_vector_type(8))) short frag8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float acc16_t;

constexpr int kTP = 32;  // positions per tile (one MFMA tile)

// x_dbl group of one direction: [dt ranks 0..R-1, zero pad to R8 = 8*ceil(R/8) | B, C, 2 pad] -- every 8-rank run
// is a whole 16-byte-aligned pair of 16-byte loads and (B, C) one aligned 8-byte load, for any R.
__host__ __device__ inline int xdbl_rank_pad(int r) { return (r + 7) & ~7; }
__host__ __device__ inline int xdbl_group_stride(int r) { return xdbl_rank_pad(r) + 4; }
inline double rg_bytes(int r) { return 4.0 * xdbl_group_stride(r); }

// split 8 floats into bf16 hi (+ bf16 lo = rounding residual) fragments
template <bool SPLIT>
__device__ __forceinline__ void pack_frag(const float (&v)[8], frag8_t &hi, frag8_t &lo)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __hip_bfloat16 h = __float2bfloat16(v[j]);
        hi[j] = __builtin_bit_cast(short, h);
        if (SPLIT) lo[j] = __builtin_bit_cast(short, __float2bfloat16(v[j] - __bfloat162float(h)));
    }
}

__device__ __forceinline__ acc16_t mfma_bf16(frag8_t a, frag8_t b, acc16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- buffer (SRD) accessors.  Every gather / store of the scan goes through a wave-uniform buffer
// descriptor + a 32-bit per-lane BYTE offset: no 64-bit address arithmetic in the VALU (the kernels are
// VALU-bound), row strides ride in the scalar `soffset`, and an element outside the tensor is given an
// out-of-range vector offset which the hardware range check drops: no branches around the stores.
template <typename T> struct RawOf { typedef unsigned short type; };
template <> struct RawOf<float> { typedef unsigned type; };

// An activation element is loaded as RAW bits and converted where it is consumed: a conversion written at the
// load site makes hipcc wait for the just-issued gather (the whole memory latency, every tile).
template <typename T> __device__ __forceinline__ typename RawOf<T>::type buf_load_raw(__amdgpu_buffer_rsrc_t r, unsigned voff);
template <> __device__ __forceinline__ unsigned buf_load_raw<float>(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
}
template <> __device__ __forceinline__ unsigned short buf_load_raw<__hip_bfloat16>(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, 0);
}
template <> __device__ __forceinline__ unsigned short buf_load_raw<__half>(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, 0);
}
template <typename T> __device__ __forceinline__ float raw_to_f(unsigned v);
template <> __device__ __forceinline__ float raw_to_f<float>(unsigned v) { return __builtin_bit_cast(float, v); }
template <> __device__ __forceinline__ float raw_to_f<__hip_bfloat16>(unsigned v)
{
    return __builtin_bit_cast(float, v << 16);   // v arrives zero-extended
}
template <> __device__ __forceinline__ float raw_to_f<__half>(unsigned v)
{
    return (float)__builtin_bit_cast(_Float16, (unsigned short)v);
}
template <typename TY>
__device__ __forceinline__ void buf_store_elem(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v);
template <>
__device__ __forceinline__ void buf_store_elem<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
template <>
__device__ __forceinline__ void buf_store_elem<__hip_bfloat16>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff,
                                                               float v)
{
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, __float2bfloat16(v)), r, voff, soff, 0);
}
template <>
__device__ __forceinline__ void buf_store_elem<__half>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), r, voff, soff, 0);
}

// Two elements of one lane (same channel, two positions): ONE packed conversion, the lower half stored by
// buffer_store_short and the upper half by buffer_store_short_d16_hi -- a conversion per element would cost the VALU-bound
// scan kernels 8 more instructions per tile.  hipcc has no pattern from (pk >> 16) to the d16_hi store, hence the asm;
// device pass only (the host pass sees the generic pair of stores, which it never runs).
template <typename TY>
__device__ __forceinline__ void buf_store_pair(__amdgpu_buffer_rsrc_t r, unsigned v0, unsigned s0, unsigned v1, unsigned s1,
                                               float a, float b)
{
    buf_store_elem<TY>(r, v0, s0, a);
    buf_store_elem<TY>(r, v1, s1, b);
}
#if defined(__HIP_DEVICE_COMPILE__)
template <>
__device__ __forceinline__ void buf_store_pair<__hip_bfloat16>(__amdgpu_buffer_rsrc_t r, unsigned v0, unsigned s0, unsigned v1,
                                                               unsigned s1, float a, float b)
{
    unsigned pk;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
    asm volatile("buffer_store_short %0, %1, %2, %3 offen" ::"v"(pk), "v"(v0), "s"(r), "s"(s0) : "memory");
    asm volatile("buffer_store_short_d16_hi %0, %1, %2, %3 offen" ::"v"(pk), "v"(v1), "s"(r), "s"(s1) : "memory");
}
template <>
__device__ __forceinline__ void buf_store_pair<__half>(__amdgpu_buffer_rsrc_t r, unsigned v0, unsigned s0, unsigned v1,
                                                       unsigned s1, float a, float b)
{
    unsigned pk;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
    asm volatile("buffer_store_short %0, %1, %2, %3 offen" ::"v"(pk), "v"(v0), "s"(r), "s"(s0) : "memory");
    asm volatile("buffer_store_short_d16_hi %0, %1, %2, %3 offen" ::"v"(pk), "v"(v1), "s"(r), "s"(s1) : "memory");
}
#endif

template <typename T, int NK>
struct TileOps {
    float araw[NK][8];               // x_dbl row of MY position (lane & 31), ranks 16kk + 8hi .. +7
    float bv, cv;                    // B, C of MY position
    typename RawOf<T>::type u[16];   // x of my 16 (position, channel) elements, raw bits
};

// One wave's view of a (direction, 32-channel tile): loop-invariant operands + the per-tile steps shared by
// the chained and the wave-segment kernels.  Lane (r32, hi): channel r32 of the tile; as a POSITION owner
// (gathers, staging) position r32; as an ELEMENT owner the 16 positions POS(r) = (r&3) + 8(r>>2) + 4hi.
//
// Arithmetic per element (the kernels are VALU-bound, so every instruction here was counted):
//   x' = (dt_raw + bias) * log2(e)   straight out of the MFMA: log2(e) is folded into the dt_w fragments and
//                                    the accumulator starts at bias * log2(e)
//   t  = med3(log2(1 + exp2(x')), x', 128)             = softplus(dt_raw + bias) * log2(e)
//        (the reference's threshold form, x > 20 -> x, equals this to below fp32 resolution)
//   a  = exp2(t * A)                                   = exp(dt * A)
//   bb = t * (B*ln2 * u)                               = dt * B * u      (ln2 folded into the staged B)
// i.e. 3 transcendentals + one med3 and 4 packed-able mul/add per element.  Positions past the end of the
// sequence start the accumulator at -1e30 instead: t = 0 exactly, hence a = 1, bb = 0 (the identity).
// lower-half / upper-half broadcast of a wave-wide value (gfx950 v_permlane32_swap)
__device__ __forceinline__ void half_swap(float x, float &lower, float &upper)
{
    const unsigned b = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    const unsigned lo = r[0], up = r[1];
    lower = __builtin_bit_cast(float, lo);
    upper = __builtin_bit_cast(float, up);
}

template <typename T, int NK, bool SPLIT>
struct ScanWave {
    frag8_t wh[NK], wl[NK];  // dt_w[k][channel][16kk + 8hi + j] * log2(e), bf16 hi (+ lo residual)
    float bias2, An, Dk;
    unsigned roff[NK];       // byte offset of my 8-rank run inside an x_dbl group
    unsigned bcoff;          // byte offset of (B, C) inside the group
    unsigned cx;             // byte offset of my channel inside an x row
    unsigned xrow, prow;     // bytes per x row / per x_dbl row
    int hi, r32, R;
    float *st;               // this wave's LDS stage: [0] x-row byte offsets, [1] B*ln2, [2] C per position

    // a_log (training): `Aneg` holds the PARAMETER A_logs and the kernel forms A = -exp(A_logs) (vmamba.py:246) itself --
    // one exp per lane at start-up instead of an exp and a negation launch per block and step (and two more for their gradient)
    __device__ __forceinline__ void init(const float *dt_w, const float *dt_bias, const float *Aneg, const float *Ds,
                                         long kd, int R_, int RG, int PC, int D, int lane, int cc_, float *st_,
                                         int a_log = 0)
    {
        r32 = lane & 31;
        hi = lane >> 5;
        R = R_;
        st = st_;
        const int R8 = xdbl_rank_pad(R);
        const float *wrow = dt_w + kd * R;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = kk * 16 + hi * 8 + j;
                t[j] = r < R ? wrow[r] * 1.44269504088896f : 0.f;
            }
            pack_frag<SPLIT>(t, wh[kk], wl[kk]);
            // a run past the padded ranks meets all-zero dt_w fragments: read run 0 instead of predicating
            const int r0 = kk * 16 + hi * 8;
            roff[kk] = (unsigned)((r0 < R8 ? r0 : 0) * 4);
        }
        bias2 = dt_bias[kd] * 1.44269504088896f;
        An = a_log ? -__expf(Aneg[kd]) : Aneg[kd];
        Dk = Ds[kd];
        bcoff = (unsigned)(R8 * 4);
        cx = (unsigned)(cc_ * (int)sizeof(T));
        xrow = (unsigned)(D * (int)sizeof(T));
        prow = (unsigned)(PC * 4);
    }

    // my 16 elements' values of a per-position array
    __device__ __forceinline__ void read_stage4(int which, float (&out)[16]) const
    {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(st + which * kTP + 8 * g + 4 * hi);
            out[4 * g + 0] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
        }
    }

    // issue every gather of one tile; idxv = pixel index of MY position in that tile (always a valid pixel)
    __device__ __forceinline__ void fetch(__amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rp, int idxv,
                                          TileOps<T, NK> &o) const
    {
        if (hi == 0) st[r32] = __builtin_bit_cast(float, (unsigned)idxv * xrow);
        __builtin_amdgcn_wave_barrier();
        float xo[16];
        read_stage4(0, xo);
        __builtin_amdgcn_wave_barrier();
        const unsigned pr = (unsigned)idxv * prow;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            // (bit_cast the WHOLE vector: hipcc miscompiles __builtin_bit_cast(float, vec[j]) to element 0)
            const v4f v0 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rp, pr + roff[kk], 0, 0));
            const v4f v1 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rp, pr + roff[kk] + 16, 0, 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o.araw[kk][j] = v0[j];
                o.araw[kk][4 + j] = v1[j];
            }
        }
        const v2f bc = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rp, pr + bcoff, 0, 0));
        o.bv = bc.x;
        o.cv = bc.y;
#pragma unroll
        for (int r = 0; r < 16; ++r) o.u[r] = buf_load_raw<T>(rx, __builtin_bit_cast(unsigned, xo[r]) + cx);
    }

    // publish B*ln2 and C of the current tile's positions, read back my 16 of each
    __device__ __forceinline__ void stage_bc(const TileOps<T, NK> &cur, float (&Bp)[16], float (&Cp)[16], bool want_c) const
    {
        if (hi == 0) {
            st[kTP + r32] = cur.bv * 0.693147180559945f;
            st[2 * kTP + r32] = cur.cv;
        }
        __builtin_amdgcn_wave_barrier();
        read_stage4(1, Bp);
        if (want_c) read_stage4(2, Cp);
    }

    // dt_proj on the matrix core + per-element decay / input terms.  nvalid = positions of this tile
    // inside the sequence (>= 32: all of them)
    // UF_READY: uf already holds the 16 inputs as floats (the LDS-DMA kernel reads bf16 straight into the upper half of
    // a zeroed register), cur.u is not looked at
    // BIAS_MFMA (LDS-DMA kernel, dt_rank padded to 8 of the MFMA's 16 k slots): the bias rides in two spare k slots -- the
    // upper half-wave supplies A = (1, 1, 0..) and B = (bf16 hi, bf16 lo of bias * log2 e, 0..) -- so the accumulator starts
    // at the inline constant 0 instead of 16 register copies of the bias per tile (2^-17 relative on the bias)
    template <bool UF_READY = false, bool BIAS_MFMA = false>
    __device__ __forceinline__ void terms(const TileOps<T, NK> &cur, const float (&Bp)[16], int nvalid, float (&a)[16],
                                          float (&bb)[16], float (&uf)[16], float *tl = nullptr) const
    {
        if constexpr (!UF_READY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) uf[r] = raw_to_f<T>(cur.u[r]);
        }
        const float b0 = BIAS_MFMA ? 0.f : bias2;
        acc16_t acc;
        {
            frag8_t ah, al;
            pack_frag<SPLIT>(cur.araw[0], ah, al);
            // two MFMA sites: the full tile's accumulator operand is a constant splat (the inline 0 with BIAS_MFMA), not a
            // register tuple that a merged path would have to fill per tile
            if (nvalid < kTP) {  // wave-uniform: only the last tile of a sequence / segment
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (r & 3) + 8 * (r >> 2) + 4 * hi < nvalid ? b0 : -1.0e30f;
                acc = mfma_bf16(ah, wh[0], acc);
            } else {
                acc16_t c0;
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] = b0;
                acc = mfma_bf16(ah, wh[0], c0);
            }
            if (SPLIT) {
                acc = mfma_bf16(ah, wl[0], acc);
                acc = mfma_bf16(al, wh[0], acc);
            }
        }
#pragma unroll
        for (int kk = 1; kk < NK; ++kk) {
            frag8_t ah, al;
            pack_frag<SPLIT>(cur.araw[kk], ah, al);
            acc = mfma_bf16(ah, wh[kk], acc);
            if (SPLIT) {
                acc = mfma_bf16(ah, wl[kk], acc);
                acc = mfma_bf16(al, wh[kk], acc);
            }
        }
        const v2f one = {1.f, 1.f}, av = {An, An};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const v2f xr = {acc[r], acc[r + 1]};
            // log2(1 + 2^x') == x' to within an fp32 ulp once x' > 30, but 2^x' overflows to +inf from x' = 128 on.  The
            // median of (log2(1 + 2^x'), x', 128) is the first below 128 and x' itself beyond -- ONE v_med3_f32 where a
            // compare + select pair stood (the kernel is VALU-issue bound: 16 instructions per tile)
            v2f z = {__builtin_amdgcn_exp2f(xr.x), __builtin_amdgcn_exp2f(xr.y)};
            z = z + one;
            const v2f t = {__builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(z.x), xr.x, 128.f),
                           __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(z.y), xr.y, 128.f)};
            const v2f e = t * av;
            a[r] = __builtin_amdgcn_exp2f(e.x);
            a[r + 1] = __builtin_amdgcn_exp2f(e.y);
            const v2f bu = v2f{Bp[r], Bp[r + 1]} * v2f{uf[r], uf[r + 1]};
            const v2f bt = t * bu;
            bb[r] = bt.x;
            bb[r + 1] = bt.y;
            if (tl) {   // (backward only; the pointer is a compile-time null in the forward kernels)
                tl[r] = t.x;
                tl[r + 1] = t.y;
            }
        }
    }

    // (decay, state) prefix entering each of my 4 runs, relative to the tile start, and the tile aggregate.
    // Runs of 4 consecutive positions; runs of the two half-waves interleave (lower half first).
    __device__ __forceinline__ void prefix(const float (&a)[16], const float (&bb)[16], float (&preA)[4],
                                           float (&preH)[4], float &tA, float &tH) const
    {
        float sa[4], sh[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float pa = a[4 * g], ph = bb[4 * g];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                ph = fmaf(a[4 * g + q], ph, bb[4 * g + q]);
                pa *= a[4 * g + q];
            }
            sa[g] = pa;
            sh[g] = ph;
        }
        tA = 1.f;
        tH = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // v_permlane32_swap(X, X): first result = X's lower half in both halves, second = its upper half --
            // both half-waves' run aggregates in one instruction each, no LDS permute, no selects
            float ea, eh, fa, fh;   // run 2g (lower half-wave) / run 2g+1 (upper half-wave)
            half_swap(sa[g], ea, fa);
            half_swap(sh[g], eh, fh);
            const float midH = fmaf(ea, tH, eh), midA = ea * tA;
            preA[g] = hi ? midA : tA;
            preH[g] = hi ? midH : tH;
            tH = fmaf(fa, midH, fh);
            tA = fa * midA;
        }
    }

    // replay my 16 steps from the state entering the tile and stream y = C h + D u out.
    // yv: byte offset of (first position of the tile + 4hi, my channel); row strides ride in the scalar
    // soffset.  MASKED (ragged tile / channel edge, wave-uniform choice): an element outside the tensor gets
    // the out-of-range VECTOR offset, which the range check drops whatever the scalar offset is.
    // PAIR (the LDS-DMA kernel only): two elements per conversion, stored by inline-asm buffer stores.  hipcc's vmcnt
    // bookkeeping does not see those, which is only sound where every wait on vector memory is hand-counted.
    template <typename TY, bool MASKED, bool PAIR = false>
    __device__ __forceinline__ void replay(__amdgpu_buffer_rsrc_t ry, unsigned yv, unsigned yrow, const float (&uf)[16],
                                           const float (&a)[16], const float (&bb)[16], const float (&Cp)[16],
                                           const float (&preA)[4], const float (&preH)[4], float hin, bool cok,
                                           int nvalid) const
    {
        const v2f dv = {Dk, Dk};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float hh = fmaf(preA[g], hin, preH[g]);
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const int r = 4 * g + q;
                const float h0 = fmaf(a[r], hh, bb[r]);
                const float h1 = fmaf(a[r + 1], h0, bb[r + 1]);
                hh = h1;
                const v2f o = v2f{Cp[r], Cp[r + 1]} * v2f{h0, h1} + dv * v2f{uf[r], uf[r + 1]};
                const int p0 = q + 8 * g + 4 * hi;  // position of element r inside the tile
                const unsigned v0 = !MASKED || (cok && p0 < nvalid) ? yv : kOutOfRange;
                const unsigned v1 = !MASKED || (cok && p0 + 1 < nvalid) ? yv : kOutOfRange;
                if constexpr (PAIR) {
                    buf_store_pair<TY>(ry, v0, (unsigned)(q + 8 * g) * yrow, v1, (unsigned)(q + 1 + 8 * g) * yrow, o.x, o.y);
                } else {
                    buf_store_elem<TY>(ry, v0, (unsigned)(q + 8 * g) * yrow, o.x);
                    buf_store_elem<TY>(ry, v1, (unsigned)(q + 1 + 8 * g) * yrow, o.y);
                }
            }
        }
    }
};

// Carry hand-over between the W waves that share a sequence, WITHOUT a workgroup barrier (r03).
// Tile t = s W + w is owned by wave w in step s; the state entering it is the state leaving tile t - 1, owned by wave w - 1
// (wave W - 1 of the previous step for w = 0).  Every wave publishes the state LEAVING its tile -- value, then a tag t + 1,
// two LDS writes in program order -- into its own mailbox and reads its predecessor's mailbox -- tag, then value -- until
// the tag matches.  The hand-over is one fused multiply-add per wave (h_out = A_tile h_in + H_tile) instead of a W-step
// fold behind a barrier, and, more important, the waves of a workgroup are no longer in lock-step: measured on the Helix
// launch (ablation builds) no single phase of the lock-step kernel -- stores, barrier, fold, LDS-DMA,
// transcendentals -- was worth more than 10 % of its time; the step was a chain of phases every wave of the CU entered
// together.  Decoupled, a wave does phase 1 of its next tile while its successors still wait for their carry.
// No write-after-read hazard: wave w rewrites its mailbox for step s + 1 only after it has the carry of that step, which
// (through waves w + 1 .. W - 1 of step s and waves 0 .. w - 1 of step s + 1) descends from wave w + 1 having READ the step-s
// value.  Forward progress: all waves of a workgroup are resident and run the same number of steps; the spin sleeps between
// polls and is bounded (a run that ever hit the bound would end with NaNs in its output, not hang the GPU).
// The device error word (common.h): its address sits in a __device__ variable of this translation unit, written once by the host
// (set_mailbox_err_word, called from mailbox_ctl() before the first launch that can raise it).  A wave whose poll runs out only
// drops a flag into the third row of the mailbox it was polling (one ds_write on a path that is never taken); every wave looks at
// that row ONCE, after its last tile, and raises the word from there (CarryLink::report).  Carried through the tile loop as a
// kernel argument -- or called out of line from it -- the report cost the Helix launch 35 scalar and 8 vector registers (97 / 128
// instead of 62 / 120) and ~1.4 % of a forward (same-box A/B against the r03 library, profiles/r04_ab_lib.txt).
__device__ unsigned *g_mailbox_err_word = nullptr;

constexpr int kMailboxPolls = 1 << 20;   // ~0.1 s of polling (s_sleep between polls) before a wave gives up

struct CarryLink {
    unsigned rd, wr;   // LDS byte addresses of (value[kTP], tag[kTP]) of my predecessor's / my own mailbox, at my channel
    int skip;          // tests only (MailboxCtl, common.h): the tile whose hand-over is withheld, -1 = none

    // reverse: the chain runs from the last wave to the first (the adjoint sweep of the backward kernel)
    __device__ __forceinline__ void init(float *box /* [W][3][kTP] */, int wv, int W, int r32, MailboxCtl c, bool reverse = false)
    {
        skip = c.skip;
        const int pred = reverse ? (wv == W - 1 ? 0 : wv + 1) : (wv == 0 ? W - 1 : wv - 1);
        rd = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(box + (pred * 3) * kTP + r32);
        wr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(box + (wv * 3) * kTP + r32);
    }
    // after the wave's last tile: a time-out flag in the mailbox I polled -> the library's device error word
    __device__ __forceinline__ void report() const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned f;
        asm volatile("ds_read_b32 %0, %1 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(rd) : "memory");
        if (__builtin_amdgcn_ballot_w64(f != 0u) != 0) {
            unsigned *p = g_mailbox_err_word;
            if (p && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(p, TRAMBA_DEVERR_MAILBOX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#endif
    }
    // state leaving tile `tile - 1` (tile >= 1, in chain order); 0 for the first tile of the chain.  `base`: tag offset
    // of this chain (a kernel that runs two chains through the same mailboxes gives the second one tags above the first's)
    __device__ __forceinline__ float acquire(int tile, unsigned base = 0u) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (tile == 0) return 0.f;
        float v;
        unsigned t;
        int guard = 0;
        // (the budget is a COMPILE-TIME constant: with a run-time bound hipcc compiled the Helix launch to 93 scalar / 128
        //  vector registers instead of 62 / 124 and the forward lost 1.4 %, profiles/r04_ab_lib.txt)
        for (; guard < kMailboxPolls; ++guard) {
            asm volatile("ds_read_b32 %0, %2 offset:128\n\tds_read_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t), "=&v"(v)
                         : "v"(rd)
                         : "memory");
            if (__builtin_amdgcn_ballot_w64(t != base + (unsigned)tile) == 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        // A wave that ever ran out of polls (~0.1 s at the default budget) hands on NaN -- the launch ends, and its output cannot
        // pass for a result -- and raises the library's device error word (host-mapped memory, one system-scope atomic on this
        // path only): the next library call that finds it set returns TRAMBA_ERR_HIP (TRAMBA_LAUNCH_CHECK; no sync is added).
        if (__builtin_expect(guard < kMailboxPolls, 1)) return v;
        asm volatile("ds_write_b32 %0, %1 offset:256" ::"v"(rd), "v"(1u) : "memory");
        return __builtin_nanf("");
#else
        return 0.f;
#endif
    }
    // state leaving tile `tile` (every lane of the wave holds its channel's value; the lower half-wave writes)
    __device__ __forceinline__ void publish(float h, int tile, int hi, unsigned base = 0u) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (hi == 0 && tile != skip) {      // (skip: TRAMBA_TUNE_MAILBOX_SKIP, a test withholding one hand-over; -1 in production)
            const unsigned tg = base + (unsigned)tile + 1u;
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:128" ::"v"(wr), "v"(h), "v"(tg) : "memory");
        }
#endif
    }
};

// XCD-aware work order of the chained kernels (xcd_work_item, common.h).  With the plain (x = channel tile) order the
// channel tiles of ONE sequence sit on 8 different XCDs: every gathered token row is then fetched as 64-byte halves of
// 128-byte lines by two different L2s (PMC: 2.2x the algorithmic bytes on the Helix launch, 1.09x with the remap), and
// the x_proj rows 8 times.  Channel tile fastest inside an XCD's run: the tiles sharing lines run side by side on one
// XCD and the second reader hits in its L2.

// SAVE (training): the state entering every tile also goes to hst (B, K, NTA, D) f32, NTA = ceil(L / 32) + kMaxW -- the
// backward kernel then skips its first sweep, which recomputes exactly these values.
template <typename T, typename TY, int NK, bool SPLIT, bool SAVE = false>
__global__ __launch_bounds__(kMaxW * kWave) void ss2d_scan_cl_kernel(
    const T *__restrict__ x, const float *__restrict__ xdbl, const int32_t *__restrict__ table,
    const float *__restrict__ dt_w, const float *__restrict__ dt_bias, const float *__restrict__ Aneg,
    const float *__restrict__ Ds, TY *__restrict__ ys, int L, int D, int K, int R, int W, float *__restrict__ hst, int a_log,
    MailboxCtl mctl)
{
    __shared__ float mbox[kMaxW][3][kTP];   // carry mailboxes (CarryLink): value, tag, time-out flag per wave and channel
    __shared__ __attribute__((aligned(16))) float stage[kMaxW][3][kTP];

    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, hi = lane >> 5;
    unsigned ct_, k_, b_;
    xcd_work_item(ct_, k_, b_);
    const int k = (int)k_, b = (int)b_, ctile = (int)ct_;
    const int c = ctile * kTP + r32;
    const bool cok = c < D;
    const int cc_ = cok ? c : D - 1;
    const int RG = xdbl_group_stride(R);
    const int PC = K * RG;

    ScanWave<T, NK, SPLIT> w;
    w.init(dt_w, dt_bias, Aneg, Ds, (long)k * D + cc_, R, RG, PC, D, lane, cc_, &stage[wv][0][0], a_log);
    CarryLink link;
    link.init(&mbox[0][0][0], wv, W, r32, mctl);

    // wave-uniform descriptors (host guarantees every extent < 2^31 bytes)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (long)b * L * D, (unsigned)L * w.xrow);
    const __amdgpu_buffer_rsrc_t rp =
        make_rsrc(xdbl + (long)b * L * PC + (long)k * RG, ((unsigned)(L - 1) * PC + RG) * 4u);
    const unsigned yrow = (unsigned)D * (unsigned)sizeof(TY);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(ys + ((long)b * K + k) * L * D, (unsigned)L * yrow);
    const int32_t *tk = table + (long)k * L;

    const int span = W * kTP;
    const int nsuper = (L + span - 1) / span;
    const int last = nsuper - 1;

    auto load_idx = [&](int s) -> int {  // unconditional (clamped): predicated loads serialise the pipeline
        const int l = (s < last ? s : last) * span + wv * kTP + r32;
        return tk[l < L ? l : L - 1];
    };

    // Gather pipeline: tile s computes while tiles s+1 (and s+2 when registers allow, NK <= 2) are in
    // flight.  The operand stages form a STATIC ring (tile t lives in ops[t % NS]) and the tile loop is
    // unrolled by NS: rotating registers that have loads in flight, or predicating the loads, makes
    // hipcc wait vmcnt(0) every tile.  Loads past the end of the sequence are clamped, not skipped.
    constexpr int AHEAD = NK <= 2 ? 2 : 1;
    constexpr int NS = AHEAD + 1;
    TileOps<T, NK> ops[NS];
    // The index vectors run TWO further tiles ahead in their own static ring (idxr[t % NS]): vmcnt retires in
    // order, so an index loaded in the same step as the gathers it follows, and needed one step later, would
    // drain those gathers every tile and void the prefetch.
    // Prologue: every index load is issued BEFORE the first gathers, so the loop is entered with the same
    // in-flight picture the back-edge has (a younger index load at entry would make hipcc drain there).
    int idxr[NS], idx0[AHEAD];
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) idx0[t] = load_idx(t);
#pragma unroll
    for (int t = 0; t < NS; ++t) idxr[(AHEAD + t) % NS] = max(load_idx(AHEAD + t), 0);   // tiles AHEAD .. AHEAD + NS - 1
    // (the max() is a no-op on a valid table; it makes the ring values ARRIVED at loop entry, so hipcc merges the
    //  entry and back-edge wait states without a drain)
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) w.fetch(rx, rp, idx0[t], ops[t]);

    // mailbox tags start at 0 = "nothing published" (tile t publishes tag t + 1)
    for (int i = threadIdx.x; i < 3 * kMaxW * kTP; i += blockDim.x) (&mbox[0][0][0])[i] = 0.f;
    __syncthreads();
    const bool cfull = ctile * kTP + kTP <= D;  // block-uniform: no channel masking needed
    // one super-chunk; STI = ring slot of tile s (compile-time: the ring must be indexed statically)
    auto step = [&](auto sti_c, int s) {
        constexpr int sti = decltype(sti_c)::value;
        TileOps<T, NK> &cur = ops[sti];
        const int l0 = s * span + wv * kTP;
        float Bp[16], Cp[16];
        w.stage_bc(cur, Bp, Cp, true);
        // tile s + AHEAD goes into the stage tile s - 1 just vacated
        w.fetch(rx, rp, idxr[(sti + AHEAD) % NS], ops[(sti + AHEAD) % NS]);
        idxr[(sti + AHEAD) % NS] = load_idx(s + AHEAD + NS);   // consumed NS steps from now

        float a[16], bb[16], uf[16], preA[4], preH[4], runA, runH;
        w.terms(cur, Bp, L - l0, a, bb, uf);
        w.prefix(a, bb, preA, preH, runA, runH);

        // ---- no barrier: the state entering my tile comes from my predecessor's mailbox, the state leaving it goes into mine
        const float hin = link.acquire(s * W + wv);
        link.publish(fmaf(runA, hin, runH), s * W + wv, hi);
        if constexpr (SAVE) {
            const long nta = (L + kTP - 1) / kTP + kMaxW;
            if (hi == 0 && cok && l0 < L) hst[(((long)b * K + k) * nta + (s * W + wv)) * D + c] = hin;
        }
        const unsigned yv = (unsigned)((l0 + 4 * hi) * D + cc_) * (unsigned)sizeof(TY);
        if (cfull && l0 + kTP <= L) {  // wave-uniform: ragged only at the sequence / channel edge
            w.template replay<TY, false>(ry, yv, yrow, uf, a, bb, Cp, preA, preH, hin, true, kTP);
        } else {
            w.template replay<TY, true>(ry, yv, yrow, uf, a, bb, Cp, preA, preH, hin, cok, L - l0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // Whole ring trips in the loop, the 0..NS-1 remaining super-chunks after it: an exit from the MIDDLE of the
    // unrolled body (a `break`) makes hipcc drain every outstanding gather at the loop head.
    int s0 = 0;
    for (; s0 + NS <= nsuper; s0 += NS) {
        step(std::integral_constant<int, 0>{}, s0);
        step(std::integral_constant<int, 1>{}, s0 + 1);
        if constexpr (NS > 2) step(std::integral_constant<int, 2>{}, s0 + 2);
    }
    if (s0 < nsuper) step(std::integral_constant<int, 0>{}, s0);
    if constexpr (NS > 2) {
        if (s0 + 1 < nsuper) step(std::integral_constant<int, 1>{}, s0 + 1);
    }
    link.report();
}

// ---------------------------------------------------------------------------------------------
// Chained form on LDS-DMA staged operands (16-bit activations, dt_rank padded to R8 in {8, 16, 32}; W = 16 or 8 waves per
// sequence).  The register-ring kernel above holds two tiles of gathered operands in VGPRs (~180 registers: 2 waves per SIMD,
// and the SQ counters show each SIMD's VALU busy only ~55 % of the time at that occupancy).  Here the gathers of the next
// tile -- 32 token rows of 64 bytes and 32 x_dbl rows of (R8 + 4) floats per wave -- are written straight into a
// wave-private LDS slot by buffer_load ... lds (no VGPRs): the kernel fits 128 registers = 4 waves per SIMD (16 waves per
// workgroup, or two workgroups of 8 per CU).
//   step s:  s_waitcnt vmcnt(16)      tile s has landed in my slot, the index vector of tile s + 1 in idxbuf: everything but
//                                     the 16 y stores of tile s - 1, the only vector-memory ops issued after those DMAs (the
//                                     counter retires in order)
//            read the slot            hand-written ds_reads into registers (the operands of ONE tile: 8 NK + 2 + 16 values)
//            DMA tile s + 1           into the SAME slot (its reads have completed), index vector of tile s + 2 into idxbuf
//            compute, barrier, fold, replay, 16 stores
// No vector-memory LOAD the compiler can see lives in the loop, and every read of DMA-written LDS is hand-written: hipcc
// orders a load result it can see, or an LDS read that may alias a pending DMA, with a vmcnt that -- in order -- also waits
// for most of the previous tile's y stores; with all waves of a workgroup leaving the barrier together the CU then idled for
// a store round trip every step (first version: 83-90 us on the Helix launch, SQ_WAIT_ANY 49 % of wave-cycles; now 77-82).
#define TRAMBA_LDS_RD_(INSTR, OUT, ADDR, OFF) asm volatile(INSTR " %0, %1 offset:" #OFF : "=v"(OUT) : "v"(ADDR) : "memory")

template <typename T, typename TY, int NK, int R8, int W, bool SAVE = false>
__global__ __launch_bounds__(W * kWave, W == 16 ? 1 : 4) void ss2d_scan_dma_kernel(
    const T *__restrict__ x, const float *__restrict__ xdbl, const int32_t *__restrict__ table,
    const float *__restrict__ dt_w, const float *__restrict__ dt_bias, const float *__restrict__ Aneg,
    const float *__restrict__ Ds, TY *__restrict__ ys, int L, int D, int K, int R, float *__restrict__ hst, int a_log,
    MailboxCtl mctl)
{
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only: the host pass needs this kernel's launch stub, not its body, and hipcc
                                      // silently drops the stub of a kernel template whose body holds vector-register asm
    static_assert(sizeof(T) == 2, "16-bit activations");
    static_assert((R8 == 8 && NK == 1) || (R8 == 16 && NK == 1) || (R8 == 32 && NK == 2), "padded rank / MFMA steps");
    constexpr int kRowP = (R8 + 4) * 4;         // bytes of one x_dbl group: R8 ranks + B, C + 2 pad
    constexpr int kCPR = kRowP / 16;            // 16-byte chunks per row: 3, 5, 9
    constexpr int kNCH = kTP * kCPR;            // chunks per tile: 96, 160, 288
    constexpr int kNP = (kNCH + kWave - 1) / kWave;   // 64-lane DMA pieces: 2, 3, 5 (the last one overshoots into padding)
    constexpr int kUB = kTP * kTP * 2;          // token tile: 32 positions x 32 channels x 2 bytes
    constexpr int kSlot = kUB + kNP * 1024;
    __shared__ float mbox[W][3][kTP];   // carry mailboxes (CarryLink): value, tag, time-out flag
    __shared__ __attribute__((aligned(16))) float stage[W][3][kTP];
    __shared__ __attribute__((aligned(16))) unsigned char ring[W][kSlot];
    __shared__ __attribute__((aligned(16))) int idxbuf[W][kWave];   // the index vector of the next tile to fetch, by DMA too

    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, hi = lane >> 5;
    unsigned ct_, k_, b_;
    xcd_work_item(ct_, k_, b_);
    const int k = (int)k_, b = (int)b_, ctile = (int)ct_;
    const int c = ctile * kTP + r32;            // D % 32 == 0 (host-checked): every channel exists
    const int RG = R8 + 4;
    const int PC = K * RG;

    ScanWave<T, NK, false> w;
    w.init(dt_w, dt_bias, Aneg, Ds, (long)k * D + c, R, RG, PC, D, lane, c, &stage[wv][0][0], a_log);
    // rank 8: k slots 8..15 of the one MFMA are spare (the upper half-wave's fragments) -- the bias goes there (terms())
    constexpr bool kBiasMfma = R8 == 8;
    __shared__ __attribute__((aligned(16))) float onesrow[8];
    if constexpr (kBiasMfma) {
        if (threadIdx.x < 8) onesrow[threadIdx.x] = threadIdx.x < 2 ? 1.f : 0.f;
        const __hip_bfloat16 bh = __float2bfloat16(w.bias2);
        const __hip_bfloat16 bl = __float2bfloat16(w.bias2 - __bfloat162float(bh));
        frag8_t f = {0, 0, 0, 0, 0, 0, 0, 0};
        f[0] = __builtin_bit_cast(short, bh);
        f[1] = __builtin_bit_cast(short, bl);
        if (hi) w.wh[0] = f;
    }
    CarryLink link;
    link.init(&mbox[0][0][0], wv, W, r32, mctl);
    for (int i = threadIdx.x; i < 3 * W * kTP; i += blockDim.x) (&mbox[0][0][0])[i] = 0.f;   // tags: nothing published
    __syncthreads();   // the only workgroup barrier of the kernel

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (long)b * L * D, (unsigned)L * w.xrow);
    const __amdgpu_buffer_rsrc_t rp =
        make_rsrc(xdbl + (long)b * L * PC + (long)k * RG, ((unsigned)(L - 1) * PC + RG) * 4u);
    const unsigned yrow = (unsigned)D * (unsigned)sizeof(TY);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(ys + ((long)b * K + k) * L * D, (unsigned)L * yrow);
    const int32_t *tk = table + (long)k * L;

    const int span = W * kTP;
    const int nsuper = (L + span - 1) / span;
    const int last = nsuper - 1;
    const __amdgpu_buffer_rsrc_t rt = make_rsrc(tk, (unsigned)L * 4u);
    auto idx_off = [&](int s) -> unsigned {     // byte offset of MY entry of tile s's index vector (clamped past the end)
        const int l = (s < last ? s : last) * span + wv * kTP + r32;
        return (unsigned)(l < L ? l : L - 1) * 4u;
    };

    // loop-invariant lane maps of the DMA pieces (16 bytes per lane, LDS destination = piece base + 16 * lane):
    //   token tile  : 2 pieces; piece j, lane i -> position 16j + i/4, bytes 16*(i%4) .. +15 of this channel tile's 64
    //   x_dbl rows  : kNCH chunks of 16 bytes, kCPR per position; piece j = chunks 64j .. 64j + 63 (past the end: the last
    //                 chunk again, landing in the slot's padding)
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned char *slot = &ring[wv][0];
    int *ib = &idxbuf[wv][0];
    const unsigned iba = (unsigned)(size_t)(__attribute__((address_space(3))) int *)ib;
    const unsigned ucol = (unsigned)(ctile * kTP * 2 + (lane & 3) * 16);
    unsigned ia[2 + kNP];                       // LDS addresses of the index entries my pieces need
    unsigned ppart[kNP];
    ia[0] = iba + 4u * (unsigned)(lane >> 2);
    ia[1] = iba + 4u * (unsigned)(16 + (lane >> 2));
#pragma unroll
    for (int j = 0; j < kNP; ++j) {
        const int ch = 64 * j + lane < kNCH ? 64 * j + lane : kNCH - 1;
        const int pos = ch / kCPR;
        ia[2 + j] = iba + 4u * (unsigned)pos;
        ppart[j] = (unsigned)(ch - pos * kCPR) * 16u;
    }

    auto dma = [&](int s_idx) {   // operands of the tile whose index vector sits in idxbuf -> my slot, then the index
        unsigned iv[7] = {0, 0, 0, 0, 0, 0, 0};   // vector of tile s_idx -> idxbuf
#pragma unroll
        for (int j = 0; j < 2 + kNP; ++j) TRAMBA_LDS_RD_("ds_read_b32", iv[j], ia[j], 0);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(iv[0]), "+v"(iv[1]), "+v"(iv[2]), "+v"(iv[3]), "+v"(iv[4]), "+v"(iv[5]), "+v"(iv[6])
                     :
                     : "memory");
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(slot), 16, iv[0] * w.xrow + ucol, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(slot + 1024), 16, iv[1] * w.xrow + ucol, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < kNP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_void *)(slot + kUB + 1024 * j), 16, iv[2 + j] * w.prow + ppart[j],
                                                     0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (lds_void *)ib, 4, idx_off(s_idx), 0, 0, 0);
    };
    // my x_dbl row: ranks 16 kk + 8 hi .. +7 (a run past the padded ranks meets all-zero dt_w fragments: run 0 instead),
    // then (B, C); my 16 token elements: positions (r&3) + 8 (r>>2) + 4 hi of channel r32
    const unsigned prow_a = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)(slot + kUB + r32 * kRowP);
    const unsigned pa0 = kBiasMfma && hi
                             ? (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)&onesrow[0]
                             : prow_a + (unsigned)((hi * 8 < R8 ? hi * 8 : 0) * 4);
    const unsigned pa1 = prow_a + (unsigned)((16 + hi * 8 < R8 ? 16 + hi * 8 : 0) * 4);
    const unsigned pb = prow_a + (unsigned)(R8 * 4);
    const unsigned ua = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)(slot + hi * 4 * kTP * 2 + r32 * 2);
    // bf16 inputs: ds_read_u16_d16_hi drops the 16 bits into the UPPER half of its destination and leaves the lower half
    // alone -- registers zeroed once before the loop then hold the element as an fp32 value, no widening shift per element
    // (16 VALU instructions per tile of a kernel that is VALU-issue bound)
    constexpr bool kHiLoad = std::is_same<T, __hip_bfloat16>::value;
    unsigned ubits[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ubits[r] = 0u;
    auto from_lds = [&](TileOps<T, NK> &o, float (&uf)[16]) {
        v4f v[4];
        v2f bc;
        unsigned ulocal[16];
        unsigned (&u)[16] = *(kHiLoad ? &ubits : &ulocal);    // no copy between a read and the wait that covers it
        TRAMBA_LDS_RD_("ds_read_b128", v[0], pa0, 0);
        TRAMBA_LDS_RD_("ds_read_b128", v[1], pa0, 16);
        if constexpr (NK == 2) {
            TRAMBA_LDS_RD_("ds_read_b128", v[2], pa1, 0);
            TRAMBA_LDS_RD_("ds_read_b128", v[3], pa1, 16);
        } else {
            v[2] = v[3] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        TRAMBA_LDS_RD_("ds_read_b64", bc, pb, 0);
#define TRAMBA_U16_(R, OFF)                                                                         \
    if constexpr (kHiLoad) {                                                                        \
        asm volatile("ds_read_u16_d16_hi %0, %1 offset:" #OFF : "+v"(u[R]) : "v"(ua) : "memory");  \
    } else {                                                                                        \
        TRAMBA_LDS_RD_("ds_read_u16", u[R], ua, OFF);                                               \
    }
        TRAMBA_U16_(0, 0) TRAMBA_U16_(1, 64) TRAMBA_U16_(2, 128) TRAMBA_U16_(3, 192)
        TRAMBA_U16_(4, 512) TRAMBA_U16_(5, 576) TRAMBA_U16_(6, 640) TRAMBA_U16_(7, 704)
        TRAMBA_U16_(8, 1024) TRAMBA_U16_(9, 1088) TRAMBA_U16_(10, 1152) TRAMBA_U16_(11, 1216)
        TRAMBA_U16_(12, 1536) TRAMBA_U16_(13, 1600) TRAMBA_U16_(14, 1664) TRAMBA_U16_(15, 1728)
#undef TRAMBA_U16_
        // every destination named: no consumer may be scheduled above this wait (the reads above count as "done" to hipcc)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(bc), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]),
                       "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]), "+v"(u[8]), "+v"(u[9]), "+v"(u[10]), "+v"(u[11]),
                       "+v"(u[12]), "+v"(u[13]), "+v"(u[14]), "+v"(u[15])
                     :
                     : "memory");
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            o.araw[kk][0] = v[2 * kk].x; o.araw[kk][1] = v[2 * kk].y; o.araw[kk][2] = v[2 * kk].z; o.araw[kk][3] = v[2 * kk].w;
            o.araw[kk][4] = v[2 * kk + 1].x; o.araw[kk][5] = v[2 * kk + 1].y; o.araw[kk][6] = v[2 * kk + 1].z;
            o.araw[kk][7] = v[2 * kk + 1].w;
        }
        o.bv = bc.x;
        o.cv = bc.y;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (kHiLoad) {
                uf[r] = __builtin_bit_cast(float, u[r]);
            } else {
                uf[r] = raw_to_f<T>(u[r]);
            }
        }
    };

    __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (lds_void *)ib, 4, idx_off(0), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dma(1);                                                    // tile 0 -> my slot, index vector of tile 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (no stores outstanding yet: a counted wait would not wait)
    // SAVE: the state entering every tile -> hst (see ss2d_scan_cl_kernel); one more store per tile behind the DMAs, issued
    // for every tile (an out-of-range offset past the sequence end) so that the counted wait stays a constant
    const long nta = (L + kTP - 1) / kTP + kMaxW;
    const __amdgpu_buffer_rsrc_t rh =
        make_rsrc(SAVE ? hst + ((long)b * K + k) * nta * D : nullptr, SAVE ? (unsigned)(nta * D) * 4u : 0u);
    for (int s = 0; s < nsuper; ++s) {
        if (s > 0) {
            if constexpr (SAVE) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        TileOps<T, NK> cur;
        float uf[16];
        from_lds(cur, uf);
        dma(s + 2);                                             // tile s + 1 (clamped past the end: fetched, never used)
        const int l0 = s * span + wv * kTP;
        float Bp[16], Cp[16];
        w.stage_bc(cur, Bp, Cp, false);
        float a[16], bb[16], preA[4], preH[4], runA, runH;
        w.template terms<true, kBiasMfma>(cur, Bp, L - l0, a, bb, uf);
        w.prefix(a, bb, preA, preH, runA, runH);
        // no barrier, no fold: carry by mailbox (CarryLink); the spin waits on lgkmcnt only, the DMAs above stay in flight
        const float hin = link.acquire(s * W + wv);
        link.publish(fmaf(runA, hin, runH), s * W + wv, hi);
        if constexpr (SAVE) {
            const unsigned ho = (hi == 0 && l0 < L) ? (unsigned)((s * W + wv) * D + c) * 4u : kOutOfRange;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hin), rh, ho, 0, 0);
        }
        w.read_stage4(2, Cp);                                  // C of my 16 positions, read where it is used
        const unsigned yv = (unsigned)((l0 + 4 * hi) * D + c) * (unsigned)sizeof(TY);
        if (l0 + kTP <= L) {  // wave-uniform: ragged only at the sequence end
            w.template replay<TY, false, sizeof(TY) == 2>(ry, yv, yrow, uf, a, bb, Cp, preA, preH, hin, true, kTP);
        } else {
            w.template replay<TY, true, sizeof(TY) == 2>(ry, yv, yrow, uf, a, bb, Cp, preA, preH, hin, true, L - l0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // The last step has requested one more tile (clamped past the end, never read): those LDS-DMA writes must have LANDED before
    // this wave ends.  A workgroup's LDS is released when its waves have ended, and a write still in flight then lands in
    // whatever workgroup the CU gives that LDS to next -- under one stream there is none (one workgroup per CU, and the next
    // kernel starts after this one has drained), but a GEMM workgroup of ANOTHER stream takes the CU at once and finds 1 KB
    // pieces of a token tile in its staged operands: r04, the training step with the guide branches on a side stream ended in
    // NaN gradients about one run in four (scripts/dev/debug_overlap2.py), and the inference forward has run this way since r01.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    link.report();
#endif
}
#undef TRAMBA_LDS_RD_

// ---------------------------------------------------------------------------------------------
// Backward of the fused channels-last scan (training).  Same decomposition as ss2d_scan_cl_kernel -- a
// workgroup of W waves owns 32 channels of one direction, tiles of 32 positions, dt_proj recomputed on the
// matrix core from the gathered x_proj rows -- run twice over the sequence:
//   sweep 1 (left to right)  the state entering every tile -> hst (B,K,NT,D), a small scratch
//   sweep 2 (right to left)  per tile: forward replay for h, then the adjoint recurrence
//        gh_l = C_l gy_l + z_{l+1},   z_l = a_l gh_l          (z: the adjoint carried leftwards)
//     with the same run / half-wave / wave fold as the forward, mirrored.
// gy is the gradient of the MERGED output gathered through the table (ys' gradient is never materialised).
// Outputs: gu (B,K,L,D) = dL/d(gathered x), graw (B,K,L,D) = dL/d(dt_raw = x_proj ranks . dt_w) in sequence
// order (the small projections around them are left to batched GEMMs), gB / gC (B,K,L) summed over channels
// (LDS transpose + atomics over the D/32 channel tiles), gpar (B,3,K,D) = per-channel dA, dD, dbias.
template <typename T, int NK, bool SPLIT, typename TG>   // TG: dtype of the merged map's gradient (float or T)
__global__ __launch_bounds__(kMaxW * kWave) void ss2d_scan_bwd_cl_kernel(
    const T *__restrict__ x, const float *__restrict__ xdbl, const int32_t *__restrict__ table,
    const float *__restrict__ dt_w, const float *__restrict__ dt_bias, const float *__restrict__ Aneg,
    const float *__restrict__ Ds, const TG *__restrict__ gym, T *__restrict__ gu, T *__restrict__ graw,
    float *__restrict__ gB, float *__restrict__ gC, float *__restrict__ gpar, float *__restrict__ hst, int L, int D,
    int K, int R, int W, int bcs, int have_states, int flags, MailboxCtl mctl)
{
    // flags & 1: `Aneg` holds A_logs (see ScanWave::init) and gpar's first plane leaves as dL/dA_logs = dL/dA * A
    // flags & 2: gB / gC are (B, K, CT, L) tables of per-channel-tile partial sums, each element WRITTEN by exactly one wave
    //            (no zero fill, no atomics: the caller adds the CT = ceil(D / 32) partials in a fixed order)
    constexpr int kTS = 36;   // LDS row stride (floats) of the position-sum transposes: 16-byte aligned rows
    __shared__ float mbox[kMaxW][3][kTP];   // carry mailboxes (CarryLink): the forward states, then the adjoint; time-out flags
    __shared__ __attribute__((aligned(16))) float stage[kMaxW][3][kTP];
    __shared__ __attribute__((aligned(16))) float tr[kMaxW][kTP][kTS];
    __shared__ float red[kMaxW][3][kTP];

    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, hi = lane >> 5;
    unsigned ct_, k_, b_;
    xcd_work_item(ct_, k_, b_);
    const int k = (int)k_, b = (int)b_, ctile = (int)ct_;
    const int c = ctile * kTP + r32;
    const bool cok = c < D;
    const int cc_ = cok ? c : D - 1;
    const int RG = xdbl_group_stride(R);
    const int PC = K * RG;

    ScanWave<T, NK, SPLIT> w;
    w.init(dt_w, dt_bias, Aneg, Ds, (long)k * D + cc_, R, RG, PC, D, lane, cc_, &stage[wv][0][0], flags & 1);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (long)b * L * D, (unsigned)L * w.xrow);
    const __amdgpu_buffer_rsrc_t rp =
        make_rsrc(xdbl + (long)b * L * PC + (long)k * RG, ((unsigned)(L - 1) * PC + RG) * 4u);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gym + (long)b * L * D, (unsigned)L * (unsigned)D * (unsigned)sizeof(TG));
    const unsigned orow = (unsigned)D * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(gu + ((long)b * K + k) * L * D, (unsigned)L * orow);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(graw + ((long)b * K + k) * L * D, (unsigned)L * orow);
    const int32_t *tk = table + (long)k * L;

    const int span = W * kTP;
    const int nsuper = (L + span - 1) / span;
    float *hs = hst + ((long)b * K + k) * (long)((L + kTP - 1) / kTP + kMaxW) * D;   // (B, K, NTA, D), tile = position / 32

    for (int i = threadIdx.x; i < 3 * kMaxW * kTP; i += blockDim.x) (&mbox[0][0][0])[i] = 0.f;   // tags: nothing published
    __syncthreads();
    CarryLink fwd_link, adj_link;
    fwd_link.init(&mbox[0][0][0], wv, W, r32, mctl);
    adj_link.init(&mbox[0][0][0], wv, W, r32, mctl, true);
    const unsigned adj_base = (unsigned)(nsuper * W);   // the adjoint chain's tags lie above the forward chain's

    auto load_idx = [&](int s) -> int {
        const int l = s * span + wv * kTP + r32;
        return tk[l < L ? l : L - 1];
    };

    // ---- sweep 1: the state entering every tile (skipped when the forward launch has saved them, block-uniform)
    if (!have_states) {
        int idx_cur = load_idx(0);
        for (int s = 0; s < nsuper; ++s) {
            TileOps<T, NK> cur;
            w.fetch(rx, rp, idx_cur, cur);
            idx_cur = load_idx(s + 1 < nsuper ? s + 1 : s);   // next tile's index vector rides behind the gathers
            const int l0 = s * span + wv * kTP;
            float Bp[16], Cp[16], a[16], bb[16], uf[16], preA[4], preH[4], runA, runH;
            w.stage_bc(cur, Bp, Cp, false);
            w.terms(cur, Bp, L - l0, a, bb, uf);
            w.prefix(a, bb, preA, preH, runA, runH);
            const float hin = fwd_link.acquire(s * W + wv);
            fwd_link.publish(fmaf(runA, hin, runH), s * W + wv, hi);
            if (hi == 0 && cok) hs[(long)(s * W + wv) * D + c] = hin;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __threadfence_block();
    __syncthreads();

    // ---- sweep 2: right to left
    float accA = 0.f, accD = 0.f, accb = 0.f;
    float *trw = &tr[wv][0][0];
    int idx_rev = load_idx(nsuper - 1);
    for (int s = nsuper - 1; s >= 0; --s) {
        TileOps<T, NK> cur;
        w.fetch(rx, rp, idx_rev, cur);
        idx_rev = load_idx(s > 0 ? s - 1 : 0);
        const int l0 = s * span + wv * kTP;
        const int nvalid = L - l0;
        // gradient rows of my 16 positions: the x-row byte offsets staged by fetch() scale to fp32 rows
        float gy[16];
        {
            float xo[16];
            w.read_stage4(0, xo);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned off = __builtin_bit_cast(unsigned, xo[r]) / (unsigned)sizeof(T) * (unsigned)sizeof(TG) +
                                     (unsigned)cc_ * (unsigned)sizeof(TG);
                const bool ok = (r & 3) + 8 * (r >> 2) + 4 * hi < nvalid;
                gy[r] = raw_to_f<TG>(buf_load_raw<TG>(rg, ok ? off : kOutOfRange));
            }
        }
        float Bp[16], Cp[16], a[16], bb[16], uf[16], tl[16], preA[4], preH[4], tA, tH;
        w.stage_bc(cur, Bp, Cp, true);
        w.terms(cur, Bp, nvalid, a, bb, uf, tl);
        w.prefix(a, bb, preA, preH, tA, tH);
        // (a tile past the sequence end is all identity elements; its entering state is not among the saved ones)
        const float hin = l0 < L ? hs[(long)(s * W + wv) * D + cc_] : 0.f;
        // forward replay: h after / before every element
        float hh[16], hp[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float hcur = fmaf(preA[g], hin, preH[g]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hp[4 * g + q] = hcur;
                hcur = fmaf(a[4 * g + q], hcur, bb[4 * g + q]);
                hh[4 * g + q] = hcur;
            }
        }
        // adjoint run aggregates (zero entering from the right of the run), right to left inside a run
        float cg[16], sa[4], sz[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) cg[r] = Cp[r] * gy[r];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float pa = 1.f, pz = 0.f;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                pz = a[4 * g + q] * (cg[4 * g + q] + pz);
                pa *= a[4 * g + q];
            }
            sa[g] = pa;
            sz[g] = pz;
        }
        // runs right to left: 2g+1 (upper half-wave) before 2g (lower); (RA, RZ) = everything to the right so far
        float qA[4], qZ[4], RA = 1.f, RZ = 0.f;
#pragma unroll
        for (int g = 3; g >= 0; --g) {
            float la, lz, ua, uz;
            half_swap(sa[g], la, ua);
            half_swap(sz[g], lz, uz);
            const float mA = ua * RA, mZ = fmaf(ua, RZ, uz);     // after the upper run
            qA[g] = hi ? RA : mA;
            qZ[g] = hi ? RZ : mZ;
            RZ = fmaf(la, mZ, lz);
            RA = la * mA;
        }
        // adjoint entering my tile from the right: the chain runs over the tiles in DEScending order (no barrier, CarryLink)
        const int rt = (nsuper - 1 - s) * W + (W - 1 - wv);
        const float zin = adj_link.acquire(rt, adj_base);
        adj_link.publish(fmaf(RA, zin, RZ), rt, hi, adj_base);
        // replay right to left, emit
        const bool full = ctile * kTP + kTP <= D && l0 + kTP <= L;
        const unsigned ov = (unsigned)((l0 + 4 * hi) * D + cc_) * (unsigned)sizeof(T);
        float eb[16], ec[16];
#pragma unroll
        for (int g = 3; g >= 0; --g) {
            float zr = fmaf(qA[g], zin, qZ[g]);
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const int r = 4 * g + q;
                const float gh = cg[r] + zr;
                zr = a[r] * gh;
                const float dtv = tl[r] * 0.693147180559945f;                 // dt
                const float gdt = gh * fmaf(a[r] * w.An, hp[r], Bp[r] * uf[r] * 1.44269504088896f);
                const float gr = gdt * (1.f - __builtin_amdgcn_exp2f(-tl[r]));   // softplus' = 1 - exp(-dt)
                const float guv = fmaf(w.Dk, gy[r], gh * (tl[r] * Bp[r]));       // dt*B = t * (B ln2)
                eb[r] = gh * dtv * uf[r];
                ec[r] = gy[r] * hh[r];
                accA = fmaf(gh * hp[r], a[r] * dtv, accA);
                accD = fmaf(gy[r], uf[r], accD);
                accb += gr;
                const int p0 = q + 8 * g + 4 * hi;
                const unsigned vo = (full || (cok && p0 < nvalid)) ? ov : kOutOfRange;
                buf_store_elem<T>(ru, vo, (unsigned)(q + 8 * g) * orow, guv);
                buf_store_elem<T>(rr, vo, (unsigned)(q + 8 * g) * orow, gr);
            }
        }
        // position sums over this tile's 32 channels: transpose through LDS (16 channels per lane, then pair the
        // half-waves), once for gB and once for gC
        float sbc[2];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pos = (r & 3) + 8 * (r >> 2) + 4 * hi;
                trw[pos * kTS + r32] = cok ? (which == 0 ? eb[r] : ec[r]) : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            float sv = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(trw + r32 * kTS + 16 * hi + 4 * j);
                sv += (v.x + v.y) + (v.z + v.w);
            }
            __builtin_amdgcn_wave_barrier();
            sbc[which] = sv + __shfl_xor(sv, 32, 64);
        }
        if (hi == 0 && l0 + r32 < L) {
            if (flags & 2) {
                const long o = (((long)b * K + k) * gridDim.x + ctile) * L + l0 + r32;
                gB[o] = sbc[0];
                gC[o] = sbc[1];
            } else {
                atomicAdd(gB + (((long)b * K + k) * L + l0 + r32) * bcs, sbc[0]);   // bcs: floats between positions (1, or
                atomicAdd(gC + (((long)b * K + k) * L + l0 + r32) * bcs, sbc[1]);   // the row stride of an x_dbl-gradient table)
            }
        }
    }
    fwd_link.report();
    adj_link.report();
    // ---- per-channel sums: two half-waves, then the W waves
    accA += __shfl_xor(accA, 32, 64);
    accD += __shfl_xor(accD, 32, 64);
    accb += __shfl_xor(accb, 32, 64);
    if (hi == 0) {
        red[wv][0][r32] = accA;
        red[wv][1][r32] = accD;
        red[wv][2][r32] = accb;
    }
    __syncthreads();
    if (wv == 0 && hi == 0 && cok) {
        float sA = 0.f, sD = 0.f, sb2 = 0.f;
        for (int q = 0; q < W; ++q) {
            sA += red[q][0][r32];
            sD += red[q][1][r32];
            sb2 += red[q][2][r32];
        }
        // (B, 3, K, D): the three parameter gradients are contiguous (K, D) planes once the batch is summed
        float *gp = gpar + ((long)b * 3 * K + k) * D + c;
        gp[0] = (flags & 1) ? sA * w.An : sA;
        gp[(long)K * D] = sD;
        gp[2l * K * D] = sb2;
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-segment form of the same scan (used when a workspace is supplied): NO barriers and no
// per-sequence chain.  A sequence of one (b, k, 32-channel tile) is cut into NSEG segments of NT
// tiles; every WAVE owns one segment and walks its tiles alone.
//   PASS 0  reduces the segment to its (decay, state) pair            -> agg   (B,K,NSEG,D) float2
//   PASS 1  folds the preceding segments' pairs into its carry-in, recomputes the segment and streams y out
// The per-element terms are evaluated twice (~1.6x the arithmetic of the chained kernel), but the
// launch has B*K*(D/32)*NSEG independent waves instead of B*K*(D/32) serial chains of 8 waves, so the
// whole chip is busy: 2-3x faster on this model's shapes.  Segments of one sequence are independent,
// tiles past a segment's end are computed on the identity (dt = 0) and never stored.
struct SegPlan {
    int ntiles, nt, nseg;
};

__host__ inline SegPlan seg_plan(int l, long rowtiles)
{
    SegPlan p;
    p.ntiles = (l + kTP - 1) / kTP;
    long want = 4096 / (rowtiles > 0 ? rowtiles : 1);
    if (want < 1) want = 1;
    int nt = (int)((p.ntiles + want - 1) / want);
    nt = (nt + 2) / 3 * 3;  // multiple of the operand ring (no wasted padding tiles)
    if (p.ntiles <= 6) nt = (p.ntiles + 2) / 3 * 3;  // short sequences: one segment, single pass
    p.nt = nt;
    p.nseg = (p.ntiles + nt - 1) / nt;
    return p;
}

template <typename T, typename TY, int NK, bool SPLIT, int PASS>
__global__ __launch_bounds__(256) void ss2d_seg_kernel(
    const T *__restrict__ x, const float *__restrict__ xdbl, const int32_t *__restrict__ table,
    const float *__restrict__ dt_w, const float *__restrict__ dt_bias, const float *__restrict__ Aneg,
    const float *__restrict__ Ds, TY *__restrict__ ys, float2 *__restrict__ agg, const float *__restrict__ carry,
    int L, int D, int K, int R, int CT, int NT, int NSEG, long nwaves)
{
    __shared__ __attribute__((aligned(16))) float stage[4][3][kTP];

    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long gid = (long)blockIdx.x * 4 + wv;
    if (gid >= nwaves) return;  // wave-uniform, the kernel has no barriers
    // (32-bit: nwaves < 2^31, host-checked -- emulated 64-bit divisions cost hundreds of instructions per wave)
    const unsigned g32 = (unsigned)gid;
    // (segment fastest.  Channel-tile fastest -- the 4 waves of a block reading neighbouring 64-byte pieces of the same
    // rows -- was measured 4.6 % SLOWER on the replay pass: the waves of a block then all fold the same number of
    // preceding segments, and the long folds bunch up on some CUs.)
    const int sgm = (int)(g32 % (unsigned)NSEG);
    unsigned rt = g32 / (unsigned)NSEG;
    const int ctile = (int)(rt % (unsigned)CT);
    rt /= (unsigned)CT;
    const int k = (int)(rt % (unsigned)K), b = (int)(rt / (unsigned)K);

    const int r32 = lane & 31, hi = lane >> 5;
    const int c = ctile * kTP + r32;
    const bool cok = c < D;
    const int cc_ = cok ? c : D - 1;
    const int RG = xdbl_group_stride(R);
    const int PC = K * RG;

    ScanWave<T, NK, SPLIT> w;
    w.init(dt_w, dt_bias, Aneg, Ds, (long)k * D + cc_, R, RG, PC, D, lane, cc_, &stage[wv][0][0]);

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (long)b * L * D, (unsigned)L * w.xrow);
    const __amdgpu_buffer_rsrc_t rp =
        make_rsrc(xdbl + (long)b * L * PC + (long)k * RG, ((unsigned)(L - 1) * PC + RG) * 4u);
    const unsigned yrow = (unsigned)D * (unsigned)sizeof(TY);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(ys + ((long)b * K + k) * L * D, (unsigned)L * yrow);
    const int32_t *tk = table + (long)k * L;

    const int t0 = sgm * NT;                                  // first tile of my segment
    const int lend = (t0 + NT) * kTP < L ? (t0 + NT) * kTP : L;  // my segment is [t0*32, lend)
    const int tlast = (L - 1) / kTP;                          // loads are clamped to the last real tile

    auto load_idx = [&](int t) -> int {
        const int l = (t < tlast ? t : tlast) * kTP + r32;
        return tk[l < L ? l : L - 1];
    };
    constexpr int AHEAD = NK <= 2 ? 2 : 1;
    constexpr int NS = AHEAD + 1;
    TileOps<T, NK> ops[NS];
    int idxr[NS], idx0[AHEAD];   // index vectors, NS tiles ahead of their gathers (see ss2d_scan_cl_kernel)
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) idx0[t] = load_idx(t0 + t);
#pragma unroll
    for (int t = 0; t < NS; ++t) idxr[(AHEAD + t) % NS] = max(load_idx(t0 + AHEAD + t), 0);
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) w.fetch(rx, rp, idx0[t], ops[t]);

    float runA = 1.f, runH = 0.f;  // PASS 0: aggregate of the segment so far
    float hin = 0.f;               // PASS 1: state entering the current tile
    if (PASS == 1 && NSEG > 1) {
        // carry-in = fold of the preceding segments' (decay, state) pairs: <= NSEG independent 8-byte loads per lane
        // and a short fma chain, instead of a separate latency-bound carry-scan launch between the two passes
        const float2 *ag = agg + ((long)b * K + k) * NSEG * D + cc_;
        int s2 = 0;
        for (; s2 + 8 <= sgm; s2 += 8) {
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ag[(long)(s2 + j) * D];
#pragma unroll
            for (int j = 0; j < 8; ++j) hin = fmaf(v[j].x, hin, v[j].y);
        }
        for (; s2 < sgm; ++s2) {
            const float2 v = ag[(long)s2 * D];
            hin = fmaf(v.x, hin, v.y);
        }
    }
    const bool cfull = ctile * kTP + kTP <= D;

    const int ntp = (NT + NS - 1) / NS * NS;  // whole ring trips; padding tiles run on the identity
    for (int i0 = 0; i0 < ntp; i0 += NS) {
#pragma unroll
      for (int sti = 0; sti < NS; ++sti) {
        const int t = t0 + i0 + sti;
        TileOps<T, NK> &cur = ops[sti];
        const int l0 = t * kTP;
        float Bp[16], Cp[16];
        w.stage_bc(cur, Bp, Cp, PASS == 1);
        w.fetch(rx, rp, idxr[(sti + AHEAD) % NS], ops[(sti + AHEAD) % NS]);
        idxr[(sti + AHEAD) % NS] = load_idx(t + AHEAD + NS);

        float a[16], bb[16], uf[16], preA[4], preH[4], tA, tH;
        w.terms(cur, Bp, lend - l0, a, bb, uf);   // positions at or past the segment end are the identity
        w.prefix(a, bb, preA, preH, tA, tH);
        if (PASS == 0) {
            runH = fmaf(tA, runH, tH);
            runA *= tA;
        } else {
            const unsigned yv = (unsigned)((l0 + 4 * hi) * D + cc_) * (unsigned)sizeof(TY);
            if (cfull && l0 + kTP <= lend) {  // wave-uniform
                w.template replay<TY, false>(ry, yv, yrow, uf, a, bb, Cp, preA, preH, hin, true, kTP);
            } else {
                w.template replay<TY, true>(ry, yv, yrow, uf, a, bb, Cp, preA, preH, hin, cok, lend - l0);
            }
            hin = fmaf(tA, hin, tH);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (PASS == 0 && hi == 0 && cok) agg[(((long)b * K + k) * NSEG + sgm) * D + c] = make_float2(runA, runH);
}

// ---------------------------------------------------------------------------------------------
// ss2d_merge_norm_cl, streaming form (bijective orders on large maps):
// y[b, p, :] = act(LayerNorm(sum over the CSR entries e of pixel p of ys[b, entry e, :])).
// One wave owns P consecutive pixels of one image.  Their CSR window (P+1 pointers, then up to 64 entries)
// is fetched ONCE, so a pixel costs one memory latency -- the row gathers -- instead of three dependent
// ones (pointer -> entry -> rows); the rows of the next batch are requested before the current batch is
// reduced (2 static register slots, loop unrolled by two, exit test at the bottom only).  The loop body is
// branch-free: rows past a batch's count are clamped re-reads weighted 0, LayerNorm runs for every batch
// and only a pixel's last batch stores (the others get the out-of-range vector offset).
// NIT = wave iterations per row (D <= 64*V*NIT); BATCH = rows requested per step.
template <typename TY> struct YsRaw;
template <> struct YsRaw<float> { static constexpr int kBytes = 4; };
template <> struct YsRaw<__half> { static constexpr int kBytes = 2; };
template <> struct YsRaw<__hip_bfloat16> { static constexpr int kBytes = 2; };

template <typename TY, int V>
__device__ __forceinline__ void buf_load_row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float (&out)[V])
{
    if constexpr (sizeof(TY) == 4) {
        static_assert(V == 4 || V == 2 || V == 1, "vector width");
        if constexpr (V == 4) {
            const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else if constexpr (V == 2) {
            const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
            out[0] = v.x; out[1] = v.y;
        } else {
            out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
        }
    } else {
        Pack<TY, V> pk;
        if constexpr (V == 4) {
            pk = __builtin_bit_cast(Pack<TY, V>, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        } else if constexpr (V == 2) {
            pk = __builtin_bit_cast(Pack<TY, V>, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
        } else {
            pk = __builtin_bit_cast(Pack<TY, V>, __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0));
        }
#pragma unroll
        for (int v = 0; v < V; ++v) out[v] = Cvt<TY>::to_f(pk.v[v]);
    }
}

template <typename TY, typename T, int V, int NIT, int BATCH>
__global__ __launch_bounds__(256) void ss2d_merge_norm_stream_kernel(
    const TY *__restrict__ ys, const int32_t *__restrict__ inv_ptr, const int32_t *__restrict__ inv_idx,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, T *__restrict__ y, long nwaves, int nchunk,
    int P, int L, int D, int K, float eps, int act, int B, int H)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wid >= nwaves) return;   // wave-uniform, no barriers below
    int b, pl0;   // 32-bit: nwaves < 2^31 (host-checked)
    if (H > 0) {  // many-to-one table on a square map whose rows are whole chunks: image rows centre-out (long pixels first)
        const unsigned cpr = (unsigned)(L / H) / (unsigned)P, per = (unsigned)B * cpr;
        const unsigned rank = (unsigned)wid / per, rem = (unsigned)wid % per;
        b = (int)(rem / cpr);
        const int row = H / 2 + ((rank & 1u) ? -(int)((rank + 1) >> 1) : (int)(rank >> 1));
        pl0 = row * (L / H) + (int)(rem % cpr) * P;
    } else {
        b = (int)((unsigned)wid / (unsigned)nchunk);
        pl0 = (int)((unsigned)wid % (unsigned)nchunk) * P;
    }
    const int np = L - pl0 < P ? L - pl0 : P;   // pixels of this wave

    // CSR window: pointers of pixels pl0 .. pl0+np in lanes 0..np, then 64 entries from the first one
    const int ptr = inv_ptr[pl0 + (lane < np ? lane : np)];
    const int e_lo = __builtin_amdgcn_readlane(ptr, 0), e_hi = __builtin_amdgcn_readlane(ptr, np);
    int ebase = e_lo;
    int ent = inv_idx[ebase + lane < e_hi ? ebase + lane : e_hi - 1];

    const unsigned yrow = (unsigned)D * (unsigned)sizeof(T), srow = (unsigned)D * (unsigned)sizeof(TY);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(ys + (long)b * K * L * D, (unsigned)K * L * srow);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(y + ((long)b * L + pl0) * D, (unsigned)np * yrow);

    // loop-invariant per-lane operands
    float gw[NIT][V], gb[NIT][V];
    unsigned coff[NIT];       // byte offset of my channels in a ys row, out of range past D
    unsigned ooff[NIT];       // same in an output row
    bool cin[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c0 = (it * kWave + lane) * V;
        cin[it] = c0 + V <= D;
        coff[it] = cin[it] ? (unsigned)c0 * (unsigned)sizeof(TY) : kOutOfRange;
        ooff[it] = cin[it] ? (unsigned)c0 * (unsigned)sizeof(T) : kOutOfRange;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            gw[it][v] = cin[it] ? ln_w[c0 + v] : 0.f;
            gb[it][v] = cin[it] ? ln_b[c0 + v] : 0.f;
        }
    }

    struct Batch { int j, e0, cnt; bool last, valid; };
    int sj = 0, se = e_lo;   // stream state: current pixel, next entry
    auto next_batch = [&]() -> Batch {
        Batch q;
        q.valid = sj < np;
        const int jj = q.valid ? sj : np - 1;
        const int eend = __builtin_amdgcn_readlane(ptr, jj + 1);
        q.j = jj;
        q.e0 = q.valid ? se : e_hi - 1;
        const int left = q.valid ? eend - se : 1;
        q.cnt = left < BATCH ? left : BATCH;
        q.last = q.valid && q.cnt == left;
        if (q.valid) {
            se += q.cnt;
            if (q.last) ++sj;
        }
        return q;
    };
    auto issue = [&](const Batch &q, float (&t)[BATCH][NIT][V]) {
        if (q.e0 + q.cnt > ebase + kWave) {   // wave-uniform, rare: slide the entry window
            ebase = q.e0;
            ent = inv_idx[ebase + lane < e_hi ? ebase + lane : e_hi - 1];
        }
#pragma unroll
        for (int r = 0; r < BATCH; ++r) {
            const int ei = q.e0 - ebase + (r < q.cnt ? r : q.cnt - 1);   // clamped: weighted 0 below
            const unsigned so = (unsigned)__builtin_amdgcn_readlane(ent, ei) * srow;
#pragma unroll
            for (int it = 0; it < NIT; ++it) buf_load_row<TY, V>(rs, coff[it], so, t[r][it]);
        }
    };
    float acc[kNormMaxIt][V];
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[it][v] = 0.f;
    auto consume = [&](const Batch &q, const float (&t)[BATCH][NIT][V]) {
#pragma unroll
        for (int r = 0; r < BATCH; ++r) {
            const float wr = (q.valid && r < q.cnt) ? 1.f : 0.f;   // scalar
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[it][v] = fmaf(t[r][it][v], wr, acc[it][v]);
        }
        float mean, rstd;
        wave_layernorm<V>(acc, NIT, D, lane, eps, mean, rstd);
        const unsigned so = (unsigned)q.j * yrow;
        const float keep = q.last ? 0.f : 1.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float o[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                o[v] = apply_act((acc[it][v] - mean) * rstd * gw[it][v] + gb[it][v], act);
                acc[it][v] *= keep;
            }
            // The row offset goes into the VECTOR offset, the scalar offset stays 0: a 16-byte buffer store with
            // an SGPR soffset is exempt from hipcc's "VALU overwrites the data of a >64-bit store" wait state,
            // yet on gfx950 under memory back-pressure the store then picked up the overwritten registers
            // (seen as corrupted dword 0 of lanes 12-15 per 16 when two streams ran; tests/test_gpu_kernels.py::test_two_stream_concurrency_is_bitwise_stable).
            const unsigned vo = (q.last ? ooff[it] : kOutOfRange) + so;
            const Pack<T, V> pk = [&] { Pack<T, V> z; for (int v = 0; v < V; ++v) z.v[v] = Cvt<T>::from_f(o[v]); return z; }();
            if constexpr (sizeof(T) * V == 16) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, pk), ro, vo, 0, 0);
            } else if constexpr (sizeof(T) * V == 8) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, pk), ro, vo, 0, 0);
            } else if constexpr (sizeof(T) * V == 4) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), ro, vo, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, pk), ro, vo, 0, 0);
            }
        }
    };

    float t0[BATCH][NIT][V], t1[BATCH][NIT][V];
    Batch c0 = next_batch();
    issue(c0, t0);
    while (true) {
        const Batch c1 = next_batch();
        issue(c1, t1);
        consume(c0, t0);
        c0 = next_batch();
        issue(c0, t0);
        consume(c1, t1);
        if (!c0.valid) break;
    }
}

// ss2d_merge_norm_cl, one-wave-per-pixel form (many-to-one Helix tables, small maps, merge-only calls).
// A Helix pixel lists 6 .. 4 + 2H entries (196 at 96x96: the Bresenham lines all cross the image centre), so the
// kernel is built around the long pixels instead of suffering them:
//  * the entry list of a pixel is fetched 64 entries per load (one per lane), not 4;
//  * a short pixel (<= RB rows) requests all its rows at once -- one memory latency; a long one walks its rows RB at a
//    time on two static register buffers, branch-free (rows past the end are clamped re-reads weighted 0), so 2*RB row
//    requests stay in flight per wave;
//  * on a square map the waves take the image rows centre-out (all images' centre rows first): the few hundred long
//    pixels start at t = 0 and finish under the cover of the short ones instead of forming the kernel's tail
//    (r01: 97 us for 302 MB, of which ~70 us was one wave summing 196 rows four at a time).
// Rows are kept as raw bits until they are summed (a conversion at the load site makes hipcc wait for the load there);
// the summation order is the CSR order, fixed.
template <typename TY, int V>
__device__ __forceinline__ Pack<TY, V> buf_load_pack(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    constexpr int kB = (int)sizeof(TY) * V;
    static_assert(kB == 16 || kB == 8 || kB == 4 || kB == 2, "row piece of 2..16 bytes");
    if constexpr (kB == 16) return __builtin_bit_cast(Pack<TY, V>, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    else if constexpr (kB == 8) return __builtin_bit_cast(Pack<TY, V>, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    else if constexpr (kB == 4) return __builtin_bit_cast(Pack<TY, V>, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    else return __builtin_bit_cast(Pack<TY, V>, __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0));
}

template <typename TY, typename T, int V, int NIT>
__global__ __launch_bounds__(256) void ss2d_merge_norm_deep_kernel(
    const TY *__restrict__ ys, const int32_t *__restrict__ inv_ptr, const int32_t *__restrict__ inv_idx,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, T *__restrict__ y, long npix, int B, int L,
    int D, int K, int H, float eps, int act, const T *__restrict__ addend = nullptr, const T *__restrict__ zpre = nullptr)
{
    // addend / zpre (training, merge-only calls: the gradient of SS2D's input): y = (sum + addend) * silu'(zpre) -- the x_proj
    // branch's share of the gradient added and the SiLU in front of the core (vmamba.py:283-285) differentiated in the pass
    // that merges the scan's input gradient, instead of an add and a silu_backward launch over the map.
    // rows per batch: 32 bytes per lane and buffer (2 buffers in flight = 4 KB per wave; 58 VGPRs = 8 waves per SIMD)
    constexpr int RB = 32 / (NIT * V * (int)sizeof(TY)) > 0 ? 32 / (NIT * V * (int)sizeof(TY)) : 1;
    const int lane = threadIdx.x & (kWave - 1);
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wid >= npix) return;   // wave-uniform, no barriers below
    int b, p;
    if (H > 0) {   // square map: image rows centre-out, every image's row of one rank side by side (npix < 2^31)
        const unsigned Wd = (unsigned)L / (unsigned)H, per = (unsigned)B * Wd;
        const unsigned rank = (unsigned)wid / per, rem = (unsigned)wid % per;
        b = (int)(rem / Wd);
        const int row = H / 2 + ((rank & 1u) ? -(int)((rank + 1) >> 1) : (int)(rank >> 1));
        p = row * (int)Wd + (int)(rem % Wd);
    } else {
        b = (int)((unsigned)wid / (unsigned)L);
        p = (int)((unsigned)wid % (unsigned)L);
    }
    const int e0 = __builtin_amdgcn_readfirstlane(inv_ptr[p]), e1 = __builtin_amdgcn_readfirstlane(inv_ptr[p + 1]);
    const int n = e1 - e0;

    const unsigned srow = (unsigned)D * (unsigned)sizeof(TY);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(ys + (long)b * K * L * D, (unsigned)K * (unsigned)L * srow);
    unsigned coff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c0 = (it * kWave + lane) * V;
        coff[it] = c0 + V <= D ? (unsigned)c0 * (unsigned)sizeof(TY) : kOutOfRange;   // past D: reads as zero
    }
    float acc[kNormMaxIt][V];
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[it][v] = 0.f;

    typedef Pack<TY, V> Raw;
    for (int base = 0; base < n; base += kWave) {   // one trip except for the few pixels with > 64 entries
        const int cn = n - base < kWave ? n - base : kWave;
        const int ent = inv_idx[e0 + base + (lane < cn ? lane : cn - 1)];   // entry = k*L + l: row of ys
        auto row_off = [&](int j) -> unsigned {   // j wave-uniform, < cn
            return (unsigned)__builtin_amdgcn_readlane(ent, j) * srow;
        };
        Raw t0[RB][NIT], t1[RB][NIT];
        if (cn <= RB) {
            // short pixel: every row requested before any is summed (wave-uniform guards, no wasted requests)
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (j < cn) {
                    const unsigned so = row_off(j);
#pragma unroll
                    for (int it = 0; it < NIT; ++it) t0[j][it] = buf_load_pack<TY, V>(rs, coff[it], so);
                }
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (j < cn)
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[it][v] += Cvt<TY>::to_f(t0[j][it].v[v]);
        } else {
            // long pixel: batches of RB rows on two static buffers, branch-free
            const int nb = (cn + RB - 1) / RB;
            auto issue = [&](int q, Raw (&t)[RB][NIT]) {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const int r = q * RB + j;
                    const unsigned so = row_off(r < cn ? r : cn - 1);
#pragma unroll
                    for (int it = 0; it < NIT; ++it) t[j][it] = buf_load_pack<TY, V>(rs, coff[it], so);
                }
            };
            auto consume = [&](int q, const Raw (&t)[RB][NIT]) {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const float wr = q * RB + j < cn ? 1.f : 0.f;   // scalar
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[it][v] = fmaf(Cvt<TY>::to_f(t[j][it].v[v]), wr, acc[it][v]);
                }
            };
            issue(0, t0);
            for (int q = 0; q < nb; q += 2) {
                issue(q + 1, t1);
                consume(q, t0);
                issue(q + 2, t0);
                consume(q + 1, t1);
            }
        }
    }
    T *orow = y + ((long)b * L + p) * D;
    if (eps < 0.f) {   // merge only (CrossMerge without out_norm: the training path and gradient merges)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c0 = (it * kWave + lane) * V;
            if (c0 + V <= D) {
                if (zpre) {   // (wave-uniform)
                    float av[V], zv[V];
                    load_pack<T, V>(zpre + ((long)b * L + p) * D + c0, zv);
                    if (addend) {
                        load_pack<T, V>(addend + ((long)b * L + p) * D + c0, av);
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[it][v] += av[v];
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[it][v] *= silu_gradf_(zv[v]);
                }
                store_pack<T, V>(orow + c0, acc[it]);
            }
        }
        return;
    }
    float mean, rstd;
    wave_layernorm<V>(acc, NIT, D, lane, eps, mean, rstd);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c0 = (it * kWave + lane) * V;
        if (c0 + V <= D) {
            float o[V];
#pragma unroll
            for (int v = 0; v < V; ++v)
                o[v] = apply_act((acc[it][v] - mean) * rstd * ln_w[c0 + v] + ln_b[c0 + v], act);
            store_pack<T, V>(orow + c0, o);
        }
    }
}

// ss2d_merge_norm_cl, split-row form (small maps with wide rows: 24x24 D=1024, 12x12 D=2048): the WPP waves of
// a workgroup share ONE pixel, each owning every WPP-th 64*V-channel slice of the row, so a 2304-pixel map
// puts 9216 waves in flight instead of 2304; the LayerNorm statistics cross the waves through LDS.
template <typename TY, typename T, int V, int NIT, int WPP>
__global__ __launch_bounds__(256) void ss2d_merge_norm_split_kernel(
    const TY *__restrict__ ys, const int32_t *__restrict__ inv_ptr, const int32_t *__restrict__ inv_idx,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, T *__restrict__ y, long npix, int L,
    int D, int K, float eps, int act)
{
    static_assert(NIT % WPP == 0 && 4 % WPP == 0, "waves per pixel must divide the iterations and the block");
    constexpr int NITW = NIT / WPP;            // slices per wave
    constexpr int PPB = 4 / WPP;               // pixels per workgroup
    constexpr int BATCH = NITW == 1 ? 4 : 2;
    __shared__ float red[2][4];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = wv % WPP, pb = wv / WPP;
    long pix = (long)blockIdx.x * PPB + pb;
    const bool live = pix < npix;              // wave-uniform; dead waves still join the barriers
    if (!live) pix = npix - 1;
    const int b = (int)((unsigned)pix / (unsigned)L), p = (int)((unsigned)pix % (unsigned)L);   // npix < 2^31
    float acc[NITW][V];
#pragma unroll
    for (int it = 0; it < NITW; ++it)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[it][v] = 0.f;

    const TY *yb = ys + (long)b * K * L * D;
    const int e0 = inv_ptr[p], e1 = inv_ptr[p + 1];
    for (int e = e0; e < e1; e += BATCH) {
        const int mine = (lane < BATCH && e + lane < e1) ? inv_idx[e + lane] : 0;
        float t[BATCH][NITW][V];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (e + j < e1) {  // wave-uniform
                const TY *row = yb + (long)__builtin_amdgcn_readlane(mine, j) * D;  // entry = k*L + l
#pragma unroll
                for (int it = 0; it < NITW; ++it) {
                    const int c0 = ((it * WPP + sub) * kWave + lane) * V;
                    if (c0 + V <= D) {
                        load_pack<TY, V>(row + c0, t[j][it]);
                    } else {
#pragma unroll
                        for (int v = 0; v < V; ++v) t[j][it][v] = 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
            if (e + j < e1)
#pragma unroll
                for (int it = 0; it < NITW; ++it)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[it][v] += t[j][it][v];
    }
    // two-pass statistics over the WPP slices of the row (channels past D hold 0 and are masked out)
    float sl = 0.f;
#pragma unroll
    for (int it = 0; it < NITW; ++it)
#pragma unroll
        for (int v = 0; v < V; ++v) sl += acc[it][v];
    sl = wave_sum(sl);
    if (lane == 0) red[0][wv] = sl;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < WPP; ++q) tot += red[0][pb * WPP + q];
    const float mean = tot / (float)D;
    float ql = 0.f;
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
        const int c0 = ((it * WPP + sub) * kWave + lane) * V;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float dlt = acc[it][v] - mean;
            if (c0 + v < D) ql = fmaf(dlt, dlt, ql);
        }
    }
    ql = wave_sum(ql);
    if (lane == 0) red[1][wv] = ql;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int q = 0; q < WPP; ++q) qt += red[1][pb * WPP + q];
    const float rstd = rsqrtf(qt / (float)D + eps);
    if (!live) return;
    T *orow = y + pix * D;
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
        const int c0 = ((it * WPP + sub) * kWave + lane) * V;
        if (c0 + V <= D) {
            float o[V];
#pragma unroll
            for (int v = 0; v < V; ++v)
                o[v] = apply_act((acc[it][v] - mean) * rstd * ln_w[c0 + v] + ln_b[c0 + v], act);
            store_pack<T, V>(orow + c0, o);
        }
    }
}

void set_mailbox_err_word(unsigned *dev_ptr)
{
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mailbox_err_word), &dev_ptr, sizeof(dev_ptr));
}

}  // namespace tramba

using namespace tramba;

extern "C" size_t tramba_ss2d_scan_workspace(int batch, int l, int d, int k)
{
    if (batch <= 0 || l <= 0 || d <= 0 || k <= 0) return 0;
    const SegPlan p = seg_plan(l, (long)batch * k * ((d + kTP - 1) / kTP));
    if (p.nseg <= 1) return 16;  // single pass, nothing to exchange (non-zero: "use the segment kernel")
    return (size_t)batch * k * p.nseg * d * (sizeof(float2) + sizeof(float));
}

// the chained forms, with or without the per-tile states of the training path (saved only where ys is in the input dtype:
// the discarded branch keeps the other combinations from being instantiated)
template <typename T, typename TY, int NK, int R8, int W>
static void launch_scan_dma(dim3 grid, dim3 block, hipStream_t s, const void *x, const float *xdbl, const int32_t *table,
                            const float *dt_w, const float *dt_bias, const float *A, const float *Ds, void *ys, int l, int d,
                            int k, int r, float *states, int a_log)
{
    if constexpr (std::is_same<T, TY>::value) {
        if (states) {
            hipLaunchKernelGGL((ss2d_scan_dma_kernel<T, TY, NK, R8, W, true>), grid, block, 0, s, (const T *)x, xdbl, table,
                               dt_w, dt_bias, A, Ds, (TY *)ys, l, d, k, r, states, a_log, mailbox_ctl(s));
            return;
        }
    }
    hipLaunchKernelGGL((ss2d_scan_dma_kernel<T, TY, NK, R8, W>), grid, block, 0, s, (const T *)x, xdbl, table, dt_w, dt_bias, A,
                       Ds, (TY *)ys, l, d, k, r, (float *)nullptr, a_log, mailbox_ctl(s));
}

template <typename T, typename TY, int NK, bool SP>
static void launch_scan_ring(dim3 grid, dim3 block, hipStream_t s, const void *x, const float *xdbl, const int32_t *table,
                             const float *dt_w, const float *dt_bias, const float *A, const float *Ds, void *ys, int l, int d,
                             int k, int r, int W, float *states, int a_log)
{
    if constexpr (std::is_same<T, TY>::value) {
        if (states) {
            hipLaunchKernelGGL((ss2d_scan_cl_kernel<T, TY, NK, SP, true>), grid, block, 0, s, (const T *)x, xdbl, table, dt_w,
                               dt_bias, A, Ds, (TY *)ys, l, d, k, r, W, states, a_log, mailbox_ctl(s));
            return;
        }
    }
    hipLaunchKernelGGL((ss2d_scan_cl_kernel<T, TY, NK, SP>), grid, block, 0, s, (const T *)x, xdbl, table, dt_w, dt_bias, A, Ds,
                       (TY *)ys, l, d, k, r, W, (float *)nullptr, a_log, mailbox_ctl(s));
}

extern "C" int tramba_ss2d_scan_cl(const void *x, const float *xdbl, const int32_t *table,
                                   const float *dt_w, const float *dt_bias, const float *A,
                                   const float *Ds, void *ys, void *workspace, size_t workspace_bytes,
                                   int batch, int l, int d, int k, int r, int dtype, int ys_dtype,
                                   float *states, int a_log, void *stream)
{
    TRAMBA_CHECK(x && xdbl && table && dt_w && dt_bias && A && Ds && ys, "ss2d_scan_cl: null tensor");
    TRAMBA_CHECK(!states || ys_dtype == dtype, "ss2d_scan_cl: states are saved by the forms whose ys is in the input dtype");
    TRAMBA_CHECK(batch > 0 && l > 0 && d > 0 && k > 0 && r > 0, "ss2d_scan_cl: empty shape");
    TRAMBA_CHECK(batch <= 65535 && k <= 65535, "ss2d_scan_cl: B or K exceeds grid limits");
    TRAMBA_CHECK(r <= 64, "ss2d_scan_cl: dt_rank %d > 64 unsupported", r);
    TRAMBA_CHECK(ys_dtype == TRAMBA_F32 || ys_dtype == dtype, "ss2d_scan_cl: ys must be f32 or the input dtype");
    TRAMBA_CHECK(aligned16(xdbl), "ss2d_scan_cl: xdbl must be 16-byte aligned");
    TRAMBA_CHECK((double)batch * k * l * (double)d < 2.0e9, "ss2d_scan_cl: tensor too large for this build");
    // 32-bit byte offsets inside one image / one (b, k) sequence (buffer descriptors), incl. the padding tiles
    TRAMBA_CHECK(((double)l + 1024.0) * d * 4.0 < 2147483648.0 && (double)l * k * rg_bytes(r) < 2147483648.0,
                 "ss2d_scan_cl: L*D too large for 32-bit offsets");
    hipStream_t s = (hipStream_t)stream;
    const int nk = (r + 15) / 16;
    const int rg = xdbl_group_stride(r);
    // algorithmic bytes of this op: x read once, the low-rank x_proj rows, ys written
    ProfScope prof(TRAMBA_PROF_SCAN_FUSED, s,
                   (double)batch * l * d * dtype_size(dtype) + (double)batch * l * k * rg * 4.0 +
                       (double)batch * k * l * (double)d * dtype_size(ys_dtype));
#define BY_NK_(MAC, T, TY, SP_)             \
    switch (nk) {                           \
    case 1: MAC(T, TY, 1, SP_); break;      \
    case 2: MAC(T, TY, 2, SP_); break;      \
    case 3: MAC(T, TY, 3, SP_); break;      \
    default: MAC(T, TY, 4, SP_); break;     \
    }
#define BY_DTYPE_(MAC)                                                                                          \
    if (dtype == TRAMBA_F32) {                                                                                  \
        BY_NK_(MAC, float, float, true)                                                                         \
    } else if (dtype == TRAMBA_BF16) {                                                                          \
        if (ys_dtype == TRAMBA_F32) { BY_NK_(MAC, __hip_bfloat16, float, false) }                               \
        else { BY_NK_(MAC, __hip_bfloat16, __hip_bfloat16, false) }                                             \
    } else if (dtype == TRAMBA_F16) {                                                                           \
        if (ys_dtype == TRAMBA_F32) { BY_NK_(MAC, __half, float, false) } else { BY_NK_(MAC, __half, __half, false) } \
    } else {                                                                                                    \
        set_error("ss2d_scan_cl: bad dtype %d", dtype);                                                         \
        return TRAMBA_ERR_ARG;                                                                                  \
    }

    const int ct = (d + kTP - 1) / kTP;
    const SegPlan p = seg_plan(l, (long)batch * k * ct);
    // Form selection (measured, scripts/bench_scan.py): the kernels are VALU-throughput bound once the chip
    // is full, so the segment form (1.6x the arithmetic) only wins where the chained form leaves CUs idle:
    // a single-segment sequence (no recompute at all, no barriers), or <= 128 sequences of >= 64 tiles.
    // (A caller that passes no workspace gets the chained form.)
    // r03, after the chained kernels lost their barrier (scripts/sweep_scan_forms.py, batch 4 and batch 1): a single-segment
    // sequence is faster on the chained kernel with its gather pipeline (12x12 maps: 15.3 against 18.1 us at batch 4, 9.8
    // against 16.3 at batch 1); the segment form keeps the long maps of a small batch (<= 128 sequences of >= 128 tiles).
    const bool seg_wins = (long)batch * k * ct <= 128 && p.ntiles >= 128;   // (48x48, 72 tiles: the ring at W = 8 is ahead)
    const int form_tune = tramba_tune_get(TRAMBA_TUNE_SCAN_FORM);      // 1 = chained, 2 = wave-segment, 3 = chained on LDS-DMA
    const bool use_seg = !states && !a_log && workspace != nullptr && workspace_bytes >= tramba_ss2d_scan_workspace(batch, l, d, k) &&
                         (form_tune == 2 || (form_tune == 0 && seg_wins));   // (the LDS-DMA form below is tried first;
                                                                             //  the chained forms are the ones that save states)
    // ---- chained form on LDS-DMA staged operands (4 waves per SIMD): 16-bit maps with dt_rank <= 32 whose sequences fill
    //      the chip at 16 or at 8 waves each
    {
        const int scan_tune = tramba_tune_get(TRAMBA_TUNE_SCAN_FORM);
        const long seqs = (long)batch * k * ct;
        const int r8 = xdbl_rank_pad(r);
        const int wdma = seqs * 16 <= 4096 ? 16 : 8;
        const bool dma_ok = dtype != TRAMBA_F32 && (r8 == 8 || r8 == 16 || r8 == 32) && d % kTP == 0 && l >= 2 * wdma * kTP &&
                            seqs * wdma <= 4096;
        // (measured, scripts/bench_scan.py / bench_scan_w.py, B=4: Helix 96x96 K=8, 256 sequences: 77-82 us against 96 on the
        //  register ring; raster 96x96 K=4, 128 sequences -- half the CUs -- 72 against 85-89 for the wave-segment form; Helix
        //  48x48, 512 sequences x 8 waves: 42 against 52; no gain below ~4 super-chunks per sequence -- 24x24: 25 = 25 --
        //  where the two memory round trips of the prologue weigh as much as the tiles)
        // (r03 sweep: with the mailbox carry the register ring at 8 waves per sequence is ahead on the 48x48 K = 4 maps --
        //  25.4 against 28.1 us -- so the LDS-DMA form wants >= 8 steps per sequence, or 8-wave workgroups with >= 8 tiles each)
        if (dma_ok && (scan_tune == 3 || (scan_tune == 0 && seqs * wdma >= 2048 &&
                                          p.ntiles >= 8 * wdma))) {
            dim3 grid(ct, k, batch), block(wdma * kWave);
#define DMA_(T, TY, NK_, R8_, W_) \
    launch_scan_dma<T, TY, NK_, R8_, W_>(grid, block, s, x, xdbl, table, dt_w, dt_bias, A, Ds, ys, l, d, k, r, states, a_log)
#define DMA_W_(T, TY, NK_, R8_) \
    if (wdma == 16) { DMA_(T, TY, NK_, R8_, 16); } else { DMA_(T, TY, NK_, R8_, 8); }
#define DMA_R_(T, TY) \
    if (r8 == 8) { DMA_W_(T, TY, 1, 8) } else if (r8 == 16) { DMA_W_(T, TY, 1, 16) } else { DMA_W_(T, TY, 2, 32) }
            if (dtype == TRAMBA_BF16) {
                if (ys_dtype == TRAMBA_F32) { DMA_R_(__hip_bfloat16, float) } else { DMA_R_(__hip_bfloat16, __hip_bfloat16) }
            } else {
                if (ys_dtype == TRAMBA_F32) { DMA_R_(__half, float) } else { DMA_R_(__half, __half) }
            }
#undef DMA_R_
#undef DMA_W_
#undef DMA_
            TRAMBA_LAUNCH_CHECK();
            return TRAMBA_OK;
        }
    }
    if (use_seg) {
        // ---- wave-segment form: pass 0 -> pass 1 (single pass when one segment suffices)
        TRAMBA_CHECK(aligned16(workspace), "ss2d_scan_cl: workspace must be 16-byte aligned");
        const long nwaves = (long)batch * k * ct * p.nseg;
        float2 *agg = reinterpret_cast<float2 *>(workspace);
        float *carry = reinterpret_cast<float *>(agg + (size_t)batch * k * p.nseg * d);
        dim3 grid((unsigned)((nwaves + 3) / 4)), block(256);
#define SEG0_(T, TY, NK_, SP_)                                                                                   \
    hipLaunchKernelGGL((ss2d_seg_kernel<T, TY, NK_, SP_, 0>), grid, block, 0, s, (const T *)x, xdbl, table, dt_w, \
                       dt_bias, A, Ds, (TY *)ys, agg, (const float *)nullptr, l, d, k, r, ct, p.nt, p.nseg, nwaves)
#define SEG1_(T, TY, NK_, SP_)                                                                                   \
    hipLaunchKernelGGL((ss2d_seg_kernel<T, TY, NK_, SP_, 1>), grid, block, 0, s, (const T *)x, xdbl, table, dt_w, \
                       dt_bias, A, Ds, (TY *)ys, agg, (const float *)(p.nseg > 1 ? carry : nullptr), l, d, k, r, \
                       ct, p.nt, p.nseg, nwaves)
        if (p.nseg > 1) {
            BY_DTYPE_(SEG0_)
        }
        BY_DTYPE_(SEG1_)
#undef SEG0_
#undef SEG1_
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    // ---- chained form (no workspace): one workgroup of W waves per sequence
    // W waves per sequence: as many as keep the whole launch resident at once (2 waves per SIMD at this
    // register budget = 2048 wave slots) -- beyond that, extra waves only add barrier partners and idle tile
    // slots in the last super-chunk (measured, scripts/probe_scan_latency.py: 24x24 K=8 is 1.8x faster at W=2).
    int W = kMaxW;
    while (W > 1 && (long)batch * k * ct * W > 2048) W >>= 1;
    // ... and at least ~4 tiles per wave (r03 sweep, mailbox carry: 24x24 at batch 1, 128 sequences of 18 tiles: 11.9 us at
    // W = 4 against 14.5 at W = 8; 12x12, 5 tiles: one wave per sequence)
    while (W > 1 && (l + kTP - 1) / kTP < 4 * W) W >>= 1;
    if (W > (l + kTP - 1) / kTP) W = (l + kTP - 1) / kTP;
    if (tramba_tune_get(TRAMBA_TUNE_SCAN_W) > 0) {   // A/B timing hook (scripts/bench_scan.py)
        W = tramba_tune_get(TRAMBA_TUNE_SCAN_W) < kMaxW ? tramba_tune_get(TRAMBA_TUNE_SCAN_W) : kMaxW;
        if (W > (l + kTP - 1) / kTP) W = (l + kTP - 1) / kTP;
    }
    dim3 grid(ct, k, batch), block(W * kWave);
#define GO_(T, TY, NK_, SP_) \
    launch_scan_ring<T, TY, NK_, SP_>(grid, block, s, x, xdbl, table, dt_w, dt_bias, A, Ds, ys, l, d, k, r, W, states, a_log)
    BY_DTYPE_(GO_)
#undef GO_
#undef BY_DTYPE_
#undef BY_NK_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" size_t tramba_ss2d_scan_bwd_workspace(int batch, int l, int d, int k)
{
    if (batch <= 0 || l <= 0 || d <= 0 || k <= 0) return 0;
    const size_t ntile = ((size_t)l + kTP - 1) / kTP + kMaxW;   // tiles rounded up to whole super-chunks
    return (size_t)batch * k * ntile * d * sizeof(float);
}

extern "C" int tramba_ss2d_scan_bwd_cl(const void *x, const float *xdbl, const int32_t *table, const float *dt_w,
                                       const float *dt_bias, const float *A, const float *Ds, const void *gym,
                                       void *gu, void *graw, float *gB, float *gC, int bc_stride, float *gpar,
                                       void *workspace, size_t workspace_bytes, int have_states, int batch, int l, int d,
                                       int k, int r, int dtype, int gym_dtype, int flags, void *stream)
{
    TRAMBA_CHECK(bc_stride >= 1, "ss2d_scan_bwd_cl: bc_stride must be >= 1");
    TRAMBA_CHECK(x && xdbl && table && dt_w && dt_bias && A && Ds && gym && gu && graw && gB && gC && gpar && workspace,
                 "ss2d_scan_bwd_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && l > 0 && d > 0 && k > 0 && r > 0, "ss2d_scan_bwd_cl: empty shape");
    TRAMBA_CHECK(batch <= 65535 && k <= 65535 && r <= 64, "ss2d_scan_bwd_cl: B, K or dt_rank exceeds this build's limits");
    TRAMBA_CHECK(gym_dtype == TRAMBA_F32 || gym_dtype == dtype, "ss2d_scan_bwd_cl: gym must be f32 or the activation dtype");
    TRAMBA_CHECK(aligned16(xdbl) && aligned16(workspace), "ss2d_scan_bwd_cl: xdbl / workspace must be 16-byte aligned");
    TRAMBA_CHECK(workspace_bytes >= tramba_ss2d_scan_bwd_workspace(batch, l, d, k), "ss2d_scan_bwd_cl: workspace too small");
    TRAMBA_CHECK(((double)l + 1024.0) * d * 4.0 < 2147483648.0 && (double)l * k * rg_bytes(r) < 2147483648.0,
                 "ss2d_scan_bwd_cl: L*D too large for 32-bit offsets");
    hipStream_t s = (hipStream_t)stream;
    const int nk = (r + 15) / 16;
    const int ct = (d + kTP - 1) / kTP;
    // SURVEY 8(d), backward of the op this replaces: u, delta, dout read + du, ddelta written per (b, k, d, l) element
    ProfScope prof(TRAMBA_PROF_SCAN_BWD, s, (double)batch * k * l * (double)d * (dtype == TRAMBA_F32 ? 20.0 : 12.0));
    int W = kMaxW;   // ~220-250 VGPRs: 2 waves per SIMD = 2048 wave slots; size W for one resident wavefront of work
    while (W > 1 && (long)batch * k * ct * W > 2048) W >>= 1;
    if (W > (l + kTP - 1) / kTP) W = (l + kTP - 1) / kTP;
    dim3 grid(ct, k, batch), block(W * kWave);
#define BWD_G_(T, NK_, SP_, TG)                                                                                     \
    hipLaunchKernelGGL((ss2d_scan_bwd_cl_kernel<T, NK_, SP_, TG>), grid, block, 0, s, (const T *)x, xdbl, table, dt_w, \
                       dt_bias, A, Ds, (const TG *)gym, (T *)gu, (T *)graw, gB, gC, gpar, (float *)workspace, l, d, k, r, W, bc_stride, \
                       have_states, flags, mailbox_ctl(s))
#define BWD_(T, NK_, SP_)                                                          \
    if (gym_dtype == TRAMBA_F32) { BWD_G_(T, NK_, SP_, float); } else { BWD_G_(T, NK_, SP_, T); }
#define BWD_NK_(T, SP_)                \
    switch (nk) {                      \
    case 1: BWD_(T, 1, SP_); break;    \
    case 2: BWD_(T, 2, SP_); break;    \
    case 3: BWD_(T, 3, SP_); break;    \
    default: BWD_(T, 4, SP_); break;   \
    }
    if (dtype == TRAMBA_F32) { BWD_NK_(float, true) }
    else if (dtype == TRAMBA_BF16) { BWD_NK_(__hip_bfloat16, false) }
    else if (dtype == TRAMBA_F16) { BWD_NK_(__half, false) }
    else {
        set_error("ss2d_scan_bwd_cl: bad dtype %d", dtype);
        return TRAMBA_ERR_ARG;
    }
#undef BWD_NK_
#undef BWD_
#undef BWD_G_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_ss2d_merge_norm_cl(const void *ys, const int32_t *inv_ptr, const int32_t *inv_idx,
                                         const float *ln_w, const float *ln_b, void *y, int batch, int l,
                                         int d, int k, float eps, int act, int ys_dtype, int dtype,
                                         void *stream)
{
    return tramba_ss2d_merge_grad_cl(ys, inv_ptr, inv_idx, ln_w, ln_b, y, nullptr, nullptr, batch, l, d, k, eps, act, ys_dtype,
                                     dtype, stream);
}

extern "C" int tramba_ss2d_merge_grad_cl(const void *ys, const int32_t *inv_ptr, const int32_t *inv_idx,
                                         const float *ln_w, const float *ln_b, void *y, const void *addend,
                                         const void *zpre, int batch, int l, int d, int k, float eps, int act,
                                         int ys_dtype, int dtype, void *stream)
{
    TRAMBA_CHECK(ys && inv_ptr && inv_idx && y && (eps < 0.f || (ln_w && ln_b)), "ss2d_merge_norm_cl: null tensor");
    TRAMBA_CHECK((!addend && !zpre) || (eps < 0.f && zpre && aligned16(zpre) && (!addend || aligned16(addend))),
                 "ss2d_merge_grad_cl: the SiLU-gradient epilogue belongs to a merge-only call (eps < 0) and needs zpre");
    TRAMBA_CHECK(batch > 0 && l > 0 && d > 0 && k > 0, "ss2d_merge_norm_cl: empty shape");
    TRAMBA_CHECK(ys_dtype == TRAMBA_F32 || ys_dtype == dtype, "ss2d_merge_norm_cl: ys must be f32 or dtype");
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)batch * l;
    TRAMBA_CHECK(npix < 2147483647L, "ss2d_merge_norm_cl: B*L exceeds the 32-bit pixel index of this build");
    // vector width: largest of 4/2/1 dividing D such that the row fits kNormMaxIt iterations
    int v = (d % 4 == 0) ? 4 : ((d % 2 == 0) ? 2 : 1);
    TRAMBA_CHECK((d + kWave * v - 1) / (kWave * v) <= kNormMaxIt, "ss2d_merge_norm_cl: D=%d too large", d);
    TRAMBA_CHECK(aligned16(ys) && aligned16(y), "ss2d_merge_norm_cl: tensors must be 16-byte aligned");
    const int nit_need = (d + kWave * v - 1) / (kWave * v);
    const int nit = nit_need <= 1 ? 1 : (nit_need <= 2 ? 2 : (nit_need <= 4 ? 4 : 8));
    // Streaming form (measured, scripts/bench_scan.py): wins where one batch covers a pixel (K <= 4), the row
    // is at most two wave iterations and the map is large enough to fill the chip with 16-pixel waves
    // (6 TB/s on 96x96 D=256); elsewhere one wave per pixel is faster.
    ProfScope prof(TRAMBA_PROF_MERGE, s, (double)batch * k * l * (double)d * dtype_size(ys_dtype) +
                                              (double)batch * l * (double)d * dtype_size(dtype));
    const bool sum_only = eps < 0.f;   // no LayerNorm: only the per-pixel form implements it
    const int tune = tramba_tune_get(TRAMBA_TUNE_MERGE_FORM);
    const bool stream_ok = !sum_only && nit <= 2 && k <= 8 && (double)k * l * d * 4.0 < 4294967296.0;
    const bool stream_form = stream_ok && (tune == 2 || (tune == 0 && k <= 4 && npix >= 8192));
    const bool wide = k > 4;               // 8 rows per batch, 4 pixels per wave (a Helix pixel lists >= 6 entries)
    const int pw = wide ? 4 : 16;
    const int nchunk = (l + pw - 1) / pw;
    const long nwaves = (long)batch * nchunk;
    int hs = 0;                            // many-to-one tables on square maps: image rows centre-out
    if (wide) {
        hs = 1;
        while ((long)hs * hs < l) ++hs;
        if ((long)hs * hs != l || hs % pw != 0) hs = 0;
    }
    // split-row form: a wide row (>= 4 wave iterations) on a map too small to fill the chip one wave per pixel
    // (bijective K = 4 orders, or fp32 rows: on the Helix tables with 2-byte rows the one-wave-per-pixel form with 16-byte
    // accesses is faster -- 19.6 vs 26.5 us at 24x24 D=1024 K=8 -- while at K = 4 the split form wins, 11.7 vs 17.3 us)
    const bool split_form = !sum_only && !stream_form && tune != 1 && (ys_dtype == TRAMBA_F32 || k <= 4) && nit >= 4 &&
                            npix <= 16384;
    if (!stream_form && !split_form) {
        // one wave per pixel, deep row pipeline (Helix tables, small maps, merge-only): widest access that keeps 64 lanes
        // busy, 16 bytes at most
        const int vd = norm_vec(d, ys_dtype == TRAMBA_F32 ? 4 : 8);
        const int nd_need = (d + kWave * vd - 1) / (kWave * vd);
        TRAMBA_CHECK(nd_need <= kNormMaxIt, "ss2d_merge_norm_cl: D=%d too large", d);
        TRAMBA_CHECK((double)k * l * d * 4.0 < 4294967296.0, "ss2d_merge_norm_cl: K*L*D too large for 32-bit offsets");
        const int nd = nd_need <= 1 ? 1 : (nd_need <= 2 ? 2 : (nd_need <= 4 ? 4 : 8));
        int hh = 1;
        while ((long)hh * hh < l) ++hh;
        if ((long)hh * hh != l) hh = 0;   // not a square map: plain pixel order
        dim3 gridd((unsigned)((npix + 3) / 4)), blockd(256);
#define DEEP_(TY, T, V_, N_)                                                                                      \
    hipLaunchKernelGGL((ss2d_merge_norm_deep_kernel<TY, T, V_, N_>), gridd, blockd, 0, s, (const TY *)ys, inv_ptr, \
                       inv_idx, ln_w, ln_b, (T *)y, npix, batch, l, d, k, hh, eps, act, (const T *)addend, (const T *)zpre)
#define DEEP_N_(TY, T, V_)                    \
    if (nd == 1) { DEEP_(TY, T, V_, 1); }     \
    else if (nd == 2) { DEEP_(TY, T, V_, 2); } \
    else if (nd == 4) { DEEP_(TY, T, V_, 4); } \
    else { DEEP_(TY, T, V_, 8); }
#define DEEP_V_(TY, T)                                                  \
    if (vd == 8) {                                                      \
        if constexpr (sizeof(TY) == 2) { DEEP_N_(TY, T, 8) }            \
    } else if (vd == 4) { DEEP_N_(TY, T, 4) }                           \
    else if (vd == 2) { DEEP_N_(TY, T, 2) }                             \
    else { DEEP_N_(TY, T, 1) }
        TRAMBA_DISPATCH_DTYPE(dtype, T, {
            if (ys_dtype == TRAMBA_F32) { DEEP_V_(float, T) } else { DEEP_V_(T, T) }
        });
#undef DEEP_V_
#undef DEEP_N_
#undef DEEP_
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    dim3 grid(stream_form ? (unsigned)((nwaves + 3) / 4) : (unsigned)npix), block(256);
#define GO_(TY, T, V_, N_)                                                                                    \
    if (split_form)                                                                                           \
        hipLaunchKernelGGL((ss2d_merge_norm_split_kernel<TY, T, V_, (N_ >= 4 ? N_ : 4), 4>), grid, block, 0, s, \
                           (const TY *)ys, inv_ptr, inv_idx, ln_w, ln_b, (T *)y, npix, l, d, k, eps, act);    \
    else if (wide)                                                                                            \
        hipLaunchKernelGGL((ss2d_merge_norm_stream_kernel<TY, T, V_, (N_ <= 2 ? N_ : 2), 8>), grid, block, 0, s, \
                           (const TY *)ys, inv_ptr, inv_idx, ln_w, ln_b, (T *)y, nwaves, nchunk, pw, l, d, k, eps, \
                           act, batch, hs);                                                                   \
    else                                                                                                      \
        hipLaunchKernelGGL((ss2d_merge_norm_stream_kernel<TY, T, V_, (N_ <= 2 ? N_ : 2), 4>), grid, block, 0, s, \
                           (const TY *)ys, inv_ptr, inv_idx, ln_w, ln_b, (T *)y, nwaves, nchunk, pw, l, d, k, eps, \
                           act, batch, 0)
#define BY_N_(TY, T, V_)               \
    if (nit == 1) { GO_(TY, T, V_, 1); }   \
    else if (nit == 2) { GO_(TY, T, V_, 2); } \
    else if (nit == 4) { GO_(TY, T, V_, 4); } \
    else { GO_(TY, T, V_, 8); }
#define BY_V_(TY, T)                   \
    if (v == 4) { BY_N_(TY, T, 4) }    \
    else if (v == 2) { BY_N_(TY, T, 2) } \
    else { BY_N_(TY, T, 1) }
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (ys_dtype == TRAMBA_F32) { BY_V_(float, T) } else { BY_V_(T, T) }
    });
#undef BY_V_
#undef BY_N_
#undef GO_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_ss2d_group_stride(int r) { return tramba::xdbl_group_stride(r); }
