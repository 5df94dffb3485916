// Selective scan in the reference's own tensor layout (the L0 drop-in boundary).
//
// Replaces selective_scan_cuda_oflex.fwd / .bwd (call sites Models/SS2D/csms6s.py:910,
// :920-922).  Layout: u, delta, out (B, KD, L) with L contiguous; B/C (B, K, N, L).
//
// MI355X mapping.  The op is HBM-bound (8 B/element at bf16-in/fp32-out, ~10 flop): the
// design goal is full-width coalesced streams with enough waves in flight, not MFMA.
//   * one row (b, d) is owned by LPR consecutive lanes of a wave (LPR = 64, or 32/16 for
//     short rows so a wave carries 2/4 rows); a chunk is LPR*E consecutive positions and
//     each lane moves its E=8 elements with 16-byte accesses -> every wave-level load/store
//     covers contiguous 1 KiB (bf16) / 2 KiB (fp32) segments;
//   * inside a chunk: per-lane serial recurrence over E elements, then a log-step
//     wave-level scan of the (decay, state) pairs with DPP/shuffle (the recurrence
//     h = a h + b is associative under (a2 a1, a2 b1 + b2)), then the per-lane replay
//     with the now-known carry-in; the chunk carry is broadcast from the row's last lane;
//   * the next chunk's operands are fetched before the current chunk's scan so the
//     dependent carry chain never exposes HBM latency;
//   * state in fp32; softplus threshold 20 like the reference; exp via v_exp_f32.
// Chunk-end states go to `ckpt` for the backward (same chunking), which replays each chunk
// forward from its checkpoint and runs the adjoint recurrence right-to-left.
#include "common.h"

namespace tramba {

constexpr int kE = 8;        // elements per lane per chunk
constexpr int kMaxN = 4;     // d_state supported by the generic path (Tramba uses 1)
constexpr int kWavesPerBlock = 4;

__host__ __device__ inline int lanes_per_row(int l)
{
    const int need = (l + kE - 1) / kE;
    return need <= 16 ? 16 : (need <= 32 ? 32 : 64);
}

// DPP data movement (gfx9 family): a VALU-rate cross-lane read, no LDS crossbar round trip.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, ROW_MASK, 0xf, false));
}

// one step of the (decay, state) scan: lanes that receive a valid source combine, the others see the
// identity (1, 0) through the `old` operand
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_scan_step(float &a, float &h)
{
    const float ap = dpp_move<CTRL, ROW_MASK>(1.f, a);
    const float hp = dpp_move<CTRL, ROW_MASK>(0.f, h);
    h = fmaf(a, hp, h);
    a *= ap;
}

// inclusive scan over all 64 lanes: row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast15 into
// rows 1 and 3, then row_bcast31 into rows 2 and 3
__device__ __forceinline__ void wave_scan_dpp(float &a, float &h)
{
    dpp_scan_step<0x111, 0xf>(a, h);
    dpp_scan_step<0x112, 0xf>(a, h);
    dpp_scan_step<0x114, 0xf>(a, h);
    dpp_scan_step<0x118, 0xf>(a, h);
    dpp_scan_step<0x142, 0xa>(a, h);
    dpp_scan_step<0x143, 0xc>(a, h);
}

// inclusive scan of (a, h) pairs over the LPR lanes that own one row
template <int LPR>
__device__ __forceinline__ void row_scan(float &a, float &h, int sub)
{
    if constexpr (LPR == 64) {
        wave_scan_dpp(a, h);
        return;
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
        const float ap = __shfl_up(a, o, LPR);
        const float hp = __shfl_up(h, o, LPR);
        if (sub >= o) {
            h = fmaf(a, hp, h);
            a *= ap;
        }
    }
}

// softplus with the reference's threshold semantics, lean form: ln2*log2(1 + exp2(min(x,60)*log2e))
// max-ed with x (equal to "x > 20 -> x" to below fp32 resolution), 2 transcendentals.
__device__ __forceinline__ float softplus_lean(float x)
{
    const float z = __builtin_amdgcn_exp2f(fminf(x, 60.f) * 1.44269504088896f);
    return fmaxf(x, __builtin_amdgcn_logf(1.f + z) * 0.693147180559945f);
}

// VEC: L % 8 == 0 and 16-byte aligned tensors (every shape of the model) -> 16-byte accesses only.
template <typename T, typename TO, int LPR, int N, bool VEC>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void selective_scan_fwd_kernel(
    const T *__restrict__ u, const T *__restrict__ delta, const float *__restrict__ A,
    const T *__restrict__ Bm, const T *__restrict__ Cm, const float *__restrict__ Dskip,
    const float *__restrict__ dbias, TO *__restrict__ out, float *__restrict__ ckpt, int rows_total,
    int kd, int K, int L, int nchunk, int softplus)
{
    constexpr bool vec_ok = VEC;
    constexpr int RPW = kWave / LPR;  // rows per wave
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int sub = lane % LPR;
    const long row = (long)wave * RPW + lane / LPR;
    const bool row_ok = row < rows_total;
    const long rrow = row_ok ? row : 0;
    const int b = (int)(rrow / kd), d = (int)(rrow % kd);
    const int k = d / (kd / K);

    const T *ur = u + rrow * L;
    const T *dr = delta + rrow * L;
    TO *yr = out + rrow * L;
    const T *Br = Bm + ((long)b * K + k) * N * L;
    const T *Cr = Cm + ((long)b * K + k) * N * L;

    float A2[N];  // A * log2(e): decay = exp2(dt * A2)
#pragma unroll
    for (int n = 0; n < N; ++n) A2[n] = A[(long)d * N + n] * 1.44269504088896f;
    const float bias = dbias ? dbias[d] : 0.f;
    const float skip = Dskip ? Dskip[d] : 0.f;

    float carry[N];
#pragma unroll
    for (int n = 0; n < N; ++n) carry[n] = 0.f;

    // Operand pipeline: a static ring of NS stages holding the RAW packed loads (converted on use), the
    // chunk loop unrolled by NS so stage indices are compile-time; loads are unconditional from clamped
    // offsets (masking happens in the arithmetic).  Two chunks stay in flight under the current one.
    constexpr int NS = (N == 1 && VEC) ? 3 : 2;
    struct Stage {
        Pack<T, kE> u, d, B[N], C[N];
    };
    Stage ring[NS];
    const int lmax = L >= kE ? L - kE : 0;
    auto fetch = [&](int chunk, Stage &st) {
        int l0 = (chunk < nchunk ? chunk : nchunk - 1) * (LPR * kE) + sub * kE;
        if (l0 > lmax) l0 = lmax;  // lanes past the row end re-read the tail; their results are masked
        if constexpr (VEC) {
            st.u = *reinterpret_cast<const Pack<T, kE> *>(ur + l0);
            st.d = *reinterpret_cast<const Pack<T, kE> *>(dr + l0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                st.B[n] = *reinterpret_cast<const Pack<T, kE> *>(Br + (long)n * L + l0);
                st.C[n] = *reinterpret_cast<const Pack<T, kE> *>(Cr + (long)n * L + l0);
            }
        } else {  // unaligned / ragged rows: element loads, clamped per element
            const int lb = chunk * (LPR * kE) + sub * kE;
#pragma unroll
            for (int j = 0; j < kE; ++j) {
                const int l = lb + j < L ? lb + j : L - 1;
                st.u.v[j] = ur[l];
                st.d.v[j] = dr[l];
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    st.B[n].v[j] = Br[(long)n * L + l];
                    st.C[n].v[j] = Cr[(long)n * L + l];
                }
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) fetch(i, ring[i]);

    for (int c0 = 0; c0 < nchunk; c0 += NS) {
#pragma unroll
        for (int si = 0; si < NS; ++si) {
            const int chunk = c0 + si;
            if (chunk >= nchunk) break;  // wave-uniform
            fetch(chunk + NS - 1, ring[(si + NS - 1) % NS]);
            const Stage &st = ring[si];
            const int l0 = chunk * (LPR * kE) + sub * kE;
            // only the last chunk of a row can be ragged (wave-uniform test): dt = 0 is the identity (a=1, b=0)
            const bool ragged = (chunk + 1) * (LPR * kE) > L;
            float cu[kE], dt[kE];
#pragma unroll
            for (int j = 0; j < kE; ++j) {
                cu[j] = Cvt<T>::to_f(st.u.v[j]);
                const float x = Cvt<T>::to_f(st.d.v[j]) + bias;
                dt[j] = softplus ? softplus_lean(x) : x;
            }
            if (ragged) {
#pragma unroll
                for (int j = 0; j < kE; ++j)
                    if (l0 + j >= L) dt[j] = 0.f;
            }
            float y[kE];
#pragma unroll
            for (int j = 0; j < kE; ++j) y[j] = skip * cu[j];
#pragma unroll
            for (int n = 0; n < N; ++n) {
                float a[kE], bb[kE];
                float pa = 1.f, ph = 0.f;
#pragma unroll
                for (int j = 0; j < kE; ++j) {
                    a[j] = __builtin_amdgcn_exp2f(dt[j] * A2[n]);
                    bb[j] = dt[j] * (Cvt<T>::to_f(st.B[n].v[j]) * cu[j]);
                    ph = fmaf(a[j], ph, bb[j]);
                    pa *= a[j];
                }
                row_scan<LPR>(pa, ph, sub);
                // exclusive prefix for this lane, then fold in the chunk carry
                float ea, eh, la, lh;
                if constexpr (LPR == 64) {
                    ea = dpp_move<0x138, 0xf>(1.f, pa);  // wave_shr:1, lane 0 keeps the identity
                    eh = dpp_move<0x138, 0xf>(0.f, ph);
                    la = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pa), 63));
                    lh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ph), 63));
                } else {
                    ea = __shfl_up(pa, 1, LPR);
                    eh = __shfl_up(ph, 1, LPR);
                    if (sub == 0) { ea = 1.f; eh = 0.f; }
                    la = __shfl(pa, LPR - 1, LPR);
                    lh = __shfl(ph, LPR - 1, LPR);
                }
                float h = fmaf(ea, carry[n], eh);
                // new chunk carry = inclusive prefix of the row's last lane applied to the old carry
                carry[n] = fmaf(la, carry[n], lh);
#pragma unroll
                for (int j = 0; j < kE; ++j) {
                    h = fmaf(a[j], h, bb[j]);
                    y[j] = fmaf(Cvt<T>::to_f(st.C[n].v[j]), h, y[j]);
                }
                if (ckpt && row_ok && sub == 0) ckpt[(rrow * nchunk + chunk) * N + n] = carry[n];
            }
            if (row_ok) {
                if (vec_ok) {
                    if (l0 + kE <= L) store_pack<TO, kE>(yr + l0, y);
                } else {
#pragma unroll
                    for (int j = 0; j < kE; ++j)
                        if (l0 + j < L) yr[l0 + j] = Cvt<TO>::from_f(y[j]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ backward
// Adjoint of the recurrence, per state component n (N = 1, 2, 4 instantiated).  With g_l = dLoss/dh_l:
//     g_l      = C_l dout_l + a_{l+1} g_{l+1}                      (right-to-left scan)
//     d dt_l   = g_l (h_{l-1} a_l A + B_l u_l)        d u_l = dout_l D + g_l dt_l B_l
//     dB_l    += g_l dt_l u_l   dC_l += dout_l h_l    dA += g_l h_{l-1} a_l dt_l   dD += dout_l u_l
//     d delta  = d dt * sigmoid(delta + bias)  (1 beyond the softplus threshold)
// Chunks are walked last-to-first; each chunk is replayed forward from its checkpoint (row_scan)
// to recover h_{l-1}, then the adjoint runs as a mirrored wave scan.  dB/dC are shared by the
// KD/K rows of a group: they are reduced with fp32 atomics (one add per element); dA, dD and
// d delta_bias are reduced per row in registers, then one atomic per row.
template <int LPR>
__device__ __forceinline__ void row_scan_rev(float &a, float &g, int sub)
{
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
        const float an = __shfl_down(a, o, LPR);
        const float gn = __shfl_down(g, o, LPR);
        if (sub + o < LPR) {
            g = fmaf(a, gn, g);
            a *= an;
        }
    }
}

template <int LPR>
__device__ __forceinline__ float row_sum(float v)
{
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
    return v;
}

template <typename T, int LPR, int N>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void selective_scan_bwd_kernel(
    const T *__restrict__ u, const T *__restrict__ delta, const float *__restrict__ A,
    const T *__restrict__ Bm, const T *__restrict__ Cm, const float *__restrict__ Dskip,
    const float *__restrict__ dbias, const float *__restrict__ dout, const float *__restrict__ ckpt,
    T *__restrict__ du, T *__restrict__ ddelta, float *__restrict__ dA, float *__restrict__ dB,
    float *__restrict__ dC, float *__restrict__ dD, float *__restrict__ ddbias, int rows_total, int kd,
    int K, int L, int nchunk, int softplus, int vec_ok, int ncopy, long copy_stride)
{
    // N state components per row (d_state): the recurrences are independent, the gradients of u and delta sum over them;
    // B, C (B, K, N, L), A (KD, N), the forward's checkpoints (rows, nchunk, N).
    constexpr int RPW = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int sub = lane % LPR;
    const long row = (long)wave * RPW + lane / LPR;
    const bool row_ok = row < rows_total;
    const long rrow = row_ok ? row : 0;
    const int b = (int)(rrow / kd), d = (int)(rrow % kd);
    const int k = d / (kd / K);

    const T *ur = u + rrow * L;
    const T *dr = delta + rrow * L;
    const float *gr = dout + rrow * L;
    const long bc = ((long)b * K + k) * N * L;
    const T *Br = Bm + bc;
    const T *Cr = Cm + bc;
    // dB/dC are shared by the KD/K rows of a group: spread the adds over `ncopy` private copies
    // (summed by the caller) so that KD/K/ncopy, not KD/K, rows contend for one address
    const long pc = (long)(d % ncopy) * copy_stride;
    float An[N];
#pragma unroll
    for (int n = 0; n < N; ++n) An[n] = A[(long)d * N + n];
    const float bias = dbias ? dbias[d] : 0.f;
    const float skip = Dskip ? Dskip[d] : 0.f;

    // dB/dC contributions leave through LDS so that every atomic wave-instruction adds 64 CONSECUTIVE
    // floats (256 contiguous bytes): in registers lane i holds elements 8i..8i+7, which as atomics would be
    // 64 scattered 4-byte adds per instruction (an order of magnitude slower, MI355X global float atomics)
    __shared__ float tr[kWavesPerBlock][2][kWave * kE];
    float *trB = tr[threadIdx.x >> 6][0], *trC = tr[threadIdx.x >> 6][1];

    float g_carry[N], a_carry[N], accA[N];   // g / a of the first element of the chunk to the right, per component
#pragma unroll
    for (int n = 0; n < N; ++n) {
        g_carry[n] = 0.f;
        a_carry[n] = 1.f;
        accA[n] = 0.f;
    }
    float accD = 0.f, accBias = 0.f;

    for (int chunk = nchunk - 1; chunk >= 0; --chunk) {
        const int l0 = chunk * (LPR * kE) + sub * kE;
        float cu[kE], cd[kE], go[kE];
        if (vec_ok && l0 + kE <= L) {
            load_pack<T, kE>(ur + l0, cu);
            load_pack<T, kE>(dr + l0, cd);
            load_pack<float, kE>(gr + l0, go);
        } else {
#pragma unroll
            for (int j = 0; j < kE; ++j) {
                const bool ok = l0 + j < L;
                cu[j] = ok ? Cvt<T>::to_f(ur[l0 + j]) : 0.f;
                cd[j] = ok ? Cvt<T>::to_f(dr[l0 + j]) : 0.f;
                go[j] = ok ? gr[l0 + j] : 0.f;
            }
        }
        float dt[kE], raw[kE], odu[kE], ddt[kE];
#pragma unroll
        for (int j = 0; j < kE; ++j) {
            raw[j] = cd[j] + bias;
            dt[j] = softplus ? softplus20(raw[j]) : raw[j];
            if (l0 + j >= L) dt[j] = 0.f;
            odu[j] = go[j] * skip;
            ddt[j] = 0.f;
            accD = fmaf(go[j], cu[j], accD);
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float cB[kE], cC[kE];
            if (vec_ok && l0 + kE <= L) {
                load_pack<T, kE>(Br + (long)n * L + l0, cB);
                load_pack<T, kE>(Cr + (long)n * L + l0, cC);
            } else {
#pragma unroll
                for (int j = 0; j < kE; ++j) {
                    const bool ok = l0 + j < L;
                    cB[j] = ok ? Cvt<T>::to_f(Br[(long)n * L + l0 + j]) : 0.f;
                    cC[j] = ok ? Cvt<T>::to_f(Cr[(long)n * L + l0 + j]) : 0.f;
                }
            }
            float a[kE], bb[kE];
            float pa = 1.f, ph = 0.f;
#pragma unroll
            for (int j = 0; j < kE; ++j) {
                a[j] = __expf(dt[j] * An[n]);
                bb[j] = dt[j] * cB[j] * cu[j];
                ph = fmaf(a[j], ph, bb[j]);
                pa *= a[j];
            }
            // forward replay from the checkpoint: state entering this lane
            row_scan<LPR>(pa, ph, sub);
            float ea = __shfl_up(pa, 1, LPR), eh = __shfl_up(ph, 1, LPR);
            if (sub == 0) { ea = 1.f; eh = 0.f; }
            const float hstart = chunk > 0 ? ckpt[(rrow * nchunk + chunk - 1) * N + n] : 0.f;
            float h = fmaf(ea, hstart, eh);
            float hprev[kE], hcur[kE];
#pragma unroll
            for (int j = 0; j < kE; ++j) {
                hprev[j] = h;
                h = fmaf(a[j], h, bb[j]);
                hcur[j] = h;
            }
            // adjoint scan: element j carries (a_{j+1}, C_j dout_j)
            float a_right = __shfl_down(a[0], 1, LPR);          // first a of the lane to the right
            if (sub == LPR - 1) a_right = a_carry[n];
            float an[kE], cg[kE];
#pragma unroll
            for (int j = 0; j < kE; ++j) {
                an[j] = j + 1 < kE ? a[j + 1] : a_right;
                cg[j] = cC[j] * go[j];
            }
            float qa = 1.f, qg = 0.f;
#pragma unroll
            for (int j = kE - 1; j >= 0; --j) {
                qg = fmaf(an[j], qg, cg[j]);
                qa *= an[j];
            }
            row_scan_rev<LPR>(qa, qg, sub);
            float xa = __shfl_down(qa, 1, LPR), xg = __shfl_down(qg, 1, LPR);   // exclusive suffix
            if (sub == LPR - 1) { xa = 1.f; xg = 0.f; }
            float g = fmaf(xa, g_carry[n], xg);
            // carries for the chunk to the left: full-row suffix applied to the old carry
            const float fa = __shfl(qa, 0, LPR), fg = __shfl(qg, 0, LPR);
            g_carry[n] = fmaf(fa, g_carry[n], fg);
            a_carry[n] = __shfl(a[0], 0, LPR);

#pragma unroll
            for (int j = kE - 1; j >= 0; --j) {
                g = fmaf(an[j], g, cg[j]);                       // g_j
                const bool ok = l0 + j < L;
                ddt[j] = fmaf(g, fmaf(hprev[j] * a[j], An[n], cB[j] * cu[j]), ddt[j]);
                odu[j] = fmaf(g * dt[j], cB[j], odu[j]);
                accA[n] = fmaf(g * hprev[j], a[j] * dt[j], accA[n]);
                trB[lane * kE + j] = ok ? g * dt[j] * cu[j] : 0.f;
                trC[lane * kE + j] = ok ? go[j] * hcur[j] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            if (row_ok) {
                // lane i, step j -> element (lane/LPR)*LPR*kE + j*LPR + sub of the wave's staging area, i.e.
                // position chunk*LPR*kE + j*LPR + sub of the row: consecutive lanes, consecutive addresses
                const int rbase = (lane / LPR) * (LPR * kE);
#pragma unroll
                for (int j = 0; j < kE; ++j) {
                    const int e = j * LPR + sub;
                    const int l = chunk * (LPR * kE) + e;
                    if (l < L) {
                        atomicAdd(dB + pc + bc + (long)n * L + l, trB[rbase + e]);
                        atomicAdd(dC + pc + bc + (long)n * L + l, trC[rbase + e]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        float odd[kE];
#pragma unroll
        for (int j = 0; j < kE; ++j) {
            float dr_ = ddt[j];
            if (softplus && raw[j] <= 20.f) dr_ = ddt[j] * sigmoidf_(raw[j]);
            odd[j] = l0 + j < L ? dr_ : 0.f;
            accBias += odd[j];
        }
        if (row_ok) {
            if (vec_ok && l0 + kE <= L) {
                store_pack<T, kE>(du + rrow * L + l0, odu);
                store_pack<T, kE>(ddelta + rrow * L + l0, odd);
            } else {
#pragma unroll
                for (int j = 0; j < kE; ++j)
                    if (l0 + j < L) {
                        du[rrow * L + l0 + j] = Cvt<T>::from_f(odu[j]);
                        ddelta[rrow * L + l0 + j] = Cvt<T>::from_f(odd[j]);
                    }
            }
        }
    }
    accD = row_sum<LPR>(accD);
    accBias = row_sum<LPR>(accBias);
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float sA = row_sum<LPR>(accA[n]);
        if (row_ok && sub == 0) atomicAdd(dA + (long)d * N + n, sA);
    }
    if (row_ok && sub == 0) {
        if (dD) atomicAdd(dD + d, accD);
        if (ddbias) atomicAdd(ddbias + d, accBias);
    }
}

template <typename T, typename TO, int N>
static int launch_fwd(const void *u, const void *delta, const float *A, const void *Bm, const void *Cm,
                      const float *D, const float *bias, void *out, float *ckpt, int batch, int kd,
                      int K, int L, int softplus, hipStream_t s)
{
    const int lpr = lanes_per_row(L);
    const int nchunk = (L + lpr * kE - 1) / (lpr * kE);
    const long rows = (long)batch * kd;
    const int rpw = kWave / lpr;
    const long waves = (rows + rpw - 1) / rpw;
    const long blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    const int vec_ok = (L % kE == 0) && aligned16(u) && aligned16(delta) && aligned16(Bm) &&
                       aligned16(Cm) && aligned16(out);
    dim3 grid((unsigned)blocks), block(kWavesPerBlock * kWave);
#define LAUNCH_(LPR_, VEC_)                                                                             \
    hipLaunchKernelGGL((selective_scan_fwd_kernel<T, TO, LPR_, N, VEC_>), grid, block, 0, s, (const T *)u, \
                       (const T *)delta, A, (const T *)Bm, (const T *)Cm, D, bias, (TO *)out, ckpt,        \
                       (int)rows, kd, K, L, nchunk, softplus)
    if (vec_ok) {
        if (lpr == 16) LAUNCH_(16, true);
        else if (lpr == 32) LAUNCH_(32, true);
        else LAUNCH_(64, true);
    } else {
        if (lpr == 16) LAUNCH_(16, false);
        else if (lpr == 32) LAUNCH_(32, false);
        else LAUNCH_(64, false);
    }
#undef LAUNCH_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_selective_scan_nchunk(int l, int io_dtype)
{
    (void)io_dtype;
    if (l <= 0) return 0;
    const int lpr = lanes_per_row(l);
    return (l + lpr * kE - 1) / (lpr * kE);
}

extern "C" int tramba_selective_scan_fwd(const void *u, const void *delta, const float *A,
                                         const void *Bm, const void *Cm, const float *D,
                                         const float *delta_bias, void *out, float *ckpt, int batch,
                                         int kd, int k, int n, int l, int io_dtype, int out_dtype,
                                         int delta_softplus, void *stream)
{
    TRAMBA_CHECK(u && delta && A && Bm && Cm && out, "selective_scan_fwd: null tensor");
    TRAMBA_CHECK(batch > 0 && kd > 0 && k > 0 && l > 0, "selective_scan_fwd: empty shape");
    TRAMBA_CHECK(kd % k == 0, "selective_scan_fwd: KD=%d is not a multiple of K=%d", kd, k);
    TRAMBA_CHECK(n >= 1 && n <= kMaxN, "selective_scan_fwd: d_state=%d unsupported (1..%d)", n, kMaxN);
    TRAMBA_CHECK(out_dtype == TRAMBA_F32 || out_dtype == io_dtype,
                 "selective_scan_fwd: out dtype must be f32 (oflex) or the input dtype");
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes at the op boundary: u, delta read + out written (+ the small B/C rows)
    const double esz = (double)dtype_size(io_dtype), osz = (double)dtype_size(out_dtype);
    ProfScope prof(TRAMBA_PROF_SCAN_BOUNDARY, s,
                   (double)batch * kd * l * (2.0 * esz + osz) + 2.0 * batch * k * n * (double)l * esz);
#define GO_(T, TO, N_) \
    return launch_fwd<T, TO, N_>(u, delta, A, Bm, Cm, D, delta_bias, out, ckpt, batch, kd, k, l, delta_softplus, s)
#define BY_N_(T, TO)                                                           \
    switch (n) {                                                               \
    case 1: GO_(T, TO, 1);                                                     \
    case 2: GO_(T, TO, 2);                                                     \
    case 4: GO_(T, TO, 4);                                                     \
    default: set_error("selective_scan_fwd: d_state=%d not instantiated (1,2,4)", n); \
             return TRAMBA_ERR_UNSUPPORTED;                                    \
    }
    TRAMBA_DISPATCH_DTYPE(io_dtype, T, {
        if (out_dtype == TRAMBA_F32) { BY_N_(T, float) } else { BY_N_(T, T) }
    });
#undef BY_N_
#undef GO_
    return TRAMBA_OK;
}

extern "C" int tramba_selective_scan_bwd(const void *u, const void *delta, const float *A, const void *Bm,
                                         const void *Cm, const float *D, const float *delta_bias,
                                         const float *dout, const float *ckpt, void *du, void *ddelta,
                                         float *dA, float *dB, float *dC, float *dD, float *ddelta_bias,
                                         int batch, int kd, int k, int n, int l, int io_dtype,
                                         int delta_softplus, int ncopy, void *stream)
{
    TRAMBA_CHECK(u && delta && A && Bm && Cm && dout && ckpt && du && ddelta && dA && dB && dC,
                 "selective_scan_bwd: null tensor");
    TRAMBA_CHECK(batch > 0 && kd > 0 && k > 0 && l > 0, "selective_scan_bwd: empty shape");
    TRAMBA_CHECK(kd % k == 0, "selective_scan_bwd: KD=%d is not a multiple of K=%d", kd, k);
    TRAMBA_CHECK(ncopy >= 1, "selective_scan_bwd: ncopy must be >= 1");
    if (n != 1 && n != 2 && n != 4) {
        set_error("selective_scan_bwd: d_state=%d not instantiated (1,2,4)", n);
        return TRAMBA_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    const int lpr = lanes_per_row(l);
    const int nchunk = (l + lpr * kE - 1) / (lpr * kE);
    const long rows = (long)batch * kd;
    const int rpw = kWave / lpr;
    const long waves = (rows + rpw - 1) / rpw;
    const long blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    const int vec_ok = (l % kE == 0) && aligned16(u) && aligned16(delta) && aligned16(Bm) && aligned16(Cm) &&
                       aligned16(dout) && aligned16(du) && aligned16(ddelta);
    dim3 grid((unsigned)blocks), block(kWavesPerBlock * kWave);
#define LAUNCH_N_(T, LPR_, N_)                                                                                    \
    hipLaunchKernelGGL((selective_scan_bwd_kernel<T, LPR_, N_>), grid, block, 0, s, (const T *)u, (const T *)delta, \
                       A, (const T *)Bm, (const T *)Cm, D, delta_bias, dout, ckpt, (T *)du, (T *)ddelta, dA, dB,    \
                       dC, dD, ddelta_bias, (int)rows, kd, k, l, nchunk, delta_softplus, vec_ok, ncopy,             \
                       (long)batch * k * n * l)
#define LAUNCH_(T, LPR_)                                        \
    if (n == 1) { LAUNCH_N_(T, LPR_, 1); }                      \
    else if (n == 2) { LAUNCH_N_(T, LPR_, 2); }                 \
    else { LAUNCH_N_(T, LPR_, 4); }
    TRAMBA_DISPATCH_DTYPE(io_dtype, T, {
        if (lpr == 16) { LAUNCH_(T, 16) }
        else if (lpr == 32) { LAUNCH_(T, 32) }
        else { LAUNCH_(T, 64) }
    });
#undef LAUNCH_
#undef LAUNCH_N_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
