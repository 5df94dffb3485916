// Channels-last normalisation and depth-wise stencil kernels (all HBM/L2-bound).
//
//   layernorm_cl      LayerNorm2d (Models/modules.py:22-27) without the two permute copies:
//                     in (B, L, C) the normalised axis is already contiguous.
//   shuffle_norm_cl   rearrange 'b (p1 p2 c) h w -> b c (h p1) (w p2)' + LayerNorm2d of
//                     PatchExpand / FinalPatchExpand_X4 / FreqExpand2D (modules.py:209-218,
//                     240-249, 687-696): in channels-last the shuffle only re-addresses whole
//                     C-vectors, so it is fused into the norm's store.
//   dwconv_cl         SS2D's depth-wise 3x3 + SiLU (vmamba.py:283-285) and, with the stencil packed
//                     by dw_pack, DWMSMlp's h + dw3(h) + dw5(h) + dw7(h) -> GELU (vmamba.py:624-625):
//                     the four linear terms are ONE 7x7 depth-wise stencil (identity + 3x3 + 5x5 +
//                     7x7 summed at pack time), so h is read once.
// Lanes map to channels with 4..16-byte accesses.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "norm.h"

namespace tramba {

// rows of C channels; if P > 1 the input is (B, H, W, P*P*C) and row q = ((b*H+h)*W+w)*P*P + p1*P+p2
// is written to output pixel (b, h*P+p1, w*P+p2).
template <typename T, int V>
__global__ __launch_bounds__(256) void layernorm_cl_kernel(const T *__restrict__ x,
                                                          const float *__restrict__ w,
                                                          const float *__restrict__ bvec, T *__restrict__ y,
                                                          long rows, int C, float eps, int act, int P, int H,
                                                          int W)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;
    const int nit = (C + kWave * V - 1) / (kWave * V);
    float acc[kNormMaxIt][V];
    const T *xr = x + row * C;
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it) {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[it][v] = 0.f;
        if (it < nit) {
            const int c0 = (it * kWave + lane) * V;
            if (c0 + V <= C) load_pack<T, V>(xr + c0, acc[it]);
        }
    }
    float mean, rstd;
    wave_layernorm<V>(acc, nit, C, lane, eps, mean, rstd);
    long orow = row;
    if (P > 1) {
        // 32-bit index arithmetic (rows < 2^31, host-checked): six emulated 64-bit divisions per lane were most of
        // this kernel's instructions
        const unsigned r32u = (unsigned)row, pp2 = (unsigned)(P * P);
        const unsigned pp = r32u % pp2, pix = r32u / pp2;
        const unsigned wi = pix % (unsigned)W, bh = pix / (unsigned)W;
        const unsigned hi = bh % (unsigned)H, b = bh / (unsigned)H;
        orow = ((long)b * (H * P) + (long)(hi * P + pp / P)) * (long)(W * P) + (long)(wi * P + pp % P);
    }
    T *yr = y + orow * C;
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it) {
        if (it < nit) {
            const int c0 = (it * kWave + lane) * V;
            if (c0 + V <= C) {
                float o[V];
#pragma unroll
                for (int v = 0; v < V; ++v)
                    o[v] = apply_act((acc[it][v] - mean) * rstd * w[c0 + v] + bvec[c0 + v], act);
                store_pack<T, V>(yr + c0, o);
            }
        }
    }
}

// Short rows (C <= 64*V): LPR lanes per row, 64/LPR rows per wave, 16-byte accesses -- a wave keeps
// 64/LPR independent rows in flight instead of one (the one-row form is latency-bound at C = 128).
template <typename T, int V, int LPR>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bvec, T *__restrict__ y,
                                                            long rows, int C, float eps, int act, int P, int H,
                                                            int W, const float *__restrict__ head_w, float head_b,
                                                            float *__restrict__ head_out)
{
    constexpr int RPW = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane % LPR;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long row = wave * RPW + lane / LPR;
    const bool rok = row < rows;
    const int c0 = sub * V;
    const bool cok = c0 + V <= C;
    float v[V];
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = 0.f;
    if (rok && cok) load_pack<T, V>(x + row * C + c0, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) s += v[i];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float t = cok ? v[i] - mean : 0.f;
        q = fmaf(t, t, q);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, LPR);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!(rok && cok)) return;
    long orow = row;
    if (P > 1) {
        // 32-bit index arithmetic (rows < 2^31, host-checked): six emulated 64-bit divisions per lane were most of
        // this kernel's instructions
        const unsigned r32u = (unsigned)row, pp2 = (unsigned)(P * P);
        const unsigned pp = r32u % pp2, pix = r32u / pp2;
        const unsigned wi = pix % (unsigned)W, bh = pix / (unsigned)W;
        const unsigned hi = bh % (unsigned)H, b = bh / (unsigned)H;
        orow = ((long)b * (H * P) + (long)(hi * P + pp / P)) * (long)(W * P) + (long)(wi * P + pp % P);
    }
    float wv[V], bv[V], o[V];
    load_pack<float, V>(w + c0, wv);
    load_pack<float, V>(bvec + c0, bv);
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = apply_act((v[i] - mean) * rstd * wv[i] + bv[i], act);
    if (head_w) {   // fused 1x1 "segmentation" head (C -> 1): the normalised row never reaches memory
        float hv[V], d = 0.f;
        load_pack<float, V>(head_w + c0, hv);
#pragma unroll
        for (int i = 0; i < V; ++i) d = fmaf(Cvt<T>::to_f(Cvt<T>::from_f(o[i])), hv[i], d);   // as if stored in T
#pragma unroll
        for (int of = LPR / 2; of > 0; of >>= 1) d += __shfl_xor(d, of, LPR);
        if (sub == 0) head_out[orow] = d + head_b;
        return;
    }
    store_pack<T, V>(y + orow * C + c0, o);
}

// Backward of layernorm_rows_kernel's fused head (training path of the last decoder stage, Trambav6.py:132-137: pixel-shuffle
// + LayerNorm over C + the C -> 1 head): with logit = sum_c (xhat_c gamma_c + beta_c) hw_c + hb and q = hw * gamma,
//   dx_c = rstd g (q_c - mean(q) - xhat_c mean(q xhat)),      g = d loss / d logit of the row's output pixel,
// and every parameter gradient follows from two sums over the rows, A_c = sum g xhat_c and G = sum g:
//   d gamma = hw A,  d beta = hw G,  d hw = gamma A + beta G,  d hb = G.
// One pass: x rows in (16-byte accesses, 64 / LPR rows of a wave in flight, a wave walks `passes` consecutive groups), dx rows
// out -- the normalised (B, 4H, 4W, C) map (302 MB at batch 8) and its gradient never exist.  part[workgroup][C + 4]: the
// workgroup's share of A (C floats) and of G (slot C), waves folded in a fixed order.
template <typename T, int V, int LPR>
__global__ __launch_bounds__(256) void norm_head_bwd_rows_kernel(const T *__restrict__ x, const float *__restrict__ g,
                                                                const float *__restrict__ ln_w,
                                                                const float *__restrict__ head_w, T *__restrict__ dx,
                                                                float *__restrict__ part, long rows, int C, float eps, int P,
                                                                int H, int W, int passes)
{
    constexpr int RPW = kWave / LPR;
    __shared__ float red[4][kWave * V + 1];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const int sub = lane % LPR;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wv;
    const int c0 = sub * V;
    const bool cok = c0 + V <= C;
    float q[V], accA[V], accG = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) q[i] = accA[i] = 0.f;
    if (cok) {
        float gam[V], hw[V];
        load_pack<float, V>(ln_w + c0, gam);
        load_pack<float, V>(head_w + c0, hw);
#pragma unroll
        for (int i = 0; i < V; ++i) q[i] = gam[i] * hw[i];
    }
    float qs = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) qs += q[i];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) qs += __shfl_xor(qs, o, LPR);
    const float qbar = qs / (float)C;
    for (int ps = 0; ps < passes; ++ps) {
        const long row = (wave * passes + ps) * RPW + lane / LPR;
        const bool rok = row < rows;
        float v[V];
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = 0.f;
        if (rok && cok) load_pack<T, V>(x + row * C + c0, v);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) s += v[i];
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
        const float mean = s / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float t = cok ? v[i] - mean : 0.f;
            var = fmaf(t, t, var);
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) var += __shfl_xor(var, o, LPR);
        const float rstd = rsqrtf(var / (float)C + eps);
        float xh[V], sq = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            xh[i] = cok ? (v[i] - mean) * rstd : 0.f;
            sq = fmaf(q[i], xh[i], sq);
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, LPR);
        sq /= (float)C;
        float gi = 0.f;
        if (rok) {
            long orow = row;
            if (P > 1) {   // (32-bit index arithmetic, rows < 2^31 host-checked: see layernorm_rows_kernel)
                const unsigned r32u = (unsigned)row, pp2 = (unsigned)(P * P);
                const unsigned pp = r32u % pp2, pix = r32u / pp2;
                const unsigned wi = pix % (unsigned)W, bh = pix / (unsigned)W;
                const unsigned hi = bh % (unsigned)H, b = bh / (unsigned)H;
                orow = ((long)b * (H * P) + (long)(hi * P + pp / P)) * (long)(W * P) + (long)(wi * P + pp % P);
            }
            gi = g[orow];
        }
        if (rok && cok) {
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                o[i] = rstd * gi * (q[i] - qbar - xh[i] * sq);
                accA[i] = fmaf(gi, xh[i], accA[i]);
            }
            store_pack<T, V>(dx + row * C + c0, o);
        }
        if (sub == 0) accG += gi;
    }
    // the wave's RPW row groups, then the four waves in order (fixed: reproducible)
#pragma unroll
    for (int o = kWave / 2; o >= LPR; o >>= 1) {
#pragma unroll
        for (int i = 0; i < V; ++i) accA[i] += __shfl_xor(accA[i], o, kWave);
        accG += __shfl_xor(accG, o, kWave);
    }
    if (lane < LPR) {
#pragma unroll
        for (int i = 0; i < V; ++i) red[wv][c0 + i] = accA[i];
        if (sub == 0) red[wv][kWave * V] = accG;
    }
    __syncthreads();
    float *pr = part + (long)blockIdx.x * (C + 4);
    for (int c = threadIdx.x; c < C + 4; c += blockDim.x) {
        const int src = c < C ? c : kWave * V;
        pr[c] = c <= C ? ((red[0][src] + red[1][src]) + red[2][src]) + red[3][src] : 0.f;
    }
}

// y[row] = <x[row, :], w> + b   (a 1x1 convolution to ONE channel: the decoder's deep-supervision heads)
template <typename T, int V, int LPR>
__global__ __launch_bounds__(256) void rowdot_rows_kernel(const T *__restrict__ x, const float *__restrict__ w, float b,
                                                         float *__restrict__ y, long rows, int C)
{
    constexpr int RPW = kWave / LPR;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane % LPR;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long row = wave * RPW + lane / LPR;
    const bool rok = row < rows;
    float d = 0.f;
    for (int c0 = sub * V; c0 < C; c0 += LPR * V) {   // LPR * V divides C (host-checked)
        float v[V], wv[V];
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = 0.f;
        if (rok) load_pack<T, V>(x + row * C + c0, v);
        load_pack<float, V>(w + c0, wv);
#pragma unroll
        for (int i = 0; i < V; ++i) d = fmaf(v[i], wv[i], d);
    }
#pragma unroll
    for (int of = LPR / 2; of > 0; of >>= 1) d += __shfl_xor(d, of, LPR);
    if (rok && sub == 0) y[row] = d + b;
}

template <typename T>
static int launch_layernorm(const void *x, const float *w, const float *b, void *y, long rows, int c,
                            float eps, int act, int P, int H, int W, hipStream_t s,
                            const float *head_w = nullptr, float head_b = 0.f, float *head_out = nullptr)
{
    constexpr int VM = sizeof(T) == 2 ? 8 : 4;
    if (rows >= 2147483647L) {
        set_error("layernorm: %ld rows exceed the 32-bit row index of this build", rows);
        return TRAMBA_ERR_UNSUPPORTED;
    }
    if (c % VM == 0 && c / VM <= kWave && aligned16(w) && aligned16(b)) {
        const int need = c / VM;
        int lpr = 1;
        while (lpr < need) lpr <<= 1;
        const long waves = (rows + (kWave / lpr) - 1) / (kWave / lpr);
        dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define GOR_(L_)                                                                                            \
    hipLaunchKernelGGL((layernorm_rows_kernel<T, VM, L_>), grid, block, 0, s, (const T *)x, w, b, (T *)y, rows, c, \
                       eps, act, P, H, W, head_w, head_b, head_out)
        switch (lpr) {
        case 1: GOR_(1); break;
        case 2: GOR_(2); break;
        case 4: GOR_(4); break;
        case 8: GOR_(8); break;
        case 16: GOR_(16); break;
        case 32: GOR_(32); break;
        default: GOR_(64); break;
        }
#undef GOR_
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    if (head_w) {
        set_error("shuffle_norm_head: C=%d needs C %% %d == 0 and C <= %d", c, VM, kWave * VM);
        return TRAMBA_ERR_UNSUPPORTED;
    }
    const int maxv = sizeof(T) == 2 ? 8 : 4;
    const int v = norm_vec(c, maxv);
    if ((c + kWave * v - 1) / (kWave * v) > kNormMaxIt) {
        set_error("layernorm: C=%d too large", c);
        return TRAMBA_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define GO_(V_)                                                                                       \
    hipLaunchKernelGGL((layernorm_cl_kernel<T, V_>), grid, block, 0, s, (const T *)x, w, b, (T *)y, rows, c, \
                       eps, act, P, H, W)
    switch (v) {
    case 8: if constexpr (sizeof(T) == 2) { GO_(8); } break;
    case 4: GO_(4); break;
    case 2: GO_(2); break;
    default: GO_(1); break;
    }
#undef GO_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

// row q = ((b*H + h)*W + w)*P*P + p1*P + p2 of a (B, H, W, P*P*C) map -> row of output pixel (b, h*P + p1, w*P + p2): the
// pixel shuffle 'b (p1 p2 c) h w -> b c (h p1) (w p2)' in channels-last rows (32-bit arithmetic, rows < 2^31)
__device__ __forceinline__ long shuffled_row(long row, int P, int H, int W)
{
    if (P <= 1) return row;
    const unsigned r32u = (unsigned)row, pp2 = (unsigned)(P * P);
    const unsigned pp = r32u % pp2, pix = r32u / pp2;
    const unsigned wi = pix % (unsigned)W, bh = pix / (unsigned)W;
    const unsigned hi = bh % (unsigned)H, b = bh / (unsigned)H;
    return ((long)b * (H * P) + (long)(hi * P + pp / P)) * (long)(W * P) + (long)(wi * P + pp % P);
}

// LayerNorm backward (training), channels-last rows: a wave walks RPW rows, all C channels in registers.
//   xh = (x - mean) * rstd,  g = dy * gamma
//   dx = rstd * (g - mean_C(g) - xh * mean_C(g * xh)),   dgamma += dy * xh,   dbeta += dy
// mean / rstd are recomputed from x (cheaper than storing them); dgamma / dbeta accumulate over the wave's rows
// in registers, the four waves of a workgroup are folded in LDS and leave as ONE partial row per workgroup,
// part[block][0 = dgamma, 1 = dbeta][C]; the caller sums the partial rows (fp32 atomics from ~4096 waves onto C addresses serialised in L2: 500 us per call, 10x the rest).
template <typename T, int V>
__global__ __launch_bounds__(256) void layernorm_bwd_cl_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                              const float *__restrict__ w, T *__restrict__ dx,
                                                              float *__restrict__ part, long rows, int C, float eps,
                                                              int rpw, const T *__restrict__ gres,
                                                              const float *__restrict__ mask, long rps,
                                                              T *__restrict__ dxm, int P = 1, int H = 1, int W = 1)
{
    // P > 1: the forward was tramba_shuffle_norm_cl -- dy is indexed by the SHUFFLED row of x's row (shuffled_row)
    // gres / mask / dxm (round 3, the residual stream of a block): dx <- dx + gres, the gradient reaching the block input
    // through the skip connection added here instead of by a separate launch; dxm = that sum * mask[sample], the gradient
    // of the PREVIOUS residual branch under stochastic depth (x = x_prev + branch * mask) -- one pass, two outputs.
    extern __shared__ float ln_red[];          // [4 waves][2 C]: the waves' (dgamma, dbeta) rows, folded before they leave
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wv;
    const long r0 = wave * rpw;                // a wave past the end runs no row and contributes zeros
    const int nit = (C + kWave * V - 1) / (kWave * V);
    float gam[kNormMaxIt][V], gw[kNormMaxIt][V], gb[kNormMaxIt][V];
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it)
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = (it * kWave + lane) * V + v;
            gam[it][v] = (it < nit && c < C) ? w[c] : 0.f;
            gw[it][v] = 0.f;
            gb[it][v] = 0.f;
        }
    const long rend = r0 + rpw < rows ? r0 + rpw : rows;
    for (long r = r0; r < rend; ++r) {
        float xv[kNormMaxIt][V], gv[kNormMaxIt][V];
#pragma unroll
        for (int it = 0; it < kNormMaxIt; ++it) {
#pragma unroll
            for (int v = 0; v < V; ++v) xv[it][v] = gv[it][v] = 0.f;
            if (it < nit) {
                const int c0 = (it * kWave + lane) * V;
                if (c0 + V <= C) {
                    load_pack<T, V>(x + r * C + c0, xv[it]);
                    load_pack<T, V>(dy + shuffled_row(r, P, H, W) * C + c0, gv[it]);
                }
            }
        }
        float mean, rstd;
        wave_layernorm<V>(xv, nit, C, lane, eps, mean, rstd);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < kNormMaxIt; ++it)
            if (it < nit)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const bool ok = (it * kWave + lane) * V + v < C;
                    const float xh = ok ? (xv[it][v] - mean) * rstd : 0.f;
                    const float g = gv[it][v] * gam[it][v];
                    xv[it][v] = xh;          // keep the normalised value
                    s1 += g;
                    s2 = fmaf(g, xh, s2);
                    gw[it][v] = fmaf(gv[it][v], xh, gw[it][v]);
                    gb[it][v] += gv[it][v];
                }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int it = 0; it < kNormMaxIt; ++it)
            if (it < nit) {
                const int c0 = (it * kWave + lane) * V;
                if (c0 + V <= C) {
                    float o[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) o[v] = rstd * (gv[it][v] * gam[it][v] - s1 - xv[it][v] * s2);
                    if (gres) {
                        float rv[V];
                        load_pack<T, V>(gres + r * C + c0, rv);
#pragma unroll
                        for (int v = 0; v < V; ++v) o[v] += rv[v];
                    }
                    store_pack<T, V>(dx + r * C + c0, o);
                    if (dxm) {
                        const float m = mask ? mask[r / rps] : 1.f;
#pragma unroll
                        for (int v = 0; v < V; ++v) o[v] = Cvt<T>::to_f(Cvt<T>::from_f(o[v])) * m;
                        store_pack<T, V>(dxm + r * C + c0, o);
                    }
                }
            }
    }
    float *mine = ln_red + (long)wv * 2 * C;
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it)
        if (it < nit)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int c = (it * kWave + lane) * V + v;
                if (c < C) {
                    mine[c] = gw[it][v];
                    mine[C + c] = gb[it][v];
                }
            }
    __syncthreads();
    // ONE partial row per block, waves added in a fixed order
    float *pw = part + (long)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        pw[i] = (ln_red[i] + ln_red[2 * C + i]) + (ln_red[4 * C + i] + ln_red[6 * C + i]);
}

// Wide 16-bit rows (512 < C <= 2048, C % 8 == 0: out_norm of the 24x24 / 12x12 SS2D cores, D = 1024 / 2048), r03.  The
// general kernel above walks kNormMaxIt = 8 predicated iterations of 8-byte pieces whatever C is: 190 VGPRs (2 waves per
// SIMD) and 34 us for 4608 x 1024 -- 47 MB, ~7 us of HBM time.  Here the iteration count is a template parameter, a lane
// moves 16 bytes per iteration (NIT = 2 at C = 1024), and the operands of the wave's NEXT row are requested before the
// current row is reduced.  Same contract and the same summation order per partial row as layernorm_bwd_cl_kernel.
template <typename T, int NIT>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                                const float *__restrict__ w, T *__restrict__ dx,
                                                                float *__restrict__ part, long rows, int C, float eps,
                                                                int rpw, const T *__restrict__ gres,
                                                                const float *__restrict__ mask, long rps,
                                                                T *__restrict__ dxm, int P = 1, int H = 1, int W = 1)
{
    constexpr int V = 8;
    extern __shared__ float ln_red[];          // [4 waves][2 C]
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wv;
    const long r0 = wave * rpw;
    const long rend = r0 + rpw < rows ? r0 + rpw : rows;
    bool ok[NIT];
    float gam[NIT][V], gw[NIT][V], gb[NIT][V];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c0 = (it * kWave + lane) * V;
        ok[it] = c0 + V <= C;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            gam[it][v] = ok[it] ? w[c0 + v] : 0.f;
            gw[it][v] = 0.f;
            gb[it][v] = 0.f;
        }
    }
    typedef Pack<T, V> Raw;
    Raw rx[NIT], rg[NIT], rr[NIT];
    auto request = [&](long r) {   // raw bits; converted where they are used (a conversion at the load site waits for the load)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c0 = (it * kWave + lane) * V;
            if (ok[it]) {
                rx[it] = *reinterpret_cast<const Raw *>(x + r * C + c0);
                rg[it] = *reinterpret_cast<const Raw *>(dy + shuffled_row(r, P, H, W) * C + c0);
                if (gres) rr[it] = *reinterpret_cast<const Raw *>(gres + r * C + c0);
            }
        }
    };
    if (r0 < rend) request(r0);
    for (long r = r0; r < rend; ++r) {
        float xv[NIT][V], gv[NIT][V], rv[NIT][V];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                xv[it][v] = ok[it] ? Cvt<T>::to_f(rx[it].v[v]) : 0.f;
                gv[it][v] = ok[it] ? Cvt<T>::to_f(rg[it].v[v]) : 0.f;
                rv[it][v] = (ok[it] && gres) ? Cvt<T>::to_f(rr[it].v[v]) : 0.f;
            }
        if (r + 1 < rend) request(r + 1);      // the next row is in flight during this row's three wave reductions
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int v = 0; v < V; ++v) sum += xv[it][v];          // (channels past C hold 0)
        const float mean = wave_sum(sum) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float t = ok[it] ? xv[it][v] - mean : 0.f;
                q = fmaf(t, t, q);
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float xh = ok[it] ? (xv[it][v] - mean) * rstd : 0.f;
                const float g = gv[it][v] * gam[it][v];
                xv[it][v] = xh;
                s1 += g;
                s2 = fmaf(g, xh, s2);
                gw[it][v] = fmaf(gv[it][v], xh, gw[it][v]);
                gb[it][v] += gv[it][v];
            }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
        const float m = (dxm && mask) ? mask[r / rps] : 1.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c0 = (it * kWave + lane) * V;
            if (ok[it]) {
                float o[V];
#pragma unroll
                for (int v = 0; v < V; ++v) o[v] = rstd * (gv[it][v] * gam[it][v] - s1 - xv[it][v] * s2) + rv[it][v];
                store_pack<T, V>(dx + r * C + c0, o);
                if (dxm) {
#pragma unroll
                    for (int v = 0; v < V; ++v) o[v] = Cvt<T>::to_f(Cvt<T>::from_f(o[v])) * m;
                    store_pack<T, V>(dxm + r * C + c0, o);
                }
            }
        }
    }
    float *mine = ln_red + (long)wv * 2 * C;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c0 = (it * kWave + lane) * V;
        if (ok[it]) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                mine[c0 + v] = gw[it][v];
                mine[C + c0 + v] = gb[it][v];
            }
        }
    }
    __syncthreads();
    float *pw = part + (long)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        pw[i] = (ln_red[i] + ln_red[2 * C + i]) + (ln_red[4 * C + i] + ln_red[6 * C + i]);
}

// Short rows (C <= 64 * V, a multiple of V): LPR lanes per row, 64 / LPR rows of a wave in flight at once, 16-byte
// accesses -- the one-row-per-wave form above is latency-bound there (rows of 128 channels keep half a wave idle and
// every row waits for four dependent wave reductions): 1.27 ms for the 384x384 map of the last decoder stage at batch 8.
// Same contract: the wave owns rows [wave * rpw, +rpw); the workgroup leaves ONE partial (dgamma, dbeta) row.
template <typename T, int V, int LPR>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                                const float *__restrict__ w, T *__restrict__ dx,
                                                                float *__restrict__ part, long rows, int C, float eps,
                                                                int rpw, const T *__restrict__ gres,
                                                                const float *__restrict__ mask, long rps,
                                                                T *__restrict__ dxm, int P = 1, int H = 1, int W = 1)
{
    constexpr int RPW = kWave / LPR;
    __shared__ float red[4][2 * LPR * V];     // the four waves' (dgamma, dbeta) rows, folded before they leave the block
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wv;
    const long r0 = wave * rpw;               // a wave past the end runs no row and contributes zeros
    const long rend = r0 + rpw < rows ? r0 + rpw : rows;
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = sub * V;
    const bool cok = c0 + V <= C;
    float gam[V], gw[V], gb[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        gam[v] = cok ? w[c0 + v] : 0.f;
        gw[v] = 0.f;
        gb[v] = 0.f;
    }
    const float inv_c = 1.f / (float)C;
    for (long rb = r0; rb < rend; rb += RPW) {
        const long r = rb + slot;
        const bool ok = r < rend && cok;
        float xv[V], gv[V];
#pragma unroll
        for (int v = 0; v < V; ++v) xv[v] = gv[v] = 0.f;
        if (ok) {
            load_pack<T, V>(x + r * C + c0, xv);
            load_pack<T, V>(dy + shuffled_row(r, P, H, W) * C + c0, gv);
        }
        float sm = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) sm += xv[v];
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o, LPR);
        const float mean = sm * inv_c;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float t = cok ? xv[v] - mean : 0.f;
            q = fmaf(t, t, q);
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, LPR);
        const float rstd = rsqrtf(q * inv_c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float xh = cok ? (xv[v] - mean) * rstd : 0.f;
            const float g = gv[v] * gam[v];
            xv[v] = xh;
            s1 += g;
            s2 = fmaf(g, xh, s2);
            gw[v] = fmaf(gv[v], xh, gw[v]);    // rows past the end were loaded as zeros: they add nothing
            gb[v] += gv[v];
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, LPR);
            s2 += __shfl_xor(s2, o, LPR);
        }
        s1 *= inv_c;
        s2 *= inv_c;
        if (ok) {
            float o_[V];
#pragma unroll
            for (int v = 0; v < V; ++v) o_[v] = rstd * (gv[v] * gam[v] - s1 - xv[v] * s2);
            if (gres) {
                float rv[V];
                load_pack<T, V>(gres + r * C + c0, rv);
#pragma unroll
                for (int v = 0; v < V; ++v) o_[v] += rv[v];
            }
            store_pack<T, V>(dx + r * C + c0, o_);
            if (dxm) {
                const float m = mask ? mask[r / rps] : 1.f;
#pragma unroll
                for (int v = 0; v < V; ++v) o_[v] = Cvt<T>::to_f(Cvt<T>::from_f(o_[v])) * m;
                store_pack<T, V>(dxm + r * C + c0, o_);
            }
        }
    }
    // fold the RPW row slots of the wave (lanes with equal `sub`), then slot 0 writes the partial row
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1)
#pragma unroll
        for (int v = 0; v < V; ++v) {
            gw[v] += __shfl_xor(gw[v], o, kWave);
            gb[v] += __shfl_xor(gb[v], o, kWave);
        }
    if (slot == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            red[wv][c0 + v] = gw[v];
            red[wv][LPR * V + c0 + v] = gb[v];
        }
    }
    __syncthreads();
    // ONE partial row per block (a quarter of the rows the caller has to sum), waves added in a fixed order
    float *pw = part + (long)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int j = i < C ? i : LPR * V + (i - C);
        pw[i] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    }
}

template <typename T>
static int launch_rowdot(const void *x, const float *w, float bias, float *y, long rows, int c, hipStream_t s)
{
    constexpr int VM = sizeof(T) == 2 ? 8 : 4;
    TRAMBA_CHECK(c % VM == 0, "rowdot_cl: C=%d must be a multiple of %d", c, VM);
    int lpr = 1;   // lanes per row: the largest power of two <= 64 with lpr * VM dividing C
    while (lpr < kWave && c % (2 * lpr * VM) == 0) lpr <<= 1;
    const long waves = (rows + (kWave / lpr) - 1) / (kWave / lpr);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define GOD_(L_) hipLaunchKernelGGL((rowdot_rows_kernel<T, VM, L_>), grid, block, 0, s, (const T *)x, w, bias, y, rows, c)
    switch (lpr) {
    case 1: GOD_(1); break;
    case 2: GOD_(2); break;
    case 4: GOD_(4); break;
    case 8: GOD_(8); break;
    case 16: GOD_(16); break;
    case 32: GOD_(32); break;
    default: GOD_(64); break;
    }
#undef GOD_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

// ------------------------------------------------------------------------------ depth-wise stencils
// Weights arrive TAP-MAJOR (ks*ks, C): lanes map to channels, so every tap is one coalesced load
// (the reference's (C,1,ks,ks) layout would make each lane walk its own 49-float row).
// dw_pack_kernel builds that layout once per weight version; for the multi-scale MLP it also folds
// identity + 3x3 + 5x5 + 7x7 (and the three biases) into ONE 7x7 stencil.
__device__ __forceinline__ void dw_pack_body(const float *__restrict__ w, const float *__restrict__ bias,
                                             const float *__restrict__ w3, const float *__restrict__ b3,
                                             const float *__restrict__ w5, const float *__restrict__ b5,
                                             float *__restrict__ wt, float *__restrict__ bt, int C, int ks, int multiscale, int c)
{
    if (c >= C) return;
    const int R = ks / 2;
    for (int dy = 0; dy < ks; ++dy)
        for (int dx = 0; dx < ks; ++dx) {
            float v = w[(long)c * ks * ks + dy * ks + dx];
            if (multiscale) {  // ks == 7
                const int y5 = dy - 1, x5 = dx - 1, y3 = dy - 2, x3 = dx - 2;
                if (y5 >= 0 && y5 < 5 && x5 >= 0 && x5 < 5) v += w5[(long)c * 25 + y5 * 5 + x5];
                if (y3 >= 0 && y3 < 3 && x3 >= 0 && x3 < 3) v += w3[(long)c * 9 + y3 * 3 + x3];
                if (dy == R && dx == R) v += 1.f;
            }
            wt[(long)(dy * ks + dx) * C + c] = v;
        }
    float bv = bias ? bias[c] : 0.f;
    if (multiscale) bv += b3[c] + b5[c];
    bt[c] = bv;
}

__global__ __launch_bounds__(256) void dw_pack_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                     const float *__restrict__ w3, const float *__restrict__ b3,
                                                     const float *__restrict__ w5, const float *__restrict__ b5,
                                                     float *__restrict__ wt, float *__restrict__ bt, int C, int ks,
                                                     int multiscale)
{
    dw_pack_body(w, bias, w3, b3, w5, b5, wt, bt, C, ks, multiscale, blockIdx.x * blockDim.x + threadIdx.x);
}

// every depth-wise stencil of a model in one launch (after the optimizer step: the packs the NEXT forward reads); the
// items travel by value (blockIdx.y = item)
constexpr int kDwMulti = 40;
struct DwPackItem {
    const float *w, *bias, *w3, *b3, *w5, *b5;
    float *wt, *bt;
    int c, ks, ms, pad;
};
struct DwPackArgs {
    DwPackItem it[kDwMulti];
};
static_assert(sizeof(DwPackArgs) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void dw_pack_multi_kernel(DwPackArgs a)
{
    const DwPackItem it = a.it[blockIdx.y];
    dw_pack_body(it.w, it.bias, it.w3, it.b3, it.w5, it.b5, it.wt, it.bt, it.c, it.ks, it.ms, blockIdx.x * blockDim.x + threadIdx.x);
}

// thread = V channels x TW consecutive output columns of one row.
template <typename T, int KS, int V, int TW>
__global__ __launch_bounds__(256) void dwconv_cl_kernel(const T *__restrict__ x, const float *__restrict__ wt,
                                                       const float *__restrict__ bt, T *__restrict__ y, int B,
                                                       int H, int W, int C, int act, long nthreads,
                                                       T *__restrict__ ypre = nullptr, int flip = 0)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nthreads) return;
    const int cg = C / V, wtiles = (W + TW - 1) / TW;
    const int c0 = (int)(tid % cg) * V;
    long rest = tid / cg;
    const int w0 = (int)(rest % wtiles) * TW;
    rest /= wtiles;
    const int h = (int)(rest % H);
    const int b = (int)(rest / H);
    constexpr int R = KS / 2;

    float acc[TW][V];
    {
        float bs[V];
        load_pack<float, V>(bt + c0, bs);
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[t][v] = bs[v];
    }
    const T *xb = x + (long)b * H * W * C + c0;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
        const int hy = h + dy - R;
        if (hy < 0 || hy >= H) continue;
        float wr[KS][V];
#pragma unroll
        for (int dx = 0; dx < KS; ++dx)   // flip: the taps mirrored through the centre (the stencil's adjoint)
            load_pack<float, V>(wt + (long)(flip ? KS * KS - 1 - (dy * KS + dx) : dy * KS + dx) * C + c0, wr[dx]);
        float xin[TW + KS - 1][V];
#pragma unroll
        for (int i = 0; i < TW + KS - 1; ++i) {
            const int wx = w0 + i - R;
            if (wx >= 0 && wx < W) {
                load_pack<T, V>(xb + ((long)hy * W + wx) * C, xin[i]);
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) xin[i][v] = 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[t][v] = fmaf(wr[dx][v], xin[t + dx][v], acc[t][v]);
    }
    T *yb = y + (long)b * H * W * C + c0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (w0 + t < W) {
            float o[V];
            if (ypre) {   // training: the pre-activation leaves too, and the activation is taken of it AS STORED
                store_pack<T, V>(ypre + (long)b * H * W * C + c0 + ((long)h * W + w0 + t) * C, acc[t]);
#pragma unroll
                for (int v = 0; v < V; ++v) acc[t][v] = Cvt<T>::to_f(Cvt<T>::from_f(acc[t][v]));
            }
#pragma unroll
            for (int v = 0; v < V; ++v) o[v] = apply_act(acc[t][v], act);
            store_pack<T, V>(yb + ((long)h * W + w0 + t) * C, o);
        }
    }
}

// thread = 4 channels x TW consecutive output columns of one row.
// The stencil is issue-bound (784 FMAs per thread at 7x7), so nothing but the FMAs and the unavoidable 16-bit
// -> fp32 conversions may cost VALU: every load goes through a buffer descriptor with ONE per-thread byte
// offset per column computed up front (the out-of-range offset where the column falls outside the image);
// the row (dy) and tap offsets ride in the scalar offset; rows outside the image swap in the out-of-range
// offset with one v_cndmask per load.  No 64-bit address arithmetic, no branches around loads.
typedef float dw_v2f __attribute__((ext_vector_type(2)));
typedef float dw_v4f __attribute__((ext_vector_type(4)));
typedef unsigned dw_v2u __attribute__((ext_vector_type(2)));

// 4 channels of one pixel as loaded (raw bits; converted where consumed, so a prefetched row does not make
// hipcc wait for it at the load site)
template <typename T> struct DwRaw { dw_v2u v; };
template <> struct DwRaw<float> { dw_v4f v; };

template <typename T>
__device__ __forceinline__ DwRaw<T> dw_load4(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    DwRaw<T> o;
    if constexpr (sizeof(T) == 4) {
        o.v = __builtin_bit_cast(dw_v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    } else {
        o.v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
    }
    return o;
}

template <typename T>
__device__ __forceinline__ void dw_cvt4(const DwRaw<T> &raw, float (&out)[4])
{
    if constexpr (sizeof(T) == 4) {
        out[0] = raw.v.x; out[1] = raw.v.y; out[2] = raw.v.z; out[3] = raw.v.w;
    } else {
        const Pack<T, 4> pk = __builtin_bit_cast(Pack<T, 4>, raw.v);
#pragma unroll
        for (int v = 0; v < 4; ++v) out[v] = Cvt<T>::to_f(pk.v[v]);
    }
}

template <typename T, int KS, int TW>
__global__ __launch_bounds__(256) void dwconv4_cl_kernel(const T *__restrict__ x, const float *__restrict__ wt,
                                                        const float *__restrict__ bt, T *__restrict__ y, int B,
                                                        int H, int W, int C, int act, T *__restrict__ ypre, int flip)
{
    // flip (training): the taps mirrored through the centre -- the input gradient of the stencil is the same stencil with
    // flipped taps, read in place instead of from a flipped copy made per call
    constexpr int V = 4, R = KS / 2, NI = TW + KS - 1;
    // grid (x: channel-group x column-tile items of one row, y: row, z: image): no 64-bit index arithmetic (the
    // flat-index form spent ~3k cycles per wave in three emulated 64-bit divisions)
    const int cg = C / V, wtiles = (W + TW - 1) / TW;
    // XCD-aware: each XCD takes a contiguous band of rows, so the KS - 1 halo rows an output row shares with its
    // neighbours are re-read from ONE L2 (plain order: a row's readers sit on ~4 XCDs, each fetching the line again)
    unsigned bx_, h_, b_;
    xcd_work_item(bx_, h_, b_);
    unsigned item = bx_ * blockDim.x + threadIdx.x;
    const bool live = item < (unsigned)(cg * wtiles);
    if (!live) item = (unsigned)(cg * wtiles) - 1u;   // keep the wave whole: dead lanes recompute the last item, never store
    const int c0 = (int)(item % (unsigned)cg) * V;
    const int w0 = (int)(item / (unsigned)cg) * TW;
    const int h = (int)h_;
    const int b = (int)b_;

    const unsigned rowb = (unsigned)W * (unsigned)C * (unsigned)sizeof(T);   // bytes per image row
    const unsigned colb = (unsigned)C * (unsigned)sizeof(T);                // bytes per pixel
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (long)b * H * W * C, (unsigned)H * rowb);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wt, (unsigned)(KS * KS) * (unsigned)C * 4u);

    // byte offset of (row h - R, column w0 - R + i, channel c0); out of range where the column is outside
    unsigned voff[NI];
    const int base = ((h - R) * W + (w0 - R)) * C + c0;   // may be negative: only used where the column is valid
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int wx = w0 - R + i;
        voff[i] = (wx >= 0 && wx < W) ? (unsigned)((base + i * C) * (int)sizeof(T)) : kOutOfRange;
    }
    const unsigned woff = (unsigned)c0 * 4u;

    dw_v2f acc[TW][2];
    {
        float bs[V];
        load_pack<float, V>(bt + c0, bs);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            acc[t][0] = dw_v2f{bs[0], bs[1]};
            acc[t][1] = dw_v2f{bs[2], bs[3]};
        }
    }
    // One row of taps at a time, the NEXT row's pixels already in flight (2 static register slots, the dy loop
    // runs in pairs and is NOT unrolled further: fully unrolled, hipcc hoists all KS*KS weight loads and
    // KS*(TW+KS-1) row loads to the top -- 438 VGPRs at 7x7, spilling into AGPRs).
    static_assert(KS % 2 == 1, "odd stencils");
    DwRaw<T> raw[2][NI];
    auto issue = [&](int dy, DwRaw<T> (&r)[NI]) {
        const int hy = h + dy - R;
        const bool rowok = hy >= 0 && hy < H;
        // (h - R) * W may be negative, i.e. voff wrapped: adding dy * rowb brings a valid row back into range in
        // 32-bit arithmetic, so the sum is formed in the VECTOR offset (range-checked), not in soffset
#pragma unroll
        for (int i = 0; i < NI; ++i)
            r[i] = dw_load4<T>(rx, (rowok && voff[i] != kOutOfRange) ? voff[i] + (unsigned)dy * rowb : kOutOfRange);
    };
    auto compute = [&](int dy, const DwRaw<T> (&r)[NI]) {
        float xin[NI][V];
#pragma unroll
        for (int i = 0; i < NI; ++i) dw_cvt4<T>(r[i], xin[i]);
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
            const dw_v4f v = __builtin_bit_cast(
                dw_v4f, __builtin_amdgcn_raw_buffer_load_b128(
                            rw, woff, (unsigned)(flip ? KS * KS - 1 - (dy * KS + dx) : dy * KS + dx) * (unsigned)C * 4u, 0));
            const dw_v2f w01 = {v.x, v.y}, w23 = {v.z, v.w};
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                acc[t][0] = dw_v2f{xin[t + dx][0], xin[t + dx][1]} * w01 + acc[t][0];
                acc[t][1] = dw_v2f{xin[t + dx][2], xin[t + dx][3]} * w23 + acc[t][1];
            }
        }
    };
    issue(0, raw[0]);
#pragma unroll 1
    for (int dy = 0; dy + 1 < KS; dy += 2) {
        issue(dy + 1, raw[1]);
        compute(dy, raw[0]);
        issue(dy + 2, raw[0]);   // dy + 2 <= KS - 1 for every trip (KS odd)
        compute(dy + 1, raw[1]);
    }
    compute(KS - 1, raw[0]);
    if (!live) return;
    T *yb = y + (long)b * H * W * C + c0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (w0 + t < W) {
            float p[V] = {acc[t][0].x, acc[t][0].y, acc[t][1].x, acc[t][1].y};
            if (ypre) {   // training (block-uniform): the pre-activation leaves too, the activation is taken of it AS STORED
                store_pack<T, V>(ypre + (long)b * H * W * C + c0 + ((long)h * W + w0 + t) * C, p);
#pragma unroll
                for (int v = 0; v < V; ++v) p[v] = Cvt<T>::to_f(Cvt<T>::from_f(p[v]));
            }
            float o[V] = {apply_act(p[0], act), apply_act(p[1], act), apply_act(p[2], act), apply_act(p[3], act)};
            store_pack<T, V>(yb + ((long)h * W + w0 + t) * C, o);
        }
    }
}

// ---- 7x7, marching down the rows (r04) ----
// The one-row-per-thread kernel above issues 2260 vector instructions per wave of which 784 are its FMAs (counters,
// scripts/dev/pmc_stream.sh): every output row converts its 7 x 14 input pixels again and re-reads the taps.  Here a lane owns
// TWO channels and 4 output columns and walks a band of TH output rows from top to bottom: each input row is loaded and
// converted ONCE (10 pixels) and feeds the seven output rows it touches, whose accumulators sit in a rotating window of
// 7 x 4 register pairs (static indices: the row loop is unrolled by 7); all 49 tap pairs stay in registers.  Per input row:
// 10 loads, 20 conversions, 196 packed FMAs.  Same products in the same order as the one-row kernel (bias first, then dy, dx
// ascending): bit-identical results.
template <typename T> struct DwRaw2 { unsigned v; };
template <> struct DwRaw2<float> { dw_v2u v; };

template <typename T>
__device__ __forceinline__ DwRaw2<T> dw_load2_raw(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    DwRaw2<T> o;
    if constexpr (sizeof(T) == 4) o.v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    else o.v = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    return o;
}

template <typename T>
__device__ __forceinline__ dw_v2f dw_cvt2(const DwRaw2<T> &raw)
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(dw_v2f, raw.v);
    } else {
        const Pack<T, 2> pk = __builtin_bit_cast(Pack<T, 2>, raw.v);
        return dw_v2f{Cvt<T>::to_f(pk.v[0]), Cvt<T>::to_f(pk.v[1])};
    }
}

template <typename T>
__device__ __forceinline__ void dw_store2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, dw_v2f v)
{
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(dw_v2u, v), r, voff, soff, 0);
    } else {
        Pack<T, 2> pk;
        pk.v[0] = Cvt<T>::from_f(v.x);
        pk.v[1] = Cvt<T>::from_f(v.y);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), r, voff, soff, 0);
    }
}

constexpr int kDwmTW = 4;    // output columns per lane
constexpr int kDwmD = 4;     // input rows in flight ahead of the row being consumed (forward; LDS-DMA ring of kDwmD + 1 rows per wave)
constexpr int kDwgD = 3;     // the same for the weight gradient (x and gy rows: ring of kDwgD + 1)
typedef __attribute__((address_space(3))) void dw_lds_void;
struct alignas(16) DwQuad { unsigned v[4]; };      // element type of the LDS rings below
template <int N> __device__ __forceinline__ void dw_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DUAL: the pre-activation map leaves too (training forward): 8 instead of 4 stores per finished row, which the counted
// waits below have to know.  The input rows of a wave travel global -> LDS by LDS-DMA (16 bytes per lane: three instructions move
// the 10 pixels x 128 channels of a row), kDwmD rows ahead of the row being consumed, into a ring of the wave's own -- no barriers,
// only s_waitcnt vmcnt(N) with N counted by hand: loads, LDS-DMA and stores retire in issue order, every phase issues exactly 3
// DMA (rows outside the image through a zero-length descriptor: zeros, no traffic) and, from phase 6 on, exactly 4 (8) stores.
// (Register-staged with one row in flight the kernel waited for memory once per row: 87 us at batch 8, 96x96, C = 512; with
// one 4-byte LDS-DMA per pixel 79 us -- an LDS-DMA instruction costs its wave ~100 issue cycles whatever it moves.)
template <typename T, bool DUAL>
__global__ __launch_bounds__(256) void dwconv7_march_kernel(const T *__restrict__ x, const float *__restrict__ wt,
                                                           const float *__restrict__ bt, T *__restrict__ y, int B, int H,
                                                           int W, int C, int act, T *__restrict__ ypre, int flip, int TH,
                                                           int nbands, int nct, int ncg)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (a 16-byte LDS-DMA builtin in a kernel template: the host pass would drop the launch stub, see ss2d_scan_dma_kernel)
    constexpr int PX = sizeof(T) == 4 ? 2 : 1;      // dwords per lane and pixel
    constexpr int KS = 7, R = 3, TW = kDwmTW, NI = TW + KS - 1, D = PX == 2 ? 2 : kDwmD, NS = D + 1, S = DUAL ? 2 * TW : TW;
    // a pixel of the wave's channel tile = 128 channels = PXB bytes; one 16-byte LDS-DMA instruction moves 1 KB = 4 (2) pixels
    // (lane -> pixel lane / 16 (8), 16-byte piece lane % 16 (8)); a ring slot = the 10 pixels of a row, padded to whole KB
    constexpr int PXB = 128 * (int)sizeof(T), NDMA = (NI * PXB + 1023) / 1024, SLOTW = NDMA * 256;
    static_assert((D - 1) * (NDMA + S) <= 63 && D <= 4, "vmcnt is a 6-bit field; the ramp of the store count below covers D <= 4");
    __shared__ DwQuad ring4[4 * NS * SLOTW / 4];      // (16-byte elements: the LDS-DMA destinations are 16-byte aligned)
    // work item of a WAVE = (channel tile of 128, column group of 4, image); a band of TH rows of it per workgroup row.
    // Linear block id -> (XCD, slot): an XCD takes whole items with all their bands (band fastest), so the 6 halo rows two
    // neighbouring bands share are read through one L2.
    const unsigned lin = blockIdx.x, nblk = gridDim.x;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = lin & 7, slot8 = lin >> 3;
    const unsigned t_ = xcd * q8 + (xcd < r8 ? xcd : r8) + slot8;     // position in the (item block, band) list, band fastest
    const unsigned band = t_ % (unsigned)nbands, iblk = t_ / (unsigned)nbands;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned item = iblk * 4u + (unsigned)wave;
    if (item >= (unsigned)(nct * ncg * B)) return;
    const int ct = (int)(item % (unsigned)nct);
    const int cgp = (int)((item / (unsigned)nct) % (unsigned)ncg);
    const int b = (int)(item / (unsigned)(nct * ncg));
    const int c0 = (ct * 64 + lane) * 2;
    const bool cok = c0 + 2 <= C;
    const int w0 = cgp * TW;
    const int h0 = (int)band * TH, hend = h0 + TH < H ? h0 + TH : H;

    const unsigned colb = (unsigned)C * (unsigned)sizeof(T), rowb = (unsigned)W * colb;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + (long)b * H * W * C, (unsigned)H * rowb);
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(DUAL ? ypre + (long)b * H * W * C : y, DUAL ? (unsigned)H * rowb : 0u);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wt, (unsigned)(KS * KS) * (unsigned)C * 4u);

    unsigned voff[NI];      // byte offset of (column w0 - R + i, channel c0) inside a row; out of range where it does not exist
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int wx = w0 - R + i;
        voff[i] = (cok && wx >= 0 && wx < W) ? (unsigned)(wx * C + c0) * (unsigned)sizeof(T) : kOutOfRange;
    }
    unsigned *const myring = reinterpret_cast<unsigned *>(ring4) + wave * (NS * SLOTW);
    unsigned vd[NDMA];      // my 16-byte piece of LDS-DMA instruction k: byte offset inside an image row, or out of range
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        const int e = k * 1024 + lane * 16, i = e / PXB, ch = ct * 128 + (e % PXB) / (int)sizeof(T), wx = w0 - R + i;
        vd[k] = (i < NI && wx >= 0 && wx < W && ch + 16 / (int)sizeof(T) <= C) ? (unsigned)(wx * C + ch) * (unsigned)sizeof(T)
                                                                               : kOutOfRange;
    }
    auto issue = [&](int r, int sl) {      // row r of the image -> ring slot sl (always NDMA LDS-DMA instructions)
        const bool ok = r >= 0 && r < H;      // (a row outside the image: a zero-length descriptor, every lane out of range -> zeros)
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(x + (long)b * H * W * C, ok ? (unsigned)H * rowb : 0u);
        const unsigned so = ok ? (unsigned)r * rowb : 0u;
#pragma unroll
        for (int k = 0; k < NDMA; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (dw_lds_void *)(myring + sl * SLOTW + k * 256), 16, vd[k], so, 0, 0);
    };
    dw_v2f wv[KS * KS];
    {
        const unsigned wo = cok ? (unsigned)c0 * 4u : kOutOfRange;
#pragma unroll
        for (int i = 0; i < KS * KS; ++i)
            wv[i] = __builtin_bit_cast(dw_v2f, __builtin_amdgcn_raw_buffer_load_b64(
                                                   rw, wo, (unsigned)(flip ? KS * KS - 1 - i : i) * (unsigned)C * 4u, 0));
    }
    dw_v2f bias = {0.f, 0.f};
    if (cok) bias = dw_v2f{bt[c0], bt[c0 + 1]};
#pragma unroll
    for (int d = 0; d < D; ++d) issue(h0 - R + d, d);
    dw_v2f acc[KS][TW];
#pragma unroll
    for (int sl = 0; sl < KS; ++sl)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[sl][t] = bias;

    // phase q: input row r = h0 - R + q feeds output rows r + R - dy (dy = 0 .. 6), whose window slot is (q + 6 - dy) % 7;
    // after it output row r - R is complete (slot q % 7) and the slot starts over for row r + R + 1
    const int nq = (hend - h0) + KS - 1;
    int srd = 0, swr = D;      // ring slots: the row consumed now, the row issued now
    for (int qb = 0; qb < nq; qb += KS) {
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int q = qb + u;
            if (q < nq) {
                const int r = h0 - R + q;
                // newer than row q's DMA: the DMA of rows q + 1 .. q + D - 1 and the stores of phases q - D + 1 .. q - 1 (a phase
                // stores from q = 6 on)
                if (qb == 0) dw_vm_wait<(D - 1) * NDMA>();
                else if (qb == KS && u == 0) dw_vm_wait<(D - 1) * NDMA + (D - 1 < 1 ? D - 1 : 1) * S>();
                else if (qb == KS && u == 1) dw_vm_wait<(D - 1) * NDMA + (D - 1 < 2 ? D - 1 : 2) * S>();
                else if (qb == KS && u == 2) dw_vm_wait<(D - 1) * NDMA + (D - 1 < 3 ? D - 1 : 3) * S>();
                else dw_vm_wait<(D - 1) * (NDMA + S)>();
                if (r >= 0 && r < H) {
                    dw_v2f xin[NI];
                    const unsigned *src = myring + srd * SLOTW + lane * PX;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        DwRaw2<T> raw;
                        if constexpr (sizeof(T) == 4) raw.v = *reinterpret_cast<const dw_v2u *>(src + i * 2 * kWave);
                        else raw.v = src[i * kWave];
                        xin[i] = dw_cvt2<T>(raw);
                    }
#pragma unroll
                    for (int dy = 0; dy < KS; ++dy) {
                        const int o = r + R - dy;
                        if (o >= h0 && o < hend) {
                            const int sl = (u + 6 - dy) % KS;
#pragma unroll
                            for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                                for (int t = 0; t < TW; ++t) acc[sl][t] = xin[t + dx] * wv[dy * KS + dx] + acc[sl][t];
                        }
                    }
                }
                if (q >= KS - 1) {      // output row r - R = h0 + q - 6 is complete
                    const unsigned so = (unsigned)(r - R) * rowb;
                    dw_v2f p[TW];
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        p[t] = acc[u % KS][t];
                        acc[u % KS][t] = bias;
                    }
                    if constexpr (DUAL) {   // training: the pre-activation leaves too, the activation is taken of it AS STORED
#pragma unroll
                        for (int t = 0; t < TW; ++t) {
                            dw_store2<T>(rp, voff[t + R], so, p[t]);
                            p[t] = dw_v2f{Cvt<T>::to_f(Cvt<T>::from_f(p[t].x)), Cvt<T>::to_f(Cvt<T>::from_f(p[t].y))};
                        }
                    }
                    if (act == TRAMBA_ACT_GELU) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) p[t] = dw_v2f{geluf_(p[t].x), geluf_(p[t].y)};
                    } else if (act == TRAMBA_ACT_SILU) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) p[t] = dw_v2f{siluf_(p[t].x), siluf_(p[t].y)};
                    }
#pragma unroll
                    for (int t = 0; t < TW; ++t) dw_store2<T>(ry, voff[t + R], so, p[t]);
                }
                issue(q + D < nq ? r + D : -1, swr);
                srd = srd + 1 == NS ? 0 : srd + 1;
                swr = swr + 1 == NS ? 0 : swr + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA of this wave may land after its workgroup's LDS is handed on
#endif
}

// Weight / bias gradient of the depth-wise stencil (training): gw[tap][c] += sum over pixels of
// gy[b,h,w,c] * x[b,h+dy-R,w+dx-R,c], gb[c] += sum gy.  Lanes = channel pairs, a wave owns RPW image rows
// of one 128-channel tile and walks them 4 columns at a time (the forward kernel's register reuse with the
// roles swapped: the accumulators are indexed by tap), then writes its ks*ks + 1 partial sums as one row of
// part[wave-slot][ks*ks + 1][C] (last plane = bias); the caller sums the rows (atomics onto ks*ks*C addresses from
// thousands of waves serialise in L2).
template <typename T>
__device__ __forceinline__ dw_v2f dw_load2(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(dw_v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
    } else {
        const unsigned raw = __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
        const Pack<T, 2> pk = __builtin_bit_cast(Pack<T, 2>, raw);
        return dw_v2f{Cvt<T>::to_f(pk.v[0]), Cvt<T>::to_f(pk.v[1])};
    }
}

// waves per workgroup of the depth-wise weight-gradient kernel: 16 for the 3x3 stencil, 8 for the wider ones (their ks*ks
// accumulator pairs need more than the 128 registers a 1024-thread workgroup leaves a lane)
constexpr int dwg_waves(int ks) { return ks <= 3 ? 16 : 8; }

template <typename T, int KS>
__global__ __launch_bounds__(dwg_waves(KS) * kWave) void dwconv_wgrad_cl_kernel(const T *__restrict__ x,
                                                                          const T *__restrict__ gy,
                                                                          float *__restrict__ part, int H, int W, int C,
                                                                          int RPW, int CS)
{
    // A workgroup of NW (16 or 8) waves owns one image, one 128-channel tile, a band of NW * RPW rows and one of CS column
    // ranges; wave w walks rows [band + w RPW, +RPW).  The waves' accumulators are folded through LDS in wave order (fixed
    // -> reproducible) and leave as ONE partial row per workgroup: several times the waves of the 4-wave form for the same
    // number of partial rows (r02: 384 waves for a 96x96 map at batch 8 ran 179 us for the 7x7 stencil).
    constexpr int V = 2, R = KS / 2, TW = 4, NI = TW + KS - 1, kDwgWaves = dwg_waves(KS);
    __shared__ float red[KS * KS + 1][2 * kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = (blockIdx.x * kWave + lane) * V;
    const bool cok = c0 + V <= C;
    const int band = blockIdx.y / CS, cs = blockIdx.y % CS;
    const int row0 = (band * kDwgWaves + wv) * RPW;     // first image row of this wave
    const int wseg = ((W + CS - 1) / CS + TW - 1) / TW * TW;   // columns per range, whole groups of TW
    const int wlo = cs * wseg, whi = wlo + wseg < W ? wlo + wseg : W;
    const int b = blockIdx.z;
    // (waves past the last row still run: they contribute zeros)
    const unsigned colb = (unsigned)C * (unsigned)sizeof(T), rowb = (unsigned)W * colb;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (long)b * H * W * C, (unsigned)H * rowb);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(gy + (long)b * H * W * C, (unsigned)H * rowb);
    const unsigned cb = cok ? (unsigned)c0 * (unsigned)sizeof(T) : kOutOfRange;

    dw_v2f acc[KS][KS], accb = {0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) acc[dy][dx] = dw_v2f{0.f, 0.f};

    const int rend = row0 + RPW < H ? row0 + RPW : H;
    if constexpr (KS <= 3) {
        // small stencil: every tap row unrolled -- all KS * NI loads of a column group are in flight together (one memory
        // latency per group instead of one per tap row) and acc[][] is indexed statically
        for (int h = row0; h < rend; ++h) {
            for (int w0 = wlo; w0 < whi; w0 += TW) {
                dw_v2f g[TW];
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    const unsigned vo = (w0 + t < W && cok) ? (unsigned)((h * W + w0 + t) * C) * (unsigned)sizeof(T) + cb : kOutOfRange;
                    g[t] = dw_load2<T>(rg, vo);
                    accb = accb + g[t];
                }
                dw_v2f xin[KS][NI];
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    const int hy = h + dy - R;
                    const bool rowok = hy >= 0 && hy < H;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int wx = w0 - R + i;
                        const unsigned vo = (rowok && wx >= 0 && wx < W && cok)
                                                ? (unsigned)((hy * W + wx) * C) * (unsigned)sizeof(T) + cb : kOutOfRange;
                        xin[dy][i] = dw_load2<T>(rx, vo);
                    }
                }
#pragma unroll
                for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                        for (int t = 0; t < TW; ++t) acc[dy][dx] = g[t] * xin[dy][t + dx] + acc[dy][dx];
            }
        }
    } else {
        // wide stencil: the tap row is the OUTER, unrolled loop -- acc[dy][] is indexed statically (the run-time-dy form
        // updated all KS * KS accumulators under a select, KS times per column group, and paid one memory latency per tap
        // row), only KS accumulators are hot per copy, and the column loop is unrolled so that several groups' loads are in
        // flight.  gy is re-read once per tap row (from L2).
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
            for (int h = row0; h < rend; ++h) {
                const int hy = h + dy - R;
                if (hy < 0 || hy >= H) continue;   // (dy == R is never skipped: the bias sum below sees every row)
#pragma unroll 2
                for (int w0 = wlo; w0 < whi; w0 += TW) {
                    dw_v2f g[TW], xin[NI];
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        const unsigned vo = (w0 + t < W && cok) ? (unsigned)((h * W + w0 + t) * C) * (unsigned)sizeof(T) + cb : kOutOfRange;
                        g[t] = dw_load2<T>(rg, vo);
                    }
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int wx = w0 - R + i;
                        const unsigned vo = (wx >= 0 && wx < W && cok) ? (unsigned)((hy * W + wx) * C) * (unsigned)sizeof(T) + cb : kOutOfRange;
                        xin[i] = dw_load2<T>(rx, vo);
                    }
                    if (dy == R) {   // (compile-time after unrolling)
#pragma unroll
                        for (int t = 0; t < TW; ++t) accb = accb + g[t];
                    }
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                        for (int t = 0; t < TW; ++t) acc[dy][dx] = g[t] * xin[t + dx] + acc[dy][dx];
                }
            }
        }
    }
    // ordered fold of the 16 waves: wave 0 stores, waves 1..15 add in turn
    for (int q = 0; q < kDwgWaves; ++q) {
        if (wv == q) {
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    float2 *r = reinterpret_cast<float2 *>(&red[dy * KS + dx][2 * lane]);
                    const float2 o = q == 0 ? make_float2(0.f, 0.f) : *r;
                    *r = make_float2(o.x + acc[dy][dx].x, o.y + acc[dy][dx].y);
                }
            float2 *r = reinterpret_cast<float2 *>(&red[KS * KS][2 * lane]);
            const float2 o = q == 0 ? make_float2(0.f, 0.f) : *r;
            *r = make_float2(o.x + accb.x, o.y + accb.y);
        }
        __syncthreads();
    }
    // slot = (image, band, column range)
    float *p = part + ((long)b * gridDim.y + blockIdx.y) * (long)(KS * KS + 1) * C + (long)blockIdx.x * 2 * kWave;
    const int cleft = C - blockIdx.x * 2 * kWave;      // channels of this tile that exist
    for (int i = threadIdx.x; i < (KS * KS + 1) * 2 * kWave; i += kDwgWaves * kWave) {
        const int t = i / (2 * kWave), c = i % (2 * kWave);
        if (c < cleft) p[(long)t * C + c] = red[t][c];
    }
}

// Weight / bias gradient of the 7x7 stencil, marching (r04).  The tap-row-outer kernel above re-reads and re-converts gy and the
// x window once per tap row: 28 packed FMAs per 14 loads + 28 conversions + their offsets (counters: 7664 vector and 9137 scalar
// instructions per wave, vector ALU busy 21 % of the wave's cycles).  Here a lane owns two channels and 4 columns and walks a
// band of rows: an x row is loaded and converted once and meets the SEVEN gy rows it pairs with (tap row dy <-> gy row r + 3 - dy),
// which wait converted in a rotating window of 7 x 4 register pairs; the 49 tap sums are register pairs.  Per x row: 14 loads,
// 28 conversions, 196 packed FMAs.  Waves of a workgroup = the column groups of one column range; ordered fold through LDS as
// above, one partial row per (image, band, column range).
template <typename T>
__global__ __launch_bounds__(512) void dwconv7_wgrad_march_kernel(const T *__restrict__ x, const T *__restrict__ gy,
                                                                 float *__restrict__ part, int H, int W, int C, int TH,
                                                                 int nbands, int NW, int ncg)
{
#if defined(__HIP_DEVICE_COMPILE__)   // (as in dwconv7_march_kernel)
    constexpr int PX = sizeof(T) == 4 ? 2 : 1;      // dwords per lane and pixel
    constexpr int KS = 7, R = 3, TW = kDwmTW, NI = TW + KS - 1, D = PX == 2 ? 1 : kDwgD, NS = D + 1;
    constexpr int PXB = 128 * (int)sizeof(T), NDX = (NI * PXB + 1023) / 1024, NDG = (TW * PXB + 1023) / 1024, SLOTW = (NDX + NDG) * 256;
    static_assert((D - 1) * (NDX + NDG) <= 63, "vmcnt is a 6-bit field");
    // per wave a ring of NS rows (10 x pixels, then 4 gy pixels, each padded to whole KB) filled by 16-byte LDS-DMA, D rows ahead of
    // the row being consumed (see dwconv7_march_kernel; no stores inside the loop: the counted wait is (D - 1) * 4 in every phase);
    // the fold table of the epilogue lies over the rings
    __shared__ DwQuad smem4[8 * NS * SLOTW / 4];      // (16-byte elements: the LDS-DMA destinations are 16-byte aligned)
    static_assert(sizeof(smem4) >= (KS * KS + 1) * 2 * kWave * sizeof(float), "the fold table fits the rings");
    unsigned *const smem = reinterpret_cast<unsigned *>(smem4);
    float (*red)[2 * kWave] = reinterpret_cast<float (*)[2 * kWave]>(smem4);
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int band = blockIdx.y % nbands, cs = blockIdx.y / nbands;
    const int cgp = cs * NW + wv;
    const bool active = cgp < ncg;
    const int w0 = cgp * TW;
    const int h0 = band * TH, hend = h0 + TH < H ? h0 + TH : H;
    const int b = blockIdx.z;
    const unsigned colb = (unsigned)C * (unsigned)sizeof(T), rowb = (unsigned)W * colb;

    dw_v2f acc[KS][KS], accb = {0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) acc[dy][dx] = dw_v2f{0.f, 0.f};

    if (active) {
        unsigned *const myring = smem + wv * (NS * SLOTW);
        unsigned vdx[NDX], vdg[NDG];      // my 16-byte piece of each LDS-DMA instruction: byte offset inside an image row, or out of range
#pragma unroll
        for (int k = 0; k < NDX; ++k) {
            const int e = k * 1024 + lane * 16, i = e / PXB, ch = (int)blockIdx.x * 128 + (e % PXB) / (int)sizeof(T), wx = w0 - R + i;
            vdx[k] = (i < NI && wx >= 0 && wx < W && ch + 16 / (int)sizeof(T) <= C) ? (unsigned)(wx * C + ch) * (unsigned)sizeof(T)
                                                                                    : kOutOfRange;
        }
#pragma unroll
        for (int k = 0; k < NDG; ++k) {
            const int e = k * 1024 + lane * 16, i = e / PXB, ch = (int)blockIdx.x * 128 + (e % PXB) / (int)sizeof(T), wx = w0 + i;
            vdg[k] = (i < TW && wx < W && ch + 16 / (int)sizeof(T) <= C) ? (unsigned)(wx * C + ch) * (unsigned)sizeof(T) : kOutOfRange;
        }
        // phase q: x row r = h0 - R + q; the gy row that joins the window is h0 + q (slot (q + 6) % 7); tap row dy pairs the x row
        // with gy row r + R - dy (slot (q + 6 - dy) % 7)
        auto issue = [&](int q, int sl) {      // the rows of phase q -> ring slot sl (always NL * PX LDS-DMA instructions)
            const int r = h0 - R + q, gr = h0 + q;
            const bool xok = r >= 0 && r < H, gok = gr >= h0 && gr < hend;     // (outside: a zero-length descriptor -> zeros)
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (long)b * H * W * C, xok ? (unsigned)H * rowb : 0u);
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(gy + (long)b * H * W * C, gok ? (unsigned)H * rowb : 0u);
            const unsigned sx = xok ? (unsigned)r * rowb : 0u, sg = gok ? (unsigned)gr * rowb : 0u;
#pragma unroll
            for (int k = 0; k < NDX; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (dw_lds_void *)(myring + sl * SLOTW + k * 256), 16, vdx[k], sx, 0, 0);
#pragma unroll
            for (int k = 0; k < NDG; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (dw_lds_void *)(myring + sl * SLOTW + (NDX + k) * 256), 16, vdg[k], sg, 0, 0);
        };
        auto fetch = [&](const unsigned *src, int i) {      // my channel pair of pixel i of a region
            DwRaw2<T> raw;
            if constexpr (sizeof(T) == 4) raw.v = *reinterpret_cast<const dw_v2u *>(src + i * 2 * kWave);
            else raw.v = src[i * kWave];
            return dw_cvt2<T>(raw);
        };
        dw_v2f gwin[KS][TW];
        const int nq = (hend - h0) + KS - 1;
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d, d);
        int srd = 0, swr = D;
        for (int qb = 0; qb < nq; qb += KS) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const int q = qb + u;
                if (q < nq) {
                    const int r = h0 - R + q;
                    dw_vm_wait<(D - 1) * (NDX + NDG)>();
                    const unsigned *src = myring + srd * SLOTW + lane * PX;
                    if (h0 + q < hend) {
#pragma unroll
                        for (int t = 0; t < TW; ++t) {
                            gwin[(u + 6) % KS][t] = fetch(src + NDX * 256, t);
                            accb = accb + gwin[(u + 6) % KS][t];
                        }
                    }
                    if (r >= 0 && r < H) {
                        dw_v2f xin[NI];
#pragma unroll
                        for (int i = 0; i < NI; ++i) xin[i] = fetch(src, i);
#pragma unroll
                        for (int dy = 0; dy < KS; ++dy) {
                            const int o = r + R - dy;
                            if (o >= h0 && o < hend) {
                                const int sl = (u + 6 - dy) % KS;
#pragma unroll
                                for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                                    for (int t = 0; t < TW; ++t) acc[dy][dx] = gwin[sl][t] * xin[t + dx] + acc[dy][dx];
                            }
                        }
                    }
                    issue(q + D < nq ? q + D : -1000000, swr);
                    srd = srd + 1 == NS ? 0 : srd + 1;
                    swr = swr + 1 == NS ? 0 : swr + 1;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every LDS-DMA of this wave has landed before the rings are reused
    }
    __syncthreads();
    // ordered fold of the waves: wave 0 stores, the others add in turn (idle waves add zeros)
    for (int q = 0; q < NW; ++q) {
        if (wv == q) {
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    float2 *r = reinterpret_cast<float2 *>(&red[dy * KS + dx][2 * lane]);
                    const float2 o = q == 0 ? make_float2(0.f, 0.f) : *r;
                    *r = make_float2(o.x + acc[dy][dx].x, o.y + acc[dy][dx].y);
                }
            float2 *r = reinterpret_cast<float2 *>(&red[KS * KS][2 * lane]);
            const float2 o = q == 0 ? make_float2(0.f, 0.f) : *r;
            *r = make_float2(o.x + accb.x, o.y + accb.y);
        }
        __syncthreads();
    }
    // slot = (image, column range, band)
    float *p = part + ((long)b * gridDim.y + blockIdx.y) * (long)(KS * KS + 1) * C + (long)blockIdx.x * 2 * kWave;
    const int cleft = C - blockIdx.x * 2 * kWave;      // channels of this tile that exist
    for (int i = threadIdx.x; i < (KS * KS + 1) * 2 * kWave; i += NW * kWave) {
        const int t = i / (2 * kWave), c = i % (2 * kWave);
        if (c < cleft) p[(long)t * C + c] = red[t][c];
    }
#endif
}

template <typename T, int KS>
static int launch_dw(const void *x, const float *wt, const float *bt, void *y, int B, int H, int W, int C, int act,
                     hipStream_t s, void *ypre = nullptr, int flip = 0)
{
    // columns per thread: the stencil is L1/TA-bandwidth bound (every thread re-reads the ks*ks taps and ks rows
    // of TW + ks - 1 pixels), so the wide 7x7 amortises them over 8 outputs; 3x3 / 5x5 keep 4 (more threads)
    constexpr int TW = KS >= 7 ? 8 : 4;
    const int wtiles = (W + TW - 1) / TW;
    const int v = (C % 4 == 0) ? 4 : ((C % 2 == 0) ? 2 : 1);
    const long nthreads = (long)(C / v) * wtiles * H * B;
    dim3 grid((unsigned)((nthreads + 255) / 256)), block(256);
#define GO_(V_)                                                                                            \
    hipLaunchKernelGGL((dwconv_cl_kernel<T, KS, V_, TW>), grid, block, 0, s, (const T *)x, wt, bt, (T *)y, B, H, W, \
                       C, act, nthreads, (T *)ypre, flip)
    // buffer-addressed kernel: 4 channels per lane, every wave inside one image, 32-bit byte offsets
    if (v == 4 && H <= 65535 && B <= 65535 && (double)H * W * C * sizeof(T) < 2147483648.0) {
        dim3 g3((unsigned)(((long)(C / 4) * wtiles + 255) / 256), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL((dwconv4_cl_kernel<T, KS, TW>), g3, block, 0, s, (const T *)x, wt, bt, (T *)y, B, H, W, C, act,
                           (T *)ypre, flip);
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    if (v == 4) GO_(4);
    else if (v == 2) GO_(2);
    else GO_(1);
#undef GO_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

// rows per band of the marching 7x7 kernels: the workgroups of a launch should come in whole rounds of the 256 CUs, and a
// band pays for its 6 halo rows (loads + conversions, ~50 of the ~250 vector instructions of a full row)
static int dw7_band_rows(long items4, int H)
{
    const int cand[] = {6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192};
    int best = H;
    double best_cost = 1e30;
    for (int th : cand) {
        if (th > H && th != cand[0]) continue;
        const int t = th > H ? H : th;
        const long nwg = items4 * ((H + t - 1) / t);
        const double cost = (double)((nwg + 255) / 256) * (t * 196.0 + (t + 6) * 50.0 + 150.0);
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

template <typename T>
static int launch_dw7_march(const void *x, const float *wt, const float *bt, void *y, int B, int H, int W, int C, int act,
                            hipStream_t s, void *ypre, int flip)
{
    const int nct = (C / 2 + 63) / 64, ncg = (W + kDwmTW - 1) / kDwmTW;
    const long items4 = ((long)nct * ncg * B + 3) / 4;
    int th = dw7_band_rows(items4, H);
    const int forced = tramba_tune_get(TRAMBA_TUNE_DW_ROWS);
    if (forced > 0) th = forced > H ? H : forced;
    const int nbands = (H + th - 1) / th;
    if (ypre)
        hipLaunchKernelGGL((dwconv7_march_kernel<T, true>), dim3((unsigned)(items4 * nbands)), dim3(256), 0, s, (const T *)x, wt, bt,
                           (T *)y, B, H, W, C, act, (T *)ypre, flip, th, nbands, nct, ncg);
    else
        hipLaunchKernelGGL((dwconv7_march_kernel<T, false>), dim3((unsigned)(items4 * nbands)), dim3(256), 0, s, (const T *)x, wt, bt,
                           (T *)y, B, H, W, C, act, (T *)ypre, flip, th, nbands, nct, ncg);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_layernorm_cl(const void *x, const float *w, const float *b, void *y, int64_t rows,
                                   int c, float eps, int act, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && b && y, "layernorm_cl: null tensor");
    TRAMBA_CHECK(rows > 0 && c > 0, "layernorm_cl: empty shape");
    TRAMBA_CHECK(aligned16(x) && aligned16(y), "layernorm_cl: tensors must be 16-byte aligned");
    ProfScope prof(TRAMBA_PROF_LAYERNORM, (hipStream_t)stream, 2.0 * (double)rows * c * dtype_size(dtype));
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        return launch_layernorm<T>(x, w, b, y, rows, c, eps, act, 1, 1, 1, (hipStream_t)stream));
    return TRAMBA_OK;
}

static long ln_bwd_rows_per_wave(long rows, bool rows_form)
{
    // ~4096 waves on the large maps.  The partial (dgamma, dbeta) rows leave per WORKGROUP (four waves folded in LDS), and
    // the caller sums them: at one row per wave and one partial per wave (24x24 maps) writing and summing the partials moved
    // 2.7x the bytes of the backward itself.  Short rows: >= 8 rows per wave (several rows of a wave are in flight at once).
    // Long rows (one row of a wave at a time, every row a chain of three wave reductions): >= 2, i.e. 4x the waves of the
    // 8-row rule -- the 24x24 / 12x12 maps otherwise run half a wave per SIMD (43.6 us per call at 4608 x 1024).
    long rpw = rows / 4096;
    const long lo = rows_form ? 4 : 2;   // (r03: 4, not 8 -- 4608 x 512: 17.7 -> 12.8 us, scripts/bench_ln_bwd.py; the partial rows are summed in the step's batched reduction)
    return rpw < lo ? lo : (rpw > 128 ? 128 : rpw);
}

// rows of at most 64 lanes x 16 bytes: several rows per wave (layernorm_bwd_rows_kernel), one partial row per BLOCK
static bool ln_bwd_rows_form(int c, int dtype)
{
    const int vm = dtype == TRAMBA_F32 ? 4 : 8;
    return c % vm == 0 && c <= kWave * vm;
}

extern "C" int64_t tramba_layernorm_bwd_parts(int64_t rows, int c, int dtype)
{
    if (rows <= 0 || c <= 0) return 0;
    const long rpw = ln_bwd_rows_per_wave(rows, ln_bwd_rows_form(c, dtype));
    const long waves = (rows + rpw - 1) / rpw;
    return (waves + 3) / 4;
}

extern "C" int tramba_layernorm_bwd_cl(const void *x, const void *dy, const float *w, void *dx, float *part,
                                       int64_t rows, int c, float eps, int dtype, void *stream)
{
    return tramba_layernorm_bwd_res_cl(x, dy, w, dx, part, nullptr, nullptr, 0, nullptr, rows, c, eps, dtype, stream);
}

extern "C" int tramba_layernorm_bwd_res_cl(const void *x, const void *dy, const float *w, void *dx, float *part,
                                           const void *gres, const float *mask, int64_t rows_per_sample, void *dxm,
                                           int64_t rows, int c, float eps, int dtype, void *stream)
{
    return tramba_layernorm_bwd_any_cl(x, dy, w, dx, part, gres, mask, rows_per_sample, dxm, rows, c, eps, 1, 1, 1, dtype,
                                       stream);
}

extern "C" int tramba_shuffle_norm_bwd_cl(const void *x, const void *dy, const float *w, void *dx, float *part, int batch,
                                          int h, int wd, int c, int p, float eps, int dtype, void *stream)
{
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0 && p >= 1, "shuffle_norm_bwd_cl: empty shape");
    const int64_t rows = (int64_t)batch * h * wd * p * p;
    TRAMBA_CHECK(rows < 2147483647L, "shuffle_norm_bwd_cl: too many rows for the 32-bit row index of this build");
    return tramba_layernorm_bwd_any_cl(x, dy, w, dx, part, nullptr, nullptr, 0, nullptr, rows, c, eps, p, h, wd, dtype, stream);
}

extern "C" int tramba_layernorm_bwd_any_cl(const void *x, const void *dy, const float *w, void *dx, float *part,
                                           const void *gres, const float *mask, int64_t rows_per_sample, void *dxm,
                                           int64_t rows, int c, float eps, int P, int H, int W, int dtype, void *stream)
{
    TRAMBA_CHECK(x && dy && w && dx && part, "layernorm_bwd_cl: null tensor");
    TRAMBA_CHECK(!mask || rows_per_sample > 0, "layernorm_bwd_cl: rows_per_sample must be positive with a mask");
    TRAMBA_CHECK((!gres || aligned16(gres)) && (!dxm || aligned16(dxm)), "layernorm_bwd_cl: tensors must be 16-byte aligned");
    const long rps = rows_per_sample > 0 ? (long)rows_per_sample : 1;
    TRAMBA_CHECK(rows > 0 && c > 0, "layernorm_bwd_cl: empty shape");
    TRAMBA_CHECK(aligned16(x) && aligned16(dy) && aligned16(dx), "layernorm_bwd_cl: tensors must be 16-byte aligned");
    const int v = (c % 4 == 0) ? 4 : ((c % 2 == 0) ? 2 : 1);
    TRAMBA_CHECK((c + kWave * v - 1) / (kWave * v) <= kNormMaxIt, "layernorm_bwd_cl: C=%d too large", c);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_LAYERNORM, s, (3.0 + (gres ? 1.0 : 0.0) + (dxm ? 1.0 : 0.0)) * (double)rows * c * dtype_size(dtype));
    const long rpw = ln_bwd_rows_per_wave(rows, ln_bwd_rows_form(c, dtype));
    const long waves = (rows + rpw - 1) / rpw;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    // short rows: several rows per wave, 16-byte accesses; longer rows: one row per wave
    const int vm = dtype == TRAMBA_F32 ? 4 : 8;
    if (ln_bwd_rows_form(c, dtype)) {
        int lpr = 1;
        while (lpr < c / vm) lpr <<= 1;
#define GOB_(T, V_, L_)                                                                                             \
    hipLaunchKernelGGL((layernorm_bwd_rows_kernel<T, V_, L_>), grid, block, 0, s, (const T *)x, (const T *)dy, w, (T *)dx, \
                       part, (long)rows, c, eps, (int)rpw, (const T *)gres, mask, rps, (T *)dxm, P, H, W)
#define BYL_(T, V_)                    \
    switch (lpr) {                     \
    case 1: GOB_(T, V_, 1); break;     \
    case 2: GOB_(T, V_, 2); break;     \
    case 4: GOB_(T, V_, 4); break;     \
    case 8: GOB_(T, V_, 8); break;     \
    case 16: GOB_(T, V_, 16); break;   \
    case 32: GOB_(T, V_, 32); break;   \
    default: GOB_(T, V_, 64); break;   \
    }
        if (dtype == TRAMBA_F32) { BYL_(float, 4) }
        else if (dtype == TRAMBA_BF16) { BYL_(__hip_bfloat16, 8) }
        else { BYL_(__half, 8) }
#undef BYL_
#undef GOB_
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    if (dtype != TRAMBA_F32 && c % 8 == 0 && c > 512 && c <= 2048) {   // wide 16-bit rows: compile-time iteration count
        const size_t lds = (size_t)8 * c * sizeof(float);
#define WIDE_(T, N_)                                                                                                     \
    hipLaunchKernelGGL((layernorm_bwd_wide_kernel<T, N_>), grid, block, lds, s, (const T *)x, (const T *)dy, w, (T *)dx, part, \
                       (long)rows, c, eps, (int)rpw, (const T *)gres, mask, rps, (T *)dxm, P, H, W)
#define WIDE_N_(T)                         \
    if (c <= 1024) { WIDE_(T, 2); }        \
    else if (c <= 1536) { WIDE_(T, 3); }   \
    else { WIDE_(T, 4); }
        if (dtype == TRAMBA_BF16) { WIDE_N_(__hip_bfloat16) } else { WIDE_N_(__half) }
#undef WIDE_N_
#undef WIDE_
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        const size_t lds = (size_t)8 * c * sizeof(float);
        if (v == 4) hipLaunchKernelGGL((layernorm_bwd_cl_kernel<T, 4>), grid, block, lds, s, (const T *)x, (const T *)dy, w, (T *)dx, part, (long)rows, c, eps, (int)rpw, (const T *)gres, mask, rps, (T *)dxm, P, H, W);
        else if (v == 2) hipLaunchKernelGGL((layernorm_bwd_cl_kernel<T, 2>), grid, block, lds, s, (const T *)x, (const T *)dy, w, (T *)dx, part, (long)rows, c, eps, (int)rpw, (const T *)gres, mask, rps, (T *)dxm, P, H, W);
        else hipLaunchKernelGGL((layernorm_bwd_cl_kernel<T, 1>), grid, block, lds, s, (const T *)x, (const T *)dy, w, (T *)dx, part, (long)rows, c, eps, (int)rpw, (const T *)gres, mask, rps, (T *)dxm, P, H, W);
    });
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_shuffle_norm_cl(const void *x, const float *w, const float *b, void *y, int batch,
                                      int h, int wd, int c, int p, float eps, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && b && y, "shuffle_norm_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0 && p >= 1, "shuffle_norm_cl: empty shape");
    TRAMBA_CHECK(aligned16(x) && aligned16(y), "shuffle_norm_cl: tensors must be 16-byte aligned");
    const long rows = (long)batch * h * wd * p * p;
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        return launch_layernorm<T>(x, w, b, y, rows, c, eps, TRAMBA_ACT_NONE, p, h, wd, (hipStream_t)stream));
    return TRAMBA_OK;
}

extern "C" int tramba_shuffle_norm_head_cl(const void *x, const float *w, const float *b, const float *head_w,
                                           float head_b, float *y, int batch, int h, int wd, int c, int p, float eps,
                                           int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && b && head_w && y, "shuffle_norm_head_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0 && p >= 1, "shuffle_norm_head_cl: empty shape");
    TRAMBA_CHECK(aligned16(x) && aligned16(head_w), "shuffle_norm_head_cl: tensors must be 16-byte aligned");
    const long rows = (long)batch * h * wd * p * p;
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        return launch_layernorm<T>(x, w, b, nullptr, rows, c, eps, TRAMBA_ACT_NONE, p, h, wd, (hipStream_t)stream,
                                   head_w, head_b, y));
    return TRAMBA_OK;
}

// row groups a wave of the fused norm + head backward walks: ~2048 workgroups (= partial rows) for the 1.18 M rows of a batch of 8
static int norm_head_bwd_passes(long rows, int lpr)
{
    const long per_block = 4L * (kWave / lpr);
    long passes = (rows + per_block * 2048 - 1) / (per_block * 2048);
    return (int)(passes < 1 ? 1 : passes > 64 ? 64 : passes);
}

extern "C" int64_t tramba_shuffle_norm_head_bwd_parts(int batch, int h, int wd, int c, int p, int dtype)
{
    const int vm = dtype == TRAMBA_F32 ? 4 : 8;
    if (batch <= 0 || h <= 0 || wd <= 0 || c <= 0 || p < 1 || c % vm != 0 || c / vm > kWave) return 0;
    int lpr = 1;
    while (lpr < c / vm) lpr <<= 1;
    const long rows = (long)batch * h * wd * p * p;
    const long per_block = 4L * (kWave / lpr) * norm_head_bwd_passes(rows, lpr);
    return (rows + per_block - 1) / per_block;
}

extern "C" int tramba_shuffle_norm_head_bwd_cl(const void *x, const float *g, const float *ln_w, const float *head_w, void *dx,
                                               float *part, int batch, int h, int wd, int c, int p, float eps, int dtype,
                                               void *stream)
{
    TRAMBA_CHECK(x && g && ln_w && head_w && dx && part, "shuffle_norm_head_bwd_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0 && p >= 1, "shuffle_norm_head_bwd_cl: empty shape");
    TRAMBA_CHECK(aligned16(x) && aligned16(dx) && aligned16(ln_w) && aligned16(head_w), "shuffle_norm_head_bwd_cl: tensors must be 16-byte aligned");
    const long nparts = tramba_shuffle_norm_head_bwd_parts(batch, h, wd, c, p, dtype);
    TRAMBA_CHECK(nparts > 0, "shuffle_norm_head_bwd_cl: C=%d needs C %% %d == 0 and C <= %d", c, dtype == TRAMBA_F32 ? 4 : 8,
                 kWave * (dtype == TRAMBA_F32 ? 4 : 8));
    const long rows = (long)batch * h * wd * p * p;
    TRAMBA_CHECK(rows < 2147483647L && nparts < 2147483647L, "shuffle_norm_head_bwd_cl: too many rows");
    hipStream_t s = (hipStream_t)stream;
#define GOB_(T, VM, L_)                                                                                                  \
    hipLaunchKernelGGL((norm_head_bwd_rows_kernel<T, VM, L_>), dim3((unsigned)nparts), dim3(256), 0, s, (const T *)x, g, ln_w, \
                       head_w, (T *)dx, part, rows, c, eps, p, h, wd, norm_head_bwd_passes(rows, L_))
#define GOBL_(T, VM)                                                                                                     \
    {                                                                                                                    \
        int lpr = 1;                                                                                                     \
        while (lpr < c / VM) lpr <<= 1;                                                                                  \
        switch (lpr) {                                                                                                   \
        case 1: GOB_(T, VM, 1); break;                                                                                   \
        case 2: GOB_(T, VM, 2); break;                                                                                   \
        case 4: GOB_(T, VM, 4); break;                                                                                   \
        case 8: GOB_(T, VM, 8); break;                                                                                   \
        case 16: GOB_(T, VM, 16); break;                                                                                 \
        case 32: GOB_(T, VM, 32); break;                                                                                 \
        default: GOB_(T, VM, 64); break;                                                                                 \
        }                                                                                                                \
    }
    if (dtype == TRAMBA_F32) GOBL_(float, 4)
    else if (dtype == TRAMBA_F16) GOBL_(__half, 8)
    else if (dtype == TRAMBA_BF16) GOBL_(__hip_bfloat16, 8)
    else TRAMBA_CHECK(false, "bad dtype %d", dtype);
#undef GOBL_
#undef GOB_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_rowdot_cl(const void *x, const float *w, float bias, float *y, int64_t rows, int c, int dtype,
                                void *stream)
{
    TRAMBA_CHECK(x && w && y, "rowdot_cl: null tensor");
    TRAMBA_CHECK(rows > 0 && c > 0, "rowdot_cl: empty shape");
    TRAMBA_CHECK(aligned16(x) && aligned16(w), "rowdot_cl: tensors must be 16-byte aligned");
    TRAMBA_DISPATCH_DTYPE(dtype, T, return launch_rowdot<T>(x, w, bias, y, (long)rows, c, (hipStream_t)stream));
    return TRAMBA_OK;
}

extern "C" int tramba_dw_pack(const float *w, const float *bias, const float *w3, const float *b3,
                              const float *w5, const float *b5, float *wt, float *bt, int c, int ks,
                              void *stream)
{
    TRAMBA_CHECK(w && wt && bt && c > 0, "dw_pack: null tensor");
    TRAMBA_CHECK(ks == 3 || ks == 5 || ks == 7, "dw_pack: kernel size %d unsupported (3,5,7)", ks);
    const int ms = (w3 || w5) ? 1 : 0;
    TRAMBA_CHECK(!ms || (ks == 7 && w3 && b3 && w5 && b5 && bias), "dw_pack: multi-scale needs ks=7 and all six tensors");
    hipLaunchKernelGGL(dw_pack_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, bias, w3, b3, w5, b5,
                       wt, bt, c, ks, ms);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_dw_pack_multi(const float *const *w, const float *const *bias, const float *const *w3,
                                    const float *const *b3, const float *const *w5, const float *const *b5, float *const *wt,
                                    float *const *bt, const int *c, const int *ks, int count, void *stream)
{
    TRAMBA_CHECK(w && bias && w3 && b3 && w5 && b5 && wt && bt && c && ks && count > 0, "dw_pack_multi: empty input");
    for (int base = 0; base < count; base += kDwMulti) {
        DwPackArgs a;
        const int n = count - base < kDwMulti ? count - base : kDwMulti;
        int cmax = 0;
        for (int i = 0; i < kDwMulti; ++i) {
            const int j = base + (i < n ? i : 0);        // (unused slots repeat the first item; their blocks are never launched)
            TRAMBA_CHECK(w[j] && wt[j] && bt[j] && c[j] > 0, "dw_pack_multi: item %d: null tensor", j);
            TRAMBA_CHECK(ks[j] == 3 || ks[j] == 5 || ks[j] == 7, "dw_pack_multi: item %d: kernel size %d unsupported (3,5,7)", j, ks[j]);
            const int ms = (w3[j] || w5[j]) ? 1 : 0;
            TRAMBA_CHECK(!ms || (ks[j] == 7 && w3[j] && b3[j] && w5[j] && b5[j] && bias[j]),
                         "dw_pack_multi: item %d: multi-scale needs ks=7 and all six tensors", j);
            a.it[i] = DwPackItem{w[j], bias[j], w3[j], b3[j], w5[j], b5[j], wt[j], bt[j], c[j], ks[j], ms, 0};
            if (i < n && c[j] > cmax) cmax = c[j];
        }
        hipLaunchKernelGGL(dw_pack_multi_kernel, dim3((cmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, a);
        TRAMBA_LAUNCH_CHECK();
    }
    return TRAMBA_OK;
}

// rows per wave / column ranges of the depth-wise weight gradient: bands of 16 (8) waves; maps of >= 48 columns are cut in two
static int dw_wgrad_rpw(int h) { return h > 96 ? 2 : 1; }
static int dw_wgrad_cs(int wd) { return wd >= 48 ? 2 : 1; }

// the marching 7x7 weight gradient: rows per band, column ranges and waves (= column groups) per workgroup
static bool dw7_wgrad_march(int ks, int c) { return ks == 7 && c % 8 == 0 && tramba_tune_get(TRAMBA_TUNE_DW_FORM) != 1; }
static int dw7_wgrad_rows(int h)
{
    const int forced = tramba_tune_get(TRAMBA_TUNE_DW_ROWS);
    const int th = forced > 0 ? forced : (h >= 48 ? 24 : 12);      // (measured at batch 8: 96x96 81 -> 78 us, 48x48 52 -> 41, 24x24 28 (12 rows) / 38 (24))
    return th > h ? h : th;
}
static void dw7_wgrad_cols(int wd, int &cs, int &nw, int &ncg)
{
    ncg = (wd + kDwmTW - 1) / kDwmTW;
    cs = (ncg + 7) / 8;
    nw = (ncg + cs - 1) / cs;
}

extern "C" int64_t tramba_dwconv_wgrad_parts(int batch, int h, int wd, int c, int ks)
{
    if (batch <= 0 || h <= 0 || wd <= 0 || c <= 0 || ks <= 0) return 0;
    if (dw7_wgrad_march(ks, c)) {
        int cs, nw, ncg;
        dw7_wgrad_cols(wd, cs, nw, ncg);
        const int th = dw7_wgrad_rows(h);
        return (int64_t)batch * ((h + th - 1) / th) * cs;
    }
    const int rpw = dw_wgrad_rpw(h), nw = dwg_waves(ks);
    return (int64_t)batch * ((h + nw * rpw - 1) / (nw * rpw)) * dw_wgrad_cs(wd);
}

extern "C" int tramba_dwconv_wgrad_cl(const void *x, const void *gy, float *part, int batch, int h, int wd, int c,
                                      int ks, int dtype, void *stream)
{
    TRAMBA_CHECK(x && gy && part, "dwconv_wgrad_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0, "dwconv_wgrad_cl: empty shape");
    TRAMBA_CHECK(ks == 3 || ks == 5 || ks == 7, "dwconv_wgrad_cl: kernel size %d unsupported (3,5,7)", ks);
    TRAMBA_CHECK(c % 2 == 0, "dwconv_wgrad_cl: C=%d must be even", c);
    TRAMBA_CHECK(batch <= 65535 && (double)h * wd * c * 4.0 < 2147483648.0, "dwconv_wgrad_cl: shape exceeds this build's limits");
    hipStream_t s = (hipStream_t)stream;
    if (dw7_wgrad_march(ks, c)) {
        int cs7, nw7, ncg;
        dw7_wgrad_cols(wd, cs7, nw7, ncg);
        const int th = dw7_wgrad_rows(h), nbands = (h + th - 1) / th;
        ProfScope prof(TRAMBA_PROF_DW, s, 2.0 * (double)batch * h * wd * c * dtype_size(dtype));
        dim3 grid((unsigned)((c / 2 + kWave - 1) / kWave), (unsigned)(nbands * cs7), (unsigned)batch), block(nw7 * kWave);
        TRAMBA_DISPATCH_DTYPE(dtype, T,
            hipLaunchKernelGGL((dwconv7_wgrad_march_kernel<T>), grid, block, 0, s, (const T *)x, (const T *)gy, part, h, wd, c, th,
                               nbands, nw7, ncg));
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    const int rpw = dw_wgrad_rpw(h), cs = dw_wgrad_cs(wd), nw = dwg_waves(ks);
    dim3 grid((unsigned)((c / 2 + kWave - 1) / kWave), (unsigned)(((h + nw * rpw - 1) / (nw * rpw)) * cs), (unsigned)batch),
        block(nw * kWave);
    ProfScope prof(TRAMBA_PROF_DW, s, 2.0 * (double)batch * h * wd * c * dtype_size(dtype));
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (ks == 3) hipLaunchKernelGGL((dwconv_wgrad_cl_kernel<T, 3>), grid, block, 0, s, (const T *)x, (const T *)gy, part, h, wd, c, rpw, cs);
        else if (ks == 5) hipLaunchKernelGGL((dwconv_wgrad_cl_kernel<T, 5>), grid, block, 0, s, (const T *)x, (const T *)gy, part, h, wd, c, rpw, cs);
        else hipLaunchKernelGGL((dwconv_wgrad_cl_kernel<T, 7>), grid, block, 0, s, (const T *)x, (const T *)gy, part, h, wd, c, rpw, cs);
    });
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// im2col / col2im of a dense 3x3 convolution on a channels-last map (training path of the four stride-2 convs of the VMamba
// stem and downsample layers, vmamba.py:454,481,486): the GEMMs around them are the library's own, these two kernels are
// the data movement.  cols (B*Ho*Wo, CKp): row = output pixel, column (ky*3 + kx)*C + ci -- the k-major order of
// tramba_conv3x3s2_cl's weight -- zero in the padding taps and in the CKp - 9C alignment columns.
namespace tramba {   // (kernel names carry the namespace: the profile summaries tell the library's launches by it)
template <typename T, int V>
__global__ __launch_bounds__(256) void im2col3x3_cl_kernel(const T *__restrict__ x, T *__restrict__ cols, int H, int W,
                                                          int C, int Ho, int Wo, int stride, int pad, int CKp, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per V consecutive columns of a row
    if (i >= total) return;
    const int cpr = CKp / V;
    const long row = i / cpr;
    const int col = (int)(i % cpr) * V;
    const int wo = (int)(row % Wo);
    const long t = row / Wo;
    const int ho = (int)(t % Ho), b = (int)(t / Ho);
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    const int tap = col / C, ci = col - tap * C;    // (V > 1 only with C % V == 0: the V columns share a tap)
    if (tap < 9) {
        const int hi = ho * stride + tap / 3 - pad, wi = wo * stride + tap % 3 - pad;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) load_pack<T, V>(x + (((long)b * H + hi) * W + wi) * C + ci, v);
    }
    store_pack<T, V>(cols + row * CKp + col, v);
}

// gx[b,hi,wi,ci] = sum over the taps (ky,kx) and output pixels (ho,wo) with ho*stride + ky - pad = hi, wo*stride + kx - pad
// = wi of gcols[(b,ho,wo)][(ky*3 + kx)*C + ci]: a gather per input pixel (at most 4 terms at stride 2), fp32 sums, no atomics
template <typename T, int V>
__global__ __launch_bounds__(256) void col2im3x3_cl_kernel(const T *__restrict__ gcols, T *__restrict__ gx, int H, int W,
                                                          int C, int Ho, int Wo, int stride, int pad, int CKp, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per V channels of an input pixel
    if (i >= total) return;
    const int cpp = C / V;
    const long pix = i / cpp;
    const int ci = (int)(i % cpp) * V;
    const int wi = (int)(pix % W);
    const long t = pix / W;
    const int hi = (int)(t % H), b = (int)(t / H);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hn = hi + pad - ky;
        if (hn < 0 || hn % stride) continue;
        const int ho = hn / stride;
        if (ho >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int wn = wi + pad - kx;
            if (wn < 0 || wn % stride) continue;
            const int wo = wn / stride;
            if (wo >= Wo) continue;
            float v[V];
            load_pack<T, V>(gcols + (((long)b * Ho + ho) * Wo + wo) * CKp + (ky * 3 + kx) * C + ci, v);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += v[e];
        }
    }
    store_pack<T, V>(gx + pix * C + ci, acc);
}

// Backward of F.interpolate(mode="bilinear", align_corners=False) on (B*C, h, w) fp32 planes upsampled to (H, W) -- the
// deep-supervision outputs of the loss (train.py:76-85) -- as a GATHER per input pixel: torch's kernel scatters with float
// atomics, ~64 of them onto every input pixel at 8x (183 us for an 8 x 48 x 48 map).  Same source-index rule as
// upsample_bilinear2d: src = max(scale * (dst + 0.5) - 0.5, 0), i0 = floor(src), i1 = i0 + (i0 < n - 1), weights 1 - f, f.
// r03: one WAVE per input pixel (its window of destination pixels -- 34 x 34 at 16x -- spread over the lanes, then a wave
// sum), not one thread: the 24x24 map of a batch of 8 was 18 workgroups walking 1156 loads each in series (131 us).
__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gin,
                                                                   int h, int w, int H, int W, long total)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= total) return;                       // wave-uniform
    const int x = (int)(i % w);
    const long t = i / w;
    const int y = (int)(t % h);
    const long plane = t / h;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    // destination rows / columns that can read source index y / x: a window of (2 / scale + 2) around the centre
    const int Y0 = max(0, (int)((y - 1 + 0.5f) / sy - 0.5f) - 1), Y1 = min(H - 1, (int)((y + 1 + 0.5f) / sy - 0.5f) + 1);
    const int X0 = max(0, (int)((x - 1 + 0.5f) / sx - 0.5f) - 1), X1 = min(W - 1, (int)((x + 1 + 0.5f) / sx - 0.5f) + 1);
    const int nx = X1 - X0 + 1, n = (Y1 - Y0 + 1) * nx;
    const float *g = gout + plane * (long)H * W;
    float acc = 0.f;
    for (int e = lane; e < n; e += kWave) {
        const int Y = Y0 + e / nx, X = X0 + e % nx;
        const float fy = fmaxf(sy * ((float)Y + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
        const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
        const float fx = fmaxf(sx * ((float)X + 0.5f) - 0.5f, 0.f);
        const int x0 = (int)fx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
        const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
        acc = fmaf(wy * wx, g[(long)Y * W + X], acc);
    }
    acc = wave_sum(acc);                          // fixed order: the result does not depend on the launch
    if (lane == 0) gin[i] = acc;
}
}  // namespace tramba

extern "C" int tramba_upsample_bilinear_bwd(const float *gout, float *gin, int planes, int h, int w, int hout, int wout,
                                            void *stream)
{
    TRAMBA_CHECK(gout && gin, "upsample_bilinear_bwd: null tensor");
    TRAMBA_CHECK(planes > 0 && h > 0 && w > 0 && hout >= h && wout >= w, "upsample_bilinear_bwd: upsampling only");
    const long total = (long)planes * h * w;
    hipLaunchKernelGGL(upsample_bilinear_bwd_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       gout, gin, h, w, hout, wout, total);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

static int conv3_geom(int h, int wd, int stride, int pad, int &ho, int &wo)
{
    ho = (h + 2 * pad - 3) / stride + 1;
    wo = (wd + 2 * pad - 3) / stride + 1;
    return ho > 0 && wo > 0;
}

extern "C" int tramba_im2col3x3_cl(const void *x, void *cols, int batch, int h, int wd, int c, int stride, int pad,
                                   int ckp, int dtype, void *stream)
{
    TRAMBA_CHECK(x && cols, "im2col3x3_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0 && stride >= 1 && pad >= 0, "im2col3x3_cl: empty shape");
    TRAMBA_CHECK(ckp >= 9 * c, "im2col3x3_cl: CKp=%d < 9*C", ckp);
    int ho, wo;
    TRAMBA_CHECK(conv3_geom(h, wd, stride, pad, ho, wo), "im2col3x3_cl: the map is smaller than the stencil");
    const int esz = dtype == TRAMBA_F32 ? 4 : 2;
    const int vmax = 16 / esz;
    const bool vec = c % vmax == 0 && ckp % vmax == 0 && aligned16(x) && aligned16(cols);
    const long rows = (long)batch * ho * wo;
    hipStream_t s = (hipStream_t)stream;
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        constexpr int VM = 16 / (int)sizeof(T);
        if (vec) {
            const long total = rows * (ckp / VM);
            hipLaunchKernelGGL((im2col3x3_cl_kernel<T, VM>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T *)x,
                               (T *)cols, h, wd, c, ho, wo, stride, pad, ckp, total);
        } else {
            const long total = rows * ckp;
            hipLaunchKernelGGL((im2col3x3_cl_kernel<T, 1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T *)x,
                               (T *)cols, h, wd, c, ho, wo, stride, pad, ckp, total);
        }
    });
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_col2im3x3_cl(const void *gcols, void *gx, int batch, int h, int wd, int c, int stride, int pad,
                                   int ckp, int dtype, void *stream)
{
    TRAMBA_CHECK(gcols && gx, "col2im3x3_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0 && stride >= 1 && pad >= 0, "col2im3x3_cl: empty shape");
    TRAMBA_CHECK(ckp >= 9 * c, "col2im3x3_cl: CKp=%d < 9*C", ckp);
    int ho, wo;
    TRAMBA_CHECK(conv3_geom(h, wd, stride, pad, ho, wo), "col2im3x3_cl: the map is smaller than the stencil");
    const int esz = dtype == TRAMBA_F32 ? 4 : 2;
    const int vmax = 16 / esz;
    const bool vec = c % vmax == 0 && ckp % vmax == 0 && aligned16(gcols) && aligned16(gx);
    const long pixels = (long)batch * h * wd;
    hipStream_t s = (hipStream_t)stream;
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        constexpr int VM = 16 / (int)sizeof(T);
        if (vec) {
            const long total = pixels * (c / VM);
            hipLaunchKernelGGL((col2im3x3_cl_kernel<T, VM>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                               (const T *)gcols, (T *)gx, h, wd, c, ho, wo, stride, pad, ckp, total);
        } else {
            const long total = pixels * c;
            hipLaunchKernelGGL((col2im3x3_cl_kernel<T, 1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                               (const T *)gcols, (T *)gx, h, wd, c, ho, wo, stride, pad, ckp, total);
        }
    });
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_dwconv_cl(const void *x, const float *wt, const float *bt, void *y, int batch, int h,
                                int wd, int c, int ks, int act, int dtype, void *stream)
{
    return tramba_dwconv_dual_cl(x, wt, bt, nullptr, y, batch, h, wd, c, ks, act, 0, dtype, stream);
}

extern "C" int tramba_dwconv_dual_cl(const void *x, const float *wt, const float *bt, void *y_pre, void *y, int batch,
                                     int h, int wd, int c, int ks, int act, int flip_taps, int dtype, void *stream)
{
    TRAMBA_CHECK(x && wt && bt && y, "dwconv_cl: null tensor");
    TRAMBA_CHECK(!y_pre || aligned16(y_pre), "dwconv_cl: tensors must be 16-byte aligned");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0, "dwconv_cl: empty shape");
    TRAMBA_CHECK(ks == 3 || ks == 5 || ks == 7, "dwconv_cl: kernel size %d unsupported (3,5,7)", ks);
    TRAMBA_CHECK(aligned16(x) && aligned16(y) && aligned16(wt) && aligned16(bt), "dwconv_cl: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(TRAMBA_PROF_DW, s, (y_pre ? 3.0 : 2.0) * (double)batch * h * wd * c * dtype_size(dtype));
    // 7x7: the marching kernel (a lane = 2 channels x 4 columns walking a band of rows) wherever its 32-bit offsets reach
    const bool march = ks == 7 && c % 8 == 0 && tramba_tune_get(TRAMBA_TUNE_DW_FORM) != 1 &&
                       (double)h * wd * c * 4.0 < 2147483648.0 && (double)batch * wd * c < 2147483648.0;
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (march) return launch_dw7_march<T>(x, wt, bt, y, batch, h, wd, c, act, s, y_pre, flip_taps);
        if (ks == 3) return launch_dw<T, 3>(x, wt, bt, y, batch, h, wd, c, act, s, y_pre, flip_taps);
        if (ks == 5) return launch_dw<T, 5>(x, wt, bt, y, batch, h, wd, c, act, s, y_pre, flip_taps);
        return launch_dw<T, 7>(x, wt, bt, y, batch, h, wd, c, act, s, y_pre, flip_taps);
    });
    return TRAMBA_OK;
}
