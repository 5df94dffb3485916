// Fused element kernels of the TRAINING path (round 3): what used to be chains of small framework launches between the
// library's GEMM / scan / stencil kernels of a step -- residual adds, stochastic-depth multiplies, activation forwards and
// their gradients, gathers, zero fills, casts -- folded into the neighbouring passes.  All HBM-bound row kernels.
//
//   add_layernorm      x' = x + y * mask[sample]  (residual + stochastic depth, vmamba.py:384-396)  and  n = LayerNorm(x'),
//                      optionally also act(n): ONE pass where autograd issued addcmul -> layer_norm -> gelu.
//   ss2d_bwd_prep      after the scan backward: the dt-rank rows of x_dbl gathered into sequence order (the operand of the
//                      dt_projs_weight gradient) and the per-channel-tile partial sums of dB / dC added up in a fixed order
//                      (the scan backward writes partials instead of fp32 atomics: reproducible).
//   ss2d_bwd_assemble  the x_dbl-row gradients from sequence order back to spatial order through the inverse table (a gather-
//                      sum: deterministic also for the many-to-one Helix lines, no zero fill, no index_add_), cast to the
//                      activation dtype the x_proj gradient GEMMs read.
//   dw_unpack_grad     gradient of the folded 7x7 multi-scale stencil (tramba_dw_pack) handed back to the 3x3 / 5x5 / 7x7
//                      parameters in their own (C, 1, ks, ks) layouts; single stencils: tap-major -> (C, 1, ks, ks).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "norm.h"

namespace tramba {

// ------------------------------------------------------------------------------------------------ add + LayerNorm
// rows form: LPR lanes per row (C <= 64 * V), 64 / LPR rows per wave.
template <typename T, int V, int LPR>
__global__ __launch_bounds__(256) void add_ln_rows_kernel(const T *__restrict__ x, const T *__restrict__ yadd,
                                                         const float *__restrict__ mask, long rps,
                                                         const float *__restrict__ w, const float *__restrict__ bvec,
                                                         T *__restrict__ xsum, T *__restrict__ n, T *__restrict__ nact,
                                                         long rows, int C, float eps, int act)
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
    if (rok && cok) {
        load_pack<T, V>(x + row * C + c0, v);
        if (yadd) {
            float yv[V];
            load_pack<T, V>(yadd + row * C + c0, yv);
            const float m = mask ? mask[row / rps] : 1.f;
            // the sum is what the next layers (and the backward's recomputed statistics) read: round it to T first
#pragma unroll
            for (int i = 0; i < V; ++i) v[i] = Cvt<T>::to_f(Cvt<T>::from_f(fmaf(yv[i], m, v[i])));
            store_pack<T, V>(xsum + row * C + c0, v);
        }
    }
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
    float wv[V], bv[V], o[V];
    load_pack<float, V>(w + c0, wv);
    load_pack<float, V>(bvec + c0, bv);
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = (v[i] - mean) * rstd * wv[i] + bv[i];
    store_pack<T, V>(n + row * C + c0, o);
    if (nact) {
        // the activation of the value AS STORED (the consumer's backward differentiates at the stored pre-activation)
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = apply_act(Cvt<T>::to_f(Cvt<T>::from_f(o[i])), act);
        store_pack<T, V>(nact + row * C + c0, o);
    }
}

// wave form: one wave per row, NIT = ceil(C / (64 V)) iterations.
template <typename T, int V>
__global__ __launch_bounds__(256) void add_ln_wave_kernel(const T *__restrict__ x, const T *__restrict__ yadd,
                                                         const float *__restrict__ mask, long rps,
                                                         const float *__restrict__ w, const float *__restrict__ bvec,
                                                         T *__restrict__ xsum, T *__restrict__ n, T *__restrict__ nact,
                                                         long rows, int C, float eps, int act)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= rows) return;
    const int nit = (C + kWave * V - 1) / (kWave * V);
    float acc[kNormMaxIt][V];
    const float m = (yadd && mask) ? mask[row / rps] : 1.f;
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it) {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[it][v] = 0.f;
        if (it < nit) {
            const int c0 = (it * kWave + lane) * V;
            if (c0 + V <= C) {
                load_pack<T, V>(x + row * C + c0, acc[it]);
                if (yadd) {
                    float yv[V];
                    load_pack<T, V>(yadd + row * C + c0, yv);
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[it][v] = Cvt<T>::to_f(Cvt<T>::from_f(fmaf(yv[v], m, acc[it][v])));
                    store_pack<T, V>(xsum + row * C + c0, acc[it]);
                }
            }
        }
    }
    float mean, rstd;
    wave_layernorm<V>(acc, nit, C, lane, eps, mean, rstd);
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it) {
        if (it < nit) {
            const int c0 = (it * kWave + lane) * V;
            if (c0 + V <= C) {
                float o[V];
#pragma unroll
                for (int v = 0; v < V; ++v) o[v] = (acc[it][v] - mean) * rstd * w[c0 + v] + bvec[c0 + v];
                store_pack<T, V>(n + row * C + c0, o);
                if (nact) {
#pragma unroll
                    for (int v = 0; v < V; ++v) o[v] = apply_act(Cvt<T>::to_f(Cvt<T>::from_f(o[v])), act);
                    store_pack<T, V>(nact + row * C + c0, o);
                }
            }
        }
    }
}

template <typename T>
static int launch_add_ln(const void *x, const void *y, const float *mask, long rps, const float *w, const float *b,
                         void *xsum, void *n, void *nact, long rows, int c, float eps, int act, hipStream_t s)
{
    constexpr int VM = sizeof(T) == 2 ? 8 : 4;
    if (c % VM == 0 && c / VM <= kWave && aligned16(w) && aligned16(b)) {
        const int need = c / VM;
        int lpr = 1;
        while (lpr < need) lpr <<= 1;
        const long waves = (rows + (kWave / lpr) - 1) / (kWave / lpr);
        dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define GOR_(L_)                                                                                                        \
    hipLaunchKernelGGL((add_ln_rows_kernel<T, VM, L_>), grid, block, 0, s, (const T *)x, (const T *)y, mask, rps, w, b, \
                       (T *)xsum, (T *)n, (T *)nact, rows, c, eps, act)
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
    const int v = norm_vec(c, VM);
    if ((c + kWave * v - 1) / (kWave * v) > kNormMaxIt) {
        set_error("add_layernorm: C=%d too large", c);
        return TRAMBA_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define GO_(V_)                                                                                                       \
    hipLaunchKernelGGL((add_ln_wave_kernel<T, V_>), grid, block, 0, s, (const T *)x, (const T *)y, mask, rps, w, b,    \
                       (T *)xsum, (T *)n, (T *)nact, rows, c, eps, act)
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

// ------------------------------------------------------------------------------------------------ SS2D backward glue
// thread = one (b, k, i) sequence position; R8 = padded rank count, RG = R8 + 4.
//   ranks  (B, K, L, R8) T     = xdbl[b, table[k][i], k*RG + 0 .. R8)          (the dt-rank rows in sequence order)
//   g_seq  (B, K, L, RG) f32   columns R8 / R8 + 1 = sum over the CT channel tiles of the scan backward's dB / dC partials
//                              (bpart / cpart: (B, K, CT, L) f32), columns R8 + 2, R8 + 3 = 0; the rank columns are written
//                              by tramba_rows_gemm_cl afterwards
template <typename T>
__global__ __launch_bounds__(256) void ss2d_bwd_prep_kernel(const float *__restrict__ xdbl, const int32_t *__restrict__ table,
                                                           const float *__restrict__ bpart, const float *__restrict__ cpart,
                                                           T *__restrict__ ranks, float *__restrict__ gseq, int L, int K,
                                                           int R8, int CT)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int k = blockIdx.y, b = blockIdx.z;
    const int RG = R8 + 4, PC = K * RG;
    const int p = table[(long)k * L + i];
    const float *src = xdbl + ((long)b * L + p) * PC + (long)k * RG;
    T *dst = ranks + (((long)b * K + k) * L + i) * R8;
    for (int j = 0; j < R8; j += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(src + j);
        const float o[4] = {v.x, v.y, v.z, v.w};
        store_pack<T, 4>(dst + j, o);
    }
    const float *bp = bpart + (((long)b * K + k) * CT) * L + i;
    const float *cp = cpart + (((long)b * K + k) * CT) * L + i;
    float sb = 0.f, sc = 0.f;
    for (int t = 0; t < CT; ++t) {   // fixed order: reproducible
        sb += bp[(long)t * L];
        sc += cp[(long)t * L];
    }
    *reinterpret_cast<float4 *>(gseq + (((long)b * K + k) * L + i) * RG + R8) = make_float4(sb, sc, 0.f, 0.f);
}

// block = 4 waves, wave = one pixel; lane j < RG walks the pixel's CSR entries (ascending entry = ascending direction)
// and keeps the running sum of ITS column for the current direction; out (B, L, K*RG) T.
template <typename T>
__global__ __launch_bounds__(256) void ss2d_bwd_assemble_kernel(const float *__restrict__ gseq,
                                                               const int32_t *__restrict__ inv_ptr,
                                                               const int32_t *__restrict__ inv_idx, T *__restrict__ out,
                                                               long npix, int L, int K, int R, int R8)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wid >= npix) return;
    const int b = (int)((unsigned long)wid / (unsigned)L), p = (int)((unsigned long)wid % (unsigned)L);
    const int RG = R8 + 4;
    const int e0 = __builtin_amdgcn_readfirstlane(inv_ptr[p]), e1 = __builtin_amdgcn_readfirstlane(inv_ptr[p + 1]);
    const float *gb = gseq + (long)b * K * L * RG;
    T *orow = out + ((long)b * L + p) * (long)(K * RG);
    // a lane owns column `lane` (and lane + 64 when RG > 64) of every direction group; columns R .. R8 - 1 (rank padding)
    // and R8 + 2, R8 + 3 leave as zeros.  The entry list is fetched 64 per load, the rows 8 per batch (all in flight
    // before any is added); entries ascend, i.e. they are grouped by direction: a group's sums leave when the next begins.
    const int l1 = lane + kWave;
    const bool col0 = lane < RG, col1 = l1 < RG;
    const bool live0 = lane < R || lane == R8 || lane == R8 + 1;
    const bool live1 = l1 < R || l1 == R8 || l1 == R8 + 1;
    float a0 = 0.f, a1 = 0.f;
    int k = 0;
    auto flush = [&]() {
        if (col0) orow[k * RG + lane] = Cvt<T>::from_f(live0 ? a0 : 0.f);
        if (col1) orow[k * RG + l1] = Cvt<T>::from_f(live1 ? a1 : 0.f);
        a0 = a1 = 0.f;
        ++k;
    };
    const int n = e1 - e0;
    for (int base = 0; base < n; base += kWave) {
        const int cn = n - base < kWave ? n - base : kWave;
        const int ent = inv_idx[e0 + base + (lane < cn ? lane : cn - 1)];
        for (int j0 = 0; j0 < cn; j0 += 8) {
            float v0[8], v1[8];
            int ek[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                ek[jj] = __builtin_amdgcn_readlane(ent, j0 + jj < cn ? j0 + jj : cn - 1);
                const float *row = gb + (long)ek[jj] * RG;    // entry = k * L + i
                v0[jj] = col0 ? row[lane] : 0.f;
                v1[jj] = col1 ? row[l1] : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                if (j0 + jj < cn) {   // wave-uniform
                    while (ek[jj] >= (k + 1) * L) flush();
                    a0 += v0[jj];
                    a1 += v1[jj];
                }
            }
        }
    }
    while (k < K) flush();
}

// ------------------------------------------------------------------------------------------------ depth-wise gradients
// gwt (ks*ks + 1, C) f32 = tap-major weight gradient + bias row (tramba_dwconv_wgrad_cl + tramba_slab_sum).
// single stencil:  g7 (C, ks*ks) = transpose of the taps;  gb7 (C) = bias row
// multi-scale (ks = 7, g5 / g3 given): the folded stencil's gradient restricted to each parameter's support
__device__ __forceinline__ void dw_unpack_body(const float *__restrict__ gwt, float *__restrict__ g7, float *__restrict__ g5,
                                               float *__restrict__ g3, float *__restrict__ gb, int nb, int C, int ks, int c)
{
    if (c >= C) return;
    for (int dy = 0; dy < ks; ++dy)
        for (int dx = 0; dx < ks; ++dx) {
            const float v = gwt[(long)(dy * ks + dx) * C + c];
            g7[(long)c * ks * ks + dy * ks + dx] = v;
            if (g5) {
                const int y5 = dy - 1, x5 = dx - 1, y3 = dy - 2, x3 = dx - 2;
                if (y5 >= 0 && y5 < 5 && x5 >= 0 && x5 < 5) g5[(long)c * 25 + y5 * 5 + x5] = v;
                if (y3 >= 0 && y3 < 3 && x3 >= 0 && x3 < 3) g3[(long)c * 9 + y3 * 3 + x3] = v;
            }
        }
    if (gb)   // nb copies of the bias gradient: the folded stencil's bias is the SUM of the three parameters' biases
        for (int q = 0; q < nb; ++q) gb[(long)q * C + c] = gwt[(long)ks * ks * C + c];
}

__global__ __launch_bounds__(256) void dw_unpack_grad_kernel(const float *__restrict__ gwt, float *__restrict__ g7,
                                                            float *__restrict__ g5, float *__restrict__ g3,
                                                            float *__restrict__ gb, int nb, int C, int ks)
{
    dw_unpack_body(gwt, g7, g5, g3, gb, nb, C, ks, blockIdx.x * blockDim.x + threadIdx.x);
}

// the unpacks of a whole backward pass in one launch (the gradients are leaves: nothing reads them before the optimizer);
// items by value, blockIdx.y = item
constexpr int kDwUnpackMulti = 64;
struct DwUnpackItem {
    const float *gwt;
    float *g7, *g5, *g3, *gb;
    int nb, c, ks, pad;
};
struct DwUnpackArgs {
    DwUnpackItem it[kDwUnpackMulti];
};
static_assert(sizeof(DwUnpackArgs) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void dw_unpack_multi_kernel(DwUnpackArgs a)
{
    const DwUnpackItem it = a.it[blockIdx.y];
    dw_unpack_body(it.gwt, it.g7, it.g5, it.g3, it.gb, it.nb, it.c, it.ks, blockIdx.x * blockDim.x + threadIdx.x);
}


// ------------------------------------------------------------------------------------------------ C -> 1 head, backward
// y[row] = <x[row, :], w> + b (tramba_rowdot_cl: the decoder's seg_layers, Trambav6.py:82,130).  One pass over x:
//   gx[row, :] = gy[row] * w            (dtype)
//   part[block][0 .. C) += gy[row] * x[row, :],  part[block][C] += gy[row]     (fp32 partial rows, summed by the caller)
// LPR lanes per row (C <= 64 V), 64 / LPR rows of a wave in flight; a wave walks rpw rows.
template <typename T, int V, int LPR>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const T *__restrict__ x, const float *__restrict__ gy,
                                                        const float *__restrict__ w, T *__restrict__ gx,
                                                        float *__restrict__ part, long rows, int C, int rpw)
{
    constexpr int RPW = kWave / LPR;
    __shared__ float red[4][LPR * V + 4];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wv;
    const long r0 = wave * rpw;
    const long rend = r0 + rpw < rows ? r0 + rpw : rows;
    const int sub = lane % LPR, slot = lane / LPR;
    const int c0 = sub * V;
    const bool cok = c0 + V <= C;
    float wv_[V], gw[V], gb = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        wv_[v] = cok ? w[c0 + v] : 0.f;
        gw[v] = 0.f;
    }
    for (long rb = r0; rb < rend; rb += RPW) {
        const long r = rb + slot;
        const bool ok = r < rend && cok;
        float xv[V];
#pragma unroll
        for (int v = 0; v < V; ++v) xv[v] = 0.f;
        float g = 0.f;
        if (ok) {
            load_pack<T, V>(x + r * C + c0, xv);
            g = gy[r];
            float o[V];
#pragma unroll
            for (int v = 0; v < V; ++v) o[v] = g * wv_[v];
            store_pack<T, V>(gx + r * C + c0, o);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) gw[v] = fmaf(g, xv[v], gw[v]);
        if (sub == 0) gb += g;
    }
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) {   // fold the row slots of the wave (lanes with equal `sub`)
#pragma unroll
        for (int v = 0; v < V; ++v) gw[v] += __shfl_xor(gw[v], o, kWave);
        gb += __shfl_xor(gb, o, kWave);
    }
    if (slot == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) red[wv][c0 + v] = gw[v];
        if (sub == 0) red[wv][LPR * V] = gb;
    }
    __syncthreads();
    float *pw = part + (long)blockIdx.x * (C + 4);
    for (int i = threadIdx.x; i < C + 4; i += 256) {
        const int j = i < C ? i : LPR * V + (i - C);
        pw[i] = i <= C ? (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]) : 0.f;
    }
}

static long rowdot_bwd_rpw(long rows)
{
    long rpw = rows / 4096;
    return rpw < 8 ? 8 : (rpw > 256 ? 256 : rpw);
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_add_layernorm_cl(const void *x, const void *y, const float *mask, int64_t rows_per_sample,
                                       const float *w, const float *b, void *xsum, void *n, void *n_act, int64_t rows,
                                       int c, float eps, int act, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && b && n, "add_layernorm_cl: null tensor");
    TRAMBA_CHECK(rows > 0 && c > 0, "add_layernorm_cl: empty shape");
    TRAMBA_CHECK(!y || xsum, "add_layernorm_cl: the sum x + y * mask needs an output tensor");
    TRAMBA_CHECK(!mask || rows_per_sample > 0, "add_layernorm_cl: rows_per_sample must be positive with a mask");
    TRAMBA_CHECK(aligned16(x) && aligned16(n) && (!y || (aligned16(y) && aligned16(xsum))) && (!n_act || aligned16(n_act)),
                 "add_layernorm_cl: tensors must be 16-byte aligned");
    ProfScope prof(TRAMBA_PROF_LAYERNORM, (hipStream_t)stream,
                   (2.0 + (y ? 2.0 : 0.0) + (n_act ? 1.0 : 0.0)) * (double)rows * c * dtype_size(dtype));
    TRAMBA_DISPATCH_DTYPE(dtype, T, return launch_add_ln<T>(x, y, mask, rows_per_sample > 0 ? rows_per_sample : 1, w, b, xsum,
                                                            n, n_act, rows, c, eps, act, (hipStream_t)stream));
    return TRAMBA_OK;
}

extern "C" int tramba_ss2d_bwd_prep_cl(const float *xdbl, const int32_t *table, const float *bpart, const float *cpart,
                                       void *ranks, float *gseq, int batch, int l, int k, int r, int ctiles, int dtype,
                                       void *stream)
{
    TRAMBA_CHECK(xdbl && table && bpart && cpart && ranks && gseq, "ss2d_bwd_prep_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && l > 0 && k > 0 && r > 0 && ctiles > 0 && batch <= 65535 && k <= 65535,
                 "ss2d_bwd_prep_cl: bad shape");
    TRAMBA_CHECK(aligned16(xdbl) && aligned16(ranks) && aligned16(gseq), "ss2d_bwd_prep_cl: 16-byte alignment");
    const int r8 = (r + 7) & ~7;
    dim3 grid((unsigned)((l + 255) / 256), (unsigned)k, (unsigned)batch), block(256);
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        hipLaunchKernelGGL((ss2d_bwd_prep_kernel<T>), grid, block, 0, (hipStream_t)stream, xdbl, table, bpart, cpart,
                           (T *)ranks, gseq, l, k, r8, ctiles));
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_ss2d_bwd_assemble_cl(const float *gseq, const int32_t *inv_ptr, const int32_t *inv_idx, void *out,
                                           int batch, int l, int k, int r, int dtype, void *stream)
{
    TRAMBA_CHECK(gseq && inv_ptr && inv_idx && out, "ss2d_bwd_assemble_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && l > 0 && k > 0 && r > 0, "ss2d_bwd_assemble_cl: bad shape");
    const int r8 = (r + 7) & ~7;
    TRAMBA_CHECK(r8 + 4 <= 2 * kWave, "ss2d_bwd_assemble_cl: dt_rank %d too large", r);
    const long npix = (long)batch * l;
    TRAMBA_CHECK(npix < 2147483647L, "ss2d_bwd_assemble_cl: too many pixels");
    dim3 grid((unsigned)((npix + 3) / 4)), block(256);
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        hipLaunchKernelGGL((ss2d_bwd_assemble_kernel<T>), grid, block, 0, (hipStream_t)stream, gseq, inv_ptr, inv_idx,
                           (T *)out, npix, l, k, r, r8));
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_dw_unpack_grad(const float *gwt, float *g7, float *g5, float *g3, float *gb, int nb, int c, int ks,
                                     void *stream)
{
    TRAMBA_CHECK(gwt && g7 && c > 0, "dw_unpack_grad: null tensor");
    TRAMBA_CHECK(ks == 3 || ks == 5 || ks == 7, "dw_unpack_grad: kernel size %d unsupported (3,5,7)", ks);
    TRAMBA_CHECK((g5 == nullptr) == (g3 == nullptr) && (!g5 || ks == 7), "dw_unpack_grad: multi-scale needs ks = 7, g5 and g3");
    hipLaunchKernelGGL(dw_unpack_grad_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gwt, g7, g5, g3, gb,
                       nb, c, ks);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_dw_unpack_grad_multi(const float *const *gwt, float *const *g7, float *const *g5, float *const *g3,
                                           float *const *gb, const int *nb, const int *c, const int *ks, int count, void *stream)
{
    TRAMBA_CHECK(gwt && g7 && g5 && g3 && gb && nb && c && ks && count > 0, "dw_unpack_grad_multi: empty input");
    for (int base = 0; base < count; base += kDwUnpackMulti) {
        DwUnpackArgs a;
        const int n = count - base < kDwUnpackMulti ? count - base : kDwUnpackMulti;
        int cmax = 0;
        for (int i = 0; i < kDwUnpackMulti; ++i) {
            const int j = base + (i < n ? i : 0);
            TRAMBA_CHECK(gwt[j] && g7[j] && c[j] > 0, "dw_unpack_grad_multi: item %d: null tensor", j);
            TRAMBA_CHECK(ks[j] == 3 || ks[j] == 5 || ks[j] == 7, "dw_unpack_grad_multi: item %d: kernel size %d unsupported", j, ks[j]);
            TRAMBA_CHECK((g5[j] == nullptr) == (g3[j] == nullptr) && (!g5[j] || ks[j] == 7),
                         "dw_unpack_grad_multi: item %d: multi-scale needs ks = 7, g5 and g3", j);
            a.it[i] = DwUnpackItem{gwt[j], g7[j], g5[j], g3[j], gb[j], nb[j], c[j], ks[j], 0};
            if (i < n && c[j] > cmax) cmax = c[j];
        }
        hipLaunchKernelGGL(dw_unpack_multi_kernel, dim3((cmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, a);
        TRAMBA_LAUNCH_CHECK();
    }
    return TRAMBA_OK;
}

extern "C" int64_t tramba_rowdot_bwd_parts(int64_t rows, int c, int dtype)
{
    const int vm = dtype == TRAMBA_F32 ? 4 : 8;
    if (rows <= 0 || c <= 0 || c % vm != 0 || c > kWave * vm) return 0;     // 0: shape not served by this kernel
    const long rpw = rowdot_bwd_rpw(rows);
    return ((rows + rpw - 1) / rpw + 3) / 4;
}

extern "C" int tramba_rowdot_bwd_cl(const void *x, const float *gy, const float *w, void *gx, float *part, int64_t rows,
                                    int c, int dtype, void *stream)
{
    TRAMBA_CHECK(x && gy && w && gx && part, "rowdot_bwd_cl: null tensor");
    TRAMBA_CHECK(tramba_rowdot_bwd_parts(rows, c, dtype) > 0, "rowdot_bwd_cl: C=%d is not served (rows of at most 64 x 16 bytes)", c);
    TRAMBA_CHECK(aligned16(x) && aligned16(gx) && aligned16(w), "rowdot_bwd_cl: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const long rpw = rowdot_bwd_rpw(rows);
    const long waves = (rows + rpw - 1) / rpw;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    const int vm = dtype == TRAMBA_F32 ? 4 : 8;
    int lpr = 1;
    while (lpr < c / vm) lpr <<= 1;
#define GOB_(T, V_, L_)                                                                                                   \
    hipLaunchKernelGGL((rowdot_bwd_kernel<T, V_, L_>), grid, block, 0, s, (const T *)x, gy, w, (T *)gx, part, (long)rows, c, \
                       (int)rpw)
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
