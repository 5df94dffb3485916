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
// ss2d_scan_cl:   grid (D/64, K, B); a workgroup of W waves owns 64 channels of one
//   direction for the whole sequence.  Per super-chunk of W*LC positions every wave
//   (A) gathers its LC rows, evaluates dt = softplus(<dt_w[d,:], dts> + bias) on the VALU
//   with the R-vector in SGPRs, a = exp(dt*A), b = dt*B*u, and reduces its chunk to a
//   (decay, state) pair; (B) after ONE barrier folds the preceding waves' pairs (LDS) into
//   its carry-in; (C) replays its LC steps from registers and streams y rows out.
// ss2d_merge_norm_cl: one wave per pixel sums the rows listed by the inverse table
//   (deterministic, atomic-free even for the many-to-one Helix lines), applies out_norm
//   (LayerNorm over D, two-pass fp32) and the following GELU, writes the activation dtype.
#include "common.h"
#include "norm.h"

namespace tramba {

constexpr int kLC = 16;      // positions per wave per super-chunk
constexpr int kMaxW = 8;     // waves per workgroup

template <typename T, typename TY, int RT>
__global__ __launch_bounds__(kMaxW * kWave) void ss2d_scan_cl_kernel(
    const T *__restrict__ x, const float *__restrict__ xdbl, const int32_t *__restrict__ table,
    const float *__restrict__ dt_w, const float *__restrict__ dt_bias, const float *__restrict__ Aneg,
    const float *__restrict__ Ds, TY *__restrict__ ys, int L, int D, int K, int R, int W)
{
    __shared__ float agg[2][kMaxW][2][kWave];

    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = blockIdx.y, b = blockIdx.z;
    const int d = blockIdx.x * kWave + lane;
    const bool dok = d < D;
    const int dc = dok ? d : D - 1;
    const int PC = K * (R + 2);

    float w[RT > 0 ? RT : 1];
    const float *wrow = dt_w + ((long)k * D + dc) * R;
    if (RT > 0) {
#pragma unroll
        for (int r = 0; r < RT; ++r) w[r] = wrow[r];
    }
    const float bias = dt_bias[(long)k * D + dc];
    const float An = Aneg[(long)k * D + dc];
    const float Dk = Ds[(long)k * D + dc];

    const T *xb = x + (long)b * L * D + dc;
    const float *pb = xdbl + (long)b * L * PC + (long)k * (R + 2);
    const int32_t *tk = table + (long)k * L;
    TY *yb = ys + ((long)b * K + k) * L * D + dc;

    float carry = 0.f;
    const int span = W * kLC;
    const int nsuper = (L + span - 1) / span;
    for (int s = 0; s < nsuper; ++s) {
        const int l0 = s * span + wv * kLC;
        float a[kLC], bb[kLC], cc[kLC], du[kLC];
        float pa = 1.f, ph = 0.f;
#pragma unroll
        for (int j = 0; j < kLC; ++j) {
            const int l = l0 + j;
            if (l < L) {  // wave-uniform
                const int p = tk[l];
                const float u = Cvt<T>::to_f(xb[(long)p * D]);
                const float *prow = pb + (long)p * PC;
                float dtv = bias;
                if (RT > 0) {
#pragma unroll
                    for (int r = 0; r < RT; ++r) dtv = fmaf(w[r], prow[r], dtv);
                } else {
                    for (int r = 0; r < R; ++r) dtv = fmaf(wrow[r], prow[r], dtv);
                }
                const float dt = softplus20(dtv);
                a[j] = __expf(dt * An);
                bb[j] = dt * prow[R] * u;
                cc[j] = prow[R + 1];
                du[j] = Dk * u;
            } else {
                a[j] = 1.f; bb[j] = 0.f; cc[j] = 0.f; du[j] = 0.f;
            }
            ph = fmaf(a[j], ph, bb[j]);
            pa *= a[j];
        }
        const int buf = s & 1;
        agg[buf][wv][0][lane] = pa;
        agg[buf][wv][1][lane] = ph;
        __syncthreads();
        // fold all W chunks: pick my carry-in on the way, end with the next carry
        float h = carry, hin = carry;
        for (int q = 0; q < W; ++q) {
            if (q == wv) hin = h;
            h = fmaf(agg[buf][q][0][lane], h, agg[buf][q][1][lane]);
        }
        carry = h;
        h = hin;
#pragma unroll
        for (int j = 0; j < kLC; ++j) {
            h = fmaf(a[j], h, bb[j]);
            const int l = l0 + j;
            if (l < L && dok) yb[(long)l * D] = Cvt<TY>::from_f(fmaf(cc[j], h, du[j]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <typename TY, typename T, int V>
__global__ __launch_bounds__(256) void ss2d_merge_norm_cl_kernel(
    const TY *__restrict__ ys, const int32_t *__restrict__ inv_ptr, const int32_t *__restrict__ inv_idx,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, T *__restrict__ y, long npix, int L,
    int D, int K, float eps, int act)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long pix = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (pix >= npix) return;
    const int b = (int)(pix / L), p = (int)(pix % L);
    const int nit = (D + kWave * V - 1) / (kWave * V);
    float acc[kNormMaxIt][V];
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[it][v] = 0.f;

    const TY *yb = ys + (long)b * K * L * D;
    const int e1 = inv_ptr[p + 1];
    for (int e = inv_ptr[p]; e < e1; ++e) {
        const TY *row = yb + (long)inv_idx[e] * D;  // entry = k*L + l indexes (K*L, D) rows
#pragma unroll
        for (int it = 0; it < kNormMaxIt; ++it) {
            if (it < nit) {
                const int c0 = (it * kWave + lane) * V;
                if (c0 + V <= D) {
                    float t[V];
                    load_pack<TY, V>(row + c0, t);
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[it][v] += t[v];
                }
            }
        }
    }
    float mean, rstd;
    wave_layernorm<V>(acc, nit, D, lane, eps, mean, rstd);
    T *orow = y + pix * D;
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it) {
        if (it < nit) {
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
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_ss2d_scan_cl(const void *x, const float *xdbl, const int32_t *table,
                                   const float *dt_w, const float *dt_bias, const float *A,
                                   const float *Ds, void *ys, int batch, int l, int d, int k, int r,
                                   int dtype, int ys_dtype, void *stream)
{
    TRAMBA_CHECK(x && xdbl && table && dt_w && dt_bias && A && Ds && ys, "ss2d_scan_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && l > 0 && d > 0 && k > 0 && r > 0, "ss2d_scan_cl: empty shape");
    TRAMBA_CHECK(batch <= 65535 && k <= 65535, "ss2d_scan_cl: B or K exceeds grid limits");
    TRAMBA_CHECK(ys_dtype == TRAMBA_F32 || ys_dtype == dtype, "ss2d_scan_cl: ys must be f32 or the input dtype");
    hipStream_t s = (hipStream_t)stream;
    int W = (l + kLC - 1) / kLC;
    if (W > kMaxW) W = kMaxW;
    dim3 grid((d + kWave - 1) / kWave, k, batch), block(W * kWave);
    // algorithmic bytes of this kernel: x read once, the low-rank x_proj rows, ys written
    ProfScope prof(TRAMBA_PROF_SCAN_FUSED, s,
                   (double)batch * l * d * dtype_size(dtype) + (double)batch * l * k * (r + 2) * 4.0 +
                       (double)batch * k * l * (double)d * dtype_size(ys_dtype));
#define GO_(T, TY, RT_)                                                                              \
    hipLaunchKernelGGL((ss2d_scan_cl_kernel<T, TY, RT_>), grid, block, 0, s, (const T *)x, xdbl, table, \
                       dt_w, dt_bias, A, Ds, (TY *)ys, l, d, k, r, W)
#define BY_R_(T, TY)                                   \
    switch (r) {                                       \
    case 1: GO_(T, TY, 1); break;                      \
    case 2: GO_(T, TY, 2); break;                      \
    case 4: GO_(T, TY, 4); break;                      \
    case 8: GO_(T, TY, 8); break;                      \
    case 16: GO_(T, TY, 16); break;                    \
    case 32: GO_(T, TY, 32); break;                    \
    case 64: GO_(T, TY, 64); break;                    \
    default: GO_(T, TY, 0); break;                     \
    }
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (ys_dtype == TRAMBA_F32) { BY_R_(T, float) } else { BY_R_(T, T) }
    });
#undef BY_R_
#undef GO_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_ss2d_merge_norm_cl(const void *ys, const int32_t *inv_ptr, const int32_t *inv_idx,
                                         const float *ln_w, const float *ln_b, void *y, int batch, int l,
                                         int d, int k, float eps, int act, int ys_dtype, int dtype,
                                         void *stream)
{
    TRAMBA_CHECK(ys && inv_ptr && inv_idx && ln_w && ln_b && y, "ss2d_merge_norm_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && l > 0 && d > 0 && k > 0, "ss2d_merge_norm_cl: empty shape");
    TRAMBA_CHECK(ys_dtype == TRAMBA_F32 || ys_dtype == dtype, "ss2d_merge_norm_cl: ys must be f32 or dtype");
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)batch * l;
    // vector width: largest of 4/2/1 dividing D such that the row fits kNormMaxIt iterations
    int v = (d % 4 == 0) ? 4 : ((d % 2 == 0) ? 2 : 1);
    TRAMBA_CHECK((d + kWave * v - 1) / (kWave * v) <= kNormMaxIt, "ss2d_merge_norm_cl: D=%d too large", d);
    TRAMBA_CHECK(aligned16(ys) && aligned16(y), "ss2d_merge_norm_cl: tensors must be 16-byte aligned");
    dim3 grid((unsigned)((npix + 3) / 4)), block(256);
#define GO_(TY, T, V_)                                                                                  \
    hipLaunchKernelGGL((ss2d_merge_norm_cl_kernel<TY, T, V_>), grid, block, 0, s, (const TY *)ys, inv_ptr, \
                       inv_idx, ln_w, ln_b, (T *)y, npix, l, d, k, eps, act)
#define BY_V_(TY, T)                   \
    if (v == 4) GO_(TY, T, 4);         \
    else if (v == 2) GO_(TY, T, 2);    \
    else GO_(TY, T, 1);
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (ys_dtype == TRAMBA_F32) { BY_V_(float, T) } else { BY_V_(T, T) }
    });
#undef BY_V_
#undef GO_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
