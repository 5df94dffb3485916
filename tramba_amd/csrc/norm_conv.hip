// Channels-last normalisation and depth-wise stencil kernels (all HBM/L2-bound).
//
//   layernorm_cl      LayerNorm2d (Models/modules.py:22-27) without the two permute copies:
//                     in (B, L, C) the normalised axis is already contiguous.
//   shuffle_norm_cl   rearrange 'b (p1 p2 c) h w -> b c (h p1) (w p2)' + LayerNorm2d of
//                     PatchExpand / FinalPatchExpand_X4 / FreqExpand2D (modules.py:209-218,
//                     240-249, 687-696): in channels-last the shuffle only re-addresses whole
//                     C-vectors, so it is fused into the norm's store.
//   dwconv_cl         SS2D's depth-wise 3x3 + SiLU (vmamba.py:283-285).
//   dwms_cl           DWMSMlp's h + dw3(h) + dw5(h) + dw7(h) -> GELU (vmamba.py:624-625): the
//                     four linear terms are ONE 7x7 depth-wise stencil whose taps are summed
//                     in registers (identity + 3x3 + 5x5 + 7x7), so h is read once.
// One wave normalises one row; lanes map to channels with 2..16-byte accesses.
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
        const int pp = (int)(row % (P * P));
        const long pix = row / (P * P);
        const int wi = (int)(pix % W);
        const long bh = pix / W;  // b*H + h
        const int hi = (int)(bh % H);
        const long b = bh / H;
        orow = (b * (long)(H * P) + (long)hi * P + pp / P) * (long)(W * P) + (long)wi * P + pp % P;
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

template <typename T>
static int launch_layernorm(const void *x, const float *w, const float *b, void *y, long rows, int c,
                            float eps, int act, int P, int H, int W, hipStream_t s)
{
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

// ------------------------------------------------------------------------------ depth-wise stencils
// thread = V channels x TW consecutive output columns of one row.  MS = multi-scale (dwms).
template <typename T, int KS, int V, int TW, bool MS>
__global__ __launch_bounds__(256) void dwconv_cl_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ bias,
                                                       const float *__restrict__ w3, const float *__restrict__ b3,
                                                       const float *__restrict__ w5, const float *__restrict__ b5,
                                                       T *__restrict__ y, int B, int H, int W, int C, int act,
                                                       long nthreads)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nthreads) return;
    const int cg = C / V, wt = (W + TW - 1) / TW;
    const int c0 = (int)(tid % cg) * V;
    long rest = tid / cg;
    const int w0 = (int)(rest % wt) * TW;
    rest /= wt;
    const int h = (int)(rest % H);
    const int b = (int)(rest / H);
    constexpr int R = KS / 2;

    float wr[KS * KS][V];
    float bs[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int c = c0 + v;
        bs[v] = bias ? bias[c] : 0.f;
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) wr[t][v] = w[(long)c * KS * KS + t];
        if (MS) {  // fold identity, 3x3 and 5x5 into the 7x7 taps
            bs[v] += b3[c] + b5[c];
#pragma unroll
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) wr[(dy + 1) * KS + dx + 1][v] += w5[(long)c * 25 + dy * 5 + dx];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) wr[(dy + 2) * KS + dx + 2][v] += w3[(long)c * 9 + dy * 3 + dx];
            wr[R * KS + R][v] += 1.f;
        }
    }

    float acc[TW][V];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[t][v] = bs[v];

    const T *xb = x + (long)b * H * W * C + c0;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
        const int hy = h + dy - R;
        if (hy < 0 || hy >= H) continue;
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
                for (int v = 0; v < V; ++v) acc[t][v] = fmaf(wr[dy * KS + dx][v], xin[t + dx][v], acc[t][v]);
    }
    T *yb = y + (long)b * H * W * C + c0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (w0 + t < W) {
            float o[V];
#pragma unroll
            for (int v = 0; v < V; ++v) o[v] = apply_act(acc[t][v], act);
            store_pack<T, V>(yb + ((long)h * W + w0 + t) * C, o);
        }
    }
}

template <typename T, int KS, bool MS>
static int launch_dw(const void *x, const float *w, const float *bias, const float *w3, const float *b3,
                     const float *w5, const float *b5, void *y, int B, int H, int W, int C, int act,
                     hipStream_t s)
{
    constexpr int TW = 4;
    const int cg_v2 = C / 2;
    const bool v2 = (C % 2 == 0);
    const int wt = (W + TW - 1) / TW;
    const long nthreads = (long)(v2 ? cg_v2 : C) * wt * H * B;
    dim3 grid((unsigned)((nthreads + 255) / 256)), block(256);
    if (v2)
        hipLaunchKernelGGL((dwconv_cl_kernel<T, KS, 2, TW, MS>), grid, block, 0, s, (const T *)x, w, bias, w3,
                           b3, w5, b5, (T *)y, B, H, W, C, act, nthreads);
    else
        hipLaunchKernelGGL((dwconv_cl_kernel<T, KS, 1, TW, MS>), grid, block, 0, s, (const T *)x, w, bias, w3,
                           b3, w5, b5, (T *)y, B, H, W, C, act, nthreads);
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
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        return launch_layernorm<T>(x, w, b, y, rows, c, eps, act, 1, 1, 1, (hipStream_t)stream));
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

extern "C" int tramba_dwconv_cl(const void *x, const float *w, const float *bias, void *y, int batch, int h,
                                int wd, int c, int ks, int act, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w && y, "dwconv_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0, "dwconv_cl: empty shape");
    TRAMBA_CHECK(ks == 3 || ks == 5 || ks == 7, "dwconv_cl: kernel size %d unsupported (3,5,7)", ks);
    hipStream_t s = (hipStream_t)stream;
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (ks == 3) return launch_dw<T, 3, false>(x, w, bias, nullptr, nullptr, nullptr, nullptr, y, batch, h, wd, c, act, s);
        if (ks == 5) return launch_dw<T, 5, false>(x, w, bias, nullptr, nullptr, nullptr, nullptr, y, batch, h, wd, c, act, s);
        return launch_dw<T, 7, false>(x, w, bias, nullptr, nullptr, nullptr, nullptr, y, batch, h, wd, c, act, s);
    });
    return TRAMBA_OK;
}

extern "C" int tramba_dwms_cl(const void *x, const float *w3, const float *b3, const float *w5,
                              const float *b5, const float *w7, const float *b7, void *y, int batch, int h,
                              int wd, int c, int dtype, void *stream)
{
    TRAMBA_CHECK(x && w3 && b3 && w5 && b5 && w7 && b7 && y, "dwms_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0 && c > 0, "dwms_cl: empty shape");
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        return launch_dw<T, 7, true>(x, w7, b7, w3, b3, w5, b5, y, batch, h, wd, c, TRAMBA_ACT_GELU,
                                     (hipStream_t)stream));
    return TRAMBA_OK;
}
