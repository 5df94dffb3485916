// The two ends of the reference's optimisation step that are not network layers (train.py:74-89):
//
//   * the deep-supervision loss (train.py:76-85; utils/loss.py:6-11): every output bilinearly resized to the label, then
//     binary_cross_entropy_with_logits + iou_loss, summed over the outputs.  Eagerly that is ~35 framework launches per
//     output over 4.7 MB maps (resize, sigmoid, products, four reductions, their backward); here it is one pass that leaves
//     three sums per image (the resized logit is recomputed from the small map, never stored), one finishing block, and
//     one gradient pass per output that lands directly on the output's own resolution (the resize backward is a gather
//     per source pixel, as in tramba_upsample_bilinear_bwd);
//   * Adam (train.py:266-280, torch.optim.Adam's arithmetic): one read and one write of p / exp_avg / exp_avg_sq and one
//     read of the gradient per step, the tensors of a launch described BY VALUE in the kernel arguments (nothing is
//     copied to the device, hipGraph-capture safe: the gradients of a captured step live at other addresses than the
//     warm-up's).
//
// Both are HBM-streaming kernels: the loss moves 8 B per label pixel and output, Adam 28 B per parameter.
#include "common.h"

#include <algorithm>
#include <vector>

namespace tramba {

// ------------------------------------------------------------------------------------------------- loss
// upsample_bilinear2d (align_corners=False): src = max(scale * (dst + 0.5) - 0.5, 0), i0 = floor(src),
// i1 = i0 + (i0 < n - 1), weights 1 - f, f.
struct Tap {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Tap tap(int dst, float scale, int n)
{
    const float f = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    Tap t;
    t.i0 = (int)f;
    t.i1 = t.i0 + (t.i0 < n - 1 ? 1 : 0);
    t.l1 = f - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}
__device__ __forceinline__ float resized(const float *__restrict__ z, int w, const Tap &ty, const Tap &tx)
{
    const float *r0 = z + (long)ty.i0 * w, *r1 = z + (long)ty.i1 * w;
    return ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1]);
}
// sigmoid(z) and log(1 + exp(-|z|)) from one exponential
__device__ __forceinline__ void sig_terms(float z, float &p, float &softplus_tail)
{
    const float e = expf(-fabsf(z)), r = 1.f / (1.f + e);
    p = z >= 0.f ? r : e * r;
    softplus_tail = log1pf(e);
}

constexpr int kLossThreads = 256;

// part[plane][blockIdx.x][3] = { sum bce, sum p*y, sum (p + y) } over this block's pixels of the plane
template <bool SAME>
__global__ __launch_bounds__(kLossThreads) void sod_loss_sums_kernel(const float *__restrict__ z,
                                                                    const float *__restrict__ label,
                                                                    float *__restrict__ part, int h, int w, int H, int W)
{
    __shared__ float red[kLossThreads / kWave][3];
    const int plane = blockIdx.y, nblk = gridDim.x;
    const long npix = (long)H * W;
    const float *zp = z + (long)plane * h * w, *yp = label + plane * npix;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (long i = (long)blockIdx.x * kLossThreads + threadIdx.x; i < npix; i += (long)nblk * kLossThreads) {
        float zz;
        if constexpr (SAME) {
            zz = zp[i];
        } else {
            const int Y = (int)(i / W), X = (int)(i - (long)Y * W);
            zz = resized(zp, w, tap(Y, sy, h), tap(X, sx, w));
        }
        const float y = yp[i];
        float p, tail;
        sig_terms(zz, p, tail);
        s0 += fmaxf(zz, 0.f) - zz * y + tail;
        s1 = fmaf(p, y, s1);
        s2 += p + y;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[wv][0] = s0;
        red[wv][1] = s1;
        red[wv][2] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int q = 0; q < kLossThreads / kWave; ++q) s += red[q][threadIdx.x];
        part[((long)plane * nblk + blockIdx.x) * 3 + threadIdx.x] = s;
    }
}

constexpr int kLossOutputs = 8, kLossPlanes = 512;
struct LossFinishArgs {
    const float *part[kLossOutputs];   // (planes, nblk[i], 3)
    float *coef[kLossOutputs];         // (planes, 4): { a, cI, cU, 0 }
    int nblk[kLossOutputs];
    float weight[kLossOutputs];
    int nout, planes;
    double npix;
    float *loss;
};
// One block.  Per output o and plane q:  I = sum p*y, U = sum (p + y), D = U - I + 1,
//   loss = sum_o w_o [ sum_q bce_q / (planes * npix) + mean_q (1 - (I + 1) / D) ]
// and the coefficients of d loss / d logit = a (p - y) + p (1 - p) (cI y + cU):
//   a = w_o / (planes * npix),  cI = -(w_o / planes) (U + 2) / D^2,  cU = (w_o / planes) (I + 1) / D^2.
// A wave per (output, plane) adds the partial sums (lanes stride over them, then a shuffle tree: a fixed order).
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(1024) void sod_loss_finish_kernel(LossFinishArgs a)
{
    __shared__ double term[kLossOutputs][kLossPlanes];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int item = wv; item < a.nout * a.planes; item += nwave) {
        const int o = item / a.planes, q = item - o * a.planes;
        const float *p = a.part[o] + (long)q * a.nblk[o] * 3;
        double bce = 0.0, I = 0.0, U = 0.0;
        for (int s = lane; s < a.nblk[o]; s += kWave) {
            bce += (double)p[3 * s];
            I += (double)p[3 * s + 1];
            U += (double)p[3 * s + 2];
        }
        bce = wave_sum_f64(bce);
        I = wave_sum_f64(I);
        U = wave_sum_f64(U);
        if (lane == 0) {
            const double w = (double)a.weight[o], D = U - I + 1.0;
            term[o][q] = w * (bce / ((double)a.planes * a.npix) + (1.0 - (I + 1.0) / D) / (double)a.planes);
            float *c = a.coef[o] + 4 * (long)q;
            c[0] = (float)(w / ((double)a.planes * a.npix));
            c[1] = (float)(-(w / (double)a.planes) * (U + 2.0) / (D * D));
            c[2] = (float)((w / (double)a.planes) * (I + 1.0) / (D * D));
            c[3] = 0.f;
        }
    }
    __syncthreads();
    if (wv == 0) {
        double s = 0.0;
        for (int item = lane; item < a.nout * a.planes; item += kWave) s += term[item / a.planes][item % a.planes];
        s = wave_sum_f64(s);
        if (lane == 0) *a.loss = (float)s;
    }
}

__device__ __forceinline__ float loss_grad(float zz, float y, float ca, float ci, float cu)
{
    float p, tail;
    sig_terms(zz, p, tail);
    return ca * (p - y) + p * (1.f - p) * fmaf(ci, y, cu);
}

// same resolution: one thread per pixel
__global__ __launch_bounds__(256) void sod_loss_grad_same_kernel(const float *__restrict__ z, const float *__restrict__ label,
                                                                const float *__restrict__ coef,
                                                                const float *__restrict__ gscale, float *__restrict__ gz,
                                                                long npix)
{
    const int plane = blockIdx.y;
    const float gs = gscale ? *gscale : 1.f;
    const float ca = gs * coef[4 * plane], ci = gs * coef[4 * plane + 1], cu = gs * coef[4 * plane + 2];
    const long base = (long)plane * npix;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x)
        gz[base + i] = loss_grad(z[base + i], label[base + i], ca, ci, cu);
}

// resized output: the adjoint of the bilinear resize is separable, so the gradient at label resolution is formed ONCE per
// label pixel (a row band in LDS) and folded along X, then along Y.  (A gather per source pixel over its 2-D window -- the
// form of tramba_upsample_bilinear_bwd -- evaluates every label pixel ~4 times: 49 us for the 24x24 map of a batch of 8.)
// The label columns / rows that read source index i: a window of (2 / scale + 2) around its centre.
__device__ __forceinline__ void window(int i, float scale, int N, int &lo, int &hi)
{
    lo = max(0, (int)((i - 1 + 0.5f) / scale - 0.5f) - 1);
    hi = min(N - 1, (int)((i + 1 + 0.5f) / scale - 0.5f) + 1);
}
__device__ __forceinline__ float tap_weight(const Tap &t, int i)
{
    return (t.i0 == i ? t.l0 : 0.f) + (t.i1 == i ? t.l1 : 0.f);
}
constexpr int kGradRows = 4;
// rows[plane][Y][x] = sum over X of wx(X -> x) g(Y, X)
__global__ __launch_bounds__(256) void sod_loss_grad_rows_kernel(const float *__restrict__ z, const float *__restrict__ label,
                                                                const float *__restrict__ coef,
                                                                const float *__restrict__ gscale, float *__restrict__ rows,
                                                                int h, int w, int H, int W)
{
    extern __shared__ float band[];               // [kGradRows][W]
    const int plane = blockIdx.y, yb = blockIdx.x * kGradRows;
    const float gs = gscale ? *gscale : 1.f;
    const float ca = gs * coef[4 * plane], ci = gs * coef[4 * plane + 1], cu = gs * coef[4 * plane + 2];
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const float *zp = z + (long)plane * h * w, *yp = label + (long)plane * H * W;
    for (int e = threadIdx.x; e < kGradRows * W; e += blockDim.x) {
        const int r = e / W, X = e - r * W, Y = yb + r;
        if (Y < H) band[e] = loss_grad(resized(zp, w, tap(Y, sy, h), tap(X, sx, w)), yp[(long)Y * W + X], ca, ci, cu);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kGradRows * w; e += blockDim.x) {
        const int r = e / w, x = e - r * w, Y = yb + r;
        if (Y >= H) continue;
        int X0, X1;
        window(x, sx, W, X0, X1);
        float acc = 0.f;
        for (int X = X0; X <= X1; ++X) acc = fmaf(tap_weight(tap(X, sx, w), x), band[r * W + X], acc);
        rows[((long)plane * H + Y) * w + x] = acc;
    }
}
// gz[plane][y][x] = sum over Y of wy(Y -> y) rows[plane][Y][x]
__global__ __launch_bounds__(256) void sod_loss_grad_cols_kernel(const float *__restrict__ rows, float *__restrict__ gz, int h,
                                                                int w, int H, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % w);
    const long t = i / w;
    const int y = (int)(t % h);
    const long plane = t / h;
    const float sy = (float)h / (float)H;
    int Y0, Y1;
    window(y, sy, H, Y0, Y1);
    const float *r = rows + plane * (long)H * w + x;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) acc = fmaf(tap_weight(tap(Y, sy, h), y), r[(long)Y * w], acc);
    gz[i] = acc;
}

// ------------------------------------------------------------------------------------------------- Adam
constexpr int kAdamTensors = 72, kAdamChunk = 8192, kAdamThreads = 256, kBumpTensors = 448;
struct AdamTensor {
    float *p;
    const float *g;
    float *m, *v;
    const float *step;
    long n;
};
struct AdamArgs {
    AdamTensor t[kAdamTensors];
    int first[kAdamTensors + 1];   // first workgroup of each tensor (ascending); first[count] = the grid
    int count;
    double lr, beta1, beta2, eps, weight_decay;
};
struct BumpArgs {
    float *step[kBumpTensors];
    int count;
};
static_assert(sizeof(AdamArgs) <= 4096 && sizeof(BumpArgs) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(kBumpTensors) void adam_bump_kernel(BumpArgs a)
{
    if ((int)threadIdx.x < a.count) *a.step[threadIdx.x] += 1.f;
}

typedef float adam_f4 __attribute__((ext_vector_type(4)));

template <bool ALIGNED>
__device__ __forceinline__ adam_f4 adam_load(const float *p)
{
    if constexpr (ALIGNED) return *reinterpret_cast<const adam_f4 *>(p);
    adam_f4 v = {p[0], p[1], p[2], p[3]};
    return v;
}
template <bool ALIGNED>
__device__ __forceinline__ void adam_store(float *p, adam_f4 v)
{
    if constexpr (ALIGNED) {
        *reinterpret_cast<adam_f4 *>(p) = v;
    } else {
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
}

struct AdamCoef {
    float step_size, bc2_sqrt, b1w, beta2, b2w, eps, wd;
};
// torch.optim.Adam (amsgrad off, maximize off):  g += wd p;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;
// p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__device__ __forceinline__ void adam_math(float &p, float g, float &m, float &v, const AdamCoef &c)
{
    g = fmaf(c.wd, p, g);
    m = fmaf(c.b1w, g - m, m);
    v = fmaf(c.beta2, v, c.b2w * g * g);
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p -= c.step_size * m / denom;
}

template <bool ALIGNED>
__device__ __forceinline__ void adam_chunk(const AdamTensor &t, long off, const AdamCoef &c)
{
    const long left = t.n - off;
    const int here = left < kAdamChunk ? (int)left : kAdamChunk;
    const int nvec = here >> 2;
    float *p = t.p + off, *m = t.m + off, *v = t.v + off;
    const float *g = t.g + off;
    constexpr int U = 4;
    for (int j0 = threadIdx.x; j0 < nvec; j0 += U * kAdamThreads) {
        adam_f4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * kAdamThreads;
            if (j < nvec) {
                gg[u] = adam_load<ALIGNED>(g + 4 * j);
                pp[u] = adam_load<ALIGNED>(p + 4 * j);
                mm[u] = adam_load<ALIGNED>(m + 4 * j);
                vv[u] = adam_load<ALIGNED>(v + 4 * j);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * kAdamThreads;
            if (j < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pp[u][e], me = mm[u][e], ve = vv[u][e];
                    adam_math(pe, gg[u][e], me, ve, c);
                    pp[u][e] = pe;
                    mm[u][e] = me;
                    vv[u][e] = ve;
                }
                adam_store<ALIGNED>(p + 4 * j, pp[u]);
                adam_store<ALIGNED>(m + 4 * j, mm[u]);
                adam_store<ALIGNED>(v + 4 * j, vv[u]);
            }
        }
    }
    const int j = (nvec << 2) + threadIdx.x;     // < 4 trailing elements of the tensor
    if (j < here) {
        float pe = p[j], me = m[j], ve = v[j];
        adam_math(pe, g[j], me, ve, c);
        p[j] = pe;
        m[j] = me;
        v[j] = ve;
    }
}

__global__ __launch_bounds__(kAdamThreads) void adam_kernel(AdamArgs a)
{
    __shared__ float corr[2];
    int lo = 0, hi = a.count;                     // the tensor this workgroup works on: first[lo] <= blockIdx.x < first[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const AdamTensor t = a.t[lo];
    const long off = (long)((int)blockIdx.x - a.first[lo]) * kAdamChunk;
    if (threadIdx.x == 0) {                       // the bias corrections in double, as the reference's optimizer computes them
        const double s = (double)*t.step;
        corr[0] = (float)(a.lr / (1.0 - pow(a.beta1, s)));
        corr[1] = (float)sqrt(1.0 - pow(a.beta2, s));
    }
    __syncthreads();
    AdamCoef c;
    c.step_size = corr[0];
    c.bc2_sqrt = corr[1];
    c.b1w = (float)(1.0 - a.beta1);
    c.beta2 = (float)a.beta2;
    c.b2w = (float)(1.0 - a.beta2);
    c.eps = (float)a.eps;
    c.wd = (float)a.weight_decay;
    const bool al = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                      reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    if (al) adam_chunk<true>(t, off, c); else adam_chunk<false>(t, off, c);
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_sod_loss_sums(const float *logits, const float *label, float *part, int planes, int h, int w,
                                    int hout, int wout, int nblk, void *stream)
{
    TRAMBA_CHECK(logits && label && part, "sod_loss_sums: null tensor");
    TRAMBA_CHECK(planes > 0 && planes <= 65535 && h > 0 && w > 0 && hout > 0 && wout > 0 && nblk > 0, "sod_loss_sums: bad shape");
    const dim3 grid((unsigned)nblk, (unsigned)planes);
    if (h == hout && w == wout)
        hipLaunchKernelGGL(sod_loss_sums_kernel<true>, grid, dim3(kLossThreads), 0, (hipStream_t)stream, logits, label, part,
                           h, w, hout, wout);
    else
        hipLaunchKernelGGL(sod_loss_sums_kernel<false>, grid, dim3(kLossThreads), 0, (hipStream_t)stream, logits, label, part,
                           h, w, hout, wout);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_sod_loss_finish(const float *const *parts, const int *nblk, const float *weights, float *const *coefs,
                                      int nout, int planes, int64_t npix, float *loss, void *stream)
{
    TRAMBA_CHECK(parts && nblk && coefs && loss, "sod_loss_finish: null argument");
    TRAMBA_CHECK(nout > 0 && nout <= kLossOutputs, "sod_loss_finish: 1..%d outputs (got %d)", kLossOutputs, nout);
    TRAMBA_CHECK(planes > 0 && planes <= kLossPlanes && npix > 0, "sod_loss_finish: 1..%d planes (got %d)", kLossPlanes, planes);
    LossFinishArgs a;
    for (int o = 0; o < kLossOutputs; ++o) {
        const bool on = o < nout;
        TRAMBA_CHECK(!on || (parts[o] && coefs[o] && nblk[o] > 0), "sod_loss_finish: output %d: null table", o);
        a.part[o] = on ? parts[o] : nullptr;
        a.coef[o] = on ? coefs[o] : nullptr;
        a.nblk[o] = on ? nblk[o] : 0;
        a.weight[o] = on ? (weights ? weights[o] : 1.f) : 0.f;
    }
    a.nout = nout;
    a.planes = planes;
    a.npix = (double)npix;
    a.loss = loss;
    hipLaunchKernelGGL(sod_loss_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" size_t tramba_sod_loss_grad_workspace(int planes, int h, int w, int hout, int wout)
{
    return (h == hout && w == wout) ? 0 : (size_t)planes * (size_t)hout * (size_t)w * sizeof(float);
}

extern "C" int tramba_sod_loss_grad(const float *logits, const float *label, const float *coef, const float *gscale,
                                    float *glogits, void *workspace, size_t workspace_bytes, int planes, int h, int w,
                                    int hout, int wout, void *stream)
{
    TRAMBA_CHECK(logits && label && coef && glogits, "sod_loss_grad: null tensor");
    TRAMBA_CHECK(planes > 0 && planes <= 65535 && h > 0 && w > 0, "sod_loss_grad: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (h == hout && w == wout) {
        const long npix = (long)h * w;
        const unsigned gx = (unsigned)((npix + 1023) / 1024);
        hipLaunchKernelGGL(sod_loss_grad_same_kernel, dim3(gx, (unsigned)planes), dim3(256), 0, s, logits, label, coef, gscale,
                           glogits, npix);
    } else {
        TRAMBA_CHECK(hout >= h && wout >= w, "sod_loss_grad: outputs are resized UP to the label (%dx%d -> %dx%d)", h, w, hout, wout);
        TRAMBA_CHECK(wout <= 4096, "sod_loss_grad: label rows of at most 4096 pixels (got %d)", wout);
        TRAMBA_CHECK(workspace && workspace_bytes >= tramba_sod_loss_grad_workspace(planes, h, w, hout, wout),
                     "sod_loss_grad: workspace of %zu bytes needed", tramba_sod_loss_grad_workspace(planes, h, w, hout, wout));
        float *rows = (float *)workspace;
        hipLaunchKernelGGL(sod_loss_grad_rows_kernel, dim3((unsigned)((hout + kGradRows - 1) / kGradRows), (unsigned)planes),
                           dim3(256), (size_t)kGradRows * wout * sizeof(float), s, logits, label, coef, gscale, rows, h, w, hout,
                           wout);
        TRAMBA_LAUNCH_CHECK();
        const long total = (long)planes * h * w;
        TRAMBA_CHECK((total + 255) / 256 < 2147483647L, "sod_loss_grad: too many workgroups");
        hipLaunchKernelGGL(sod_loss_grad_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rows, glogits, h, w,
                           hout, total);
    }
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_adam_step(float *const *params, const float *const *grads, float *const *exp_avg,
                                float *const *exp_avg_sq, float *const *steps, const int64_t *numel, int count, double lr,
                                double beta1, double beta2, double eps, double weight_decay, void *stream)
{
    TRAMBA_CHECK(params && grads && exp_avg && exp_avg_sq && steps && numel && count > 0, "adam_step: empty input");
    TRAMBA_CHECK(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0,
                 "adam_step: bad hyper-parameters");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < count; ++i)
        TRAMBA_CHECK(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && steps[i] && numel[i] > 0,
                     "adam_step: tensor %d: null pointer or no elements", i);
    for (int base = 0; base < count; base += kBumpTensors) {     // t += 1 on the device (the counters are part of the state_dict)
        BumpArgs b;
        b.count = count - base < kBumpTensors ? count - base : kBumpTensors;
        for (int i = 0; i < kBumpTensors; ++i) b.step[i] = i < b.count ? steps[base + i] : nullptr;
        hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(kBumpTensors), 0, s, b);
        TRAMBA_LAUNCH_CHECK();
    }
    // Launches of equal weight: the tensors are dealt to ceil(count / 72) launches largest first, in snake order (a launch of
    // 72 bias vectors alone would put 72 workgroups on 256 CUs; the order of the updates does not matter, they are independent).
    const int nlaunch = (count + kAdamTensors - 1) / kAdamTensors;
    std::vector<int> order(count);
    for (int i = 0; i < count; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return numel[x] > numel[y]; });
    std::vector<std::vector<int>> bins(nlaunch);
    for (int i = 0; i < count; ++i) {
        const int round = i / nlaunch, pos = i % nlaunch;
        bins[(round & 1) ? nlaunch - 1 - pos : pos].push_back(order[i]);
    }
    for (int l = 0; l < nlaunch; ++l) {
        AdamArgs a;
        long blocks = 0;
        const int c = (int)bins[l].size();           // <= kAdamTensors by construction
        for (int i = 0; i < c; ++i) {
            const int j = bins[l][i];
            const long nb = (numel[j] + kAdamChunk - 1) / kAdamChunk;
            TRAMBA_CHECK(blocks + nb < 2147483647L, "adam_step: too many workgroups");
            a.t[i] = AdamTensor{params[j], grads[j], exp_avg[j], exp_avg_sq[j], steps[j], (long)numel[j]};
            a.first[i] = (int)blocks;
            blocks += nb;
        }
        a.count = c;
        for (int i = c; i <= kAdamTensors; ++i) a.first[i] = (int)blocks;
        for (int i = c; i < kAdamTensors; ++i) a.t[i] = AdamTensor{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kAdamThreads), 0, s, a);
        TRAMBA_LAUNCH_CHECK();
    }
    return TRAMBA_OK;
}
