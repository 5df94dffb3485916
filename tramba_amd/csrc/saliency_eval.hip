// Per-image statistics of a saliency map against its mask: everything the reference's evaluation loop
// (train.py:101-139 test_one_epoch; Evaluation/metrics.py) needs for MAE, F-measure (adaptive + 256-threshold
// curve, precision / recall, FNR), E-measure (adaptive + curve) and S-measure, reduced on the GPU in one launch,
// so the (B, H, W) prediction never has to travel to the host.  One block per image, three sweeps:
//   1  min / max of pred, mask area and centroid sums (Evaluation/metrics.py:13-19, 201-212);
//   2  p = (pred - min) / (max - min) in fp32 exactly as numpy does it; sum p, sum |p - g|, the two 256-bin
//      histograms of uint8(p * 255) over mask / background (:60-64, 323-327), the object-score sums (:175-187) and
//      the four quadrant sums about the centroid (:189-260);
//   3  the two counts at the adaptive threshold min(2 * mean(p), 1) (:22-23, 47-57, 287-312) and the centred second
//      moments of the quadrants and of the object scores (two-pass, as numpy computes them).
// Integer outputs are exact; floating sums are accumulated in fp64 in a fixed order (bitwise reproducible).
#include "common.h"

namespace tramba {

constexpr int kEvalThreads = 1024;
constexpr int kNI = TRAMBA_EVAL_NINT, kND = TRAMBA_EVAL_NDBL;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sums acc[0..N) over the block in a fixed order; result valid in every thread (through LDS)
template <typename T, int N>
__device__ __forceinline__ void block_sum(T (&acc)[N], T *lds /* [16][N] + [N] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const T s = wave_sum(acc[i]);
        if (lane == 0) lds[wave * N + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        T s = 0;
        for (int w = 0; w < kEvalThreads / 64; ++w) s += lds[w * N + threadIdx.x];
        lds[(kEvalThreads / 64) * N + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = lds[(kEvalThreads / 64) * N + i];
    __syncthreads();
}

__global__ __launch_bounds__(kEvalThreads) void saliency_stats_kernel(const float *__restrict__ pred,
                                                                     const unsigned char *__restrict__ gt,
                                                                     long long *__restrict__ ints,
                                                                     double *__restrict__ dbl, int H, int W)
{
    __shared__ double lds_d[17 * 24];
    __shared__ long long lds_i[17 * 4];
    __shared__ int hist[512];
    __shared__ float lds_f[2 * 16];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = H * W;
    const float *p = pred + (size_t)img * n;
    const unsigned char *g = gt + (size_t)img * n;
    long long *oi = ints + (size_t)img * kNI;
    double *od = dbl + (size_t)img * kND;
    if (tid < 512) hist[tid] = 0;

    // ---- sweep 1
    float mn = INFINITY, mx = -INFINITY;
    long long ia[4] = {0, 0, 0, 0};   // area, sum g*col, sum g*row
    for (int i = tid; i < n; i += kEvalThreads) {
        const float v = p[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        if (g[i]) {
            ia[0] += 1;
            ia[1] += i % W;
            ia[2] += i / W;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) {
        lds_f[wave] = mn;
        lds_f[16 + wave] = mx;
    }
    __syncthreads();
    for (int w = 0; w < 16; ++w) {
        mn = fminf(mn, lds_f[w]);
        mx = fmaxf(mx, lds_f[16 + w]);
    }
    block_sum<long long, 4>(ia, lds_i);
    const long long area = ia[0];
    // centroid: numpy's round (half to even) of the exact quotient, + 1; the image centre for an empty mask
    int cx, cy;
    if (area == 0) {
        cx = (int)rint((double)W / 2.0) + 1;
        cy = (int)rint((double)H / 2.0) + 1;
    } else {
        cx = (int)rint((double)ia[1] / (double)area) + 1;
        cy = (int)rint((double)ia[2] / (double)area) + 1;
    }
    const bool norm = mx != mn;
    const float span = mx - mn;

    // ---- sweep 2
    double da[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) da[i] = 0.0;
    for (int i = tid; i < n; i += kEvalThreads) {
        float v = p[i];
        if (norm) v = (v - mn) / span;
        const int m = g[i] ? 1 : 0;
        const float gf = (float)m;
        da[2] += (double)v;
        da[3] += (double)fabsf(v - gf);
        const int q8 = (int)(unsigned char)(v * 255.0f);
        atomicAdd(&hist[(m ? 0 : 256) + q8], 1);
        if (m) {
            da[4] += (double)v;
            da[5] += (double)v * (double)v;
        } else {
            const float u = 1.0f - v;
            da[6] += (double)u;
            da[7] += (double)u * (double)u;
        }
        const int r = i / W, c = i % W;
        const int quad = (r >= cy ? 2 : 0) + (c >= cx ? 1 : 0);
        // quadrant sums through selects (no dynamic register indexing)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double on = quad == k ? 1.0 : 0.0;
            da[8 + 4 * k + 0] += on * (double)v;
            da[8 + 4 * k + 1] += on * (double)v * (double)v;
            da[8 + 4 * k + 2] += on * (double)gf;
            da[8 + 4 * k + 3] += on * (double)v * (double)gf;
        }
    }
    block_sum<double, 24>(da, lds_d);   // its barriers also publish the LDS histogram
    const float mean32 = (float)(da[2] / (double)n);
    const float thr = fminf(2.0f * mean32, 1.0f);

    // ---- sweep 3: adaptive-threshold counts, and the CENTRED second moments (two-pass like numpy, so a constant
    // region gives exactly 0 and the S-measure's degenerate branches, metrics.py:253-258, are taken identically)
    double qn[4], mp[4], mg[4];
    qn[0] = (double)cy * cx; qn[1] = (double)cy * (W - cx); qn[2] = (double)(H - cy) * cx; qn[3] = (double)(H - cy) * (W - cx);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mp[k] = da[8 + 4 * k + 0] / qn[k];
        mg[k] = da[8 + 4 * k + 2] / qn[k];
    }
    const double mfg = da[4] / (double)area, mbg = da[6] / (double)(n - area);
    long long ib[4] = {0, 0, 0, 0};
    double dc[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) dc[i] = 0.0;
    for (int i = tid; i < n; i += kEvalThreads) {
        float v = p[i];
        if (norm) v = (v - mn) / span;
        const int m = g[i] ? 1 : 0;
        if (v >= thr) ib[m ? 0 : 1] += 1;
        const int r = i / W, c = i % W;
        const int quad = (r >= cy ? 2 : 0) + (c >= cx ? 1 : 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double on = quad == k ? 1.0 : 0.0;
            const double dp = (double)v - mp[k], dg = (double)m - mg[k];
            dc[3 * k + 0] += on * dp * dp;
            dc[3 * k + 1] += on * dg * dg;
            dc[3 * k + 2] += on * dp * dg;
        }
        if (m) {
            const double d = (double)v - mfg;
            dc[12] += d * d;
        } else {
            const double d = (double)(1.0f - v) - mbg;
            dc[13] += d * d;
        }
    }
    block_sum<long long, 4>(ib, lds_i);
    block_sum<double, 14>(dc, lds_d);

    if (tid == 0) {
        oi[0] = area; oi[1] = ia[1]; oi[2] = ia[2]; oi[3] = cx; oi[4] = cy; oi[5] = ib[0]; oi[6] = ib[1]; oi[7] = n;
        od[0] = (double)mn; od[1] = (double)mx;
        for (int i = 2; i < 24; ++i) od[i] = da[i];
        od[24] = (double)thr;
        for (int i = 25; i < 32; ++i) od[i] = 0.0;
        for (int i = 0; i < 14; ++i) od[32 + i] = dc[i];
        for (int i = 46; i < kND; ++i) od[i] = 0.0;
    }
    if (tid < 512) oi[8 + tid] = hist[tid];
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_saliency_stats(const float *pred, const unsigned char *gt, long long *ints, double *dbl, int batch,
                                     int h, int w, void *stream)
{
    TRAMBA_CHECK(pred && gt && ints && dbl, "saliency_stats: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && w > 0, "saliency_stats: empty shape");
    TRAMBA_CHECK((long)h * w < (1L << 30), "saliency_stats: image too large");
    hipLaunchKernelGGL(saliency_stats_kernel, dim3(batch), dim3(kEvalThreads), 0, (hipStream_t)stream, pred, gt, ints, dbl, h,
                       w);
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
