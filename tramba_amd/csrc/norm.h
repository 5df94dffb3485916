// Wave-level LayerNorm helper shared by the norm kernels.
#pragma once
#include "common.h"

namespace tramba {

constexpr int kNormMaxIt = 8;

// One wave normalises one row held as acc[it][v] (channel = (it*64 + lane)*V + v).
template <int V>
__device__ __forceinline__ void wave_layernorm(float (&acc)[kNormMaxIt][V], int nit, int C, int lane,
                                               float eps, float &mean, float &rstd)
{
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it)
        if (it < nit)
#pragma unroll
            for (int v = 0; v < V; ++v)
                if ((it * kWave + lane) * V + v < C) s += acc[it][v];
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < kNormMaxIt; ++it)
        if (it < nit)
#pragma unroll
            for (int v = 0; v < V; ++v)
                if ((it * kWave + lane) * V + v < C) {
                    const float t = acc[it][v] - mean;
                    q = fmaf(t, t, q);
                }
    rstd = rsqrtf(wave_sum(q) / (float)C + eps);
}

// vector width for a row of C channels: the largest of 8/4/2/1 that divides C and keeps the
// row within kNormMaxIt wave iterations while using as many lanes as possible
inline int norm_vec(int c, int maxv)
{
    int v = maxv;
    while (v > 1 && (c % v != 0 || c / v < kWave)) v >>= 1;
    while ((c + kWave * v - 1) / (kWave * v) > kNormMaxIt && v < maxv && c % (2 * v) == 0) v <<= 1;
    return v;
}

}  // namespace tramba
