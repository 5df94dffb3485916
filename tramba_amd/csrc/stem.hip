// Patch-embed stem of the VMamba encoder, fused: conv 3x3 / stride 2 / pad 1 (3 -> 64 channels) + bias +
// LayerNorm2d(64) + GELU  (Models/vmamba.py:481-485, the first three stages of patch_embed).
//
// Reads the image in its native NCHW layout and dtype (fp32 or the activation dtype), writes the
// channels-last activation.  The layer is tiny in flops (27 MACs x 64 outputs per pixel) and HBM-bound on
// its 128-byte-per-pixel output; MIOpen's generic grouped-conv kernel spends ~250 us on it.  Here 4 lanes
// share one output pixel (16 output channels each), the 27x64 filter sits in LDS tap-major, and the
// LayerNorm reduction is two DPP/shuffle steps across those 4 lanes.
#include "common.h"

namespace tramba {

constexpr int kStemCout = 64, kStemTaps = 27;

template <typename TI, typename T>
__global__ __launch_bounds__(256) void stem_conv_ln_gelu_kernel(const TI *__restrict__ img, const float *__restrict__ w,
                                                               const float *__restrict__ bias,
                                                               const float *__restrict__ ln_w,
                                                               const float *__restrict__ ln_b, T *__restrict__ y,
                                                               int B, int H, int W, int Ho, int Wo, float eps)
{
    __shared__ __attribute__((aligned(16))) float wl[kStemTaps][kStemCout];  // [ci*9 + ky*3 + kx][cout]
    for (int t = threadIdx.x; t < kStemTaps * kStemCout; t += blockDim.x) {
        const int co = t / kStemTaps, tap = t % kStemTaps;  // reference layout (Cout, Cin, 3, 3)
        wl[tap][co] = w[t];
    }
    __syncthreads();
    const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int part = threadIdx.x & 3;  // 16 output channels each
    const long npix = (long)B * Ho * Wo;
    const bool ok = pix < npix;
    const long pp = ok ? pix : npix - 1;
    const int wo = (int)(pp % Wo);
    const long t2 = pp / Wo;
    const int ho = (int)(t2 % Ho), b = (int)(t2 / Ho);

    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[part * 16 + j];
    // (one input channel per trip, NOT unrolled: fully unrolled, hipcc hoists all 108 16-byte weight reads to the top
    //  and parks them in AGPRs -- 242 v_accvgpr_read per thread and one wave per SIMD)
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const int hy = 2 * ho + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int wx = 2 * wo + kx - 1;
                float v = 0.f;
                if (hy >= 0 && hy < H && wx >= 0 && wx < W) v = Cvt<TI>::to_f(img[(((long)b * 3 + ci) * H + hy) * W + wx]);
                const float4 *wp = reinterpret_cast<const float4 *>(&wl[ci * 9 + ky * 3 + kx][part * 16]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 ww = wp[q];
                    acc[4 * q + 0] = fmaf(v, ww.x, acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(v, ww.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(v, ww.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(v, ww.w, acc[4 * q + 3]);
                }
            }
        }
    // LayerNorm over the 64 channels of the pixel = 4 lanes x 16
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j];
    s += __shfl_xor(s, 1, 4);
    s += __shfl_xor(s, 2, 4);
    const float mean = s * (1.f / kStemCout);
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float d = acc[j] - mean;
        q2 = fmaf(d, d, q2);
    }
    q2 += __shfl_xor(q2, 1, 4);
    q2 += __shfl_xor(q2, 2, 4);
    const float rstd = rsqrtf(q2 * (1.f / kStemCout) + eps);
    if (!ok) return;
    T *yo = y + pix * kStemCout + part * 16;
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = part * 16 + h8 * 8 + j;
            o[j] = geluf_((acc[h8 * 8 + j] - mean) * rstd * ln_w[c] + ln_b[c]);
        }
        store_pack<T, 8>(yo + h8 * 8, o);
    }
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_stem_conv_ln_gelu(const void *img, const float *w, const float *bias, const float *ln_w,
                                        const float *ln_b, void *y, int batch, int h, int wd, float eps,
                                        int img_dtype, int dtype, void *stream)
{
    TRAMBA_CHECK(img && w && bias && ln_w && ln_b && y, "stem_conv_ln_gelu: null tensor");
    TRAMBA_CHECK(batch > 0 && h > 0 && wd > 0, "stem_conv_ln_gelu: empty shape");
    TRAMBA_CHECK(img_dtype == TRAMBA_F32 || img_dtype == dtype, "stem_conv_ln_gelu: image must be f32 or the activation dtype");
    TRAMBA_CHECK(aligned16(y), "stem_conv_ln_gelu: output must be 16-byte aligned");
    const int ho = (h + 1) / 2, wo = (wd + 1) / 2;
    const long npix = (long)batch * ho * wo;
    dim3 grid((unsigned)((npix + 63) / 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (img_dtype == TRAMBA_F32 && dtype != TRAMBA_F32)
            hipLaunchKernelGGL((stem_conv_ln_gelu_kernel<float, T>), grid, block, 0, s, (const float *)img, w, bias, ln_w,
                               ln_b, (T *)y, batch, h, wd, ho, wo, eps);
        else
            hipLaunchKernelGGL((stem_conv_ln_gelu_kernel<T, T>), grid, block, 0, s, (const T *)img, w, bias, ln_w, ln_b,
                               (T *)y, batch, h, wd, ho, wo, eps);
    });
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
