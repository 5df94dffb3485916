// 1x1 convolution == row-major GEMM in channels-last:  Y[M,N] = epilogue(X[M,K] W[N,K]^T).
//
// Replaces Linear2d.forward (Models/modules.py:10-13: F.conv2d with a (out,in,1,1) weight) and
// the grouped x_proj conv1d (vmamba.py:233-234, evaluated once in spatial order, see
// ss2d_fused.hip).  Both operands are K-contiguous, which is exactly the MFMA A/B fragment
// order on gfx950: lane l of v_mfma_f32_32x32x16_{bf16,f16} holds A[row l&31][k = 8(l>>5)+j]
// and B[k = 8(l>>5)+j][col l&31], j = 0..7 -- one 16-byte load per fragment, no transpose.
// fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
//
// v1 structure: 256 threads = 2x2 waves, each wave a 64x64 output tile (2x2 MFMA tiles, 64
// accumulator registers), fragments loaded straight from global/L2 (the activations of this
// network are short-K: K = 128..4096, most GEMMs are HBM/L2-bound on the M x N output, so the
// fused epilogue matters more than LDS staging; an LDS-staged variant is the next step).
// Epilogue fused: + bias, activation, + residual, cast.
#include "common.h"

namespace tramba {

typedef __attribute__((ext_vector_type(8))) short frag8_t;     // 8 x 16-bit
typedef __attribute__((ext_vector_type(8))) _Float16 frag8h_t; // 8 x f16
typedef __attribute__((ext_vector_type(16))) float acc16_t;

template <typename T> struct Mfma;
template <> struct Mfma<__hip_bfloat16> {
    static constexpr int KSTEP = 16;
    using frag = frag8_t;
    static __device__ __forceinline__ acc16_t run(frag a, frag b, acc16_t c)
    {
        typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<__half> {
    static constexpr int KSTEP = 16;
    using frag = frag8_t;
    static __device__ __forceinline__ acc16_t run(frag a, frag b, acc16_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(frag8h_t, a), __builtin_bit_cast(frag8h_t, b), c, 0, 0, 0);
    }
};

// 16-bit operands: fragment = 8 consecutive k of one row
template <typename T>
__device__ __forceinline__ frag8_t load_frag16(const T *__restrict__ base, long row, long nrows, int k0, int K,
                                               bool vec)
{
    frag8_t f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nrows) {
        const T *p = base + row * K + k0;
        if (vec && k0 + 8 <= K) {
            f = *reinterpret_cast<const frag8_t *>(p);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < K) f[j] = *reinterpret_cast<const short *>(p + j);
        }
    }
    return f;
}

template <typename T, typename TO>
__device__ __forceinline__ void store_tile(const acc16_t &acc, TO *__restrict__ y, const float *__restrict__ bias,
                                           const T *__restrict__ res, long m0, int n0, long M, int N, int act,
                                           int lane)
{
    const int col = n0 + (lane & 31);
    if (col >= N) return;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) {
            float v = apply_act(acc[r] + bv, act);
            if (res) v += Cvt<T>::to_f(res[row * N + col]);
            y[row * N + col] = Cvt<TO>::from_f(v);
        }
    }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void linear16_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                      const float *__restrict__ bias, const T *__restrict__ res,
                                                      TO *__restrict__ y, long M, int N, int K, int act, int vec)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.y * 128 + (wave >> 1) * 64;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (m0 >= M || n0 >= N) return;
    acc16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r32 = lane & 31, kh = (lane >> 5) * 8;
    for (int k0 = 0; k0 < K; k0 += 16) {
        frag8_t a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = load_frag16<T>(x, m0 + i * 32 + r32, M, k0 + kh, K, vec);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = load_frag16<T>(w, n0 + j * 32 + r32, N, k0 + kh, K, vec);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Mfma<T>::run(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<T, TO>(acc[i][j], y, bias, res, m0 + i * 32, n0 + j * 32, M, N, act, lane);
}

// fp32: v_mfma_f32_32x32x2_f32, A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
template <typename TO>
__global__ __launch_bounds__(256) void linear32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, const float *__restrict__ res,
                                                      TO *__restrict__ y, long M, int N, int K, int act)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long m0 = (long)blockIdx.y * 128 + (wave >> 1) * 64;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (m0 >= M || n0 >= N) return;
    acc16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r32 = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += 2) {
        const int k = k0 + kh;
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long row = m0 + i * 32 + r32;
            a[i] = (row < M && k < K) ? x[row * K + k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + j * 32 + r32;
            b[j] = (col < N && k < K) ? w[(long)col * K + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_tile<float, TO>(acc[i][j], y, bias, res, m0 + i * 32, n0 + j * 32, M, N, act, lane);
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_linear_cl(const void *x, const void *w, const float *bias, const void *residual,
                                void *y, int64_t m, int n, int k, int act, int dtype, int out_dtype,
                                void *stream)
{
    TRAMBA_CHECK(x && w && y, "linear_cl: null tensor");
    TRAMBA_CHECK(m > 0 && n > 0 && k > 0, "linear_cl: empty shape");
    TRAMBA_CHECK(out_dtype == dtype || out_dtype == TRAMBA_F32, "linear_cl: out dtype must be dtype or f32");
    const long gy = (m + 127) / 128, gx = (n + 127) / 128;
    TRAMBA_CHECK(gy <= 65535, "linear_cl: M=%ld exceeds grid limits", (long)m);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)gx, (unsigned)gy), block(256);
    const int vec = (k % 8 == 0) && aligned16(x) && aligned16(w);
    if (dtype == TRAMBA_F32) {
        hipLaunchKernelGGL(linear32_kernel<float>, grid, block, 0, s, (const float *)x, (const float *)w, bias,
                           (const float *)residual, (float *)y, (long)m, n, k, act);
    } else if (dtype == TRAMBA_BF16) {
        using T = __hip_bfloat16;
        if (out_dtype == TRAMBA_F32)
            hipLaunchKernelGGL((linear16_kernel<T, float>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (float *)y, (long)m, n, k, act, vec);
        else
            hipLaunchKernelGGL((linear16_kernel<T, T>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (T *)y, (long)m, n, k, act, vec);
    } else if (dtype == TRAMBA_F16) {
        using T = __half;
        if (out_dtype == TRAMBA_F32)
            hipLaunchKernelGGL((linear16_kernel<T, float>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (float *)y, (long)m, n, k, act, vec);
        else
            hipLaunchKernelGGL((linear16_kernel<T, T>), grid, block, 0, s, (const T *)x, (const T *)w, bias,
                               (const T *)residual, (T *)y, (long)m, n, k, act, vec);
    } else {
        set_error("linear_cl: bad dtype %d", dtype);
        return TRAMBA_ERR_ARG;
    }
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
