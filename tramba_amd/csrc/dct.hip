// 2-D DCT frequency split, channels-last (Models/DCT_2D.py:6-77).
//
// The reference evaluates Y = Wy X Wx^T as 2n broadcast-multiply + 2n reductions per call and
// then keeps only the LL quadrant ("low") and the HH quadrant ("high").  Here it is the
// separable product it is, in two passes with lanes = channels (coalesced C-vectors, the
// (n,n) cosine tables are wave-uniform scalars):
//   pass 1   T[b,i,u,:] = sum_j Wx[u,j] X[b,i,j,:]          all u           (fp32 scratch)
//   pass 2   low [b,v,u,:] = sum_i Wy[v,i]     T[b,i,u,:]    v,u <  n/2
//            high[b,v,u,:] = sum_i Wy[v+n/2,i] T[b,i,u+n/2,:]
// i.e. only the two kept quadrants are ever produced (half of pass 2 is skipped).
// ~0.5 GFLOP per image in total: VALU-resident, bound by the 1.5x-of-input HBM traffic.
#include "common.h"

namespace tramba {

constexpr int kUT = 8;  // outputs per register tile
constexpr int kJT = 4;  // inputs per register tile (n % 4 == 0 required)

// out[o, :] = sum_j Wm[row0 + o, j] * in[j*in_stride + :]   for o in [0, nout)
template <typename TI, typename TOUT>
__device__ __forceinline__ void dct_rows(const TI *__restrict__ in, long in_stride,
                                         const float *__restrict__ Wm, int n, int row0, int nout,
                                         TOUT *__restrict__ out, long out_stride, bool cok)
{
    for (int o0 = 0; o0 < nout; o0 += kUT) {
        float acc[kUT];
#pragma unroll
        for (int a = 0; a < kUT; ++a) acc[a] = 0.f;
        for (int j0 = 0; j0 < n; j0 += kJT) {
            float xv[kJT];
#pragma unroll
            for (int q = 0; q < kJT; ++q) xv[q] = cok ? Cvt<TI>::to_f(in[(long)(j0 + q) * in_stride]) : 0.f;
#pragma unroll
            for (int a = 0; a < kUT; ++a) {
                const int o = o0 + a < nout ? o0 + a : nout - 1;  // clamp: masked at the store
                const float *wr = Wm + (long)(row0 + o) * n + j0;
#pragma unroll
                for (int q = 0; q < kJT; ++q) acc[a] = fmaf(wr[q], xv[q], acc[a]);
            }
        }
#pragma unroll
        for (int a = 0; a < kUT; ++a)
            if (o0 + a < nout && cok) out[(long)(o0 + a) * out_stride] = Cvt<TOUT>::from_f(acc[a]);
    }
}

// One block = one input row i (pass 1) / one output column u (pass 2); its 4 waves split the
// outputs of that row in quarters, so every wave runs a 4x shorter FMA chain and the grid has 4x
// the waves (the one-wave-per-row form left most of the chip idle at n = 96).
template <typename T>
__global__ __launch_bounds__(256) void dct_pass1_kernel(const T *__restrict__ x, const float *__restrict__ wx,
                                                       float *__restrict__ tmp, int n, int C)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.y;
    const int b = blockIdx.z;
    const int c = blockIdx.x * kWave + lane;
    const bool cok = c < C;
    const int per = ((n + 3) / 4 + kUT - 1) / kUT * kUT;   // outputs per wave, multiple of the register tile
    const int o0 = wv * per;
    if (o0 >= n) return;
    const int cnt = o0 + per <= n ? per : n - o0;
    const long base = (((long)b * n + i) * n) * C + (cok ? c : 0);
    dct_rows<T, float>(x + base, C, wx, n, o0, cnt, tmp + base + (long)o0 * C, C, cok);
}

template <typename T>
__global__ __launch_bounds__(256) void dct_pass2_kernel(const float *__restrict__ tmp,
                                                       const float *__restrict__ wy, T *__restrict__ high,
                                                       T *__restrict__ low, int n, int C)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = blockIdx.y;
    const int b = blockIdx.z;
    const int c = blockIdx.x * kWave + lane;
    const bool cok = c < C;
    const int hn = n / 2;
    const bool hi = u >= hn;
    const int per = ((hn + 3) / 4 + kUT - 1) / kUT * kUT;
    const int o0 = wv * per;
    if (o0 >= hn) return;
    const int cnt = o0 + per <= hn ? per : hn - o0;
    const float *in = tmp + ((long)b * n * n + u) * C + (cok ? c : 0);   // T[b, i, u, c], stride n*C over i
    T *out = (hi ? high : low) + (((long)b * hn + o0) * hn + (hi ? u - hn : u)) * C + (cok ? c : 0);
    dct_rows<float, T>(in, (long)n * C, wy, n, (hi ? hn : 0) + o0, cnt, out, (long)hn * C, cok);
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_dct_split_cl(const void *x, const float *wx, const float *wy, float *tmp, void *high,
                                   void *low, int batch, int n, int c, int dtype, void *stream)
{
    TRAMBA_CHECK(x && wx && wy && tmp && high && low, "dct_split_cl: null tensor");
    TRAMBA_CHECK(batch > 0 && n > 0 && c > 0, "dct_split_cl: empty shape");
    TRAMBA_CHECK(n % kJT == 0, "dct_split_cl: n=%d must be a multiple of %d", n, kJT);
    TRAMBA_CHECK(batch <= 65535, "dct_split_cl: batch exceeds grid limits");
    hipStream_t s = (hipStream_t)stream;
    TRAMBA_CHECK(n <= 65535, "dct_split_cl: n exceeds grid limits");
    dim3 grid((c + kWave - 1) / kWave, n, batch), block(256);
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL(dct_pass1_kernel<T>, grid, block, 0, s, (const T *)x, wx, tmp, n, c);
        hipLaunchKernelGGL(dct_pass2_kernel<T>, grid, block, 0, s, (const float *)tmp, wy, (T *)high, (T *)low, n, c);
    });
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
