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
// ~0.5 GFLOP per image in total: FMA issue (lanes = channels, weights broadcast from LDS), then the 1.5x-of-input HBM
// traffic.
#include "common.h"

namespace tramba {

constexpr int kJT = 4;  // n % 4 == 0 required (16-byte aligned rows of the transposed cosine table)

// The (n, n) cosine table, TRANSPOSED into LDS (wl[j * n + r] = Wm[r, j], + PER floats of padding): for one input j
// the weights of a wave's PER consecutive outputs are then PER / 4 broadcast 16-byte LDS reads, and the wave walks its
// input row ONCE with all PER accumulators live.  (The first form tiled 8 outputs x 4 inputs in registers with the
// weights as scalar loads: one s_load per FMA, and every input re-read once per 8 outputs -- 72 us for the 96x96 map
// against ~12 us of FMA issue.)
// (row stride n + 4 floats: rows stay 16-byte aligned and the transposing fill is 4-way instead of 32-way conflicting)
__device__ __forceinline__ void dct_table_to_lds(const float *__restrict__ Wm, float *wl, int n, int pad)
{
    const int ldw = n + 4;
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) {
        const int r = t / n, j = t - r * n;      // coalesced read along j
        wl[j * ldw + r] = Wm[t];
    }
    for (int t = threadIdx.x; t < 4 * n + pad; t += blockDim.x) {   // the 4 pad columns of every row, and the tail
        if (t < 4 * n) wl[(t >> 2) * ldw + n + (t & 3)] = 0.f;
        else wl[n * ldw + (t - 4 * n)] = 0.f;
    }
    __syncthreads();
}

// out[o, :] = sum_j Wm[row0 + o, j] * in[j*in_stride + :]   for o in [0, nout), nout <= PER
template <typename TI, typename TOUT, int PER>
__device__ __forceinline__ void dct_rows(const TI *__restrict__ in, long in_stride, const float *wl, int n, int row0,
                                         int nout, TOUT *__restrict__ out, long out_stride, bool cok)
{
    const int LDW = n + 4;
    float acc[PER];
#pragma unroll
    for (int a = 0; a < PER; ++a) acc[a] = 0.f;
    // inputs in batches of kJB loads issued together (one 2-byte element per lane and input: with 4 in flight the loop
    // ran at memory latency / 4 per input)
    constexpr int kJB = 12;   // divides 12, 24, 48, 96; n % 4 == 0 in general: the tail batch is masked
    for (int j0 = 0; j0 < n; j0 += kJB) {
        TI raw[kJB];
#pragma unroll
        for (int q = 0; q < kJB; ++q) {
            const int j = j0 + q < n ? j0 + q : n - 1;
            raw[q] = in[(long)j * in_stride];
        }
#pragma unroll
        for (int q = 0; q < kJB; ++q) {
            const float xv = (cok && j0 + q < n) ? Cvt<TI>::to_f(raw[q]) : 0.f;
            const int j = j0 + q < n ? j0 + q : n - 1;
            const float4 *w4 = reinterpret_cast<const float4 *>(wl + j * LDW + row0);   // wave-uniform address: broadcast
#pragma unroll
            for (int a = 0; a < PER / 4; ++a) {
                const float4 ww = w4[a];
                acc[4 * a + 0] = fmaf(ww.x, xv, acc[4 * a + 0]);
                acc[4 * a + 1] = fmaf(ww.y, xv, acc[4 * a + 1]);
                acc[4 * a + 2] = fmaf(ww.z, xv, acc[4 * a + 2]);
                acc[4 * a + 3] = fmaf(ww.w, xv, acc[4 * a + 3]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < PER; ++a)
        if (a < nout && cok) out[(long)a * out_stride] = Cvt<TOUT>::from_f(acc[a]);
}

// One block = one input row i (pass 1) / one output column u (pass 2); its 4 waves split the outputs of that row in
// quarters of PER (a multiple of 8 >= n / 4).
template <typename T, int PER>
__global__ __launch_bounds__(256) void dct_pass1_kernel(const T *__restrict__ x, const float *__restrict__ wx,
                                                       float *__restrict__ tmp, int n, int C)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];
    dct_table_to_lds(wx, wl, n, PER);
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.y;
    const int b = blockIdx.z;
    const int c = blockIdx.x * kWave + lane;
    const bool cok = c < C;
    const int o0 = wv * PER;
    if (o0 >= n) return;
    const int cnt = o0 + PER <= n ? PER : n - o0;
    const long base = (((long)b * n + i) * n) * C + (cok ? c : 0);
    dct_rows<T, float, PER>(x + base, C, wl, n, o0, cnt, tmp + base + (long)o0 * C, C, cok);
}

template <typename T, int PER>
__global__ __launch_bounds__(256) void dct_pass2_kernel(const float *__restrict__ tmp,
                                                       const float *__restrict__ wy, T *__restrict__ high,
                                                       T *__restrict__ low, int n, int C)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];
    dct_table_to_lds(wy, wl, n, PER);
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u = blockIdx.y;
    const int b = blockIdx.z;
    const int c = blockIdx.x * kWave + lane;
    const bool cok = c < C;
    const int hn = n / 2;
    const bool hi = u >= hn;
    const int o0 = wv * PER;
    if (o0 >= hn) return;
    const int cnt = o0 + PER <= hn ? PER : hn - o0;
    const float *in = tmp + ((long)b * n * n + u) * C + (cok ? c : 0);   // T[b, i, u, c], stride n*C over i
    T *out = (hi ? high : low) + (((long)b * hn + o0) * hn + (hi ? u - hn : u)) * C + (cok ? c : 0);
    dct_rows<float, T, PER>(in, (long)n * C, wl, n, (hi ? hn : 0) + o0, cnt, out, (long)hn * C, cok);
}

// ---- scalar-weight form (first version): kept for maps whose cosine table does not fit LDS (n > 112, e.g. 192 at 768x768)
constexpr int kUT = 8;  // outputs per register tile
constexpr int kJTo = 4; // inputs per register tile of the scalar-weight form

// out[o, :] = sum_j Wm[row0 + o, j] * in[j*in_stride + :]   for o in [0, nout)
template <typename TI, typename TOUT>
__device__ __forceinline__ void dct_rows_sw(const TI *__restrict__ in, long in_stride,
                                         const float *__restrict__ Wm, int n, int row0, int nout,
                                         TOUT *__restrict__ out, long out_stride, bool cok)
{
    for (int o0 = 0; o0 < nout; o0 += kUT) {
        float acc[kUT];
#pragma unroll
        for (int a = 0; a < kUT; ++a) acc[a] = 0.f;
        for (int j0 = 0; j0 < n; j0 += kJTo) {
            float xv[kJTo];
#pragma unroll
            for (int q = 0; q < kJTo; ++q) xv[q] = cok ? Cvt<TI>::to_f(in[(long)(j0 + q) * in_stride]) : 0.f;
#pragma unroll
            for (int a = 0; a < kUT; ++a) {
                const int o = o0 + a < nout ? o0 + a : nout - 1;  // clamp: masked at the store
                const float *wr = Wm + (long)(row0 + o) * n + j0;
#pragma unroll
                for (int q = 0; q < kJTo; ++q) acc[a] = fmaf(wr[q], xv[q], acc[a]);
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
__global__ __launch_bounds__(256) void dct_pass1_sw_kernel(const T *__restrict__ x, const float *__restrict__ wx,
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
    dct_rows_sw<T, float>(x + base, C, wx, n, o0, cnt, tmp + base + (long)o0 * C, C, cok);
}

template <typename T>
__global__ __launch_bounds__(256) void dct_pass2_sw_kernel(const float *__restrict__ tmp,
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
    dct_rows_sw<float, T>(in, (long)n * C, wy, n, (hi ? hn : 0) + o0, cnt, out, (long)hn * C, cok);
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
    if (n > 112) {   // the transposed cosine table (n x (n + 4) floats) would not fit 64 KB of LDS
        TRAMBA_DISPATCH_DTYPE(dtype, T, {
            hipLaunchKernelGGL(dct_pass1_sw_kernel<T>, grid, block, 0, s, (const T *)x, wx, tmp, n, c);
            hipLaunchKernelGGL(dct_pass2_sw_kernel<T>, grid, block, 0, s, (const float *)tmp, wy, (T *)high, (T *)low, n, c);
        });
        TRAMBA_LAUNCH_CHECK();
        return TRAMBA_OK;
    }
    const int per1 = ((n + 3) / 4 + 7) / 8 * 8, per2 = ((n / 2 + 3) / 4 + 7) / 8 * 8;   // outputs per wave
    const size_t lds1 = (size_t)(n * (n + 4) + per1) * 4, lds2 = (size_t)(n * (n + 4) + per2) * 4;
#define P1_(T, P_) hipLaunchKernelGGL((dct_pass1_kernel<T, P_>), grid, block, lds1, s, (const T *)x, wx, tmp, n, c)
#define P2_(T, P_) hipLaunchKernelGGL((dct_pass2_kernel<T, P_>), grid, block, lds2, s, (const float *)tmp, wy, (T *)high, (T *)low, n, c)
    TRAMBA_DISPATCH_DTYPE(dtype, T, {
        if (per1 <= 8) P1_(T, 8); else if (per1 <= 16) P1_(T, 16); else if (per1 <= 24) P1_(T, 24); else P1_(T, 32);
        if (per2 <= 8) P2_(T, 8); else if (per2 <= 16) P2_(T, 16); else P2_(T, 24);
    });
#undef P1_
#undef P2_
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
