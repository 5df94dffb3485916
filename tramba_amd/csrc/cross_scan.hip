// Scan-order gather / merge in the reference's NCHW layout (the L1 plugin boundary).
//
// Replaces CrossScan*/CrossMerge* forward and backward (Models/SS2D/csms6s.py:13-216) and the
// index_select / scatter_add_ loops of SpiralLine.py:85-133, Window.py:38-86,
// Dilation.py:48-96.  The merge is a GATHER through an inverse (CSR) table, so it is
// deterministic and atomic-free even for the many-to-one Helix line directions (the
// reference's scatter_add_ is order-nondeterministic on a GPU).
//
// These two kernels exist for drop-in parity with user-supplied scan classes; the model's
// own forward never materialises (B,K,D,L) in NCHW -- see ss2d_fused.hip.
#include "common.h"

namespace tramba {

template <typename T>
__global__ __launch_bounds__(256) void cross_scan_kernel(const T *__restrict__ x,
                                                        const int32_t *__restrict__ table,
                                                        T *__restrict__ xs, int C, int L, int K)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int bk = blockIdx.z;
    if (l >= L) return;
    const int b = bk / K, k = bk % K;
    const int p = table[(long)k * L + l];
    xs[(((long)b * K + k) * C + c) * L + l] = x[((long)b * C + c) * L + p];
}

template <typename T>
__global__ __launch_bounds__(256) void cross_merge_kernel(const T *__restrict__ ys,
                                                         const int32_t *__restrict__ inv_ptr,
                                                         const int32_t *__restrict__ inv_idx,
                                                         T *__restrict__ y, int C, int L, int K)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int b = blockIdx.z;
    if (p >= L) return;
    float acc = 0.f;
    const int e1 = inv_ptr[p + 1];
    for (int e = inv_ptr[p]; e < e1; ++e) {
        const int kl = inv_idx[e];
        const int k = kl / L, l = kl - k * L;
        acc += Cvt<T>::to_f(ys[(((long)b * K + k) * C + c) * L + l]);
    }
    y[((long)b * C + c) * L + p] = Cvt<T>::from_f(acc);
}

}  // namespace tramba

using namespace tramba;

extern "C" int tramba_cross_scan(const void *x, const int32_t *table, void *xs, int batch, int c,
                                 int l, int k, int dtype, void *stream)
{
    TRAMBA_CHECK(x && table && xs, "cross_scan: null tensor");
    TRAMBA_CHECK(batch > 0 && c > 0 && l > 0 && k > 0, "cross_scan: empty shape");
    TRAMBA_CHECK(c <= 65535 && (long)batch * k <= 65535, "cross_scan: C or B*K exceeds grid limits");
    dim3 grid((l + 255) / 256, c, batch * k), block(256);
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        hipLaunchKernelGGL(cross_scan_kernel<T>, grid, block, 0, (hipStream_t)stream, (const T *)x,
                           table, (T *)xs, c, l, k));
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}

extern "C" int tramba_cross_merge(const void *ys, const int32_t *inv_ptr, const int32_t *inv_idx,
                                  void *y, int batch, int c, int l, int k, int dtype, void *stream)
{
    TRAMBA_CHECK(ys && inv_ptr && inv_idx && y, "cross_merge: null tensor");
    TRAMBA_CHECK(batch > 0 && c > 0 && l > 0 && k > 0, "cross_merge: empty shape");
    TRAMBA_CHECK(c <= 65535 && batch <= 65535, "cross_merge: C or B exceeds grid limits");
    dim3 grid((l + 255) / 256, c, batch), block(256);
    TRAMBA_DISPATCH_DTYPE(dtype, T,
        hipLaunchKernelGGL(cross_merge_kernel<T>, grid, block, 0, (hipStream_t)stream, (const T *)ys,
                           inv_ptr, inv_idx, (T *)y, c, l, k));
    TRAMBA_LAUNCH_CHECK();
    return TRAMBA_OK;
}
