// Shared device/host helpers for libtramba_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tramba_hip.h"

namespace tramba {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

#define TRAMBA_CHECK(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::tramba::set_error(__VA_ARGS__);   \
            return TRAMBA_ERR_ARG;              \
        }                                       \
    } while (0)

// Device error word: ONE 32-bit word in host-mapped (coherent) memory that a kernel raises on a failure it cannot report any
// other way -- today the carry mailbox of the fused scans running out of polls (ss2d_fused.hip, CarryLink::acquire).  The host
// never synchronises for it: every entry point reads the word after its launch (TRAMBA_LAUNCH_CHECK) and, when it is set,
// clears it, sets tramba_last_error() and returns TRAMBA_ERR_HIP -- so a poisoned (NaN) result is reported by the next library
// call that runs after the failing kernel, or by tramba_device_error() after the caller's own synchronisation.
#define TRAMBA_DEVERR_MAILBOX 1u
extern unsigned *g_deverr_host;                 // null until the first kernel that can raise it is launched (runtime.hip)
int dev_error_report(const char *file, int line);
struct MailboxCtl {
    int skip;        // tests: the tile (chain order) whose carry hand-over is withheld, so that its successor's poll times out; -1
};
MailboxCtl mailbox_ctl(hipStream_t s);              // (also allocates the error word before the first launch that can raise it)
void set_mailbox_err_word(unsigned *dev_ptr);        // ss2d_fused.hip: the word's device address into that file's __device__ variable

#define TRAMBA_LAUNCH_CHECK()                                                         \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            ::tramba::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,        \
                                hipGetErrorString(e_));                               \
            return TRAMBA_ERR_HIP;                                                    \
        }                                                                             \
        if (::tramba::g_deverr_host && *(volatile unsigned *)::tramba::g_deverr_host) \
            return ::tramba::dev_error_report(__FILE__, __LINE__);                    \
    } while (0)

// ---- profiling hooks (HIP events around one kernel class; see tramba_profile_enable) ----
struct ProfScope {
    int which;
    hipStream_t stream;
    bool on;
    hipEvent_t start;
    ProfScope(int which, hipStream_t s, double units);
    ~ProfScope();
};

// ---- XCD-aware work order ----
// Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MiB L2) in linear block-id order.  This maps the
// block to the work item (i0 fastest, then i1, then i2; extents = the grid's) such that every XCD takes one CONTIGUOUS
// run of the item list, in order: neighbouring items -- which share cache lines, halo rows or operand panels -- then
// meet in one L2 instead of being fetched by several.
__device__ __forceinline__ void xcd_work_item(unsigned &i0, unsigned &i1, unsigned &i2)
{
    const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned q = nblk >> 3, r = nblk & 7, xcd = lin & 7, slot = lin >> 3;
    const unsigned t = xcd * q + (xcd < r ? xcd : r) + slot;
    i0 = t % gridDim.x;
    const unsigned u = t / gridDim.x;
    i1 = u % gridDim.y;
    i2 = u / gridDim.y;
}

// ---- dtype traits ----
template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<__half> {
    static __device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
    static __device__ __forceinline__ __half from_f(float v) { return __float2half(v); }
};
template <> struct Cvt<__hip_bfloat16> {
    static __device__ __forceinline__ float to_f(__hip_bfloat16 v) { return __bfloat162float(v); }
    static __device__ __forceinline__ __hip_bfloat16 from_f(float v) { return __float2bfloat16(v); }
};

// A pack of V elements of T moved with one (8/16-byte where possible) access.
template <typename T, int V> struct alignas(sizeof(T) * V > 16 ? 16 : sizeof(T) * V) Pack {
    T v[V];
};

template <typename T, int V>
__device__ __forceinline__ void load_pack(const T *p, float (&out)[V])
{
    Pack<T, V> pk = *reinterpret_cast<const Pack<T, V> *>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = Cvt<T>::to_f(pk.v[i]);
}

template <typename T, int V>
__device__ __forceinline__ void store_pack(T *p, const float (&in)[V])
{
    Pack<T, V> pk;
#pragma unroll
    for (int i = 0; i < V; ++i) pk.v[i] = Cvt<T>::from_f(in[i]);
    *reinterpret_cast<Pack<T, V> *>(p) = pk;
}

// ---- buffer (SRD) addressing: wave-uniform descriptor + 32-bit per-lane byte offset.  Accesses at or beyond
// `bytes` are dropped (loads return 0) by the hardware range check.  Build descriptors from wave-uniform
// values only (kernel arguments, blockIdx, readfirstlane results).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
constexpr unsigned kOutOfRange = 0x80000000u;  // a vector offset >= any num_records the host admits

// ---- math ----
// softplus with the reference's threshold (x > 20 -> x).  log1p via the w = 1+z trick keeps
// full relative precision for small exp(x) while using the hardware exp/log.
__device__ __forceinline__ float softplus20(float x)
{
    // branch-free: evaluate log1p(exp(min(x,20))) and select x beyond the threshold
    const float z = __expf(fminf(x, 20.f));
    const float w = 1.f + z;
    const float d = w - 1.f;
    const float sp = d == 0.f ? z : __logf(w) * z * __builtin_amdgcn_rcpf(d);
    return x > 20.f ? x : sp;
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup, ~10 instructions): the
// SiLU of every depth-wise 3x3 and the FreqSS2D gate evaluate this once per output element
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32-level): 1 rcp + 1 exp + 5 fma
// instead of libm erff's ~40 instructions -- the exact (erf) GELU sits in the epilogue of the widest GEMMs.
__device__ __forceinline__ float erf_as(float x)
{
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896f);
    const float r = 1.f - p * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float geluf_(float x)
{
    return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752440f));
}
// d/dx of the exact GELU: Phi(x) + x phi(x), with the same erf as the forward
__device__ __forceinline__ float gelu_gradf_(float x)
{
    const float cdf = 0.5f * (1.f + erf_as(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.5f * x * x * 1.44269504088896f);
    return fmaf(x, pdf, cdf);
}
// d/dx of SiLU: s (1 + x (1 - s)), s = sigmoid(x)
__device__ __forceinline__ float silu_gradf_(float x)
{
    const float sg = sigmoidf_(x);
    return sg * fmaf(x, 1.f - sg, 1.f);
}
// how a GEMM epilogue combines its activated value with the `residual` operand
__device__ __forceinline__ float combine_residual(float o, float r, int act)
{
    if (act == TRAMBA_ACT_SIGMOID_GATE) return o * r;
    if (act == TRAMBA_ACT_GELU_GRAD_MUL) return o * gelu_gradf_(r);
    return o + r;
}
__device__ __forceinline__ float apply_act(float x, int act)
{
    if (act == TRAMBA_ACT_SILU) return siluf_(x);
    if (act == TRAMBA_ACT_GELU) return geluf_(x);
    if (act == TRAMBA_ACT_SIGMOID_GATE) return sigmoidf_(x);   // the caller multiplies by the gated tensor
    return x;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// dispatch a runtime dtype to a template
#define TRAMBA_DISPATCH_DTYPE(dt, T, ...)                                   \
    switch (dt) {                                                           \
    case TRAMBA_F32: { using T = float; __VA_ARGS__; } break;               \
    case TRAMBA_F16: { using T = __half; __VA_ARGS__; } break;              \
    case TRAMBA_BF16: { using T = __hip_bfloat16; __VA_ARGS__; } break;     \
    default: ::tramba::set_error("bad dtype %d", (int)(dt)); return TRAMBA_ERR_ARG; \
    }

inline size_t dtype_size(int dt) { return dt == TRAMBA_F32 ? 4 : 2; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace tramba
