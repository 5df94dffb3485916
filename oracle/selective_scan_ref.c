/* Oracle: selective-scan recurrence, plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates the operator behind the reference's L0 boundary
 *   selective_scan_cuda_oflex.fwd / .bwd   (call sites Models/SS2D/csms6s.py:910, :920-922)
 * whose source is NOT in the reference tree (third-party MzeroMiko/VMamba
 * kernels/selective_scan, version unpinned).  PARITY UNPINNED for this file: it follows
 * the published recurrence
 *     dt_l  = softplus(delta_l + delta_bias[d])      (threshold 20: x>20 -> x)
 *     h_l   = exp(dt_l * A[d,n]) * h_{l-1} + dt_l * B[b,k(d),n,l] * u_l,   h_0 = 0
 *     y_l   = sum_n C[b,k(d),n,l] * h_l[n] + D[d] * u_l
 *     k(d)  = d / (KD / K)
 * Layouts (row-major, contiguous):  u, delta, out, dout, du, ddelta: (B, KD, L);
 * A: (KD, N);  Bm, Cm, dB, dC: (B, K, N, L);  D, delta_bias, dD, dbias: (KD).
 *
 * All arithmetic in double; I/O in double (the Python wrapper converts).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double softplus20(double x) { return x > 20.0 ? x : log1p(exp(x)); }
static inline double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

void oracle_selective_scan_fwd(const double *u, const double *delta, const double *A,
                               const double *Bm, const double *Cm, const double *D,
                               const double *delta_bias, int softplus, int nb, int kd, int K,
                               int N, int L, double *out)
{
    const int dper = kd / K;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < nb; ++b) {
        for (int d = 0; d < kd; ++d) {
            const int k = d / dper;
            const double *ur = u + ((size_t)b * kd + d) * L;
            const double *dr = delta + ((size_t)b * kd + d) * L;
            double *yr = out + ((size_t)b * kd + d) * L;
            const double bias = delta_bias ? delta_bias[d] : 0.0;
            const double skip = D ? D[d] : 0.0;
            double h[256];
            for (int n = 0; n < N; ++n) h[n] = 0.0;
            for (int l = 0; l < L; ++l) {
                double dt = dr[l] + bias;
                if (softplus) dt = softplus20(dt);
                double y = 0.0;
                for (int n = 0; n < N; ++n) {
                    const size_t bc = (((size_t)b * K + k) * N + n) * L + l;
                    h[n] = exp(dt * A[(size_t)d * N + n]) * h[n] + dt * Bm[bc] * ur[l];
                    y += Cm[bc] * h[n];
                }
                yr[l] = y + skip * ur[l];
            }
        }
    }
}

/* Backward.  g_l[n] = dLoss/dh_l[n] obeys g_l = C_l*dout_l + a_{l+1} g_{l+1}.
 * Outputs must be zero-initialised by the caller for the reduced ones
 * (dA, dB, dC, dD, dbias); du, ddelta are fully overwritten.
 * Serial over (b,d) so the reductions are deterministic. */
void oracle_selective_scan_bwd(const double *u, const double *delta, const double *A,
                               const double *Bm, const double *Cm, const double *D,
                               const double *delta_bias, const double *dout, int softplus,
                               int nb, int kd, int K, int N, int L, double *du, double *ddelta,
                               double *dA, double *dB, double *dC, double *dD, double *dbias)
{
    const int dper = kd / K;
    double *hs = (double *)malloc(sizeof(double) * (size_t)(L + 1) * N);
    double *dts = (double *)malloc(sizeof(double) * (size_t)L);
    double *g = (double *)malloc(sizeof(double) * (size_t)N);
    for (int b = 0; b < nb; ++b) {
        for (int d = 0; d < kd; ++d) {
            const int k = d / dper;
            const size_t row = ((size_t)b * kd + d) * L;
            const double bias = delta_bias ? delta_bias[d] : 0.0;
            const double skip = D ? D[d] : 0.0;
            /* forward replay, keep every state */
            for (int n = 0; n < N; ++n) hs[n] = 0.0;
            for (int l = 0; l < L; ++l) {
                double dt = delta[row + l] + bias;
                if (softplus) dt = softplus20(dt);
                dts[l] = dt;
                for (int n = 0; n < N; ++n) {
                    const size_t bc = (((size_t)b * K + k) * N + n) * L + l;
                    hs[(size_t)(l + 1) * N + n] =
                        exp(dt * A[(size_t)d * N + n]) * hs[(size_t)l * N + n] +
                        dt * Bm[bc] * u[row + l];
                }
            }
            for (int n = 0; n < N; ++n) g[n] = 0.0;
            for (int l = L - 1; l >= 0; --l) {
                const double dt = dts[l];
                const double go = dout[row + l];
                double ddt = 0.0, dul = go * skip;
                if (dD) dD[d] += go * u[row + l];
                for (int n = 0; n < N; ++n) {
                    const size_t bc = (((size_t)b * K + k) * N + n) * L + l;
                    const double a = exp(dt * A[(size_t)d * N + n]);
                    const double hprev = hs[(size_t)l * N + n];
                    const double hcur = hs[(size_t)(l + 1) * N + n];
                    dC[bc] += go * hcur;
                    const double gn = g[n] + Cm[bc] * go; /* dLoss/dh_l[n] */
                    /* h_l = a*hprev + dt*B*u */
                    ddt += gn * (hprev * a * A[(size_t)d * N + n] + Bm[bc] * u[row + l]);
                    dA[(size_t)d * N + n] += gn * hprev * a * dt;
                    dB[bc] += gn * dt * u[row + l];
                    dul += gn * dt * Bm[bc];
                    g[n] = gn * a; /* contribution to h_{l-1} */
                }
                du[row + l] = dul;
                double draw = ddt;
                if (softplus) {
                    const double raw = delta[row + l] + bias;
                    draw = raw > 20.0 ? ddt : ddt * sigmoid(raw);
                }
                ddelta[row + l] = draw;
                if (dbias) dbias[d] += draw;
            }
        }
    }
    free(hs);
    free(dts);
    free(g);
}

/* float32-arithmetic forward (what a straightforward fp32 device kernel computes);
 * used only to size tolerances, never as truth. */
void oracle_selective_scan_fwd_f32(const float *u, const float *delta, const float *A,
                                   const float *Bm, const float *Cm, const float *D,
                                   const float *delta_bias, int softplus, int nb, int kd, int K,
                                   int N, int L, float *out)
{
    const int dper = kd / K;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < nb; ++b) {
        for (int d = 0; d < kd; ++d) {
            const int k = d / dper;
            const size_t row = ((size_t)b * kd + d) * L;
            const float bias = delta_bias ? delta_bias[d] : 0.f;
            const float skip = D ? D[d] : 0.f;
            float h[256];
            for (int n = 0; n < N; ++n) h[n] = 0.f;
            for (int l = 0; l < L; ++l) {
                float dt = delta[row + l] + bias;
                if (softplus) dt = dt > 20.f ? dt : log1pf(expf(dt));
                float y = 0.f;
                for (int n = 0; n < N; ++n) {
                    const size_t bc = (((size_t)b * K + k) * N + n) * L + l;
                    h[n] = expf(dt * A[(size_t)d * N + n]) * h[n] + dt * Bm[bc] * u[row + l];
                    y += Cm[bc] * h[n];
                }
                out[row + l] = y + skip * u[row + l];
            }
        }
    }
}
