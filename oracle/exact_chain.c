/* ORACLE — test infrastructure only; never linked or loaded by the product path.
 *
 * Exact-order fp32 restatement of the encoder -> bottleneck-linear -> VQ sub-path, the part
 * of the reference whose *integer* result (the code indices, vqema_bn.py:141) must be
 * reproduced bit-exactly.  The reference leaves the summation order to ATen; this file
 * fixes one order, documents it, and the HIP kernels are written to the same order:
 *
 *   conv  : acc = 0; for tap in 0..f-1: for ci in 0..Cin-1:
 *               acc = fmaf(x[t*stride+tap][ci], W[co][ci][tap], acc)
 *           y = relu(acc + bias[co]) (+ x[t+lw][co] if residual) (wave_encoder.py:39-43)
 *           ksplit = S in {2, 4} (round 5; the product's aew_gemm_nt_t.k_split): the k axis (k = tap * Cin + ci, f * Cin a
 *           multiple of 32 * S) is cut into S contiguous ranges, each its own ascending fmaf chain from 0, and
 *               acc = p0 + p1  (S = 2)        acc = (p0 + p1) + (p2 + p3)  (S = 4)       (plain fp32 adds)
 *           - the canonical order of every encoder layer and of the bottleneck's linear map whose k axis divides
 *           that way (oracle/exact.py: ksplit_for), so that the MI355X kernel can run the ranges on separate workgroups
 *   dist  : dd = fma-chain_j (z_j-q_j)^2 ; zz = fma-chain_j z_j^2 ; qq = fma-chain_j q_j^2
 *           scaled_l2 = sqrtf(dd) / (sqrtf(zz) + sqrtf(qq))    (vqema_bn.py:67-76)
 *           sq_l2     = dd                                      (vq_bn.py:39)
 *           argmin over k ascending, strict '<' (first minimum wins, torch.min)
 *   stats : z_sum[k] += ze[q] in ascending query order; n_sum[k] = count (vqema_bn.py:172-188)
 *   ema   : numer = g*numer + c*z_sum (three roundings, no contraction), g=(float)gamma,
 *           c=(float)(1.0-gamma)                                (vqema_bn.py:190-195)
 *   emb   : numer / denom                                       (vqema_bn.py:216-222)
 *
 * All tensors channels-last: x[b][t][c].  Weights in the reference's layout W[co][ci][tap].
 * Build: see oracle/Makefile (-ffp-contract=off so only the explicit fmaf()s fuse).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int aewo_conv_cl_split(const float* x, int B, int L, int Cin, const float* W, const float* bias,
                       int Cout, int f, int stride, int relu, int res_lw, float* y, int ksplit);

int aewo_conv_cl(const float* x, int B, int L, int Cin, const float* W, const float* bias,
                 int Cout, int f, int stride, int relu, int res_lw, float* y)
{
    return aewo_conv_cl_split(x, B, L, Cin, W, bias, Cout, f, stride, relu, res_lw, y, 1);
}

int aewo_conv_cl_split(const float* x, int B, int L, int Cin, const float* W, const float* bias,
                       int Cout, int f, int stride, int relu, int res_lw, float* y, int ksplit)
{
    const int Lout = (L - f) / stride + 1;
    if (Lout <= 0) return 1;
    if (ksplit != 1 && ksplit != 2 && ksplit != 4) return 3;
    if (ksplit > 1 && (f * Cin) % (32 * ksplit)) return 4;
    const int krange = f * Cin / ksplit;                     /* k = tap * Cin + ci */
    /* re-lay weights as Wt[tap][ci][co] so the inner loop over co is contiguous */
    float* Wt = (float*)malloc(sizeof(float) * (size_t)f * Cin * Cout);
    if (!Wt) return 2;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < f; ++k)
                Wt[((size_t)k * Cin + ci) * Cout + co] = W[((size_t)co * Cin + ci) * f + k];
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < Lout; ++t) {
            float* acc = y + ((size_t)b * Lout + t) * Cout;
            if (ksplit == 1) {
                for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
                for (int k = 0; k < f; ++k) {
                    const float* xr = x + ((size_t)b * L + (size_t)t * stride + k) * Cin;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float xv = xr[ci];
                        const float* wr = Wt + ((size_t)k * Cin + ci) * Cout;
                        for (int co = 0; co < Cout; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                    }
                }
            } else {
                float* part = (float*)malloc(sizeof(float) * (size_t)ksplit * Cout);
                for (int sp = 0; sp < ksplit; ++sp) {
                    float* p = part + (size_t)sp * Cout;
                    for (int co = 0; co < Cout; ++co) p[co] = 0.0f;
                    for (int kk = sp * krange; kk < (sp + 1) * krange; ++kk) {
                        const int k = kk / Cin, ci = kk - k * Cin;
                        const float xv = x[((size_t)b * L + (size_t)t * stride + k) * Cin + ci];
                        const float* wr = Wt + (size_t)kk * Cout;
                        for (int co = 0; co < Cout; ++co) p[co] = fmaf(xv, wr[co], p[co]);
                    }
                }
                for (int co = 0; co < Cout; ++co)
                    acc[co] = ksplit == 2 ? part[co] + part[Cout + co]
                                          : (part[co] + part[Cout + co]) + (part[2 * Cout + co] + part[3 * Cout + co]);
                free(part);
            }
            if (bias)
                for (int co = 0; co < Cout; ++co) acc[co] = acc[co] + bias[co];
            if (relu)
                for (int co = 0; co < Cout; ++co) acc[co] = acc[co] > 0.0f ? acc[co] : 0.0f;
            if (res_lw >= 0) {
                const float* xr = x + ((size_t)b * L + t + res_lw) * Cin;   /* Cin == Cout */
                for (int co = 0; co < Cout; ++co) acc[co] = acc[co] + xr[co];
            }
        }
    free(Wt);
    return 0;
}

static float chain_sq(const float* a, int d)
{
    float s = 0.0f;
    for (int j = 0; j < d; ++j) s = fmaf(a[j], a[j], s);
    return s;
}

/* metric: 0 = scaled_l2 (VQEMA), 1 = sq_l2 (VQ) */
int aewo_vq_nearest(const float* ze, const float* emb, int Q, int K, int d, int metric,
                    int64_t* ind, float* dist, float* second)
{
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        const float* z = ze + (size_t)q * d;
        const float zn = sqrtf(chain_sq(z, d));
        float best = INFINITY, sec = INFINITY;
        int64_t bi = 0;
        for (int k = 0; k < K; ++k) {
            const float* c = emb + (size_t)k * d;
            float dd = 0.0f;
            for (int j = 0; j < d; ++j) {
                const float t = z[j] - c[j];
                dd = fmaf(t, t, dd);
            }
            float v;
            if (metric == 0) v = sqrtf(dd) / (zn + sqrtf(chain_sq(c, d)));
            else v = dd;
            if (v < best) { sec = best; best = v; bi = k; }
            else if (v < sec) sec = v;
        }
        ind[q] = bi;
        dist[q] = best;
        if (second) second[q] = sec;
    }
    return 0;
}

int aewo_vq_stats(const float* ze, const int64_t* ind, int Q, int K, int d,
                  float* z_sum, float* n_sum)
{
    memset(z_sum, 0, sizeof(float) * (size_t)K * d);
    memset(n_sum, 0, sizeof(float) * (size_t)K);
    for (int q = 0; q < Q; ++q) {
        const int64_t k = ind[q];
        if (k < 0 || k >= K) return 1;
        for (int j = 0; j < d; ++j) z_sum[k * d + j] = z_sum[k * d + j] + ze[(size_t)q * d + j];
        n_sum[k] = n_sum[k] + 1.0f;
    }
    return 0;
}

int aewo_ema(float* numer, float* denom, const float* z_sum, const float* n_sum, int K, int d,
             double gamma)
{
    const float g = (float)gamma, c = (float)(1.0 - gamma);
    for (int k = 0; k < K; ++k) {
        for (int j = 0; j < d; ++j) {
            const float a = g * numer[k * d + j];
            const float b = c * z_sum[k * d + j];
            numer[k * d + j] = a + b;
        }
        const float a = g * denom[k];
        const float b = c * n_sum[k];
        denom[k] = a + b;
    }
    return 0;
}

int aewo_codebook(const float* numer, const float* denom, float* emb, int K, int d)
{
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < d; ++j) emb[k * d + j] = numer[k * d + j] / denom[k];
    return 0;
}
