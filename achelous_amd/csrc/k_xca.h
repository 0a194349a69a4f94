// k_xca.h — EdgeNeXt cross-covariance attention (XCA, edgenext_modules/sdta_encoder.py:162-185).
//
// Attention is over CHANNELS: per (sample, head) a d x d matrix, d = C/heads in {8..72}; the token axis N (<= 1600) is only
// reduced over.  Both kernels are compiled for d <= 48 (EN-S0: 44, EN-S2 36 at 8 heads) and d <= 64 (EN-S1: 56) — the LDS tiles and the
// per-thread accumulator count are sized by the template bound.  Because the attention matrix multiplies v from the left and the projection from the right,
//     proj(attn @ v)[n, co] = sum_k v[n, k] * Weff[b][co][k],   Weff[b][co][h*d+j] = gamma[co] * sum_i A[b,h,i,j] * Wproj[co][h*d+i]
// the whole "attn @ v -> proj -> layer scale -> + residual" tail is ONE MFMA GEMM with per-sample weights.  So:
//   xca_gram    : partial Gram matrices q.k^T and squared norms over a slice of the tokens  (grid: B*heads x S, LDS staged)
//   xca_finalize: sum the partials, F.normalize over tokens folded in, temperature, row softmax, then write Weff of this
//                 (sample, head) straight into MFMA fragment order (k_gemm.h wfrag_offset)
// qkv comes from the MFMA GEMM with the LayerNorm prologue; v is a channel slice of that buffer.
#pragma once
#include "ach_platform.h"
#include "k_gemm.h"

namespace ach {

constexpr int XCA_CT = 32;       // output-channel tile of xca_finalize: one workgroup per (sample, head, tile) — the softmax is recomputed
                                 // per tile (d x d, cheap) so that the fold runs on 4-6x more workgroups instead of a serial loop

struct XcaGramParams { const void* qkv; long ld; float* partial; int B, N, C, heads, S; };

// partial layout per (b, h, s): [d*d gram | d sum q^2 | d sum k^2]
template <class T, int XCA_DMAX>
__global__ __launch_bounds__(256) void xca_gram_kernel(const XcaGramParams p) {
    constexpr int TOK = 16;
    __shared__ float qs[TOK][XCA_DMAX];
    __shared__ float ks[TOK][XCA_DMAX];
    const int bh = blockIdx.x, sp = blockIdx.y;
    const int b = bh / p.heads, h = bh % p.heads;
    const int d = p.C / p.heads;
    const int tid = threadIdx.x;
    const int per = (p.N + p.S - 1) / p.S;
    const int n_lo = sp * per, n_hi = (n_lo + per < p.N) ? n_lo + per : p.N;
    const T* base = static_cast<const T*>(p.qkv) + long(b) * p.N * p.ld;
    const int npair = d * d;
    constexpr int PP = (XCA_DMAX * XCA_DMAX + 255) / 256;
    float acc[PP];
    ACH_UNROLL
    for (int e = 0; e < PP; ++e) acc[e] = 0.f;
    float nacc = 0.f;
    for (int n0 = n_lo; n0 < n_hi; n0 += TOK) {
        for (int e = tid; e < TOK * d * 2; e += 256) {
            const int which = e / (TOK * d);
            const int r = e - which * TOK * d;
            const int t = r / d, i = r - t * d;
            const int n = n0 + t;
            const float v = n < n_hi ? Store<T>::ld(base + long(n) * p.ld + which * p.C + h * d + i) : 0.f;
            if (which == 0) qs[t][i] = v; else ks[t][i] = v;
        }
        __syncthreads();
        ACH_UNROLL
        for (int e = 0; e < PP; ++e) {
            const int pr = tid + e * 256;
            if (pr < npair) {
                const int i = pr / d, j = pr - i * d;
                float s = acc[e];
                ACH_UNROLL
                for (int t = 0; t < TOK; ++t) s += qs[t][i] * ks[t][j];
                acc[e] = s;
            }
        }
        if (tid < 2 * d) {
            const int i = tid < d ? tid : tid - d;
            ACH_UNROLL
            for (int t = 0; t < TOK; ++t) { const float v = tid < d ? qs[t][i] : ks[t][i]; nacc += v * v; }
        }
        __syncthreads();
    }
    float* out = p.partial + (long(bh) * p.S + sp) * (npair + 2 * d);
    ACH_UNROLL
    for (int e = 0; e < PP; ++e) {
        const int pr = tid + e * 256;
        if (pr < npair) out[pr] = acc[e];
    }
    if (tid < 2 * d) out[npair + tid] = nacc;
}

struct XcaFinalParams {
    const float* partial; int S;
    const float* temperature;     // [heads]
    const float* Wproj;           // [C][C] fp32
    const float* gamma;           // [C] layer scale
    float* attn;                  // optional [B][heads][d][d] (tap) or nullptr
    void* Weff; long group_stride;   // packed per-sample weights (elements of T)
    int B, C, heads, NT, ksteps;
};

template <class T, int XCA_DMAX>
__global__ __launch_bounds__(256) void xca_finalize_kernel(const XcaFinalParams p) {
    __shared__ float A[XCA_DMAX][XCA_DMAX + 1];
    __shared__ float nq[XCA_DMAX], nk[XCA_DMAX];
    __shared__ float wps[XCA_CT][XCA_DMAX + 1];
    const int bh = blockIdx.x;
    const int b = bh / p.heads, h = bh % p.heads;
    const int d = p.C / p.heads, npair = d * d;
    const int tid = threadIdx.x;
    const float* part = p.partial + long(bh) * p.S * (npair + 2 * d);
    for (int e = tid; e < npair + 2 * d; e += 256) {
        float s = 0.f;
        for (int sp = 0; sp < p.S; ++sp) s += part[long(sp) * (npair + 2 * d) + e];
        if (e < npair) A[e / d][e % d] = s;
        else if (e < npair + d) nq[e - npair] = fmaxf(sqrtf(s), 1e-12f);
        else nk[e - npair - d] = fmaxf(sqrtf(s), 1e-12f);
    }
    __syncthreads();
    const float temp = p.temperature[h];
    for (int e = tid; e < npair; e += 256) { const int i = e / d, j = e % d; A[i][j] = A[i][j] / (nq[i] * nk[j]) * temp; }
    __syncthreads();
    if (tid < d) {
        float mx = -3.0e38f;
        for (int j = 0; j < d; ++j) mx = fmaxf(mx, A[tid][j]);
        float sum = 0.f;
        for (int j = 0; j < d; ++j) { const float e = expf(A[tid][j] - mx); A[tid][j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < d; ++j) A[tid][j] *= inv;
    }
    __syncthreads();
    if (p.attn && blockIdx.y == 0) for (int e = tid; e < npair; e += 256) p.attn[long(bh) * npair + e] = A[e / d][e % d];
    // Weff rows in tiles of XCA_CT output channels: the Wproj slice [tile][d] of this head is staged in LDS with coalesced loads
    // first (a thread-private walk over Wproj rows is a chain of dependent L1/L2 latencies: 67 us for a 2 us job).
    T* W = static_cast<T*>(p.Weff) + long(b) * p.group_stride;
    {
        const int c0 = blockIdx.y * XCA_CT;
        const int rows = (p.C - c0 < XCA_CT) ? p.C - c0 : XCA_CT;
        __syncthreads();
        for (int e = tid; e < rows * d; e += 256) { const int r = e / d, i = e - r * d; wps[r][i] = p.Wproj[long(c0 + r) * p.C + h * d + i]; }
        __syncthreads();
        for (int e = tid; e < rows * d; e += 256) {
            const int r = e / d, j = e - r * d;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // 4 independent chains: LDS reads of a group are all in flight
            int i = 0;
            for (; i + 4 <= d; i += 4) {
                s0 += A[i][j] * wps[r][i]; s1 += A[i + 1][j] * wps[r][i + 1]; s2 += A[i + 2][j] * wps[r][i + 2]; s3 += A[i + 3][j] * wps[r][i + 3];
            }
            for (; i < d; ++i) s0 += A[i][j] * wps[r][i];
            const float s = (s0 + s1) + (s2 + s3);
            const int co = c0 + r;
            Store<T>::st(W + wfrag_offset(co, h * d + j, p.NT, p.ksteps, Store<T>::VEC), s * p.gamma[co]);
        }
    }
}

}  // namespace ach
