// k_xca.h — EdgeNeXt cross-covariance attention (XCA, edgenext_modules/sdta_encoder.py:162-185).
//
// Attention is over CHANNELS: per (sample, head) a d x d matrix, d = C/heads in {8..44}; the token axis N
// (<= 1600) is only reduced over.  Two small kernels:
//   xca_attn : A[b,h] = softmax_j( (q_i . k_j) / (max(|q_i|,eps) max(|k_j|,eps)) * temperature_h )
//              (F.normalize over the N tokens folded into the Gram matrix)
//   xca_apply: out[b,n,h*d+i] = sum_j A[b,h,i,j] * v[b,n,h*d+j]
// qkv comes from the MFMA GEMM (LayerNorm fused as its prologue); the projection (+ layer scale + residual)
// is another MFMA GEMM.
#pragma once
#include "ach_platform.h"

namespace ach {

struct XcaAttnParams { const void* qkv; long ld; float* attn; const float* temperature; int B, N, C, heads; };

template <class T>
__global__ __launch_bounds__(256) void xca_attn_kernel(const XcaAttnParams p) {
    constexpr int TOK = 16;                 // tokens staged per step
    constexpr int DMAX = 48;
    __shared__ float qs[TOK][DMAX];
    __shared__ float ks[TOK][DMAX];
    __shared__ float gram[DMAX][DMAX + 1];
    __shared__ float nq[DMAX], nk[DMAX];
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int d = p.C / p.heads;
    const int tid = threadIdx.x;
    const T* base = static_cast<const T*>(p.qkv) + long(b) * p.N * p.ld;
    const int npair = d * d;
    constexpr int PP = (DMAX * DMAX + 255) / 256;      // (i,j) pairs per thread
    float acc[PP];
    ACH_UNROLL
    for (int e = 0; e < PP; ++e) acc[e] = 0.f;
    float nacc = 0.f;                                   // threads < d: |q_i|^2 ; d <= threads < 2d: |k_j|^2
    for (int n0 = 0; n0 < p.N; n0 += TOK) {
        for (int e = tid; e < TOK * d * 2; e += 256) {
            const int which = e / (TOK * d);
            const int r = e - which * TOK * d;
            const int t = r / d, i = r - t * d;
            const int n = n0 + t;
            const float v = n < p.N ? Store<T>::ld(base + long(n) * p.ld + which * p.C + h * d + i) : 0.f;
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
    if (tid < d) nq[tid] = fmaxf(sqrtf(nacc), 1e-12f);
    else if (tid < 2 * d) nk[tid - d] = fmaxf(sqrtf(nacc), 1e-12f);
    __syncthreads();
    const float temp = p.temperature[h];
    ACH_UNROLL
    for (int e = 0; e < PP; ++e) {
        const int pr = tid + e * 256;
        if (pr < npair) { const int i = pr / d, j = pr - i * d; gram[i][j] = acc[e] / (nq[i] * nk[j]) * temp; }
    }
    __syncthreads();
    if (tid < d) {                                       // row softmax
        float mx = -3.0e38f;
        for (int j = 0; j < d; ++j) mx = fmaxf(mx, gram[tid][j]);
        float sum = 0.f;
        for (int j = 0; j < d; ++j) { const float e = expf(gram[tid][j] - mx); gram[tid][j] = e; sum += e; }
        const float inv = 1.0f / sum;
        float* out = p.attn + ((long(b) * p.heads + h) * d + tid) * d;
        for (int j = 0; j < d; ++j) out[j] = gram[tid][j] * inv;
    }
}

struct XcaApplyParams { const void* qkv; long ld; const float* attn; void* Y; long ldy; int B, N, C, heads; };
template <class T>
__global__ __launch_bounds__(256) void xca_apply_kernel(const XcaApplyParams p) {
    const long total = long(p.B) * p.N * p.C;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = int(idx % p.C);
    const long row = idx / p.C;               // b*N + n
    const long b = row / p.N;
    const int d = p.C / p.heads, h = c / d, i = c - h * d;
    const T* v = static_cast<const T*>(p.qkv) + row * p.ld + 2 * p.C + h * d;
    const float* a = p.attn + ((b * p.heads + h) * d + i) * d;
    float s = 0.f;
    for (int j = 0; j < d; ++j) s += a[j] * Store<T>::ld(v + j);
    Store<T>::st(static_cast<T*>(p.Y) + row * p.ldy + c, s);
}

}  // namespace ach
