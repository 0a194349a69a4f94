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

#ifndef ACH_XCA_CT
#define ACH_XCA_CT 32
#endif
constexpr int XCA_CT = ACH_XCA_CT;       // output-channel tile of xca_finalize: one workgroup per (sample, head, tile) — the softmax is recomputed
                                 // per tile (d x d, cheap) so that the fold runs on 4-6x more workgroups instead of a serial loop

struct XcaGramParams { const void* qkv; long ld; float* partial; int B, N, C, heads, S; int hg; int per = 0; };   // hg: heads per workgroup (xca_gram_mfma_kernel); per > 0: tokens per slice (else ceil(N / S))

// partial layout per (b, h, s): [d*d gram | d sum q^2 | d sum k^2]
template <class T, int XCA_DMAX>
__global__ __launch_bounds__(256) void xca_gram_kernel(const XcaGramParams p) { f16_sat_mode<T>();
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

// The same partial sums on the matrix cores (option `xca_mfma`, default): gram = Q^T K is an MFMA with the TOKENS as the k index, so both
// operands are token-contiguous fragments of one channel — the transpose of how qkv is stored.  A workgroup takes a GROUP of `hg`
// consecutive heads (hg d <= XCA_DMAX channels: 8 heads of 8, 4 of 12, 2 of 24, one of 36 / 44 / 56) so that a token row contributes one
// contiguous run of hg d channels, and a round stages 4 k-steps of tokens (128 in bf16, 64 in fp32) of q and k TRANSPOSED into LDS
// ([channel][token], 4-byte / 8-byte global loads of channel pairs); every fragment is then one 16-byte LDS read.  Work items, dealt to the
// four waves: per head the ceil(d/16)^2 gram tiles plus, for the squared norms, the 2 ceil(d/16) diagonal tiles of Q^T Q and K^T K (only
// their diagonals are stored).  A head's tiles start at its first channel, not at a multiple of 16; rows that belong to the next head
// produce entries that are simply not stored.  Same output layout as xca_gram_kernel; the sums differ from it only in the order of the
// fp32 additions.  (One head per workgroup was slower than the VALU kernel for d <= 18: 16- / 24-byte pieces of every token row.)
// register budget of the MFMA Gram kernel: left to itself the compiler takes 188 VGPRs (two workgroups per CU) for a kernel that fits 97 without a
// spill; at four waves per SIMD the three Gram launches go 20 / 16 / 20 -> 16 / 13 / 19 us and the step gains 1.6 % (round 3, occupancy table of DESIGN 7)
#ifndef ACH_XCA_WAVES
#define ACH_XCA_WAVES 4
#endif
#if ACH_XCA_WAVES > 0
#define ACH_XCA_BOUNDS __launch_bounds__(256, ACH_XCA_WAVES)
#else
#define ACH_XCA_BOUNDS __launch_bounds__(256)
#endif
// The body for one (sample b, head group grp, token slice sp) on a workgroup of NWV waves; `tsf` = the workgroup's LDS staging tile [2][ROWS][PITCH] of T.
// (xca_gram_mfma_kernel: NWV = 4, one call per workgroup; xca_frame_kernel, k_xcaframe.h: NWV = 16, one call per head group of the frame.)
template <class T, int XCA_DMAX> struct XcaGramTile {
    static constexpr int VEC = Store<T>::VEC, KC = 4 * VEC;       // tokens per MFMA k-step
    static constexpr int CH = 4 * KC;                             // tokens staged per round
    static constexpr int PITCH = CH + VEC;                        // + 16 bytes: rows land on different banks
    static constexpr int ROWS = XCA_DMAX + 16;                    // a head's last tile may reach 15 rows past the group's channels (zeros)
    static constexpr int ELEMS = 2 * ROWS * PITCH;
};
template <class T, int XCA_DMAX, int NWV>
__device__ __forceinline__ void xca_gram_mfma_body(const XcaGramParams& p, T* tsf, int b, int grp, int sp) {
    using G = XcaGramTile<T, XCA_DMAX>;
    constexpr int VEC = G::VEC, KC = G::KC, CH = G::CH, PITCH = G::PITCH, ROWS = G::ROWS;
    constexpr int MAXIT = (24 + NWV - 1) / NWV;            // items per wave: 8 heads x 3, 3 heads x 8, one head of 4 x 4 + 8 -> 24 in all
    constexpr int NTH = 64 * NWV;
    auto ts = [&](int which, int row, int t) -> T& { return tsf[(which * ROWS + row) * PITCH + t]; };
    const int h0 = grp * p.hg, nh = (h0 + p.hg <= p.heads) ? p.hg : p.heads - h0;
    const int d = p.C / p.heads, gc2 = (nh * d) >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int per = p.per > 0 ? p.per : (p.N + p.S - 1) / p.S;
    const int n_lo = sp * per, n_hi = (n_lo + per < p.N) ? n_lo + per : p.N;
    const T* base = static_cast<const T*>(p.qkv) + long(b) * p.N * p.ld + h0 * d;
    const int tm = (d + 15) >> 4, ngram = tm * tm, per_head = ngram + 2 * tm, nitems = nh * per_head;
    {   // rows past the group's channels stay zero (16-byte stores; the staged rows are rewritten every round)
        const int r0 = nh * d, nv = (ROWS - r0) * (PITCH / VEC);
        for (int e = tid; e < 2 * nv; e += NTH) {
            const int which = e >= nv, k = e - which * nv;
            reinterpret_cast<uint4*>(&ts(which, r0, 0))[k] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // a token row's q (or k) channels of this group as 16-byte pieces when everything is 16-byte aligned (12 x 4, 24 x 2, 8 x 8 channels),
    // else as channel pairs
    const bool wide = ((h0 * d) % VEC == 0) && ((nh * d) % VEC == 0) && (p.C % VEC == 0) && (p.ld % VEC == 0);
    const int gv = (nh * d) / VEC;
    f32x4 acc[MAXIT];
    ACH_UNROLL
    for (int it = 0; it < MAXIT; ++it) { acc[it][0] = 0.f; acc[it][1] = 0.f; acc[it][2] = 0.f; acc[it][3] = 0.f; }
    // this wave's items: (head, operand arrays, first rows) are the same in every round
    int rowa[MAXIT], rowb[MAXIT];                           // row index + which * ROWS
    ACH_UNROLL
    for (int it = 0; it < MAXIT; ++it) {
        const int item = wave + NWV * it;
        rowa[it] = rowb[it] = 0;
        if (item >= nitems) continue;
        const int hh = item / per_head, k = item - hh * per_head;
        if (k < ngram) { const int ti = k / tm, tj = k - ti * tm; rowa[it] = hh * d + ti * 16; rowb[it] = ROWS + hh * d + tj * 16; }
        else { const int w = (k - ngram) / tm, t = (k - ngram) - w * tm; rowa[it] = rowb[it] = w * ROWS + hh * d + t * 16; }
    }
    for (int n0 = n_lo; n0 < n_hi; n0 += CH) {
        __syncthreads();
        if (wide) {
            for (int e = tid; e < 2 * CH * gv; e += NTH) {
                const int which = e / (CH * gv), r = e - which * CH * gv;
                const int t = r / gv, cv = r - t * gv, n = n0 + t;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (n < n_hi) v = *reinterpret_cast<const uint4*>(base + long(n) * p.ld + which * p.C + cv * VEC);
                T el[VEC];
                __builtin_memcpy(el, &v, sizeof(v));
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) ts(which, cv * VEC + i, t) = el[i];
            }
        } else
        for (int e = tid; e < 2 * CH * gc2; e += NTH) {
            const int which = e / (CH * gc2), r = e - which * CH * gc2;
            const int t = r / gc2, cp = r - t * gc2, n = n0 + t;
            T v0 = T{}, v1 = v0;
            if (n < n_hi) { const T* src = base + long(n) * p.ld + which * p.C + 2 * cp; v0 = src[0]; v1 = src[1]; }
            ts(which, 2 * cp, t) = v0; ts(which, 2 * cp + 1, t) = v1;
        }
        __syncthreads();
        ACH_UNROLL
        for (int ks = 0; ks < 4; ++ks) {
            if (n0 + ks * KC >= n_hi) continue;
            ACH_UNROLL
            for (int it = 0; it < MAXIT; ++it) {
                if (wave + NWV * it >= nitems) continue;
                const uint4 fa = *reinterpret_cast<const uint4*>(tsf + (rowa[it] + col) * PITCH + ks * KC + g * VEC);
                const uint4 fb = *reinterpret_cast<const uint4*>(tsf + (rowb[it] + col) * PITCH + ks * KC + g * VEC);
                mfma16<T>(fa, fb, acc[it]);
            }
        }
    }
    ACH_UNROLL
    for (int it = 0; it < MAXIT; ++it) {
        const int item = wave + NWV * it;
        if (item >= nitems) continue;
        const int hh = item / per_head, k = item - hh * per_head;
        float* out = p.partial + ((long(b) * p.heads + h0 + hh) * p.S + sp) * (d * d + 2 * d);
        if (k < ngram) {                                        // acc[r] = D[4 g + r][col] of the tile
            const int ti = k / tm, tj = k - ti * tm, j = tj * 16 + col;
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { const int i = ti * 16 + 4 * g + r; if (i < d && j < d) out[i * d + j] = acc[it][r]; }
        } else {
            const int w = (k - ngram) / tm, t = (k - ngram) - w * tm, i = t * 16 + col;
            const int rr = col & 3;
            const float dv = rr == 0 ? acc[it][0] : (rr == 1 ? acc[it][1] : (rr == 2 ? acc[it][2] : acc[it][3]));
            if ((col >> 2) == g && i < d) out[d * d + w * d + i] = dv;
        }
    }
}
template <class T, int XCA_DMAX>
__global__ ACH_XCA_BOUNDS void xca_gram_mfma_kernel(const XcaGramParams p) { f16_sat_mode<T>();
    __shared__ __attribute__((aligned(16))) T ts[XcaGramTile<T, XCA_DMAX>::ELEMS];
    const int ngroups = (p.heads + p.hg - 1) / p.hg;
    const int b = blockIdx.x / ngroups, grp = blockIdx.x - b * ngroups;
    xca_gram_mfma_body<T, XCA_DMAX, 4>(p, ts, b, grp, int(blockIdx.y));
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
__global__ __launch_bounds__(256) void xca_finalize_kernel(const XcaFinalParams p) { f16_sat_mode<T>();
    __shared__ __attribute__((aligned(16))) float A[XCA_DMAX][XCA_DMAX + 4];     // rows 16-byte aligned: the fold reads four columns at once
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
    for (int e = tid; e < d * 4; e += 256) A[e >> 2][d + (e & 3)] = 0.f;       // the fold's last column group reads up to three columns past d
    __syncthreads();
    const float temp = p.temperature[h];
    for (int e = tid; e < npair; e += 256) { const int i = e / d, j = e % d; A[i][j] = A[i][j] / (nq[i] * nk[j]) * temp; }
    __syncthreads();
    {   // row softmax, four lanes per row (d <= 64 rows: all 256 threads): each lane takes every fourth column, the row maximum and the
        // row sum are combined over the four lanes with two butterfly steps (one thread per row was a chain of 3 d dependent LDS round trips)
        const int i = tid >> 2, q = tid & 3;
        const bool row = i < d;
        float mx = -3.0e38f;
        if (row) for (int j = q; j < d; j += 4) mx = fmaxf(mx, A[i][j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
        if (row) for (int j = q; j < d; j += 4) { const float e = expf(A[i][j] - mx); A[i][j] = e; sum += e; }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
        const float inv = 1.0f / sum;
        if (row) for (int j = q; j < d; j += 4) A[i][j] *= inv;
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
        // 2 x 4 register tiles (two output channels x four head columns): per i two broadcast reads of wps and ONE 16-byte read of A feed
        // eight FMAs — with one output per thread the fold was LDS-bandwidth bound (two 4-byte reads per FMA, 10 us of the stage-3 launch)
        const int jq = (d + 3) >> 2, ntile = ((rows + 1) >> 1) * jq;
        for (int e = tid; e < ntile; e += 256) {
            const int rp = e / jq, r0 = rp * 2, j0 = (e - rp * jq) * 4;
            const int r1 = r0 + 1 < rows ? r0 + 1 : r0;
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < d; ++i) {
                const float4 av = *reinterpret_cast<const float4*>(&A[i][j0]);
                const float w0 = wps[r0][i], w1 = wps[r1][i];
                a0[0] += av.x * w0; a0[1] += av.y * w0; a0[2] += av.z * w0; a0[3] += av.w * w0;
                a1[0] += av.x * w1; a1[1] += av.y * w1; a1[2] += av.z * w1; a1[3] += av.w * w1;
            }
            ACH_UNROLL
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q;
                if (j >= d) continue;
                Store<T>::st(W + wfrag_offset(c0 + r0, h * d + j, p.NT, p.ksteps, Store<T>::VEC), a0[q] * p.gamma[c0 + r0]);
                if (r1 != r0) Store<T>::st(W + wfrag_offset(c0 + r1, h * d + j, p.NT, p.ksteps, Store<T>::VEC), a1[q] * p.gamma[c0 + r1]);
            }
        }
    }
}

}  // namespace ach
