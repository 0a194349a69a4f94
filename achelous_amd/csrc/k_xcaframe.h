// k_xcaframe.h — a whole XCA (cross-covariance attention, edgenext_modules/sdta_encoder.py:162-185 + the residual of :60-62) as ONE launch,
// one workgroup per frame (16-bit engines; round 6).
//
//     qkv  = Wqkv LN(y) + b                         (k_gemm.h gemm_body, LayerNorm prologue)
//     G_h  = q_h^T k_h, |q_i|, |k_j| over the frame's tokens          (k_xca.h xca_gram_mfma_body)
//     P_h  = softmax(G_h / (|q_i| |k_j|) * temperature_h)             (rows of d <= 64 entries)
//     Weff = gamma * Wproj * blockdiag(P_h)                           (MFMA: the fold of "attn @ v -> proj -> layer scale" into per-frame weights)
//     t2   = y + Weff v + gamma * bproj                               (gemm_body, per-frame weights, residual)
//
// Until round 5 these were four launches (qkv GEMM, Gram partials over token slices, finalize, projection GEMM) per SDTA block — 12 of the 113 launches of
// EN-GDF-PN-S0, 187 us of isolated time for ~50 MFLOP per frame and, on the caller's stream, 0.120 ms of the 1.53 ms step (profiles/r05_skip_ops_en_s0.txt): each of
// them a full-chip launch that lives for one or two L2 round trips.  The attention matrix couples all tokens of a frame, and nothing couples two frames: a frame is
// the natural unit of work.  A workgroup of 16 waves owns one frame from the LayerNorm to the residual; q, k, v, the Gram sums and the folded weights go through
// global scratch that only this compute unit touches (its own writes, read back behind a workgroup barrier: L2 hits, no agent-scope fence anywhere), the softmax and
// the fold's A operand live in LDS.  64 workgroups leave 192 compute units to the two side streams for the launch's whole life.
//
// Work is dealt to the waves in (16-token tile, 64-channel chunk) units of gemm_body — the same per-row sums as the separate launches, so qkv, the Gram sums and
// t2's GEMM are bit-identical to them given the same Weff; the fold runs on the matrix cores here (P and gamma * Wproj rounded to the storage type, fp32
// accumulation) instead of fp32 VALU FMAs: Weff differs from the four-launch path by its last bit now and then (test: within 2 ulp of the storage type).
#pragma once
#include "k_gemm.h"
#include "k_xca.h"

namespace ach {

// the two instantiated LDS budgets (A floats, Pt elements): EdgeNeXt-S0 (C <= 176, 4 heads, d <= 44) and S1 / S2 (C <= 288, d <= 56 at 4 heads / d <= 36 at 8)
constexpr int XCAF_SMALL_AFL = 176 * 45, XCAF_SMALL_PEL = 4 * 48 * 72;
constexpr int XCAF_BIG_AFL = 224 * 57, XCAF_BIG_PEL = 8 * 48 * 72;

struct XcaFrameParams {
    GemmParams qkv;                 // groups = B, M_per_group = N tokens, shared weights, ln = 1, chunks_per_block = 1
    GemmParams proj;                // X = v (channel slice of qkv), W = Weff (per-frame stride), R = y, chunks_per_block = 1
    XcaGramParams gram;             // S = 1, partial = [B][heads][d*d + 2d] scratch
    const float* temperature;       // [heads]
    const void* Wpg;                // gamma[co] * Wproj[co][h*d + i] as MFMA B fragments: [((h * CT + ct) * KS + s) * 64 + lane] x 16 bytes
    float* attn;                    // optional [B][heads][d][d] (tap) or nullptr
    int B, N, C, heads, d, KS, CT;
};

// LDS of the attention phase: A = the scaled Gram matrices, fp32 [C rows = (head, i)][d + 1]; Pt = softmax(A) TRANSPOSED per head as the fold's A operand,
// [head][DR = 16 ceil(d/16) rows j][KP = KS * KC + VEC columns i] of T, zero outside d x d.  The Gram phase's staging tile shares the bytes.
template <class T, int XCA_DMAX, int AFL, int PEL> struct XcaFrameLds {
    static constexpr int GRAM_BYTES = XcaGramTile<T, XCA_DMAX>::ELEMS * int(sizeof(T));
    static constexpr int A_BYTES = AFL * 4, P_BYTES = PEL * int(sizeof(T));
    static constexpr int BYTES = (GRAM_BYTES > A_BYTES + P_BYTES ? GRAM_BYTES : A_BYTES + P_BYTES + 15) / 16 * 16;
};

template <class T, int XCA_DMAX, int AFL, int PEL, int NWV>
__global__ __launch_bounds__(64 * NWV) void xca_frame_kernel(const XcaFrameParams p) { f16_sat_mode<T>();
    using L = XcaFrameLds<T, XCA_DMAX, AFL, PEL>;
    constexpr int VEC = Store<T>::VEC, KC = 4 * VEC, NTH = 64 * NWV;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::BYTES];
    const int b = int(blockIdx.x);
    const int tid = int(threadIdx.x), lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int ntt = (p.N + 15) >> 4;                                  // 16-token tiles of the frame
    const unsigned nbx = unsigned((p.N + 63) >> 6);
    // ---- 1. qkv
    for (int u = wave; u < ntt * p.qkv.nchunks; u += NWV) {
        const int tile = u % ntt, chunk = u / ntt;
        gemm_body<T, 4, 1>(p.qkv, unsigned(tile >> 2), nbx, unsigned(b), unsigned(chunk), tile & 3);
    }
    __syncthreads();
    // ---- 2. Gram sums + squared norms of every head group, over all tokens of the frame
    {
        T* ts = reinterpret_cast<T*>(smem);
        const int ngroups = (p.heads + p.gram.hg - 1) / p.gram.hg;
        for (int grp = 0; grp < ngroups; ++grp) xca_gram_mfma_body<T, XCA_DMAX, NWV>(p.gram, ts, b, grp, 0);
    }
    __syncthreads();
    // ---- 3. A = G / (|q_i| |k_j|) * temperature (xca_finalize_kernel's expression), Pt cleared
    float* A = reinterpret_cast<float*>(smem);
    T* Pt = reinterpret_cast<T*>(smem + L::A_BYTES);
    const int d = p.d, dp = d + 1, npair = d * d, tm = (d + 15) >> 4, DR = tm * 16, KP = p.KS * KC + VEC;
    {
        const float* part = p.gram.partial + long(b) * p.heads * (npair + 2 * d);
        for (int e = tid; e < p.C * d; e += NTH) {
            const int R = e / d, j = e - R * d, h = R / d, i = R - h * d;
            const float* ph = part + long(h) * (npair + 2 * d);
            const float nq = fmaxf(sqrtf(ph[npair + i]), 1e-12f), nk = fmaxf(sqrtf(ph[npair + d + j]), 1e-12f);
            A[R * dp + j] = ph[i * d + j] / (nq * nk) * p.temperature[h];
        }
        uint4* pz = reinterpret_cast<uint4*>(Pt);
        const int nz = (p.heads * DR * KP * int(sizeof(T)) + 15) / 16;
        for (int e = tid; e < nz; e += NTH) pz[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    // ---- 4. row softmax, four lanes per row (every lane of a wave takes part in the butterflies); the result goes to Pt[h][j][i]
    for (int R0 = 0; R0 < p.C; R0 += NTH / 4) {
        const int R = R0 + (tid >> 2), q = tid & 3;
        const bool row = R < p.C;
        const float* Ar = A + (row ? R : 0) * dp;
        float mx = -3.0e38f;
        if (row) for (int j = q; j < d; j += 4) mx = fmaxf(mx, Ar[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
        if (row) for (int j = q; j < d; j += 4) sum += expf(Ar[j] - mx);
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
        const float inv = 1.0f / sum;
        if (row) {
            const int h = R / d, i = R - h * d;
            for (int j = q; j < d; j += 4) {
                const float v = expf(Ar[j] - mx) * inv;
                Store<T>::st(Pt + (h * DR + j) * KP + i, v);
                if (p.attn) p.attn[(long(b) * p.heads + h) * npair + i * d + j] = v;
            }
        }
    }
    __syncthreads();
    // ---- 5. the fold on the matrix cores: D[j][co] = sum_i Pt[h][j][i] * (gamma[co] Wproj[co][h d + i]) = Weff[co][h d + j], written in the projection GEMM's
    //         fragment order (wfrag_offset); tiles (head, 16 rows j, 16 channels co) dealt to the waves
    {
        const int col = lane & 15, g = lane >> 4;
        T* W = const_cast<T*>(static_cast<const T*>(p.proj.W)) + long(b) * p.proj.w_group_stride;
        const uint4* Wpg = static_cast<const uint4*>(p.Wpg) + lane;
        const int ntile = p.heads * tm * p.CT;
        for (int tile = wave; tile < ntile; tile += NWV) {
            const int h = tile / (tm * p.CT), rem = tile - h * tm * p.CT, jt = rem / p.CT, ct = rem - jt * p.CT;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < p.KS; ++s) {
                const uint4 fa = *reinterpret_cast<const uint4*>(Pt + (h * DR + jt * 16 + col) * KP + s * KC + g * VEC);
                const uint4 fb = Wpg[(long(h * p.CT + ct) * p.KS + s) * 64];
                mfma16<T>(fa, fb, acc);
            }
            const int co = ct * 16 + col;
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + 4 * g + r;
                if (j < d && co < p.C) Store<T>::st(W + wfrag_offset(co, h * d + j, 4, p.proj.ksteps, VEC), acc[r]);
            }
        }
    }
    __syncthreads();
    // ---- 6. t2 = y + Weff v + bias
    for (int u = wave; u < ntt * p.proj.nchunks; u += NWV) {
        const int tile = u % ntt, chunk = u / ntt;
        gemm_body<T, 4, 1>(p.proj, unsigned(tile >> 2), nbx, unsigned(b), unsigned(chunk), tile & 3);
    }
}

}  // namespace ach
