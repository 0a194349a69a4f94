// k_xcaframe.h — EdgeNeXt's cross-covariance attention (XCA, edgenext_modules/sdta_encoder.py:162-185 + the residual of :60-62) in TWO launches instead of four
// (16-bit engines; round 6).
//
//     qkv  = Wqkv LN(y) + b                         (k_gemm.h gemm_body, LayerNorm prologue)
//     G_h  = q_h^T k_h, |q_i|, |k_j| over the frame's tokens          (k_xca.h xca_gram_mfma_body)
//     P_h  = softmax(G_h / (|q_i| |k_j|) * temperature_h)             (rows of d <= 64 entries)
//     Weff = gamma * Wproj * blockdiag(P_h)                           (the fold of "attn @ v -> proj -> layer scale" into per-frame weights, on the matrix cores)
//     t2   = y + Weff v + gamma * bproj                               (gemm_body, per-frame weights, residual)
//
// Until round 5 these were four launches (qkv GEMM, Gram partials over token slices, finalize, projection GEMM) per SDTA block — 12 of the 113 launches of
// EN-GDF-PN-S0, 190 us of isolated time for ~50 MFLOP per frame and, on the caller's stream, 0.120 ms of the 1.53 ms step (profiles/r05_skip_ops_en_s0.txt): each
// of them a full-chip launch that lives for a few L2 round trips.  The front kernel computes a token slice's qkv and, from the q and k it has just written (read back
// by the compute unit that wrote them, behind a workgroup barrier: L2 hits, no agent-scope fence), the slice's partial Gram sums.  The back kernel's workgroups each
// rebuild their frame's Weff from the partials — softmax of a d x d matrix per head and a C x d x d fold per head on the matrix cores cost less than the launch they
// replace, and every workgroup of a frame writes the SAME bytes, so there is no hand-off between workgroups — and then run the projection GEMM of their 64 tokens.
//
// Work is dealt to a workgroup's waves in (16-token tile, 64-channel chunk) units of gemm_body — the same per-row sums as the separate launches.  The Gram partials
// are summed over other slice boundaries than before (fp32, last-bit differences), and the fold runs on the matrix cores (P and gamma * Wproj rounded to the storage
// type, fp32 accumulation) instead of fp32 VALU FMAs: Weff differs from the four-launch path by an ulp of the storage type now and then.
#pragma once
#include "k_gemm.h"
#include "k_xca.h"

namespace ach {

// the two instantiated LDS budgets (A floats incl. the 2 C norms, Pt elements): EdgeNeXt-S0 (C <= 176, 4 heads, d <= 44) and S1 / S2 (C <= 288, d <= 56 at 4 heads / d <= 36 at 8)
constexpr int XCAF_TINY_AFL = 3072, XCAF_TINY_PEL = 8 * 32 * 40;            // the 40 x 40 / 20 x 20 stages (C <= 144, d <= 24): 32 KB, several workgroups per compute unit
constexpr int XCAF_SMALL_AFL = 176 * 47, XCAF_SMALL_PEL = 4 * 48 * 72;
constexpr int XCAF_BIG_AFL = 224 * 59, XCAF_BIG_PEL = 8 * 48 * 72;

// the attention phase shared by xca_frame_kernel and xca_back_kernel: partial Gram sums (S token slices, summed in slice order as xca_finalize_kernel does) -> Weff of
// frame b in the projection GEMM's fragment order (global, p.proj.W + b * stride).  All threads of the workgroup; ends behind a barrier.
struct XcaFoldParams {
    const float* partial; int S;    // [B][heads][S][d*d + 2d]
    const float* temperature;       // [heads]
    const void* Wpg;                // gamma[co] * Wproj[co][h*d + i] as MFMA B fragments: [((h * CT + ct) * KS + s) * 64 + lane] x 16 bytes
    float* attn;                    // optional [B][heads][d][d] (tap) or nullptr
    int C, heads, d, KS, CT;
};
template <class T, int AFL, int PEL, int NWV>
__device__ __forceinline__ void xca_fold_body(const XcaFoldParams& f, const GemmParams& proj, unsigned char* smem, int b, int hb = 0, int he = -1) {
    if (he < 0) he = f.heads;                                       // heads [hb, he) of the frame (xca_finalize_mfma_kernel: one head per workgroup)
    constexpr int VEC = Store<T>::VEC, KC = 4 * VEC, NTH = 64 * NWV;
    const int tid = int(threadIdx.x), lane = tid & 63, wave = wave_uniform(tid >> 6);
    float* A = reinterpret_cast<float*>(smem);
    T* Pt = reinterpret_cast<T*>(smem + AFL * 4);
    const int d = f.d, dp = d + 1, npair = d * d, stride = npair + 2 * d, tm = (d + 15) >> 4, DR = tm * 16, KP = f.KS * KC + VEC;
    // ---- A = G / (|q_i| |k_j|) * temperature (xca_finalize_kernel's expression); the norms first, in a strip behind A's rows
    float* nrm = A + f.C * dp;                                      // [2][C]: |q_(h,i)|, |k_(h,j)|
    const float* part = f.partial + long(b) * f.heads * f.S * stride;
    const int R_lo = hb * d, R_hi = he * d;                         // rows (head, i) of A this call works on
    for (int e = tid; e < 2 * f.C; e += NTH) {
        const int w = e >= f.C, R = e - w * f.C, h = R / d, i = R - h * d;
        if (R < R_lo || R >= R_hi) continue;
        const float* ph = part + long(h) * f.S * stride + npair + w * d + i;
        float s = 0.f;
        for (int sp = 0; sp < f.S; ++sp) s += ph[long(sp) * stride];
        nrm[e] = fmaxf(sqrtf(s), 1e-12f);
    }
    {
        uint4* pz = reinterpret_cast<uint4*>(Pt);
        const int z0 = hb * DR * KP * int(sizeof(T)) / 16, nz = (he * DR * KP * int(sizeof(T)) + 15) / 16;      // (DR KP sizeof(T) is a multiple of 16)
        for (int e = z0 + tid; e < nz; e += NTH) pz[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    for (int e = R_lo * d + tid; e < R_hi * d; e += NTH) {
        const int R = e / d, j = e - R * d, h = R / d, i = R - h * d;
        const float* ph = part + long(h) * f.S * stride + i * d + j;
        float s = 0.f;
        for (int sp = 0; sp < f.S; ++sp) s += ph[long(sp) * stride];
        A[R * dp + j] = s / (nrm[R] * nrm[f.C + h * d + j]) * f.temperature[h];
    }
    __syncthreads();
    // ---- row softmax, four lanes per row (every lane of a wave takes part in the butterflies); the result goes to Pt[h][j][i]
    for (int R0 = R_lo; R0 < R_hi; R0 += NTH / 4) {
        const int R = R0 + (tid >> 2), q = tid & 3;
        const bool row = R < R_hi;
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
                if (f.attn) f.attn[(long(b) * f.heads + h) * npair + i * d + j] = v;
            }
        }
    }
    __syncthreads();
    // ---- the fold on the matrix cores: D[j][co] = sum_i Pt[h][j][i] * (gamma[co] Wproj[co][h d + i]) = Weff[co][h d + j], written in the projection GEMM's
    //      fragment order (wfrag_offset); tiles (head, 16 rows j, 16 channels co) dealt to the waves
    {
        const int col = lane & 15, g = lane >> 4;
        T* W = const_cast<T*>(static_cast<const T*>(proj.W)) + long(b) * proj.w_group_stride;
        const uint4* Wpg = static_cast<const uint4*>(f.Wpg) + lane;
        const int ntile = he * tm * f.CT;
        for (int tile = hb * tm * f.CT + wave; tile < ntile; tile += NWV) {
            const int h = tile / (tm * f.CT), rem = tile - h * tm * f.CT, jt = rem / f.CT, ct = rem - jt * f.CT;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < f.KS; ++s) {
                const uint4 fa = *reinterpret_cast<const uint4*>(Pt + (h * DR + jt * 16 + col) * KP + s * KC + g * VEC);
                const uint4 fb = Wpg[(long(h * f.CT + ct) * f.KS + s) * 64];
                mfma16<T>(fa, fb, acc);
            }
            const int co = ct * 16 + col;
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + 4 * g + r;
                if (j < d && co < f.C) Store<T>::st(W + wfrag_offset(co, h * d + j, 4, proj.ksteps, VEC), acc[r]);
            }
        }
    }
    __syncthreads();
}

struct XcaFrameParams {
    GemmParams qkv;                 // groups = B, M_per_group = N tokens, shared weights, ln = 1, chunks_per_block = 1
    GemmParams proj;                // X = v (channel slice of qkv), W = Weff (per-frame stride), R = y, chunks_per_block = 1
    XcaGramParams gram;             // S = 1, partial = [B][heads][d*d + 2d] scratch
    XcaFoldParams fold;
    int N, heads;
};
// LDS of the attention phase: A = the scaled Gram matrices, fp32 [C rows = (head, i)][d + 1] followed by the 2 C norms; Pt = softmax(A) TRANSPOSED per head as the fold's
// A operand, [head][DR = 16 ceil(d/16) rows j][KP = KS * KC + VEC columns i] of T, zero outside d x d.  In xca_frame_kernel the Gram phase's staging tile shares the bytes.
template <class T, int XCA_DMAX, int AFL, int PEL> struct XcaFrameLds {
    static constexpr int GRAM_BYTES = XcaGramTile<T, XCA_DMAX>::ELEMS * int(sizeof(T));
    static constexpr int A_BYTES = AFL * 4, P_BYTES = PEL * int(sizeof(T));
    static constexpr int BYTES = (GRAM_BYTES > A_BYTES + P_BYTES ? GRAM_BYTES : A_BYTES + P_BYTES + 15) / 16 * 16;
};

// (row tile, 64-channel chunk) units of gemm_body over the 16-token tiles [t0, t1) of frame b, dealt to the workgroup's NWV waves
template <class T, int NWV>
__device__ __forceinline__ void xca_gemm_units(const GemmParams& g, int b, int t0, int t1, int N) {
    const int wave = wave_uniform(int(threadIdx.x) >> 6), nt = t1 - t0;
    const unsigned nbx = unsigned((N + 63) >> 6);
    for (int u = wave; u < nt * g.nchunks; u += NWV) {
        const int tile = t0 + u % nt, chunk = u / nt;
        gemm_body<T, 4, 1>(g, unsigned(tile >> 2), nbx, unsigned(b), unsigned(chunk), tile & 3);
    }
}

// ---- ONE launch, one workgroup per frame.  MEASURED SLOWER than the four launches it replaces (EN-S0, batch 64: 121 / 85 / 96 us against 64 / 56 / 70 us isolated,
// 40.4 k against 41.6 k frames/s, profiles/r06_xca/): every phase is a chain of L2 round trips and 16 waves per frame cannot overlap them.  Kept as option xca_frame = 1
// (the correct statement of the experiment, and the batch-1 form); the default is the two-launch form below.
template <class T, int XCA_DMAX, int AFL, int PEL, int NWV>
__global__ __launch_bounds__(64 * NWV) void xca_frame_kernel(const XcaFrameParams p) { f16_sat_mode<T>();
    using L = XcaFrameLds<T, XCA_DMAX, AFL, PEL>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::BYTES];
    const int b = int(blockIdx.x), ntt = (p.N + 15) >> 4;
    xca_gemm_units<T, NWV>(p.qkv, b, 0, ntt, p.N);
    __syncthreads();
    {
        T* ts = reinterpret_cast<T*>(smem);
        const int ngroups = (p.heads + p.gram.hg - 1) / p.gram.hg;
        for (int grp = 0; grp < ngroups; ++grp) xca_gram_mfma_body<T, XCA_DMAX, NWV>(p.gram, ts, b, grp, 0);
    }
    __syncthreads();
    xca_fold_body<T, AFL, PEL, NWV>(p.fold, p.proj, smem, b);
    xca_gemm_units<T, NWV>(p.proj, b, 0, ntt, p.N);
}

// ---- TWO launches (option xca_frame = 2, the default of the 16-bit engines):
//   xca_front_kernel, workgroup = (frame, slice of `per` tokens): qkv of the slice, then the slice's partial Gram sums — what the qkv GEMM and the Gram launch did, the
//                     slice's q and k read back by the compute unit that wrote them;
//   xca_back_kernel,  workgroup = (frame, block of 64 tokens): sum of the partials + softmax + fold (every workgroup of a frame computes the SAME Weff and writes the
//                     same bytes: no hand-off between workgroups, no agent-scope fence), then the projection GEMM + residual of its tokens.
struct XcaFrontParams { GemmParams qkv; XcaGramParams gram; int N, S; };
template <class T, int XCA_DMAX, int NWV>
__global__ __launch_bounds__(64 * NWV, 4) void xca_front_kernel(const XcaFrontParams p) { f16_sat_mode<T>();
    __shared__ __attribute__((aligned(16))) T ts[XcaGramTile<T, XCA_DMAX>::ELEMS];
    const int b = int(blockIdx.x) / p.S, sp = int(blockIdx.x) - b * p.S;
    const int ntt = (p.N + 15) >> 4, tps = p.gram.per >> 4;
    const int t0 = sp * tps, t1 = (t0 + tps < ntt) ? t0 + tps : ntt;
    xca_gemm_units<T, NWV>(p.qkv, b, t0, t1, p.N);
    __syncthreads();
    const int ngroups = (p.gram.heads + p.gram.hg - 1) / p.gram.hg;
    for (int grp = 0; grp < ngroups; ++grp) xca_gram_mfma_body<T, XCA_DMAX, NWV>(p.gram, ts, b, grp, sp);
}
struct XcaBackParams { GemmParams proj; XcaFoldParams fold; int N, RB; };
template <class T, int AFL, int PEL, int NWV>
__global__ __launch_bounds__(64 * NWV, 4) void xca_back_kernel(const XcaBackParams p) { f16_sat_mode<T>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[(AFL * 4 + PEL * int(sizeof(T)) + 15) / 16 * 16];
    const int b = int(blockIdx.x) / p.RB, rb = int(blockIdx.x) - b * p.RB;
    const int ntt = (p.N + 15) >> 4;
    const int t0 = rb * 4, t1 = (t0 + 4 < ntt) ? t0 + 4 : ntt;
    xca_fold_body<T, AFL, PEL, NWV>(p.fold, p.proj, smem, b);
    xca_gemm_units<T, NWV>(p.proj, b, t0, t1, p.N);
}


// ---- the finalize launch of the FOUR-launch form with the fold on the matrix cores (option xca_fold_mfma): workgroup = (frame, head)
struct XcaFinalMfmaParams { GemmParams proj; XcaFoldParams fold; };
template <class T, int AFL, int PEL>
__global__ __launch_bounds__(256, 4) void xca_finalize_mfma_kernel(const XcaFinalMfmaParams p) { f16_sat_mode<T>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[(AFL * 4 + PEL * int(sizeof(T)) + 15) / 16 * 16];
    const int b = int(blockIdx.x) / p.fold.heads, h = int(blockIdx.x) - b * p.fold.heads;
    xca_fold_body<T, AFL, PEL, 4>(p.fold, p.proj, smem, b, h, h + 1);
}

}  // namespace ach
