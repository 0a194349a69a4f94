// k_mvit.h — multi-head self-attention of the MobileViT blocks (backbone/vision/mobilevit_modules/mobilevit.py:48-73,134-165).
//
// MobileViT unfolds the map into 2x2 patches: 'b d (h ph) (w pw) -> b (ph pw) (h w) d'.  Attention runs among the N = (H/2)(W/2)
// tokens that share the same position (ph, pw) inside their patch — 4 independent token groups per sample — with 4 heads of
// width 8 (scale 8^-0.5).  Everything else in the transformer (LayerNorm, linears, SiLU) is per token, so the tensor stays in
// NHWC and only this kernel knows about the grouping: token n of group (ph, pw) is pixel (2*(n / w2) + ph, 2*(n % w2) + pw).
//   qkv [B,H,W,96] = [q | k | v], each [heads=4][dim_head=8]   ->   out [B,H,W,32]  ([heads][dim_head])
// One workgroup = one (sample, group, head) and a slice of queries; K and V of the group are staged in LDS (N <= 1600 tokens x 8),
// each thread owns one query and runs an online softmax over the keys.
#pragma once
#include "ach_platform.h"

namespace ach {

struct MvitAttnParams { const void* qkv; long ld; void* Y; long ldy; int B, H, Wd, heads; float scale; };
constexpr int MVIT_DH = 8;
constexpr int MVIT_NMAX = 1600;

// NMAX sizes the LDS staging: 400 tokens (a 40x40 map: 25 KB, six workgroups per CU) or 1600 (100 KB, one workgroup per CU —
// with the large instantiation on a 40x40 map the kernel ran at one wave per SIMD and 2.5x slower)
template <class T, int NMAX>
__global__ __launch_bounds__(256) void mvit_attn_kernel(const MvitAttnParams p) { f16_sat_mode<T>();
    __shared__ float ks[NMAX * MVIT_DH];
    __shared__ float vs[NMAX * MVIT_DH];
    const int h2 = p.H / 2, w2 = p.Wd / 2, N = h2 * w2;
    int id = blockIdx.x;
    const int head = id % p.heads; id /= p.heads;
    const int grp = id % 4;
    const long b = id / 4;
    const int ph = grp >> 1, pw = grp & 1;
    const int inner = p.heads * MVIT_DH;
    const T* base = static_cast<const T*>(p.qkv) + b * p.H * long(p.Wd) * p.ld;
    auto pixel = [&](int n) { return long(2 * (n / w2) + ph) * p.Wd + 2 * (n % w2) + pw; };
    for (int e = threadIdx.x; e < N * 2; e += 256) {
        const int n = e >> 1, which = e & 1;                     // 0: k, 1: v
        const T* src = base + pixel(n) * p.ld + (1 + which) * inner + head * MVIT_DH;
        float a[4], c[4];
        Store<T>::ld4(src, a);
        Store<T>::ld4(src + 4, c);
        float* dst = (which ? vs : ks) + n * MVIT_DH;
        dst[0] = a[0]; dst[1] = a[1]; dst[2] = a[2]; dst[3] = a[3]; dst[4] = c[0]; dst[5] = c[1]; dst[6] = c[2]; dst[7] = c[3];
    }
    __syncthreads();
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= N) return;
    float q[MVIT_DH];
    {
        const T* src = base + pixel(n) * p.ld + head * MVIT_DH;
        float a[4], c[4];
        Store<T>::ld4(src, a);
        Store<T>::ld4(src + 4, c);
        // softmax in base 2: log2(e) is folded into the query scale, so every exponential is one v_exp_f32
        const float sc = p.scale * 1.44269504088896341f;
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) { q[i] = a[i] * sc; q[4 + i] = c[i] * sc; }
    }
    float m = -3.0e38f, l = 0.f, acc[MVIT_DH];
    ACH_UNROLL
    for (int i = 0; i < MVIT_DH; ++i) acc[i] = 0.f;
    // online softmax over chunks of 8 keys: one running-max correction per chunk instead of per key
    for (int j0 = 0; j0 < N; j0 += 8) {
        float sj[8];
        float cm = -3.0e38f;
        ACH_UNROLL
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u;
            const float4 k0 = *reinterpret_cast<const float4*>(ks + (j < N ? j : 0) * MVIT_DH);
            const float4 k1 = *reinterpret_cast<const float4*>(ks + (j < N ? j : 0) * MVIT_DH + 4);
            float d = q[0] * k0.x + q[1] * k0.y + q[2] * k0.z + q[3] * k0.w + q[4] * k1.x + q[5] * k1.y + q[6] * k1.z + q[7] * k1.w;
            sj[u] = j < N ? d : -3.0e38f;
            cm = fmaxf(cm, sj[u]);
        }
        const float mn = fmaxf(m, cm);
        const float corr = fast_exp2(m - mn);
        l *= corr;
        ACH_UNROLL
        for (int i = 0; i < MVIT_DH; ++i) acc[i] *= corr;
        ACH_UNROLL
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u;
            const float pj = fast_exp2(sj[u] - mn);                 // padded keys: exp2(-huge) = 0
            const float4 v0 = *reinterpret_cast<const float4*>(vs + (j < N ? j : 0) * MVIT_DH);
            const float4 v1 = *reinterpret_cast<const float4*>(vs + (j < N ? j : 0) * MVIT_DH + 4);
            l += pj;
            acc[0] += pj * v0.x; acc[1] += pj * v0.y; acc[2] += pj * v0.z; acc[3] += pj * v0.w;
            acc[4] += pj * v1.x; acc[5] += pj * v1.y; acc[6] += pj * v1.z; acc[7] += pj * v1.w;
        }
        m = mn;
    }
    const float inv = 1.0f / l;
    float o0[4], o1[4];
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) { o0[i] = acc[i] * inv; o1[i] = acc[4 + i] * inv; }
    T* dst = static_cast<T*>(p.Y) + (b * p.H * long(p.Wd) + pixel(n)) * p.ldy + head * MVIT_DH;
    Store<T>::st4(dst, o0);
    Store<T>::st4(dst + 4, o1);
}

// ------------------------------------------------------------------------------------------ the same attention on the matrix cores
// One workgroup = one (sample, patch group, head) and up to 256 queries; a wave takes 16 queries at a time.
//   S^T = K Q^T   per 16 keys: A = the keys (16 x d), B = Q^T (d x 16 queries).  The accumulator layout puts a lane's four values on
//                 four KEYS of one query, so exp2(S^T - max) of one (bf16: two) key tile(s), packed, IS the B fragment of
//   O^T = V^T P^T : A = V^T (d x keys of the chunk), staged in LDS in exactly the key order the lanes hold.
// The softmax is two-pass over the 400 (<= NMAX) keys — pass 1 recomputes nothing but the maxima (MFMA + max), pass 2 recomputes
// the scores, exponentiates and accumulates — which costs 2 x 25 extra MFMAs per 16 queries and saves the online rescaling's
// cross-lane traffic per chunk; the two reductions across the four lanes that share a query (max, sum) are two xor-shuffles each.
// log2(e) is folded into the query scale (one v_exp_f32 per score).  Head width 8 uses a quarter (bf16) / half (fp32) of the MFMA's
// k-range; the kernel is bound by the exponentials, not by the matrix cores.  Replaces one-query-per-thread VALU dot products:
// 165 us -> see DESIGN (MV-GDF-PN-S2, 40x40 map, batch 64).
template <class T, int NMAX>
__global__ __launch_bounds__(256) void mvit_attn_mfma_kernel(const MvitAttnParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC, CH = 4 * VEC;              // keys per P.V chunk: 32 (bf16) / 16 (fp32)
    constexpr int NT = (NMAX + 15) / 16, NC = (NMAX + CH - 1) / CH;
    __shared__ __attribute__((aligned(16))) T ks[NT * 16 * MVIT_DH];                 // [key][d]
    __shared__ __attribute__((aligned(16))) T vt[NC * MVIT_DH * 4 * VEC];           // [chunk][d][g][j]: key(chunk, g, j) as the lanes hold them
    const int h2 = p.H / 2, w2 = p.Wd / 2, N = h2 * w2;
    int id = blockIdx.x;
    const int head = id % p.heads; id /= p.heads;
    const int grp = id % 4;
    const long b = id / 4;
    const int ph = grp >> 1, pw = grp & 1;
    const int inner = p.heads * MVIT_DH;
    const T* base = static_cast<const T*>(p.qkv) + b * p.H * long(p.Wd) * p.ld;
    auto pixel = [&](int n) { return long(2 * (n / w2) + ph) * p.Wd + 2 * (n % w2) + pw; };
    // key held by slot (g, j) of chunk c:  bf16: j < 4 -> tile 2c, row 4g + j ; j >= 4 -> tile 2c + 1, row 4g + j - 4.  fp32: tile c, row 4g + j
    auto slot_key = [&](int c, int g, int j) { return VEC == 8 ? 32 * c + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : 16 * c + 4 * g + j; };
    const int ntiles = (N + 15) / 16, nchunks = (N + CH - 1) / CH;
    for (int e = threadIdx.x; e < ntiles * 16; e += 256) {                            // keys (zero rows past N)
        float a[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
        if (e < N) { const T* src = base + pixel(e) * p.ld + inner + head * MVIT_DH; Store<T>::ld4(src, a); Store<T>::ld4(src + 4, c); }
        Store<T>::st4(ks + e * MVIT_DH, a);
        Store<T>::st4(ks + e * MVIT_DH + 4, c);
    }
    for (int e = threadIdx.x; e < nchunks * 4 * VEC; e += 256) {                      // values, transposed into the chunk's slot order
        const int c = e / (4 * VEC), gj = e % (4 * VEC), g = gj / VEC, j = gj % VEC;
        const int key = slot_key(c, g, j);
        float a[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
        if (key < N) { const T* src = base + pixel(key) * p.ld + 2 * inner + head * MVIT_DH; Store<T>::ld4(src, a); Store<T>::ld4(src + 4, d); }
        ACH_UNROLL
        for (int dv = 0; dv < 4; ++dv) { Store<T>::st(vt + ((c * MVIT_DH + dv) * 4 + g) * VEC + j, a[dv]); Store<T>::st(vt + ((c * MVIT_DH + 4 + dv) * 4 + g) * VEC + j, d[dv]); }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const float sc = p.scale * 1.44269504088896341f;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    // the lane's share of a K fragment (A operand: row = key, k = d): bf16: g == 0 holds d 0..7 ; fp32: g < 2 hold d 4g .. 4g+3
    // Every lane reads LDS unconditionally and the lanes that hold no part of the fragment are masked to zero afterwards: written as
    // `g == 0 ? *lds_ptr : zero` the compiler selected between the LDS POINTER and the address of `zero` in scratch and fetched the fragment
    // with four generic flat_load_dword — each waited for with vmcnt(0) lgkmcnt(0) right before its MFMA (16 of them in the chunk loop).
    const unsigned kmask = (VEC == 8 ? g == 0 : g < 2) ? 0xffffffffu : 0u;
    auto kfrag = [&](int t) -> uint4 {
        uint4 v = *reinterpret_cast<const uint4*>(ks + (t * 16 + col) * MVIT_DH + (VEC == 8 ? 0 : 4 * (g & 1)));
        v.x &= kmask; v.y &= kmask; v.z &= kmask; v.w &= kmask;
        return v;
    };
    // TWO query tiles per wave and pass (QT): every score MFMA feeds its exponentials at once, so a single tile is one dependent chain
    // (MFMA -> exp2 -> pack -> MFMA into the running output); a second, independent tile fills its bubbles (119 -> see DESIGN, 40x40 maps)
    constexpr int QT = 2;
    const int full_tiles = N / 16, full_chunks = N / CH;
    const int q_lo = blockIdx.y * 16, q_hi = (q_lo + 16 < ntiles) ? q_lo + 16 : ntiles;          // this workgroup's query tiles
    for (int q0 = q_lo + wave * QT; q0 < q_hi; q0 += 4 * QT) {
        int n[QT];
        bool live[QT];
        uint4 qf[QT];
        ACH_UNROLL
        for (int u = 0; u < QT; ++u) {
            live[u] = q0 + u < q_hi;
            n[u] = (live[u] ? q0 + u : q0) * 16 + col;
            const int nq = n[u] < N ? n[u] : N - 1;
            qf[u] = zero;                                                             // B operand: k = d, column = query
            const T* src = base + pixel(nq) * p.ld + head * MVIT_DH;
            float q8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (VEC == 8) { if (g == 0) { float a[4], c[4]; Store<T>::ld4(src, a); Store<T>::ld4(src + 4, c); for (int i = 0; i < 4; ++i) { q8[i] = a[i] * sc; q8[4 + i] = c[i] * sc; } qf[u] = frag_pack<T>(q8); } }
            else if (g < 2) { float a[4]; Store<T>::ld4(src + 4 * g, a); for (int i = 0; i < 4; ++i) q8[i] = a[i] * sc; qf[u] = frag_pack<T>(q8); }
        }
        // pass 1: the queries' maximum scores (full key tiles need no mask; a last partial tile is masked)
        float m[QT];
        ACH_UNROLL
        for (int u = 0; u < QT; ++u) m[u] = -3.0e38f;
        for (int t = 0; t < full_tiles; ++t) {
            const uint4 kf = kfrag(t);
            ACH_UNROLL
            for (int u = 0; u < QT; ++u) {
                f32x4 s4; s4[0] = s4[1] = s4[2] = s4[3] = 0.f;
                mfma16<T>(kf, qf[u], s4);
                m[u] = fmaxf(fmaxf(m[u], fmaxf(s4[0], s4[1])), fmaxf(s4[2], s4[3]));
            }
        }
        if (full_tiles < ntiles) {
            const uint4 kf = kfrag(full_tiles);
            ACH_UNROLL
            for (int u = 0; u < QT; ++u) {
                f32x4 s4; s4[0] = s4[1] = s4[2] = s4[3] = 0.f;
                mfma16<T>(kf, qf[u], s4);
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) if (full_tiles * 16 + g * 4 + r < N) m[u] = fmaxf(m[u], s4[r]);
            }
        }
        ACH_UNROLL
        for (int u = 0; u < QT; ++u) { m[u] = fmaxf(m[u], __shfl_xor(m[u], 16)); m[u] = fmaxf(m[u], __shfl_xor(m[u], 32)); }
        // pass 2: P^T = exp2(S^T - m) chunk by chunk, O^T += V^T P^T
        float l[QT];
        f32x4 o4[QT];
        ACH_UNROLL
        for (int u = 0; u < QT; ++u) { l[u] = 0.f; o4[u][0] = o4[u][1] = o4[u][2] = o4[u][3] = 0.f; }
        for (int c = 0; c < nchunks; ++c) {
            const bool partial = c >= full_chunks;                                    // the last chunk may hold keys past N
            uint4 kf[VEC / 4];
            ACH_UNROLL
            for (int w = 0; w < VEC / 4; ++w) {
                const int t = c * (VEC / 4) + w;
                kf[w] = kfrag(t < ntiles ? t : ntiles - 1);                           // same reason: clamp the tile, mask the value
                const unsigned tm = t < ntiles ? 0xffffffffu : 0u;
                kf[w].x &= tm; kf[w].y &= tm; kf[w].z &= tm; kf[w].w &= tm;
            }
            uint4 vf = *reinterpret_cast<const uint4*>(vt + ((c * MVIT_DH + (col & (MVIT_DH - 1))) * 4 + g) * VEC);                // A: row = d, k = slot
            { const unsigned vm = col < MVIT_DH ? 0xffffffffu : 0u; vf.x &= vm; vf.y &= vm; vf.z &= vm; vf.w &= vm; }
            ACH_UNROLL
            for (int u = 0; u < QT; ++u) {
                float pj[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                ACH_UNROLL
                for (int w = 0; w < VEC / 4; ++w) {
                    const int t = c * (VEC / 4) + w;
                    f32x4 s4; s4[0] = s4[1] = s4[2] = s4[3] = 0.f;
                    mfma16<T>(kf[w], qf[u], s4);
                    ACH_UNROLL
                    for (int r = 0; r < 4; ++r) {
                        const float e = (!partial || t * 16 + g * 4 + r < N) ? fast_exp2(s4[r] - m[u]) : 0.f;
                        pj[4 * w + r] = e; l[u] += e;
                    }
                }
                mfma16<T>(vf, frag_pack<T>(pj), o4[u]);
            }
        }
        ACH_UNROLL
        for (int u = 0; u < QT; ++u) {
            l[u] += __shfl_xor(l[u], 16);
            l[u] += __shfl_xor(l[u], 32);
            if (live[u] && n[u] < N && g < 2) {                                       // rows (d) 4g .. 4g+3 of this query's column
                const float inv = 1.0f / l[u];
                const float o[4] = {o4[u][0] * inv, o4[u][1] * inv, o4[u][2] * inv, o4[u][3] * inv};
                Store<T>::st4(static_cast<T*>(p.Y) + (b * p.H * long(p.Wd) + pixel(n[u])) * p.ldy + head * MVIT_DH + 4 * g, o);
            }
        }
    }
}

}  // namespace ach
