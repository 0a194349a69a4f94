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
__global__ __launch_bounds__(256) void mvit_attn_kernel(const MvitAttnParams p) {
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

}  // namespace ach
