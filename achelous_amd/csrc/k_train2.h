// k_train2.h — the remaining training-mode primitives: with k_train.h (fp32 MFMA GEMM, BatchNorm batch statistics,
// depthwise 3x3, PointNet glue) these are every arithmetic operation `Achelous.forward` performs in `.train()` — forward AND backward —
// for the EdgeNeXt / Ghost-Dual-FPN / RCNet / nano-head / PointNet model (utils/utils_fit.py:37-166 runs them through ATen autograd).
// fp32, NCHW-contiguous tensors as PyTorch lays them out, one thread per output element or one workgroup per reduction: written for
// correctness first (the measured hot path of this repository is inference).  Every backward is in GATHER form — an output
// element sums its contributions in a fixed order, so results are run-to-run identical — except the input gradient of the deformable
// sampling, which scatters with fp32 atomics as torchvision's kernel does.
#pragma once
#include "ach_platform.h"
#include "k_train.h"

namespace ach {

// ------------------------------------------------------------------------------------------ block reduction helper
__device__ __forceinline__ float block_sum_256(float v, float* sh) {       // sh: 256 floats of LDS; every thread gets the total
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}

// i / d for index arithmetic: 32-bit division where both fit (this target has no integer divider: a 64-bit division is a long sequence even through the compiler's run-time 32-bit bypass, a 32-bit one ~30 instructions;
// the one-thread-per-element kernels below split their linear index three to five times per element)
__device__ __forceinline__ long tdiv(long i, long d) { return ((static_cast<unsigned long>(i) | static_cast<unsigned long>(d)) >> 32) == 0 ? long(unsigned(i) / unsigned(d)) : i / d; }

// ------------------------------------------------------------------------------------------ activations
// kind 0 ReLU, 1 SiLU, 2 GELU (erf form, nn.GELU default), 3 sigmoid.  dy == nullptr: out = f(x); else out = dy * f'(x).
struct TrainActParams { const float* x; const float* dy; float* out; long n; int kind; };
__device__ __forceinline__ float train_act_value(float x, float dy, bool bwd, int kind) {
    float f, d;
    if (kind == 0) { f = x > 0.f ? x : 0.f; d = x > 0.f ? 1.f : 0.f; }
    else if (kind == 1) { const float s = 1.f / (1.f + expf(-x)); f = x * s; d = s * (1.f + x * (1.f - s)); }
    else if (kind == 2) {
        const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
        f = x * cdf; d = cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
    } else { const float s = 1.f / (1.f + expf(-x)); f = s; d = s * (1.f - s); }
    return bwd ? dy * d : f;
}
static __global__ __launch_bounds__(256) void train_act_kernel(const TrainActParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.n) return;
    p.out[i] = train_act_value(p.x[i], p.dy ? p.dy[i] : 0.f, p.dy != nullptr, p.kind);
}
// four elements per thread with 16-byte accesses (n a multiple of four, 16-byte aligned tensors: every activation map of the model)
static __global__ __launch_bounds__(256) void train_act4_kernel(const TrainActParams p) {
    const long i = (long(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (i >= p.n) return;
    const float4 x = *reinterpret_cast<const float4*>(p.x + i);
    const float4 g = p.dy ? *reinterpret_cast<const float4*>(p.dy + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool bwd = p.dy != nullptr;
    *reinterpret_cast<float4*>(p.out + i) = make_float4(train_act_value(x.x, g.x, bwd, p.kind), train_act_value(x.y, g.y, bwd, p.kind),
                                                        train_act_value(x.z, g.z, bwd, p.kind), train_act_value(x.w, g.w, bwd, p.kind));
}

// element-wise product (gates that are full tensors: shuffle_attention.py:66); its backward is the same kernel with the other factor
struct TrainMulParams { const float* a; const float* b; float* out; long n; };
static __global__ __launch_bounds__(256) void train_mul_kernel(const TrainMulParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i < p.n) p.out[i] = p.a[i] * p.b[i];
}

// ------------------------------------------------------------------------------------------ LayerNorm over C
// element (r, c, i) at (r * C + c) * inner + i; one normalisation group per (r, i): channels-last rows (inner = 1, edgenext_modules/layers.py
// "channels_last") and channels_first maps [B, C, H*W] (inner = H*W).  Biased variance, eps inside the square root.
struct TrainLnParams { const float* x; const float* gamma; const float* beta; float* y; float* mean; float* rstd; long rows; int C; long inner; float eps; };
static __global__ __launch_bounds__(256) void train_ln_fwd_kernel(const TrainLnParams p) {
    const long gidx = long(blockIdx.x) * 256 + threadIdx.x;
    if (gidx >= p.rows * p.inner) return;
    const long r = gidx / p.inner, i = gidx - r * p.inner;
    const float* x = p.x + r * p.C * p.inner + i;
    float m = 0.f;
    for (int c = 0; c < p.C; ++c) m += x[c * p.inner];
    m /= float(p.C);
    float v = 0.f;
    for (int c = 0; c < p.C; ++c) { const float d = x[c * p.inner] - m; v += d * d; }
    const float rs = 1.f / sqrtf(v / float(p.C) + p.eps);
    float* y = p.y + r * p.C * p.inner + i;
    for (int c = 0; c < p.C; ++c) y[c * p.inner] = (x[c * p.inner] - m) * rs * p.gamma[c] + p.beta[c];
    p.mean[gidx] = m; p.rstd[gidx] = rs;
}
// the same with four consecutive `inner` positions per thread and 16-byte accesses (inner a multiple of four: every channels-first map of the model)
static __global__ __launch_bounds__(256) void train_ln_fwd4_kernel(const TrainLnParams p) {
    const long q = long(blockIdx.x) * 256 + threadIdx.x, iq = p.inner >> 2;
    if (q >= p.rows * iq) return;
    const long r = tdiv(q, iq), i = (q - r * iq) * 4;
    const float* x = p.x + r * p.C * p.inner + i;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < p.C; ++c) { const float4 t = *reinterpret_cast<const float4*>(x + c * p.inner); m[0] += t.x; m[1] += t.y; m[2] += t.z; m[3] += t.w; }
    ACH_UNROLL
    for (int k = 0; k < 4; ++k) m[k] /= float(p.C);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < p.C; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(x + c * p.inner);
        const float d0 = t.x - m[0], d1 = t.y - m[1], d2 = t.z - m[2], d3 = t.w - m[3];
        v[0] += d0 * d0; v[1] += d1 * d1; v[2] += d2 * d2; v[3] += d3 * d3;
    }
    float rs[4];
    ACH_UNROLL
    for (int k = 0; k < 4; ++k) rs[k] = 1.f / sqrtf(v[k] / float(p.C) + p.eps);
    float* y = p.y + r * p.C * p.inner + i;
    for (int c = 0; c < p.C; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(x + c * p.inner);
        const float g = p.gamma[c], b = p.beta[c];
        *reinterpret_cast<float4*>(y + c * p.inner) = make_float4((t.x - m[0]) * rs[0] * g + b, (t.y - m[1]) * rs[1] * g + b, (t.z - m[2]) * rs[2] * g + b, (t.w - m[3]) * rs[3] * g + b);
    }
    const long gidx = r * p.inner + i;
    *reinterpret_cast<float4*>(p.mean + gidx) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(p.rstd + gidx) = make_float4(rs[0], rs[1], rs[2], rs[3]);
}

struct TrainLnBwdParams { const float* x; const float* dy; const float* gamma; const float* mean; const float* rstd; float* dx; float* dgamma; float* dbeta;
                          long rows; int C; long inner;
                          int S; float* ws; };     // S > 1: the parameter gradients' (rows, inner) range in S slices (grid C x S), partials in ws [2][C][S], summed in order by the finalize kernel
static __global__ __launch_bounds__(256) void train_ln_bwd_dx_kernel(const TrainLnBwdParams p) {
    const long gidx = long(blockIdx.x) * 256 + threadIdx.x;
    if (gidx >= p.rows * p.inner) return;
    const long r = gidx / p.inner, i = gidx - r * p.inner;
    const long base = r * p.C * p.inner + i;
    const float m = p.mean[gidx], rs = p.rstd[gidx];
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < p.C; ++c) {
        const float g = p.dy[base + c * p.inner] * p.gamma[c], xh = (p.x[base + c * p.inner] - m) * rs;
        s1 += g; s2 += g * xh;
    }
    s1 /= float(p.C); s2 /= float(p.C);
    for (int c = 0; c < p.C; ++c) {
        const float g = p.dy[base + c * p.inner] * p.gamma[c], xh = (p.x[base + c * p.inner] - m) * rs;
        p.dx[base + c * p.inner] = rs * (g - s1 - xh * s2);
    }
}
static __global__ __launch_bounds__(256) void train_ln_bwd_dx4_kernel(const TrainLnBwdParams p) {      // (four `inner` positions per thread: see train_ln_fwd4_kernel)
    const long q = long(blockIdx.x) * 256 + threadIdx.x, iq = p.inner >> 2;
    if (q >= p.rows * iq) return;
    const long r = tdiv(q, iq), i = (q - r * iq) * 4;
    const long base = r * p.C * p.inner + i, gidx = r * p.inner + i;
    const float4 m4 = *reinterpret_cast<const float4*>(p.mean + gidx), r4 = *reinterpret_cast<const float4*>(p.rstd + gidx);
    const float m[4] = {m4.x, m4.y, m4.z, m4.w}, rs[4] = {r4.x, r4.y, r4.z, r4.w};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < p.C; ++c) {
        const float4 d = *reinterpret_cast<const float4*>(p.dy + base + c * p.inner), x = *reinterpret_cast<const float4*>(p.x + base + c * p.inner);
        const float gm = p.gamma[c];
        const float dv[4] = {d.x, d.y, d.z, d.w}, xv[4] = {x.x, x.y, x.z, x.w};
        ACH_UNROLL
        for (int k = 0; k < 4; ++k) { const float g = dv[k] * gm, xh = (xv[k] - m[k]) * rs[k]; s1[k] += g; s2[k] += g * xh; }
    }
    ACH_UNROLL
    for (int k = 0; k < 4; ++k) { s1[k] /= float(p.C); s2[k] /= float(p.C); }
    for (int c = 0; c < p.C; ++c) {
        const float4 d = *reinterpret_cast<const float4*>(p.dy + base + c * p.inner), x = *reinterpret_cast<const float4*>(p.x + base + c * p.inner);
        const float gm = p.gamma[c];
        const float dv[4] = {d.x, d.y, d.z, d.w}, xv[4] = {x.x, x.y, x.z, x.w};
        float o[4];
        ACH_UNROLL
        for (int k = 0; k < 4; ++k) { const float g = dv[k] * gm, xh = (xv[k] - m[k]) * rs[k]; o[k] = rs[k] * (g - s1[k] - xh * s2[k]); }
        *reinterpret_cast<float4*>(p.dx + base + c * p.inner) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// dgamma / dbeta: one workgroup per (channel, slice).  (Round 5: it was one workgroup per CHANNEL — 32 to 176 workgroups on a 256-CU chip — with a 64-bit division per element:
// 96 us on average, 1.8 ms of a batch-32 step.)
static __global__ __launch_bounds__(256) void train_ln_bwd_param_kernel(const TrainLnBwdParams p) {
    __shared__ float sh[256];
    const int c = blockIdx.x;
    const long groups = p.rows * p.inner;
    const long per = p.S > 1 ? (groups + p.S - 1) / p.S : groups, lo = p.S > 1 ? long(blockIdx.y) * per : 0, hi = lo + per < groups ? lo + per : groups;
    float a = 0.f, b = 0.f;
    if (p.inner < (1L << 30)) {
        BnWalk w(lo + threadIdx.x, int(p.inner));
        for (long gidx = lo + threadIdx.x; gidx < hi; gidx += 256, w.step()) {
            const long e = w.offset(p.C, c);
            const float dy = p.dy[e];
            a += dy * (p.x[e] - p.mean[gidx]) * p.rstd[gidx];
            b += dy;
        }
    } else
    for (long gidx = lo + threadIdx.x; gidx < hi; gidx += 256) {
        const long r = gidx / p.inner, i = gidx - r * p.inner;
        const long e = (r * p.C + c) * p.inner + i;
        const float dy = p.dy[e];
        a += dy * (p.x[e] - p.mean[gidx]) * p.rstd[gidx];
        b += dy;
    }
    a = block_sum_256(a, sh); b = block_sum_256(b, sh);
    if (threadIdx.x == 0) {
        if (p.S > 1) { p.ws[long(c) * p.S + blockIdx.y] = a; p.ws[(long(p.C) + c) * p.S + blockIdx.y] = b; }
        else { p.dgamma[c] = a; p.dbeta[c] = b; }
    }
}
static __global__ __launch_bounds__(256) void train_ln_bwd_param_finalize_kernel(const TrainLnBwdParams p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    float a = 0.f, b = 0.f;
    for (int j = 0; j < p.S; ++j) { a += p.ws[long(c) * p.S + j]; b += p.ws[(long(p.C) + c) * p.S + j]; }
    p.dgamma[c] = a; p.dbeta[c] = b;
}

// ------------------------------------------------------------------------------------------ depthwise k x k, stride 1, pad k/2
// x [B,C,H,W], w [C,k*k]; flip mirrors the taps (the input gradient of the same layer); bias optional
struct TrainDwParams { const float* x; const float* w; const float* bias; float* y; int B, C, H, W, k, flip; };
// KT > 0: the kernel size as a compile-time constant (3 / 5 / 7 / 9: every depthwise layer of the model) — the tap loops unroll, a row's bounds test is hoisted, the taps' addresses
// are constant offsets; KT = 0: any odd k at run time.  (The run-time form was 6 % of a batch-32 training step.)
template <int KT>
static __global__ __launch_bounds__(256) void train_dwconv_kernel(const TrainDwParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    const long total = long(p.B) * p.C * p.H * p.W;
    if (i >= total) return;
    const long row = tdiv(i, p.W), pl = tdiv(row, p.H);
    const int ox = int(i - row * p.W), oy = int(row - pl * p.H), c = int(pl - tdiv(pl, p.C) * p.C);
    const float* xp = p.x + (i - long(oy) * p.W - ox);
    const int k = KT > 0 ? KT : p.k;
    const float* w = p.w + long(c) * k * k;
    const int r = k / 2;
    float acc = p.bias ? p.bias[c] : 0.f;
    if constexpr (KT > 0) {
        ACH_UNROLL
        for (int ky = 0; ky < KT; ++ky) {
            const int iy = oy + ky - r;
            if (iy < 0 || iy >= p.H) continue;
            const float* xr = xp + long(iy) * p.W;
            ACH_UNROLL
            for (int kx = 0; kx < KT; ++kx) {
                const int ix = ox + kx - r;
                const float wv = w[p.flip ? (KT - 1 - ky) * KT + (KT - 1 - kx) : ky * KT + kx];
                const float xv = xr[ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix)];             // unconditional load from a clamped address, then a select (no divergence at the map's edges)
                acc += ((ix >= 0 && ix < p.W) ? xv : 0.f) * wv;
            }
        }
    } else
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy + ky - r;
        if (iy < 0 || iy >= p.H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox + kx - r;
            if (ix < 0 || ix >= p.W) continue;
            const int t = p.flip ? (k - 1 - ky) * k + (k - 1 - kx) : ky * k + kx;
            acc += xp[long(iy) * p.W + ix] * w[t];
        }
    }
    p.y[i] = acc;
}
// The same convolution with FOUR consecutive outputs of a row per thread (W a multiple of four; one plane per blockIdx.y): the index split — three divisions per output above, ~100 of
// the thread's ~150 instructions — is one division per four outputs, a tap row is K + 3 loads for 4 K products instead of 4 K loads, the store is 16 bytes.  (The per-output form ran at
// 0.9 TB/s on the 320 x 320 layers — VALU-bound by its own index arithmetic — and was 3.1 ms of a batch-32 training step.)
template <int KT>
static __global__ __launch_bounds__(256) void train_dwconv4_kernel(const TrainDwParams p) {
    constexpr int R = KT / 2;
    const int wq = p.W >> 2;
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= unsigned(p.H) * unsigned(wq)) return;
    const int oy = int(q / unsigned(wq)), ox = int(q - unsigned(oy) * unsigned(wq)) * 4;
    const long plane = blockIdx.y;                                   // b * C + c
    const int c = int(plane % p.C);
    const float* xp = p.x + plane * long(p.H) * p.W;
    const float* w = p.w + long(c) * KT * KT;
    const float b0 = p.bias ? p.bias[c] : 0.f;
    float acc[4] = {b0, b0, b0, b0};
    ACH_UNROLL
    for (int ky = 0; ky < KT; ++ky) {
        const int iy = oy + ky - R;
        if (iy < 0 || iy >= p.H) continue;
        const float* xr = xp + long(iy) * p.W;
        float v[KT + 3];
        ACH_UNROLL
        for (int j = 0; j < KT + 3; ++j) {
            const int ix = ox + j - R;
            const float t = xr[ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix)];             // unconditional load from a clamped address, then a select
            v[j] = (ix >= 0 && ix < p.W) ? t : 0.f;
        }
        ACH_UNROLL
        for (int kx = 0; kx < KT; ++kx) {
            const float wv = w[p.flip ? (KT - 1 - ky) * KT + (KT - 1 - kx) : ky * KT + kx];
            ACH_UNROLL
            for (int i = 0; i < 4; ++i) acc[i] += v[kx + i] * wv;
        }
    }
    *reinterpret_cast<float4*>(p.y + plane * long(p.H) * p.W + long(oy) * p.W + ox) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}
// one workgroup per (channel, tap) and SLICE of the (B, H, W) range: a 16-channel 3x3 layer at 320 x 320 is 144 (channel, tap) pairs of
// 819 200 terms each at batch 8 — 144 workgroups walking 3 200 iterations on a 256-CU chip took 2 ms; S > 1: partials into ws [C*k*k][S],
// summed in slice order by train_dwconv_wgrad_finalize_kernel
struct TrainDwWgradParams { const float* x; const float* dz; float* dw; int B, C, H, W, k; int S; float* ws; };
static __global__ __launch_bounds__(256) void train_dwconv_wgrad_kernel(const TrainDwWgradParams p) {
    __shared__ float sh[256];
    const int c = blockIdx.x / (p.k * p.k), t = blockIdx.x % (p.k * p.k);
    const int ky = t / p.k - p.k / 2, kx = t % p.k - p.k / 2;
    const long hw = long(p.H) * p.W, total = long(p.B) * hw;
    const long per = p.S > 1 ? (total + p.S - 1) / p.S : total, lo = p.S > 1 ? long(blockIdx.y) * per : 0, hi = lo + per < total ? lo + per : total;
    float a = 0.f;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
        const long b = e / hw, pix = e - b * hw;
        const int oy = int(pix / p.W), ox = int(pix - long(oy) * p.W);
        const int iy = oy + ky, ix = ox + kx;
        if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) continue;
        const long base = (b * p.C + c) * hw;
        a += p.dz[base + pix] * p.x[base + long(iy) * p.W + ix];
    }
    a = block_sum_256(a, sh);
    if (threadIdx.x == 0) { if (p.S > 1) p.ws[long(blockIdx.x) * p.S + blockIdx.y] = a; else p.dw[blockIdx.x] = a; }
}
// Round 5: ALL k x k taps of a channel in one pass.  The kernel above walks the (B, H, W) range once per TAP — 9 to 81 passes over dz and x per layer, two 64-bit divisions per
// element — and was 9 % of a batch-32 training step (6.7 ms; 1.2 ms for one 16-channel 3x3 layer at 320 x 320).  Here a workgroup owns (channel, slice), a thread keeps k x k
// accumulators in registers, reads dz once per position and the k x k neighbours of x through L1; 32-bit index arithmetic (B H W < 2^31 is checked by the caller); one block
// reduction per tap at the end.  Same partial layout (ws[(c k k + t) S + s]) and finalize kernel; the sums differ from the per-tap kernel's only in the order of the fp32 additions.
template <int K>
static __global__ __launch_bounds__(256) void train_dwconv_wgrad_taps_kernel(const TrainDwWgradParams p) {
    constexpr int R = K / 2, KK = K * K;
    __shared__ float part[4][KK];
    const int c = blockIdx.x;
    const unsigned hw = unsigned(p.H) * unsigned(p.W), total = unsigned(p.B) * hw;
    const unsigned per = p.S > 1 ? (total + unsigned(p.S) - 1) / unsigned(p.S) : total, lo = p.S > 1 ? blockIdx.y * per : 0u, hi = (lo + per < total) ? lo + per : total;
    float acc[KK];
    ACH_UNROLL
    for (int t = 0; t < KK; ++t) acc[t] = 0.f;
    // (a four-positions-per-step walk — K (K + 3) loads of x per four positions instead of 4 K K — was measured SLOWER here: 89 us per call against 70)
    for (unsigned e = lo + threadIdx.x; e < hi; e += 256) {
        const unsigned b = e / hw, pix = e - b * hw;
        const int oy = int(pix / unsigned(p.W)), ox = int(pix - unsigned(oy) * unsigned(p.W));
        const float* xc = p.x + (long(b) * p.C + c) * long(hw);
        const float g = p.dz[(long(b) * p.C + c) * long(hw) + pix];
        ACH_UNROLL
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy + ky - R;
            const bool row = iy >= 0 && iy < p.H;
            const float* xr = xc + long(row ? iy : 0) * p.W;
            ACH_UNROLL
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox + kx - R;
                const bool in = row && ix >= 0 && ix < p.W;
                acc[ky * K + kx] += g * (in ? xr[ix] : 0.f);
            }
        }
    }
    // per tap: butterfly inside the wave (no barrier), the four waves' sums through LDS, ONE barrier for all taps
    const int lane = int(threadIdx.x) & 63, wave = int(threadIdx.x) >> 6;
    ACH_UNROLL
    for (int t = 0; t < KK; ++t) {
        float v = acc[t];
        v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        if (lane == 0) part[wave][t] = v;
    }
    __syncthreads();
    for (int t = int(threadIdx.x); t < KK; t += 256) {
        const float a = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
        if (p.S > 1) p.ws[(long(c) * KK + t) * p.S + blockIdx.y] = a; else p.dw[c * KK + t] = a;
    }
}
static __global__ __launch_bounds__(256) void train_dwconv_wgrad_finalize_kernel(const TrainDwWgradParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.C * p.k * p.k) return;
    float a = 0.f;
    for (int j = 0; j < p.S; ++j) a += p.ws[long(i) * p.S + j];
    p.dw[i] = a;
}

// ------------------------------------------------------------------------------------------ im2col / col2im (dense convolutions through train_gemm)
// col [B][(ci * kh + ky) * kw + kx][oy * Wo + ox] = x[b][ci][oy * sh - ph + ky][ox * sw - pw + kx]   (0 outside)
struct TrainColParams { const float* src; float* dst; int B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo; };
static __global__ __launch_bounds__(256) void train_im2col_kernel(const TrainColParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    const long K = long(p.C) * p.kh * p.kw, O = long(p.Ho) * p.Wo;
    if (i >= long(p.B) * K * O) return;
    const long io = tdiv(i, O), o = i - io * O, b = tdiv(io, K), kk = io - b * K;
    const int oy = int(tdiv(o, p.Wo)), ox = int(o - long(oy) * p.Wo);
    const int kq = int(kk) / p.kw, kx = int(kk) - kq * p.kw, ci = kq / p.kh, ky = kq - ci * p.kh;
    const int iy = oy * p.sh - p.ph + ky, ix = ox * p.sw - p.pw + kx;
    p.dst[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? p.src[((b * p.C + ci) * p.H + iy) * long(p.W) + ix] : 0.f;
}
static __global__ __launch_bounds__(256) void train_col2im_kernel(const TrainColParams p) {      // dx[b][ci][iy][ix] = sum of the col entries that read it
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= long(p.B) * p.C * p.H * p.W) return;
    const long row = tdiv(i, p.W), pl = tdiv(row, p.H), b = tdiv(pl, p.C);
    const int ix = int(i - row * p.W), iy = int(row - pl * p.H), ci = int(pl - b * p.C);
    const long K = long(p.C) * p.kh * p.kw, O = long(p.Ho) * p.Wo;
    float a = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
        const int ty = iy + p.ph - ky;
        if (ty < 0 || ty % p.sh) continue;
        const int oy = ty / p.sh;
        if (oy >= p.Ho) continue;
        for (int kx = 0; kx < p.kw; ++kx) {
            const int tx = ix + p.pw - kx;
            if (tx < 0 || tx % p.sw) continue;
            const int ox = tx / p.sw;
            if (ox >= p.Wo) continue;
            a += p.src[(b * K + (long(ci) * p.kh + ky) * p.kw + kx) * O + long(oy) * p.Wo + ox];
        }
    }
    p.dst[i] = a;
}

// ------------------------------------------------------------------------------------------ softmax over the last dimension
struct TrainSoftmaxParams { const float* x; float* y; const float* dy; float* dx; long rows; int d; };
static __global__ __launch_bounds__(256) void train_softmax_kernel(const TrainSoftmaxParams p) {
    const long r = long(blockIdx.x) * 256 + threadIdx.x;
    if (r >= p.rows) return;
    if (!p.dy) {
        const float* x = p.x + r * p.d;
        float m = x[0];
        for (int j = 1; j < p.d; ++j) m = fmaxf(m, x[j]);
        float s = 0.f;
        for (int j = 0; j < p.d; ++j) s += expf(x[j] - m);
        for (int j = 0; j < p.d; ++j) p.y[r * p.d + j] = expf(x[j] - m) / s;
    } else {
        const float* y = p.y + r * p.d; const float* dy = p.dy + r * p.d;
        float s = 0.f;
        for (int j = 0; j < p.d; ++j) s += dy[j] * y[j];
        for (int j = 0; j < p.d; ++j) p.dx[r * p.d + j] = y[j] * (dy[j] - s);
    }
}

// ------------------------------------------------------------------------------------------ bilinear x2, align_corners=True
__device__ __forceinline__ void up2_src(int o, int n_in, float scale, int& i0, int& i1, float& l) {
    const float f = scale * float(o);
    i0 = int(f);
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 < n_in - 1 ? i0 + 1 : i0;
    l = f - float(i0);
}
struct TrainUpParams { const float* src; float* dst; long planes; int h, w; float sy, sx; };
static __global__ __launch_bounds__(256) void train_up2_fwd_kernel(const TrainUpParams p) {
    const int H = 2 * p.h, W = 2 * p.w;
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.planes * H * W) return;
    const long row = tdiv(i, W), pl = tdiv(row, H);
    const int ox = int(i - row * W), oy = int(row - pl * H);
    int y0, y1, x0, x1; float ly, lx;
    up2_src(oy, p.h, p.sy, y0, y1, ly); up2_src(ox, p.w, p.sx, x0, x1, lx);
    const float* s = p.src + pl * p.h * p.w;
    const float hy = 1.f - ly, hx = 1.f - lx;
    p.dst[i] = hy * (hx * s[y0 * p.w + x0] + lx * s[y0 * p.w + x1]) + ly * (hx * s[y1 * p.w + x0] + lx * s[y1 * p.w + x1]);
}
static __global__ __launch_bounds__(256) void train_up2_bwd_kernel(const TrainUpParams p) {      // src = dy [planes,2h,2w], dst = dx [planes,h,w]
    const int H = 2 * p.h, W = 2 * p.w;
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.planes * p.h * p.w) return;
    const long row = tdiv(i, p.w), pl = tdiv(row, p.h);
    const int ix = int(i - row * p.w), iy = int(row - pl * p.h);
    const float* dy = p.src + pl * H * W;
    float a = 0.f;
    for (int oy = 2 * iy - 2 < 0 ? 0 : 2 * iy - 2; oy <= 2 * iy + 2 && oy < H; ++oy) {
        int y0, y1; float ly;
        up2_src(oy, p.h, p.sy, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int ox = 2 * ix - 2 < 0 ? 0 : 2 * ix - 2; ox <= 2 * ix + 2 && ox < W; ++ox) {
            int x0, x1; float lx;
            up2_src(ox, p.w, p.sx, x0, x1, lx);
            const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            a += wy * wx * dy[long(oy) * W + ox];
        }
    }
    p.dst[i] = a;
}

// ------------------------------------------------------------------------------------------ max pool k x k, stride 1, pad k/2 (SPP) and avg pool 3x3
struct TrainPoolParams { const float* x; float* y; int* idx; const float* dy; float* dx; long planes; int H, W, k; };
static __global__ __launch_bounds__(256) void train_maxpool_kernel(const TrainPoolParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.planes * p.H * p.W) return;
    const int x0 = int(i % p.W), y0 = int((i / p.W) % p.H), r = p.k / 2;
    const long base = i - long(y0) * p.W - x0;
    if (!p.dy) {                                   // forward: first maximum in row-major order, NaN wins (ATen's rule)
        float best = -INFINITY;
        int bi = (y0 - r < 0 ? 0 : y0 - r) * p.W + (x0 - r < 0 ? 0 : x0 - r);
        for (int y = y0 - r < 0 ? 0 : y0 - r; y <= y0 + r && y < p.H; ++y)
            for (int x = x0 - r < 0 ? 0 : x0 - r; x <= x0 + r && x < p.W; ++x) {
                const float v = p.x[base + long(y) * p.W + x];
                if (v > best || v != v) { best = v; bi = y * p.W + x; }
            }
        p.y[i] = best; p.idx[i] = bi;
    } else {                                       // backward: this input collects from the windows whose arg-max it is
        const int me = y0 * p.W + x0;
        float a = 0.f;
        for (int y = y0 - r < 0 ? 0 : y0 - r; y <= y0 + r && y < p.H; ++y)
            for (int x = x0 - r < 0 ? 0 : x0 - r; x <= x0 + r && x < p.W; ++x)
                if (p.idx[base + long(y) * p.W + x] == me) a += p.dy[base + long(y) * p.W + x];
        p.dx[i] = a;
    }
}
static __global__ __launch_bounds__(256) void train_avgpool3_kernel(const TrainPoolParams p) {     // count_include_pad: always / 9; self-adjoint
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.planes * p.H * p.W) return;
    const int x0 = int(i % p.W), y0 = int((i / p.W) % p.H);
    const long base = i - long(y0) * p.W - x0;
    float a = 0.f;
    for (int y = y0 - 1 < 0 ? 0 : y0 - 1; y <= y0 + 1 && y < p.H; ++y)
        for (int x = x0 - 1 < 0 ? 0 : x0 - 1; x <= x0 + 1 && x < p.W; ++x) a += p.x[base + long(y) * p.W + x];
    p.y[i] = a * (1.0f / 9.0f);
}

// ------------------------------------------------------------------------------------------ row / column reductions and scalings
// row_reduce: out[r] = scale * sum_i a[r,i] * (b ? b[r,i] : 1)            one workgroup per row
// row_scale : out[r,i] = (x ? x[r,i] : 1) * s[r % S]
// col_reduce: out[c] = sum_r a[r,c] * (b ? b[r,c] : 1)                    one workgroup per column
// col_scale : out[r,c] = x[r,c] * g[c]
struct TrainRowParams { const float* a; const float* b; float* out; long rows; long N; long S; float scale; };
static __global__ __launch_bounds__(256) void train_row_reduce_kernel(const TrainRowParams p) {
    __shared__ float sh[256];
    const long r = blockIdx.x;
    float v = 0.f;
    if ((p.N & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.a) | reinterpret_cast<uintptr_t>(p.b)) & 15u) == 0) {       // 16-byte loads (rows of a multiple of four floats)
        const float4* a4 = reinterpret_cast<const float4*>(p.a + r * p.N);
        const float4* b4 = p.b ? reinterpret_cast<const float4*>(p.b + r * p.N) : nullptr;
        for (long i = threadIdx.x; i < (p.N >> 2); i += 256) {
            const float4 a = a4[i];
            if (b4) { const float4 b = b4[i]; v += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
            else v += (a.x + a.y) + (a.z + a.w);
        }
    } else
    for (long i = threadIdx.x; i < p.N; i += 256) v += p.a[r * p.N + i] * (p.b ? p.b[r * p.N + i] : 1.f);
    v = block_sum_256(v, sh);
    if (threadIdx.x == 0) p.out[r] = v * p.scale;
}
static __global__ __launch_bounds__(256) void train_row_scale_kernel(const TrainRowParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.rows * p.N) return;
    const long r = i / p.N;
    p.out[i] = (p.a ? p.a[i] : 1.f) * p.b[r % p.S];
}
static __global__ __launch_bounds__(256) void train_col_reduce_kernel(const TrainRowParams p) {      // N = number of columns
    __shared__ float sh[256];
    const long c = blockIdx.x;
    float v = 0.f;
    for (long r = threadIdx.x; r < p.rows; r += 256) v += p.a[r * p.N + c] * (p.b ? p.b[r * p.N + c] : 1.f);
    v = block_sum_256(v, sh);
    if (threadIdx.x == 0) p.out[c] = v * p.scale;
}
static __global__ __launch_bounds__(256) void train_col_scale_kernel(const TrainRowParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= p.rows * p.N) return;
    p.out[i] = p.a[i] * p.b[i % p.N];
}

// ------------------------------------------------------------------------------------------ per-row normalisations
// instance norm (GroupNorm with one channel per group, shuffle_attention.py:20): row r = (b, c), statistics over its N elements
struct TrainInParams { const float* x; const float* dy; const float* gamma; const float* beta; float* y; float* mean; float* rstd; float* dx; float* dg; float* db;
                       long rows; long N; int C; float eps; };
static __global__ __launch_bounds__(256) void train_instnorm_kernel(const TrainInParams p) {
    __shared__ float sh[256];
    const long r = blockIdx.x;
    const int c = int(r % p.C);
    const float* x = p.x + r * p.N;
    if (!p.dy) {
        float s = 0.f;
        for (long i = threadIdx.x; i < p.N; i += 256) s += x[i];
        const float m = block_sum_256(s, sh) / float(p.N);
        float v = 0.f;
        for (long i = threadIdx.x; i < p.N; i += 256) { const float d = x[i] - m; v += d * d; }
        const float rs = 1.f / sqrtf(block_sum_256(v, sh) / float(p.N) + p.eps);
        for (long i = threadIdx.x; i < p.N; i += 256) p.y[r * p.N + i] = (x[i] - m) * rs * p.gamma[c] + p.beta[c];
        if (threadIdx.x == 0) { p.mean[r] = m; p.rstd[r] = rs; }
    } else {
        const float m = p.mean[r], rs = p.rstd[r], g = p.gamma[c];
        const float* dy = p.dy + r * p.N;
        float s1 = 0.f, s2 = 0.f;
        for (long i = threadIdx.x; i < p.N; i += 256) { s1 += dy[i]; s2 += dy[i] * (x[i] - m) * rs; }
        s1 = block_sum_256(s1, sh); s2 = block_sum_256(s2, sh);
        for (long i = threadIdx.x; i < p.N; i += 256) p.dx[r * p.N + i] = g * rs * (dy[i] - s1 / float(p.N) - (x[i] - m) * rs * s2 / float(p.N));
        if (threadIdx.x == 0) { p.dg[r] = s2; p.db[r] = s1; }         // per row: the caller sums over the batch
    }
}
// F.normalize(x, dim=-1): y = x / max(||x||, eps)   (xca.py: q, k over the tokens)
struct TrainL2Params { const float* x; float* y; float* norm; const float* dy; float* dx; long rows; long N; float eps; };
static __global__ __launch_bounds__(256) void train_l2norm_kernel(const TrainL2Params p) {
    __shared__ float sh[256];
    const long r = blockIdx.x;
    const float* x = p.x + r * p.N;
    if (!p.dy) {
        float s = 0.f;
        for (long i = threadIdx.x; i < p.N; i += 256) s += x[i] * x[i];
        const float n = sqrtf(block_sum_256(s, sh));
        const float d = n > p.eps ? n : p.eps;
        for (long i = threadIdx.x; i < p.N; i += 256) p.y[r * p.N + i] = x[i] / d;
        if (threadIdx.x == 0) p.norm[r] = n;
    } else {
        const float n = p.norm[r];
        const float* dy = p.dy + r * p.N;
        if (n > p.eps) {
            float s = 0.f;
            for (long i = threadIdx.x; i < p.N; i += 256) s += dy[i] * x[i];
            s = block_sum_256(s, sh) / (n * n);
            for (long i = threadIdx.x; i < p.N; i += 256) p.dx[r * p.N + i] = (dy[i] - x[i] * s) / n;
        } else {
            for (long i = threadIdx.x; i < p.N; i += 256) p.dx[r * p.N + i] = dy[i] / p.eps;
        }
    }
}

// ------------------------------------------------------------------------------------------ modulated deformable 3x3 sampling (DCNv2)
// torchvision 0.12 deform_conv2d semantics (deformable_im2col / bilinear_interpolate; dcn.py:49-63 calls it with padding 1, one offset
// group): sample position (oy * s - pad + ky + off_y, ox * s - pad + kx + off_x); a sample at or beyond -1 / H (W) is 0; corners
// outside the map contribute 0.  offset [B,18,Ho,Wo] = (dy, dx) per tap, mask [B,9,Ho,Wo].
//   col  [B][(ci*9 + k)][o] = mask * bilinear(x[b,ci], p)
// backward from dcol = W^T dY:
//   dmask[b][k][o]   = sum_ci dcol * bilinear
//   doffset[b][2k+{0,1}][o] = sum_ci dcol * mask * d bilinear / d{py, px}
//   dx: each sample scatters dcol * mask * corner weight to its four corners (atomicAdd; dx zeroed by the caller)
struct TrainDeformParams { const float* x; const float* offset; const float* mask; float* col; const float* dcol; float* dx; float* doffset; float* dmask;
                           int B, C, H, W, Ho, Wo, stride, pad; };
__device__ __forceinline__ float dcn_corner(const float* img, int H, int W, int y, int x) { return (y >= 0 && y < H && x >= 0 && x < W) ? img[long(y) * W + x] : 0.f; }
static __global__ __launch_bounds__(256) void train_deform_im2col_kernel(const TrainDeformParams p) {
    const long O = long(p.Ho) * p.Wo;
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= long(p.B) * p.C * 9 * O) return;
    const long io = tdiv(i, O), o = i - io * O, ic = tdiv(io, 9), b = tdiv(ic, p.C);
    const int k = int(io - ic * 9), ci = int(ic - b * p.C);
    const int oy = int(tdiv(o, p.Wo)), ox = int(o - long(oy) * p.Wo);
    const float py = float(oy * p.stride - p.pad + k / 3) + p.offset[(b * 18 + 2 * k) * O + o];
    const float px = float(ox * p.stride - p.pad + k % 3) + p.offset[(b * 18 + 2 * k + 1) * O + o];
    float v = 0.f;
    if (py > -1.f && px > -1.f && py < float(p.H) && px < float(p.W)) {
        const float* img = p.x + (b * p.C + ci) * long(p.H) * p.W;
        const int y0 = int(floorf(py)), x0 = int(floorf(px));
        const float ly = py - float(y0), lx = px - float(x0), hy = 1.f - ly, hx = 1.f - lx;
        v = hy * hx * dcn_corner(img, p.H, p.W, y0, x0) + hy * lx * dcn_corner(img, p.H, p.W, y0, x0 + 1) +
            ly * hx * dcn_corner(img, p.H, p.W, y0 + 1, x0) + ly * lx * dcn_corner(img, p.H, p.W, y0 + 1, x0 + 1);
    }
    p.col[i] = v * p.mask[(b * 9 + k) * O + o];
}
static __global__ __launch_bounds__(256) void train_deform_bwd_coord_kernel(const TrainDeformParams p) {     // one thread per (b, k, o): doffset, dmask
    const long O = long(p.Ho) * p.Wo;
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= long(p.B) * 9 * O) return;
    const long io = tdiv(i, O), o = i - io * O, b = tdiv(io, 9);
    const int k = int(io - b * 9);
    const int oy = int(tdiv(o, p.Wo)), ox = int(o - long(oy) * p.Wo);
    const float py = float(oy * p.stride - p.pad + k / 3) + p.offset[(b * 18 + 2 * k) * O + o];
    const float px = float(ox * p.stride - p.pad + k % 3) + p.offset[(b * 18 + 2 * k + 1) * O + o];
    const float m = p.mask[(b * 9 + k) * O + o];
    float gm = 0.f, gy = 0.f, gx = 0.f;
    if (py > -1.f && px > -1.f && py < float(p.H) && px < float(p.W)) {
        const int y0 = int(floorf(py)), x0 = int(floorf(px));
        const float ly = py - float(y0), lx = px - float(x0), hy = 1.f - ly, hx = 1.f - lx;
        for (int ci = 0; ci < p.C; ++ci) {
            const float* img = p.x + (b * p.C + ci) * long(p.H) * p.W;
            const float v00 = dcn_corner(img, p.H, p.W, y0, x0), v01 = dcn_corner(img, p.H, p.W, y0, x0 + 1);
            const float v10 = dcn_corner(img, p.H, p.W, y0 + 1, x0), v11 = dcn_corner(img, p.H, p.W, y0 + 1, x0 + 1);
            const float d = p.dcol[((b * p.C + ci) * 9 + k) * O + o];
            gm += d * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11);
            gy += d * m * (hx * (v10 - v00) + lx * (v11 - v01));
            gx += d * m * (hy * (v01 - v00) + ly * (v11 - v10));
        }
    }
    p.dmask[(b * 9 + k) * O + o] = gm;
    p.doffset[(b * 18 + 2 * k) * O + o] = gy;
    p.doffset[(b * 18 + 2 * k + 1) * O + o] = gx;
}
// fp32 add at the L2 (global_atomic_add_f32, no return value): without -munsafe-fp-atomics the plain atomicAdd(float*) is a compare-and-swap LOOP per element —
// the 354 M adds of the first radar block's input gradient at batch 32 took 2.6 ms that way.  The buffers are hipMalloc'd (coarse-grained) device memory, where the
// hardware instruction is valid; the result is the same sum in another (equally unspecified) order.
#if defined(ACH_HOSTEMU)
static inline void train_atomic_add(float* p, float v) { atomicAdd(p, v); }
#else
static __device__ __forceinline__ void train_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
#endif
static __global__ __launch_bounds__(256) void train_deform_bwd_input_kernel(const TrainDeformParams p) {     // one thread per col element: 4 atomic adds
    const long O = long(p.Ho) * p.Wo;
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= long(p.B) * p.C * 9 * O) return;
    const long io = tdiv(i, O), o = i - io * O, ic = tdiv(io, 9), b = tdiv(ic, p.C);
    const int k = int(io - ic * 9), ci = int(ic - b * p.C);
    const int oy = int(tdiv(o, p.Wo)), ox = int(o - long(oy) * p.Wo);
    const float py = float(oy * p.stride - p.pad + k / 3) + p.offset[(b * 18 + 2 * k) * O + o];
    const float px = float(ox * p.stride - p.pad + k % 3) + p.offset[(b * 18 + 2 * k + 1) * O + o];
    if (!(py > -1.f && px > -1.f && py < float(p.H) && px < float(p.W))) return;
    const float g = p.dcol[i] * p.mask[(b * 9 + k) * O + o];
    const int y0 = int(floorf(py)), x0 = int(floorf(px));
    const float ly = py - float(y0), lx = px - float(x0), hy = 1.f - ly, hx = 1.f - lx;
    float* dimg = p.dx + (b * p.C + ci) * long(p.H) * p.W;
    const float wts[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
    ACH_UNROLL
    for (int q = 0; q < 4; ++q) {
        const int y = y0 + (q >> 1), x = x0 + (q & 1);
        if (y >= 0 && y < p.H && x >= 0 && x < p.W) train_atomic_add(dimg + long(y) * p.W + x, g * wts[q]);
    }
}

}  // namespace ach
