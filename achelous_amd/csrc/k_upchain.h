// k_upchain.h — a decoder level and the NEXT level's low-resolution conv pair in one kernel (bf16 engine; round 3).
//
//     x1 = relu(bilinear x2 (t))      x2 = relu(dw3x3(x1) + b)      y = [x1 | x2]                       (GhostModule at full resolution, upghost_kernel)
//     t' = Wp · relu(Wu y + bu) + bp                                                                     (next level: Upsample's 1x1 + BN + ReLU, Ghost primary conv;
//                                                                                                        neck/ghostdualfpn.py:28-39, 217-236; backbone/ghostnet.py GhostModule)
// `upghost_kernel` wrote y — 2 Cg channels at the level's full resolution, 105 MB per decoder at 160 x 160, batch 64 — and `chain_kernel`
// (the `.lowres_pair` launch of the next level) read it straight back to produce 16 channels.  Here y only exists as the bf16 B fragments of
// the pair's first GEMM, in LDS: the level writes t' (16 channels) and the `.lowres_pair` launch disappears.
//   0. the tile's source pixels of t and the interpolation geometry of its 18 rows / columns -> LDS;
//   1. the tile's x1 with a one-pixel halo -> LDS (fp32), upghost_kernel's arithmetic on the staged values;
//   2. thread = (pixel, 4-channel group): depthwise 3 x 3, ReLU; [x1 | x2] of the pixel rounded to bf16 (the rounding the stored y had) and,
//      once every thread is done with x1, written OVER it as B fragments: 16-pixel tile row t, k-step s = channel / 32, lane group (channel % 32) / 8;
//   3. wave per tile row: chain_kernel's register chain (same packed weights, same MFMA order: bit-identical to the two launches) and
//      one 16-byte store per lane of the first two lane groups.
#pragma once
#include "k_mlp.h"
#include "k_nhwc.h"

namespace ach {

struct UpGhostChainParams {
    UpGhostParams u;                      // Tq / ldt / Wdw / bdw / B / h / w / Cg  (u.Y unused)
    void* Tn; long ldn;                   // t' [B, 2h, 2w, Cout] bf16
    const void* W1; const float* b1;      // chain_kernel's packing (DT = 2): Wu fragments [K1][2][64], bias [32]
    const void* W2; const float* b2;      // Wp fragments [1][2][64], bias [16 + pad]
    int Cout;
};

#ifndef ACH_UPC_WAVES
#define ACH_UPC_WAVES 0            // > 0: register budget for this many waves per SIMD at every width (experiments)
#endif
// register budget: four waves per SIMD at Cg = 16 (four workgroups per CU; measured 59 -> 49 us on the 160 x 160 level against the
// compiler's free choice of 132), three at Cg = 24 / 32 (two workgroups of 6 / 8 waves per CU)
#define ACH_UPC_BOUNDS(CG) __launch_bounds__(16 * CG, (ACH_UPC_WAVES > 0 ? ACH_UPC_WAVES : (CG <= 16 ? 4 : 3)))
template <class T, int CG>
__global__ ACH_UPC_BOUNDS(CG) void upghost_chain_kernel(const UpGhostChainParams q) { f16_sat_mode<T>();
    constexpr int TS = UPG_TS, HS = TS + 2, CQ = CG / 4, K1 = (2 * CG + 31) / 32, NW = 16 * CG / 64;
    // one buffer: x1 (fp32, tile + halo) in phases 1-2, then the B fragments of the tile's 16 rows (the depthwise results wait in registers
    // across the barrier): 21 / 33 / 42 KB for Cg = 16 / 24 / 32 instead of 37 / 64 / 74 — the kernel is latency-bound, workgroups per CU matter
    constexpr int X1_FLOATS = HS * HS * CG, XS_FLOATS = TS * K1 * 64 * 4;
    __shared__ __attribute__((aligned(16))) float smem[X1_FLOATS > XS_FLOATS ? X1_FLOATS : XS_FLOATS];
    float* const x1 = smem;
    uint4* const xs = reinterpret_cast<uint4*>(smem);
    constexpr int SRC = TS / 2 + 4;                          // source rows / columns a halo tile can touch
    __shared__ __attribute__((aligned(16))) float src[SRC * SRC * CG];
    struct GeoTab { int i0, i1; float l; int ok; };
    __shared__ GeoTab ytab[HS], xtab[HS];
    const UpGhostParams& p = q.u;
    const int H = 2 * p.h, Wd = 2 * p.w;
    const int tiles_x = (Wd + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);
    const int bx = int(wg % tiles_x) * TS, by = int((wg / tiles_x) % tiles_y) * TS;
    const long b = wg / (unsigned(tiles_x) * tiles_y);
    const int c = (threadIdx.x % CQ) * 4, slot = threadIdx.x / CQ;
    const float sy = H > 1 ? float(p.h - 1) / float(H - 1) : 0.f, sx = Wd > 1 ? float(p.w - 1) / float(Wd - 1) : 0.f;
    const T* Tq = static_cast<const T*>(p.Tq) + b * p.h * long(p.w) * p.ldt + c;
    // the pair's weights (per wave: up to 6 fragments + 16 biases): at Cg = 16 requested first, their latency hides behind phases 0-2; at the wider
    // levels the 40 registers are what keeps the kernel from two workgroups per CU, so they are fetched in front of phase 3
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const uint4* W1f = static_cast<const uint4*>(q.W1) + lane;
    const uint4* W2f = static_cast<const uint4*>(q.W2) + lane;
    uint4 w1[K1][2], w2[2];
    float b1[8], b2[8];
    auto fetch_weights = [&]() {
        ACH_UNROLL
        for (int s = 0; s < K1; ++s) { w1[s][0] = W1f[(s * 2) * 64]; w1[s][1] = W1f[(s * 2 + 1) * 64]; }
        w2[0] = W2f[0]; w2[1] = W2f[64];
        ACH_UNROLL
        for (int i = 0; i < 8; ++i) { b1[i] = q.b1[g * 8 + i]; b2[i] = q.b2[g * 8 + i]; }
    };
    if (CG <= 16) fetch_weights();
    // ---- 0. the tile's SOURCE pixels of t (at most 12 x 12 for the 18 x 18 halo tile at scale ~ 1/2) -> LDS as fp32, and the interpolation
    // geometry of the tile's 18 rows and 18 columns (upghost_kernel's float arithmetic, once per row / column instead of once per
    // (position, channel quad): the bilinear phase was half of the kernel's ~1 000 VALU instructions per wave, 30 of every 80 of them this)
    const int oy_first = by > 0 ? by - 1 : 0, ox_first = bx > 0 ? bx - 1 : 0;
    int ys0 = int(sy * float(oy_first)), xs0 = int(sx * float(ox_first));
    if (ys0 > p.h - 1) ys0 = p.h - 1;
    if (xs0 > p.w - 1) xs0 = p.w - 1;
    for (int i = slot; i < SRC * SRC; i += 64) {
        const int r = i / SRC, cc = i % SRC;
        const int yy = ys0 + r < p.h ? ys0 + r : p.h - 1, xx = xs0 + cc < p.w ? xs0 + cc : p.w - 1;
        float a[4];
        Store<T>::ld4(Tq + (long(yy) * p.w + xx) * p.ldt, a);
        *reinterpret_cast<float4*>(src + i * CG + c) = make_float4(a[0], a[1], a[2], a[3]);
    }
    if (threadIdx.x < 2 * HS) {
        const bool isx = threadIdx.x >= HS;
        const int k = isx ? int(threadIdx.x) - HS : int(threadIdx.x);
        const int o = (isx ? bx : by) + k - 1, lim = isx ? Wd : H, n = isx ? p.w : p.h;
        const int cl = o < 0 ? 0 : (o >= lim ? lim - 1 : o);
        const float f = (isx ? sx : sy) * float(cl);
        int i0 = int(f);
        if (i0 > n - 1) i0 = n - 1;
        const int i1 = i0 + (i0 < n - 1 ? 1 : 0);
        GeoTab g;
        g.i0 = i0 - (isx ? xs0 : ys0); g.i1 = i1 - (isx ? xs0 : ys0); g.l = f - float(i0); g.ok = (o >= 0 && o < lim) ? 1 : 0;
        (isx ? xtab : ytab)[k] = g;
    }
    __syncthreads();
    // ---- 1. relu(bilinear(t)) on the tile + halo, from LDS (two rounds in flight: fully unrolled the six rounds' 24 float4 loads were
    // hoisted together and the Cg = 24 kernel took 242 registers)
    constexpr int ROUNDS = (HS * HS + 63) / 64;
    _Pragma("unroll 2")
    for (int r = 0; r < ROUNDS; ++r) {
        const int pos_raw = slot + r * 64;
        const int pos = pos_raw < HS * HS ? pos_raw : HS * HS - 1;
        const GeoTab gy = ytab[pos / HS], gx = xtab[pos % HS];
        const bool ok = gy.ok && gx.ok;                           // outside the map: the dw conv's zero padding
        const float ly = gy.l, lx = gx.l, hy = 1.f - ly, hx = 1.f - lx;
        const float4 a = *reinterpret_cast<const float4*>(src + (gy.i0 * SRC + gx.i0) * CG + c);
        const float4 bq = *reinterpret_cast<const float4*>(src + (gy.i0 * SRC + gx.i1) * CG + c);
        const float4 cc = *reinterpret_cast<const float4*>(src + (gy.i1 * SRC + gx.i0) * CG + c);
        const float4 d = *reinterpret_cast<const float4*>(src + (gy.i1 * SRC + gx.i1) * CG + c);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w}, cv[4] = {cc.x, cc.y, cc.z, cc.w}, dv[4] = {d.x, d.y, d.z, d.w};
        float v[4];
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) { const float t = hy * (hx * av[i] + lx * bv[i]) + ly * (hx * cv[i] + lx * dv[i]); v[i] = (ok && t > 0.f) ? t : 0.f; }
        if (pos_raw < HS * HS) *reinterpret_cast<float4*>(x1 + pos * CG + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
    float wk[9][4];
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) { const float4 w = *reinterpret_cast<const float4*>(p.Wdw + k * CG + c); wk[k][0] = w.x; wk[k][1] = w.y; wk[k][2] = w.z; wk[k][3] = w.w; }
    const float4 bb = *reinterpret_cast<const float4*>(p.bdw + c);
    __syncthreads();
    // ---- 2. depthwise 3 x 3 + ReLU; [x1 | x2] of the thread's pixels rounded to bf16, kept in registers until every thread is done with x1
    constexpr int NPIX = TS * TS / 64;
    uint2 fr[NPIX][2];
    ACH_UNROLL
    for (int it = 0; it < NPIX; ++it) {
        const int pix = slot + it * 64;
        const int ty = pix / TS, tx = pix % TS;
        float acc[4] = {bb.x, bb.y, bb.z, bb.w};
        float o1[4] = {0.f, 0.f, 0.f, 0.f};
        ACH_UNROLL
        for (int k = 0; k < 9; ++k) {
            const float4 s = *reinterpret_cast<const float4*>(x1 + ((ty + k / 3) * HS + tx + k % 3) * CG + c);
            acc[0] += s.x * wk[k][0]; acc[1] += s.y * wk[k][1]; acc[2] += s.z * wk[k][2]; acc[3] += s.w * wk[k][3];
            if (k == 4) { o1[0] = s.x; o1[1] = s.y; o1[2] = s.z; o1[3] = s.w; }
        }
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) acc[i] = acc[i] > 0.f ? acc[i] : 0.f;
        fr[it][0] = make_uint2(H16<T>::pack(o1[0], o1[1]), H16<T>::pack(o1[2], o1[3]));
        fr[it][1] = make_uint2(H16<T>::pack(acc[0], acc[1]), H16<T>::pack(acc[2], acc[3]));
#if !defined(ACH_HOSTEMU)
        // the packed results must EXIST before the barrier: left alone, the compiler (Cg = 24) read all 36 taps, sank the arithmetic below the
        // barrier and carried 144 registers across it (242 VGPRs, or 81 spilled at a 168 budget)
        asm volatile("" : "+v"(fr[it][0].x), "+v"(fr[it][0].y), "+v"(fr[it][1].x), "+v"(fr[it][1].y));
#endif
    }
    __syncthreads();
    // B fragments: 16-pixel tile row ty, k-step s = channel / 32, lane group (channel % 32) / 8; channels past 2 Cg of the last k-step are zero
    // (pixels outside the map hold finite values of the clamped interpolation: their columns are never stored)
    if (2 * CG < K1 * 32) {
        constexpr int PADG = (K1 * 32 - 2 * CG) / 8;                       // whole lane groups of padding (Cg = 24: 2)
        for (int i = threadIdx.x; i < TS * PADG * 16; i += 16 * CG) {
            const int ty = i / (PADG * 16), r = i % (PADG * 16);
            xs[(ty * K1 + K1 - 1) * 64 + (4 - PADG) * 16 + r] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    ACH_UNROLL
    for (int it = 0; it < NPIX; ++it) {
        const int pix = slot + it * 64;
        const int ty = pix / TS, tx = pix % TS;
        ACH_UNROLL
        for (int half = 0; half < 2; ++half) {
            const int ch = half * CG + c;                                        // channel of y
            const int s = ch >> 5, gg = (ch & 31) >> 3, e = ch & 7;
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(xs + (ty * K1 + s) * 64 + gg * 16 + tx) + e * 2) = fr[it][half];
        }
    }
    __syncthreads();
    // ---- 3. the next level's conv pair on the tile rows (chain_kernel<bf16, K1, 1>'s chain)
    if (CG > 16) fetch_weights();
    T* Tn = static_cast<T*>(q.Tn);
    for (int ty = wave; ty < TS; ty += NW) {
        f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
        ACH_UNROLL
        for (int s = 0; s < K1; ++s) { const uint4 xf = xs[(ty * K1 + s) * 64 + lane]; mfma16<T>(w1[s][0], xf, h0); mfma16<T>(w1[s][1], xf, h1); }
        float h[8];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { h[r] = h0[r] + b1[r]; h[4 + r] = h1[r] + b1[4 + r]; }
        apply_act_n<T, 8>(h, ACT_RELU);
        const uint4 hf = frag_pack<T>(h);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        mfma16<T>(w2[0], hf, acc0);
        mfma16<T>(w2[1], hf, acc1);
        const int oy = by + ty, ox = bx + px, nb = g * 8;
        if (oy < H && ox < Wd && nb < q.Cout) {
            float o[8];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { o[r] = acc0[r] + b2[r]; o[4 + r] = acc1[r] + b2[4 + r]; }
            Store<T>::st8(Tn + ((b * H + oy) * long(Wd) + ox) * q.ldn + nb, o);
        }
    }
}

}  // namespace ach
