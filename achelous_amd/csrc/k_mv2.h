// k_mv2.h — a whole MobileNetV2 inverted-residual block of MobileViT (backbone/vision/mobilevit_modules/mobilevit.py:93-131) in ONE
// launch:   x -> 1x1 (Cin -> hid) + BN + SiLU -> depthwise 3x3 (stride 1 | 2) + BN + SiLU -> 1x1 (hid -> Cout) + BN [+ x].
//
// Layer-wise this was three launches whose traffic is the 4x-expanded hidden tensor: at 160x160 the 128-channel hidden map of
// mv2.1 is 419 MB per batch of 64, written once and read once (the stride-1 blocks write and read it twice), against 105 MB of
// block input — the MV2 blocks were 1.5 ms of MV-GDF-PN-S2's 6.1 ms of kernel time.  Here the hidden map never leaves the CU:
//   phase 1  a workgroup owns an output tile; the 1x1 expansion runs on MFMA over the tile's input REGION (tile * stride + halo),
//            bias + SiLU in the epilogue, and the hidden activations go to LDS in the storage type.  Region pixels outside the
//            map are written as ZERO: they are the depthwise conv's zero padding, which pads the hidden map, not the input.
//   phase 2  per 16 output pixels: lane (pixel, g) owns VEC hidden channels of k-step s — it gathers their nine taps from LDS,
//            applies the folded depthwise weights + bias + SiLU, and the packed result IS the B fragment of k-step s of the
//            projection GEMM (weights as the A operand, so a lane's accumulators are consecutive output channels of its pixel).
//            k-steps are the OUTER loop: the 9 x VEC depthwise weights of a k-step are loaded once and serve every pixel tile of the wave.
// The 1x1 expansion is recomputed on the halo (region / tile = 1.4 for 16x8 stride-1 tiles, 1.2 for 8x4 stride-2 tiles).
#pragma once
#include "ach_platform.h"

namespace ach {

#ifndef ACH_MV2_DEBUG
#define ACH_MV2_DEBUG 0                   // timing builds only: 1 = ReLU instead of SiLU on the hidden map (wrong results)
#endif
__device__ __forceinline__ float mv2_silu(float x) { return (ACH_MV2_DEBUG & 1) ? fmaxf(x, 0.f) : x * sigmoidf_(x); }

struct Mv2Params {
    const void* X; long ldx;                  // NHWC [B,H,W,Cin], ldx >= k1 * 4 * VEC (channel padding zero)
    void* Y; long ldy;                        // NHWC [B,Ho,Wo,Cout]
    const void* W1; const float* b1;          // expansion: A fragments [hid/16][k1][64] x 16 B, bias [hid]          (BN folded)
    const float* Wdw; const float* bdw;       // depthwise: [9][hid] fp32, bias [hid]                                 (BN folded)
    const void* W2; const float* b2;          // projection: A fragments [Cout/16][hid/(4 VEC)][64] x 16 B, bias [Cout] (BN folded)
    const void* R; long ldr;                  // residual (the block input) or nullptr
    int B, H, Wd, Ho, Wo, hid, Cout, k1;
};

// fragment element (row n, column k) of a [N x K] matrix packed as A operands: tile n / 16, k-step k / (4 VEC)
__host__ __device__ __forceinline__ long mv2_frag_offset(int n, int k, int ksteps, int VEC) {
    const int KC = 4 * VEC, t = n >> 4, i = n & 15, s = k / KC, kk = k % KC, kg = kk / VEC, j = kk % VEC;
    return ((long(t) * ksteps + s) * 64 + kg * 16 + i) * VEC + j;
}

// output tile per workgroup; the fp32 parity engine takes shorter tiles so that the hidden region still fits 64 KB of static LDS
template <int STRIDE, int ESZ> struct Mv2Tile;
template <> struct Mv2Tile<1, 2> { static constexpr int TW = 16, TH = 8; };     // 128 outputs, region 18 x 10
template <> struct Mv2Tile<2, 2> { static constexpr int TW = 8, TH = 4; };      //  32 outputs, region 17 x 9
template <> struct Mv2Tile<1, 4> { static constexpr int TW = 16, TH = 4; };     //  64 outputs, region 18 x 6
template <> struct Mv2Tile<2, 4> { static constexpr int TW = 8, TH = 2; };      //  16 outputs, region 17 x 5

// HID: hidden width (64 | 128); NT2 = Cout / 16 (1 | 2 | 4).  Static LDS: region pixels x (HID elements + 16 B of padding).
template <class T, int STRIDE, int HID, int NT2>
__global__ __launch_bounds__(256, STRIDE == 2 ? 3 : 4) void mv2_kernel(const Mv2Params p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC, KC = 4 * VEC;
    constexpr int TW = Mv2Tile<STRIDE, int(sizeof(T))>::TW, TH = Mv2Tile<STRIDE, int(sizeof(T))>::TH;
    constexpr int RW = TW * STRIDE + (STRIDE == 1 ? 2 : 1), RH = TH * STRIDE + (STRIDE == 1 ? 2 : 1), RP = RW * RH;
    constexpr int PITCH = HID * int(sizeof(T)) + 16;               // bytes per region pixel (padding: conflict-free 16-byte reads)
    constexpr int NT1 = HID / 16, KS2 = HID / KC;
    __shared__ __attribute__((aligned(16))) unsigned char hs[RP * PITCH];
    // the depthwise weights and bias ([9][HID] + [HID] floats, 2.5 / 5 KB) staged once per workgroup (round 4): phase 2 fetched a tap's weights from global memory
    // inside the tap loop, behind a lane-condition block per pixel tile — nine exposed L1 round trips per k-step (vmcnt(0) in front of every tap's FMAs)
    __shared__ __attribute__((aligned(16))) float wdl[10 * HID];
    for (int i = threadIdx.x; i < 10 * HID / 4; i += 256)
        reinterpret_cast<float4*>(wdl)[i] = i < 9 * HID / 4 ? reinterpret_cast<const float4*>(p.Wdw)[i] : reinterpret_cast<const float4*>(p.bdw)[i - 9 * HID / 4];
    const int lane = threadIdx.x & 63, wave = wave_uniform(int(threadIdx.x) >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);          // neighbouring tiles share halo lines: keep them in one L2
    const int bx = int(wg % unsigned(tiles_x)) * TW, by = int((wg / unsigned(tiles_x)) % unsigned(tiles_y)) * TH;
    const long b = wg / (unsigned(tiles_x) * unsigned(tiles_y));
    const int ry0 = by * STRIDE - 1, rx0 = bx * STRIDE - 1;        // region origin in the input map (3x3, pad 1)

    // ---- phase 1: hidden = SiLU(W1 x + b1) on the region, to LDS
    {
        const T* Xb = static_cast<const T*>(p.X) + b * p.H * long(p.Wd) * p.ldx;
        const uint4* W1 = static_cast<const uint4*>(p.W1) + lane;
        for (int pt = wave; pt * 16 < RP; pt += 4) {
            const int rp_raw = pt * 16 + px, rp = rp_raw < RP ? rp_raw : RP - 1;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int iy = ry0 + ry, ix = rx0 + rx;
            const bool in_map = iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd;
            const T* xp = Xb + (long(in_map ? iy : 0) * p.Wd + (in_map ? ix : 0)) * p.ldx;
            f32x4 acc[NT1];
            ACH_UNROLL
            for (int t = 0; t < NT1; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f; }
            for (int s = 0; s < p.k1; ++s) {
                // (k-slots past the pixel's stored channels meet zero weights; do not read the neighbouring pixel for them)
                // (an unconditional load from a clamped offset, zeroed afterwards: a load under a lane condition is waited for alone)
                const int xo = s * KC + g * VEC;
                const bool xin = xo < int(p.ldx);
                const uint4 xr = *reinterpret_cast<const uint4*>(xp + (xin ? xo : 0));
                const uint4 xf = make_uint4(xin ? xr.x : 0u, xin ? xr.y : 0u, xin ? xr.z : 0u, xin ? xr.w : 0u);
                ACH_UNROLL
                for (int t = 0; t < NT1; ++t) mfma16<T>(W1[(t * p.k1 + s) * 64], xf, acc[t]);
            }
            if (rp_raw < RP) {
                unsigned char* dst = hs + rp * PITCH;
                ACH_UNROLL
                for (int t = 0; t < NT1; ++t) {
                    const int ch = t * 16 + g * 4;
                    const float4 bb = *reinterpret_cast<const float4*>(p.b1 + ch);
                    float v[4] = {acc[t][0] + bb.x, acc[t][1] + bb.y, acc[t][2] + bb.z, acc[t][3] + bb.w};
                    ACH_UNROLL
                    for (int r = 0; r < 4; ++r) v[r] = in_map ? mv2_silu(v[r]) : 0.f;
                    Store<T>::st4(reinterpret_cast<T*>(dst) + ch, v);
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: depthwise 3x3 + SiLU from LDS -> B fragments -> projection
    constexpr int PT = TW * TH / 16;                               // 16-pixel tiles of the output tile
    // four or more pixel tiles: a wave owns tiles wave, wave + 4, .. and runs every k-step.  Fewer (the stride-2 tiles): the waves
    // split the K-STEPS of a tile between them as well, and the partial sums meet in LDS — otherwise half the workgroup idles here
    constexpr int KSPLIT = PT >= 4 ? 1 : 4 / PT, KPER = KS2 / KSPLIT;
    static_assert(KS2 % KSPLIT == 0, "k-steps must split evenly over the waves");
    constexpr int PPW = KSPLIT == 1 ? (PT + 3) / 4 : 1;           // pixel tiles per wave
    const int tile0 = KSPLIT == 1 ? wave : wave % PT, kpart = KSPLIT == 1 ? 0 : wave / PT;
    f32x4 acc[PPW][NT2];
    int rbase[PPW];                                                // LDS byte offset of the top-left tap of this lane's pixel
    ACH_UNROLL
    for (int q = 0; q < PPW; ++q) {
        ACH_UNROLL
        for (int t = 0; t < NT2; ++t) { acc[q][t][0] = acc[q][t][1] = acc[q][t][2] = acc[q][t][3] = 0.f; }
        const int pt = tile0 + 4 * q, o = pt * 16 + px;
        const int oy = o / TW, ox = o - oy * TW;
        rbase[q] = ((oy * STRIDE) * RW + ox * STRIDE) * PITCH;
    }
    const uint4* W2 = static_cast<const uint4*>(p.W2) + lane;
    ACH_NO_UNROLL
    for (int si = 0; si < KPER; ++si) {
        const int s = kpart * KPER + si;
        const int ch = s * KC + g * VEC;
        float bd[VEC];
        ACH_UNROLL
        for (int j = 0; j < VEC; j += 4) { const float4 w = *reinterpret_cast<const float4*>(wdl + 9 * HID + ch + j); bd[j] = w.x; bd[j + 1] = w.y; bd[j + 2] = w.z; bd[j + 3] = w.w; }
        uint4 wf[NT2];
        ACH_UNROLL
        for (int t = 0; t < NT2; ++t) wf[t] = W2[(t * KS2 + s) * 64];
        // taps are the OUTER loop inside a k-step: one tap's VEC weights are live at a time (8 registers instead of 72 for the whole 3x3 —
        // what kept the stride-1 instantiations at ~200 VGPRs, two waves per SIMD) and serve every pixel tile of the wave
        float a[PPW][VEC];
        ACH_UNROLL
        for (int q = 0; q < PPW; ++q)
            ACH_UNROLL
            for (int j = 0; j < VEC; ++j) a[q][j] = bd[j];
        ACH_UNROLL
        for (int k = 0; k < 9; ++k) {
            float wd[VEC];
            ACH_UNROLL
            for (int j = 0; j < VEC; j += 4) {
                const float4 w = *reinterpret_cast<const float4*>(wdl + k * HID + ch + j);
                wd[j] = w.x; wd[j + 1] = w.y; wd[j + 2] = w.z; wd[j + 3] = w.w;
            }
            ACH_UNROLL
            for (int q = 0; q < PPW; ++q) {
                if (tile0 + 4 * q >= PT) continue;
                float v[8];
                frag_unpack<T>(*reinterpret_cast<const uint4*>(hs + rbase[q] + ((k / 3) * RW + (k % 3)) * PITCH + ch * int(sizeof(T))), v);
                ACH_UNROLL
                for (int j = 0; j < VEC; ++j) a[q][j] += v[j] * wd[j];
            }
        }
        ACH_UNROLL
        for (int q = 0; q < PPW; ++q) {
            if (tile0 + 4 * q >= PT) continue;
            float h[8];
            ACH_UNROLL
            for (int j = 0; j < VEC; ++j) h[j] = mv2_silu(a[q][j]);
            const uint4 bf = frag_pack<T>(h);
            ACH_UNROLL
            for (int t = 0; t < NT2; ++t) mfma16<T>(wf[t], bf, acc[q][t]);
        }
    }
    if (KSPLIT > 1) {                                              // partial sums of the k-step groups -> wave group 0, through the (now free) LDS tile
        __syncthreads();
        f32x4* part = reinterpret_cast<f32x4*>(hs);
        if (kpart > 0) {
            ACH_UNROLL
            for (int t = 0; t < NT2; ++t) part[(((kpart - 1) * PT + tile0) * NT2 + t) * 64 + lane] = acc[0][t];
        }
        __syncthreads();
        if (kpart > 0) return;
        ACH_UNROLL
        for (int kp = 1; kp < KSPLIT; ++kp)
            ACH_UNROLL
            for (int t = 0; t < NT2; ++t) {
                const f32x4 v = part[(((kp - 1) * PT + tile0) * NT2 + t) * 64 + lane];
                acc[0][t][0] += v[0]; acc[0][t][1] += v[1]; acc[0][t][2] += v[2]; acc[0][t][3] += v[3];
            }
    }
    // ---- epilogue: + bias [+ residual], one 4-channel store per 16-channel tile
    ACH_UNROLL
    for (int q = 0; q < PPW; ++q) {
        const int pt = tile0 + 4 * q;
        if (pt >= PT) continue;
        const int o = pt * 16 + px, oyl = o / TW, oxl = o - oyl * TW;
        const int oy = by + oyl, ox = bx + oxl;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const long pix = (b * p.Ho + oy) * long(p.Wo) + ox;
        ACH_UNROLL
        for (int t = 0; t < NT2; ++t) {
            const int co = t * 16 + g * 4;
            if (co >= p.Cout) continue;
            const float4 bb = *reinterpret_cast<const float4*>(p.b2 + co);
            float v[4] = {acc[q][t][0] + bb.x, acc[q][t][1] + bb.y, acc[q][t][2] + bb.z, acc[q][t][3] + bb.w};
            if (p.R) {
                float r4[4];
                Store<T>::ld4(static_cast<const T*>(p.R) + pix * p.ldr + co, r4);
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) v[r] += r4[r];
            }
            Store<T>::st4(static_cast<T*>(p.Y) + pix * p.ldy + co, v);
        }
    }
}

template <class T>
inline bool launch_mv2(const Mv2Params& p, int stride, hipStream_t stream) {
    constexpr int ESZ = int(sizeof(T));
    const int TW = stride == 1 ? Mv2Tile<1, ESZ>::TW : Mv2Tile<2, ESZ>::TW, TH = stride == 1 ? Mv2Tile<1, ESZ>::TH : Mv2Tile<2, ESZ>::TH;
    const dim3 grid(unsigned((p.Wo + TW - 1) / TW) * unsigned((p.Ho + TH - 1) / TH) * unsigned(p.B)), block(256);
    const int nt2 = (p.Cout + 15) / 16;
#define ACH_MV2_CASE(st, hd, nt) if (stride == st && p.hid == hd && nt2 == nt) { ACH_LAUNCH((mv2_kernel<T, st, hd, nt>), grid, block, stream, p); return true; }
    ACH_MV2_CASE(1, 64, 2) ACH_MV2_CASE(2, 128, 2) ACH_MV2_CASE(1, 128, 2) ACH_MV2_CASE(2, 128, 4)
    ACH_MV2_CASE(1, 64, 1) ACH_MV2_CASE(2, 64, 2) ACH_MV2_CASE(2, 64, 1) ACH_MV2_CASE(1, 128, 4)
#undef ACH_MV2_CASE
    return false;
}
inline bool mv2_supported(int stride, int hid, int cout) {
    const int nt2 = (cout + 15) / 16;
    return (stride == 1 || stride == 2) && (hid == 64 || hid == 128) && cout % 4 == 0 && (nt2 == 1 || nt2 == 2 || nt2 == 4) && !(hid == 64 && nt2 == 4);
}

}  // namespace ach
