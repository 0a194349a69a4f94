// k_sdta.h — the front of an SDTA encoder as ONE launch (backbone/edgenext_utils/sdta_encoder.py:39-58):
//
//     spx = split(x, width);   sp_0 = conv_0(spx_0);   sp_i = conv_i(sp_{i-1} + spx_i)   (depthwise 3x3, bias);
//     y = cat(sp_0 .. sp_{nums-1}, spx_nums) [+ positional encoding]
//
// It was nums depthwise launches + a tail copy + an in-place positional add (3-5 launches of 7-14 us on maps of 10 x 10 .. 40 x 40: launch
// floors on the caller's stream, the step's critical path).  Channel j of split i only ever meets channel j of the other splits, and the maps
// are small: a workgroup owns (frame, Q quads of four channel positions) with the WHOLE map in LDS and walks the splits in order —
// no halo exchange, no intermediate in HBM.  Workgroups past the conv quads copy the tail split.  Rounding points are the separate launches'
// (each sp_i is rounded to the storage type where it was stored; the positional term is added to the rounded value): bit-identical.
#pragma once
#include "ach_platform.h"

namespace ach {

template <class T> __device__ __forceinline__ float round_to(float v);                  // the value a store of v to the storage type reads back as
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }
template <> __device__ __forceinline__ float round_to<f16_t>(float v) { return f16_to_f32(f32_to_f16(v)); }

struct SdtaPreParams {
    const void* X; long ldx;
    void* Y; long ldy;
    const float* W;          // [nums][9][width]
    const float* bias;       // [nums][width]
    const float* posenc;     // [H*W][C] fp32 or nullptr
    int B, H, Wd, C, width, nums;
    int Q;                   // quads per workgroup (1, 2, 4 or 8): SDTA_THREADS / Q pixel slots
    int conv_wgs, tail_wgs;  // workgroups per frame: conv quads / Q, tail quads / Q (rounded up)
};
#ifndef ACH_SDTA_THREADS
#define ACH_SDTA_THREADS 1024
#define ACH_SDTA_MAXPPT 4
#define ACH_SDTA_LDS 14336
#endif
constexpr int SDTA_THREADS = ACH_SDTA_THREADS;
constexpr int SDTA_MAXPPT = ACH_SDTA_MAXPPT;            // pixels per thread (H * W <= 4 * 1024 / Q): the cascade's previous outputs stay in registers
constexpr int SDTA_LDS_FLOATS = ACH_SDTA_LDS;    // 56 KB: H * W * Q float4

template <class T>
__global__ __launch_bounds__(SDTA_THREADS) void sdta_pre_kernel(const SdtaPreParams p) { f16_sat_mode<T>();
    __shared__ __attribute__((aligned(16))) float s[SDTA_LDS_FLOATS];
    __shared__ float4 wl[10 * 8];            // [tap 0..8, bias][quad]
    const int HW = p.H * p.Wd, per = p.conv_wgs + p.tail_wgs;
    const int b = int(blockIdx.x) / per, wgi = int(blockIdx.x) % per;
    const int Q = p.Q, q = int(threadIdx.x) % Q, slot = int(threadIdx.x) / Q, slots = SDTA_THREADS / Q;
    const T* X = static_cast<const T*>(p.X) + long(b) * HW * p.ldx;
    T* Y = static_cast<T*>(p.Y) + long(b) * HW * p.ldy;
    if (wgi >= p.conv_wgs) {              // tail split: copy (+ positional encoding)
        const int tq = (p.C - p.nums * p.width) / 4, j = (wgi - p.conv_wgs) * Q + q;
        if (j >= tq) return;
        const int c = p.nums * p.width + 4 * j;
        for (int px = slot; px < HW; px += slots) {
            float a[4];
            Store<T>::ld4(X + long(px) * p.ldx + c, a);
            if (p.posenc) { const float* pe = p.posenc + long(px) * p.C + c; ACH_UNROLL for (int i = 0; i < 4; ++i) a[i] += pe[i]; }
            Store<T>::st4(Y + long(px) * p.ldy + c, a);
        }
        return;
    }
    const int j = wgi * Q + q;                              // channel position 4j .. 4j+3 inside every split
    const bool live = j < p.width / 4;
    // the map sits in LDS with a one-pixel ZERO border (the conv's padding): taps are unconditional reads at constant offsets
    const int WB = p.Wd + 2, nb = (p.H + 2) * WB;
    for (int i0 = int(threadIdx.x); i0 < nb * Q; i0 += SDTA_THREADS) {
        const int pos = i0 / Q, by = pos / WB, bx = pos % WB;
        if (by == 0 || by == p.H + 1 || bx == 0 || bx == p.Wd + 1) *reinterpret_cast<float4*>(s + long(i0) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // this thread's pixels: slot, slot + slots, ... as (row, column) without a division per pixel
    int oy[SDTA_MAXPPT], ox[SDTA_MAXPPT];
    {
        const int dy = slots / p.Wd, dx = slots % p.Wd;
        int y = slot / p.Wd, x = slot % p.Wd;
        ACH_UNROLL
        for (int k = 0; k < SDTA_MAXPPT; ++k) { oy[k] = y; ox[k] = x; y += dy; x += dx; if (x >= p.Wd) { x -= p.Wd; ++y; } }
    }
    float yr[SDTA_MAXPPT][4];
    for (int i = 0; i < p.nums; ++i) {
        const int c = i * p.width + 4 * j;
        // stage sp_{i-1} + spx_i of the whole map
        ACH_UNROLL
        for (int k = 0; k < SDTA_MAXPPT; ++k) {
            const int px = slot + k * slots;
            if (px >= HW) break;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                Store<T>::ld4(X + long(px) * p.ldx + c, a);
                if (i > 0) { ACH_UNROLL for (int e = 0; e < 4; ++e) a[e] += yr[k][e]; }
            }
            *reinterpret_cast<float4*>(s + (long((oy[k] + 1) * WB + ox[k] + 1) * Q + q) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
        if (int(threadIdx.x) < 10 * Q) {             // the split's nine weight vectors and its bias per quad: LDS, not 37 registers per thread
            const int t = int(threadIdx.x) / Q, qq = int(threadIdx.x) % Q, jj = wgi * Q + qq;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jj < p.width / 4) w = *reinterpret_cast<const float4*>((t < 9 ? p.W + (long(i) * 9 + t) * p.width : p.bias + long(i) * p.width) + 4 * jj);
            wl[t * Q + qq] = w;
        }
        __syncthreads();
        ACH_UNROLL
        for (int k = 0; k < SDTA_MAXPPT; ++k) {
            const int px = slot + k * slots;
            if (px >= HW) break;
            if (!live) continue;
            const float* base = s + (long(oy[k] * WB + ox[k]) * Q + q) * 4;          // tap (0, 0) of the bordered map
            const float4 bb = wl[9 * Q + q];
            float acc[4] = {bb.x, bb.y, bb.z, bb.w};
            ACH_UNROLL
            for (int ky = 0; ky < 3; ++ky) {
                ACH_UNROLL
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = *reinterpret_cast<const float4*>(base + long(ky * WB + kx) * Q * 4);
                    const float4 w = wl[(ky * 3 + kx) * Q + q];
                    acc[0] += v.x * w.x; acc[1] += v.y * w.y; acc[2] += v.z * w.z; acc[3] += v.w * w.w;
                }
            }
            ACH_UNROLL
            for (int e = 0; e < 4; ++e) yr[k][e] = round_to<T>(acc[e]);            // what the separate launch stored and the next split read back
            float o[4] = {yr[k][0], yr[k][1], yr[k][2], yr[k][3]};
            if (p.posenc) { const float* pe = p.posenc + long(px) * p.C + c; ACH_UNROLL for (int e = 0; e < 4; ++e) o[e] += pe[e]; }
            Store<T>::st4(Y + long(px) * p.ldy + c, o);
        }
        __syncthreads();
    }
}

// quads per workgroup: as many as the LDS tile and the per-thread pixel registers allow (adjacent lanes then read adjacent 8 / 16 bytes)
inline int sdta_pre_quads(int H, int Wd, int wq) {
    const int HW = H * Wd, bordered = (H + 2) * (Wd + 2);
    for (int Q = 8; Q >= 1; Q >>= 1)
        if (long(bordered) * Q * 4 <= SDTA_LDS_FLOATS && HW <= SDTA_MAXPPT * (SDTA_THREADS / Q) && (Q == 1 || Q / 2 < wq)) return Q;
    return 0;
}

}  // namespace ach
