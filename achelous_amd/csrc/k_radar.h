// k_radar.h — the radar-map branch (RCNet: 8 x RCBlock, backbone/radar/RadarEncoder.py:38-109).
//
// Channel counts are tiny (3..72) and the maps are read by gathers, so this branch stays in planar NCHW
// (coalesced along x for every channel plane) and runs on the VALU with wave-uniform weight loads; no MFMA.
// Per block:  radar_offmask (AvgPool3x3 + offset conv 3x3 -> 18 + modulator conv 3x3 -> 9, 2*sigmoid)
//             radar_deform  (modulated deformable 3x3 sampling + contraction + 1x1(bias) + BN + ReLU + residual)
//             conv_planar   (weight_conv2: 1x1 or 3x3 stride 2)
// deform_conv2d semantics restated from torchvision 0.12.0 (see oracle/deform_conv.py header).
#pragma once
#include "ach_platform.h"

namespace ach {

struct OffMaskParams {
    const void* X;            // [B,C,H,W] block input
    void* pooled;             // [B,C,H,W] AvgPool2d(3,1,1) (count_include_pad) of X
    float* offmask;           // [B,27,H,W] fp32: 18 offsets (dy,dx interleaved per tap) then 9 masks (2*sigmoid)
    const float* W;           // [27][C][9]  (offset_conv rows 0..17, modulator_conv rows 18..26)
    const float* bias;        // [27]
    int B, C, H, Wd;
};

template <class T>
__global__ __launch_bounds__(256) void radar_offmask_kernel(const OffMaskParams p) {
    const long total = long(p.B) * p.H * p.Wd;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = int(idx % p.Wd);
    const int y = int((idx / p.Wd) % p.H);
    const long b = idx / (long(p.Wd) * p.H);
    const long HW = long(p.H) * p.Wd;
    float acc[27];
    ACH_UNROLL
    for (int o = 0; o < 27; ++o) acc[o] = p.bias[o];
    for (int c = 0; c < p.C; ++c) {
        const T* plane = static_cast<const T*>(p.X) + (b * p.C + c) * HW;
        float win[5][5];
        ACH_UNROLL
        for (int dy = 0; dy < 5; ++dy)
            ACH_UNROLL
            for (int dx = 0; dx < 5; ++dx) {
                const int iy = y + dy - 2, ix = x + dx - 2;
                win[dy][dx] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd) ? Store<T>::ld(plane + long(iy) * p.Wd + ix) : 0.f;
            }
        // pooled value at each of the 3x3 conv taps; a tap outside the map is the conv's zero padding
        float pl[9];
        ACH_UNROLL
        for (int ky = 0; ky < 3; ++ky)
            ACH_UNROLL
            for (int kx = 0; kx < 3; ++kx) {
                const int py = y + ky - 1, px = x + kx - 1;
                float s = 0.f;
                ACH_UNROLL
                for (int a = 0; a < 3; ++a)
                    ACH_UNROLL
                    for (int bb = 0; bb < 3; ++bb) s += win[ky + a][kx + bb];
                pl[ky * 3 + kx] = (py >= 0 && py < p.H && px >= 0 && px < p.Wd) ? s * (1.0f / 9.0f) : 0.f;
            }
        Store<T>::st(static_cast<T*>(p.pooled) + (b * p.C + c) * HW + long(y) * p.Wd + x, pl[4]);
        const float* w = p.W + long(c) * 9;
        ACH_UNROLL
        for (int o = 0; o < 27; ++o) {
            const float* wo = w + long(o) * p.C * 9;
            float s = 0.f;
            ACH_UNROLL
            for (int k = 0; k < 9; ++k) s += wo[k] * pl[k];
            acc[o] += s;
        }
    }
    float* om = p.offmask + b * 27 * HW + long(y) * p.Wd + x;
    ACH_UNROLL
    for (int o = 0; o < 18; ++o) om[o * HW] = acc[o];
    ACH_UNROLL
    for (int o = 18; o < 27; ++o) om[o * HW] = 2.0f * sigmoidf_(acc[o]);
}

struct DeformParams {
    const void* pooled;       // [B,C,H,W]  sampled tensor (the block's avg-pooled input)
    const float* offmask;     // [B,27,H,W]
    const void* res;          // [B,C,H,W]  block input (residual)
    void* Y;                  // [B,C,H,W]  relu(bn(conv1x1(dcn))) + res
    const float* Wd3;         // regular_conv [C][C][9]
    const float* W1;          // weight_conv1 [C][C] with BN scale folded
    const float* b1;          // [C] (bias and BN shift folded)
    int B, H, Wd;
};

template <class T, int C>
__global__ __launch_bounds__(256) void radar_deform_kernel(const DeformParams p) {
    const long total = long(p.B) * p.H * p.Wd;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = int(idx % p.Wd);
    const int y = int((idx / p.Wd) % p.H);
    const long b = idx / (long(p.Wd) * p.H);
    const long HW = long(p.H) * p.Wd;
    const float* om = p.offmask + b * 27 * HW + long(y) * p.Wd + x;
    const T* base = static_cast<const T*>(p.pooled) + b * C * HW;
    float acc[C];
    ACH_UNROLL
    for (int o = 0; o < C; ++o) acc[o] = 0.f;
    for (int k = 0; k < 9; ++k) {
        const float sy = float(y - 1 + k / 3) + om[(2 * k) * HW];
        const float sx = float(x - 1 + k % 3) + om[(2 * k + 1) * HW];
        const float mk = om[(18 + k) * HW];
        if (!(sy > -1.f && sy < float(p.H) && sx > -1.f && sx < float(p.Wd))) continue;   // sample is 0
        const float fy = floorf(sy), fx = floorf(sx);
        const int y0 = int(fy), x0 = int(fx), y1 = y0 + 1, x1 = x0 + 1;
        const float ly = sy - fy, lx = sx - fx, hy = 1.f - ly, hx = 1.f - lx;
        const bool oky0 = y0 >= 0, oky1 = y1 <= p.H - 1, okx0 = x0 >= 0, okx1 = x1 <= p.Wd - 1;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const long o00 = long(y0) * p.Wd + x0;
        for (int c = 0; c < C; ++c) {
            const T* pl = base + c * HW;
            const float v1 = (oky0 && okx0) ? Store<T>::ld(pl + o00) : 0.f;
            const float v2 = (oky0 && okx1) ? Store<T>::ld(pl + o00 + 1) : 0.f;
            const float v3 = (oky1 && okx0) ? Store<T>::ld(pl + o00 + p.Wd) : 0.f;
            const float v4 = (oky1 && okx1) ? Store<T>::ld(pl + o00 + p.Wd + 1) : 0.f;
            const float v = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * mk;
            const float* w = p.Wd3 + long(c) * 9 + k;
            ACH_UNROLL
            for (int o = 0; o < C; ++o) acc[o] += w[long(o) * C * 9] * v;
        }
    }
    const long pix = long(y) * p.Wd + x;
    for (int o = 0; o < C; ++o) {
        float s = p.b1[o];
        ACH_UNROLL
        for (int c = 0; c < C; ++c) s += p.W1[o * C + c] * acc[c];
        s = s > 0.f ? s : 0.f;
        s += Store<T>::ld(static_cast<const T*>(p.res) + (b * C + o) * HW + pix);
        Store<T>::st(static_cast<T*>(p.Y) + (b * C + o) * HW + pix, s);
    }
}

template <class T>
inline bool launch_radar_deform(const DeformParams& p, int C, hipStream_t s) {
    const long total = long(p.B) * p.H * p.Wd;
    const dim3 grid(unsigned(cdivl(total, 256))), block(256);
    switch (C) {
        case 3: ACH_LAUNCH((radar_deform_kernel<T, 3>), grid, block, s, p); return true;
        case 8: ACH_LAUNCH((radar_deform_kernel<T, 8>), grid, block, s, p); return true;
        case 12: ACH_LAUNCH((radar_deform_kernel<T, 12>), grid, block, s, p); return true;
        case 16: ACH_LAUNCH((radar_deform_kernel<T, 16>), grid, block, s, p); return true;
        case 24: ACH_LAUNCH((radar_deform_kernel<T, 24>), grid, block, s, p); return true;
        case 30: ACH_LAUNCH((radar_deform_kernel<T, 30>), grid, block, s, p); return true;
        case 36: ACH_LAUNCH((radar_deform_kernel<T, 36>), grid, block, s, p); return true;
        default: return false;
    }
}

// planar direct convolution (k = 1 or 3, stride 1 or 2, zero padding k/2): CO output channels per thread
struct ConvPlanarParams {
    const void* X; void* Y;
    const float* W;           // [Cout][Cin][k*k]
    const float* bias;        // [Cout]
    int B, Cin, H, Wd, Cout, Ho, Wo, k, stride, act;
};
template <class T, int CO>
__global__ __launch_bounds__(256) void conv_planar_kernel(const ConvPlanarParams p) {
    const long total = long(p.B) * p.Ho * p.Wo;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co0 = blockIdx.y * CO;
    const int ox = int(idx % p.Wo);
    const int oy = int((idx / p.Wo) % p.Ho);
    const long b = idx / (long(p.Wo) * p.Ho);
    const long HW = long(p.H) * p.Wd;
    const int pad = p.k / 2, kk = p.k * p.k;
    float acc[CO];
    ACH_UNROLL
    for (int o = 0; o < CO; ++o) acc[o] = (co0 + o < p.Cout) ? p.bias[co0 + o] : 0.f;
    for (int c = 0; c < p.Cin; ++c) {
        const T* pl = static_cast<const T*>(p.X) + (b * p.Cin + c) * HW;
        for (int ky = 0; ky < p.k; ++ky) {
            const int iy = oy * p.stride - pad + ky;
            if (iy < 0 || iy >= p.H) continue;
            for (int kx = 0; kx < p.k; ++kx) {
                const int ix = ox * p.stride - pad + kx;
                if (ix < 0 || ix >= p.Wd) continue;
                const float v = Store<T>::ld(pl + long(iy) * p.Wd + ix);
                const float* w = p.W + (long(co0) * p.Cin + c) * kk + ky * p.k + kx;
                ACH_UNROLL
                for (int o = 0; o < CO; ++o)
                    if (co0 + o < p.Cout) acc[o] += w[long(o) * p.Cin * kk] * v;
            }
        }
    }
    const long OHW = long(p.Ho) * p.Wo;
    ACH_UNROLL
    for (int o = 0; o < CO; ++o)
        if (co0 + o < p.Cout)
            Store<T>::st(static_cast<T*>(p.Y) + (b * p.Cout + co0 + o) * OHW + long(oy) * p.Wo + ox, apply_act(acc[o], p.act));
}

}  // namespace ach
