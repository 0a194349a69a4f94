// k_radar.h — the radar-map branch (RCNet: 8 x RCBlock, backbone/radar/RadarEncoder.py:38-109).
//
// Layout: NHWC with the channel count padded to a multiple of 8 (3 -> 8, 12 -> 16, 44 -> 48; padding lanes stay 0),
// so that a bilinear corner of the deformable sampling is ONE vector load for all channels and the dense 3x3
// convolutions (offset/modulator conv, stride-2 weight_conv2) run as implicit GEMMs on MFMA (k_gemm.h conv mode).
// Per RCBlock:
//   avgpool3x3        AvgPool2d(3,1,1), count_include_pad                                          (VALU, streaming)
//   gemm conv3x3      offset_conv (18) + modulator_conv (9) in one 27-wide implicit GEMM          (MFMA)
//   deform            modulated deformable 3x3 sampling (torchvision 0.12.0 semantics, oracle/deform_conv.py)
//                     + [regular_conv folded with weight_conv1 and BatchNorm] + ReLU + residual:
//                       C <= 8 : one fused kernel, contraction on the VALU with wave-uniform weights
//                       C >= 12: sampling kernel writes the 9*Cp "columns", contraction is an MFMA GEMM
//   gemm              weight_conv2: 1x1, or 3x3 stride 2 as implicit GEMM                         (MFMA)
#pragma once
#include "ach_platform.h"

namespace ach {

// NCHW [B,C,H,W] -> NHWC [B,H,W,ld] (channels C..ld-1 are left untouched = 0)
struct ToNhwcParams { const void* X; void* Y; int B, C, H, Wd; long ld; };
template <class T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const ToNhwcParams p) {
    const long total = long(p.B) * p.H * p.Wd;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long HW = long(p.H) * p.Wd;
    const long b = idx / HW, pix = idx - b * HW;
    const T* x = static_cast<const T*>(p.X) + b * p.C * HW + pix;
    T* y = static_cast<T*>(p.Y) + idx * p.ld;
    for (int c = 0; c < p.C; ++c) y[c] = x[c * HW];
}

// AvgPool2d(3, stride 1, pad 1, count_include_pad=True): always divides by 9
struct PoolParams { const void* X; long ldx; void* Y; long ldy; int B, H, Wd, C; };
template <class T>
__global__ __launch_bounds__(256) void avgpool3x3_kernel(const PoolParams p) {
    const int cq = (p.C + 3) >> 2;
    const long total = long(p.B) * p.H * p.Wd * cq;
    const long idx = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = int(idx % cq) * 4;
    long pix = idx / cq;
    const int x = int(pix % p.Wd); pix /= p.Wd;
    const int y = int(pix % p.H);
    const long b = pix / p.H;
    const T* X = static_cast<const T*>(p.X);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // nine unconditional loads from clamped addresses (all in flight together); taps outside the map are zeroed afterwards
    float v[9][4];
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) {
        const int iy = y + k / 3 - 1, ix = x + k % 3 - 1;
        const int cy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy), cx = ix < 0 ? 0 : (ix >= p.Wd ? p.Wd - 1 : ix);
        Store<T>::ld4(X + ((b * p.H + cy) * p.Wd + cx) * p.ldx + c, v[k]);
    }
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) {
        const int iy = y + k / 3 - 1, ix = x + k % 3 - 1;
        const float ok = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd) ? 1.f : 0.f;
        acc[0] += v[k][0] * ok; acc[1] += v[k][1] * ok; acc[2] += v[k][2] * ok; acc[3] += v[k][3] * ok;
    }
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) acc[i] *= (1.0f / 9.0f);
    Store<T>::st4(static_cast<T*>(p.Y) + ((b * p.H + y) * p.Wd + x) * p.ldy + c, acc);
}

// ---- shared by both deformable kernels: one tap's bilinear footprint (zero outside, torchvision semantics)
struct BilinearTap {
    long o00, o01, o10, o11;      // pixel indices of the four corners (clamped into the map)
    float w00, w01, w10, w11;     // bilinear weights, already 0 for corners outside the map and multiplied by the mask
};
__device__ __forceinline__ BilinearTap make_tap(float sy, float sx, float mask, int H, int Wd, long base) {
    BilinearTap t;
    const bool inside = sy > -1.f && sy < float(H) && sx > -1.f && sx < float(Wd);
    const float fy = floorf(sy), fx = floorf(sx);
    const int y0 = int(fy), x0 = int(fx), y1 = y0 + 1, x1 = x0 + 1;
    const float ly = sy - fy, lx = sx - fx, hy = 1.f - ly, hx = 1.f - lx;
    const float m = inside ? mask : 0.f;
    const bool oy0 = y0 >= 0 && y0 <= H - 1, oy1 = y1 >= 0 && y1 <= H - 1, ox0 = x0 >= 0 && x0 <= Wd - 1, ox1 = x1 >= 0 && x1 <= Wd - 1;
    t.w00 = (oy0 && ox0) ? hy * hx * m : 0.f;
    t.w01 = (oy0 && ox1) ? hy * lx * m : 0.f;
    t.w10 = (oy1 && ox0) ? ly * hx * m : 0.f;
    t.w11 = (oy1 && ox1) ? ly * lx * m : 0.f;
    const int cy0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0), cy1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    const int cx0 = x0 < 0 ? 0 : (x0 > Wd - 1 ? Wd - 1 : x0), cx1 = x1 < 0 ? 0 : (x1 > Wd - 1 ? Wd - 1 : x1);
    t.o00 = base + long(cy0) * Wd + cx0; t.o01 = base + long(cy0) * Wd + cx1;
    t.o10 = base + long(cy1) * Wd + cx0; t.o11 = base + long(cy1) * Wd + cx1;
    return t;
}

struct DeformParams {
    const void* pooled; long ldp;     // sampled tensor (avg-pooled block input), NHWC
    const void* om; long ldo;         // [pixels, 27]: 18 offsets (dy,dx per tap) + 9 modulator logits
    const void* res; long ldr;        // block input (residual)
    void* Y; long ldy;                // fused: relu(Wf . col + bf) + res ; sample: the columns [pixels, 9*Cp]
    const float* Wf;                  // fused: [C][9][Cp] folded (weight_conv1 . BN . regular_conv)
    const float* bf;                  // fused: [C]
    int B, H, Wd, Cp;
};

// C <= 8: sampling + contraction + ReLU + residual in one pass, one thread per pixel
template <class T, int C, int CP>
__global__ __launch_bounds__(256) void deform_fused_kernel(const DeformParams p) {
    const long total = long(p.B) * p.H * p.Wd;
    const long idx = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = int(idx % p.Wd);
    const int y = int((idx / p.Wd) % p.H);
    const long b = idx / (long(p.Wd) * p.H);
    const T* om = static_cast<const T*>(p.om) + idx * p.ldo;
    float o[28];
    ACH_UNROLL
    for (int i = 0; i < 7; ++i) { float v[4]; Store<T>::ld4(om + i * 4, v); o[i * 4] = v[0]; o[i * 4 + 1] = v[1]; o[i * 4 + 2] = v[2]; o[i * 4 + 3] = v[3]; }
    const T* P0 = static_cast<const T*>(p.pooled);
    float acc[C];
    ACH_UNROLL
    for (int i = 0; i < C; ++i) acc[i] = p.bf[i];
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) {
        const BilinearTap t = make_tap(float(y - 1 + k / 3) + o[2 * k], float(x - 1 + k % 3) + o[2 * k + 1],
                                       2.0f * sigmoidf_(o[18 + k]), p.H, p.Wd, b * p.H * long(p.Wd));
        float v[CP];
        ACH_UNROLL
        for (int c = 0; c < CP; c += 4) {
            float a[4], bq[4], cc[4], d[4];
            Store<T>::ld4(P0 + t.o00 * p.ldp + c, a);
            Store<T>::ld4(P0 + t.o01 * p.ldp + c, bq);
            Store<T>::ld4(P0 + t.o10 * p.ldp + c, cc);
            Store<T>::ld4(P0 + t.o11 * p.ldp + c, d);
            ACH_UNROLL
            for (int i = 0; i < 4; ++i) v[c + i] = t.w00 * a[i] + t.w01 * bq[i] + t.w10 * cc[i] + t.w11 * d[i];
        }
        const float* w = p.Wf + k * p.Cp;                         // host layout [C][9][Cp], Cp = channel stride of the buffers
        ACH_UNROLL
        for (int co = 0; co < C; ++co)
            ACH_UNROLL
            for (int c = 0; c < C; ++c) acc[co] += w[co * 9 * p.Cp + c] * v[c];
    }
    float outv[CP];
    ACH_UNROLL
    for (int i = 0; i < CP; ++i) outv[i] = 0.f;
    ACH_UNROLL
    for (int i = 0; i < C; ++i) outv[i] = acc[i] > 0.f ? acc[i] : 0.f;
    const T* r = static_cast<const T*>(p.res) + idx * p.ldr;
    T* yo = static_cast<T*>(p.Y) + idx * p.ldy;
    ACH_UNROLL
    for (int c = 0; c < CP; c += 4) {
        float rr[4], ov[4];
        Store<T>::ld4(r + c, rr);                 // padding lanes of the residual are zero
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) ov[i] = outv[c + i] + rr[i];
        Store<T>::st4(yo + c, ov);
    }
}

// generic: one thread per (pixel, tap, 4-channel group) writes the masked bilinear sample into the column buffer
template <class T>
__global__ __launch_bounds__(256) void deform_sample_kernel(const DeformParams p) {
    const int cq = p.Cp >> 2;
    const long total = long(p.B) * p.H * p.Wd * 9 * cq;
    const long idx = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = int(idx % cq) * 4;
    long r = idx / cq;
    const int k = int(r % 9);
    const long pix = r / 9;
    const int x = int(pix % p.Wd);
    const int y = int((pix / p.Wd) % p.H);
    const long b = pix / (long(p.Wd) * p.H);
    const T* om = static_cast<const T*>(p.om) + pix * p.ldo;
    const float dy = Store<T>::ld(om + 2 * k), dx = Store<T>::ld(om + 2 * k + 1), ml = Store<T>::ld(om + 18 + k);
    const BilinearTap t = make_tap(float(y - 1 + k / 3) + dy, float(x - 1 + k % 3) + dx, 2.0f * sigmoidf_(ml), p.H, p.Wd, b * p.H * long(p.Wd));
    const T* P0 = static_cast<const T*>(p.pooled);
    float a[4], bq[4], cc[4], d[4], v[4];
    Store<T>::ld4(P0 + t.o00 * p.ldp + c, a);
    Store<T>::ld4(P0 + t.o01 * p.ldp + c, bq);
    Store<T>::ld4(P0 + t.o10 * p.ldp + c, cc);
    Store<T>::ld4(P0 + t.o11 * p.ldp + c, d);
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) v[i] = t.w00 * a[i] + t.w01 * bq[i] + t.w10 * cc[i] + t.w11 * d[i];
    Store<T>::st4(static_cast<T*>(p.Y) + pix * p.ldy + k * p.Cp + c, v);
}

}  // namespace ach
