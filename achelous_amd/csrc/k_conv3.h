// k_conv3.h — dense 3x3 / stride 1 / pad 1 convolution over a narrow NHWC map (a few 16-byte vectors per pixel), N <= 32
// outputs: the offset + modulator conv of every RCBlock (radar_models/dcn.py:24-47: 18 + 9 = 27 channels), which at
// 320x320 and 160x160 is the largest MFMA workload of the radar branch.
//
// The generic implicit-GEMM mode of gemm_kernel (k_gemm.h) spends ~700 VALU+SALU instructions per 16 pixels on this shape
// (64-bit divisions to recover (b, y, x) from the row index, tap decoding per k-step, weight fragment loads) for six MFMAs:
// rocprofv3 showed it VALU-issue bound at 1.4 TB/s.  Here a workgroup owns one output ROW: (b, y) come from the workgroup id
// (scalar), the tap geometry of a lane's k-slots is computed once, the weight fragments stay in registers, and each wave
// walks the row 16 pixels at a time — per tile only the loads, MFMAs and one wide store remain.  The input carries a
// one-pixel zero border in memory (its producer, avgpool3x3, writes into such a buffer), so there is no bounds logic either.
// K is ordered (tap, channel vector) exactly as conv_lin packs it, so the weights are the ones gemm_kernel would use.
#pragma once
#include "ach_platform.h"
#include "k_gemm.h"

namespace ach {

struct Conv3Params {
    const void* X; long ldx;        // NHWC input, ldx = cv * VEC, with a one-pixel ZERO BORDER around every sample: X is pixel (0,0) of
    long xpr, xpi;                  //   sample 0, xpr / xpi the row / image pitches in elements — the conv's zero padding is in memory
    void* Y; long ldy;              // NHWC output [B,H,W,ldy], ldy >= 32
    const void* W; const float* bias;   // packed NT = 2, one chunk, KS k-steps ; bias[32]
    int B, H, Wd, cv, act;
};

template <class T, int KS>
__global__ __launch_bounds__(256) void conv3x3_rows_kernel(const Conv3Params p) {
    constexpr int VEC = Store<T>::VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);          // consecutive rows of a sample share an L2 (halo rows)
    const int b = int(wg / unsigned(p.H)), oy = int(wg - unsigned(b) * unsigned(p.H));
    const int ldx = int(p.ldx);

    int off[KS];
    bool live[KS];
    ACH_UNROLL
    for (int s = 0; s < KS; ++s) {
        const int q = 4 * s + g;
        const int tap = q / p.cv, c = q - tap * p.cv;
        const int ty = tap / 3, tx = tap - 3 * ty;
        live[s] = tap < 9;                                             // k-slots past the ninth tap: zero weights, but keep the address in bounds
        off[s] = live[s] ? (ty - 1) * int(p.xpr) + (tx - 1) * ldx + c * VEC : 0;
    }
    const uint4* Wf = static_cast<const uint4*>(p.W) + lane;
    uint4 wf[KS][2];
    ACH_UNROLL
    for (int s = 0; s < KS; ++s) { wf[s][0] = Wf[(s * 2) * 64]; wf[s][1] = Wf[(s * 2 + 1) * 64]; }
    float bv[8];
    ACH_UNROLL
    for (int i = 0; i < 8; ++i) bv[i] = p.bias[g * 8 + i];

    const T* xrow = static_cast<const T*>(p.X) + long(b) * p.xpi + long(oy) * p.xpr;
    T* yrow = static_cast<T*>(p.Y) + (long(b) * p.H + oy) * p.Wd * p.ldy + g * 8;
    for (int tile = wave; tile * 16 < p.Wd; tile += 4) {
        const int xr = tile * 16 + px;
        const bool valid = xr < p.Wd;
        const int x = valid ? xr : p.Wd - 1;
        const T* xp = xrow + long(x) * ldx;
        uint4 xf[KS];
        ACH_UNROLL
        for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const uint4*>(xp + off[s]);
        f32x4 a0, a1;
        a0[0] = a0[1] = a0[2] = a0[3] = 0.f;
        a1[0] = a1[1] = a1[2] = a1[3] = 0.f;
        ACH_UNROLL
        for (int s = 0; s < KS; ++s) { mfma16<T>(wf[s][0], xf[s], a0); mfma16<T>(wf[s][1], xf[s], a1); }
        float o[8];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { o[r] = apply_act_t<T>(a0[r] + bv[r], p.act); o[4 + r] = apply_act_t<T>(a1[r] + bv[4 + r], p.act); }
        if (valid) Store<T>::st8(yrow + long(x) * p.ldy, o);
    }
}

template <class T>
inline bool launch_conv3(const Conv3Params& p, int ksteps, hipStream_t stream) {
    const dim3 grid(unsigned(p.B) * unsigned(p.H)), block(256);
    if (ksteps == 3) { ACH_LAUNCH((conv3x3_rows_kernel<T, 3>), grid, block, stream, p); return true; }
    if (ksteps == 5) { ACH_LAUNCH((conv3x3_rows_kernel<T, 5>), grid, block, stream, p); return true; }
    if (ksteps == 9) { ACH_LAUNCH((conv3x3_rows_kernel<T, 9>), grid, block, stream, p); return true; }
    return false;
}

}  // namespace ach
