// k_headdw.h — one layer of the detection head's two towers, fused (bf16 engine, nano head; round 3):
//
//     y[:, br*64 : br*64+64] = relu( Wp_br · dw5x5(x_br) + b_br )        br = cls / reg       head/decouplehead.py:38-52, 70-92
//                                                                        (BaseConv with ds_conv: dw 5x5 -> pw 1x1 -> BN -> ReLU,
//                                                                         backbone/conv_utils/normal_conv.py:23-52)
// It was two launches per layer for the three pyramid levels: `dwconv_strip_multi<5>` — every input element fetched 25 times through the
// texture path (TA 49 % busy, 58-64 us, and the reason the decoders' tail stretches beside it) — writing a 128-channel tensor that the
// block-diagonal 128 x 128 GEMM (31 us) read straight back.  Here a workgroup (256 threads, see ACH_HDW_THREADS) owns (frame, band of rows, tower):
//   0. the band's halo (rows + 4, columns + 4, the tower's 64 input channels) is staged in LDS once, unpacked to fp32;
//   1. depthwise 5 x 5 from LDS: thread = 5-pixel strip x 4 channels (a tap row: 9 LDS reads for 5 x 5 x 4 FMAs); the sums are written
//      back to LDS as the bf16 B fragments of the pointwise GEMM — the depthwise output never exists in HBM;
//   2. the tower's 64 x 64 pointwise conv on MFMA (weights: 8 fragments, register-resident), + bias, ReLU, 16-byte stores.
// The three levels are one launch (a job per level: 40 x 40 in 4-row bands, 20 x 20 in 5-row bands, 10 x 10 whole).
#pragma once
#include "ach_platform.h"

namespace ach {

#ifndef ACH_HDW_THREADS
#define ACH_HDW_THREADS 256           // measured (one box, alternating, three runs): 512 threads 37.57 k frames/s (60 us isolated), 256: 38.55 k (47 us), 128: 38.10 k (70 us).
                                      // (The first comparison of this knob measured a kernel that the fp32 engine's translation unit had ALSO instantiated: the runtime ran
                                      // that copy — compiled for 512 threads — under a 128-thread launch, i.e. a quarter of the work.  The launch is behind `if constexpr` now.)
#endif
#ifndef ACH_HDW_WGS
#define ACH_HDW_WGS 4
#endif
#ifndef ACH_HDW_ALIAS
#define ACH_HDW_ALIAS 1               // the B fragments are written over the halo tile (the depthwise sums wait in registers across a barrier): 45 KB of LDS, three workgroups per CU; 49 -> 43 us, +0.7 % end to end
#endif
constexpr int HDW_MAXIT = 1024 / ACH_HDW_THREADS;        // strips x channel quads per thread in the aliased form (engine: rows * strips * 16 <= 1024)
#ifndef ACH_HDW_HALO_ROWS
#define ACH_HDW_HALO_ROWS 8           // rows of the staged halo tile (44 columns): 8 = 4-row bands on the 40 x 40 level (every input row staged twice: the launch's PMC traffic is 1.85x its
#endif                                // algorithmic bytes); 12 with ACH_HDW_MAXT 20 = 8-row bands (1.5x), 67 KB of LDS — A/B in profiles/r05_sweep_headdw_bands.txt
#ifndef ACH_HDW_MAXT
#define ACH_HDW_MAXT 10
#endif
constexpr int HDW_THREADS = ACH_HDW_THREADS, HDW_SP = 5, HDW_C = 64, HDW_MAXPOS = ACH_HDW_HALO_ROWS * 44, HDW_MAXT = ACH_HDW_MAXT;
#ifndef ACH_HDW_F32
#define ACH_HDW_F32 0             // 1: the halo tile is staged as fp32 (90 KB: one workgroup per CU, 78 us); 0: as bf16 (45 KB: two per CU, taps unpacked in the loop, 48 us)
#endif
constexpr bool HDW_F32 = ACH_HDW_F32 != 0;
constexpr int HDW_WGS = ACH_HDW_WGS > 0 ? ACH_HDW_WGS : (HDW_F32 ? 1 : 2);
struct HeadDwJob {
    const void* X; void* Y; long ldx, ldy;
    const float* Wdw;                 // [25][128] fp32 (both towers side by side)
    const uint4* Wp;                  // [2 towers][2 k-steps][4 tiles][64 lanes] bf16 A fragments
    const float* bias;                // [128]
    int H, W, rb, bands, wg0;         // wg0: first workgroup of the job
};
struct HeadDwParams {
    HeadDwJob job[3];
    int njobs, B;
    int dbg;                          // timing experiments only (option head_fuse_dbg; wrong results): bit 0 no staging, 1 no depthwise, 2 no GEMM / stores
    int shared_in;                    // 1: both towers read the SAME 64 input channels (first layer, fed by the stem); 0: tower br reads channels br*64 ..
};

template <class T>          // (bf16_t / f16_t: the 16-bit storage types)
__global__ __launch_bounds__(HDW_THREADS, HDW_WGS) void headdw_kernel(const HeadDwParams p) { f16_sat_mode<T>();
    constexpr int C = HDW_C, SP = HDW_SP, KS = 5;
    __shared__ __attribute__((aligned(16))) float xin[HDW_F32 ? HDW_MAXPOS * C : HDW_MAXPOS * C / 2];       // 90 KB (fp32) / 45 KB (bf16)
#if ACH_HDW_ALIAS
    uint4* const xs = reinterpret_cast<uint4*>(xin);            // the B fragments go OVER the halo tile
#else
    __shared__ uint4 xs[HDW_MAXT * 2 * 64];                     // 20 KB: B fragments of the band's tiles
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, g = lane >> 4;
    int ji = 0;
    ACH_UNROLL
    for (int k = 1; k < 3; ++k) if (k < p.njobs && int(blockIdx.x) >= p.job[k].wg0) ji = k;
    const HeadDwJob& J = p.job[ji];
    const int idx = int(blockIdx.x) - J.wg0;
    const int br = idx & 1, band = (idx >> 1) % J.bands, b = (idx >> 1) / J.bands;
    const int H = J.H, W = J.W;
    const int y0 = band * J.rb, rows = (y0 + J.rb <= H) ? J.rb : H - y0;
    const int npx = rows * W, nt = (npx + 15) / 16;
    const int WCr = W + KS - 1, HRr = rows + KS - 1;
    const T* X = static_cast<const T*>(J.X) + long(b) * H * W * J.ldx + (p.shared_in ? 0 : br * C);
    // ---- 0. halo tile -> LDS (fp32)
    if (!(p.dbg & 1)) {
        constexpr int C8 = C / 8, UN = 2;
        const int total = HRr * WCr * C8;
        for (int it0 = tid; it0 < total; it0 += UN * HDW_THREADS) {
            uint4 raw[UN];
            ACH_UNROLL
            for (int u = 0; u < UN; ++u) {
                const int it = it0 + u * HDW_THREADS;
                raw[u] = make_uint4(0u, 0u, 0u, 0u);
                if (it < total) {
                    const int c8 = it % C8, pos = it / C8, wc = pos % WCr, hr = pos / WCr;
                    const int iy = y0 - KS / 2 + hr, ix = wc - KS / 2;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) raw[u] = *reinterpret_cast<const uint4*>(X + (long(iy) * W + ix) * J.ldx + c8 * 8);
                }
            }
            ACH_UNROLL
            for (int u = 0; u < UN; ++u) {
                const int it = it0 + u * HDW_THREADS;
                if (it >= total) continue;
                const int c8 = it % C8, pos = it / C8;
                if (HDW_F32) {
                    float v[8];
                    frag_unpack<T>(raw[u], v);
                    float* d = xin + pos * C + c8 * 8;
                    *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    reinterpret_cast<uint4*>(xin)[pos * C8 + c8] = raw[u];
                }
            }
        }
    }
#if ACH_HDW_ALIAS
    // ---- 1. depthwise 5 x 5 from LDS: thread = strip of SP pixels x 4 channels.  The bf16 sums wait in registers until every thread is done
    // with the halo tile, then go OVER it as the B fragments of the pointwise GEMM (one LDS buffer: 45 KB, three workgroups per CU instead of two)
    __syncthreads();
    constexpr int C4 = C / 4;
    const int nstrip = (W + SP - 1) / SP, total = rows * nstrip * C4;
    uint2 res[HDW_MAXIT][SP];
    if (!(p.dbg & 2)) {
        const float* wdw = J.Wdw + br * C;
        ACH_UNROLL
        for (int k = 0; k < HDW_MAXIT; ++k) {
            const int it = tid + k * HDW_THREADS;
            if (it >= total) break;
            const int cg = it % C4, rest = it / C4, q = rest % nstrip, r = rest / nstrip;
            const int x0 = q * SP;
            f32x2 acc[SP][2];
            ACH_UNROLL
            for (int i = 0; i < SP; ++i) { acc[i][0] = f32x2{0.f, 0.f}; acc[i][1] = f32x2{0.f, 0.f}; }
            ACH_NO_UNROLL
            for (int ty = 0; ty < KS; ++ty) {
                f32x2 v[SP + KS - 1][2];
                ACH_UNROLL
                for (int j = 0; j < SP + KS - 1; ++j) {
                    const int col = x0 + j < WCr ? x0 + j : WCr - 1;
                    if (HDW_F32) {
                        const float4 t = *reinterpret_cast<const float4*>(xin + ((r + ty) * WCr + col) * C + cg * 4);
                        v[j][0] = f32x2{t.x, t.y}; v[j][1] = f32x2{t.z, t.w};
                    } else {
                        const uint2 t = reinterpret_cast<const uint2*>(xin)[((r + ty) * WCr + col) * (C / 4) + cg];
                        v[j][0] = f32x2{H16<T>::lo(t.x), H16<T>::hi(t.x)};
                        v[j][1] = f32x2{H16<T>::lo(t.y), H16<T>::hi(t.y)};
                    }
                }
                ACH_UNROLL
                for (int tx = 0; tx < KS; ++tx) {
                    const float4 w = *reinterpret_cast<const float4*>(wdw + long(ty * KS + tx) * (2 * C) + cg * 4);
                    const f32x2 w0 = {w.x, w.y}, w1 = {w.z, w.w};
                    ACH_UNROLL
                    for (int i = 0; i < SP; ++i) { acc[i][0] += w0 * v[i + tx][0]; acc[i][1] += w1 * v[i + tx][1]; }
                }
            }
            ACH_UNROLL
            for (int i = 0; i < SP; ++i) {
                res[k][i].x = H16<T>::pack(acc[i][0][0], acc[i][0][1]);
                res[k][i].y = H16<T>::pack(acc[i][1][0], acc[i][1][1]);
#if !defined(ACH_HOSTEMU)
                asm volatile("" : "+v"(res[k][i].x), "+v"(res[k][i].y));          // the sums exist BEFORE the barrier (k_upchain.h: the compiler sinks them otherwise)
#endif
            }
        }
    }
    __syncthreads();
    // the pointwise weights of this tower: 2 k-steps x 4 output tiles
    uint4 wp[2][4];
    ACH_UNROLL
    for (int s = 0; s < 2; ++s) { ACH_UNROLL for (int t = 0; t < 4; ++t) wp[s][t] = J.Wp[((br * 2 + s) * 4 + t) * 64 + lane]; }
    // fragments of the (ragged) last tile must not hold garbage
    for (int i = tid; i < nt * 2 * 64; i += HDW_THREADS) xs[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    if (!(p.dbg & 2)) {
        ACH_UNROLL
        for (int k = 0; k < HDW_MAXIT; ++k) {
            const int it = tid + k * HDW_THREADS;
            if (it >= total) break;
            const int cg = it % C4, rest = it / C4, q = rest % nstrip, r = rest / nstrip;
            const int x0 = q * SP;
            const int c0 = cg * 4, s = c0 >> 5, gg = (c0 & 31) >> 3, e = c0 & 7;          // k-step, lane group and element of these 4 channels
            ACH_UNROLL
            for (int i = 0; i < SP; ++i) {
                if (x0 + i >= W) continue;
                const int pix = r * W + x0 + i, t = pix >> 4, pp = pix & 15;
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(xs + (t * 2 + s) * 64 + gg * 16 + pp) + e * 2) = res[k][i];
            }
        }
    }
    __syncthreads();
#else
    // the pointwise weights of this tower: 2 k-steps x 4 output tiles
    uint4 wp[2][4];
    ACH_UNROLL
    for (int s = 0; s < 2; ++s) { ACH_UNROLL for (int t = 0; t < 4; ++t) wp[s][t] = J.Wp[((br * 2 + s) * 4 + t) * 64 + lane]; }
    // fragments of the (ragged) last tile must not hold garbage
    for (int i = tid; i < nt * 2 * 64; i += HDW_THREADS) xs[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // ---- 1. depthwise 5 x 5 from LDS: thread = strip of SP pixels x 4 channels -> bf16 B fragments
    if (!(p.dbg & 2)) {
        constexpr int C4 = C / 4;
        const int nstrip = (W + SP - 1) / SP, total = rows * nstrip * C4;
        const float* wdw = J.Wdw + br * C;
        for (int it = tid; it < total; it += HDW_THREADS) {
            const int cg = it % C4, rest = it / C4, q = rest % nstrip, r = rest / nstrip;
            const int x0 = q * SP;
            f32x2 acc[SP][2];
            ACH_UNROLL
            for (int i = 0; i < SP; ++i) { acc[i][0] = f32x2{0.f, 0.f}; acc[i][1] = f32x2{0.f, 0.f}; }
            ACH_NO_UNROLL
            for (int ty = 0; ty < KS; ++ty) {
                f32x2 v[SP + KS - 1][2];
                ACH_UNROLL
                for (int j = 0; j < SP + KS - 1; ++j) {
                    const int col = x0 + j < WCr ? x0 + j : WCr - 1;
                    if (HDW_F32) {
                        const float4 t = *reinterpret_cast<const float4*>(xin + ((r + ty) * WCr + col) * C + cg * 4);
                        v[j][0] = f32x2{t.x, t.y}; v[j][1] = f32x2{t.z, t.w};
                    } else {
                        const uint2 t = reinterpret_cast<const uint2*>(xin)[((r + ty) * WCr + col) * (C / 4) + cg];
                        v[j][0] = f32x2{H16<T>::lo(t.x), H16<T>::hi(t.x)};
                        v[j][1] = f32x2{H16<T>::lo(t.y), H16<T>::hi(t.y)};
                    }
                }
                ACH_UNROLL
                for (int tx = 0; tx < KS; ++tx) {
                    const float4 w = *reinterpret_cast<const float4*>(wdw + long(ty * KS + tx) * (2 * C) + cg * 4);
                    const f32x2 w0 = {w.x, w.y}, w1 = {w.z, w.w};
                    ACH_UNROLL
                    for (int i = 0; i < SP; ++i) { acc[i][0] += w0 * v[i + tx][0]; acc[i][1] += w1 * v[i + tx][1]; }
                }
            }
            const int c0 = cg * 4, s = c0 >> 5, gg = (c0 & 31) >> 3, e = c0 & 7;          // k-step, lane group and element of these 4 channels
            ACH_UNROLL
            for (int i = 0; i < SP; ++i) {
                if (x0 + i >= W) continue;
                const int pix = r * W + x0 + i, t = pix >> 4, pp = pix & 15;
                uint2 o;
                o.x = H16<T>::pack(acc[i][0][0], acc[i][0][1]);
                o.y = H16<T>::pack(acc[i][1][0], acc[i][1][1]);
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(xs + (t * 2 + s) * 64 + gg * 16 + pp) + e * 2) = o;
            }
        }
    }
    __syncthreads();
#endif
    // ---- 2. pointwise 64 x 64 on MFMA, + bias, ReLU; lane (px, g) of tile pair q holds output channels q*32 + g*8 .. +7
    T* Y = static_cast<T*>(J.Y) + long(b) * H * W * J.ldy + br * C;
    const float* bias = J.bias + br * C;
    for (int t = wave; t < ((p.dbg & 4) ? 0 : nt); t += HDW_THREADS / 64) {
        f32x4 acc[4];
        ACH_UNROLL
        for (int k = 0; k < 4; ++k) { acc[k][0] = 0.f; acc[k][1] = 0.f; acc[k][2] = 0.f; acc[k][3] = 0.f; }
        ACH_UNROLL
        for (int s = 0; s < 2; ++s) {
            const uint4 xf = xs[(t * 2 + s) * 64 + lane];
            ACH_UNROLL
            for (int k = 0; k < 4; ++k) mfma16<T>(wp[s][k], xf, acc[k]);
        }
        const int pix = t * 16 + px;
        if (pix >= npx) continue;
        const long m = long(y0) * W + pix;
        ACH_UNROLL
        for (int q = 0; q < 2; ++q) {
            const int nb = q * 32 + g * 8;
            float o[8];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { o[r] = acc[2 * q][r] + bias[nb + r]; o[4 + r] = acc[2 * q + 1][r] + bias[nb + 4 + r]; }
            ACH_UNROLL
            for (int i = 0; i < 8; ++i) o[i] = o[i] > 0.f ? o[i] : 0.f;
            Store<T>::st8(Y + m * J.ldy + nb, o);
        }
    }
}

inline int headdw_band_rows(int H, int W) {            // the band's halo must fit HDW_MAXPOS positions and its pixels HDW_MAXT tiles
    for (int rb = H; rb >= 1; --rb)
        if ((rb + 4) * (W + 4) <= HDW_MAXPOS && (rb * W + 15) / 16 <= HDW_MAXT && (!ACH_HDW_ALIAS || rb * ((W + HDW_SP - 1) / HDW_SP) * (HDW_C / 4) <= 1024)) return rb;
    return 0;
}

}  // namespace ach
