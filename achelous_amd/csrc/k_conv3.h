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
#include "k_radar.h"

namespace ach {

struct Conv3Params {
    const void* X; long ldx;        // NHWC input, ldx = cv * VEC, with a one-pixel ZERO BORDER around every sample: X is pixel (0,0) of
    long xpr, xpi;                  //   sample 0, xpr / xpi the row / image pitches in elements — the conv's zero padding is in memory
    void* Y; long ldy;              // NHWC output [B,H,W,ldy], ldy >= 32
    const void* W; const float* bias;   // packed NT = 2, one chunk, KS k-steps ; bias[32]
    int B, H, Wd, cv, act;
    int dense;                      // 1: X is a plain dense NHWC map (no border): taps outside the map are masked instead (stride 1 only)
    const void* R; long ldr;        // optional residual added after the activation, same pixel layout as Y (nullptr: none)
    int rows4;                      // 1: a workgroup owns four consecutive output rows, one per wave (H % 4 == 0): the weight fragments every lane loads
                                    //    at the start serve a whole row per wave instead of a quarter of one
};

// NT = 1: up to 16 outputs (one 8-byte store per lane); NT = 2: up to 32.  STRIDE 1 or 2 (H, Wd are the OUTPUT map; the input is
// the bordered map of (H - 1) * STRIDE + 1 .. rows, pad 1).
// HALF: the input pixel is HALF a 16-byte vector (4 bf16 channels = 8 B: the 3-channel maps of the first RCBlock, ldx = 4, cv = 1):
// a k-slot is still one tap x 8 k-elements, its upper four elements are zero (zero weights there), and a tap is one 8-byte load.
template <class T, int KS, int NT, int STRIDE, bool HALF = false>
__global__ __launch_bounds__(256) void conv3x3_rows_kernel(const Conv3Params p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);          // consecutive rows of a sample share an L2 (halo rows)
    const unsigned grow = p.rows4 ? wg * 4u + unsigned(wave_uniform(wave)) : wg;
    const int b = int(grow / unsigned(p.H)), oy = int(grow - unsigned(b) * unsigned(p.H));
    const int ldx = int(p.ldx);

    int off[KS], dxs[KS];
    bool rowok[KS];
    ACH_UNROLL
    for (int s = 0; s < KS; ++s) {
        const int q = 4 * s + g;
        const int tap = q / p.cv, c = q - tap * p.cv;
        const int ty = tap / 3, tx = tap - 3 * ty;
        // k-slots past the ninth tap: zero weights, but keep the address in bounds
        off[s] = tap < 9 ? (ty - 1) * int(p.xpr) + (tx - 1) * ldx + c * VEC : 0;
        dxs[s] = tap < 9 ? tx - 1 : 0;
        const int iy = oy + ty - 1;
        rowok[s] = tap >= 9 || (iy >= 0 && iy < p.H);                 // dense input only
    }
    const uint4* Wf = static_cast<const uint4*>(p.W) + lane;
    uint4 wf[KS][NT];
    ACH_UNROLL
    for (int s = 0; s < KS; ++s)
        ACH_UNROLL
        for (int t = 0; t < NT; ++t) wf[s][t] = Wf[(s * NT + t) * 64];
    float bv[4 * NT];
    ACH_UNROLL
    for (int i = 0; i < 4 * NT; ++i) bv[i] = p.bias[g * 4 * NT + i];

    const T* xrow = static_cast<const T*>(p.X) + long(b) * p.xpi + long(oy) * STRIDE * p.xpr;
    T* yrow = static_cast<T*>(p.Y) + (long(b) * p.H + oy) * p.Wd * p.ldy + g * 4 * NT;
    const bool chan_ok = g * 4 * NT < int(p.ldy);
    for (int tile = p.rows4 ? 0 : wave; tile * 16 < p.Wd; tile += p.rows4 ? 1 : 4) {
        const int xr = tile * 16 + px;
        const bool valid = xr < p.Wd;
        const int x = valid ? xr : p.Wd - 1;
        const T* xp = xrow + long(x) * STRIDE * ldx;
        uint4 xf[KS];
        ACH_UNROLL
        for (int s = 0; s < KS; ++s) {
            if (p.dense) {                                             // masked taps read the centre pixel (in bounds) and are zeroed
                const int ix = x + dxs[s];
                const bool ok = rowok[s] && ix >= 0 && ix < p.Wd;
                const uint4 v = *reinterpret_cast<const uint4*>(ok ? xp + off[s] : xp);
                xf[s] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
            } else if (HALF) {
                const uint2 v = *reinterpret_cast<const uint2*>(xp + off[s]);
                xf[s] = make_uint4(v.x, v.y, 0u, 0u);
            } else {
                xf[s] = *reinterpret_cast<const uint4*>(xp + off[s]);
            }
        }
        f32x4 acc[NT];
        ACH_UNROLL
        for (int t = 0; t < NT; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; acc[t][3] = 0.f; }
        ACH_UNROLL
        for (int s = 0; s < KS; ++s)
            ACH_UNROLL
            for (int t = 0; t < NT; ++t) mfma16<T>(wf[s][t], xf[s], acc[t]);
        if (!valid || !chan_ok) continue;
        if (NT == 2) {
            float o[8];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { o[r] = acc[0][r] + bv[r]; o[4 + r] = acc[NT - 1][r] + bv[4 * (NT - 1) + r]; }
            apply_act_n<T, 8>(o, p.act);
            if (p.R) {
                float r8[8];
                Store<T>::ld8(static_cast<const T*>(p.R) + ((long(b) * p.H + oy) * p.Wd + x) * p.ldr + g * 8, r8);
                ACH_UNROLL
                for (int i = 0; i < 8; ++i) o[i] += r8[i];
            }
            Store<T>::st8(yrow + long(x) * p.ldy, o);
        } else {
            float o[4];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) o[r] = acc[0][r] + bv[r];
            apply_act_n<T, 4>(o, p.act);
            Store<T>::st4(yrow + long(x) * p.ldy, o);
        }
    }
}

template <class T>
inline bool launch_conv3(const Conv3Params& p, int ksteps, int NT, int stride, hipStream_t stream, bool half = false) {
    const dim3 grid(unsigned(p.B) * unsigned(p.H) / (p.rows4 ? 4u : 1u)), block(256);
    if (half) {
        if (Store<T>::VEC != 8 || ksteps != 3 || NT != 1 || stride != 2 || p.dense) return false;
        ACH_LAUNCH((conv3x3_rows_kernel<T, 3, 1, 2, true>), grid, block, stream, p);
        return true;
    }
#define ACH_C3_CASE(ks, nt, st) if (ksteps == ks && NT == nt && stride == st) { ACH_LAUNCH((conv3x3_rows_kernel<T, ks, nt, st>), grid, block, stream, p); return true; }
    ACH_C3_CASE(3, 2, 1) ACH_C3_CASE(5, 2, 1) ACH_C3_CASE(9, 2, 1)
    ACH_C3_CASE(3, 1, 2) ACH_C3_CASE(5, 1, 2) ACH_C3_CASE(9, 1, 2)
    ACH_C3_CASE(3, 2, 2) ACH_C3_CASE(5, 2, 2) ACH_C3_CASE(9, 2, 2)
#undef ACH_C3_CASE
    return false;
}

}  // namespace ach

namespace ach {

// ------------------------------------------------------------------------------------------ RCBlock front half, fused
// offset/modulator conv (above) + modulated deformable 3x3 sampling + [regular_conv . weight_conv1 . BN] + ReLU + residual
// (radar_models/dcn.py:49-63, RadarEncoder.py:80-92) for blocks of up to 16 channels, in ONE kernel: the 27-channel
// offset/mask tensor — 64 B per pixel, the dominant HBM traffic of both separate kernels at 320x320 and 160x160 — never leaves
// the chip.  Per 16-pixel tile of an output row:
//   1. conv: 2 x KS MFMAs -> the 27 offset/mask values of each pixel, parked in a 2 KB per-wave LDS tile;
//   2. lane (pixel, g) owns k-slots 4s+g = (tap, channel vector): it reads that tap's (dy, dx, logit) from LDS, forms the
//      bilinear footprint on the zero-bordered map (no validity logic), gathers 4 corners (16 B each), blends, and the packed
//      result IS the B fragment of k-step s;
//   3. KS MFMAs against the folded weights -> output channels, + bias, ReLU, + residual, one 8/16-byte store per lane.
struct RcFrontParams {
    const void* P; long ldp, prow, pimg;      // avg-pooled input with a zero border: pixel (0,0) of sample 0, row / image pitches
    const void* Wom; const float* bom;        // offset + modulator conv: packed NT = 2, KS k-steps; bias[32]
    const void* Wf; const float* bf;          // folded deformable weights [C][9*ldp]: packed NT = 1, KS k-steps; bias[16]
    const void* R; long ldr;                  // block input (residual)
    void* Y; long ldy, ypr, ypi;              // output: pixel (0,0) of sample 0 and row / image pitches (dense or zero-bordered)
    int B, H, Wd, cv, C;
    const unsigned short* occ; int occ_r;     // optional occupancy of P per 16-pixel row segment [B][H][Wd/16], one bit per column (avgpool3x3), and the reach of a pixel, see below
    int compact;                              // 1 (with occ, rows4, <= 4 output channels): the row's ACTIVE PIXELS are compacted into dense 16-pixel tiles (round 3)
    int rows4;                                // 1: a workgroup owns FOUR consecutive rows, one per wave (H % 4 == 0): the weight staging and the tables are paid once per four rows
    const void* Rn; int rn_bf16;              // round 4 (NARROW): the residual read straight from the caller's NCHW map [B, 3, H, Wd] (16-bit; rn_bf16: bf16 values behind fp16 storage) instead of R
    // round 6, BACKGROUND mode (compact mode only): the output map holds relu(bias) — what an unoccupied pixel with a zero input evaluates to — at every pixel that was not active
    // after the previous forward (written once when the plan is built), `prev` [B][H][Wd/16] holds those activity masks.  An inactive pixel is then neither read nor written:
    // the kernel only restores the background where a pixel was active last time and is not now.  Needs occupancy masks that include non-zero raw inputs (avgpool3x3_nchw3_kernel).
    unsigned short* prev; int bg;
};

// Empty segments (first RCBlock only: its input is the raw radar map, > 99 % zeros — radar_feature_map_generate.ipynb).  If P is
// exactly zero within `occ_r` rows / columns of a 16-pixel segment — occ_r = 2 + ceil(max |offset-conv bias|): the 3x3 conv window, plus
// the constant offset the conv then produces, plus the bilinear footprint — then the offsets are the bias, every sampled corner is 0, the
// contraction accumulates +0 and the output is relu(bias) + residual: the segment takes that shortcut, which evaluates the SAME final
// expression (bit-identical to the full path).  Dense maps only pay the mask test.  occ_r <= 15: horizontally a segment sees the last occ_r
// columns of its left neighbour and the first occ_r of its right one (column masks: with whole-segment flags 66 % of the segments of a
// 256-cells-per-frame map stayed active, with column masks 36 %).

// NARROW (bf16 storage, the first RCBlock: 3 channels): the maps are carried as 4-channel = 8-BYTE pixels (ldp = ldr = ldy = 4, cv = 1).
// A corner of the sampling and a tap of the conv are one 8-byte load each; a k-slot is one tap x [4 channels | 4 zeros].
#ifndef ACH_RCF_WIDE_WAVES
#define ACH_RCF_WIDE_WAVES 1
#endif
// fp16 storage (round 4): the four-corner blend of a deformable tap on PACKED halves — the corners are fp16 in memory and the blended value is rounded to fp16 for the
// MFMA anyway; v_pk_mul_f16 + 3 v_pk_fma_f16 per channel pair replace 2 x (2 conversions + 4 fp32 FMAs) and the final pack: ~68 -> ~20 VALU instructions per tap and lane.
// Costs 2-3 fp16 ulps on the sampled value instead of 0.5 (fp32 blend, one rounding); every radar tap of the fixtures stays at 1-3e-3 of the reference (bound 2e-2).
// bf16 storage has no packed FMA on gfx950 and keeps the fp32 blend, as does the CPU emulation.
#ifndef ACH_RCF_PK16
#define ACH_RCF_PK16 1
#endif
#if !defined(ACH_HOSTEMU)
typedef _Float16 rcf_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rcf_blend_h2(uint32_t a, uint32_t b, uint32_t c, uint32_t d, rcf_h2 w00, rcf_h2 w01, rcf_h2 w10, rcf_h2 w11) {
    const rcf_h2 r = __builtin_elementwise_fma(w11, __builtin_bit_cast(rcf_h2, d), __builtin_elementwise_fma(w10, __builtin_bit_cast(rcf_h2, c),
                     __builtin_elementwise_fma(w01, __builtin_bit_cast(rcf_h2, b), w00 * __builtin_bit_cast(rcf_h2, a))));
    return __builtin_bit_cast(uint32_t, r);
}
#endif
template <class T, int KS, bool NARROW>
__global__ __launch_bounds__(256, KS == 3 ? 4 : ACH_RCF_WIDE_WAVES) void rc_front_kernel(const RcFrontParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC;
    __shared__ float oml[4][16][36];                               // per wave: [pixel][27 values], row padded against bank conflicts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);
    const int wv = wave_uniform(wave);
    const unsigned grow = p.rows4 ? wg * 4u + unsigned(wv) : wg;   // global row index (sample * H + row)
    const int b = int(grow / unsigned(p.H)), oy = int(grow - unsigned(b) * unsigned(p.H));
    const int ldp = int(p.ldp), prow = int(p.prow);

    int off[KS], tapi[KS], cofs[KS];
    float ybase[KS], xbase[KS];
    ACH_UNROLL
    for (int s = 0; s < KS; ++s) {
        const int q = 4 * s + g;
        const int tap = q / p.cv, c = q - tap * p.cv;
        const int ty = tap / 3, tx = tap - 3 * ty;
        const bool live = tap < 9;
        tapi[s] = live ? tap : -1;
        cofs[s] = c * VEC;
        off[s] = live ? (ty - 1) * prow + (tx - 1) * ldp + c * VEC : 0;
        ybase[s] = float(oy + ty - 1);
        xbase[s] = float(tx - 1);
    }
    const uint4* Wom = static_cast<const uint4*>(p.Wom) + lane;
    const uint4* Wfp = static_cast<const uint4*>(p.Wf) + lane;
    // WLDS: the 3 KS weight fragments live in LDS (one copy per workgroup) instead of 12 KS registers per lane — for the small
    // instantiations this is what lets four waves per SIMD fit without spills
    constexpr bool WLDS = KS <= 3;
    __shared__ uint4 wsh[WLDS ? KS * 3 * 64 : 1];
    uint4 wom[WLDS ? 1 : KS][2], wfd[WLDS ? 1 : KS];
    if (WLDS) {
        for (int i = threadIdx.x; i < KS * 3 * 64; i += 256) {
            const int f = i >> 6, l = i & 63;                       // fragment f: 0 .. 2KS-1 conv, 2KS .. 3KS-1 folded contraction
            wsh[i] = f < 2 * KS ? static_cast<const uint4*>(p.Wom)[f * 64 + l] : static_cast<const uint4*>(p.Wf)[(f - 2 * KS) * 64 + l];
        }
        __syncthreads();
    } else {
        ACH_UNROLL
        for (int s = 0; s < KS; ++s) { wom[s][0] = Wom[(s * 2) * 64]; wom[s][1] = Wom[(s * 2 + 1) * 64]; wfd[s] = Wfp[s * 64]; }
    }
    float bv[8], bo[4];
    ACH_UNROLL
    for (int i = 0; i < 8; ++i) bv[i] = p.bom[g * 8 + i];
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) bo[i] = p.bf[g * 4 + i];

    const T* Pimg = static_cast<const T*>(p.P) + long(b) * p.pimg;
    const T* xrow = Pimg + long(oy) * p.prow;
    const float prow_b = float(prow) * float(sizeof(T)), ld_b = float(ldp) * float(sizeof(T));
    const long rowpix = (long(b) * p.H + oy) * p.Wd;
    const int ntiles = (p.Wd + 15) / 16;
    // the conv inputs of the NEXT tile are fetched while the current tile is being sampled (one memory round trip hidden)
    uint4 xf[KS];
    const unsigned short* plist_w = nullptr;                        // compact mode (below): x of the tile's pixels comes from the wave's list
    int plist_n = 0;
    auto fetch = [&](int tile) {
        const int xr = tile * 16 + px;
        const int x = plist_w ? int(plist_w[xr < plist_n ? xr : plist_n - 1]) : (xr < p.Wd ? xr : p.Wd - 1);
        const T* xp = xrow + long(x) * ldp;
        ACH_UNROLL
        for (int s = 0; s < KS; ++s) {
            if constexpr (NARROW) { const uint2 v = *reinterpret_cast<const uint2*>(xp + off[s]); xf[s] = make_uint4(v.x, v.y, 0u, 0u); }
            else xf[s] = *reinterpret_cast<const uint4*>(xp + off[s]);
        }
    };
    unsigned active = 0xffffffffu;                                  // bit t: segment t of this row needs the full path
    if (p.occ) {                                                    // ntiles <= 32 (engine)
        unsigned cols = 0;                                          // lane t: columns of segment t occupied in rows oy - occ_r .. oy + occ_r
        if (lane < ntiles) {                                        // fixed trip count (rows clamped into the map: duplicates do not change an OR)
            const unsigned short* f = p.occ + long(b) * p.H * ntiles + lane;
#pragma unroll 8
            for (int dy = -p.occ_r; dy <= p.occ_r; ++dy) {
                const int y = oy + dy < 0 ? 0 : (oy + dy > p.H - 1 ? p.H - 1 : oy + dy);
                cols |= f[y * ntiles];
            }
        }
        const unsigned own = unsigned(wave_ballot64(cols != 0));
        const unsigned to_right = unsigned(wave_ballot64((cols >> (16 - p.occ_r)) != 0));           // its last occ_r columns reach segment t + 1
        const unsigned to_left = unsigned(wave_ballot64((cols & ((1u << p.occ_r) - 1u)) != 0));     // its first occ_r columns reach segment t - 1
        active = own | (to_right << 1) | (to_left >> 1);
    }
    // Round 3: per-PIXEL activity.  Segment flags leave 36 % of the segments of the bench's maps (256 cells per frame) active, but only ~20 % of the
    // pixels have an occupied column within occ_r of them: the row's active pixels are compacted into dense tiles (their x positions in a per-wave LDS
    // list: every lane of the full path already addresses its own pixel), everything else takes the closed form, 64 pixels per wave pass.
    __shared__ unsigned short plist[4][NARROW ? 512 + 16 : 1];
    __shared__ unsigned short amask[4][NARROW ? 32 : 1];
    const bool compact = NARROW && p.compact && p.occ && p.rows4 && p.ldy <= 4 && p.Wd <= 512;
    int ntot = 0;
    if (compact) {
        if constexpr (NARROW) {
        // this lane's segment: columns occupied within occ_r rows (cols), dilated by occ_r columns with the neighbours' masks
        unsigned cols = 0;
        if (lane < ntiles) {
            const unsigned short* f = p.occ + long(b) * p.H * ntiles + lane;
#pragma unroll 8
            for (int dy = -p.occ_r; dy <= p.occ_r; ++dy) {
                const int y = oy + dy < 0 ? 0 : (oy + dy > p.H - 1 ? p.H - 1 : oy + dy);
                cols |= f[y * ntiles];
            }
        }
        const unsigned lc = unsigned(__shfl(int(cols), lane > 0 ? lane - 1 : 0)), rc = unsigned(__shfl(int(cols), lane < 63 ? lane + 1 : 63));
        const unsigned long long wide = (unsigned long long)(lane > 0 ? lc : 0u) | ((unsigned long long)cols << 16) | ((unsigned long long)(lane + 1 < ntiles ? rc : 0u) << 32);
        unsigned long long dil = wide;
        for (int d = 1; d <= p.occ_r; ++d) dil |= (wide << d) | (wide >> d);
        unsigned act16 = lane < ntiles ? unsigned(dil >> 16) & 0xffffu : 0u;
        const int lim = p.Wd - lane * 16;                                  // columns of this segment inside the map
        if (lim < 16) act16 &= lim > 0 ? ((1u << lim) - 1u) : 0u;
        const int cnt = __popcll(static_cast<unsigned long long>(act16));
        int base = 0;
        for (int l = 0; l < ntiles; ++l) { const int c = __shfl(cnt, l); if (l < lane) base += c; ntot += c; }
        if (lane < 32) amask[wave][lane] = static_cast<unsigned short>(act16);
        if (p.bg && lane < ntiles) {
            unsigned short* pw = p.prev + (long(b) * p.H + oy) * ntiles + lane;
            unsigned stale = unsigned(*pw) & ~act16;                         // active after the previous forward, background now
            *pw = static_cast<unsigned short>(act16);
            for (; stale; stale &= stale - 1u) {
                const int xx = lane * 16 + (__ffsll(static_cast<long long>(stale)) - 1);
                float ov[4];
                ACH_UNROLL
                for (int i = 0; i < 4; ++i) { const float r = 0.f + p.bf[i]; ov[i] = (r > 0.f ? r : 0.f) + 0.f; }
                Store<T>::st4(static_cast<T*>(p.Y) + long(b) * p.ypi + long(oy) * p.ypr + long(xx) * p.ldy, ov);
            }
        }
        for (unsigned m = act16; m; m &= m - 1u) plist[wave][base++] = static_cast<unsigned short>(lane * 16 + (__ffsll(static_cast<long long>(m)) - 1));
        wave_sync();
        }
    }
    const int ch = g * 4;
    // the block input of pixel x of this row (residual): from the NHWC copy, or (NARROW, round 4) from the three NCHW planes of the caller's map, each value passed
    // through the storage type exactly as the copy kernel did
    auto residual4 = [&](int x, int c0, float (&rr)[4]) {
        if constexpr (NARROW && sizeof(T) == 2) {
            if (p.Rn) {
                const uint16_t* q = static_cast<const uint16_t*>(p.Rn) + (long(b) * 3 * p.H + oy) * p.Wd + x;
                const long cs = long(p.H) * p.Wd;
                ACH_UNROLL
                for (int i = 0; i < 3; ++i) {
                    const uint32_t bits = q[i * cs];
                    rr[i] = H16<T>::lo(p.rn_bf16 ? h16_recast<bf16_t, T>(bits) : bits);
                }
                rr[3] = 0.f;
                return;
            }
        }
        Store<T>::ld4(static_cast<const T*>(p.R) + (rowpix + x) * p.ldr + c0, rr);
    };
    auto finish = [&](int x, bool valid, const f32x4& acc) {        // bias, ReLU, residual, one store
        if (valid && ch < int(p.ldy)) {
            float rr[4], ov[4];
            residual4(x, ch, rr);
            ACH_UNROLL
            for (int i = 0; i < 4; ++i) { const float r = acc[i] + bo[i]; ov[i] = (r > 0.f ? r : 0.f) + rr[i]; }
            Store<T>::st4(static_cast<T*>(p.Y) + long(b) * p.ypi + long(oy) * p.ypr + long(x) * p.ldy + ch, ov);
        }
    };
    if (compact && !p.bg) {
        if constexpr (NARROW) {
        // empty pixels: relu(bias) + residual, one pixel per lane and pass (all four output channels are lane group 0's: ldy <= 4)
        float b0[4];
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) b0[i] = p.bf[i];
        for (int xx = lane; xx < p.Wd; xx += 64) {
            if ((amask[wave][xx >> 4] >> (xx & 15)) & 1) continue;
            float rr[4], ov[4];
            residual4(xx, 0, rr);
            ACH_UNROLL
            for (int i = 0; i < 4; ++i) { const float r = 0.f + b0[i]; ov[i] = (r > 0.f ? r : 0.f) + rr[i]; }
            Store<T>::st4(static_cast<T*>(p.Y) + long(b) * p.ypi + long(oy) * p.ypr + long(xx) * p.ldy, ov);
        }
        }
    }
    // Segments are dealt to the four waves by RANK among the active (and among the empty) ones, not by position: occupied cells come
    // in clusters, and position-strided waves would leave one wave with a row's whole cluster.
    const unsigned tmask = ntiles >= 32 ? 0xffffffffu : ((1u << ntiles) - 1u);
    unsigned rem = active & tmask;
    if (compact) {
        if constexpr (NARROW) {
        const int ctiles = (ntot + 15) / 16;                       // <= 32 (Wd <= 512)
        rem = ctiles >= 32 ? 0xffffffffu : ((1u << ctiles) - 1u);
        plist_w = &plist[wave][0]; plist_n = ntot;
        }
    }
    int rank = 0;
    auto take = [&]() -> int {                                      // this wave's next segment of `rem` (consumed), or -1
        while (rem) {
            const int t = __ffsll(static_cast<long long>(rem)) - 1;
            rem &= rem - 1u;
            if (p.rows4 || (rank++ & 3) == wv) return t;
        }
        return -1;
    };
    int tile = take();
    if (tile >= 0) fetch(tile);
    if (p.occ && !compact) {                                        // empty neighbourhoods: the full path would accumulate +0
        unsigned rem_e = ~active & tmask;
        int rank_e = 0;
        while (rem_e) {
            const int t = __ffsll(static_cast<long long>(rem_e)) - 1;
            rem_e &= rem_e - 1u;
            if (!p.rows4 && (rank_e++ & 3) != wv) continue;
            f32x4 zero;
            zero[0] = zero[1] = zero[2] = zero[3] = 0.f;
            const int xr = t * 16 + px;
            finish(xr < p.Wd ? xr : p.Wd - 1, xr < p.Wd, zero);
        }
    }
    while (tile >= 0) {
        const int next = take();
        const int xr = tile * 16 + px;
        const bool valid = plist_w ? xr < plist_n : xr < p.Wd;
        const int x = plist_w ? int(plist_w[valid ? xr : plist_n - 1]) : (valid ? xr : p.Wd - 1);
        {   // 1. offsets + modulator logits of the tile
            f32x4 a0, a1;
            a0[0] = a0[1] = a0[2] = a0[3] = 0.f;
            a1[0] = a1[1] = a1[2] = a1[3] = 0.f;
            ACH_UNROLL
            for (int s = 0; s < KS; ++s) {
                const uint4 w0 = WLDS ? wsh[(s * 2) * 64 + lane] : wom[WLDS ? 0 : s][0], w1 = WLDS ? wsh[(s * 2 + 1) * 64 + lane] : wom[WLDS ? 0 : s][1];
                mfma16<T>(w0, xf[s], a0);
                mfma16<T>(w1, xf[s], a1);
            }
            float* o = &oml[wave][px][g * 8];
            *reinterpret_cast<float4*>(o) = make_float4(a0[0] + bv[0], a0[1] + bv[1], a0[2] + bv[2], a0[3] + bv[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(a1[0] + bv[4], a1[1] + bv[5], a1[2] + bv[6], a1[3] + bv[7]);
        }
        if (next >= 0) fetch(next);
        wave_sync();
        f32x4 acc;
        acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
        ACH_UNROLL
        for (int s = 0; s < KS; ++s) {   // 2. + 3.
            const int tap = tapi[s] < 0 ? 0 : tapi[s];
            const float* om = &oml[wave][px][0];
            const float dy = om[2 * tap], dx = om[2 * tap + 1], ml = om[18 + tap];
            // wave-uniform base (top-left BORDER pixel of the sample) + 32-bit byte offsets >= 0: scalar-base addressing, no 64-bit math
            const char* Pb = reinterpret_cast<const char*>(Pimg - (prow + ldp));
            constexpr unsigned esz = unsigned(sizeof(T));
            const BilinearTapB t = make_tap_bytes(ybase[s] + dy, float(x) + xbase[s] + dx, sigmoid_mod(ml), p.H, p.Wd, prow_b, ld_b, float(cofs[s]) * float(esz));
            const unsigned q0 = t.q0, q1 = q0 + unsigned(prow) * esz;
            const unsigned q0b = q0 + unsigned(ldp) * esz, q1b = q1 + unsigned(ldp) * esz;
            const float live = tapi[s] < 0 ? 0.f : 1.f;
            const float w00 = t.w00 * live, w01 = t.w01 * live, w10 = t.w10 * live, w11 = t.w11 * live;
#if !defined(ACH_HOSTEMU) && ACH_RCF_PK16
            if constexpr (std::is_same<T, f16_t>::value) {
                const rcf_h2 h00 = {_Float16(w00), _Float16(w00)}, h01 = {_Float16(w01), _Float16(w01)}, h10 = {_Float16(w10), _Float16(w10)}, h11 = {_Float16(w11), _Float16(w11)};
                uint4 fr;
                if constexpr (NARROW) {
                    const uint2 A = *reinterpret_cast<const uint2*>(Pb + q0), Bq = *reinterpret_cast<const uint2*>(Pb + q0b);
                    const uint2 Cq = *reinterpret_cast<const uint2*>(Pb + q1), D = *reinterpret_cast<const uint2*>(Pb + q1b);
                    fr = make_uint4(rcf_blend_h2(A.x, Bq.x, Cq.x, D.x, h00, h01, h10, h11), rcf_blend_h2(A.y, Bq.y, Cq.y, D.y, h00, h01, h10, h11), 0u, 0u);
                } else {
                    const uint4 A = *reinterpret_cast<const uint4*>(Pb + q0), Bq = *reinterpret_cast<const uint4*>(Pb + q0b);
                    const uint4 Cq = *reinterpret_cast<const uint4*>(Pb + q1), D = *reinterpret_cast<const uint4*>(Pb + q1b);
                    fr = make_uint4(rcf_blend_h2(A.x, Bq.x, Cq.x, D.x, h00, h01, h10, h11), rcf_blend_h2(A.y, Bq.y, Cq.y, D.y, h00, h01, h10, h11),
                                    rcf_blend_h2(A.z, Bq.z, Cq.z, D.z, h00, h01, h10, h11), rcf_blend_h2(A.w, Bq.w, Cq.w, D.w, h00, h01, h10, h11));
                }
                mfma16<T>(WLDS ? wsh[(2 * KS + s) * 64 + lane] : wfd[WLDS ? 0 : s], fr, acc);
                continue;
            }
#endif
            float v[8];
            if constexpr (NARROW && VEC == 8) {
                float a[4], bq[4], cc[4], d[4];
                Store<T>::ld4(reinterpret_cast<const T*>(Pb + q0), a); Store<T>::ld4(reinterpret_cast<const T*>(Pb + q0b), bq);
                Store<T>::ld4(reinterpret_cast<const T*>(Pb + q1), cc); Store<T>::ld4(reinterpret_cast<const T*>(Pb + q1b), d);
                ACH_UNROLL
                for (int i = 0; i < 4; ++i) { v[i] = w00 * a[i] + w01 * bq[i] + w10 * cc[i] + w11 * d[i]; v[4 + i] = 0.f; }
            } else {
                float a[8], bq[8], cc[8], d[8];
                frag_unpack<T>(*reinterpret_cast<const uint4*>(Pb + q0), a);
                frag_unpack<T>(*reinterpret_cast<const uint4*>(Pb + q0b), bq);
                frag_unpack<T>(*reinterpret_cast<const uint4*>(Pb + q1), cc);
                frag_unpack<T>(*reinterpret_cast<const uint4*>(Pb + q1b), d);
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) v[i] = w00 * a[i] + w01 * bq[i] + w10 * cc[i] + w11 * d[i];
            }
            mfma16<T>(WLDS ? wsh[(2 * KS + s) * 64 + lane] : wfd[WLDS ? 0 : s], frag_pack<T>(v), acc);
        }
        wave_sync();                                                // the LDS tile is rewritten by the next iteration
        finish(x, valid, acc);
        tile = next;
    }
}

template <class T>
inline bool launch_rc_front(const RcFrontParams& p, int ksteps, hipStream_t stream) {
    const dim3 grid(unsigned(p.B) * unsigned(p.H) / (p.rows4 ? 4u : 1u)), block(256);
    const bool narrow = p.ldp == 4 && Store<T>::VEC == 8;          // 8-byte pixels (engine: radar block 0 in bf16)
    if (ksteps == 3) {
        if (narrow) ACH_LAUNCH((rc_front_kernel<T, 3, true>), grid, block, stream, p); else ACH_LAUNCH((rc_front_kernel<T, 3, false>), grid, block, stream, p);
        return true;
    }
    if (ksteps == 5) { ACH_LAUNCH((rc_front_kernel<T, 5, false>), grid, block, stream, p); return true; }
    if (ksteps == 9) { ACH_LAUNCH((rc_front_kernel<T, 9, false>), grid, block, stream, p); return true; }
    return false;
}

}  // namespace ach
