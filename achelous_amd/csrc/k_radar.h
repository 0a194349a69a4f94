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
template <class T, class IO = T>      // IO: the type of the caller's tensor (a 16-bit engine may take bf16 inputs into fp16 storage)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const ToNhwcParams p) { f16_sat_mode<T>();
    const long total = long(p.B) * p.H * p.Wd;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long HW = long(p.H) * p.Wd;
    const long b = idx / HW, pix = idx - b * HW;
    const IO* x = static_cast<const IO*>(p.X) + b * p.C * HW + pix;
    T* y = static_cast<T*>(p.Y) + idx * p.ld;
    for (int c = 0; c < p.C; ++c) Store<T>::st(y + c, Store<IO>::ld(x + c * HW));
}

// 3-channel NCHW -> 4-channel NHWC pixels (8 B in bf16, 16 B in fp32; channel 3 = 0): the layout of the first RCBlock's maps.
// One thread = 4 consecutive pixels of a row: one vector load per plane, four pixels written as one contiguous 32 / 64 bytes.
template <class T, class IO = T>
__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const ToNhwcParams p) { f16_sat_mode<T>();
    const long HW = long(p.H) * p.Wd, quads = HW / 4;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * quads) return;
    const long b = idx / quads, pix = (idx - b * quads) * 4;
    const IO* x = static_cast<const IO*>(p.X) + b * 3 * HW + pix;
    float c0[4], c1[4], c2[4];
    Store<IO>::ld4(x, c0); Store<IO>::ld4(x + HW, c1); Store<IO>::ld4(x + 2 * HW, c2);
    T* y = static_cast<T*>(p.Y) + (b * HW + pix) * 4;
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) { const float v[4] = {c0[i], c1[i], c2[i], 0.f}; Store<T>::st4(y + 4 * i, v); }
}

// AvgPool2d(3, stride 1, pad 1, count_include_pad=True): always divides by 9
// The output goes to a buffer with a one-pixel ZERO BORDER around every sample (Y points at pixel (0,0) of sample 0, ypr / ypi are
// its row / image pitches in elements): the 3x3 conv and the deformable sampling that read it need no bounds logic at all.
struct PoolParams { const void* X; long ldx; void* Y; long ldy; int B, H, Wd, C; long ypr, ypi;
                    unsigned short* occ; };    // optional [B][H][Wd/16]: bit c set where column c of a 16-pixel row segment of the OUTPUT is non-zero (rc_front's skip test)
// One thread = 4 channels x a strip of AVG_OW consecutive output pixels of one row: the 3 x (AVG_OW + 2) window is fetched once
// (18 loads for 4 outputs instead of 36), summed by columns first and then across three columns — the kernel was VALU-issue bound
// (SQ counters: 239 VALU instructions per wave against a 46 us launch at 320x320, exactly the VALU floor), and per output this
// is 4.5 loads, 20 adds and a quarter of the index arithmetic instead of 9, 36 and all of it.
// (AVG_OW = 1 for maps with fewer than 16 channels: with one or two channel groups per pixel a strip per lane spreads a wave's
// loads over 4x more cache lines and the texture path becomes the limit — measured 47 -> 55 us on the 3-channel 320x320 map.)
template <class T, int AVG_OW>
__global__ __launch_bounds__(256) void avgpool3x3_kernel(const PoolParams p) { f16_sat_mode<T>();
    const int cq = (p.C + 3) >> 2;
    const int strips = (p.Wd + AVG_OW - 1) / AVG_OW;
    const long total = long(p.B) * p.H * strips * cq;
    const long idx_raw = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    const bool live = idx_raw < total;
    if (!live && !p.occ) return;                     // with occupancy flags every lane takes part in the ballot below
    const long idx = live ? idx_raw : total - 1;
    const int c = int(idx % cq) * 4;
    long r = idx / cq;
    const int x0 = int(r % strips) * AVG_OW; r /= strips;
    const int y = int(r % p.H);
    const long b = r / p.H;
    // taps through a range-checked buffer resource (ach_platform.h): outside the map -> out-of-range offset -> zeros
    constexpr unsigned ESZ = sizeof(T);
    const unsigned pitch = unsigned(p.ldx) * ESZ;
    const BufRsrc xb = make_buf(p.X, unsigned(p.B) * unsigned(p.H) * unsigned(p.Wd) * pitch);
    unsigned coff[AVG_OW + 2];
    ACH_UNROLL
    for (int j = 0; j < AVG_OW + 2; ++j) {
        const int ix = x0 - 1 + j;
        coff[j] = (ix >= 0 && ix < p.Wd) ? unsigned(ix) * pitch + unsigned(c) * ESZ : BUF_OOB / 2;     // + row offset: >= 2^30, out of range (radar maps are < 1 GiB, see plan())
    }
    float col[AVG_OW + 2][4];
    ACH_UNROLL
    for (int j = 0; j < AVG_OW + 2; ++j) { col[j][0] = 0.f; col[j][1] = 0.f; col[j][2] = 0.f; col[j][3] = 0.f; }
    ACH_UNROLL
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = y + dy;
        const unsigned rowb = (iy >= 0 && iy < p.H) ? (unsigned(b) * unsigned(p.H) + unsigned(iy)) * unsigned(p.Wd) * pitch : BUF_OOB;
        ACH_UNROLL
        for (int j = 0; j < AVG_OW + 2; ++j) {
            float v[4];
            buf_ld4<T>(xb, rowb + coff[j], v);                  // 2^31 + 2^30 at most: no wrap
            col[j][0] += v[0]; col[j][1] += v[1]; col[j][2] += v[2]; col[j][3] += v[3];
        }
    }
    T* yrow = static_cast<T*>(p.Y) + b * p.ypi + y * p.ypr + c;
    int nz = 0;                                      // bit o: output pixel x0 + o is non-zero in some channel
    ACH_UNROLL
    for (int o = 0; o < AVG_OW; ++o) {
        if (x0 + o >= p.Wd) break;
        float acc[4];
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) { acc[i] = ((col[o][i] + col[o + 1][i]) + col[o + 2][i]) * (1.0f / 9.0f); nz |= !(acc[i] == 0.f) ? 1 << o : 0; }     // NaN counts as occupied
        if (live) Store<T>::st4(yrow + long(x0 + o) * p.ldy, acc);
    }
    if (AVG_OW == 4 && p.occ) {
        // engine guarantees one channel group per pixel and Wd % 16 == 0: four consecutive lanes (aligned: rows are multiples of 4 strips and
        // workgroups of 64 strips) hold the four strips of one 16-pixel segment; two butterfly steps gather their 4-bit masks in the first
        int m = live ? nz : 0;
        m |= __shfl_xor(m, 1) << 4;                  // valid in even lanes: strips s, s+1
        m |= __shfl_xor(m, 2) << 8;                  // valid in lanes % 4 == 0: strips s .. s+3
        if (live && (threadIdx.x & 3) == 0) p.occ[(b * p.H + y) * long(p.Wd >> 4) + (x0 >> 4)] = static_cast<unsigned short>(m & 0xffff);
    }
}

// The first RCBlock's pool straight from the caller's NCHW radar map (round 4): the NHWC copy of the map (52 MB written and read twice at batch 64) and its launch
// are gone.  One thread = a strip of 4 output pixels x the 3 channels; per (plane, row) one aligned 8-byte load (x0 .. x0+3: x0 % 4 == 0, Wd % 16 == 0) and the two
// neighbours as 2-byte loads.  Same sums in the same order as avgpool3x3_kernel<T, 4> on the 4-channel copy (values pass through the storage type first, as the copy did):
// bit-identical pooled map and occupancy masks.
// sparse (round 6; needs occ): a pixel is STORED only if its pooled value is non-zero or the buffer still holds a non-zero value there from the previous forward — the occupancy word the
// kernel is about to overwrite says which pixels those are (arena and masks are zero when the plan is built, and nothing else writes either).  The bench's maps are > 99 % zeros: the
// launch writes ~3 MB instead of 52 MB; the pooled map in memory is bit for bit what the unconditional stores leave.
struct PoolNchwParams { const void* X; void* Y; long ldy; int B, H, Wd; long ypr, ypi; unsigned short* occ; int sparse = 0; };
constexpr int POOLN_ROWS = 4;            // output rows per thread: 6 input rows are fetched and converted for 4 output rows instead of 12 (one row per thread: 50.7 us at batch 64)
template <class T, class IO>
__global__ __launch_bounds__(256) void avgpool3x3_nchw3_kernel(const PoolNchwParams p) { f16_sat_mode<T>();
    static_assert(sizeof(T) == 2 && sizeof(IO) == 2, "16-bit storage");
    const int strips = p.Wd / 4, rblocks = p.H / POOLN_ROWS;               // H % POOLN_ROWS == 0 (engine)
    const long total = long(p.B) * rblocks * strips;
    const long idx_raw = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    const bool live = idx_raw < total;
    if (!live && !p.occ) return;
    long r = live ? idx_raw : total - 1;
    const int x0 = int(r % strips) * 4; r /= strips;
    const int y0 = int(r % rblocks) * POOLN_ROWS;
    const long b = r / rblocks;
    const long cstride = long(p.H) * p.Wd;
    const uint16_t* X = static_cast<const uint16_t*>(p.X) + b * 3 * cstride + x0;
    auto val = [](uint32_t bits) { return H16<T>::lo(h16_recast<IO, T>(bits & 0xffffu)); };
    float rows[3][3][6];                                                   // rolling window: [row slot][channel][column x0-1 .. x0+4]
    const bool has_l = x0 > 0, has_r = x0 + 4 < p.Wd;
    auto fetch = [&](int iy, float (&o)[3][6]) {
        const bool rowin = iy >= 0 && iy < p.H;
        ACH_UNROLL
        for (int c = 0; c < 3; ++c) {
            const uint16_t* row = X + c * cstride + long(rowin ? iy : y0) * p.Wd;
            const uint2 mid = *reinterpret_cast<const uint2*>(row);
            // unconditional loads from clamped addresses, the padding applied afterwards: a load under a lane condition is a branch, and behind a branch the
            // compiler waits for every load by itself (40 x vmcnt(0) per thread, 50 us for 92 MB)
            const uint32_t lraw = row[has_l ? -1 : 0], rraw = row[has_r ? 4 : 3];
            const uint32_t lft = has_l ? lraw : 0u, rgt = has_r ? rraw : 0u;
            const float v[6] = {val(lft), val(mid.x), val(mid.x >> 16), val(mid.y), val(mid.y >> 16), val(rgt)};
            ACH_UNROLL
            for (int j = 0; j < 6; ++j) o[c][j] = rowin ? v[j] : 0.f;
        }
    };
    const bool sparse = p.sparse && p.occ;
    int old4[POOLN_ROWS];                                                  // bit o: pixel x0 + o of row y0 + k held a non-zero value after the previous forward
    ACH_UNROLL
    for (int k = 0; k < POOLN_ROWS; ++k) old4[k] = sparse ? (int(p.occ[(b * p.H + y0 + k) * long(p.Wd >> 4) + (x0 >> 4)]) >> (x0 & 15)) & 15 : 15;
    fetch(y0 - 1, rows[0]);
    fetch(y0, rows[1]);
    int nzr[POOLN_ROWS];
    ACH_UNROLL
    for (int k = 0; k < POOLN_ROWS; ++k) {
        float (&up)[3][6] = rows[k % 3], (&mid)[3][6] = rows[(k + 1) % 3], (&dn)[3][6] = rows[(k + 2) % 3];
        fetch(y0 + k + 1, dn);
        T* yrow = static_cast<T*>(p.Y) + b * p.ypi + long(y0 + k) * p.ypr;
        int nz = 0;
        float col[6][3];
        ACH_UNROLL
        for (int j = 0; j < 6; ++j) { ACH_UNROLL for (int c = 0; c < 3; ++c) col[j][c] = ((0.f + up[c][j]) + mid[c][j]) + dn[c][j]; }
        ACH_UNROLL
        for (int o = 0; o < 4; ++o) {
            float acc[4];
            ACH_UNROLL
            for (int i = 0; i < 3; ++i) { acc[i] = ((col[o][i] + col[o + 1][i]) + col[o + 2][i]) * (1.0f / 9.0f); nz |= !(acc[i] == 0.f) ? 1 << o : 0; }
            // a non-zero RAW value under a pooled zero (sums that cancel exactly) counts as occupied too: rc_front's background mode (k_conv3.h, round 6) relies on
            // "unoccupied => the block's input is zero there"; a superset of the occupied pixels is always exact (the full path is)
            ACH_UNROLL
            for (int i = 0; i < 3; ++i) nz |= !(mid[i][o + 1] == 0.f) ? 1 << o : 0;
            acc[3] = 0.f;
            if (live && (((nz | old4[k]) >> o) & 1)) Store<T>::st4(yrow + long(x0 + o) * p.ldy, acc);
        }
        nzr[k] = nz;
    }
    if (p.occ) {
        ACH_UNROLL
        for (int k = 0; k < POOLN_ROWS; ++k) {
            int m = live ? nzr[k] : 0;
            m |= __shfl_xor(m, 1) << 4;
            m |= __shfl_xor(m, 2) << 8;
            if (live && (threadIdx.x & 3) == 0) p.occ[(b * p.H + y0 + k) * long(p.Wd >> 4) + (x0 >> 4)] = static_cast<unsigned short>(m & 0xffff);
        }
    }
}

// every interior pixel of a bordered map of 8-byte pixels <- one value (plan time: the background of the first RCBlock's output, k_conv3.h)
struct FillPx8Params { void* Y; long ypr, ypi; int B, H, Wd; uint2 v; };
template <class T>
__global__ __launch_bounds__(256) void fill_px8_kernel(const FillPx8Params p) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x, total = long(p.B) * p.H * p.Wd;
    if (i >= total) return;
    const int x = int(i % p.Wd), y = int((i / p.Wd) % p.H);
    const long b = i / (long(p.Wd) * p.H);
    *reinterpret_cast<uint2*>(static_cast<char*>(p.Y) + (b * p.ypi + long(y) * p.ypr) * 2 + long(x) * 8) = p.v;
}

// ---- shared by both deformable kernels: one tap's bilinear footprint (torchvision 0.12.0 deform_conv2d semantics:
// a sample at or beyond -1 / H (W) is 0, corners outside the map contribute 0).  The sampled tensor carries a one-pixel zero
// border, so clamping the coordinate to [-1, H] and the top-left corner to [-1, H-1] reproduces exactly that with no
// validity logic: out-of-range samples land on border pixels and/or get weight 0.
struct BilinearTap {
    int o0, o1;                   // element offsets (relative to the sample's pixel (0,0)) of the top-left / bottom-left corners
    float w00, w01, w10, w11;     // bilinear weights multiplied by the modulation mask
};
__device__ __forceinline__ BilinearTap make_tap(float sy, float sx, float mask, int H, int Wd, int prow, int ld) {
    BilinearTap t;
    const float cy = clampf(sy, -1.f, float(H)), cx = clampf(sx, -1.f, float(Wd));
    const float fy = fminf(floorf(cy), float(H - 1)), fx = fminf(floorf(cx), float(Wd - 1));
    const float ly = cy - fy, lx = cx - fx;
    const float hym = (1.f - ly) * mask, lym = ly * mask, hx = 1.f - lx;
    t.w00 = hym * hx; t.w01 = hym * lx; t.w10 = lym * hx; t.w11 = lym * lx;
    t.o0 = int(fy) * prow + int(fx) * ld;
    t.o1 = t.o0 + prow;
    return t;
}
__device__ __forceinline__ float modulation(float logit) { return 2.0f * fast_rcp(1.0f + fast_exp2(-1.44269504f * logit)); }   // 2 sigmoid
// The same footprint for the fused front kernel: the corner address as a BYTE offset from the sample's top-left BORDER pixel,
// formed in fp32 — (fy + 1) * row pitch + (fx + 1) * pixel pitch is an integer below 2^24 for every map of the path (a bordered
// 320x320 sample of 16-byte pixels is 1.7 MB), so two FMAs and one conversion replace two conversions, a 32-bit multiply and a
// 64-bit multiply-add (both quarter rate) per tap of a VALU-bound kernel.
struct BilinearTapB {
    unsigned q0;                  // byte offset of the top-left corner (>= 0: the border pixel row / column is offset 0)
    float w00, w01, w10, w11;
};
__device__ __forceinline__ BilinearTapB make_tap_bytes(float sy, float sx, float mask, int H, int Wd, float prow_b, float ld_b, float chan_b) {
    BilinearTapB t;
    const float cy = clampf(sy, -1.f, float(H)), cx = clampf(sx, -1.f, float(Wd));
    const float fy = fminf(floorf(cy), float(H - 1)), fx = fminf(floorf(cx), float(Wd - 1));
    const float ly = cy - fy, lx = cx - fx;
    const float hym = (1.f - ly) * mask, lym = ly * mask, hx = 1.f - lx;
    t.w00 = hym * hx; t.w01 = hym * lx; t.w10 = lym * hx; t.w11 = lym * lx;
    t.q0 = unsigned(fmaf(fy + 1.f, prow_b, fmaf(fx + 1.f, ld_b, chan_b)));
    return t;
}
__device__ __forceinline__ float sigmoid_mod(float logit) { return fast_rcp(1.0f + fast_exp2(-1.44269504f * logit)); }        // the factor 2 of the modulation lives in the folded weights

struct DeformParams {
    const void* pooled; long ldp;     // sampled tensor (avg-pooled block input), NHWC with a zero border: pixel (0,0) of sample 0,
    long prow, pimg;                  //   row / image pitches in elements
    const void* om; long ldo;         // [pixels, 27]: 18 offsets (dy,dx per tap) + 9 modulator logits
    const void* res; long ldr;        // block input (residual)
    void* Y; long ldy;                // fused: relu(Wf . col + bf) + res ; sample: the columns [pixels, 9*Cp]
    const float* Wf;                  // fused: [C][9][Cp] folded (weight_conv1 . BN . regular_conv)
    const float* bf;                  // fused: [C]
    int B, H, Wd, Cp;
};

// C <= 8: sampling + contraction + ReLU + residual in one pass, one thread per pixel.  CP = channels fetched per corner (4 or 8).
// The folded weights are wave-uniform: passed as __restrict__ kernel arguments so they are fetched with scalar loads.
template <class T, int C, int CP>
__global__ __launch_bounds__(256) void deform_fused_kernel(const DeformParams p, const float* __restrict__ Wf, const float* __restrict__ bf) { f16_sat_mode<T>();
    const long total = long(p.B) * p.H * p.Wd;
    const long idx = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = int(idx % p.Wd);
    const int y = int((idx / p.Wd) % p.H);
    const long b = idx / (long(p.Wd) * p.H);
    const T* om = static_cast<const T*>(p.om) + idx * p.ldo;
    float o[32];
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) {
        float v[8];
        Store<T>::ld8(om + i * 8, v);                 // row stride 32: offsets 0..17, logits 18..26, padding
        ACH_UNROLL
        for (int j = 0; j < 8; ++j) o[i * 8 + j] = v[j];
    }
    const T* P = static_cast<const T*>(p.pooled) + b * p.pimg;
    const int prow = int(p.prow), ld = int(p.ldp);
    f32x2 acc[C];
    ACH_UNROLL
    for (int i = 0; i < C; ++i) { acc[i][0] = bf[i]; acc[i][1] = 0.f; }
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) {
        const BilinearTap t = make_tap(float(y - 1 + k / 3) + o[2 * k], float(x - 1 + k % 3) + o[2 * k + 1], modulation(o[18 + k]),
                                       p.H, p.Wd, prow, ld);
        float a[CP], bq[CP], cc[CP], d[CP];
        if constexpr (CP == 8) {
            Store<T>::ld8(P + t.o0, a);  Store<T>::ld8(P + t.o0 + ld, bq);
            Store<T>::ld8(P + t.o1, cc); Store<T>::ld8(P + t.o1 + ld, d);
        } else {
            Store<T>::ld4(P + t.o0, a);  Store<T>::ld4(P + t.o0 + ld, bq);
            Store<T>::ld4(P + t.o1, cc); Store<T>::ld4(P + t.o1 + ld, d);
        }
        f32x2 v[CP / 2];
        const f32x2 w00 = {t.w00, t.w00}, w01 = {t.w01, t.w01}, w10 = {t.w10, t.w10}, w11 = {t.w11, t.w11};
        ACH_UNROLL
        for (int c = 0; c < CP / 2; ++c) {
            const f32x2 av = {a[2 * c], a[2 * c + 1]}, bv = {bq[2 * c], bq[2 * c + 1]}, cv = {cc[2 * c], cc[2 * c + 1]}, dv = {d[2 * c], d[2 * c + 1]};
            v[c] = w00 * av + w01 * bv + w10 * cv + w11 * dv;
        }
        const float* w = Wf + k * p.Cp;                           // host layout [C][9][Cp], Cp = channel stride of the buffers (zero beyond C)
        ACH_UNROLL
        for (int co = 0; co < C; ++co)
            ACH_UNROLL
            for (int c = 0; c < CP / 2; ++c) { const f32x2 wv = {w[co * 9 * p.Cp + 2 * c], w[co * 9 * p.Cp + 2 * c + 1]}; acc[co] += wv * v[c]; }
    }
    float outv[CP];
    ACH_UNROLL
    for (int i = 0; i < CP; ++i) outv[i] = 0.f;
    ACH_UNROLL
    for (int i = 0; i < C; ++i) { const float r = acc[i][0] + acc[i][1]; outv[i] = r > 0.f ? r : 0.f; }
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
__global__ __launch_bounds__(256) void deform_sample_kernel(const DeformParams p) { f16_sat_mode<T>();
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
    const int ld = int(p.ldp);
    const BilinearTap t = make_tap(float(y - 1 + k / 3) + dy, float(x - 1 + k % 3) + dx, modulation(ml), p.H, p.Wd, int(p.prow), ld);
    const T* P = static_cast<const T*>(p.pooled) + b * p.pimg + c;
    float a[4], bq[4], cc[4], d[4], v[4];
    Store<T>::ld4(P + t.o0, a);
    Store<T>::ld4(P + t.o0 + ld, bq);
    Store<T>::ld4(P + t.o1, cc);
    Store<T>::ld4(P + t.o1 + ld, d);
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) v[i] = t.w00 * a[i] + t.w01 * bq[i] + t.w10 * cc[i] + t.w11 * d[i];
    Store<T>::st4(static_cast<T*>(p.Y) + pix * p.ldy + k * p.Cp + c, v);
}

}  // namespace ach
