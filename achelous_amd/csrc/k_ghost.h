// k_ghost.h — the Ghost bottlenecks of the Dual-FPN's top-down path as band kernels (16-bit engines; round 4, VERDICT r3 item 3 ii).
//
//   GhostModule      y[:, :init] = act(W x + b)                     backbone/conv_utils/ghost_conv.py:6-29   (primary 1x1 + BN [+ ReLU])
//                    y[:, init:] = act(dw3x3(y[:, :init]) + b')                                              (cheap operation + BN [+ ReLU])
//   shortcut         y = Wp dw3x3(x) + b + r                        ghost_conv.py:47-56, 58-70               (dw3x3 + BN, 1x1 + BN, + ghost2's output)
//
// A GhostBottleneck (ghost1 -> ghost2 -> + shortcut) was six launches of 10-19 us on 20 x 20 / 40 x 40 maps — GEMM, depthwise, GEMM, depthwise,
// depthwise, GEMM — every one a latency floor on the caller's stream, each intermediate written and read back.  Here a workgroup owns
// (frame, band of rows) and the spatial neighbour of each 1x1 conv stays on the CU:
//   ghost_kernel   1. primary conv on MFMA for the band + one halo row either side, result to an LDS tile with a zero border (and, for the
//                     band's own rows, to y[:, :init]);   2. depthwise 3x3 from the LDS tile -> y[:, init:].
//   dwpw_kernel    0. the band's halo of x -> LDS;   1. depthwise 3x3 from LDS, the sums written back to LDS as the B fragments of
//                     2. the pointwise conv on MFMA, + bias + residual.
// Three launches per bottleneck instead of six; the halo rows of the primary conv are recomputed (7 rows for 5).
#pragma once
#include "ach_platform.h"
#include "k_gemm.h"

namespace ach {

constexpr int GH_THREADS = 512;                  // 8 waves: the 1x1 convs are dealt to the waves as (16-pixel tile, 32-channel chunk) items
constexpr int GH_WAVES = GH_THREADS / 64;
constexpr int GH_LDS_BYTES = 48 * 1024;          // ghost_kernel: the x1 tile
constexpr int DP_HALO_BYTES = 36 * 1024, DP_FRAG_BYTES = 20 * 1024;     // dwpw_kernel: halo tile of x, B fragments of the band (two workgroups per CU)
constexpr int UC_LDS_BYTES = 40 * 1024;

// One (tile, chunk) item of a 1x1 conv: A = the chunk's 2 x k1 weight fragments (NT = 2 packing: a lane ends up with 8 consecutive output
// channels), B = the tile's k1 activation fragments.  Every load of the item is issued before the first MFMA (one L2 round trip per item).
// KMAX: compile-time bound of the k-steps (register arrays).
template <class T, int KMAX, class LoadB>
__device__ __forceinline__ void band_gemm_item(const uint4* __restrict__ wchunk, int k1, int lane, LoadB load_b, f32x4& a0, f32x4& a1) {
    uint4 wf[KMAX][2], xf[KMAX];
    ACH_UNROLL
    for (int s = 0; s < KMAX; ++s) {
        if (s < k1) { wf[s][0] = wchunk[(s * 2) * 64 + lane]; wf[s][1] = wchunk[(s * 2 + 1) * 64 + lane]; xf[s] = load_b(s); }
    }
    ACH_UNROLL
    for (int s = 0; s < KMAX; ++s) {
        if (s < k1) { mfma16<T>(wf[s][0], xf[s], a0); mfma16<T>(wf[s][1], xf[s], a1); }
    }
}

struct GhostParams {
    const void* X; long ldx;              // [B, H, W, ldx], K = Cin real channels
    void* Y; long ldy;                    // [B, H, W, ldy]: channels [0, init) primary, [init, 2 init) cheap
    const uint4* Wp; const float* bp;     // primary conv: NT = 2 fragments [chunk][k-step][2][64], bias [32 * chunks] (zero tail)
    const float* Wdw; const float* bdw;   // cheap operation: [9][init], [init]   (BN folded)
    int B, H, W, K, k1, init, chunks, act, rb, bands;
};

template <class T, int KMAX>
__global__ __launch_bounds__(GH_THREADS) void ghost_kernel(const GhostParams p) { f16_sat_mode<T>();
    static_assert(Store<T>::VEC == 8, "16-bit storage");
    __shared__ __attribute__((aligned(16))) unsigned char smem[GH_LDS_BYTES];
    T* const x1s = reinterpret_cast<T*>(smem);                       // [(rows + 2) x (W + 2)][init]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);
    const int band = int(wg % unsigned(p.bands)), b = int(wg / unsigned(p.bands));
    const int H = p.H, W = p.W, init = p.init;
    const int y0 = band * p.rb, rows = (y0 + p.rb <= H) ? p.rb : H - y0;
    const int WC = W + 2, HR = rows + 2;
    // zero border / rows outside the map: the depthwise conv's zero padding
    for (int i = tid; i < HR * WC * init / 8; i += GH_THREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // ---- 1. primary 1x1 on the band + halo rows: (tile, chunk) items over the waves
    const T* X = static_cast<const T*>(p.X) + long(b) * H * W * p.ldx;
    T* Y = static_cast<T*>(p.Y) + long(b) * H * W * p.ldy;
    const int npos = HR * W, nt = (npos + 15) / 16;
    for (int item = wave; item < nt * p.chunks; item += GH_WAVES) {
        const int t = item / p.chunks, c = item - t * p.chunks;
        const int pos = t * 16 + px;
        const int hr = pos / W, col = pos - hr * W;
        const int iy = y0 - 1 + hr;
        const bool valid = pos < npos && iy >= 0 && iy < H;
        const T* xrow = X + (long(valid ? iy : 0) * W + (valid ? col : 0)) * p.ldx + g * 8;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        band_gemm_item<T, KMAX>(p.Wp + long(c) * p.k1 * 2 * 64, p.k1, lane,
                                [&](int s) { return (valid && s * 32 + g * 8 < p.K) ? *reinterpret_cast<const uint4*>(xrow + s * 32) : make_uint4(0u, 0u, 0u, 0u); }, a0, a1);
        const int nb = c * 32 + g * 8;                                 // this lane's 8 consecutive output channels
        if (!valid || nb >= init) continue;
        float o[8];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { o[r] = a0[r] + p.bp[nb + r]; o[4 + r] = a1[r] + p.bp[nb + 4 + r]; }
        apply_act_n<T, 8>(o, p.act);
        const uint4 packed = frag_pack<T>(o);
        *reinterpret_cast<uint4*>(x1s + (hr * WC + col + 1) * init + nb) = packed;
        if (hr >= 1 && hr <= rows) *reinterpret_cast<uint4*>(Y + (long(iy) * W + col) * p.ldy + nb) = packed;
    }
    __syncthreads();
    // ---- 2. cheap operation: depthwise 3x3 (+ folded BN, act) of x1 -> channels [init, 2 init)
    const int c4n = init / 4, items = rows * W * c4n;
    for (int it = tid; it < items; it += GH_THREADS) {
        const int cg = it % c4n, pix = it / c4n;
        const int r = pix / W, col = pix - r * W;
        const float4 bb = *reinterpret_cast<const float4*>(p.bdw + cg * 4);
        float acc[4] = {bb.x, bb.y, bb.z, bb.w};
        ACH_UNROLL
        for (int k = 0; k < 9; ++k) {
            float v[4];
            Store<T>::ld4(x1s + ((r + k / 3) * WC + col + k % 3) * init + cg * 4, v);
            const float4 w = *reinterpret_cast<const float4*>(p.Wdw + k * init + cg * 4);
            acc[0] += v[0] * w.x; acc[1] += v[1] * w.y; acc[2] += v[2] * w.z; acc[3] += v[3] * w.w;
        }
        apply_act_n<T, 4>(acc, p.act);
        Store<T>::st4(Y + (long(y0 + r) * W + col) * p.ldy + init + cg * 4, acc);
    }
}

struct DwPwParams {
    const void* X; long ldx;              // [B, H, W, ldx], C real channels
    const void* R; long ldr;              // residual [B, H, W, ldr] (N channels) or nullptr
    void* Y; long ldy;                    // [B, H, W, ldy], N channels
    const float* Wdw; const float* bdw;   // depthwise 3x3: [9][C], [C]   (BN folded)
    const uint4* Wp; const float* bp;     // pointwise conv: NT = 2 fragments [chunk][k-step][2][64], bias [32 * chunks]
    int B, H, W, C, k1, N, chunks, rb, bands;
};

template <class T, int KMAX>
__global__ __launch_bounds__(GH_THREADS) void dwpw_kernel(const DwPwParams p) { f16_sat_mode<T>();
    static_assert(Store<T>::VEC == 8, "16-bit storage");
    __shared__ __attribute__((aligned(16))) unsigned char halo[DP_HALO_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char frag[DP_FRAG_BYTES];
    T* const xin = reinterpret_cast<T*>(halo);                       // [(rows + 2) x (W + 2)][C]
    uint4* const xs = reinterpret_cast<uint4*>(frag);                // [tile][k-step][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);
    const int band = int(wg % unsigned(p.bands)), b = int(wg / unsigned(p.bands));
    const int H = p.H, W = p.W, C = p.C;
    const int y0 = band * p.rb, rows = (y0 + p.rb <= H) ? p.rb : H - y0;
    const int WC = W + 2, HR = rows + 2, c8n = C / 8;
    const int npx = rows * W, nt = (npx + 15) / 16;
    const T* X = static_cast<const T*>(p.X) + long(b) * H * W * p.ldx;
    // ---- 0. halo tile (zero outside the map) and zeroed fragments (ragged last tile, k padding)
    for (int i = tid; i < HR * WC * c8n; i += GH_THREADS) {
        const int c8 = i % c8n, pos = i / c8n, wc = pos % WC, hr = pos / WC;
        const int iy = y0 - 1 + hr, ix = wc - 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(X + (long(iy) * W + ix) * p.ldx + c8 * 8);
        reinterpret_cast<uint4*>(halo)[i] = v;
    }
    for (int i = tid; i < nt * p.k1 * 64; i += GH_THREADS) xs[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // ---- 1. depthwise 3x3 (+ folded BN) -> B fragments: lane (pixel % 16, channel group) of (tile, k-step)
    const int c4n = C / 4, items = npx * c4n;
    for (int it = tid; it < items; it += GH_THREADS) {
        const int cg = it % c4n, pix = it / c4n;
        const int r = pix / W, col = pix - r * W;
        const float4 bb = *reinterpret_cast<const float4*>(p.bdw + cg * 4);
        float acc[4] = {bb.x, bb.y, bb.z, bb.w};
        ACH_UNROLL
        for (int k = 0; k < 9; ++k) {
            float v[4];
            Store<T>::ld4(xin + ((r + k / 3) * WC + col + k % 3) * C + cg * 4, v);
            const float4 w = *reinterpret_cast<const float4*>(p.Wdw + k * C + cg * 4);
            acc[0] += v[0] * w.x; acc[1] += v[1] * w.y; acc[2] += v[2] * w.z; acc[3] += v[3] * w.w;
        }
        const int c0 = cg * 4, s = c0 >> 5, gg = (c0 & 31) >> 3, e = c0 & 7;
        const int t = pix >> 4, pp = pix & 15;
        uint2 o;
        o.x = H16<T>::pack(acc[0], acc[1]);
        o.y = H16<T>::pack(acc[2], acc[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(xs + (t * p.k1 + s) * 64 + gg * 16 + pp) + e * 2) = o;
    }
    __syncthreads();
    // ---- 2. pointwise conv on MFMA, + bias (+ residual): (tile, chunk) items over the waves
    T* Y = static_cast<T*>(p.Y) + long(b) * H * W * p.ldy;
    const T* R = p.R ? static_cast<const T*>(p.R) + long(b) * H * W * p.ldr : nullptr;
    for (int item = wave; item < nt * p.chunks; item += GH_WAVES) {
        const int t = item / p.chunks, c = item - t * p.chunks;
        const int pix = t * 16 + px;
        const bool valid = pix < npx;
        const long m = long(y0) * W + pix;
        const int nb = c * 32 + g * 8;
        float r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (R && valid && nb < p.N) Store<T>::ld8(R + m * p.ldr + nb, r8);          // requested with the operands
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const uint4* xt = xs + (t * p.k1) * 64 + lane;
        band_gemm_item<T, KMAX>(p.Wp + long(c) * p.k1 * 2 * 64, p.k1, lane, [&](int s) { return xt[s * 64]; }, a0, a1);
        if (!valid || nb >= p.N) continue;
        float o[8];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { o[r] = a0[r] + p.bp[nb + r] + r8[r]; o[4 + r] = a1[r] + p.bp[nb + 4 + r] + r8[4 + r]; }
        Store<T>::st8(Y + m * p.ldy + nb, o);
    }
}

// ------------------------------------------------------------------------------------------ Upsample = 1x1 conv + BN + ReLU, bilinear x2
// neck/ghostdualfpn.py:28-39 (align_corners=True).  A workgroup owns (frame, band of OUTPUT rows): 1. the conv on MFMA for the source rows the
// band's interpolation touches -> LDS (storage type: the rounding the separate launch's output had);  2. the band's pixels interpolated from
// LDS with upsample2x_kernel's float arithmetic, written into the destination (a channel slice of the neck's concat buffer).
struct UpConvParams {
    const void* X; long ldx;              // [B, h, w, ldx], K real channels
    void* Y; long ldy;                    // [B, 2h, 2w, ldy], N channels written at Y
    const uint4* Wp; const float* bp;     // NT = 2 fragments, bias [32 * chunks]
    int B, h, w, K, k1, N, chunks, rb, bands;
};

template <class T, int KMAX>
__global__ __launch_bounds__(GH_THREADS) void upconv_kernel(const UpConvParams p) { f16_sat_mode<T>();
    static_assert(Store<T>::VEC == 8, "16-bit storage");
    __shared__ __attribute__((aligned(16))) unsigned char smem[UC_LDS_BYTES];
    T* const ts = reinterpret_cast<T*>(smem);                        // [source rows][w][N]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const unsigned wg = xcd_block(blockIdx.x, gridDim.x);
    const int band = int(wg % unsigned(p.bands)), b = int(wg / unsigned(p.bands));
    const int h = p.h, w = p.w, Ho = 2 * h, Wo = 2 * w, N = p.N;
    const int oy0 = band * p.rb, orows = (oy0 + p.rb <= Ho) ? p.rb : Ho - oy0;
    const float sy = Ho > 1 ? float(h - 1) / float(Ho - 1) : 0.f, sx = Wo > 1 ? float(w - 1) / float(Wo - 1) : 0.f;
    // source rows of the band: first = floor(sy * oy0), last = the upper neighbour of the band's last row
    int ys0 = int(sy * float(oy0));
    if (ys0 > h - 1) ys0 = h - 1;
    int ys1 = int(sy * float(oy0 + orows - 1));
    if (ys1 > h - 1) ys1 = h - 1;
    ys1 = ys1 + (ys1 < h - 1 ? 1 : 0);
    const int srows = ys1 - ys0 + 1;
    // ---- 1. conv + BN + ReLU on the source rows
    const T* X = static_cast<const T*>(p.X) + (long(b) * h + ys0) * w * p.ldx;
    const int npos = srows * w, nt = (npos + 15) / 16;
    for (int item = wave; item < nt * p.chunks; item += GH_WAVES) {
        const int t = item / p.chunks, c = item - t * p.chunks;
        const int pos = t * 16 + px;
        const bool valid = pos < npos;
        const T* xrow = X + long(valid ? pos : 0) * p.ldx + g * 8;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        band_gemm_item<T, KMAX>(p.Wp + long(c) * p.k1 * 2 * 64, p.k1, lane,
                                [&](int s) { return (valid && s * 32 + g * 8 < p.K) ? *reinterpret_cast<const uint4*>(xrow + s * 32) : make_uint4(0u, 0u, 0u, 0u); }, a0, a1);
        const int nb = c * 32 + g * 8;
        if (!valid || nb >= N) continue;
        float o[8];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { o[r] = a0[r] + p.bp[nb + r]; o[4 + r] = a1[r] + p.bp[nb + 4 + r]; }
        apply_act_n<T, 8>(o, ACT_RELU);
        *reinterpret_cast<uint4*>(ts + long(pos) * N + nb) = frag_pack<T>(o);
    }
    __syncthreads();
    // ---- 2. bilinear x2 of the band's rows (upsample2x_kernel's arithmetic)
    T* Y = static_cast<T*>(p.Y) + long(b) * Ho * Wo * p.ldy;
    const int cq = N / 4, items = orows * Wo * cq;
    for (int it = tid; it < items; it += GH_THREADS) {
        const int c = (it % cq) * 4, pix = it / cq;
        const int r = pix / Wo, ox = pix - r * Wo, oy = oy0 + r;
        const float fy = sy * float(oy), fx = sx * float(ox);
        int y0 = int(fy), x0 = int(fx);
        if (y0 > h - 1) y0 = h - 1;
        if (x0 > w - 1) x0 = w - 1;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - float(y0), lx = fx - float(x0);
        const float hy = 1.f - ly, hx = 1.f - lx;
        float a[4], bq[4], cc[4], d[4], o[4];
        Store<T>::ld4(ts + (long(y0 - ys0) * w + x0) * N + c, a);
        Store<T>::ld4(ts + (long(y0 - ys0) * w + x1) * N + c, bq);
        Store<T>::ld4(ts + (long(y1 - ys0) * w + x0) * N + c, cc);
        Store<T>::ld4(ts + (long(y1 - ys0) * w + x1) * N + c, d);
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) o[i] = hy * (hx * a[i] + lx * bq[i]) + ly * (hx * cc[i] + lx * d[i]);
        Store<T>::st4(Y + (long(oy) * Wo + ox) * p.ldy + c, o);
    }
}


// ------------------------------------------------------------------------------------------ SPP / SPPF (neck/spp.py:41-67) as one launch
// cv1 (1x1 + BN + SiLU, C -> c_) -> max pools 5 / 9 / 13 (stride 1, -inf padding; SPPF's three chained 5x5 pools are the same windows) -> cat ->
// cv2 (1x1 + BN + SiLU, 4 c_ -> C) on a map of at most 16 x 16.  A workgroup owns (frame, share of cv2's output chunks) and recomputes cv1 + the
// pools for its frame (cheap: 100 pixels): cv1's output goes to LDS in the storage type (the rounding the concat buffer had), the pools are
// computed separably (row maxima, then column maxima) and written straight into the B fragments of cv2 — the concat buffer never exists.
struct SppFusedParams {
    const void* X; long ldx;              // [B, H, W, ldx], K1 = C channels
    void* Y; long ldy;                    // [B, H, W, ldy], N = C channels
    const uint4* W1; const float* b1;     // cv1: NT = 2 fragments, k1a k-steps, chunks1 chunks
    const uint4* W2; const float* b2;     // cv2: k1b k-steps (K = 4 c_), chunks2 chunks
    int B, H, W, C, cmid, k1a, chunks1, k1b, chunks2, split;
};
constexpr int SPPF_MAXPIX = 112, SPPF_MAXMID = 88, SPPF_K2MAX = 11;     // LDS: 19.7 + 59.1 + 78.8 KB = 157.7 of 160 KB (EdgeNeXt-S0: 10 x 10 x 176 -> 88)
template <class T>
__global__ __launch_bounds__(GH_THREADS) void spp_fused_kernel(const SppFusedParams p) { f16_sat_mode<T>();
    static_assert(Store<T>::VEC == 8, "16-bit storage");
    __shared__ __attribute__((aligned(16))) T x1s[SPPF_MAXPIX * SPPF_MAXMID];                  // cv1 output [pix][cmid]
    __shared__ __attribute__((aligned(16))) T rmax[3][SPPF_MAXPIX * SPPF_MAXMID];              // row maxima of radius 2 / 4 / 6
    __shared__ __attribute__((aligned(16))) uint4 xs[(SPPF_MAXPIX / 16) * SPPF_K2MAX * 64];    // B fragments of cv2: [tile][k-step][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int b = int(blockIdx.x) / p.split, part = int(blockIdx.x) % p.split;
    const int HW = p.H * p.W, nt = (HW + 15) / 16, cm = p.cmid;
    const T* X = static_cast<const T*>(p.X) + long(b) * HW * p.ldx;
    for (int i = tid; i < nt * p.k1b * 64; i += GH_THREADS) xs[i] = make_uint4(0u, 0u, 0u, 0u);
    // ---- 1. cv1 -> LDS
    for (int item = wave; item < nt * p.chunks1; item += GH_WAVES) {
        const int t = item / p.chunks1, c = item - t * p.chunks1;
        const int pix = t * 16 + px;
        const bool valid = pix < HW;
        const T* xrow = X + long(valid ? pix : 0) * p.ldx + g * 8;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        band_gemm_item<T, 6>(p.W1 + long(c) * p.k1a * 2 * 64, p.k1a, lane,
                             [&](int s) { return (valid && s * 32 + g * 8 < p.C) ? *reinterpret_cast<const uint4*>(xrow + s * 32) : make_uint4(0u, 0u, 0u, 0u); }, a0, a1);
        const int nb = c * 32 + g * 8;
        if (!valid || nb >= cm) continue;
        float o[8];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { o[r] = a0[r] + p.b1[nb + r]; o[4 + r] = a1[r] + p.b1[nb + 4 + r]; }
        apply_act_n<T, 8>(o, ACT_SILU);
        *reinterpret_cast<uint4*>(x1s + pix * cm + nb) = frag_pack<T>(o);
    }
    __syncthreads();
    // ---- 2a. row maxima (radius 2 / 4 / 6) per (pixel, channel quad); x1 itself -> fragments (concat slot 0)
    const int cq = cm / 4, items = HW * cq;
    auto to_frag = [&](int pix, int ccat, const float (&v)[4]) {      // 4 consecutive concat channels of a pixel -> its B-fragment slot
        const int s = ccat >> 5, gg = (ccat & 31) >> 3, e = ccat & 7;
        uint2 o;
        o.x = H16<T>::pack(v[0], v[1]); o.y = H16<T>::pack(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(xs + ((pix >> 4) * p.k1b + s) * 64 + gg * 16 + (pix & 15)) + e * 2) = o;
    };
    for (int it = tid; it < items; it += GH_THREADS) {
        const int q = it % cq, pix = it / cq, y = pix / p.W, x = pix - y * p.W;
        float m[4], m5[4], m9[4];
        Store<T>::ld4(x1s + pix * cm + q * 4, m);
        to_frag(pix, q * 4, m);
        ACH_UNROLL
        for (int d = 1; d <= 6; ++d) {
            float v[4];
            if (x - d >= 0) { Store<T>::ld4(x1s + (pix - d) * cm + q * 4, v); ACH_UNROLL for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]); }
            if (x + d < p.W) { Store<T>::ld4(x1s + (pix + d) * cm + q * 4, v); ACH_UNROLL for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]); }
            if (d == 2) { ACH_UNROLL for (int e = 0; e < 4; ++e) m5[e] = m[e]; }
            if (d == 4) { ACH_UNROLL for (int e = 0; e < 4; ++e) m9[e] = m[e]; }
        }
        Store<T>::st4(rmax[0] + pix * cm + q * 4, m5);
        Store<T>::st4(rmax[1] + pix * cm + q * 4, m9);
        Store<T>::st4(rmax[2] + pix * cm + q * 4, m);
    }
    __syncthreads();
    // ---- 2b. column maxima -> fragments (concat slots 1-3)
    for (int it = tid; it < items; it += GH_THREADS) {
        const int q = it % cq, pix = it / cq, y = pix / p.W;
        float m5[4], m9[4], m13[4];
        Store<T>::ld4(rmax[0] + pix * cm + q * 4, m5);
        Store<T>::ld4(rmax[1] + pix * cm + q * 4, m9);
        Store<T>::ld4(rmax[2] + pix * cm + q * 4, m13);
        ACH_UNROLL
        for (int d = 1; d <= 6; ++d) {
            ACH_UNROLL
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                const int yy = y + sgn * d;
                if (yy < 0 || yy >= p.H) continue;
                const int j = (pix + sgn * d * p.W) * cm + q * 4;
                float v[4];
                if (d <= 2) { Store<T>::ld4(rmax[0] + j, v); ACH_UNROLL for (int e = 0; e < 4; ++e) m5[e] = fmaxf(m5[e], v[e]); }
                if (d <= 4) { Store<T>::ld4(rmax[1] + j, v); ACH_UNROLL for (int e = 0; e < 4; ++e) m9[e] = fmaxf(m9[e], v[e]); }
                Store<T>::ld4(rmax[2] + j, v);
                ACH_UNROLL for (int e = 0; e < 4; ++e) m13[e] = fmaxf(m13[e], v[e]);
            }
        }
        to_frag(pix, cm + q * 4, m5);
        to_frag(pix, 2 * cm + q * 4, m9);
        to_frag(pix, 3 * cm + q * 4, m13);
    }
    __syncthreads();
    // ---- 3. cv2 on the fragments: this workgroup's share of the output chunks
    T* Y = static_cast<T*>(p.Y) + long(b) * HW * p.ldy;
    const int cper = (p.chunks2 + p.split - 1) / p.split, c_lo = part * cper, c_hi = (c_lo + cper < p.chunks2) ? c_lo + cper : p.chunks2;
    const int nc = c_hi - c_lo;
    for (int item = wave; item < nt * nc; item += GH_WAVES) {
        const int t = item / nc, c = c_lo + (item - t * nc);
        const int pix = t * 16 + px;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const uint4* xt = xs + (t * p.k1b) * 64 + lane;
        band_gemm_item<T, SPPF_K2MAX>(p.W2 + long(c) * p.k1b * 2 * 64, p.k1b, lane, [&](int s) { return xt[s * 64]; }, a0, a1);
        const int nb = c * 32 + g * 8;
        if (pix >= HW || nb >= p.C) continue;
        float o[8];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { o[r] = a0[r] + p.b2[nb + r]; o[4 + r] = a1[r] + p.b2[nb + 4 + r]; }
        apply_act_n<T, 8>(o, ACT_SILU);
        Store<T>::st8(Y + long(pix) * p.ldy + nb, o);
    }
}
inline bool spp_fused_supported(int H, int W, int C, int cmid) {
    return H * W <= SPPF_MAXPIX && cmid <= SPPF_MAXMID && cmid % 8 == 0 && C % 8 == 0 && (C + 31) / 32 <= 6 && (4 * cmid + 31) / 32 <= SPPF_K2MAX;
}

// KMAX instantiations: 3 / 6 / 10 k-steps (96 / 192 / 320 input channels)
#define ACH_BAND_LAUNCH(KERN, k1, grid, block, s, prm)                                                          \
    do {                                                                                                            \
        if ((k1) <= 3) ACH_LAUNCH((KERN<T, 3>), grid, block, s, prm);                                               \
        else if ((k1) <= 6) ACH_LAUNCH((KERN<T, 6>), grid, block, s, prm);                                          \
        else ACH_LAUNCH((KERN<T, 10>), grid, block, s, prm);                                                        \
    } while (0)

// output rows per band: at most `cap`, the band's source rows (rb / 2 + 2 at most) must fit the LDS tile
inline int upconv_band_rows(int h, int w, int N, int cap = 8) {
    for (int rb = (2 * h < cap ? 2 * h : cap); rb >= 2; rb -= 2)
        if ((rb / 2 + 2) * w * N * 2 <= UC_LDS_BYTES) return rb;
    return 0;
}

// rows per band: the largest count (at most `cap`) whose tiles fit the kernels' LDS buffers; 0 = does not fit
inline int ghost_band_rows(int H, int W, int init, int cap = 5) {
    for (int rb = (H < cap ? H : cap); rb >= 1; --rb)
        if ((rb + 2) * (W + 2) * init * 2 <= GH_LDS_BYTES) return rb;
    return 0;
}
inline int dwpw_band_rows(int H, int W, int C, int cap = 5) {
    const int k1 = (C + 31) / 32;
    for (int rb = (H < cap ? H : cap); rb >= 1; --rb)
        if ((rb + 2) * (W + 2) * C * 2 <= DP_HALO_BYTES && ((rb * W + 15) / 16) * k1 * 64 * 16 <= DP_FRAG_BYTES) return rb;
    return 0;
}

}  // namespace ach
