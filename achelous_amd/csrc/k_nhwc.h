// k_nhwc.h — the bandwidth-bound NHWC kernels of the image path: depthwise kxk conv, EdgeNeXt stem, channel
// LayerNorm, bilinear x2 (align_corners), SPP max-pools, per-channel statistics, ShuffleAttention, ECA fusion,
// element-wise add.  One thread owns 4 consecutive channels of one pixel (8 B bf16 / 16 B fp32), so a wave reads
// and writes whole contiguous NHWC rows.  All arithmetic in fp32.
#pragma once
#include "ach_platform.h"

namespace ach {

// ------------------------------------------------------------------------------------------ depthwise conv
struct DwParams {
    const void* X; long ldx;        // input  [B,H,W,(C)] view, channel stride ldx
    const void* X2; long ldx2;      // optional second input added tap-wise (SDTA cascade: conv(sp_prev + spx_i))
    const float* W;                 // [KS*KS][C] fp32 (BatchNorm scale folded)
    const float* bias;              // [C] fp32
    void* Y; long ldy;              // output [B,Ho,Wo,(C)]
    int B, H, Wd, C, Ho, Wo, stride, act;
    int cin_mod;                    // > 0: output channel c reads input channel c % cin_mod (two filter banks over one input)
    int tile;                       // 1: LDS-tiled kernel (10x10 maps; measured slower than the strip kernel from 20x20 up), 0: strip kernel
};

// stride-1 kernel: one thread = 4 channels x a strip of OW consecutive output pixels of one row.  A row of the receptive
// field is loaded once (KS + OW - 1 vector loads) and feeds all OW outputs from registers: k^2 -> k (k + OW - 1) / OW loads
// per output (49 -> 17.5 for the 7x7 of EdgeNeXt stage 2), with the KS weight vectors of the row held in registers.
template <class T, int KS, int OW>
__device__ __forceinline__ void dwconv_strip_body(const DwParams& p, unsigned bx, unsigned nbx) {
    const int cq = p.C >> 2;
    const int strips = (p.Wo + OW - 1) / OW;
    const long total = long(p.B) * p.Ho * strips * cq;
    const long idx = long(xcd_block(bx, nbx)) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = int(idx % cq) * 4;
    long r = idx / cq;
    const int ox0 = int(r % strips) * OW; r /= strips;
    const int oy = int(r % p.Ho);
    const long b = r / p.Ho;
    constexpr int PAD = KS / 2;
    const int ci = p.cin_mod > 0 ? c % p.cin_mod : c;
    // taps through range-checked buffer resources (ach_platform.h): a column outside the map gets an out-of-range offset and
    // reads zeros — one add + one load per fetched pixel instead of a clamp, 64-bit address arithmetic and four selects
    constexpr unsigned ESZ = sizeof(T);
    const unsigned npix = unsigned(p.B) * unsigned(p.H) * unsigned(p.Wd);
    const BufRsrc xb = make_buf(p.X, npix * unsigned(p.ldx) * ESZ);
    const BufRsrc xb2 = make_buf(p.X2 ? p.X2 : p.X, p.X2 ? npix * unsigned(p.ldx2) * ESZ : 0u);
    const unsigned pitch = unsigned(p.ldx) * ESZ, pitch2 = unsigned(p.ldx2) * ESZ;
    const unsigned pix0 = unsigned(b) * unsigned(p.H) * unsigned(p.Wd);
    const unsigned base = pix0 * pitch + unsigned(ci) * ESZ, base2 = pix0 * pitch2 + unsigned(ci) * ESZ;
    unsigned coff[KS + OW - 1], coff2[KS + OW - 1];
    ACH_UNROLL
    for (int j = 0; j < KS + OW - 1; ++j) {
        const int ix = ox0 - PAD + j;
        const bool ok = ix >= 0 && ix < p.Wd;
        coff[j] = ok ? unsigned(ix) * pitch : BUF_OOB;
        coff2[j] = ok ? unsigned(ix) * pitch2 : BUF_OOB;           // dead (and removed) when there is no second input
    }
    float acc[OW][4];
    {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + c);
        ACH_UNROLL
        for (int o = 0; o < OW; ++o) { acc[o][0] = bb.x; acc[o][1] = bb.y; acc[o][2] = bb.z; acc[o][3] = bb.w; }
    }
    // (measured: unrolling the row loop with clamped, always-issued loads is 2-3x SLOWER here — the extra live registers cost
    //  more occupancy than the batched loads gain; rows outside the map are skipped instead)
    for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy - PAD + ky;
        if (iy < 0 || iy >= p.H) continue;
        float4 w[KS];
        ACH_UNROLL
        for (int kx = 0; kx < KS; ++kx) w[kx] = *reinterpret_cast<const float4*>(p.W + long(ky * KS + kx) * p.C + c);
        const unsigned rowb = base + unsigned(iy) * unsigned(p.Wd) * pitch, rowb2 = base2 + unsigned(iy) * unsigned(p.Wd) * pitch2;
        ACH_UNROLL
        for (int j = 0; j < KS + OW - 1; ++j) {
            float v[4];
            buf_ld4<T>(xb, rowb + coff[j], v);
            if (p.X2) { float u[4]; buf_ld4<T>(xb2, rowb2 + coff2[j], u); v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
            ACH_UNROLL
            for (int o = 0; o < OW; ++o) {
                const int kx = j - o;
                if (kx >= 0 && kx < KS) { acc[o][0] += v[0] * w[kx].x; acc[o][1] += v[1] * w[kx].y; acc[o][2] += v[2] * w[kx].z; acc[o][3] += v[3] * w[kx].w; }
            }
        }
    }
    T* Y = static_cast<T*>(p.Y) + ((b * p.Ho + oy) * long(p.Wo) + ox0) * p.ldy + c;
    ACH_UNROLL
    for (int o = 0; o < OW; ++o) {
        if (ox0 + o >= p.Wo) break;
        float t[4];
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) t[i] = acc[o][i];
        apply_act_n<float, 4>(t, p.act);
        Store<T>::st4(Y + long(o) * p.ldy, t);
    }
}

template <class T, int KS, int OW>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(const DwParams p) { f16_sat_mode<T>(); dwconv_strip_body<T, KS, OW>(p, blockIdx.x, gridDim.x); }
// up to three independent maps in one launch (blockIdx.y = job; the detection head's three pyramid levels)
struct DwJobs { DwParams p[3]; unsigned nbx[3]; int n; };
template <class T, int KS, int OW>
__global__ __launch_bounds__(256) void dwconv_strip_multi_kernel(const DwJobs m) { f16_sat_mode<T>();
    const unsigned j = blockIdx.y;
    if (blockIdx.x >= m.nbx[j]) return;
    dwconv_strip_body<T, KS, OW>(m.p[j], blockIdx.x, m.nbx[j]);
}

// general kernel (any stride): one thread = 4 channels of one output pixel
#ifndef ACH_DWK_WAVES
#define ACH_DWK_WAVES 0
#endif
#if ACH_DWK_WAVES > 0
#define ACH_DWK_BOUNDS __launch_bounds__(256, ACH_DWK_WAVES)
#else
#define ACH_DWK_BOUNDS __launch_bounds__(256)
#endif
template <class T, int KS>
__global__ ACH_DWK_BOUNDS void dwconv_kernel(const DwParams p) { f16_sat_mode<T>();
    const int cq = p.C >> 2;
    const long total = long(p.B) * p.Ho * p.Wo * cq;
    const long idx = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = int(idx % cq) * 4;
    long pix = idx / cq;
    const int ox = int(pix % p.Wo); pix /= p.Wo;
    const int oy = int(pix % p.Ho);
    const long b = pix / p.Ho;
    constexpr int PAD = KS / 2;
    // branch-free: every tap is fetched through a range-checked buffer resource (ach_platform.h), taps outside the map read zeros
    constexpr unsigned ESZ = sizeof(T);
    const unsigned npix = unsigned(p.B) * unsigned(p.H) * unsigned(p.Wd);
    const BufRsrc xb = make_buf(p.X, npix * unsigned(p.ldx) * ESZ);
    const BufRsrc xb2 = make_buf(p.X2 ? p.X2 : p.X, p.X2 ? npix * unsigned(p.ldx2) * ESZ : 0u);
    const unsigned pitch = unsigned(p.ldx) * ESZ, pitch2 = unsigned(p.ldx2) * ESZ;
    const unsigned pix0 = unsigned(b) * unsigned(p.H) * unsigned(p.Wd);
    float acc[4] = {p.bias[c], p.bias[c + 1], p.bias[c + 2], p.bias[c + 3]};
    ACH_UNROLL
    for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy * p.stride - PAD + ky;
        const bool rok = iy >= 0 && iy < p.H;
        ACH_UNROLL
        for (int kx = 0; kx < KS; ++kx) {
            const int ix = ox * p.stride - PAD + kx;
            const bool ok = rok && ix >= 0 && ix < p.Wd;
            const unsigned pix = pix0 + unsigned(iy) * unsigned(p.Wd) + unsigned(ix);
            float v[4];
            buf_ld4<T>(xb, ok ? pix * pitch + unsigned(c) * ESZ : BUF_OOB, v);
            if (p.X2) { float u[4]; buf_ld4<T>(xb2, ok ? pix * pitch2 + unsigned(c) * ESZ : BUF_OOB, u); v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
            const float4 w = *reinterpret_cast<const float4*>(p.W + long(ky * KS + kx) * p.C + c);
            acc[0] += v[0] * w.x; acc[1] += v[1] * w.y; acc[2] += v[2] * w.z; acc[3] += v[3] * w.w;
        }
    }
    apply_act_n<float, 4>(acc, p.act);
    const long op = (b * p.Ho + oy) * p.Wo + ox;
    Store<T>::st4(static_cast<T*>(p.Y) + op * p.ldy + c, acc);
}

// stride-1 kernel for the smallest maps (10x10), where the strip kernel above is a chain of dependent global
// loads (rocprofv3: waves 65 % waiting, 2.8 waves per SIMD): a workgroup stages an 8x8 output tile + halo of 16 channels in LDS
// with every global load in flight at once (unpacked to fp32), plus the KS*KS weight vectors, and the taps then run from LDS:
// two ds_read_b128 + two packed FMAs per tap per thread (thread = 4 channels x 1 output).
constexpr int DWT_TS = 8;
template <class T, int KS>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(const DwParams p) { f16_sat_mode<T>();
    constexpr int TS = DWT_TS, HS = TS + KS - 1, PAD = KS / 2;
    __shared__ float4 xs[HS * HS * 4];
    __shared__ float4 ws[KS * KS * 4];
    const int cq = p.C >> 2, cgroups = (cq + 3) >> 2;
    const int tiles_x = (p.Wo + TS - 1) / TS, tiles_y = (p.Ho + TS - 1) / TS;
    unsigned wg = blockIdx.x;
    const int cg = int(wg % unsigned(cgroups)); wg /= unsigned(cgroups);
    const int bx = int(wg % unsigned(tiles_x)) * TS; wg /= unsigned(tiles_x);
    const int by = int(wg % unsigned(tiles_y)) * TS;
    const long b = wg / unsigned(tiles_y);
    const int q = threadIdx.x & 3, pos = threadIdx.x >> 2;
    const int c = (cg * 4 + q) * 4;
    const bool cok = c < p.C;
    const int ci = p.cin_mod > 0 ? c % p.cin_mod : c;
    const T* X = static_cast<const T*>(p.X) + b * p.H * long(p.Wd) * p.ldx + ci;
    const T* X2 = p.X2 ? static_cast<const T*>(p.X2) + b * p.H * long(p.Wd) * p.ldx2 + ci : nullptr;
    for (int i = pos; i < HS * HS; i += 64) {
        const int hy = i / HS, hx = i - hy * HS;
        const int iy = by + hy - PAD, ix = bx + hx - PAD;
        const bool ok = cok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd;
        const long ip = long(ok ? iy : 0) * p.Wd + (ok ? ix : 0);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (cok) {
            Store<T>::ld4(X + ip * p.ldx, v);
            if (X2) { float u[4]; Store<T>::ld4(X2 + ip * p.ldx2, u); v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3]; }
        }
        xs[i * 4 + q] = ok ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = pos; i < KS * KS; i += 64)
        ws[i * 4 + q] = cok ? *reinterpret_cast<const float4*>(p.W + long(i) * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int ty = pos / TS, tx = pos - ty * TS;
    const int oy = by + ty, ox = bx + tx;
    if (!cok || oy >= p.Ho || ox >= p.Wo) return;
    const float4 bb = *reinterpret_cast<const float4*>(p.bias + c);
    f32x2 a0 = {bb.x, bb.y}, a1 = {bb.z, bb.w};
    ACH_UNROLL
    for (int ky = 0; ky < KS; ++ky)
        ACH_UNROLL
        for (int kx = 0; kx < KS; ++kx) {
            const float4 v = xs[((ty + ky) * HS + tx + kx) * 4 + q];
            const float4 w = ws[(ky * KS + kx) * 4 + q];
            const f32x2 v0 = {v.x, v.y}, v1 = {v.z, v.w}, w0 = {w.x, w.y}, w1 = {w.z, w.w};
            a0 += v0 * w0; a1 += v1 * w1;
        }
    float t[4] = {a0[0], a0[1], a1[0], a1[1]};
    apply_act_n<float, 4>(t, p.act);
    Store<T>::st4(static_cast<T*>(p.Y) + ((b * p.Ho + oy) * long(p.Wo) + ox) * p.ldy + c, t);
}

template <class T>
inline void launch_dwconv(const DwParams& p, int ks, hipStream_t s) {
    const dim3 block(256);
    if (p.stride == 1 && p.tile && p.Ho == p.H && p.Wo == p.Wd) {
        const dim3 grid(unsigned(p.B) * unsigned(cdiv(p.Ho, DWT_TS)) * unsigned(cdiv(p.Wo, DWT_TS)) * unsigned((p.C / 4 + 3) / 4));
        switch (ks) {
            case 3: ACH_LAUNCH((dwconv_tile_kernel<T, 3>), grid, block, s, p); return;
            case 5: ACH_LAUNCH((dwconv_tile_kernel<T, 5>), grid, block, s, p); return;
            case 7: ACH_LAUNCH((dwconv_tile_kernel<T, 7>), grid, block, s, p); return;
            case 9: ACH_LAUNCH((dwconv_tile_kernel<T, 9>), grid, block, s, p); return;
            default: break;
        }
    }
    if (p.stride == 1) {
        constexpr int OW = 4;
        const long total = long(p.B) * p.Ho * cdiv(p.Wo, OW) * (p.C / 4);
        const dim3 grid(unsigned(cdivl(total, 256)));
        switch (ks) {
            case 3: ACH_LAUNCH((dwconv_strip_kernel<T, 3, OW>), grid, block, s, p); break;
            case 5: ACH_LAUNCH((dwconv_strip_kernel<T, 5, OW>), grid, block, s, p); break;
            case 7: ACH_LAUNCH((dwconv_strip_kernel<T, 7, OW>), grid, block, s, p); break;
            case 9: ACH_LAUNCH((dwconv_strip_kernel<T, 9, OW>), grid, block, s, p); break;
            default: break;
        }
        return;
    }
    const long total = long(p.B) * p.Ho * p.Wo * (p.C / 4);
    const dim3 grid(unsigned(cdivl(total, 256)));
    switch (ks) {
        case 3: ACH_LAUNCH((dwconv_kernel<T, 3>), grid, block, s, p); break;
        case 5: ACH_LAUNCH((dwconv_kernel<T, 5>), grid, block, s, p); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------ EdgeNeXt stem
// conv 4x4 stride 4 (3 -> 32, bias) + channels-first LayerNorm(eps) with affine; NCHW image in, NHWC out.
struct StemParams {
    const void* X; void* Y;
    const float* W;      // [48][32]: k = (c*4 + dy)*4 + dx
    const float* bias; const float* lnw; const float* lnb;
    int B, H, Wd; float eps;
};
template <class T, class IO = T>       // IO: the type of the caller's image
__global__ __launch_bounds__(256) void stem_kernel(const StemParams p) { f16_sat_mode<T>();
    constexpr int CO = 32;
    const int Ho = p.H / 4, Wo = p.Wd / 4;
    const long total = long(p.B) * Ho * Wo;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ox = int(idx % Wo);
    const int oy = int((idx / Wo) % Ho);
    const long b = idx / (long(Wo) * Ho);
    const IO* X = static_cast<const IO*>(p.X);
    float acc[CO];
    ACH_UNROLL
    for (int o = 0; o < CO; ++o) acc[o] = p.bias[o];
    for (int c = 0; c < 3; ++c)
        for (int dy = 0; dy < 4; ++dy) {
            const IO* row = X + ((b * 3 + c) * p.H + (oy * 4 + dy)) * long(p.Wd) + ox * 4;
            float v4[4];
            Store<IO>::ld4(row, v4);                               // the patch row: one 8 / 16-byte load (W is a multiple of 4)
            ACH_UNROLL
            for (int dx = 0; dx < 4; ++dx) {
                const float v = v4[dx];
                const float* w = p.W + ((c * 4 + dy) * 4 + dx) * CO;
                ACH_UNROLL
                for (int o = 0; o < CO; ++o) acc[o] += v * w[o];
            }
        }
    float mu = 0.f;
    ACH_UNROLL
    for (int o = 0; o < CO; ++o) mu += acc[o];
    mu *= (1.0f / CO);
    float var = 0.f;
    ACH_UNROLL
    for (int o = 0; o < CO; ++o) { const float d = acc[o] - mu; var += d * d; }
    const float rs = 1.0f / sqrtf(var * (1.0f / CO) + p.eps);
    T* y = static_cast<T*>(p.Y) + idx * CO;
    ACH_UNROLL
    for (int o = 0; o < CO; o += 4) {
        float v4[4];
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) v4[i] = (acc[o + i] - mu) * rs * p.lnw[o + i] + p.lnb[o + i];
        Store<T>::st4(y + o, v4);
    }
}

// The same stem on the matrix cores: a [B*Ho*Wo, 48] x [48, 32] GEMM whose rows are gathered straight from the NCHW image.  The
// reduction index is k = c*16 + dy*4 + dx (the conv weight's own order), so the 4 (fp32) / 8 (bf16) consecutive k values of a
// lane's B fragment are one / two patch rows of 4 pixels: one 16-byte / two 8-byte loads, no unpacking.  Weights are packed like
// any 32-wide GEMM chunk (NT = 2, k_gemm.h), which leaves every lane with 8 consecutive output channels of its pixel: the
// channels-first LayerNorm is two xor-shuffles across the four lane groups, and the row goes out in one 16 / 32-byte store.
// 4 MFMAs replace 1536 scalar FMAs per 16 pixels; the kernel becomes a stream over the image (40 -> 25 us at batch 64, 2.6 TB/s).
struct StemMfmaParams {
    const void* X; void* Y;
    const void* W; const float* bias; const float* lnw; const float* lnb;   // W: packed NT = 2, ksteps = ceil(48 / KC)
    int B, H, Wd, ksteps; float eps;
};
template <class T, class IO = T>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemMfmaParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC, KC = 4 * VEC, SEG = VEC / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int Ho = p.H / 4, Wo = p.Wd / 4;
    const long total = long(p.B) * Ho * Wo;
    const long mraw = (long(blockIdx.x) * 4 + wave) * 16 + px;
    const bool valid = mraw < total;
    const long m = valid ? mraw : 0;
    const int ox = int(m % Wo);
    const int oy = int((m / Wo) % Ho);
    const long b = m / (long(Wo) * Ho);
    const IO* X = static_cast<const IO*>(p.X) + (b * 3 * p.H + oy * 4) * long(p.Wd) + ox * 4;     // channel 0, patch row 0
    const long cstride = long(p.H) * p.Wd;
    f32x4 acc[2];
    acc[0][0] = acc[0][1] = acc[0][2] = acc[0][3] = 0.f;
    acc[1][0] = acc[1][1] = acc[1][2] = acc[1][3] = 0.f;
    const uint4* Wf = static_cast<const uint4*>(p.W) + lane;
    ACH_UNROLL
    for (int s = 0; s < 48 / KC + (48 % KC ? 1 : 0); ++s) {
        unsigned raw[4] = {0u, 0u, 0u, 0u};
        ACH_UNROLL
        for (int h = 0; h < SEG; ++h) {
            const int kk = s * KC + g * VEC + 4 * h;                 // first k of this 4-pixel patch row
            if (valid && kk < 48) {
                const IO* row = X + (kk >> 4) * cstride + ((kk >> 2) & 3) * long(p.Wd);
                if (sizeof(T) == 2) { const uint2 v = *reinterpret_cast<const uint2*>(row); raw[2 * h] = h16_recast<IO, T>(v.x); raw[2 * h + 1] = h16_recast<IO, T>(v.y); }
                else { const uint4 v = *reinterpret_cast<const uint4*>(row); raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w; }
            }
        }
        const uint4 xf = make_uint4(raw[0], raw[1], raw[2], raw[3]);
        mfma16<T>(Wf[(s * 2) * 64], xf, acc[0]);
        mfma16<T>(Wf[(s * 2 + 1) * 64], xf, acc[1]);
    }
    // lane: channels g*8 .. g*8+7 of pixel px (chunk_channel with NT = 2)
    float v[8];
    ACH_UNROLL
    for (int r = 0; r < 4; ++r) { v[r] = acc[0][r] + p.bias[g * 8 + r]; v[4 + r] = acc[1][r] + p.bias[g * 8 + 4 + r]; }
    float mu = 0.f;
    ACH_UNROLL
    for (int i = 0; i < 8; ++i) mu += v[i];
    mu += __shfl_xor(mu, 16); mu += __shfl_xor(mu, 32);
    mu *= (1.0f / 32.0f);
    float var = 0.f;
    ACH_UNROLL
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mu; var += d * d; }
    var += __shfl_xor(var, 16); var += __shfl_xor(var, 32);
    const float rs = 1.0f / sqrtf(var * (1.0f / 32.0f) + p.eps);
    if (!valid) return;
    float o[8];
    ACH_UNROLL
    for (int i = 0; i < 8; ++i) o[i] = (v[i] - mu) * rs * p.lnw[g * 8 + i] + p.lnb[g * 8 + i];
    Store<T>::st8(static_cast<T*>(p.Y) + m * 32 + g * 8, o);
}

// ------------------------------------------------------------------------------------------ MobileViT stem (round 4)
// conv1 = 3x3 / stride 2 / pad 1, 3 -> 16 channels, + BatchNorm + SiLU (mobilevit.py:6-19, 198-203) gathered straight from the NCHW image, like the EdgeNeXt stem
// above: the NHWC copy of the image (3 channels in an 8-channel pitch: 105 MB written and read back at batch 64) and its launch disappear — 301 MB of traffic
// for the pair become 39 MB in + 52 MB out.  The reduction index is k = (c*3 + ky)*4 + kx with kx = 3 a zero weight (K = 36 -> two k-steps of 32): a lane's 8
// consecutive k values are TWO image rows of [x-1, x, x+1, -], each one aligned dword (x, x+1: x = 2 ox is even) + one 2-byte load (x-1).
struct MvStemParams { const void* X; void* Y; const void* W; const float* bias; int B, H, Wd; };
constexpr int MVSTEM_TPW = 4;            // 16-pixel tiles per wave: every load of the four tiles is in flight before the first MFMA (one tile per wave: 51.8 us at batch 64)
template <class T, class IO>
__global__ __launch_bounds__(256) void mvstem_kernel(const MvStemParams p) { f16_sat_mode<T>();
    static_assert(Store<T>::VEC == 8, "16-bit storage");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const int Ho = p.H / 2, Wo = p.Wd / 2;
    const long total = long(p.B) * Ho * Wo;
    const long cstride = long(p.H) * p.Wd;
    const uint4* Wf = static_cast<const uint4*>(p.W) + lane;
    const uint4 w0 = Wf[0], w1 = Wf[64];
    float bias[4];
    ACH_UNROLL
    for (int r = 0; r < 4; ++r) bias[r] = p.bias[g * 4 + r];
    // phase 1: every address and every load of the wave's four tiles (32-bit index arithmetic: B * Ho * Wo < 2^31); phase 2 converts.  All loads are
    // unconditional, from clamped addresses — padding and idle k-slots are zeroed afterwards (a load under a lane condition is a branch, and behind a branch the
    // compiler waits for each load by itself)
    uint32_t mid[MVSTEM_TPW][2][2], lraw[MVSTEM_TPW][2][2];
    bool use[MVSTEM_TPW][2][2], hasl[MVSTEM_TPW];
    unsigned mm[MVSTEM_TPW];
    bool ok[MVSTEM_TPW];
    const unsigned utotal = unsigned(total), uWo = unsigned(Wo), uHo = unsigned(Ho);
    ACH_UNROLL
    for (int t = 0; t < MVSTEM_TPW; ++t) {
        const unsigned mraw = ((blockIdx.x * 4u + unsigned(wave)) * MVSTEM_TPW + unsigned(t)) * 16u + unsigned(px);
        ok[t] = mraw < utotal;
        const unsigned m = ok[t] ? mraw : 0u;
        mm[t] = m;
        const unsigned q = m / uWo;
        const int ox = int(m - q * uWo);
        const unsigned b = q / uHo;
        const int oy = int(q - b * uHo);
        hasl[t] = ox > 0;
        const IO* X = static_cast<const IO*>(p.X) + long(b) * 3 * cstride + 2 * ox;
        ACH_UNROLL
        for (int s = 0; s < 2; ++s) {
            ACH_UNROLL
            for (int h = 0; h < 2; ++h) {
                const int r = (s * 4 + g) * 2 + h;                       // image row (c, ky) of this half of the fragment (r >= 9: zero weights, nothing to fetch)
                const int rc = r < 9 ? r : 8;
                const int c = rc / 3, ky = rc - 3 * c;
                const int y = 2 * oy - 1 + ky;                           // <= H - 1 (H even); -1 = the conv's zero padding above the image
                const IO* row = X + c * cstride + long(y < 0 ? 0 : y) * p.Wd;
                mid[t][s][h] = *reinterpret_cast<const uint32_t*>(row);                                            // x, x + 1
                lraw[t][s][h] = uint32_t(reinterpret_cast<const uint16_t*>(row)[hasl[t] ? -1 : 0]);                // x - 1
                use[t][s][h] = ok[t] && r < 9 && y >= 0;
            }
        }
    }
    unsigned raw[MVSTEM_TPW][2][4];
    ACH_UNROLL
    for (int t = 0; t < MVSTEM_TPW; ++t) {
        ACH_UNROLL
        for (int s = 0; s < 2; ++s) {
            ACH_UNROLL
            for (int h = 0; h < 2; ++h) {
                const uint32_t left = hasl[t] ? lraw[t][s][h] : 0u;                                                // the conv's zero padding at the left edge
                raw[t][s][2 * h] = use[t][s][h] ? h16_recast<IO, T>(left | (mid[t][s][h] << 16)) : 0u;
                raw[t][s][2 * h + 1] = use[t][s][h] ? h16_recast<IO, T>(mid[t][s][h] >> 16) : 0u;
            }
        }
    }
    ACH_UNROLL
    for (int t = 0; t < MVSTEM_TPW; ++t) {
        f32x4 acc;
        acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
        mfma16<T>(w0, make_uint4(raw[t][0][0], raw[t][0][1], raw[t][0][2], raw[t][0][3]), acc);
        mfma16<T>(w1, make_uint4(raw[t][1][0], raw[t][1][1], raw[t][1][2], raw[t][1][3]), acc);
        if (!ok[t]) continue;
        float o[4];                                                      // NT = 1: channels 4g .. 4g+3 of pixel px
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) o[r] = apply_act(acc[r] + bias[r], ACT_SILU);
        Store<T>::st4(static_cast<T*>(p.Y) + long(mm[t]) * 16 + g * 4, o);
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm over C
// channels-first LayerNorm in front of the three 2x2/s2 down-sampling convs.  A row (pixel) is owned by a group of G lanes
// (G = power of two >= C/4, <= 64), each lane holding 4 channels per step; reductions are xor-shuffles inside the group.
struct LnParams { const void* X; long ldx; void* Y; long ldy; const float* w; const float* b; long rows; int C; float eps; int G; };
template <class T>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnParams p) { f16_sat_mode<T>();
    const int G = p.G, rows_per_block = 256 / G;
    const int gl = threadIdx.x % G;
    const long row = long(blockIdx.x) * rows_per_block + threadIdx.x / G;
    const bool ok = row < p.rows;
    const T* x = static_cast<const T*>(p.X) + (ok ? row : 0) * p.ldx;
    const int cq = p.C >> 2;
    float s1 = 0.f, s2 = 0.f;
    for (int q = gl; q < cq; q += G) {
        float v[4]; Store<T>::ld4(x + q * 4, v);
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
    }
    for (int m = G >> 1; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
    const float mu = s1 / float(p.C);
    float var = s2 / float(p.C) - mu * mu;
    var = var > 0.f ? var : 0.f;
    const float rs = 1.0f / sqrtf(var + p.eps);
    if (!ok) return;
    T* y = static_cast<T*>(p.Y) + row * p.ldy;
    for (int q = gl; q < cq; q += G) {
        float v[4]; Store<T>::ld4(x + q * 4, v);
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) v[i] = (v[i] - mu) * rs * p.w[q * 4 + i] + p.b[q * 4 + i];
        Store<T>::st4(y + q * 4, v);
    }
}

// ------------------------------------------------------------------------------------------ bilinear x2
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True): src = dst * (in-1)/(out-1)
struct UpParams { const void* X; long ldx; void* Y; long ldy; int B, H, Wd, C; };
template <class T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const UpParams p) { f16_sat_mode<T>();
    const int Ho = p.H * 2, Wo = p.Wd * 2, cq = p.C >> 2;
    const long total = long(p.B) * Ho * Wo * cq;
    const long idx = long(xcd_block(blockIdx.x, gridDim.x)) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = int(idx % cq) * 4;
    long pix = idx / cq;
    const int ox = int(pix % Wo); pix /= Wo;
    const int oy = int(pix % Ho);
    const long b = pix / Ho;
    const float sy = Ho > 1 ? float(p.H - 1) / float(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? float(p.Wd - 1) / float(Wo - 1) : 0.f;
    const float fy = sy * float(oy), fx = sx * float(ox);
    int y0 = int(fy), x0 = int(fx);
    if (y0 > p.H - 1) y0 = p.H - 1;
    if (x0 > p.Wd - 1) x0 = p.Wd - 1;
    const int y1 = y0 + (y0 < p.H - 1 ? 1 : 0), x1 = x0 + (x0 < p.Wd - 1 ? 1 : 0);
    const float ly = fy - float(y0), lx = fx - float(x0);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* X = static_cast<const T*>(p.X);
    float a[4], bq[4], cc[4], d[4], o[4];
    Store<T>::ld4(X + ((b * p.H + y0) * p.Wd + x0) * p.ldx + c, a);
    Store<T>::ld4(X + ((b * p.H + y0) * p.Wd + x1) * p.ldx + c, bq);
    Store<T>::ld4(X + ((b * p.H + y1) * p.Wd + x0) * p.ldx + c, cc);
    Store<T>::ld4(X + ((b * p.H + y1) * p.Wd + x1) * p.ldx + c, d);
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) o[i] = hy * (hx * a[i] + lx * bq[i]) + ly * (hx * cc[i] + lx * d[i]);
    Store<T>::st4(static_cast<T*>(p.Y) + ((b * Ho + oy) * Wo + ox) * p.ldy + c, o);
}

// ------------------------------------------------------------------------------------------ SPP max pools
// reads channels [0,C) of the concat buffer, writes the 5x5 / 9x9 / 13x13 stride-1 max pools (-inf padding)
// to channels [C,2C) [2C,3C) [3C,4C).  (SPPF's three chained 5x5 pools are the same three windows.)
struct SppParams { void* buf; long ld; int B, H, Wd, C, cqb; };
// One workgroup = one sample x `cqb` channel quads: the map slice is staged in LDS once and the three square windows are
// computed separably (row maxima of radius 2 / 4 / 6, then column maxima) — 2 x 13 LDS reads per output instead of 169
// global loads.  H * W * cqb <= SPP_TILE.
constexpr int SPP_TILE = 768;
template <class T>
__global__ __launch_bounds__(256) void spp_pool_kernel(const SppParams p) { f16_sat_mode<T>();
    __shared__ float4 src[SPP_TILE], r5[SPP_TILE], r9[SPP_TILE], r13[SPP_TILE];
    const int cq = p.C >> 2, groups = (cq + p.cqb - 1) / p.cqb;
    const int b = blockIdx.x / groups, q0 = (blockIdx.x % groups) * p.cqb;
    const int nq = (cq - q0) < p.cqb ? (cq - q0) : p.cqb;
    const int HW = p.H * p.Wd, n = HW * nq;
    T* buf = static_cast<T*>(p.buf) + long(b) * HW * p.ld;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int pix = i / nq, q = i - pix * nq;
        float v[4];
        Store<T>::ld4(buf + long(pix) * p.ld + (q0 + q) * 4, v);
        src[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    auto mx = [](float4 a, const float4& c) { a.x = fmaxf(a.x, c.x); a.y = fmaxf(a.y, c.y); a.z = fmaxf(a.z, c.z); a.w = fmaxf(a.w, c.w); return a; };
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int pix = i / nq, q = i - pix * nq, y = pix / p.Wd, x = pix - y * p.Wd;
        float4 m5 = src[i], m9, m13;
        for (int d = 1; d <= 2; ++d) { if (x - d >= 0) m5 = mx(m5, src[(pix - d) * nq + q]); if (x + d < p.Wd) m5 = mx(m5, src[(pix + d) * nq + q]); }
        m9 = m5;
        for (int d = 3; d <= 4; ++d) { if (x - d >= 0) m9 = mx(m9, src[(pix - d) * nq + q]); if (x + d < p.Wd) m9 = mx(m9, src[(pix + d) * nq + q]); }
        m13 = m9;
        for (int d = 5; d <= 6; ++d) { if (x - d >= 0) m13 = mx(m13, src[(pix - d) * nq + q]); if (x + d < p.Wd) m13 = mx(m13, src[(pix + d) * nq + q]); }
        r5[i] = m5; r9[i] = m9; r13[i] = m13;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int pix = i / nq, q = i - pix * nq, y = pix / p.Wd;
        float4 m5 = r5[i], m9 = r9[i], m13 = r13[i];
        for (int d = 1; d <= 6; ++d) {
            if (y - d >= 0) { const int j = (pix - d * p.Wd) * nq + q; if (d <= 2) m5 = mx(m5, r5[j]); if (d <= 4) m9 = mx(m9, r9[j]); m13 = mx(m13, r13[j]); }
            if (y + d < p.H) { const int j = (pix + d * p.Wd) * nq + q; if (d <= 2) m5 = mx(m5, r5[j]); if (d <= 4) m9 = mx(m9, r9[j]); m13 = mx(m13, r13[j]); }
        }
        T* o = buf + long(pix) * p.ld + (q0 + q) * 4;
        const float a5[4] = {m5.x, m5.y, m5.z, m5.w}, a9[4] = {m9.x, m9.y, m9.z, m9.w}, a13[4] = {m13.x, m13.y, m13.z, m13.w};
        Store<T>::st4(o + p.C, a5);
        Store<T>::st4(o + 2 * p.C, a9);
        Store<T>::st4(o + 3 * p.C, a13);
    }
}

// ------------------------------------------------------------------------------------------ element-wise
struct AddParams { const void* A; long lda; const void* Bp; long ldb; void* Y; long ldy; long rows; int C; };
template <class T>
__global__ __launch_bounds__(256) void add_kernel(const AddParams p) { f16_sat_mode<T>();
    const int cq = p.C >> 2;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.rows * cq) return;
    const int c = int(idx % cq) * 4;
    const long r = idx / cq;
    float a[4], b[4];
    Store<T>::ld4(static_cast<const T*>(p.A) + r * p.lda + c, a);
    Store<T>::ld4(static_cast<const T*>(p.Bp) + r * p.ldb + c, b);
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) a[i] += b[i];
    Store<T>::st4(static_cast<T*>(p.Y) + r * p.ldy + c, a);
}
// up to three independent adds in one launch (the neck's three residual outputs q3 / q4 / q5: blockIdx.y = job)
struct AddJobs { AddParams j[3]; int n; };
template <class T>
__global__ __launch_bounds__(256) void add_multi_kernel(const AddJobs m) { f16_sat_mode<T>();
    const AddParams& p = m.j[blockIdx.y];
    const int cq = p.C >> 2;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.rows * cq) return;
    const int c = int(idx % cq) * 4;
    const long r = idx / cq;
    float a[4], b[4];
    Store<T>::ld4(static_cast<const T*>(p.A) + r * p.lda + c, a);
    Store<T>::ld4(static_cast<const T*>(p.Bp) + r * p.ldb + c, b);
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) a[i] += b[i];
    Store<T>::st4(static_cast<T*>(p.Y) + r * p.ldy + c, a);
}
// copy a [rows, C] view (optionally adding a per-position constant [HW][C] fp32: the folded Fourier pos-enc)
struct CopyParams { const void* X; long ldx; void* Y; long ldy; long rows; int C; const float* posenc; int HW; };
template <class T>
__global__ __launch_bounds__(256) void copy_kernel(const CopyParams p) { f16_sat_mode<T>();
    const int cq = p.C >> 2;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.rows * cq) return;
    const int c = int(idx % cq) * 4;
    const long r = idx / cq;
    float a[4];
    Store<T>::ld4(static_cast<const T*>(p.X) + r * p.ldx + c, a);
    if (p.posenc) {
        const float* pe = p.posenc + (r % p.HW) * p.C + c;
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) a[i] += pe[i];
    }
    Store<T>::st4(static_cast<T*>(p.Y) + r * p.ldy + c, a);
}

// ------------------------------------------------------------------------------------------ channel statistics
// per (sample, channel) sum and sum of squares over the H*W positions of an NHWC view.
// grid (B, S): block (b, s) reduces positions s, s+S, ... and writes partial[b][s][2][C]; the consumers add the S partials.
struct StatParams { const void* X; long ldx; float* partial; int HW, C, S; };
template <class T>
__device__ __forceinline__ void chan_stats_body(const StatParams& p, int b, int s) {
    __shared__ float4 red[2][256];
    const int tid = threadIdx.x;
    const T* X = static_cast<const T*>(p.X) + long(b) * p.HW * p.ldx;
    float* out = p.partial + (long(b) * p.S + s) * 2 * p.C;
    // threads = (channel QUAD q) x (row lane rl): one 8 / 16-byte load per position and thread (round 3: was one element per thread —
    // 2-byte loads, 4x the iterations); every thread strides over positions.  A last quad past C reads the row's zero padding (ld is a
    // multiple of 8 elements) and is not written.
    const int cq = (p.C + 3) >> 2;
    const int tc = cq < 256 ? cq : 256;         // quads handled per pass
    const int rl_n = 256 / tc;                  // row lanes per quad (>= 1)
    for (int q0 = 0; q0 < cq; q0 += tc) {
        const int q = q0 + (tid % tc);
        const int rl = tid / tc;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < rl_n && q < cq)
            for (int i = s * rl_n + rl; i < p.HW; i += p.S * rl_n) {
                float v[4];
                Store<T>::ld4(X + long(i) * p.ldx + 4 * q, v);
                ACH_UNROLL
                for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
            }
        red[0][tid] = make_float4(s1[0], s1[1], s1[2], s1[3]); red[1][tid] = make_float4(s2[0], s2[1], s2[2], s2[3]);
        __syncthreads();
        if (tid < tc && q0 + tid < cq) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), w = a;
            for (int r = 0; r < rl_n; ++r) {
                const float4 u = red[0][r * tc + tid], z = red[1][r * tc + tid];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; w.x += z.x; w.y += z.y; w.z += z.z; w.w += z.w;
            }
            const int c = 4 * (q0 + tid);
            const float av[4] = {a.x, a.y, a.z, a.w}, wv[4] = {w.x, w.y, w.z, w.w};
            ACH_UNROLL
            for (int e = 0; e < 4; ++e) if (c + e < p.C) { out[c + e] = av[e]; out[p.C + c + e] = wv[e]; }
        }
        __syncthreads();
    }
}
template <class T>
__global__ __launch_bounds__(256) void chan_stats_kernel(const StatParams p) { f16_sat_mode<T>(); chan_stats_body<T>(p, blockIdx.x, blockIdx.y); }
// up to 6 independent jobs in one launch (the six ECA inputs of the fusion stage): blockIdx.z = job
template <class P> struct Multi6 { P j[6]; int n; };
template <class T>
__global__ __launch_bounds__(256) void chan_stats_multi_kernel(const Multi6<StatParams> m) { f16_sat_mode<T>();
    const StatParams& p = m.j[blockIdx.z];
    if (int(blockIdx.y) >= p.S) return;
    chan_stats_body<T>(p, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------ ShuffleAttention
// coefficients: out = x * sigmoid(a*x + d) per (sample, channel)
//   channel-attention half: a = 0, d = cweight*mean + cbias
//   spatial half (GroupNorm with one channel per group): a = sweight*gnw*rstd, d = sweight*(gnb - gnw*mean*rstd) + sbias
// Both decoders start with a ShuffleAttention of the SAME tensor (ghostdualfpn.py:175,187): the channel statistics are
// computed once and the two modules (weight sets w[0], w[1]) share one coefficient launch and one apply launch that reads the
// input once and writes both outputs.
struct SaWeights { const float* cw; const float* cb; const float* sw; const float* sb; const float* gnw; const float* gnb; };
struct SaCoefParams {
    const float* partial; int S; float* coef;   // coef [2][B][C][2]
    SaWeights w[2];
    int B, C, G, HW; float eps;
};
static __global__ void sa_coef_kernel(const SaCoefParams p) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (gidx >= 2 * p.B * p.C) return;
    const int m = gidx / (p.B * p.C), idx = gidx - m * p.B * p.C;
    const SaWeights& w = p.w[m];
    const int b = idx / p.C, c = idx % p.C;
    float s1 = 0.f, s2 = 0.f;
    for (int s = 0; s < p.S; ++s) { const float* q = p.partial + (long(b) * p.S + s) * 2 * p.C; s1 += q[c]; s2 += q[p.C + c]; }
    const float mean = s1 / float(p.HW);
    const int cg = p.C / p.G, half = cg / 2;        // channels per group, per half
    const int j = c % cg;
    float a, d;
    if (j < half) { a = 0.f; d = w.cw[j] * mean + w.cb[j]; }
    else {
        const int jj = j - half;
        float var = s2 / float(p.HW) - mean * mean;
        if (var < 0.f) var = 0.f;
        const float rstd = 1.0f / sqrtf(var + p.eps);
        a = w.sw[jj] * w.gnw[jj] * rstd;
        d = w.sw[jj] * (w.gnb[jj] - w.gnw[jj] * mean * rstd) + w.sb[jj];
    }
    p.coef[2 * gidx] = a;
    p.coef[2 * gidx + 1] = d;
}
// apply + channel_shuffle(groups=2): input channel c -> output channel (c % (C/2)) * 2 + c / (C/2)
struct SaApplyParams { const void* X; long ldx; void* Y0; void* Y1; long ldy; const float* coef; int B, HW, C; };
template <class T>
__global__ __launch_bounds__(256) void sa_apply_kernel(const SaApplyParams p) { f16_sat_mode<T>();
    const long total = long(p.B) * p.HW * p.C;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = int(idx % p.C);
    const long pix = idx / p.C;
    const long b = pix / p.HW;
    const int c = (co & 1) * (p.C / 2) + (co >> 1);          // inverse of the shuffle
    const float x = Store<T>::ld(static_cast<const T*>(p.X) + pix * p.ldx + c);
    const float* k0 = p.coef + (b * p.C + c) * 2;
    const float* k1 = k0 + long(p.B) * p.C * 2;
    Store<T>::st(static_cast<T*>(p.Y0) + pix * p.ldy + co, x * sigmoidf_(k0[0] * x + k0[1]));
    Store<T>::st(static_cast<T*>(p.Y1) + pix * p.ldy + co, x * sigmoidf_(k1[0] * x + k1[1]));
}

// Eight output channels per thread (round 4): outputs co .. co + 7 are inputs c0 .. c0 + 3 of the first half interleaved with the same four of
// the second half — two 8-byte loads, two 16-byte stores instead of eight 2-byte loads and sixteen 2-byte stores (19 -> 8 us at 40 x 40).  C % 8 == 0.
template <class T>
__global__ __launch_bounds__(256) void sa_apply8_kernel(const SaApplyParams p) { f16_sat_mode<T>();
    const int c8n = p.C / 8;
    const long total = long(p.B) * p.HW * c8n;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int q = int(idx % c8n);
    const long pix = idx / c8n;
    const long b = pix / p.HW;
    const int co = q * 8, c0 = co >> 1, half = p.C / 2;
    float xa[4], xb[4];
    Store<T>::ld4(static_cast<const T*>(p.X) + pix * p.ldx + c0, xa);
    Store<T>::ld4(static_cast<const T*>(p.X) + pix * p.ldx + half + c0, xb);
    const float* k0 = p.coef + (b * p.C) * 2;
    const float* k1 = k0 + long(p.B) * p.C * 2;
    float o0[8], o1[8];
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) {
        const int ca = c0 + i, cb = half + c0 + i;
        o0[2 * i] = xa[i] * sigmoidf_(k0[2 * ca] * xa[i] + k0[2 * ca + 1]);
        o0[2 * i + 1] = xb[i] * sigmoidf_(k0[2 * cb] * xb[i] + k0[2 * cb + 1]);
        o1[2 * i] = xa[i] * sigmoidf_(k1[2 * ca] * xa[i] + k1[2 * ca + 1]);
        o1[2 * i + 1] = xb[i] * sigmoidf_(k1[2 * cb] * xb[i] + k1[2 * cb + 1]);
    }
    Store<T>::st8(static_cast<T*>(p.Y0) + pix * p.ldy + co, o0);
    Store<T>::st8(static_cast<T*>(p.Y1) + pix * p.ldy + co, o1);
}

// The same with the coefficients computed by the workgroup itself (round 4: one launch fewer on the caller's stream; measured SLOWER end to end —
// 38.3 k against 38.9 k frames/s: 19 200 workgroups each re-derive 96 coefficients behind a barrier — option sa_fuse, off): a block of 256 consecutive
// elements lies inside ONE sample when HW * C is a multiple of 256 (the engine checks), so its first 2 C threads evaluate sa_coef_kernel's formulas
// for that sample into LDS — same operations in the same order, bit-identical results.
struct SaFusedParams { SaCoefParams c; SaApplyParams a; };
template <class T>
__global__ __launch_bounds__(256) void sa_apply_fused_kernel(const SaFusedParams q) { f16_sat_mode<T>();
    __shared__ float coef[2][256][2];
    const SaCoefParams& p = q.c;
    const SaApplyParams& ap = q.a;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    const long b = (long(blockIdx.x) * blockDim.x) / (long(ap.HW) * ap.C);
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
        const int m = i / p.C, c = i - m * p.C;
        const SaWeights& w = p.w[m];
        float s1 = 0.f, s2 = 0.f;
        for (int s = 0; s < p.S; ++s) { const float* qq = p.partial + (b * p.S + s) * 2 * p.C; s1 += qq[c]; s2 += qq[p.C + c]; }
        const float mean = s1 / float(p.HW);
        const int cg = p.C / p.G, half = cg / 2;
        const int j = c % cg;
        float a, d;
        if (j < half) { a = 0.f; d = w.cw[j] * mean + w.cb[j]; }
        else {
            const int jj = j - half;
            float var = s2 / float(p.HW) - mean * mean;
            if (var < 0.f) var = 0.f;
            const float rstd = 1.0f / sqrtf(var + p.eps);
            a = w.sw[jj] * w.gnw[jj] * rstd;
            d = w.sw[jj] * (w.gnb[jj] - w.gnw[jj] * mean * rstd) + w.sb[jj];
        }
        coef[m][c][0] = a; coef[m][c][1] = d;
    }
    __syncthreads();
    if (idx >= long(ap.B) * ap.HW * ap.C) return;
    const int co = int(idx % ap.C);
    const long pix = idx / ap.C;
    const int c = (co & 1) * (ap.C / 2) + (co >> 1);          // inverse of the shuffle
    const float x = Store<T>::ld(static_cast<const T*>(ap.X) + pix * ap.ldx + c);
    Store<T>::st(static_cast<T*>(ap.Y0) + pix * ap.ldy + co, x * sigmoidf_(coef[0][c][0] * x + coef[0][c][1]));
    Store<T>::st(static_cast<T*>(ap.Y1) + pix * ap.ldy + co, x * sigmoidf_(coef[1][c][0] * x + coef[1][c][1]));
}

// ------------------------------------------------------------------------------------------ ECA + fusion
// scale[b][c] = sigmoid(conv1d_k(mean over HW)) * bn_scale[c] ; shift = bn_shift[c]
struct EcaParams { const float* partial; int S; const float* w; int k; const float* bn_scale; float* scale; int B, C, HW; };
__device__ __forceinline__ void eca_scale_body(const EcaParams& p, int idx) {
    if (idx >= p.B * p.C) return;
    const int b = idx / p.C, c = idx % p.C;
    float g = 0.f;
    for (int t = 0; t < p.k; ++t) {
        const int cc = c + t - (p.k - 1) / 2;
        if (cc < 0 || cc >= p.C) continue;
        float s1 = 0.f;
        for (int s = 0; s < p.S; ++s) s1 += p.partial[(long(b) * p.S + s) * 2 * p.C + cc];
        g += p.w[t] * (s1 / float(p.HW));
    }
    p.scale[idx] = sigmoidf_(g) * p.bn_scale[c];
}
static __global__ void eca_scale_multi_kernel(const Multi6<EcaParams> m) { eca_scale_body(m.j[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x); }
// Y[b,pix,c] = relu(X[b,pix,c] * scale[b][c] + shift[c]); X is NHWC (x_nchw = 0) or NCHW (radar branch)
struct FuseParams { const void* X; long ldx; int x_nchw; void* Y; long ldy; const float* scale; const float* shift; int B, HW, C; };
// one thread = 4 consecutive channels of one pixel (C % 4 == 0 for NHWC inputs; the NCHW input form stays scalar)
template <class T>
__device__ __forceinline__ void fuse_scale_body(const FuseParams& p, long idx) {
    const T* X = static_cast<const T*>(p.X);
    if (p.x_nchw || (p.C & 3)) {
        const long total = long(p.B) * p.HW * p.C;
        for (int e = 0; e < 4; ++e) {
            const long i = idx * 4 + e;
            if (i >= total) return;
            const int c = int(i % p.C);
            const long pix = i / p.C;
            const long b = pix / p.HW;
            const float x = p.x_nchw ? Store<T>::ld(X + (b * p.C + c) * p.HW + (pix - b * p.HW)) : Store<T>::ld(X + pix * p.ldx + c);
            const float v = x * p.scale[b * p.C + c] + p.shift[c];
            Store<T>::st(static_cast<T*>(p.Y) + pix * p.ldy + c, v > 0.f ? v : 0.f);
        }
        return;
    }
    const int cq = p.C >> 2;
    const long total = long(p.B) * p.HW * cq;
    if (idx >= total) return;
    const int c = int(idx % cq) * 4;
    const long pix = idx / cq;
    const long b = pix / p.HW;
    float x[4], o[4];
    Store<T>::ld4(X + pix * p.ldx + c, x);
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + b * p.C + c), sh = *reinterpret_cast<const float4*>(p.shift + c);
    o[0] = fmaxf(x[0] * sc.x + sh.x, 0.f); o[1] = fmaxf(x[1] * sc.y + sh.y, 0.f); o[2] = fmaxf(x[2] * sc.z + sh.z, 0.f); o[3] = fmaxf(x[3] * sc.w + sh.w, 0.f);
    Store<T>::st4(static_cast<T*>(p.Y) + pix * p.ldy + c, o);
}
template <class T>
__global__ __launch_bounds__(256) void fuse_scale_multi_kernel(const Multi6<FuseParams> m) { f16_sat_mode<T>(); fuse_scale_body<T>(m.j[blockIdx.y], long(blockIdx.x) * blockDim.x + threadIdx.x); }

}  // namespace ach

namespace ach {

// ------------------------------------------------------------------------------------------ decoder level (fused)
// One segmentation-decoder level (neck/ghostdualfpn.py:175-197) is
//     u = relu(bn(conv1x1(x)))  ->  bilinear x2 (align_corners)  ->  x1 = relu(bn(conv1x1(.)))  ->  x2 = relu(bn(dw3x3(x1)))
// Bilinear interpolation is linear and its weights sum to 1, so it commutes with the Ghost primary 1x1 conv + folded BN:
//     x1 = relu( bilinear( Wp u + bp ) ).
// Both 1x1 convs therefore run at LOW resolution (4x fewer pixels, MFMA GEMMs) and the full-resolution work is this
// kernel: x1 = relu(bilinear(t)), x2 = relu(dw3x3(x1) + b), written side by side as [x1 | x2] — the full-resolution
// tensor is written exactly once and never re-read by the level itself.  LDS-staged halo tile (16x16 outputs).
struct UpGhostParams {
    const void* Tq; long ldt;       // t = Wp u + bp at low resolution, NHWC [B,h,w,Cg]
    void* Y; long ldy;              // [B,2h,2w,2Cg]
    const float* Wdw; const float* bdw;   // cheap operation: [9][Cg] (BN folded), [Cg]
    int B, h, w, Cg;
};
constexpr int UPG_TS = 16;
constexpr int UPG_CMAX = 32;
// CG = Ghost half-width (16 / 24 / 32).  64 * CG/4 threads: a thread owns one 4-channel group for the whole tile (its nine
// depthwise weight vectors live in registers) and walks the tile's pixels 64 at a time.
template <class T, int CG>
__global__ __launch_bounds__(16 * CG) void upghost_kernel(const UpGhostParams p) { f16_sat_mode<T>();
    constexpr int TS = UPG_TS, HS = TS + 2, CQ = CG / 4;
    __shared__ float x1[HS * HS * CG];
    const int H = 2 * p.h, Wd = 2 * p.w;
    const int tiles_x = (Wd + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;      // XCD-aware tile order (see upghost_head_kernel)
    const unsigned nwg = gridDim.x;
    const unsigned wg = xcd_block(blockIdx.x, nwg);
    const int bx = int(wg % tiles_x) * TS, by = int((wg / tiles_x) % tiles_y) * TS;
    const long b = wg / (unsigned(tiles_x) * tiles_y);
    const int c = (threadIdx.x % CQ) * 4, slot = threadIdx.x / CQ;
    const float sy = H > 1 ? float(p.h - 1) / float(H - 1) : 0.f, sx = Wd > 1 ? float(p.w - 1) / float(Wd - 1) : 0.f;
    const T* Tq = static_cast<const T*>(p.Tq) + b * p.h * long(p.w) * p.ldt + c;
    constexpr int ROUNDS = (HS * HS + 63) / 64;
    ACH_UNROLL
    for (int r = 0; r < ROUNDS; ++r) {                             // unconditional clamped loads: all rounds in flight together
        const int pos_raw = slot + r * 64;
        const int pos = pos_raw < HS * HS ? pos_raw : HS * HS - 1;
        const int oy = by + pos / HS - 1, ox = bx + pos % HS - 1;
        const bool ok = oy >= 0 && oy < H && ox >= 0 && ox < Wd;  // outside the map: the dw conv's zero padding
        const int cy = oy < 0 ? 0 : (oy >= H ? H - 1 : oy), cx = ox < 0 ? 0 : (ox >= Wd ? Wd - 1 : ox);
        const float fy = sy * float(cy), fx = sx * float(cx);
        int y0 = int(fy), x0 = int(fx);
        if (y0 > p.h - 1) y0 = p.h - 1;
        if (x0 > p.w - 1) x0 = p.w - 1;
        const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1i = x0 + (x0 < p.w - 1 ? 1 : 0);
        const float ly = fy - float(y0), lx = fx - float(x0), hy = 1.f - ly, hx = 1.f - lx;
        float a[4], bq[4], cc[4], d[4], v[4];
        Store<T>::ld4(Tq + (long(y0) * p.w + x0) * p.ldt, a);
        Store<T>::ld4(Tq + (long(y0) * p.w + x1i) * p.ldt, bq);
        Store<T>::ld4(Tq + (long(y1) * p.w + x0) * p.ldt, cc);
        Store<T>::ld4(Tq + (long(y1) * p.w + x1i) * p.ldt, d);
        ACH_UNROLL
        for (int i = 0; i < 4; ++i) { const float t = hy * (hx * a[i] + lx * bq[i]) + ly * (hx * cc[i] + lx * d[i]); v[i] = (ok && t > 0.f) ? t : 0.f; }
        if (pos_raw < HS * HS) *reinterpret_cast<float4*>(x1 + pos * CG + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
    float wk[9][4];
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) { const float4 w = *reinterpret_cast<const float4*>(p.Wdw + k * CG + c); wk[k][0] = w.x; wk[k][1] = w.y; wk[k][2] = w.z; wk[k][3] = w.w; }
    const float4 bb = *reinterpret_cast<const float4*>(p.bdw + c);
    __syncthreads();
    T* Y = static_cast<T*>(p.Y);
    ACH_UNROLL
    for (int pix = slot; pix < TS * TS; pix += 64) {
        const int ty = pix / TS, tx = pix % TS;
        const int oy = by + ty, ox = bx + tx;
        if (oy >= H || ox >= Wd) continue;
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
        T* yo = Y + ((b * H + oy) * long(Wd) + ox) * p.ldy + c;
        Store<T>::st4(yo, o1);
        Store<T>::st4(yo + CG, acc);
    }
}

}  // namespace ach

namespace ach {

// ------------------------------------------------------------------------------------------ last decoder level + head (fused)
// upghost (above) followed by the Ghost segmentation head, in one kernel: the 32-channel full-resolution tensor — the
// single largest activation of the network (6.5 MB per frame per decoder in bf16) — is never written to HBM.
//   x1  = relu(bilinear(t))                 on the tile + 2-pixel halo   (LDS)
//   f   = [x1 | relu(dw3x3(x1) + b)]        on the tile + 1-pixel halo   (registers)
//   h1  = relu(Wh f + bh)   (init channels) on the tile + 1-pixel halo   (LDS)
//   out = [h1 | relu(dw3x3(h1) + b')][:oup] on the tile, scattered to NCHW
// Tile 32x8 outputs, 256 threads.  `F` (optional) receives f for the parity tests that tap this boundary.
struct UpGhostHeadParams {
    const void* Tq; long ldt;                 // t at low resolution [B,h,w,16]
    void* F; long ldf;                        // optional [B,2h,2w,32] tap (nullptr in production plans)
    void* out;                                // NCHW [B,oup,2h,2w]
    const float* Wdw; const float* bdw;       // level cheap op: [9][16], [16]
    const float* Wh; const float* bh;         // head primary: [init][32], [init]
    const float* Wdh; const float* bdh;       // head cheap op: [9][nch], [nch]
    int B, h, w, init, nch, oup;
    float sy, sx;                             // (h-1)/(2h-1), (w-1)/(2w-1): align_corners source scale (computed on the host, same float division)
    int out_bf16 = 0;                         // fp16-storage engine: the caller's output tensor is bf16 (st_user)
};
constexpr int UGH_TW = 30, UGH_TH = 6, UGH_CG = 16, UGH_IMAX = 8, UGH_THREADS = 256;

// Phases 2 and 3 of the fused last level, shared by both variants below: thread = position of the 1-halo tile (W1 x H1 <= 256).
//   x1s: relu(bilinear(t)) on the 2-halo tile, [position][CS] floats (zero outside the map)
//   hs : head init channels on the 1-halo tile, planar [channel][position]
// DBG (timing experiments only, engine option head_debug; results are wrong): bit 0 skip phase 1, 1 skip the level's depthwise taps,
// 2 skip the head 1x1, 3 skip the head's depthwise conv and the stores.
template <class T, int TW, int TH, int DBG = 0>
__device__ __forceinline__ void upghost_head_tail(const UpGhostHeadParams& p, const float* x1s, float (*hs)[UGH_THREADS], long b, int bx, int by, int H, int Wd,
                                                  const float* __restrict__ Wdw, const float* __restrict__ bdw, const float* __restrict__ Wh,
                                                  const float* __restrict__ bh, const float* __restrict__ Wdh, const float* __restrict__ bdh) {
    constexpr int CG = UGH_CG, CS = CG + 4, W2 = TW + 4, W1 = TW + 2, H1 = TH + 2;
    // a row of positions starts on a multiple of 16 threads: the 16 lanes that share an LDS access cycle then read 16 consecutive positions of
    // ONE row — stride CS = 20 dwords, conflict-free — instead of straddling two rows whose pitch may be a multiple of the 64 banks
    // (measured on a 30-wide row packed densely: the depthwise taps 52 -> 78 us)
    constexpr int W1P = (W1 + 15) / 16 * 16;
    static_assert(W1P * H1 <= UGH_THREADS, "one halo position per thread");
    const int tid = threadIdx.x;
    const bool live = tid % W1P < W1 && tid / W1P < H1;
    const int ly_ = live ? tid / W1P : 0, lx_ = live ? tid % W1P : 0;
    const int oy = by + ly_ - 1, ox = bx + lx_ - 1;
    const bool inside = live && oy >= 0 && oy < H && ox >= 0 && ox < Wd;
    {
        float hv[UGH_IMAX];
        ACH_UNROLL
        for (int j = 0; j < UGH_IMAX; ++j) hv[j] = 0.f;
        if (inside) {
            float f[2 * CG];
            ACH_UNROLL
            for (int c = 0; c < CG; ++c) f[CG + c] = bdw[c];
            if (!(DBG & 2))
            ACH_UNROLL
            for (int k = 0; k < 9; ++k) {
                const float* s = x1s + ((ly_ + k / 3) * W2 + lx_ + k % 3) * CS;     // 2-halo coords of the tap
                ACH_UNROLL
                for (int c4 = 0; c4 < CG; c4 += 4) {
                    const float4 sv = *reinterpret_cast<const float4*>(s + c4);
                    const float* wk = Wdw + k * CG + c4;
                    f[CG + c4] += sv.x * wk[0]; f[CG + c4 + 1] += sv.y * wk[1]; f[CG + c4 + 2] += sv.z * wk[2]; f[CG + c4 + 3] += sv.w * wk[3];
                    if (k == 4) { f[c4] = sv.x; f[c4 + 1] = sv.y; f[c4 + 2] = sv.z; f[c4 + 3] = sv.w; }
                }
            }
            ACH_UNROLL
            for (int c = 0; c < CG; ++c) f[CG + c] = f[CG + c] > 0.f ? f[CG + c] : 0.f;
            if (p.F && ly_ >= 1 && ly_ <= TH && lx_ >= 1 && lx_ <= TW) {
                T* fo = static_cast<T*>(p.F) + ((b * H + oy) * long(Wd) + ox) * p.ldf;
                ACH_UNROLL
                for (int c4 = 0; c4 < 2 * CG; c4 += 4) { const float v4[4] = {f[c4], f[c4 + 1], f[c4 + 2], f[c4 + 3]}; Store<T>::st4(fo + c4, v4); }
            }
            if (!(DBG & 4))
            ACH_UNROLL
            for (int j = 0; j < UGH_IMAX; ++j)
                if (j < p.init) {                                   // two packed partial sums per output channel (v_pk_fma_f32)
                    const float* w = Wh + j * 2 * CG;
                    f32x2 a = {bh[j], 0.f};
                    ACH_UNROLL
                    for (int c = 0; c < 2 * CG; c += 2) { const f32x2 wv = {w[c], w[c + 1]}, fv = {f[c], f[c + 1]}; a += wv * fv; }
                    const float r = a[0] + a[1];
                    hv[j] = r > 0.f ? r : 0.f;
                }
        }
        ACH_UNROLL
        for (int j = 0; j < UGH_IMAX; ++j) hs[j][tid] = hv[j];
    }
    __syncthreads();
    // ---- outputs: the interior positions
    if (inside && ly_ >= 1 && ly_ <= TH && lx_ >= 1 && lx_ <= TW && !(DBG & 8)) {
        const long HW = long(H) * Wd;
        const long o0 = b * p.oup * HW + long(oy) * Wd + ox;
        for (int j = 0; j < p.init && j < p.oup; ++j) st_user<T>(p.out, o0 + j * HW, hs[j][tid], p.out_bf16 != 0);
        for (int j = 0; j < p.nch; ++j) {
            float a = bdh[j];
            ACH_UNROLL
            for (int k = 0; k < 9; ++k) a += hs[j][(ly_ - 1 + k / 3) * W1P + lx_ - 1 + k % 3] * Wdh[k * p.nch + j];
            st_user<T>(p.out, o0 + (p.init + j) * HW, a > 0.f ? a : 0.f, p.out_bf16 != 0);
        }
    }
}

// Tile 30x6 outputs: its 1-pixel halo is exactly 32x8 = 256 positions = one per thread.  LDS layouts are chosen for the
// access patterns: x1 rows padded to 20 floats (thread = position reads 16 consecutive floats: stride 20 dwords is
// conflict-free for ds_read_b128), h1 stored planar [channel][position] (thread = pixel reads one channel at a time).
// (Measured and rejected in round 2: a 28x12 tile on 512 threads, whose 2-halo is exactly one position per thread — 1.5 instead of
//  2.8 thread-slots of the bilinear phase per output pixel, 25 % fewer VALU instructions per output in total — ran SLOWER, 160 -> 166 us
//  (lane) and 215 -> 233 us (semantic): the kernel is not bound by its instruction count)
template <class T, int DBG = 0>
__global__ __launch_bounds__(UGH_THREADS) void upghost_head_kernel(const UpGhostHeadParams p, const float* __restrict__ Wdw, const float* __restrict__ bdw,
                                                           const float* __restrict__ Wh, const float* __restrict__ bh,
                                                           const float* __restrict__ Wdh, const float* __restrict__ bdh) { f16_sat_mode<T>();
    constexpr int TW = UGH_TW, TH = UGH_TH, CG = UGH_CG, CS = CG + 4;
    constexpr int W2 = TW + 4, H2 = TH + 4;
    __shared__ float x1s[H2 * W2 * CS];
    __shared__ float hs[UGH_IMAX][UGH_THREADS];
    const int H = 2 * p.h, Wd = 2 * p.w;
    // XCD-aware tile order: workgroup w runs on XCD w % 8 (observed dispatch order; a speed assumption only).  Give every XCD
    // a contiguous run of tiles, i.e. whole samples, so the halo re-reads of t and the partial-line NCHW writes of neighbouring
    // tiles meet in ONE L2 instead of going to HBM from eight (rocprofv3 FETCH_SIZE showed 3.4x the algorithmic reads before).
    const int tiles_x = (Wd + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const unsigned nwg = gridDim.x;
    const unsigned wg = xcd_block(blockIdx.x, nwg);
    const int bx = int(wg % tiles_x) * TW, by = int((wg / tiles_x) % tiles_y) * TH;
    const long b = wg / (unsigned(tiles_x) * tiles_y);
    const int tid = threadIdx.x;
    if (!(DBG & 1))
    {   // ---- x1 on the 2-halo tile: thread = position (all 16 channels), so the bilinear geometry is computed once per position;
        // loads are unconditional from clamped coordinates (positions outside the map / past the tile are zeroed afterwards)
        const float sy = p.sy, sx = p.sx;
        const T* Tq = static_cast<const T*>(p.Tq) + b * p.h * long(p.w) * p.ldt;
        const int ldt = int(p.ldt);
        constexpr int ROUNDS = (H2 * W2 + 255) / 256;
        ACH_UNROLL
        for (int r = 0; r < ROUNDS; ++r) {
            const int pos_raw = tid + r * 256;
            const int pos = pos_raw < H2 * W2 ? pos_raw : H2 * W2 - 1;
            const int py = pos / W2, oy = by + py - 2, ox = bx + (pos - py * W2) - 2;
            const bool ok = oy >= 0 && oy < H && ox >= 0 && ox < Wd;
            const int cy = oy < 0 ? 0 : (oy >= H ? H - 1 : oy), cx = ox < 0 ? 0 : (ox >= Wd ? Wd - 1 : ox);
            const float fy = sy * float(cy), fx = sx * float(cx);
            int y0 = int(fy), x0 = int(fx);
            if (y0 > p.h - 1) y0 = p.h - 1;
            if (x0 > p.w - 1) x0 = p.w - 1;
            const int dy = y0 < p.h - 1 ? p.w * ldt : 0, dx = x0 < p.w - 1 ? ldt : 0;
            const float ly = fy - float(y0), lx = fx - float(x0), hy = 1.f - ly, hx = 1.f - lx;
            const float w00 = ok ? hy * hx : 0.f, w01 = ok ? hy * lx : 0.f, w10 = ok ? ly * hx : 0.f, w11 = ok ? ly * lx : 0.f;
            const T* t0 = Tq + (y0 * p.w + x0) * ldt;
            ACH_UNROLL
            for (int c8 = 0; c8 < CG; c8 += 8) {
                float a[8], bq[8], cc[8], d[8];
                Store<T>::ld8(t0 + c8, a);
                Store<T>::ld8(t0 + dx + c8, bq);
                Store<T>::ld8(t0 + dy + c8, cc);
                Store<T>::ld8(t0 + dy + dx + c8, d);
                float v[8];
                ACH_UNROLL
                for (int i = 0; i < 8; ++i) { const float t = w00 * a[i] + w01 * bq[i] + w10 * cc[i] + w11 * d[i]; v[i] = t > 0.f ? t : 0.f; }
                if (pos_raw < H2 * W2) {
                    *reinterpret_cast<float4*>(x1s + pos * CS + c8) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(x1s + pos * CS + c8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
        }
    }
    __syncthreads();
    upghost_head_tail<T, TW, TH, DBG>(p, x1s, hs, b, bx, by, H, Wd, Wdw, bdw, Wh, bh, Wdh, bdh);
}

// The same fused level with the bilinear phase on the matrix cores (bf16 storage).  x2 bilinear interpolation is separable:
//   v[i][x][c]  = sum_j Ax[x][j] t[i][j][c]            along x: a 16-pixel segment needs <= 10 source columns -> ONE MFMA per (source row,
//                                                      segment), K = 16 columns x {hi, lo}: the fp32 interpolation weight is split into two bf16
//                                                      halves so that the product is exact to 2^-17 (plain bf16 weights would cost 2^-9)
//   x1[y][x][c] = relu(hy v[y0][x][c] + ly v[y0+1][x][c])   along y on the VALU, fp32: 2 FMA per value instead of 4, no bf16 unpacking,
//                                                      one 16-byte load per (source row, channel, segment) instead of four corner gathers per position
// For the MFMA's A operand (16 channels x K source columns, 8 consecutive k per lane) t must be CHANNEL-PLANAR per row, [B*h][16][w]:
// its producer (chain_kernel, planar_w) writes it that way.  The weights are the B operand, built once per wave; the product lands as
// lane (pixel, 4 channels) and goes to LDS in the [position][CS] layout of the tail phases.  Requires w % 4 == 0 (8-byte aligned window).
// NSEG segments of 16 columns x H2 rows, split over the four waves as (segment, RPW consecutive rows); RPW = 5 -> 4 source rows per wave.
constexpr int UGM_NSEG = 2, UGM_TH = 6;       // 28 x 6 outputs per tile
constexpr int UGM_GRID = 0;                   // > 0: that many persistent workgroups walk the tiles (option head_grid); 0: one tile per workgroup
// Instruction budget of phase 1 (the kernel is VALU-issue bound at ~80 % busy, so its time is its instruction count): everything that is
// the same for the whole wave stays off the vector pipe — the per-row geometry (y0, weights) is computed ONCE by lanes 0..RPW-1 and read
// back with v_readlane, the row select is a scalar branch, the load addresses are a constant per-lane offset + a scalar offset (MUBUF
// soffset), the past-the-row mask is a scalar branch taken only by the right-most tiles.
// UGM_PREFETCH (persistent form only): request the next tile's source fragments before the current tile's tail phases
template <int NSEG, int TH, bool UGM_PREFETCH, int DBG = 0>
__global__ __launch_bounds__(UGH_THREADS, 4) void upghost_head_mfma_kernel(const UpGhostHeadParams p, const float* __restrict__ Wdw, const float* __restrict__ bdw,
                                                                const float* __restrict__ Wh, const float* __restrict__ bh,
                                                                const float* __restrict__ Wdh, const float* __restrict__ bdh) {
    typedef bf16_t T;
    constexpr int TW = 16 * NSEG - 4, CG = UGH_CG, CS = CG + 4;
    constexpr int W2 = TW + 4, H2 = TH + 4;
    constexpr int RPW = H2 * NSEG / 4, NSRC = (RPW - 1) / 2 + 2;
    static_assert(RPW * 4 == H2 * NSEG && (NSEG == 1 || NSEG == 2 || NSEG == 4), "rows split evenly over the four waves");
    static_assert(NSRC <= 4, "source rows per wave");
    __shared__ float x1s[H2 * W2 * CS];
    __shared__ float hs[UGH_IMAX][UGH_THREADS];
    const int H = 2 * p.h, Wd = 2 * p.w;
    const int tiles_x = (Wd + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const unsigned per_frame = unsigned(tiles_x) * tiles_y, ntiles = per_frame * unsigned(p.B);
    // tiles of this workgroup: workgroup w runs on XCD w % 8; every XCD takes one contiguous eighth of the tiles (whole frames: halo
    // re-reads of t and the partial-line NCHW writes of neighbouring tiles meet in one L2), its workgroups interleaved over that run
    unsigned t_begin, t_end, t_step;
    if (gridDim.x % 8 == 0) {
        const unsigned xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per = gridDim.x >> 3;
        const unsigned lo = unsigned((unsigned long long)(ntiles) * xcd / 8), hi = unsigned((unsigned long long)(ntiles) * (xcd + 1) / 8);
        t_begin = lo + li; t_end = hi; t_step = per;
    } else {
        t_begin = blockIdx.x; t_end = ntiles; t_step = gridDim.x;
    }
    const int lane = threadIdx.x & 63, wave = wave_uniform(int(threadIdx.x >> 6));
    const int px = lane & 15, g = lane >> 4;
    const int seg = wave % NSEG, r0 = (wave / NSEG) * RPW;
    const float sy = p.sy, sx = p.sx;
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    // the engine keeps 16 readable bytes behind t: the window of the last row's last channel may start in its final 16 bytes
    const BufRsrc tb = make_buf(p.Tq, unsigned(size_t(p.B) * p.h * CG * p.w * sizeof(T)) + 16u);
    const unsigned voff = unsigned((px * p.w + (g & 1) * 8) * int(sizeof(T)));          // channel px, window columns (g & 1) * 8 ..
    struct Geo { int bx, by, win0, ymin, k0; float wa, wb; long b; };
    auto geometry = [&](unsigned t) {
        Geo q;
        q.b = t / per_frame;
        const unsigned rem = t - unsigned(q.b) * per_frame;
        q.by = int(rem / unsigned(tiles_x)) * TH; q.bx = int(rem % unsigned(tiles_x)) * TW;
        // source window of this wave's segment: 16 columns from a multiple of 4 at or below the first pixel's left neighbour
        int xfirst = wave_uniform(int(sx * float(clampi(q.bx - 2 + seg * 16, Wd - 1))));
        if (xfirst > p.w - 1) xfirst = p.w - 1;
        q.win0 = xfirst & ~3;
        // row geometry: lane r < RPW holds output row r0 + r of this wave (other lanes compute a copy of the last row: harmless)
        const int oy = q.by - 2 + r0 + (lane < RPW ? lane : RPW - 1);
        const bool oky = oy >= 0 && oy < H;
        const float fy = sy * float(clampi(oy, H - 1));
        int y0 = int(fy);
        if (y0 > p.h - 1) y0 = p.h - 1;
        const float ly = oky ? fy - float(y0) : 0.f, hy = oky ? 1.f - (fy - float(y0)) : 0.f;
        const bool last = y0 >= p.h - 1;                            // bottom row: both taps are row y0
        q.wa = last ? hy + ly : hy; q.wb = last ? 0.f : ly;
        q.ymin = wave_lane_i32(y0, 0);
        q.k0 = y0 - q.ymin;                                         // 0 .. NSRC-2 (RPW rows span < RPW/2 source rows)
        return q;
    };
    uint4 tf[NSRC];
    auto request = [&](const Geo& q) {          // A operand fragments of source rows ymin .. ymin + NSRC - 1
        ACH_UNROLL
        for (int k = 0; k < NSRC; ++k) {
            const int sr = q.ymin + k < p.h ? q.ymin + k : p.h - 1;
            const unsigned soff = unsigned(((q.b * p.h + sr) * CG) * long(p.w) + q.win0) * unsigned(sizeof(T));
            tf[k] = buf_load16s(tb, voff, soff);
        }
    };
    Geo q = geometry(t_begin < t_end ? t_begin : 0u);
    if (t_begin < t_end && !(DBG & 1)) request(q);
    for (unsigned t = t_begin; t < t_end; t += t_step) {
        const int bx = q.bx, by = q.by;
        const long b = q.b;
        if (!(DBG & 1)) {
            uint4 wfrag;
            {   // B operand: this lane's pixel, k = g*8 + i -> window column (g&1)*8 + i, hi half (g < 2) or lo half of the weight
                const int ox = bx - 2 + seg * 16 + px;
                const bool okx = ox >= 0 && ox < Wd;
                const float fx = sx * float(clampi(ox, Wd - 1));
                int x0 = int(fx);
                if (x0 > p.w - 1) x0 = p.w - 1;
                const bool edge = x0 >= p.w - 1;                    // right column: both taps are column x0
                const float lx = edge ? 0.f : fx - float(x0), hx = edge ? 1.f : 1.f - (fx - float(x0));
                const float hxh = bf16_to_f32(f32_to_bf16(hx)), lxh = bf16_to_f32(f32_to_bf16(lx));
                const float wa = okx ? (g < 2 ? hxh : hx - hxh) : 0.f, wb = okx ? (g < 2 ? lxh : lx - lxh) : 0.f;
                const int rel = x0 - q.win0 - (g & 1) * 8;          // slot of the left tap in this lane's 8 columns (may be outside 0..7)
                float wv[8];
                ACH_UNROLL
                for (int i = 0; i < 8; ++i) wv[i] = (i == rel ? wa : 0.f) + (i == rel + 1 ? wb : 0.f);
                wfrag = frag_pack<T>(wv);
            }
            if (q.win0 + 16 > p.w) {            // right-most tiles: the window runs past the row (another channel's data, or the guard bytes): zero it
                const int colb = q.win0 + (g & 1) * 8;
                const bool keep_lo = colb + 4 <= p.w, keep_hi = colb + 8 <= p.w;
                ACH_UNROLL
                for (int k = 0; k < NSRC; ++k) {
                    tf[k].x = keep_lo ? tf[k].x : 0u; tf[k].y = keep_lo ? tf[k].y : 0u; tf[k].z = keep_hi ? tf[k].z : 0u; tf[k].w = keep_hi ? tf[k].w : 0u;
                }
            }
            f32x4 v[NSRC];
            ACH_UNROLL
            for (int k = 0; k < NSRC; ++k) {
                v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f;
                mfma16<T>(tf[k], wfrag, v[k]);
            }
            const Geo cur = q;
            if (UGM_PREFETCH && t + t_step < t_end) { q = geometry(t + t_step); request(q); }     // in flight during this tile's tail
            ACH_UNROLL
            for (int r = 0; r < RPW; ++r) {
                const int k0 = wave_lane_i32(cur.k0, r);
                const float wa = wave_lane_f32(cur.wa, r), wb = wave_lane_f32(cur.wb, r);
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                ACH_UNROLL
                for (int kk = 0; kk < NSRC - 1; ++kk)
                    if (kk == k0) {
                        ACH_UNROLL
                        for (int c = 0; c < 4; ++c) { const float s = wa * v[kk][c] + wb * v[kk + 1][c]; o[c] = s > 0.f ? s : 0.f; }
                    }
                const int pos = (r0 + r) * W2 + seg * 16 + px;
                *reinterpret_cast<float4*>(x1s + pos * CS + g * 4) = make_float4(o[0], o[1], o[2], o[3]);
            }
            if (!UGM_PREFETCH && t + t_step < t_end) { q = geometry(t + t_step); }
        } else if (t + t_step < t_end) {
            q = geometry(t + t_step);
        }
        __syncthreads();
        upghost_head_tail<T, TW, TH, DBG>(p, x1s, hs, b, bx, by, H, Wd, Wdw, bdw, Wh, bh, Wdh, bdh);
        // the next tile's phase 1 writes x1s only; its barrier orders this tile's reads of hs before the next writes to it
        if (!UGM_PREFETCH && !(DBG & 1) && t + t_step < t_end) request(q);
    }
}

}  // namespace ach
