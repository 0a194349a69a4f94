// k_dechead.h — last decoder level + Ghost segmentation head as a ROW-WALKING kernel (bf16 engine; round 3).
//
//   t  (low resolution, 16 channels)  ->  x1 = relu(bilinear x2 (t))             neck/ghostdualfpn.py:28-39 (Upsample), :175-197
//                                         x2 = relu(dw3x3(x1) + b)                backbone/conv_utils/ghost_conv.py:6-29 (cheap operation)
//                                         h  = relu(Wh [x1 | x2] + bh)            head GhostModule, primary conv (32 -> init)
//                                         out = [h | relu(dw3x3(h) + b')][:oup]   head cheap operation, NCHW
//
// The tile kernel it replaces (upghost_head_kernel, k_nhwc.h) stages x1 of a 34 x 10 halo tile in LDS, is LDS- and VALU-issue bound at
// 10 % of the HBM roofline and spends 4 of 5 VALU instructions on something other than an FMA.  Here NOTHING goes
// through LDS and nothing is recomputed along y:
//   * a WAVE owns a strip of 16 columns (12 of them produce outputs; the two cascaded 3x3 windows need 2 columns of halo per side) and
//     walks DOWN a band of rows.  Lane (n, g) = column n of the strip, channel group g (4 of the level's 16 channels).
//   * x neighbours come from DPP row shifts (row_shr:1 / row_shl:1 inside the 16-lane row that IS the strip) — no LDS, no barrier;
//     y neighbours are the two previous rows, still in registers (a three-row rolling window of x1 and of h).
//   * the bilinear source rows are kept unpacked in registers too: a new source row is fetched every second output row (two 8-byte
//     loads per lane), one row ahead of its use.
//   * lane (n, g)'s eight values [x1 c=4g..4g+3 | x2 c=4g..4g+3] ARE k-group g of the B fragment of v_mfma_f32_16x16x32_bf16, so the head's
//     1x1 conv is ONE MFMA per row of the strip (weights permuted on the host: A fragment in 4 VGPRs).  Head channel jj lands in accumulator
//     r = jj / 4 of lane group g = jj % 4, i.e. next to its x neighbours again: the head's depthwise conv uses the same DPP shifts.
// Numerics (bf16 engine): x1 / x2 in fp32 registers, rounded to bf16 once as the MFMA operand (the tile kernel kept them in fp32);
// the head weights are bf16 like every other MFMA weight of this engine.
#pragma once
#include "ach_platform.h"

namespace ach {

struct DecHeadParams {
    const void* Tq; long ldt;                 // t at low resolution [B,h,w,ldt] bf16 (16 real channels)
    void* F; long ldf;                        // optional [B,2h,2w,32] tap of [x1 | x2] (nullptr in production plans)
    void* out;                                // NCHW [B,oup,2h,2w]
    const float* Wdw; const float* bdw;       // level cheap op: [9][16], [16]   (BN folded)
    const uint4* Afrag;                       // [64] head primary conv as the bf16 A fragment (rows / k permuted, see above)
    const float* bh;                          // [8]  head primary bias by channel jj (zero beyond init)
    const float* Wdh; const float* bdh;       // head cheap op: [9][8], [8] by channel jj (zero beyond nch)
    int B, h, w, init, nch, oup;
    float sy, sx;                             // (h-1)/(2h-1), (w-1)/(2w-1): align_corners source scale
    int band_rows, bands, strips;
};
constexpr int DH_VALID = 12;                  // output columns per 16-lane strip
#ifndef ACH_DH_WAVES
#define ACH_DH_WAVES 3
#endif
constexpr int DH_WAVES = ACH_DH_WAVES;
#ifndef ACH_DH_WG_WAVES
#define ACH_DH_WG_WAVES 1          // waves (independent strips) per workgroup of the two-column kernel
#endif
#ifndef ACH_DH_MFMA32
#define ACH_DH_MFMA32 0            // 1: one v_mfma_f32_16x16x32_bf16 per column (see dh_mfma)
#endif        // register budget: waves per SIMD the compiler must fit

// a + (value of the lane one column to the left / right inside the 16-lane row; 0 at the row's ends): ONE VALU instruction each
// (v_add_f32 with a DPP row shift on its first source).  The depthwise 3x3 taps are arranged so that only the three per-column partial
// sums of a row are shifted — two shifted adds per channel instead of six shifted operands.
#if !defined(ACH_HOSTEMU) && defined(ACH_DH_NO_ASM)
__device__ __forceinline__ float add_from_left(float a, float v) { return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); }
__device__ __forceinline__ float add_from_right(float a, float v) { return a + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true)); }
#elif defined(ACH_HOSTEMU)
__device__ inline float add_from_left(float a, float v) { const int l = int(threadIdx.x) & 63; const float o = __shfl(v, (l & 15) ? l - 1 : l); return a + ((l & 15) ? o : 0.f); }
__device__ inline float add_from_right(float a, float v) { const int l = int(threadIdx.x) & 63; const float o = __shfl(v, (l & 15) != 15 ? l + 1 : l); return a + ((l & 15) != 15 ? o : 0.f); }
#else
// Inline assembly, because the compiler's vectoriser otherwise pairs the adds into v_pk_add_f32 behind two v_mov_b32_dpp.  The hazard
// recogniser cannot see into an asm statement: a DPP read of a VGPR needs two wait states after the VALU write of that VGPR (measured the
// hard way: without the s_nop the head read stale partial sums on the MI355X while the CPU emulation was right), hence the s_nop 1.
__device__ __forceinline__ float add_from_left(float a, float v) {      // row_shr:1, bound_ctrl: lanes without a source read 0
    float r;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(a));
    return r;
}
__device__ __forceinline__ float add_from_right(float a, float v) {     // row_shl:1
    float r;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(a));
    return r;
}
#endif

// Row geometry of the x2 bilinear interpolation, one entry per OUTPUT row (written by the host with the float arithmetic the tile kernel
// and torch use: fy = sy * float(i); y0 = min(int(fy), h-1); ly = y0 < h-1 ? fy - y0 : 0): the walk reads it with scalar loads instead of
// recomputing it on the VALU for every row of every strip.
struct DecHeadRow { int y0; float ly; };

// four channels at once: o[q] = (c[q] + left neighbour's l[q]) + right neighbour's r[q] — ONE leading s_nop covers all eight DPP reads
// (every l / r is written before the statement starts; the chained second add reads the first one's result as a plain operand)
#if defined(ACH_HOSTEMU) || defined(ACH_DH_NO_ASM)
__device__ inline void combine4(const float (&c)[4], const float (&l)[4], const float (&r)[4], float (&o)[4]) {
    for (int q = 0; q < 4; ++q) o[q] = add_from_right(add_from_left(c[q], l[q]), r[q]);
}
#else
__device__ __forceinline__ void combine4(const float (&c)[4], const float (&l)[4], const float (&r)[4], float (&o)[4]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %12, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %13, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %14, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %15, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
        : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]), "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
}
#endif

// The head's 1x1 on the matrix cores: D += A (16 x 32, packed once by the host) * B (this lane's 8 channels of 16 pixels), as TWO
// v_mfma_f32_16x16x16_bf16 (k = 0..15: the x1 half of the lane's channels, k = 16..31: the x2 half), not one v_mfma_f32_16x16x32_bf16.
// Measured on the MI355X (round 3, tests/test_gpu_parity.py::test_pipelined_submit_wait_equals_plain_calls): with the 16x16x32 form in
// these long-lived, > 128-VGPR waves, waves of OTHER kernels that shared a SIMD with them (the next forward's backbone on the caller's
// stream, in pipelined mode) came out a few bf16 ulps off in whole 16-pixel tiles, run to run; with the register budget forced to 128
// or with the two 16x16x16 instructions the difference is gone (0 of 60 forwards against 3 of 4).  DESIGN.md 4.14 has the experiments.
#if defined(ACH_HOSTEMU)
template <class T> __device__ inline void dh_mfma(const uint4& a, const uint4& b, f32x4& c) { mfma16<T>(a, b, c); }
#else
template <class T> __device__ __forceinline__ void dh_mfma(const uint4& a, const uint4& b, f32x4& c);
template <> __device__ __forceinline__ void dh_mfma<bf16_t>(const uint4& a, const uint4& b, f32x4& c) {
#if ACH_DH_MFMA32 == 2       // experiment: the 16x16x32 form with an EARLY-CLOBBER destination (no register shared with A / B) and C = 0
    { const buf_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w}; asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 3" : "=&v"(c) : "v"(av), "v"(bv)); }
#elif ACH_DH_MFMA32
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
#else
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, make_uint2(a.x, a.y)), __builtin_bit_cast(s16x4, make_uint2(b.x, b.y)), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, make_uint2(a.z, a.w)), __builtin_bit_cast(s16x4, make_uint2(b.z, b.w)), c, 0, 0, 0);
#endif
}
template <> __device__ __forceinline__ void dh_mfma<f16_t>(const uint4& a, const uint4& b, f32x4& c) {
#if ACH_DH_MFMA32 == 2
    { const buf_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w}; asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 3" : "=&v"(c) : "v"(av), "v"(bv)); }
#elif ACH_DH_MFMA32
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
#else
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, make_uint2(a.x, a.y)), __builtin_bit_cast(h16x4, make_uint2(b.x, b.y)), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, make_uint2(a.z, a.w)), __builtin_bit_cast(h16x4, make_uint2(b.z, b.w)), c, 0, 0, 0);
#endif
}
#endif

// max(v, 0) of a value that comes out of inline assembly: one v_max_f32 (the compiler would first canonicalise a value it cannot see into)
#if defined(ACH_HOSTEMU) || defined(ACH_DH_NO_ASM)
__device__ inline float relu_raw(float v) { return v > 0.f ? v : 0.f; }
// relu of two packed 16-bit floats (fp16 and bf16 alike): read as signed 16-bit integers every negative float is below 0 and every non-negative one keeps its
// order, so max(x, 0) per half IS relu — and rounding is monotonic, so relu-after-rounding = rounding-after-relu
__device__ inline uint32_t relu_packed16(uint32_t v) { return ((v & 0x8000u) ? 0u : (v & 0xffffu)) | ((v & 0x80000000u) ? 0u : (v & 0xffff0000u)); }
#else
__device__ __forceinline__ float relu_raw(float v) { float r; asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(v)); return r; }
// (a builtin, NOT inline asm: the value goes straight into an MFMA operand, and the wait states a VALU write needs before a matrix instruction reads the register
//  are only inserted for instructions the compiler can see — as inline asm this produced run-to-run differences whenever other kernels shared the CU)
typedef short dh_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_packed16(uint32_t v) {
    const dh_s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(dh_s16x2, v), dh_s16x2{0, 0});      // v_pk_max_i16
    return __builtin_bit_cast(uint32_t, r);
}
#endif

// DW2: the head's cheap operation has more than four channels (num_seg > 8): accumulator r = 1 takes part in it too.
// DBG (timing experiments only, results are wrong): bit 0 no bilinear, 1 no level depthwise, 2 no MFMA, 3 no head depthwise / stores.
// TAP: also write [x1 | x2] to p.F (parity tests, option full_taps).
template <class T, class IO, bool DW2, bool TAP, int DBG = 0>
__global__ __launch_bounds__(64, DH_WAVES) void dechead_rows_kernel(const DecHeadParams p, const DecHeadRow* __restrict__ rows) { f16_sat_mode<T>();
    const int H = 2 * p.h, Wd = 2 * p.w;
    const unsigned u = xcd_block(blockIdx.x, gridDim.x);
    const int strip = int(u % unsigned(p.strips)), band = int((u / unsigned(p.strips)) % unsigned(p.bands));
    const long b = long(u / (unsigned(p.strips) * unsigned(p.bands)));
    const int lane = int(threadIdx.x) & 63, n = lane & 15, g = lane >> 4;
    const int x = strip * DH_VALID - 2 + n;
    const bool in_x = x >= 0 && x < Wd;
    const bool writer = in_x && n >= 2 && n < 2 + DH_VALID;
    // ---- per-lane bilinear geometry along x (fixed for the whole band)
    const int cx = x < 0 ? 0 : (x >= Wd ? Wd - 1 : x);
    const float fx = p.sx * float(cx);
    int x0 = int(fx);
    if (x0 > p.w - 1) x0 = p.w - 1;
    const int dx = x0 < p.w - 1 ? 1 : 0;
    const float lx = fx - float(x0);
    const float wx0 = in_x ? 1.f - lx : 0.f, wx1 = in_x ? lx : 0.f;        // a column outside the map is the depthwise conv's zero padding
    const T* Tq = static_cast<const T*>(p.Tq) + b * p.h * long(p.w) * p.ldt;          // (uniform)
    const unsigned o0 = unsigned(x0 * int(p.ldt) + 4 * g), o1 = unsigned((x0 + dx) * int(p.ldt) + 4 * g);
    const int rowp = p.w * int(p.ldt);
    // ---- per-lane weights; channel PAIRS (4g, 4g+1) and (4g+2, 4g+3) as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of FMAs per issue)
    f32x2 wl[9][2], bl[2];
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) { const float4 w = *reinterpret_cast<const float4*>(p.Wdw + k * 16 + 4 * g); wl[k][0] = f32x2{w.x, w.y}; wl[k][1] = f32x2{w.z, w.w}; }
    { const float4 w = *reinterpret_cast<const float4*>(p.bdw + 4 * g); bl[0] = f32x2{w.x, w.y}; bl[1] = f32x2{w.z, w.w}; }
    constexpr int NR = DW2 ? 2 : 1;
    float wh[9][NR], bdh[NR], bhv[2];
    ACH_UNROLL
    for (int r = 0; r < NR; ++r) {
        ACH_UNROLL
        for (int k = 0; k < 9; ++k) wh[k][r] = p.Wdh[k * 8 + g + 4 * r];
        bdh[r] = p.bdh[g + 4 * r];
    }
    bhv[0] = p.bh[g]; bhv[1] = p.bh[g + 4];
    const uint4 afrag = p.Afrag[lane];
    // a lane whose column lies outside the map, or whose accumulator holds no head channel, keeps h = 0 (the head depthwise conv's zero padding)
    const bool has_h[2] = {in_x && g < p.init, in_x && g + 4 < p.init};
    const bool st_h[2] = {writer && g < p.init && g < p.oup, writer && g + 4 < p.init && g + 4 < p.oup};
    const bool st_d[2] = {writer && g < p.nch, writer && g + 4 < p.nch};
    const long HW = long(H) * Wd;
    IO* out_b = static_cast<IO*>(p.out) + b * p.oup * HW;                               // (uniform)
    const unsigned xo = unsigned(in_x ? x : 0);
    const unsigned off_h[2] = {unsigned(g * HW) + xo, unsigned((g + 4) * HW) + xo};
    const unsigned off_d[2] = {unsigned((p.init + g) * HW) + xo, unsigned((p.init + g + 4) * HW) + xo};
    // ---- source rows of t, unpacked: ta = row cy, tb = row cy + 1 (clamped), tn = raw row cy + 2 (clamped), fetched one row ahead
    const int r0 = band * p.band_rows, r1 = (r0 + p.band_rows < H) ? r0 + p.band_rows : H;
    auto load_raw = [&](int r, uint2 (&raw)[2]) {
        const int rr = r < 0 ? 0 : (r > p.h - 1 ? p.h - 1 : r);
        const T* q = Tq + long(rr) * rowp;                              // uniform base + per-lane 32-bit offsets
        raw[0] = *reinterpret_cast<const uint2*>(q + o0);
        raw[1] = *reinterpret_cast<const uint2*>(q + o1);
    };
    auto unpack = [&](const uint2 (&raw)[2], f32x2 (&o)[2][2]) {            // [column][channel pair]
        ACH_UNROLL
        for (int c = 0; c < 2; ++c) {
            o[c][0] = f32x2{H16<T>::lo(raw[c].x), H16<T>::hi(raw[c].x)};
            o[c][1] = f32x2{H16<T>::lo(raw[c].y), H16<T>::hi(raw[c].y)};
        }
    };
    const int i_first = r0 - 2 < 0 ? 0 : r0 - 2;
    int cy = rows[i_first].y0;
    f32x2 ta[2][2], tb[2][2];
    uint2 tn[2];
    { uint2 raw[2]; load_raw(cy, raw); unpack(raw, ta); load_raw(cy + 1, raw); unpack(raw, tb); load_raw(cy + 2, tn); }
    const f32x2 zero2 = {0.f, 0.f};
    f32x2 w0[2] = {zero2, zero2}, w1[2] = {zero2, zero2}, w2[2] = {zero2, zero2};                         // rolling window of x1 rows (channel pairs)
    float v0[2] = {0.f, 0.f}, v1[2] = {0.f, 0.f}, v2[2] = {0.f, 0.f};                                    // rolling window of h rows

    // One step of the walk: x1 row i into xp; x2 / h of row i-1 (x1 rows xm, xc, xp) into hp; output row i-2 (h rows hm, hc, hp).  The caller
    // rotates the roles of the three window slots, so nothing is copied between steps.  Every stage runs on every step — rows outside the
    // map get zero interpolation weights / a zeroed h, rows outside the band only lose their stores — so the steady state has no branches.
    auto step = [&](const int i, f32x2 (&xm)[2], f32x2 (&xc)[2], f32x2 (&xp)[2], float (&hm)[2], float (&hc)[2], float (&hp)[2]) {
        // ---- A: x1 row i
        {
            const bool row_ok = i >= 0 && i < H;
            const DecHeadRow rg = rows[row_ok ? i : 0];                    // scalar load: uniform address
            if (row_ok && rg.y0 > cy) {                                    // (the scale is below 1/2: at most one new source row per output row)
                // (a two-slot ping-pong instead of this copy was compiled into ~30 selects per step: slower than the four 64-bit moves)
                ACH_UNROLL
                for (int c = 0; c < 2; ++c) { ta[c][0] = tb[c][0]; ta[c][1] = tb[c][1]; }
                unpack(tn, tb);
                ++cy;
                load_raw(cy + 2, tn);
            }
            const float ly = row_ok ? rg.ly : 0.f, hy = row_ok ? 1.f - rg.ly : 0.f;
            const float w00 = hy * wx0, w01 = hy * wx1, w10 = ly * wx0, w11 = ly * wx1;
            ACH_UNROLL
            for (int q = 0; q < 2; ++q) {
                const f32x2 v = (DBG & 1) ? ta[0][q] : w00 * ta[0][q] + w01 * ta[1][q] + w10 * tb[0][q] + w11 * tb[1][q];
                xp[q] = f32x2{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f};
            }
        }
        // ---- B: x2 and h of row i-1
        {
            const int rb = i - 1;
            f32x2 x2[2] = {bl[0], bl[1]};
            if (!(DBG & 2)) {
                // per-column partial sums of the left / centre / right taps; the neighbours' sums arrive by two shifted adds per channel
                f32x2 sl[2], sr[2], sc[2];
                ACH_UNROLL
                for (int q = 0; q < 2; ++q) {
                    sl[q] = wl[0][q] * xm[q] + wl[3][q] * xc[q] + wl[6][q] * xp[q];               // what this column contributes to x + 1
                    sr[q] = wl[2][q] * xm[q] + wl[5][q] * xc[q] + wl[8][q] * xp[q];               // ... to x - 1
                    sc[q] = x2[q] + wl[1][q] * xm[q] + wl[4][q] * xc[q] + wl[7][q] * xp[q];
                }
                const float c4[4] = {sc[0][0], sc[0][1], sc[1][0], sc[1][1]}, l4[4] = {sl[0][0], sl[0][1], sl[1][0], sl[1][1]}, r4[4] = {sr[0][0], sr[0][1], sr[1][0], sr[1][1]};
                float o4[4];
                combine4(c4, l4, r4, o4);
                x2[0] = f32x2{o4[0], o4[1]}; x2[1] = f32x2{o4[2], o4[3]};
            }
            ACH_UNROLL
            for (int q = 0; q < 2; ++q) x2[q] = f32x2{relu_raw(x2[q][0]), relu_raw(x2[q][1])};
            const bool row_in = rb >= 0 && rb < H;
            if (TAP && row_in && writer && rb >= r0 && rb < r1) {
                T* fo = static_cast<T*>(p.F) + ((b * H + rb) * long(Wd) + x) * p.ldf + 4 * g;
                const float a1[4] = {xc[0][0], xc[0][1], xc[1][0], xc[1][1]}, a2[4] = {x2[0][0], x2[0][1], x2[1][0], x2[1][1]};
                Store<T>::st4(fo, a1);
                Store<T>::st4(fo + 16, a2);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (!(DBG & 4)) {
                const uint4 bfrag = make_uint4(H16<T>::pack(xc[0][0], xc[0][1]), H16<T>::pack(xc[1][0], xc[1][1]), H16<T>::pack(x2[0][0], x2[0][1]), H16<T>::pack(x2[1][0], x2[1][1]));
                dh_mfma<T>(afrag, bfrag, acc);
            }
            ACH_UNROLL
            for (int r = 0; r < 2; ++r) { const float v = acc[r] + bhv[r]; hp[r] = (row_in && has_h[r] && v > 0.f) ? v : 0.f; }
        }
        // ---- C: output row i-2
        if (!(DBG & 8)) {
            const int ro = i - 2;
            const bool row_st = ro >= r0 && ro < r1;
            IO* orow = out_b + long(row_st ? ro : r0) * Wd;            // uniform base; per-lane 32-bit offsets
            ACH_UNROLL
            for (int r = 0; r < 2; ++r)
                if (row_st && st_h[r]) Store<IO>::st(orow + off_h[r], hc[r]);
            ACH_UNROLL
            for (int r = 0; r < NR; ++r) {
                float a = bdh[r] + wh[1][r] * hm[r] + wh[4][r] * hc[r] + wh[7][r] * hp[r];
                a = add_from_left(a, wh[0][r] * hm[r] + wh[3][r] * hc[r] + wh[6][r] * hp[r]);
                a = add_from_right(a, wh[2][r] * hm[r] + wh[5][r] * hc[r] + wh[8][r] * hp[r]);
                if (row_st && st_d[r]) Store<IO>::st(orow + off_d[r], relu_raw(a));
            }
        }
    };
    ACH_NO_UNROLL
    for (int i = r0 - 2; i <= r1 + 1; i += 3) {
        step(i, w0, w1, w2, v0, v1, v2);
        step(i + 1, w1, w2, w0, v1, v2, v0);          // (up to two steps beyond the band: their stores are masked)
        step(i + 2, w2, w0, w1, v2, v0, v1);
    }
}

// ------------------------------------------------------------------------------------------ two columns per lane
// The same walk with TWO adjacent columns per lane: a strip is 32 columns wide (lane n holds columns 2n and 2n + 1), 28 of them produce
// outputs instead of 12 of 16, and the per-lane channel count stays at four — the depthwise weights are still 36 registers.  Half of the
// x neighbours are now in the lane itself: out[2n] = centre + (lane n-1's right column, one DPP add) + (own right column, a plain add),
// out[2n+1] = centre + (own left column) + (lane n+1's left column, one DPP add).  The head's 1x1 conv is one MFMA per column set; its
// depthwise conv packs the two columns of a channel into one v_pk_fma_f32; the two columns' outputs are adjacent in memory and leave as ONE
// dword store per channel.  ~145 VALU instructions per 28 columns (5.2 per pixel against 8.9).
constexpr int DH2_VALID = 28;
#ifndef ACH_DH2_HFIRST
#define ACH_DH2_HFIRST 1           // round 4: interpolate along x ONCE per source row (when it arrives), along y per output row — 2 packed operations per value and row
#endif                             // instead of 4, no per-row weight products, half the unpacked-source registers.  0 = round 3's four-corner form
#ifndef ACH_DH2_WAVES
#define ACH_DH2_WAVES 2
#endif
#ifndef ACH_DH2_BUFST
#define ACH_DH2_BUFST 1            // round 4: range-checked buffer stores + a fixed number of memory operations per step (0: exec-masked global stores, loads only when a row arrives)
#endif
#ifndef ACH_DH2_PK16
#define ACH_DH2_PK16 1             // round 4: relu on packed 16-bit pairs, non-existent rows / channels through the bias
#endif
#ifndef ACH_DH2_RPRE
#define ACH_DH2_RPRE 1             // round 4: the row record is requested a step ahead
#endif

// four values: o[q] = c[q] + (left neighbour lane's l[q]);  and  o[q] = c[q] + (right neighbour lane's r[q]) — one s_nop per group
#if defined(ACH_HOSTEMU) || defined(ACH_DH_NO_ASM)
__device__ inline void add4_from_left(const float (&c)[4], const float (&l)[4], float (&o)[4]) { for (int q = 0; q < 4; ++q) o[q] = add_from_left(c[q], l[q]); }
__device__ inline void add4_from_right(const float (&c)[4], const float (&r)[4], float (&o)[4]) { for (int q = 0; q < 4; ++q) o[q] = add_from_right(c[q], r[q]); }
#else
__device__ __forceinline__ void add4_from_left(const float (&c)[4], const float (&l)[4], float (&o)[4]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
        : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]));
}
__device__ __forceinline__ void add4_from_right(const float (&c)[4], const float (&r)[4], float (&o)[4]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
        : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
}
#endif

template <class T, class IO, bool DW2, bool TAP>
__global__ __launch_bounds__(64 * ACH_DH_WG_WAVES, ACH_DH2_WAVES) void dechead_rows2_kernel(const DecHeadParams p, const DecHeadRow* __restrict__ rows) { f16_sat_mode<T>();
    const int H = 2 * p.h, Wd = 2 * p.w;
    const unsigned u = xcd_block(blockIdx.x, gridDim.x) * ACH_DH_WG_WAVES + unsigned(wave_uniform(int(threadIdx.x) >> 6));
    if (u >= unsigned(p.strips) * unsigned(p.bands) * unsigned(p.B)) return;
    const int strip = int(u % unsigned(p.strips)), band = int((u / unsigned(p.strips)) % unsigned(p.bands));
    const long b = long(u / (unsigned(p.strips) * unsigned(p.bands)));
    const int lane = int(threadIdx.x) & 63, n = lane & 15, g = lane >> 4;
    const int xa = strip * DH2_VALID - 2 + 2 * n;                 // column A; column B = xa + 1 (xa is even, the map width is even)
    const bool in_x = xa >= 0 && xa < Wd;
    const bool writer = in_x && n >= 1 && n < 15;
    const T* Tq = static_cast<const T*>(p.Tq) + b * p.h * long(p.w) * p.ldt;
    const int rowp = p.w * int(p.ldt);
    // ---- per-lane bilinear geometry along x, per column
    unsigned o0[2], o1[2];
    float wx0[2], wx1[2];
    ACH_UNROLL
    for (int c = 0; c < 2; ++c) {
        const int x = xa + c;
        const int cx = x < 0 ? 0 : (x >= Wd ? Wd - 1 : x);
        const float fx = p.sx * float(cx);
        int x0 = int(fx);
        if (x0 > p.w - 1) x0 = p.w - 1;
        const int dx = x0 < p.w - 1 ? 1 : 0;
        const float lx = fx - float(x0);
        wx0[c] = in_x ? 1.f - lx : 0.f; wx1[c] = in_x ? lx : 0.f;
        o0[c] = unsigned(x0 * int(p.ldt) + 4 * g); o1[c] = unsigned((x0 + dx) * int(p.ldt) + 4 * g);
    }
    f32x2 wl[9][2], bl[2];
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) { const float4 w = *reinterpret_cast<const float4*>(p.Wdw + k * 16 + 4 * g); wl[k][0] = f32x2{w.x, w.y}; wl[k][1] = f32x2{w.z, w.w}; }
    { const float4 w = *reinterpret_cast<const float4*>(p.bdw + 4 * g); bl[0] = f32x2{w.x, w.y}; bl[1] = f32x2{w.z, w.w}; }
    constexpr int NR = DW2 ? 2 : 1;
    float wh[9][NR], bdh[NR], bhv[2];
    ACH_UNROLL
    for (int r = 0; r < NR; ++r) {
        ACH_UNROLL
        for (int k = 0; k < 9; ++k) wh[k][r] = p.Wdh[k * 8 + g + 4 * r];
        bdh[r] = p.bdh[g + 4 * r];
    }
    bhv[0] = p.bh[g]; bhv[1] = p.bh[g + 4];
    const uint4 afrag = p.Afrag[lane];
    const bool has_h[2] = {in_x && g < p.init, in_x && g + 4 < p.init};
    const bool st_h[2] = {writer && g < p.init && g < p.oup, writer && g + 4 < p.init && g + 4 < p.oup};
    const bool st_d[2] = {writer && g < p.nch, writer && g + 4 < p.nch};
    const long HW = long(H) * Wd;
    IO* out_b = static_cast<IO*>(p.out) + b * p.oup * HW;
    const unsigned xo = unsigned(in_x ? xa : 0);
    const unsigned off_h[2] = {unsigned(g * HW) + xo, unsigned((g + 4) * HW) + xo};
    const unsigned off_d[2] = {unsigned((p.init + g) * HW) + xo, unsigned((p.init + g + 4) * HW) + xo};
    // stores: per-lane byte offsets inside the sample's output, BUF_OOB where the lane has nothing to store (dropped by the buffer range check: no exec masks, no
    // skip branches — see buf_store4); a row outside the band stores through a zero-length resource
    const unsigned off_hb[2] = {st_h[0] ? off_h[0] * unsigned(sizeof(IO)) : BUF_OOB, st_h[1] ? off_h[1] * unsigned(sizeof(IO)) : BUF_OOB};
    const unsigned off_db[2] = {st_d[0] ? off_d[0] * unsigned(sizeof(IO)) : BUF_OOB, st_d[1] ? off_d[1] * unsigned(sizeof(IO)) : BUF_OOB};
    const unsigned out_bytes = unsigned(p.oup) * unsigned(HW) * unsigned(sizeof(IO));
    const int r0 = band * p.band_rows, r1 = (r0 + p.band_rows < H) ? r0 + p.band_rows : H;
    // source rows: [column][left / right source column] raw and unpacked (channel pairs)
    // (byte offsets from a wave-uniform base: the loads / stores then take the base from SGPRs — `global_load v, v_off, s[base]` — instead of a 64-bit
    //  VALU add per access: 23 v_lshl_add_u64 per three rows in round 3's ISA)
    const char* Tqb = reinterpret_cast<const char*>(Tq);
    const unsigned rowpb = unsigned(rowp) * unsigned(sizeof(T));
    unsigned o0b[2], o1b[2];
    ACH_UNROLL
    for (int c = 0; c < 2; ++c) { o0b[c] = o0[c] * unsigned(sizeof(T)); o1b[c] = o1[c] * unsigned(sizeof(T)); }
    auto load_raw = [&](int r, uint2 (&raw)[2][2]) {
        const int rr = r < 0 ? 0 : (r > p.h - 1 ? p.h - 1 : r);
        const char* q = Tqb + wave_uniform(int(unsigned(rr) * rowpb));            // (below 2 GiB: plan-time check)
        ACH_UNROLL
        for (int c = 0; c < 2; ++c) { raw[c][0] = *reinterpret_cast<const uint2*>(q + local_offset(o0b[c])); raw[c][1] = *reinterpret_cast<const uint2*>(q + local_offset(o1b[c])); }
    };
    auto unpack = [&](const uint2 (&raw)[2][2], f32x2 (&o)[2][2][2]) {       // [column][source column][channel pair]
        ACH_UNROLL
        for (int c = 0; c < 2; ++c) {
            ACH_UNROLL
            for (int k = 0; k < 2; ++k) {
                o[c][k][0] = f32x2{H16<T>::lo(raw[c][k].x), H16<T>::hi(raw[c][k].x)};
                o[c][k][1] = f32x2{H16<T>::lo(raw[c][k].y), H16<T>::hi(raw[c][k].y)};
            }
        }
    };
    // ACH_DH2_HFIRST: a source row blended along x for this lane's two columns: [column][channel pair]
    auto hblend = [&](const uint2 (&raw)[2][2], f32x2 (&o)[2][2]) {
        f32x2 u[2][2][2];
        unpack(raw, u);
        ACH_UNROLL
        for (int c = 0; c < 2; ++c) {
            ACH_UNROLL
            for (int q = 0; q < 2; ++q) o[c][q] = wx0[c] * u[c][0][q] + wx1[c] * u[c][1][q];
        }
    };
    const int i_first = r0 - 2 < 0 ? 0 : r0 - 2;
    int cy = rows[i_first].y0;
#if ACH_DH2_HFIRST
    // the two live source rows: `par` (wave-uniform) says which array holds the OLDER one — an arriving row overwrites it and the roles swap, instead of
    // eight register-pair moves per arrival; both orders evaluate hy * older + ly * newer
    f32x2 ha[2][2], hb[2][2];
    uint2 tn[2][2];
    int par = 0;
    { uint2 raw[2][2]; load_raw(cy, raw); hblend(raw, ha); load_raw(cy + 1, raw); hblend(raw, hb); load_raw(cy + 2, tn); }
    // as many (dropped) stores behind the first row's loads as every step issues behind its own: the compiler's count of the memory operations between a step's
    // loads and their use in the next step is then the same on the loop's entry edge and on its back edge — otherwise the first step of every trip takes the
    // smaller one, i.e. waits for the previous step's stores to be acknowledged
#if ACH_DH2_BUFST
    {
        const BufRsrc none = make_buf(out_b, 0u);
        ACH_UNROLL
        for (int r = 0; r < 2 + NR; ++r) buf_store4(none, BUF_OOB, 0u, 0u);
    }
#endif
#else
    f32x2 ta[2][2][2], tb[2][2][2];
    uint2 tn[2][2];
    { uint2 raw[2][2]; load_raw(cy, raw); unpack(raw, ta); load_raw(cy + 1, raw); unpack(raw, tb); load_raw(cy + 2, tn); }
#endif
    const f32x2 zero2 = {0.f, 0.f};
    // rolling windows: x1 [column][channel pair] and h as COLUMN pairs per accumulator r
    f32x2 w0[2][2], w1[2][2], w2[2][2];
    f32x2 v0[2] = {zero2, zero2}, v1[2] = {zero2, zero2}, v2[2] = {zero2, zero2};
    ACH_UNROLL
    for (int c = 0; c < 2; ++c) { ACH_UNROLL for (int q = 0; q < 2; ++q) { w0[c][q] = zero2; w1[c][q] = zero2; w2[c][q] = zero2; } }

    DecHeadRow rnext = rows[(r0 - 2 >= 0 && r0 - 2 < H) ? r0 - 2 : 0];
    auto step = [&](const int i, f32x2 (&xm)[2][2], f32x2 (&xc)[2][2], f32x2 (&xp)[2][2], f32x2 (&hm)[2], f32x2 (&hc)[2], f32x2 (&hp)[2]) {
        // ---- A: x1 row i, both columns
        {
            const bool row_ok = i >= 0 && i < H;
            // this row's record was requested at the end of the previous step (a scalar load waited for where it is issued costs the wave a scalar-cache
            // round trip per row, and two or three waves per SIMD do not hide it; scalar loads return out of order, so the request must not sit between
            // this use and its wait either)
#if ACH_DH2_RPRE
            DecHeadRow rg = rnext;
#else
            DecHeadRow rg = rows[row_ok ? i : 0];
#endif
            // the record is the same in every lane (i is): as scalars, the row-advance test below is a scalar branch and cy, the source-row base and
            // the blend weights stay in SGPRs (the loads inside the branch then keep their SGPR-base form instead of 64-bit per-lane addresses)
            rg.y0 = wave_uniform(rg.y0); rg.ly = __int_as_float(wave_uniform(__float_as_int(rg.ly)));
#if ACH_DH2_HFIRST
            if (row_ok && rg.y0 > cy) {
                if (par == 0) hblend(tn, ha); else hblend(tn, hb);
                par ^= 1;
                ++cy;
#if !ACH_DH2_BUFST
                load_raw(cy + 2, tn);
#endif
            }
#if ACH_DH2_BUFST
            load_raw(cy + 2, tn);          // every step (a repeat of the row already held when nothing arrived): a fixed number of memory operations per step
#endif

            const float ly = row_ok ? rg.ly : 0.f, hy = row_ok ? 1.f - rg.ly : 0.f;
            if (par == 0) {
                ACH_UNROLL
                for (int c = 0; c < 2; ++c) {
                    ACH_UNROLL
                    for (int q = 0; q < 2; ++q) {
                        const f32x2 v = hy * ha[c][q] + ly * hb[c][q];
                        xp[c][q] = f32x2{relu_raw(v[0]), relu_raw(v[1])};
                    }
                }
            } else {
                ACH_UNROLL
                for (int c = 0; c < 2; ++c) {
                    ACH_UNROLL
                    for (int q = 0; q < 2; ++q) {
                        const f32x2 v = hy * hb[c][q] + ly * ha[c][q];
                        xp[c][q] = f32x2{relu_raw(v[0]), relu_raw(v[1])};
                    }
                }
            }
#else
            if (row_ok && rg.y0 > cy) {
                ACH_UNROLL
                for (int c = 0; c < 2; ++c) { ACH_UNROLL for (int k = 0; k < 2; ++k) { ta[c][k][0] = tb[c][k][0]; ta[c][k][1] = tb[c][k][1]; } }
                unpack(tn, tb);
                ++cy;
                load_raw(cy + 2, tn);
            }
            const float ly = row_ok ? rg.ly : 0.f, hy = row_ok ? 1.f - rg.ly : 0.f;
            ACH_UNROLL
            for (int c = 0; c < 2; ++c) {
                const float w00 = hy * wx0[c], w01 = hy * wx1[c], w10 = ly * wx0[c], w11 = ly * wx1[c];
                ACH_UNROLL
                for (int q = 0; q < 2; ++q) {
                    const f32x2 v = w00 * ta[c][0][q] + w01 * ta[c][1][q] + w10 * tb[c][0][q] + w11 * tb[c][1][q];
                    xp[c][q] = f32x2{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f};
                }
            }
#endif
        }
        // ---- B: x2 and h of row i-1
        {
            const int rb = i - 1;
#if ACH_DH2_RPRE
            { const int in = i + 1; rnext = rows[(in >= 0 && in < H) ? in : 0]; }           // next step's record: a whole section B + C ahead of its wait
#endif
            f32x2 sl[2][2], sr[2][2], sc[2][2];
            ACH_UNROLL
            for (int c = 0; c < 2; ++c) {
                ACH_UNROLL
                for (int q = 0; q < 2; ++q) {
                    sl[c][q] = wl[0][q] * xm[c][q] + wl[3][q] * xc[c][q] + wl[6][q] * xp[c][q];       // what this column contributes to column + 1
                    sr[c][q] = wl[2][q] * xm[c][q] + wl[5][q] * xc[c][q] + wl[8][q] * xp[c][q];       // ... to column - 1
                    sc[c][q] = bl[q] + wl[1][q] * xm[c][q] + wl[4][q] * xc[c][q] + wl[7][q] * xp[c][q];
                }
            }
            // column A: + own B's right-tap sums, + lane n-1's B left-tap sums;   column B: + own A's left-tap sums, + lane n+1's A right-tap sums
            float ca[4], cb[4], la[4], rb4[4], oa[4], ob[4];
            ACH_UNROLL
            for (int q = 0; q < 2; ++q) {
                const f32x2 a = sc[0][q] + sr[1][q], bq = sc[1][q] + sl[0][q];
                ca[2 * q] = a[0]; ca[2 * q + 1] = a[1]; cb[2 * q] = bq[0]; cb[2 * q + 1] = bq[1];
                la[2 * q] = sl[1][q][0]; la[2 * q + 1] = sl[1][q][1]; rb4[2 * q] = sr[0][q][0]; rb4[2 * q + 1] = sr[0][q][1];
            }
            add4_from_left(ca, la, oa);
            add4_from_right(cb, rb4, ob);
            const bool row_in = rb >= 0 && rb < H;
            // x2 = relu(.) only feeds the MFMA (and the debug tap): the relu is applied to the PACKED pairs, four v_pk_max_i16 instead of eight v_max_f32
#if ACH_DH2_PK16
            const uint32_t pa0 = relu_packed16(H16<T>::pack(oa[0], oa[1])), pa1 = relu_packed16(H16<T>::pack(oa[2], oa[3]));
            const uint32_t pb0 = relu_packed16(H16<T>::pack(ob[0], ob[1])), pb1 = relu_packed16(H16<T>::pack(ob[2], ob[3]));
#else
            const uint32_t pa0 = H16<T>::pack(relu_raw(oa[0]), relu_raw(oa[1])), pa1 = H16<T>::pack(relu_raw(oa[2]), relu_raw(oa[3]));
            const uint32_t pb0 = H16<T>::pack(relu_raw(ob[0]), relu_raw(ob[1])), pb1 = H16<T>::pack(relu_raw(ob[2]), relu_raw(ob[3]));
#endif
            if (TAP && row_in && writer && rb >= r0 && rb < r1) {
                float x2a[4], x2b[4];
                ACH_UNROLL
                for (int e = 0; e < 4; ++e) { x2a[e] = relu_raw(oa[e]); x2b[e] = relu_raw(ob[e]); }
                T* fo = static_cast<T*>(p.F) + ((b * H + rb) * long(Wd) + xa) * p.ldf + 4 * g;
                const float a1[4] = {xc[0][0][0], xc[0][0][1], xc[0][1][0], xc[0][1][1]}, b1[4] = {xc[1][0][0], xc[1][0][1], xc[1][1][0], xc[1][1][1]};
                Store<T>::st4(fo, a1); Store<T>::st4(fo + 16, x2a);
                Store<T>::st4(fo + p.ldf, b1); Store<T>::st4(fo + p.ldf + 16, x2b);
            }
            f32x4 acca = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
            {
                const uint4 fa = make_uint4(H16<T>::pack(xc[0][0][0], xc[0][0][1]), H16<T>::pack(xc[0][1][0], xc[0][1][1]), pa0, pa1);
                const uint4 fb = make_uint4(H16<T>::pack(xc[1][0][0], xc[1][0][1]), H16<T>::pack(xc[1][1][0], xc[1][1][1]), pb0, pb1);
                dh_mfma<T>(afrag, fa, acca);
                dh_mfma<T>(afrag, fb, accb);
            }
            ACH_UNROLL
            for (int r = 0; r < 2; ++r) {
                // rows / channels that do not exist: a bias no accumulator survives, instead of a compare + select per value
#if ACH_DH2_PK16
                const float bsel = (row_in && has_h[r]) ? bhv[r] : -3.0e38f;
                hp[r] = f32x2{relu_raw(acca[r] + bsel), relu_raw(accb[r] + bsel)};
#else
                const float va = acca[r] + bhv[r], vb = accb[r] + bhv[r];
                const bool ok = row_in && has_h[r];
                hp[r] = f32x2{(ok && va > 0.f) ? va : 0.f, (ok && vb > 0.f) ? vb : 0.f};
#endif
            }
        }
        // ---- C: output row i-2: the two columns of a channel leave as one dword
        {
            const int ro = i - 2;
            const bool row_st = ro >= r0 && ro < r1;
#if ACH_DH2_BUFST
            const BufRsrc orow = make_buf(out_b, row_st ? out_bytes : 0u);
            const unsigned orow_off = unsigned(wave_uniform(int(unsigned(row_st ? ro : r0) * unsigned(Wd) * unsigned(sizeof(IO)))));
            ACH_UNROLL
            for (int r = 0; r < 2; ++r) buf_store4(orow, off_hb[r], orow_off, H16<IO>::pack(hc[r][0], hc[r][1]));
#else
            char* orow = reinterpret_cast<char*>(out_b) + wave_uniform(int(unsigned(row_st ? ro : r0) * unsigned(Wd) * unsigned(sizeof(IO))));
            ACH_UNROLL
            for (int r = 0; r < 2; ++r)
                if (row_st && st_h[r]) *reinterpret_cast<uint32_t*>(orow + off_hb[r]) = H16<IO>::pack(hc[r][0], hc[r][1]);
#endif
            ACH_UNROLL
            for (int r = 0; r < NR; ++r) {
                const f32x2 wv0 = {wh[0][r], wh[0][r]}, wv1 = {wh[1][r], wh[1][r]}, wv2 = {wh[2][r], wh[2][r]}, wv3 = {wh[3][r], wh[3][r]}, wv4 = {wh[4][r], wh[4][r]},
                            wv5 = {wh[5][r], wh[5][r]}, wv6 = {wh[6][r], wh[6][r]}, wv7 = {wh[7][r], wh[7][r]}, wv8 = {wh[8][r], wh[8][r]};
                const f32x2 slh = wv0 * hm[r] + wv3 * hc[r] + wv6 * hp[r];               // per column: its contribution to column + 1
                const f32x2 srh = wv2 * hm[r] + wv5 * hc[r] + wv8 * hp[r];               // ... to column - 1
                const f32x2 sch = f32x2{bdh[r], bdh[r]} + wv1 * hm[r] + wv4 * hc[r] + wv7 * hp[r];
                const float oa = add_from_left(sch[0] + srh[1], slh[1]);                 // column A: own B's right taps + lane n-1's B left taps
                const float ob = add_from_right(sch[1] + slh[0], srh[0]);                // column B: own A's left taps + lane n+1's A right taps
#if ACH_DH2_BUFST
                buf_store4(orow, off_db[r], orow_off, relu_packed16(H16<IO>::pack(oa, ob)));
#else
                if (row_st && st_d[r]) *reinterpret_cast<uint32_t*>(orow + off_db[r]) = relu_packed16(H16<IO>::pack(oa, ob));
#endif
            }
        }
    };
    ACH_NO_UNROLL
    for (int i = r0 - 2; i <= r1 + 1; i += 3) {
        step(i, w0, w1, w2, v0, v1, v2);
        step(i + 1, w1, w2, w0, v1, v2, v0);
        step(i + 2, w2, w0, w1, v2, v0, v1);
    }
}

// ------------------------------------------------------------------------------------------ the other decoder levels, same walk
// One decoder level at full resolution — x1 = relu(bilinear x2 (t)), x2 = relu(dw3x3(x1) + b), y = [x1 | x2] (NHWC) — as the row-walking
// kernel above without the head: one 3x3 window, so 14 of a strip's 16 columns produce outputs.  NP = channel pairs per lane:
// the level's Cg = 8 NP channels are dealt to the four lane groups (Cg = 16 / 24 / 32).  Replaces upghost_kernel's LDS tile (bf16 engine).
struct UpGhostRowsParams {
    const void* Tq; long ldt;                 // t at low resolution [B,h,w,ldt] bf16
    void* Y; long ldy;                        // [B,2h,2w,2Cg]
    const float* Wdw; const float* bdw;       // [9][Cg], [Cg]
    int B, h, w, Cg;
    float sx;
    int band_rows, bands, strips;
};
constexpr int UGR_VALID = 14;

template <class T, int NP>
__global__ __launch_bounds__(64, (NP <= 2 ? 4 : (NP == 3 ? 3 : 2))) void upghost_rows_kernel(const UpGhostRowsParams p, const DecHeadRow* __restrict__ rows) { f16_sat_mode<T>();
    constexpr int CL = 2 * NP;                                   // channels per lane
    const int H = 2 * p.h, Wd = 2 * p.w;
    const unsigned u = xcd_block(blockIdx.x, gridDim.x);
    const int strip = int(u % unsigned(p.strips)), band = int((u / unsigned(p.strips)) % unsigned(p.bands));
    const long b = long(u / (unsigned(p.strips) * unsigned(p.bands)));
    const int lane = int(threadIdx.x) & 63, n = lane & 15, g = lane >> 4;
    const int x = strip * UGR_VALID - 1 + n;
    const bool in_x = x >= 0 && x < Wd;
    const bool writer = in_x && n >= 1 && n < 1 + UGR_VALID;
    const int cx = x < 0 ? 0 : (x >= Wd ? Wd - 1 : x);
    const float fx = p.sx * float(cx);
    int x0 = int(fx);
    if (x0 > p.w - 1) x0 = p.w - 1;
    const int dx = x0 < p.w - 1 ? 1 : 0;
    const float lx = fx - float(x0);
    const float wx0 = in_x ? 1.f - lx : 0.f, wx1 = in_x ? lx : 0.f;
    const T* Tq = static_cast<const T*>(p.Tq) + b * p.h * long(p.w) * p.ldt;
    const unsigned o0 = unsigned(x0 * int(p.ldt) + CL * g), o1 = unsigned((x0 + dx) * int(p.ldt) + CL * g);
    const int rowp = p.w * int(p.ldt);
    f32x2 wl[9][NP], bl[NP];
    ACH_UNROLL
    for (int k = 0; k < 9; ++k) { ACH_UNROLL for (int q = 0; q < NP; ++q) wl[k][q] = f32x2{p.Wdw[k * p.Cg + CL * g + 2 * q], p.Wdw[k * p.Cg + CL * g + 2 * q + 1]}; }
    ACH_UNROLL
    for (int q = 0; q < NP; ++q) bl[q] = f32x2{p.bdw[CL * g + 2 * q], p.bdw[CL * g + 2 * q + 1]};
    T* Yb = static_cast<T*>(p.Y) + b * long(H) * Wd * p.ldy;                       // (uniform)
    const unsigned yo = unsigned(in_x ? x : 0) * unsigned(p.ldy) + unsigned(CL * g);
    const int r0 = band * p.band_rows, r1 = (r0 + p.band_rows < H) ? r0 + p.band_rows : H;
    auto load_raw = [&](int r, uint32_t (&raw)[2][NP]) {
        const int rr = r < 0 ? 0 : (r > p.h - 1 ? p.h - 1 : r);
        const T* q = Tq + long(rr) * rowp;
        ACH_UNROLL
        for (int i = 0; i < NP; ++i) { raw[0][i] = reinterpret_cast<const uint32_t*>(q + o0)[i]; raw[1][i] = reinterpret_cast<const uint32_t*>(q + o1)[i]; }
    };
    auto unpack = [&](const uint32_t (&raw)[2][NP], f32x2 (&o)[2][NP]) {
        ACH_UNROLL
        for (int c = 0; c < 2; ++c) { ACH_UNROLL for (int i = 0; i < NP; ++i) o[c][i] = f32x2{H16<T>::lo(raw[c][i]), H16<T>::hi(raw[c][i])}; }
    };
    const int i_first = r0 - 1 < 0 ? 0 : r0 - 1;
    int cy = rows[i_first].y0;
    f32x2 ta[2][NP], tb[2][NP];
    uint32_t tn[2][NP];
    { uint32_t raw[2][NP]; load_raw(cy, raw); unpack(raw, ta); load_raw(cy + 1, raw); unpack(raw, tb); load_raw(cy + 2, tn); }
    const f32x2 zero2 = {0.f, 0.f};
    f32x2 w0[NP], w1[NP], w2[NP];
    ACH_UNROLL
    for (int q = 0; q < NP; ++q) { w0[q] = zero2; w1[q] = zero2; w2[q] = zero2; }

    // one step: x1 row i into xp; output row i-1 from x1 rows xm, xc, xp (the caller rotates the window slots)
    auto step = [&](const int i, f32x2 (&xm)[NP], f32x2 (&xc)[NP], f32x2 (&xp)[NP]) {
        {
            const bool row_ok = i >= 0 && i < H;
            const DecHeadRow rg = rows[row_ok ? i : 0];
            if (row_ok && rg.y0 > cy) {
                ACH_UNROLL
                for (int c = 0; c < 2; ++c) { ACH_UNROLL for (int q = 0; q < NP; ++q) ta[c][q] = tb[c][q]; }
                unpack(tn, tb);
                ++cy;
                load_raw(cy + 2, tn);
            }
            const float ly = row_ok ? rg.ly : 0.f, hy = row_ok ? 1.f - rg.ly : 0.f;
            const float w00 = hy * wx0, w01 = hy * wx1, w10 = ly * wx0, w11 = ly * wx1;
            ACH_UNROLL
            for (int q = 0; q < NP; ++q) {
                const f32x2 v = w00 * ta[0][q] + w01 * ta[1][q] + w10 * tb[0][q] + w11 * tb[1][q];
                xp[q] = f32x2{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f};
            }
        }
        const int ro = i - 1;
        float x2[CL];
        {
            float c4[4], l4[4], r4[4], o4[4];
            ACH_UNROLL
            for (int q0 = 0; q0 < NP; q0 += 2) {                   // two channel pairs per group of shifted adds (the last group of NP = 3 is half used)
                ACH_UNROLL
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = q0 + qq < NP ? q0 + qq : NP - 1;
                    const f32x2 sl = wl[0][q] * xm[q] + wl[3][q] * xc[q] + wl[6][q] * xp[q];
                    const f32x2 sr = wl[2][q] * xm[q] + wl[5][q] * xc[q] + wl[8][q] * xp[q];
                    const f32x2 sc = bl[q] + wl[1][q] * xm[q] + wl[4][q] * xc[q] + wl[7][q] * xp[q];
                    c4[2 * qq] = sc[0]; c4[2 * qq + 1] = sc[1]; l4[2 * qq] = sl[0]; l4[2 * qq + 1] = sl[1]; r4[2 * qq] = sr[0]; r4[2 * qq + 1] = sr[1];
                }
                combine4(c4, l4, r4, o4);
                ACH_UNROLL
                for (int e = 0; e < 4; ++e) if (2 * q0 + e < CL) x2[2 * q0 + e] = relu_raw(o4[e]);
            }
        }
        if (ro >= r0 && ro < r1 && writer) {
            T* yrow = Yb + long(ro) * Wd * p.ldy + yo;
            uint32_t a[NP], c[NP];
            ACH_UNROLL
            for (int q = 0; q < NP; ++q) { a[q] = H16<T>::pack(xc[q][0], xc[q][1]); c[q] = H16<T>::pack(x2[2 * q], x2[2 * q + 1]); }
            ACH_UNROLL
            for (int q = 0; q < NP; ++q) { reinterpret_cast<uint32_t*>(yrow)[q] = a[q]; reinterpret_cast<uint32_t*>(yrow + p.Cg)[q] = c[q]; }
        }
    };
    ACH_NO_UNROLL
    for (int i = r0 - 1; i <= r1; i += 3) {
        step(i, w0, w1, w2);
        step(i + 1, w1, w2, w0);
        step(i + 2, w2, w0, w1);
    }
}

}  // namespace ach
