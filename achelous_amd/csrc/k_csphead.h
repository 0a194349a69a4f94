// k_csphead.h — last decoder level + segmentation head of the CSP-Dual-FPN as ONE row-walking kernel (16-bit engines; round 5).
//
//   x   = bilinear x2 ( u ),   u = relu(BN(conv1x1(prev)))                                  neck/cspdualfpn.py:27-39  (Upsample; u at LOW resolution)
//   a   = silu( bilinear x2 ( v ) ),  v = BN(conv1x1(u))                                     :42-56  Bottleneck.conv1 — a 1x1 conv + folded BN is affine with
//                                                                                             weights that the interpolation's (which sum to 1) commute with
//   y   = x + relu(BN(conv3x3(a)))                      16 -> 32 channels, dense 3x3         Bottleneck.conv2 + shortcut (in == out)
//   h   = silu(BN(conv1x1(y)))                          32 -> hid = num_class / 2            head Bottleneck.conv1        (:190, :178)
//   out = relu(BN(conv3x3(h)))                          hid -> num_class, NCHW               head Bottleneck.conv2 (no shortcut: in != out)
//
// Layer by layer this was five launches and 2.9 GB of HBM traffic per decoder at batch 64 (the 320 x 320 x 32 tensors x, y written and read back, a, h in
// between: 1.14 ms of a 1.56 ms decoder, profiles/r05_ops_en_s0_cdf.json).  Here nothing of full resolution but the NCHW output ever exists in memory:
//   * a WAVE owns a strip of 16 columns (12 produce outputs: the two cascaded 3x3 windows need two columns of halo per side) and walks down a band of rows, as
//     the Ghost-FPN head does (k_dechead.h); lane (n, g) = column n, lane group g.
//   * both dense 3x3 convs are MFMAs whose B fragments are built from the lane's rolling three-row windows with DPP row shifts (the x neighbours are the
//     neighbouring lanes of the 16-lane row; `row_mask` lets the four lane groups — which hold DIFFERENT taps of a k-step — take different shifts of different rows):
//       conv2  K = 9 taps x 16 channels = 144 -> 5 k-steps of 32: k-step s, group g = tap 2s + (g >> 1), channels 8 (g & 1) .. + 7; two 16-row tiles = 32 outputs;
//       head   K = 9 taps x 4 channels  =  36 -> 2 k-steps:       k-step s, group g = taps 8s + 2g, 8s + 2g + 1, four channels each.
//   * D row 4g + r of conv2's tile t is channel 16t + 4g + r, so a lane ends up with y channels {4g .. 4g+3, 16+4g .. 16+4g+3} of its column — which, with the
//     head conv1's k index permuted the same way on the host, IS its B fragment: one more MFMA.  That conv's <= 4 output rows are REPLICATED four times in A, so
//     every lane group receives all hidden channels of its column (what the last conv's taps need) without a cross-group shuffle.
//   * the bilinear source rows are blended along x once when they arrive and along y per output row (k_dechead.h, HFIRST).
// HEAD = false: the same walk for a decoder level that is NOT the last one (the 160 x 160 level of EN-S0): it stops at y, which leaves as NHWC rows of the storage
// type — a lane's eight channels as two 8-byte stores — and a strip has 14 valid columns (one 3x3 window: one column of halo per side).
// Round 5, second version (399 / 417 -> see DESIGN 4.20): a lane interpolates and activates only FOUR of its eight `a` channels — lanes (n, g) and (n, g ^ 2) need the
// same eight — and the halves are exchanged with one cross-half shuffle per dword; x stays in fp32 registers until it is added; every store is a range-checked buffer
// store at a per-lane offset computed once (no per-row branch); rows beyond the head's hidden width need no test (their A rows and biases are zero: silu(0) = 0).
// Every matrix instruction of the walk is the 16x16x16 PAIR form (mfma16_pair): these are long-lived MFMA waves.
// Numerics: u, v are stored in the engine's 16-bit type (as the layer-wise plan stores them); x, a, y, h are fp32 in registers and rounded once, where an MFMA
// consumes them (the layer-wise plan rounds each to the storage type in HBM).
#pragma once
#include "ach_platform.h"
#include "k_dechead.h"

namespace ach {

struct CspHeadParams {
    const void* UV; long lduv;                // low resolution [B, h, w, lduv]: channels 0..31 = u, 32..47 = v (storage type)
    void* out;                                // HEAD: NCHW [B, nc, 2h, 2w] (caller's type); otherwise NHWC [B, 2h, 2w, ldo] (storage type), 32 channels
    long ldo;
    const uint4* W2; const float* b2;         // [5][2][64] A fragments of conv2 (see above), bias[32]
    const uint4* Wh1; const float* bh1;       // [64] head conv1 (rows replicated, k permuted), bias[4] (zero beyond hid)
    const uint4* Wh2; const float* bh2;       // [2][64] head conv2, bias[16] (zero beyond nc)
    int B, h, w, hid, nc;
    float sy, sx;
    int band_rows, bands, strips;
};
constexpr int CSPH_VALID = 12, CSPL_VALID = 14;

// `src` of the lane one column to the left (row_shr:1: CTRL 0x111) / to the right (row_shl:1: 0x101) / of the lane itself (quad_perm [0,1,2,3]: 0xE4) — written only in
// the 16-lane rows whose bit is set in RMASK (row = lane group g), `old` elsewhere; a lane without a source (the strip's edge) receives 0.
#if defined(ACH_HOSTEMU)
template <int CTRL, int RMASK> __device__ inline uint32_t dpp_sel(uint32_t old, uint32_t src) {
    const int l = int(threadIdx.x) & 63, n = l & 15, row = l >> 4;
    const int from = CTRL == 0x111 ? l - 1 : (CTRL == 0x101 ? l + 1 : l);
    const bool has = CTRL == 0x111 ? n > 0 : (CTRL == 0x101 ? n < 15 : true);
    const uint32_t v = uint32_t(__shfl(int(src), has ? from : l));
    return ((RMASK >> row) & 1) ? (has ? v : 0u) : old;
}
#else
template <int CTRL, int RMASK> __device__ __forceinline__ uint32_t dpp_sel(uint32_t old, uint32_t src) {
    return uint32_t(__builtin_amdgcn_update_dpp(int(old), int(src), CTRL, RMASK, 0xf, true));
}
#endif
// the value of tap (dy, dx) of a three-row window {m, c, p} (rows -1, 0, +1) for the lane groups in RMASK
template <int TAP, int RMASK> __device__ __forceinline__ uint32_t csph_tap(uint32_t old, uint32_t m, uint32_t c, uint32_t p) {
    constexpr int dy = TAP / 3 - 1, dx = TAP % 3 - 1;
    constexpr int CTRL = dx < 0 ? 0x111 : (dx > 0 ? 0x101 : 0xE4);
    return dpp_sel<CTRL, RMASK>(old, dy < 0 ? m : (dy > 0 ? p : c));
}

template <class T, class IO, bool HEAD>
__global__ __launch_bounds__(64, 2) void csp_head_rows_kernel(const CspHeadParams p, const DecHeadRow* __restrict__ rows) { f16_sat_mode<T>();
    constexpr int VALID = HEAD ? CSPH_VALID : CSPL_VALID, HALO = HEAD ? 2 : 1;
    const int H = 2 * p.h, Wd = 2 * p.w;
    const unsigned u_ = xcd_block(blockIdx.x, gridDim.x);
    const int strip = int(u_ % unsigned(p.strips)), band = int((u_ / unsigned(p.strips)) % unsigned(p.bands));
    const long b = long(u_ / (unsigned(p.strips) * unsigned(p.bands)));
    const int lane = int(threadIdx.x) & 63, n = lane & 15, g = lane >> 4;
    const bool hi = g >= 2;
    const int x = strip * VALID - HALO + n;
    const bool in_x = x >= 0 && x < Wd;
    const bool writer = in_x && n >= HALO && n < HALO + VALID;
    // ---- bilinear geometry along x (fixed for the band)
    const int cx = x < 0 ? 0 : (x >= Wd ? Wd - 1 : x);
    const float fx = p.sx * float(cx);
    int x0 = int(fx);
    if (x0 > p.w - 1) x0 = p.w - 1;
    const int dxs = x0 < p.w - 1 ? 1 : 0;
    const float lx = fx - float(x0);
    const float wx0 = in_x ? 1.f - lx : 0.f, wx1 = in_x ? lx : 0.f;        // a column outside the map: zero — the convs' zero padding (silu(0) = 0)
    const char* UVb = reinterpret_cast<const char*>(static_cast<const T*>(p.UV) + b * p.h * long(p.w) * p.lduv);
    const unsigned rowpb = unsigned(p.w) * unsigned(p.lduv) * unsigned(sizeof(T));
    const unsigned c0 = unsigned(x0 * int(p.lduv)) * unsigned(sizeof(T)), c1 = unsigned((x0 + dxs) * int(p.lduv)) * unsigned(sizeof(T));
    // this lane's channels: u {4g .. 4g+3} and {16+4g ..}; v: FOUR of the eight channels 8 (g & 1) .. + 7 its fragments hold — the low half of the wave (g < 2)
    // takes the first four, lane ^ 32 the other four
    unsigned oua0 = c0 + unsigned(4 * g) * unsigned(sizeof(T)), oua1 = c1 + unsigned(4 * g) * unsigned(sizeof(T));
    unsigned oub0 = c0 + unsigned(16 + 4 * g) * unsigned(sizeof(T)), oub1 = c1 + unsigned(16 + 4 * g) * unsigned(sizeof(T));
    unsigned ov0 = c0 + unsigned(32 + 8 * (g & 1) + 4 * (g >> 1)) * unsigned(sizeof(T)), ov1 = c1 + unsigned(32 + 8 * (g & 1) + 4 * (g >> 1)) * unsigned(sizeof(T));
    // ---- weights and biases
    uint4 w2[5][2];
    ACH_UNROLL
    for (int s = 0; s < 5; ++s) { w2[s][0] = p.W2[(s * 2) * 64 + lane]; w2[s][1] = p.W2[(s * 2 + 1) * 64 + lane]; }
    uint4 wh1 = make_uint4(0u, 0u, 0u, 0u), wh2a = wh1, wh2b = wh1;
    float b2a[4], b2b[4], bh1[4] = {0.f, 0.f, 0.f, 0.f}, bo[4] = {0.f, 0.f, 0.f, 0.f};
    ACH_UNROLL
    for (int r = 0; r < 4; ++r) { b2a[r] = p.b2[4 * g + r]; b2b[r] = p.b2[16 + 4 * g + r]; }
    if (HEAD) {
        wh1 = p.Wh1[lane]; wh2a = p.Wh2[lane]; wh2b = p.Wh2[64 + lane];
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { bh1[r] = p.bh1[r]; bo[r] = p.bh2[4 * g + r]; }
    }
    const long HW = long(H) * Wd;
    const int r0 = band * p.band_rows, r1 = (r0 + p.band_rows < H) ? r0 + p.band_rows : H;
    // stores: per-lane byte offsets inside the sample's output computed once, BUF_OOB where the lane has nothing to store; a row outside the band stores through a
    // zero-length resource (k_dechead.h)
    char* out_b = HEAD ? reinterpret_cast<char*>(static_cast<IO*>(p.out) + b * p.nc * HW) : reinterpret_cast<char*>(static_cast<T*>(p.out) + b * HW * p.ldo);
    const unsigned out_bytes = HEAD ? unsigned(p.nc) * unsigned(HW) * unsigned(sizeof(IO)) : unsigned(HW) * unsigned(p.ldo) * unsigned(sizeof(T));
    unsigned off_o[4], off_y[2];
    ACH_UNROLL
    for (int r = 0; r < 4; ++r) off_o[r] = (HEAD && writer && 4 * g + r < p.nc) ? (unsigned(4 * g + r) * unsigned(HW) + unsigned(in_x ? x : 0)) * unsigned(sizeof(IO)) : BUF_OOB;
    off_y[0] = (!HEAD && writer) ? (unsigned(x) * unsigned(p.ldo) + unsigned(4 * g)) * unsigned(sizeof(T)) : BUF_OOB;
    off_y[1] = (!HEAD && writer) ? (unsigned(x) * unsigned(p.ldo) + unsigned(16 + 4 * g)) * unsigned(sizeof(T)) : BUF_OOB;

    // ---- source rows: raw (two source columns x {u 4g.., u 16+4g.., v four channels}) and blended along x (fp32 pairs: u 8 values, v 4 values)
    struct Raw { uint2 ua[2], ub[2], v[2]; };
    auto load_raw = [&](int r, Raw& q) {
        const int rr = r < 0 ? 0 : (r > p.h - 1 ? p.h - 1 : r);
        const char* base = UVb + wave_uniform(int(unsigned(rr) * rowpb));               // (below 2 GiB: plan-time check)
        q.ua[0] = *reinterpret_cast<const uint2*>(base + local_offset(oua0)); q.ua[1] = *reinterpret_cast<const uint2*>(base + local_offset(oua1));
        q.ub[0] = *reinterpret_cast<const uint2*>(base + local_offset(oub0)); q.ub[1] = *reinterpret_cast<const uint2*>(base + local_offset(oub1));
        q.v[0] = *reinterpret_cast<const uint2*>(base + local_offset(ov0)); q.v[1] = *reinterpret_cast<const uint2*>(base + local_offset(ov1));
    };
    struct Blend { f32x2 u[4], v[2]; };
    auto pr = [&](uint32_t w) { return f32x2{H16<T>::lo(w), H16<T>::hi(w)}; };
    auto hblend = [&](const Raw& q, Blend& o) {
        o.u[0] = wx0 * pr(q.ua[0].x) + wx1 * pr(q.ua[1].x); o.u[1] = wx0 * pr(q.ua[0].y) + wx1 * pr(q.ua[1].y);
        o.u[2] = wx0 * pr(q.ub[0].x) + wx1 * pr(q.ub[1].x); o.u[3] = wx0 * pr(q.ub[0].y) + wx1 * pr(q.ub[1].y);
        o.v[0] = wx0 * pr(q.v[0].x) + wx1 * pr(q.v[1].x); o.v[1] = wx0 * pr(q.v[0].y) + wx1 * pr(q.v[1].y);
    };
    const int i_lo = HEAD ? r0 - 2 : r0 - 1;                       // first row whose `a` is needed
    const int i_first = i_lo < 0 ? 0 : i_lo;
    int cy = rows[i_first].y0;
    Blend ha, hb;                           // the two live source rows; `par` says which holds the older one
    Raw tn;
    int par = 0;
    { Raw q; load_raw(cy, q); hblend(q, ha); load_raw(cy + 1, q); hblend(q, hb); load_raw(cy + 2, tn); }
    if (HEAD) { const BufRsrc none = make_buf(out_b, 0u); ACH_UNROLL for (int r = 0; r < 4; ++r) buf_store2(none, BUF_OOB, 0u, 0u); }      // (as many dropped stores behind the first loads as a step issues: k_dechead.h)
    else { const BufRsrc none = make_buf(out_b, 0u); buf_store8(none, BUF_OOB, 0u, 0u, 0u); buf_store8(none, BUF_OOB, 0u, 0u, 0u); }

    // rolling windows: a (8 channels packed: 4 dwords), h (4 channels packed: 2 dwords); x (8 channels, fp32) from the row it is interpolated in to the next
    struct W4 { uint32_t d[4]; };
    struct W2_ { uint32_t d[2]; };
    struct X8 { f32x2 v[4]; };
    const f32x2 z2 = {0.f, 0.f};
    W4 a0{{0u, 0u, 0u, 0u}}, a1 = a0, a2 = a0;
    X8 xq0{{z2, z2, z2, z2}}, xq1 = xq0, xq2 = xq0;
    W2_ h0{{0u, 0u}}, h1 = h0, h2 = h0;

    auto step = [&](const int i, W4& am, W4& ac, W4& ap, W2_& hm, W2_& hc, W2_& hp, X8& xprev, X8& xcur) {
        // ---- A: a and x of row i
        {
            const bool row_ok = i >= 0 && i < H;
            DecHeadRow rg = rows[row_ok ? i : 0];
            rg.y0 = wave_uniform(rg.y0); rg.ly = __int_as_float(wave_uniform(__float_as_int(rg.ly)));
            if (row_ok && rg.y0 > cy) {
                if (par == 0) hblend(tn, ha); else hblend(tn, hb);
                par ^= 1;
                ++cy;
            }
            load_raw(cy + 2, tn);          // every step (a repeat when nothing arrived): a fixed number of memory operations per step (k_dechead.h)
            const float ly = row_ok ? rg.ly : 0.f, hy = row_ok ? 1.f - rg.ly : 0.f;
            const float wo = par == 0 ? hy : ly, wn = par == 0 ? ly : hy;          // weights of ha / hb
            ACH_UNROLL
            for (int q = 0; q < 4; ++q) xcur.v[q] = wo * ha.u[q] + wn * hb.u[q];
            uint32_t mine[2], oth[2];
            ACH_UNROLL
            for (int q = 0; q < 2; ++q) {
                const f32x2 av = wo * ha.v[q] + wn * hb.v[q];
                mine[q] = H16<T>::pack(av[0] * sigmoidf_(av[0]), av[1] * sigmoidf_(av[1]));
            }
            oth[0] = uint32_t(__shfl_xor(int(mine[0]), 32)); oth[1] = uint32_t(__shfl_xor(int(mine[1]), 32));
            ap.d[0] = hi ? oth[0] : mine[0]; ap.d[1] = hi ? oth[1] : mine[1];
            ap.d[2] = hi ? mine[0] : oth[0]; ap.d[3] = hi ? mine[1] : oth[1];
        }
        // ---- B: y (and h) of row i - 1
        {
            const int rb = i - 1;
            f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#define CSPH_KSTEP(S)                                                                                                                      \
            {                                                                                                                              \
                uint4 f;                                                                                                                   \
                f.x = csph_tap<2 * S, 0x3>(0u, am.d[0], ac.d[0], ap.d[0]); f.y = csph_tap<2 * S, 0x3>(0u, am.d[1], ac.d[1], ap.d[1]);     \
                f.z = csph_tap<2 * S, 0x3>(0u, am.d[2], ac.d[2], ap.d[2]); f.w = csph_tap<2 * S, 0x3>(0u, am.d[3], ac.d[3], ap.d[3]);     \
                if (2 * S + 1 < 9) {                                                                                                       \
                    constexpr int TB = 2 * S + 1 < 9 ? 2 * S + 1 : 0;                                                                      \
                    f.x = csph_tap<TB, 0xC>(f.x, am.d[0], ac.d[0], ap.d[0]); f.y = csph_tap<TB, 0xC>(f.y, am.d[1], ac.d[1], ap.d[1]);     \
                    f.z = csph_tap<TB, 0xC>(f.z, am.d[2], ac.d[2], ap.d[2]); f.w = csph_tap<TB, 0xC>(f.w, am.d[3], ac.d[3], ap.d[3]);     \
                }                                                                                                                          \
                mfma16_pair<T>(w2[S][0], f, ca); mfma16_pair<T>(w2[S][1], f, cb);                                                                    \
            }
            CSPH_KSTEP(0) CSPH_KSTEP(1) CSPH_KSTEP(2) CSPH_KSTEP(3) CSPH_KSTEP(4)
#undef CSPH_KSTEP
            float y[8];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) {
                const float ra = ca[r] + b2a[r], rbv = cb[r] + b2b[r];
                y[r] = xprev.v[r >> 1][r & 1] + (ra > 0.f ? ra : 0.f);
                y[4 + r] = xprev.v[2 + (r >> 1)][r & 1] + (rbv > 0.f ? rbv : 0.f);
            }
            const uint4 yf = make_uint4(H16<T>::pack(y[0], y[1]), H16<T>::pack(y[2], y[3]), H16<T>::pack(y[4], y[5]), H16<T>::pack(y[6], y[7]));
            if (HEAD) {
                f32x4 hh = {0.f, 0.f, 0.f, 0.f};
                mfma16_pair<T>(wh1, yf, hh);
                const bool live = rb >= 0 && rb < H && in_x;                 // outside the map h is the last conv's zero padding
                float hv[4];
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) {                                // (rows beyond the hidden width: zero weights and bias -> silu(0) = 0)
                    const float t = hh[r] + bh1[r];
                    hv[r] = live ? t * sigmoidf_(t) : 0.f;
                }
                hp.d[0] = H16<T>::pack(hv[0], hv[1]); hp.d[1] = H16<T>::pack(hv[2], hv[3]);
            } else {
                const bool row_st = rb >= r0 && rb < r1;
                const BufRsrc orow = make_buf(out_b, row_st ? out_bytes : 0u);
                const unsigned soff = unsigned(wave_uniform(int(unsigned(row_st ? rb : r0) * unsigned(Wd) * unsigned(p.ldo) * unsigned(sizeof(T)))));
                buf_store8(orow, off_y[0], soff, yf.x, yf.y);
                buf_store8(orow, off_y[1], soff, yf.z, yf.w);
            }
        }
        // ---- C: output row i - 2
        if (HEAD) {
            const int ro = i - 2;
            uint4 f0, f1;
            // k-step 0: group g holds taps 2g (dwords 0, 1) and 2g + 1 (dwords 2, 3); k-step 1: tap 8 in group 0
            f0.x = csph_tap<0, 0x1>(0u, hm.d[0], hc.d[0], hp.d[0]); f0.x = csph_tap<2, 0x2>(f0.x, hm.d[0], hc.d[0], hp.d[0]);
            f0.x = csph_tap<4, 0x4>(f0.x, hm.d[0], hc.d[0], hp.d[0]); f0.x = csph_tap<6, 0x8>(f0.x, hm.d[0], hc.d[0], hp.d[0]);
            f0.y = csph_tap<0, 0x1>(0u, hm.d[1], hc.d[1], hp.d[1]); f0.y = csph_tap<2, 0x2>(f0.y, hm.d[1], hc.d[1], hp.d[1]);
            f0.y = csph_tap<4, 0x4>(f0.y, hm.d[1], hc.d[1], hp.d[1]); f0.y = csph_tap<6, 0x8>(f0.y, hm.d[1], hc.d[1], hp.d[1]);
            f0.z = csph_tap<1, 0x1>(0u, hm.d[0], hc.d[0], hp.d[0]); f0.z = csph_tap<3, 0x2>(f0.z, hm.d[0], hc.d[0], hp.d[0]);
            f0.z = csph_tap<5, 0x4>(f0.z, hm.d[0], hc.d[0], hp.d[0]); f0.z = csph_tap<7, 0x8>(f0.z, hm.d[0], hc.d[0], hp.d[0]);
            f0.w = csph_tap<1, 0x1>(0u, hm.d[1], hc.d[1], hp.d[1]); f0.w = csph_tap<3, 0x2>(f0.w, hm.d[1], hc.d[1], hp.d[1]);
            f0.w = csph_tap<5, 0x4>(f0.w, hm.d[1], hc.d[1], hp.d[1]); f0.w = csph_tap<7, 0x8>(f0.w, hm.d[1], hc.d[1], hp.d[1]);
            f1.x = csph_tap<8, 0x1>(0u, hm.d[0], hc.d[0], hp.d[0]); f1.y = csph_tap<8, 0x1>(0u, hm.d[1], hc.d[1], hp.d[1]);
            f1.z = 0u; f1.w = 0u;
            f32x4 oc = {0.f, 0.f, 0.f, 0.f};
            mfma16_pair<T>(wh2a, f0, oc); mfma16_pair<T>(wh2b, f1, oc);
            const bool row_st = ro >= r0 && ro < r1;
            const BufRsrc orow = make_buf(out_b, row_st ? out_bytes : 0u);
            const unsigned soff = unsigned(wave_uniform(int(unsigned(row_st ? ro : r0) * unsigned(Wd) * unsigned(sizeof(IO)))));
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) {
                const float v = oc[r] + bo[r];
                buf_store2(orow, off_o[r], soff, H16<IO>::pack(v > 0.f ? v : 0.f, 0.f));
            }
        }
    };
    ACH_NO_UNROLL
    for (int i = i_lo; i <= (HEAD ? r1 + 1 : r1); i += 3) {
        step(i, a0, a1, a2, h0, h1, h2, xq0, xq1);
        step(i + 1, a1, a2, a0, h1, h2, h0, xq1, xq2);          // (up to two steps beyond the band: their stores are masked)
        step(i + 2, a2, a0, a1, h2, h0, h1, xq2, xq0);
    }
}

}  // namespace ach
