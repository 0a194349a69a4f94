// k_mlp.h — one kernel for the whole pointwise half of an EdgeNeXt block:
//
//     y = r + W2 · act( W1 · LN( dw_kxk(x) + b_dw ) + b1 ) + b2
//
// (ConvEncoder, edgenext_modules/conv_encoder.py:19-32: depthwise conv -> LayerNorm -> Linear(d,4d) -> GELU ->
//  Linear(4d,d) -> layer scale -> + input;  the MLP tail of SDTAEncoder, sdta_encoder.py:68-74, is the same thing
//  without the depthwise conv and with a residual taken from a different tensor.)
// LayerNorm affine is folded into W1/b1 and the layer scale into W2/b2 on the host.  With ln = 0 and no residual the same
// kernel is a plain two-layer 1x1 chain  y = W2 · act(W1 x + b1) + b2  (the low-resolution pair of every decoder level:
// upsample conv + BN + ReLU followed by the Ghost primary conv + BN, neck/ghostdualfpn.py:175-197).
//
// Layer by layer this was three launches and a round trip of the 4d-wide hidden tensor through HBM (the largest
// activation of the backbone).  Here the hidden activations never leave registers: with weights as the MFMA *A* operand
// a lane's accumulators of one PAIR of 16-channel output tiles are, after the channel permutation chosen for the wide
// stores (chunk_channel in k_gemm.h), exactly the 8 consecutive hidden channels  j*32 + g*8 .. +7  of pixel (lane & 15)
// — which is the *B* operand fragment of k-step j of the second GEMM.  So GEMM1 -> bias -> GELU -> GEMM2 chains through
// registers with no LDS shuffle.  (fp32 storage: the same 8 values form two 4-wide k-steps; the host packs W2 with the
// matching order of hidden channels, see mlp_hidden_channel.)
//
// A wave owns 16 pixels.  Large maps (SPLIT = false): the 4 waves of a workgroup take 4 different pixel tiles.  Small
// maps (SPLIT = true; 20x20 and 10x10 at batch 64 are only 1600 / 400 tiles): the 4 waves share ONE tile — each computes
// the depthwise conv for a quarter of the channel k-steps and a quarter of the hidden chunks, inputs are exchanged and
// the partial outputs summed through LDS — four times the parallelism, a quarter of the dependent-latency chain.
#pragma once
#include <type_traits>
#include "ach_platform.h"
#include "k_gemm.h"

#ifndef ACH_MLP_DW_LDS
#define ACH_MLP_DW_LDS 1                  // one-tile-per-wave kernels: depthwise weights staged in LDS once per workgroup (0 = fetched per tap from global memory, round 4)
#endif
#ifndef ACH_MLP_MFMA_PAD
#define ACH_MLP_MFMA_PAD 0
#endif
namespace ach {

// Hidden channel that k index `kappa` of the second GEMM refers to.  bf16 (VEC 8): identity.  fp32 (VEC 4): the 8 values a
// lane group g holds for hidden chunk j (channels j*32 + g*8 + e) are consumed as two k-steps of 16: step 2j takes e = 0..3,
// step 2j+1 takes e = 4..7, and inside a step lane group g / element kj is k index g*4 + kj.
__host__ __device__ __forceinline__ int mlp_hidden_channel(int kappa, int VEC) {
    if (VEC == 8) return kappa;
    const int j = kappa >> 5, rem = kappa & 31, half = rem >> 4, g = (rem & 15) >> 2, kj = rem & 3;
    return j * 32 + g * 8 + half * 4 + kj;
}

struct MlpParams {
    const void* X; long ldx;              // LayerNorm input rows, or the depthwise conv's input map when dw_k > 0 (NHWC)
    const void* R; long ldr;              // residual rows
    void* Y; long ldy;
    const float* Wdw; const float* bdw;   // depthwise weights [dw_k*dw_k][ldc] (fp32, zero padded), bias [ldc]; ldc = k1 * 4 * VEC
    int dw_k, H, W;
    unsigned xbytes;                      // dw_k > 0: bytes of the whole input tensor (< 2^31), for the range-checked tap loads
    const void* W1; const float* b1;      // [hidden][C] packed in NT = 2 chunks with k1 k-steps ; bias padded to 32 * J
    const void* W2; const float* b2;      // [C][hidden] packed as ONE chunk of DT tiles with J * (8 / VEC) k-steps ; bias padded to 16 * DT
    long M; int C, k1, J, act; float ln_eps;
    int ln, Cout;                         // ln = 0: no normalisation (plain two-layer chain); Cout: output width (R may be null)
    int planar_w;                         // chain_kernel only, > 0: Y is written channel-planar per map row, [M / planar_w][Cout][planar_w]
    int dw_even;                          // SPLIT + depthwise: the k1 * dw_k tap ROWS are dealt evenly to the four waves (mlp_inputs), not whole k-steps
    void* Hout; long ldh;                 // chain_kernel only, optional: the hidden activations act(W1 x + b1) are stored too (rows of ldh elements) — a layer pair whose FIRST output is needed as well (k_csphead.h's [u | v])
};

constexpr int MLP_RED_TILES = 4;          // output tiles reduced per LDS round in SPLIT mode
#ifndef ACH_MLP_DEBUG
#define ACH_MLP_DEBUG 0                   // phase timing builds only (profiles/scripts/build_variant.sh): 1 = no depthwise taps, 2 = no hidden chunks, 4 = ReLU for every activation; wrong results
#endif

// depthwise k x k conv (zero padding k/2) of this lane's VEC channels at its pixel; up to 5 taps of a row in flight at a time.
// Taps are fetched through a range-checked buffer resource over the whole activation tensor (ach_platform.h): a column outside
// the map gets an out-of-range offset and comes back as zeros, so a tap costs one add and one load — no clamps, no 64-bit
// address arithmetic, no masking of the packed data (this loop is VALU-issue bound: 25 -> 15 instructions per tap).  Rows outside
// the map are skipped.  The weights are kept in fp32 so that only the activations need unpacking.
// WL (round 5): the weights come from the workgroup's LDS copy `wl` (same [tap][ldc] layout; mlp_kernel stages it once) instead of global memory: per tap a
// lane fetched its 8 fp32 weights with two 16-byte VECTOR loads — the address depends on the lane group — i.e. 100 of the 150 texture-path operations of a
// 5 x 5 / 48-channel tile were weights, on a path the counters put at 62 % busy (profiles/r04_pmc_summary_en_s0.txt, stages.1.0.block).
template <class T, int KS, bool WL = false>
__device__ __forceinline__ void mlp_dw(const MlpParams& p, const BufRsrc& xb, long pix0, int oy, int ox, int k0, float* acc, int ty0 = 0, int ty1 = KS, const float* wl = nullptr) {
    constexpr int VEC = Store<T>::VEC;
    constexpr int TG = KS <= 5 ? KS : (KS + 1) / 2;
    constexpr unsigned ESZ = sizeof(T);
    const int ldc = p.k1 * 4 * VEC;
    const float* wdw = (WL ? wl : p.Wdw) + k0;
    const unsigned pitch = unsigned(p.ldx) * ESZ, rowpitch = unsigned(p.W) * pitch;
    const unsigned base = unsigned(pix0) * pitch + unsigned(k0) * ESZ;
    unsigned coff[KS];
    ACH_UNROLL
    for (int tx = 0; tx < KS; ++tx) {
        const int ix = ox + tx - KS / 2;
        coff[tx] = (ix >= 0 && ix < p.W) ? unsigned(ix) * pitch : BUF_OOB;
    }
    if (ACH_MLP_DEBUG & 1) return;
    for (int ty = ty0; ty < ty1; ++ty) {
        const int iy = oy + ty - KS / 2;
        if (iy < 0 || iy >= p.H) continue;
        const unsigned rowb = base + unsigned(iy) * rowpitch;
        const float* wrow = wdw + long(ty * KS) * ldc;
        ACH_UNROLL
        for (int tx0 = 0; tx0 < KS; tx0 += TG) {
            uint4 xv[TG];
            f32x4 wv[TG][VEC / 4];
            ACH_UNROLL
            for (int i = 0; i < TG; ++i) {
                const int tx = tx0 + i;
                if (tx >= KS) continue;
                xv[i] = buf_load16(xb, rowb + coff[tx]);
                ACH_UNROLL
                for (int q = 0; q < VEC / 4; ++q) wv[i][q] = *reinterpret_cast<const f32x4*>(wrow + long(tx) * ldc + q * 4);
            }
            ACH_UNROLL
            for (int i = 0; i < TG; ++i) {
                const int tx = tx0 + i;
                if (tx >= KS) continue;
                float xf[8];
                frag_unpack<T>(xv[i], xf);
                ACH_UNROLL
                for (int e = 0; e < VEC; ++e) acc[e] += xf[e] * wv[i][e >> 2][e & 3];
            }
        }
    }
}

// Step 1 of mlp_kernel: this wave's input fragments and its partial LayerNorm sums.  KS = 0: plain rows; KS = 3/5/7/9: depthwise conv.
// SPLIT: wave w produces k-steps w, w+4, ... straight into LDS (`xs`); otherwise all k-steps into `xf`.
// SPLIT + dw_even (host: every wave's share spans at most two k-steps, `part` holds 4 waves x 2 slots x 64 lanes x 8 floats): with whole
// k-steps per wave, 5 k-steps (d = 144) are 2 / 1 / 1 / 1 and 3 k-steps (d = 96) 1 / 1 / 1 / 0 — the depthwise phase waits for its slowest wave.
// Dealing the k1 * KS tap ROWS in four contiguous shares makes that 9 / 9 / 9 / 8 rows instead of 14 / 7 / 7 / 7; the owner of a k-step
// (as before: wave s % 4) then sums the shares in wave order (deterministic), adds the bias and forms the fragment.
template <class T, int K1MAX, bool SPLIT, int KS, bool EVEN, bool WL = false>
__device__ __forceinline__ void mlp_inputs(const MlpParams& p, long m, bool valid, int g, int wave, int lane, uint4* xs, uint4* xf, float& s1, float& s2, float* part, const float* wl = nullptr) {
    constexpr int VEC = Store<T>::VEC;
    constexpr int KC = 4 * VEC;
    const T* X = static_cast<const T*>(p.X);
    int oy = 0, ox = 0;
    long pix0 = 0;
    const BufRsrc xb = make_buf(p.X, KS > 0 ? p.xbytes : 0u);
    if (KS > 0) {
        // 32-bit unsigned arithmetic: the depthwise input is below 2 GiB (plan-time check), so is its pixel count — the 64-bit division
        // this replaces was ~100 of the kernel's ~820 VALU instructions per tile
        const unsigned hw = unsigned(p.H) * unsigned(p.W), mu = unsigned(m);
        const unsigned b = mu / hw, rem = mu - b * hw;
        oy = int(rem / unsigned(p.W)); ox = int(rem) - oy * p.W;
        pix0 = long(b) * hw;
    }
    constexpr int NS = SPLIT ? (K1MAX + 3) / 4 : K1MAX;
    constexpr bool even = SPLIT && KS > 0 && EVEN;     // a compile-time variant: as a run-time branch it cost the plain path 116 bytes of spills at DT = 6
    const int U = p.k1 * KS, chunk = (U + 3) / 4;
    if (SPLIT && KS > 0 && even) {
        const int u0 = wave * chunk, u1 = u0 + chunk < U ? u0 + chunk : U, sa = u0 / KS;
        ACH_UNROLL
        for (int slot = 0; slot < 2; ++slot) {
            const int s = sa + slot, k0 = s * KC + g * VEC;
            const int r0 = slot == 0 ? u0 - sa * KS : 0, r1 = u1 - s * KS < KS ? u1 - s * KS : KS;
            float acc[8];
            ACH_UNROLL
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            if (u0 < u1 && s < p.k1 && r0 < r1 && valid && k0 < p.C) mlp_dw<T, (KS > 0 ? KS : 3)>(p, xb, pix0, oy, ox, k0, acc, r0, r1);
            float* dst = part + ((wave * 2 + slot) * 64 + lane) * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __syncthreads();
    }
    ACH_UNROLL
    for (int si = 0; si < NS; ++si) {
        const int s = SPLIT ? wave + 4 * si : si;
        if (s >= p.k1) continue;
        const int k0 = s * KC + g * VEC;
        uint4 frag = make_uint4(0u, 0u, 0u, 0u);
        if (SPLIT && KS > 0 && even) {
            if (valid && k0 < p.C) {
                float acc[8];
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) acc[i] = p.bdw[k0 + i];
                for (int w = 0; w < 4; ++w) {                          // shares of k-step s, in wave order
                    const int a = (w * chunk) / KS;
                    if (w * chunk >= U || (a != s && a + 1 != s)) continue;
                    const float* src = part + ((w * 2 + (s - a)) * 64 + lane) * 8;
                    ACH_UNROLL
                    for (int i = 0; i < VEC; ++i) acc[i] += src[i];
                }
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) { s1 += acc[i]; s2 += acc[i] * acc[i]; }
                frag = frag_pack<T>(acc);
            }
            xs[s * 64 + lane] = frag;
            continue;
        }
        if (valid && k0 < p.C) {
            if (KS == 0) {
                frag = *reinterpret_cast<const uint4*>(X + m * p.ldx + k0);
                float v[8];
                frag_unpack<T>(frag, v);
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
            } else {
                float acc[8];
                if constexpr (WL) {             // bias behind the taps in the LDS copy
                    const float* bl = wl + KS * KS * (p.k1 * 4 * VEC) + k0;
                    ACH_UNROLL
                    for (int q = 0; q < VEC / 4; ++q) { const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + 4 * q); acc[4 * q] = b4[0]; acc[4 * q + 1] = b4[1]; acc[4 * q + 2] = b4[2]; acc[4 * q + 3] = b4[3]; }
                } else {
                    ACH_UNROLL
                    for (int i = 0; i < VEC; ++i) acc[i] = p.bdw[k0 + i];
                }
                mlp_dw<T, (KS > 0 ? KS : 3), WL>(p, xb, pix0, oy, ox, k0, acc, 0, (KS > 0 ? KS : 3), wl);
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) { s1 += acc[i]; s2 += acc[i] * acc[i]; }     // channels >= C: zero weights, zero bias
                frag = frag_pack<T>(acc);
            }
        }
        if (SPLIT) xs[s * 64 + lane] = frag; else xf[si] = frag;
    }
}

// Workgroups per CU the register budget is sized for.  SPLIT kernels run on the small maps: few workgroups, one latency chain per
// tile — what matters there is that a wave can have a whole hidden chunk's weight fragments (2 k1 + DT loads) in flight at once.
// With the non-SPLIT budgets the compiler had 4 fragment registers to cycle through and every second MFMA waited for a fresh L2
// round trip (measured: 7 us per hidden chunk on the 10x10 maps).
#ifndef ACH_MLP_PIPE_SPLIT_DT
#define ACH_MLP_PIPE_SPLIT_DT 0           // SPLIT kernels wider than this take the branch-free, prefetching chunk loop (all of them: +0.9 % on EN-S0 over > 6)
#endif
#ifndef ACH_MLP_PIPE_FLAT_DT
#define ACH_MLP_PIPE_FLAT_DT 0            // one-tile-per-wave kernels up to this width take it too
#endif
#ifndef ACH_MLP_OCC_SPLIT_PIPE
#define ACH_MLP_OCC_SPLIT_PIPE 4
#endif
#ifndef ACH_MLP_OCC_4
#define ACH_MLP_OCC_4 ACH_MLP_OCC_SMALL
#endif
#ifndef ACH_MLP_OCC_SMALL
#define ACH_MLP_OCC_SMALL 6
#endif
#ifndef ACH_MLP_OCC_10
#define ACH_MLP_OCC_10 4
#endif
#ifndef ACH_MLP_OCC_12
#define ACH_MLP_OCC_12 3
#endif
// fp32 storage (the parity engine): a hidden chunk is two k-steps of the second GEMM, the prefetched fragments of the narrow SPLIT widths do
// not fit four workgroups per CU (332 bytes of spills at DT 6) — those keep the plain chunk loop and the small budget.
template <int DT, bool SPLIT, int VEC> struct MlpOcc {
    static constexpr int pipe_split_dt = VEC == 8 ? ACH_MLP_PIPE_SPLIT_DT : 6;
    static constexpr int blocks = (SPLIT && DT > 6) ? 2 : (SPLIT && DT > pipe_split_dt) ? ACH_MLP_OCC_SPLIT_PIPE : (DT == 4 ? ACH_MLP_OCC_4 : DT <= 6 ? ACH_MLP_OCC_SMALL : (DT <= 10 ? ACH_MLP_OCC_10 : (DT <= 12 ? ACH_MLP_OCC_12 : 2)));
};

template <class T, int DT, bool SPLIT, bool EVEN = false>
__global__ __launch_bounds__(256, (MlpOcc<DT, SPLIT, Store<T>::VEC>::blocks)) void mlp_kernel(const MlpParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC;
    constexpr int KC = 4 * VEC;
    constexpr int K1MAX = (16 * DT + KC - 1) / KC;
    constexpr int HSTEP = 8 / VEC;                       // k-steps of the second GEMM per hidden chunk of 32
    constexpr int RT = DT < MLP_RED_TILES ? DT : MLP_RED_TILES;
    __shared__ uint4 xs[SPLIT ? K1MAX * 64 : 1];         // SPLIT: exchanged input fragments
    __shared__ float st[SPLIT ? 4 * 16 * 2 : 1];         //        per-wave partial LayerNorm sums
    __shared__ float red[SPLIT ? 4 * RT * 4 * 64 : 1];   //        partial outputs, RT tiles per round
    // one-tile-per-wave kernels of up to 96 channels (stages 0 / 1), depthwise 3 x 3 / 5 x 5: [taps][ldc] weights + [ldc] bias, staged once per workgroup (mlp_dw WL)
    constexpr bool DWL = ACH_MLP_DW_LDS && !SPLIT && DT <= 6;
    __shared__ __attribute__((aligned(16))) float wl[DWL ? 26 * K1MAX * KC : 4];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    // depthwise mode re-reads halo rows across tiles: XCD-aware order keeps those re-reads inside one L2 (ach_platform.h)
    const unsigned wgid = p.dw_k > 0 ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const long tile = SPLIT ? long(wgid) : long(wgid) * 4 + wave;
    const long mraw = tile * 16 + px;
    const bool valid = mraw < p.M;
    const long m = valid ? mraw : 0;

    // ---- 1. input fragments and LayerNorm sums
    uint4 xf[K1MAX];
    ACH_UNROLL
    for (int s = 0; s < K1MAX; ++s) xf[s] = make_uint4(0u, 0u, 0u, 0u);
    float s1 = 0.f, s2 = 0.f;
    bool staged = false;
    if constexpr (DWL) {
        if (p.dw_k == 3 || p.dw_k == 5) {           // (launch-uniform)
            const int ldc = p.k1 * KC, nw = p.dw_k * p.dw_k * ldc;
            for (int i = int(threadIdx.x) * 4; i < nw + ldc; i += 1024)
                *reinterpret_cast<f32x4*>(wl + i) = *reinterpret_cast<const f32x4*>(i < nw ? p.Wdw + i : p.bdw + (i - nw));
            __syncthreads();
            staged = true;
        }
    }
    if (staged) {
        if constexpr (DWL) {
            if (p.dw_k == 3) mlp_inputs<T, K1MAX, SPLIT, 3, EVEN, true>(p, m, valid, g, wave, lane, xs, xf, s1, s2, red, wl);
            else mlp_inputs<T, K1MAX, SPLIT, 5, EVEN, true>(p, m, valid, g, wave, lane, xs, xf, s1, s2, red, wl);
        }
    } else
    switch (p.dw_k) {
        case 0: mlp_inputs<T, K1MAX, SPLIT, 0, EVEN>(p, m, valid, g, wave, lane, xs, xf, s1, s2, red); break;
        case 3: mlp_inputs<T, K1MAX, SPLIT, 3, EVEN>(p, m, valid, g, wave, lane, xs, xf, s1, s2, red); break;
        case 5: mlp_inputs<T, K1MAX, SPLIT, 5, EVEN>(p, m, valid, g, wave, lane, xs, xf, s1, s2, red); break;
        case 7: mlp_inputs<T, K1MAX, SPLIT, 7, EVEN>(p, m, valid, g, wave, lane, xs, xf, s1, s2, red); break;
        default: mlp_inputs<T, K1MAX, SPLIT, 9, EVEN>(p, m, valid, g, wave, lane, xs, xf, s1, s2, red); break;
    }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (SPLIT) {
        if (g == 0) { st[(wave * 16 + px) * 2] = s1; st[(wave * 16 + px) * 2 + 1] = s2; }
        __syncthreads();
        ACH_UNROLL
        for (int s = 0; s < K1MAX; ++s)
            if (s < p.k1) xf[s] = xs[s * 64 + lane];
        s1 = 0.f; s2 = 0.f;
        ACH_UNROLL
        for (int w = 0; w < 4; ++w) { s1 += st[(w * 16 + px) * 2]; s2 += st[(w * 16 + px) * 2 + 1]; }
    }
    // ---- 2. LayerNorm (affine folded into W1 / b1)
    if (p.ln) {
        const float mu = s1 / float(p.C);
        float var = s2 / float(p.C) - mu * mu;
        var = var > 0.f ? var : 0.f;
        const float rstd = ln_rstd(var + p.ln_eps);
        ACH_UNROLL
        for (int s = 0; s < K1MAX; ++s) {
            if (s >= p.k1) continue;
            const int k0 = s * KC + g * VEC;
            float v[8];
            frag_unpack<T>(xf[s], v);
            ACH_UNROLL
            for (int i = 0; i < VEC; ++i) v[i] = (k0 + i < p.C) ? (v[i] - mu) * rstd : 0.f;
            xf[s] = frag_pack<T>(v);
        }
    }
    // ---- 3. hidden chunks of 32 channels: GEMM1 (2 tiles) -> bias, activation -> GEMM2 accumulation
    f32x4 acc2[DT];
    ACH_UNROLL
    for (int t = 0; t < DT; ++t) { acc2[t][0] = 0.f; acc2[t][1] = 0.f; acc2[t][2] = 0.f; acc2[t][3] = 0.f; }
    const uint4* W1f = static_cast<const uint4*>(p.W1) + lane;
    const uint4* W2f = static_cast<const uint4*>(p.W2) + lane;
    // SPLIT: branch-free and software-pipelined — all of a chunk's W1 fragments are requested together one chunk ahead, its W2
    // fragments before the activation (k-steps beyond k1 re-read the last fragment against a zero input instead of branching:
    // with a branch per k-step every pair of MFMAs waited for its own L2 round trip).
    // DT <= 6 keeps the small register budget (more workgroups per CU next to the side streams: measured better end to end);
    // DT > 12: branch-free, but the next chunk's fragments are not requested ahead (they would spill).
    constexpr bool PIPE = SPLIT ? DT > MlpOcc<DT, SPLIT, VEC>::pipe_split_dt : DT <= ACH_MLP_PIPE_FLAT_DT;
    constexpr bool AHEAD = PIPE && DT <= 12;
    constexpr int JS = SPLIT ? 4 : 1;
    const int j0 = SPLIT ? wave : 0;
    uint4 wn[PIPE ? K1MAX : 1][2];
    auto load_w1 = [&](int j) {
        const uint4* w1 = W1f + long(j) * p.k1 * 2 * 64;
        ACH_UNROLL
        for (int s = 0; s < K1MAX; ++s) {
            const int sc = s < p.k1 ? s : p.k1 - 1;
            wn[s][0] = w1[(sc * 2) * 64]; wn[s][1] = w1[(sc * 2 + 1) * 64];
        }
    };
    // The activation is fixed at compile time inside the loop (GELU: EdgeNeXt, SiLU: MobileViT; anything else through the run-time form):
    // with `p.act` tested per element the chunk loop was eight serialised load -> wait -> branch ladder -> exp -> rcp chains.
    auto hidden = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
        if (AHEAD && j0 < p.J) load_w1(j0);
        for (int j = j0; j < ((ACH_MLP_DEBUG & 2) ? 0 : p.J); j += JS) {
            f32x4 a0, a1;
            a0[0] = a0[1] = a0[2] = a0[3] = 0.f;
            a1[0] = a1[1] = a1[2] = a1[3] = 0.f;
            const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b1 + j * 32 + g * 8), bB = *reinterpret_cast<const f32x4*>(p.b1 + j * 32 + g * 8 + 4);
            uint4 w2r[PIPE ? HSTEP : 1][PIPE ? DT : 1];
            if (PIPE) {
                if (!AHEAD) load_w1(j);
                ACH_UNROLL
                for (int s = 0; s < K1MAX; ++s) { mfma16<T>(wn[s][0], xf[s], a0); mfma16<T>(wn[s][1], xf[s], a1); }
                ACH_UNROLL
                for (int hh = 0; hh < HSTEP; ++hh)
                    ACH_UNROLL
                    for (int t = 0; t < DT; ++t) w2r[hh][t] = W2f[(long(j * HSTEP + hh) * DT + t) * 64];
                if (AHEAD && j + JS < p.J) load_w1(j + JS);
            } else {
            const uint4* w1 = W1f + long(j) * p.k1 * 2 * 64;
            ACH_UNROLL
            for (int s = 0; s < K1MAX; ++s) {
                if (s >= p.k1) continue;
                const uint4 wa = w1[(s * 2) * 64], wb = w1[(s * 2 + 1) * 64];
                mfma16<T>(wa, xf[s], a0);
                mfma16<T>(wb, xf[s], a1);
            }
            }
            float h[8];
#if ACH_MLP_MFMA_PAD
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // experiment: 64 extra wait states between the MFMAs and the first read of their results
#endif
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { h[r] = a0[r] + bA[r]; h[4 + r] = a1[r] + bB[r]; }
            apply_act_n<T, 8, ((ACH_MLP_DEBUG & 4) ? int(ACT_RELU) : ACT)>(h, p.act);
            ACH_UNROLL
            for (int hh = 0; hh < HSTEP; ++hh) {
                const uint4 hf = frag_pack<T>(h + hh * VEC);
                const uint4* w2 = W2f + long(j * HSTEP + hh) * DT * 64;
                ACH_UNROLL
                for (int t = 0; t < DT; ++t) mfma16<T>(PIPE ? w2r[hh][t] : w2[t * 64], hf, acc2[t]);
            }
        }
    };
    switch (p.act) {
        case ACT_GELU: hidden(std::integral_constant<int, ACT_GELU>{}); break;
        case ACT_SILU: hidden(std::integral_constant<int, ACT_SILU>{}); break;
        default: hidden(std::integral_constant<int, -1>{}); break;
    }
    // ---- 4. + bias + residual, 8 consecutive channels per lane per tile pair
    auto finish = [&](int pair, const float* v8) {
        const int nb = pair * 32 + g * 8;
        if (!valid || nb >= p.Cout) return;
        float o[8], r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.R) Store<T>::ld8(static_cast<const T*>(p.R) + m * p.ldr + nb, r8);
        ACH_UNROLL
        for (int i = 0; i < 8; ++i) o[i] = v8[i] + p.b2[nb + i] + r8[i];
        Store<T>::st8(static_cast<T*>(p.Y) + m * p.ldy + nb, o);
    };
#if ACH_MLP_MFMA_PAD
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
    if (!SPLIT) {
        if constexpr (sizeof(T) == 2) {
            // every residual row and bias vector of the tile requested first, unconditionally (clamped addresses), then the arithmetic and the stores: inside
            // `finish` each pair's loads sit behind a lane condition, i.e. a branch, and the wave paid one memory round trip PER PAIR (five at d = 144) — each of
            // them also waiting for the previous pair's store (round 4).  Same sums in the same order.
            constexpr int NP = DT / 2;
            uint4 rr[NP];
            float4 ba[NP], bb[NP];
            const bool hasR = p.R != nullptr;
            ACH_UNROLL
            for (int pair = 0; pair < NP; ++pair) {
                const int nb = pair * 32 + g * 8;
                ba[pair] = *reinterpret_cast<const float4*>(p.b2 + nb);
                bb[pair] = *reinterpret_cast<const float4*>(p.b2 + nb + 4);
                rr[pair] = make_uint4(0u, 0u, 0u, 0u);
                if (hasR) rr[pair] = *reinterpret_cast<const uint4*>(static_cast<const T*>(p.R) + m * p.ldr + (nb < p.Cout ? nb : 0));
            }
            ACH_UNROLL
            for (int pair = 0; pair < NP; ++pair) {
                const int nb = pair * 32 + g * 8;
                float r8[8], o[8];
                frag_unpack<T>(rr[pair], r8);
                const float bv[8] = {ba[pair].x, ba[pair].y, ba[pair].z, ba[pair].w, bb[pair].x, bb[pair].y, bb[pair].z, bb[pair].w};
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) { o[r] = acc2[2 * pair][r] + bv[r] + r8[r]; o[4 + r] = acc2[2 * pair + 1][r] + bv[4 + r] + r8[4 + r]; }
                if (valid && nb < p.Cout) Store<T>::st8(static_cast<T*>(p.Y) + m * p.ldy + nb, o);
            }
        } else {
        ACH_UNROLL
        for (int pair = 0; pair < DT / 2; ++pair) {
            float v8[8];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { v8[r] = acc2[2 * pair][r]; v8[4 + r] = acc2[2 * pair + 1][r]; }
            finish(pair, v8);
        }
        }
    } else {
        ACH_UNROLL
        for (int t0 = 0; t0 < DT; t0 += RT) {
            if (t0 > 0) __syncthreads();
            ACH_UNROLL
            for (int t = 0; t < RT; ++t) {
                if (t0 + t >= DT) continue;
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) red[((wave * RT + t) * 4 + r) * 64 + lane] = acc2[t0 + t][r];
            }
            __syncthreads();
            ACH_UNROLL
            for (int q = 0; q < RT / 2; ++q) {                      // pair q of this round belongs to wave q % 4
                if ((q & 3) != wave || t0 + 2 * q >= DT) continue;
                float v8[8];
                ACH_UNROLL
                for (int i = 0; i < 8; ++i) {
                    const int t = 2 * q + (i >> 2), r = i & 3;
                    float a = 0.f;
                    ACH_UNROLL
                    for (int w = 0; w < 4; ++w) a += red[((w * RT + t) * 4 + r) * 64 + lane];
                    v8[i] = a;
                }
                finish((t0 >> 1) + q, v8);
            }
        }
    }
}

template <class T>
inline bool launch_mlp(const MlpParams& p, int DT, bool split, hipStream_t stream) {
    const long tiles = (p.M + 15) / 16;
    const dim3 grid(unsigned(split ? tiles : (tiles + 3) / 4)), block(256);
    // the even depthwise row split is instantiated where it pays: d = 144 (DT 10; EN-S2 stage 2, +0.7 % frames/s).  At d = 96 (DT 6, the
    // 80-register budget of six workgroups per CU) the variant spills 80 bytes and EN-S0 loses 0.8 %: not instantiated.
    if (split && p.dw_even && DT == 10) { ACH_LAUNCH((mlp_kernel<T, 10, true, true>), grid, block, stream, p); return true; }
#define ACH_MLP_CASE(dt) \
    if (DT == dt) { if (split) ACH_LAUNCH((mlp_kernel<T, dt, true>), grid, block, stream, p); else ACH_LAUNCH((mlp_kernel<T, dt, false>), grid, block, stream, p); return true; }
    ACH_MLP_CASE(2) ACH_MLP_CASE(4) ACH_MLP_CASE(6) ACH_MLP_CASE(8) ACH_MLP_CASE(10) ACH_MLP_CASE(12) ACH_MLP_CASE(18) ACH_MLP_CASE(20)
#undef ACH_MLP_CASE
    return false;
}
// ------------------------------------------------------------------------------------------ two tiles per wave (round 5)
// The feed-forward blocks of MobileViT's transformers (LN -> fc1 -> SiLU -> fc2 -> + input; d = 144 / 192, hidden 2d) are plain-row launches of mlp_kernel with one
// 16-row tile per wave: every wave streams the layer's 2 d * 2d weights (166 / 295 KB) through L1 for its 16 rows — ~1 GB of L2 reads per layer at batch 64, and the
// measured time (99 - 105 us at d = 144) IS that volume at the L2's rate.  Here a wave owns TWO tiles: a weight fragment is loaded once and feeds
// two MFMAs, half the L2 traffic; same weight packing, the same sums in the same order per row (bit-identical to mlp_kernel).
// Its waves live for 9 - 12 hidden chunks of ~40 matrix instructions: they use the 16x16x16 PAIR form (mfma16_pair, ach_platform.h) — with v_mfma_f32_16x16x32 the co-residency
// guard saw MV-S2's outputs change beside every aggressor, poison waves included (5 - 12 of 20 passes; first tap map3), i.e. the kernel's waves disturbed each other.
#ifndef ACH_FFN2_MFMA32
#define ACH_FFN2_MFMA32 0          // 1 (profiles/scripts/ubench/ffn2_coresidency.hip only): the v_mfma_f32_16x16x32 form that the co-residency guard rejected
#endif
template <class T> __device__ __forceinline__ void ffn2_mfma(const uint4& a, const uint4& b, f32x4& c) {
#if ACH_FFN2_MFMA32
    mfma16<T>(a, b, c);
#else
    mfma16_pair<T>(a, b, c);
#endif
}
template <class T, int DT>
__global__ __launch_bounds__(256, 2) void ffn2_kernel(const MlpParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC;
    constexpr int KC = 4 * VEC;
    constexpr int K1MAX = (16 * DT + KC - 1) / KC;
    constexpr int HSTEP = 8 / VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const long pair = long(blockIdx.x) * 4 + wave;
    const T* X = static_cast<const T*>(p.X);
    long m[2]; bool valid[2];
    uint4 xf[2][K1MAX];
    ACH_UNROLL
    for (int q = 0; q < 2; ++q) {
        const long mraw = (pair * 2 + q) * 16 + px;
        valid[q] = mraw < p.M;
        m[q] = valid[q] ? mraw : 0;
        float s1 = 0.f, s2 = 0.f;
        ACH_UNROLL
        for (int s = 0; s < K1MAX; ++s) {
            xf[q][s] = make_uint4(0u, 0u, 0u, 0u);
            const int k0 = s * KC + g * VEC;
            if (s < p.k1 && valid[q] && k0 < p.C) {
                xf[q][s] = *reinterpret_cast<const uint4*>(X + m[q] * p.ldx + k0);
                float v[8];
                frag_unpack<T>(xf[q][s], v);
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) { s1 += v[i]; s2 += v[i] * v[i]; }
            }
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (p.ln) {
            const float mu = s1 / float(p.C);
            float var = s2 / float(p.C) - mu * mu;
            var = var > 0.f ? var : 0.f;
            const float rstd = ln_rstd(var + p.ln_eps);
            ACH_UNROLL
            for (int s = 0; s < K1MAX; ++s) {
                if (s >= p.k1) continue;
                const int k0 = s * KC + g * VEC;
                float v[8];
                frag_unpack<T>(xf[q][s], v);
                ACH_UNROLL
                for (int i = 0; i < VEC; ++i) v[i] = (k0 + i < p.C) ? (v[i] - mu) * rstd : 0.f;
                xf[q][s] = frag_pack<T>(v);
            }
        }
    }
    f32x4 acc2[2][DT];
    ACH_UNROLL
    for (int q = 0; q < 2; ++q)
        ACH_UNROLL
        for (int t = 0; t < DT; ++t) { acc2[q][t][0] = 0.f; acc2[q][t][1] = 0.f; acc2[q][t][2] = 0.f; acc2[q][t][3] = 0.f; }
    const uint4* W1f = static_cast<const uint4*>(p.W1) + lane;
    const uint4* W2f = static_cast<const uint4*>(p.W2) + lane;
    auto hidden = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
        for (int j = 0; j < p.J; ++j) {
            f32x4 a0[2], a1[2];
            ACH_UNROLL
            for (int q = 0; q < 2; ++q) { a0[q][0] = a0[q][1] = a0[q][2] = a0[q][3] = 0.f; a1[q][0] = a1[q][1] = a1[q][2] = a1[q][3] = 0.f; }
            const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b1 + j * 32 + g * 8), bB = *reinterpret_cast<const f32x4*>(p.b1 + j * 32 + g * 8 + 4);
            const uint4* w1 = W1f + long(j) * p.k1 * 2 * 64;
            // (k-steps beyond k1 re-read the last fragment against a zero input: branch-free, every load of the chunk can be in flight at once)
            uint4 wa[K1MAX], wb[K1MAX];
            ACH_UNROLL
            for (int s = 0; s < K1MAX; ++s) { const int sc = s < p.k1 ? s : p.k1 - 1; wa[s] = w1[(sc * 2) * 64]; wb[s] = w1[(sc * 2 + 1) * 64]; }
            ACH_UNROLL
            for (int s = 0; s < K1MAX; ++s) {
                ffn2_mfma<T>(wa[s], xf[0][s], a0[0]); ffn2_mfma<T>(wb[s], xf[0][s], a1[0]);
                ffn2_mfma<T>(wa[s], xf[1][s], a0[1]); ffn2_mfma<T>(wb[s], xf[1][s], a1[1]);
            }
            uint4 hf[2][HSTEP];
            ACH_UNROLL
            for (int q = 0; q < 2; ++q) {
                float h[8];
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) { h[r] = a0[q][r] + bA[r]; h[4 + r] = a1[q][r] + bB[r]; }
                apply_act_n<T, 8, ACT>(h, p.act);
                ACH_UNROLL
                for (int hh = 0; hh < HSTEP; ++hh) hf[q][hh] = frag_pack<T>(h + hh * VEC);
            }
            ACH_UNROLL
            for (int hh = 0; hh < HSTEP; ++hh) {
                const uint4* w2 = W2f + long(j * HSTEP + hh) * DT * 64;
                ACH_UNROLL
                for (int t = 0; t < DT; ++t) { const uint4 w = w2[t * 64]; ffn2_mfma<T>(w, hf[0][hh], acc2[0][t]); ffn2_mfma<T>(w, hf[1][hh], acc2[1][t]); }
            }
        }
    };
    switch (p.act) {
        case ACT_GELU: hidden(std::integral_constant<int, ACT_GELU>{}); break;
        case ACT_SILU: hidden(std::integral_constant<int, ACT_SILU>{}); break;
        default: hidden(std::integral_constant<int, -1>{}); break;
    }
    // + bias + residual, 8 consecutive channels per lane per tile pair (mlp_kernel's 16-bit epilogue)
    constexpr int NP = DT / 2;
    const bool hasR = p.R != nullptr;
    ACH_UNROLL
    for (int q = 0; q < 2; ++q) {
        uint4 rr[NP];
        float4 ba[NP], bb[NP];
        ACH_UNROLL
        for (int pr_ = 0; pr_ < NP; ++pr_) {
            const int nb = pr_ * 32 + g * 8;
            ba[pr_] = *reinterpret_cast<const float4*>(p.b2 + nb);
            bb[pr_] = *reinterpret_cast<const float4*>(p.b2 + nb + 4);
            rr[pr_] = make_uint4(0u, 0u, 0u, 0u);
            if (hasR) rr[pr_] = *reinterpret_cast<const uint4*>(static_cast<const T*>(p.R) + m[q] * p.ldr + (nb < p.Cout ? nb : 0));
        }
        ACH_UNROLL
        for (int pr_ = 0; pr_ < NP; ++pr_) {
            const int nb = pr_ * 32 + g * 8;
            float r8[8], o[8];
            frag_unpack<T>(rr[pr_], r8);
            const float bv[8] = {ba[pr_].x, ba[pr_].y, ba[pr_].z, ba[pr_].w, bb[pr_].x, bb[pr_].y, bb[pr_].z, bb[pr_].w};
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { o[r] = acc2[q][2 * pr_][r] + bv[r] + r8[r]; o[4 + r] = acc2[q][2 * pr_ + 1][r] + bv[4 + r] + r8[4 + r]; }
            if (valid[q] && nb < p.Cout) Store<T>::st8(static_cast<T*>(p.Y) + m[q] * p.ldy + nb, o);
        }
    }
}
// plain-row launches (no depthwise conv) of the 16-bit engines whose width is instantiated: two tiles per wave
template <class T>
inline bool launch_ffn2(const MlpParams& p, int DT, hipStream_t stream) {
    if constexpr (sizeof(T) != 2) return false;
    else {
        if (p.dw_k != 0 || (DT != 10 && DT != 12)) return false;
        const long tiles = (p.M + 15) / 16;
        const dim3 grid(unsigned((tiles + 7) / 8)), block(256);
        if (DT == 10) ACH_LAUNCH((ffn2_kernel<T, 10>), grid, block, stream, p); else ACH_LAUNCH((ffn2_kernel<T, 12>), grid, block, stream, p);
        return true;
    }
}
inline bool mlp_even_dt(int DT) { return DT == 10; }      // d = 96 (DT 6) re-measured at the 128-register budget: 45 -> 49 us per block, still not instantiated
inline int mlp_pick_dt(int C) {
    const int need = 2 * ((C + 31) / 32);
    for (int dt : {2, 4, 6, 8, 10, 12, 18, 20}) if (dt >= need) return dt;
    return 0;
}

}  // namespace ach

namespace ach {

// ------------------------------------------------------------------------------------------ small two-layer chain
// y = W2 · act(W1 x + b1) + b2 for narrow layers (<= 48 inputs, <= 64 hidden, <= 32 outputs: the low-resolution conv pair of
// every decoder level, up to 1.6 M pixels).  Same register chaining as mlp_kernel, but everything is compile-time sized so
// that ALL weight fragments live in registers, and a wave walks several 16-pixel tiles with the next tile's rows prefetched:
// per tile only the row load, 2 K1 J + J HSTEP 2 MFMAs, the activation and one 16-byte store remain — the kernel is then a
// plain stream over x and y (HBM-bound).  Weight layouts are mlp_kernel's with DT = 2.
#ifndef ACH_CHAIN_TILES
#define ACH_CHAIN_TILES 4
#endif
constexpr int CHAIN_TILES_PER_WAVE = ACH_CHAIN_TILES;
#ifndef ACH_CHAIN_WAVES
#define ACH_CHAIN_WAVES 0
#endif
#if ACH_CHAIN_WAVES > 0
#define ACH_CHAIN_BOUNDS __launch_bounds__(256, ACH_CHAIN_WAVES)
#else
#define ACH_CHAIN_BOUNDS __launch_bounds__(256)
#endif
template <class T, int K1, int J>
__global__ ACH_CHAIN_BOUNDS void chain_kernel(const MlpParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC;
    constexpr int KC = 4 * VEC;
    constexpr int HSTEP = 8 / VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const uint4* W1f = static_cast<const uint4*>(p.W1) + lane;
    const uint4* W2f = static_cast<const uint4*>(p.W2) + lane;
    uint4 w1[J][K1][2], w2[J][HSTEP][2];
    ACH_UNROLL
    for (int j = 0; j < J; ++j) {
        ACH_UNROLL
        for (int s = 0; s < K1; ++s) { w1[j][s][0] = W1f[((j * K1 + s) * 2) * 64]; w1[j][s][1] = W1f[((j * K1 + s) * 2 + 1) * 64]; }
        ACH_UNROLL
        for (int hh = 0; hh < HSTEP; ++hh) { w2[j][hh][0] = W2f[((j * HSTEP + hh) * 2) * 64]; w2[j][hh][1] = W2f[((j * HSTEP + hh) * 2 + 1) * 64]; }
    }
    float b1[J][8], b2[8];
    ACH_UNROLL
    for (int j = 0; j < J; ++j)
        ACH_UNROLL
        for (int i = 0; i < 8; ++i) b1[j][i] = p.b1[j * 32 + g * 8 + i];
    ACH_UNROLL
    for (int i = 0; i < 8; ++i) b2[i] = p.b2[g * 8 + i];

    const T* X = static_cast<const T*>(p.X);
    const long tile0 = (long(blockIdx.x) * 4 + wave) * CHAIN_TILES_PER_WAVE;
    uint4 xf[K1];
    auto fetch = [&](long tile) {
        const long mr = tile * 16 + px;
        const long m = mr < p.M ? mr : p.M - 1;
        ACH_UNROLL
        for (int s = 0; s < K1; ++s) {
            const int k0 = s * KC + g * VEC;
            xf[s] = k0 < p.C ? *reinterpret_cast<const uint4*>(X + m * p.ldx + k0) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (tile0 * 16 < p.M) fetch(tile0);
    ACH_UNROLL
    for (int t = 0; t < CHAIN_TILES_PER_WAVE; ++t) {
        const long tile = tile0 + t;
        if (tile * 16 >= p.M) break;
        f32x4 acc[2];
        acc[0][0] = acc[0][1] = acc[0][2] = acc[0][3] = 0.f;
        acc[1][0] = acc[1][1] = acc[1][2] = acc[1][3] = 0.f;
        f32x4 h0[J], h1[J];
        ACH_UNROLL
        for (int j = 0; j < J; ++j) {
            h0[j][0] = h0[j][1] = h0[j][2] = h0[j][3] = 0.f;
            h1[j][0] = h1[j][1] = h1[j][2] = h1[j][3] = 0.f;
            ACH_UNROLL
            for (int s = 0; s < K1; ++s) { mfma16<T>(w1[j][s][0], xf[s], h0[j]); mfma16<T>(w1[j][s][1], xf[s], h1[j]); }
        }
        const long mr = tile * 16 + px;
        if (t + 1 < CHAIN_TILES_PER_WAVE && (tile + 1) * 16 < p.M) fetch(tile + 1);      // xf is free: all first-layer MFMAs are issued
        ACH_UNROLL
        for (int j = 0; j < J; ++j) {
            float h[8];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { h[r] = h0[j][r] + b1[j][r]; h[4 + r] = h1[j][r] + b1[j][4 + r]; }
            apply_act_n<T, 8>(h, p.act);
            if (p.Hout && mr < p.M) Store<T>::st8(static_cast<T*>(p.Hout) + mr * p.ldh + j * 32 + g * 8, h);      // hidden channels j * 32 + 8 g .. + 7 are this lane's (see the header)
            ACH_UNROLL
            for (int hh = 0; hh < HSTEP; ++hh) {
                const uint4 hf = frag_pack<T>(h + hh * VEC);
                mfma16<T>(w2[j][hh][0], hf, acc[0]);
                mfma16<T>(w2[j][hh][1], hf, acc[1]);
            }
        }
        const int nb = g * 8;
        if (mr < p.M && nb < p.Cout) {
            float o[8];
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) { o[r] = acc[0][r] + b2[r]; o[4 + r] = acc[1][r] + b2[4 + r]; }
            if (p.planar_w > 0) {                       // [row][channel][x]: the layout upghost_head_mfma_kernel's A operand wants
                const long row = mr / p.planar_w;
                T* yo = static_cast<T*>(p.Y) + (row * p.Cout + nb) * p.planar_w + (mr - row * p.planar_w);
                ACH_UNROLL
                for (int i = 0; i < 8; ++i) Store<T>::st(yo + long(i) * p.planar_w, o[i]);
            } else {
                Store<T>::st8(static_cast<T*>(p.Y) + mr * p.ldy + nb, o);
            }
        }
    }
}

template <class T>
inline bool launch_chain(const MlpParams& p, hipStream_t stream) {
    const long tiles = (p.M + 15) / 16;
    const dim3 grid(unsigned((tiles + 4 * CHAIN_TILES_PER_WAVE - 1) / (4 * CHAIN_TILES_PER_WAVE))), block(256);
#define ACH_CHAIN_CASE(KK, JJ) if (p.k1 == KK && p.J == JJ) { ACH_LAUNCH((chain_kernel<T, KK, JJ>), grid, block, stream, p); return true; }
    ACH_CHAIN_CASE(1, 1) ACH_CHAIN_CASE(2, 1) ACH_CHAIN_CASE(3, 1) ACH_CHAIN_CASE(1, 2) ACH_CHAIN_CASE(2, 2) ACH_CHAIN_CASE(3, 2)
#undef ACH_CHAIN_CASE
    return false;
}

}  // namespace ach
