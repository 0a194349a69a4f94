// k_gemm.h — the pointwise / linear workhorse: Y[m, n] = epilogue( sum_k X[m, k] * W[n, k] ) on MFMA.
//
// Serves every 1x1 conv and nn.Linear of the path (EdgeNeXt pw-MLPs and qkv/proj, neck 1x1s, Ghost primary
// convs, head stems/pw/preds, PointNet shared MLPs and FC stack, MobileViT projections) and the 2x2/s2
// patchify convs of EdgeNeXt (rows gathered from two contiguous NHWC segments).
//
// Shape regime: M = B*H*W is huge (up to 6.5 M rows), K and N are tiny (8..704).  So:
//   * weights are the MFMA *A* operand, activations the *B* operand: a lane's accumulator registers are then
//     consecutive OUTPUT CHANNELS of one pixel, i.e. contiguous bytes of the NHWC output row (wide stores);
//   * weights are pre-packed on the host in exact fragment order (one 1 KiB coalesced read per fragment,
//     L1/L2 resident, no LDS staging and no barrier in the kernel);
//   * an activation fragment is 16 B per lane, 4 lane groups reading 64 contiguous bytes of each of 16 rows;
//     every activation row is consumed by exactly one wave;
//   * fp32 storage uses v_mfma_f32_16x16x4_f32 (exact fp32, parity path), bf16 storage v_mfma_f32_16x16x32_bf16.
// Fusions: LayerNorm over K as a prologue (affine folded into W/bias on the host), bias (BatchNorm folded),
// activation, residual add, NCHW scatter for network outputs, per-group (per-sample) weights, and a
// column max over the rows of a group (PointNet's max over points) reduced with wavefront shuffles.
#pragma once
#include "ach_platform.h"

namespace ach {

// Offset (in elements) of W[n][k] inside one group's packed weight blob.  Shared by the host packer and the
// device-side packer for data-dependent weights (PointNet feature transform).
// Output-channel order inside a chunk of 16*NT channels is chosen for the STORES: lane group g = lane>>4 of the MFMA
// result owns, per pair of 16-row tiles (t = 2j, 2j+1), the 8 consecutive channels  j*32 + g*8 .. +7  — so one 16-byte
// (bf16) / two 16-byte (fp32) stores per lane, and the 4 lane groups of a pixel write 64 / 128 contiguous bytes.
// (NT = 1: 4 consecutive channels g*4 .. +3.)
__host__ __device__ __forceinline__ int chunk_channel(int NT, int t, int g, int r) {
    return NT == 1 ? g * 4 + r : (t >> 1) * 32 + g * 8 + (t & 1) * 4 + r;
}
__host__ __device__ __forceinline__ long wfrag_offset(int n, int k, int NT, int ksteps, int VEC) {
    const int CH = 16 * NT;
    const int c = n / CH, nn = n % CH;
    int g, t, r;
    if (NT == 1) { g = nn >> 2; t = 0; r = nn & 3; }
    else { const int j = nn >> 5, w = nn & 31; g = w >> 3; t = j * 2 + ((w & 7) >> 2); r = w & 3; }
    const int i = g * 4 + r;
    const int KC = 4 * VEC;
    const int s = k / KC, kk = k % KC, kg = kk / VEC, kj = kk % VEC;
    return ((long(c) * ksteps + s) * NT + t) * (64L * VEC) + long(kg * 16 + i) * VEC + kj;
}

struct GemmParams {
    const void* X; long ldx;            // activations: row m at X + m*ldx (elements), K contiguous channels
    // conv_k > 0: implicit-GEMM convolution over an NHWC input [.,Hin,Win,(Cin<=ldx)]: row m is output pixel (b,oy,ox) of a
    // conv_k x conv_k / stride conv_s / zero-pad conv_p conv, K ordered (tap, channel) with Cin channels per tap
    // (Cin a multiple of the 16-byte vector).  Covers EdgeNeXt's 2x2/s2 patchify convs and every dense 3x3.
    int conv_k, conv_s, conv_p, Hin, Win, Cin, Ho, Wo;
    const void* W; long w_group_stride; // packed fragments; per-group stride in elements (0: shared)
    const float* bias; long bias_group_stride;
    void* Y; long ldy;                  // NHWC: row m at Y + m*ldy ; NCHW: see out_nchw
    const void* R; long ldr;            // optional residual, NHWC rows
    int M_per_group, groups, K, N;
    int nchunks, ksteps;
    int chunks_per_block;               // blockIdx.z selects a contiguous range of N-chunks (fills the chip when M is small)
    int act, ln; float ln_eps;
    int out_nchw, HW, Ctot, coff;       // NCHW scatter: Y[((b*Ctot + coff + n)*HW + p)], m = b*HW + p (2: the caller's tensor is bf16 while the engine stores fp16)
    int vec_store;                      // Y/R rows and channel offsets are 16-byte compatible
};

__device__ __forceinline__ unsigned order_encode(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float order_decode(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

#ifndef ACH_GEMM_DEPTH
#define ACH_GEMM_DEPTH 3
#endif
constexpr int GEMM_DEPTH = ACH_GEMM_DEPTH;      // operand slots in flight in gemm_body's k-loop (DEEP instantiations)
#ifndef ACH_GEMM_DEEP_KSTEPS
#define ACH_GEMM_DEEP_KSTEPS 7
#endif
constexpr int GEMM_DEEP_KSTEPS = ACH_GEMM_DEEP_KSTEPS;   // k-steps from which launch_gemm picks them

// LNTAP (round 4): conv mode with a channels-first LayerNorm of the INPUT pixels fused in — every tap of the 2x2 / stride-2 patchify convs of
// EdgeNeXt is a different input pixel with its own mean / variance over its Cin channels (edgenext.py:29-34: LayerNorm then Conv2d); the affine
// part is folded into the conv weights on the host.  Saves the LayerNorm launch and its tensor in front of each of the three convs.
template <class T, int NT, int P, bool DEEP = false, bool LNTAP = false>
__device__ __forceinline__ void gemm_body(const GemmParams& p, unsigned bx, unsigned nbx, unsigned by, unsigned bz, int wave_in = -1) {
    constexpr int VEC = Store<T>::VEC;
    constexpr int KC = 4 * VEC;
    // wave_in >= 0: the caller deals (row tile, chunk range) units to its waves itself (xca_frame_kernel, k_xcaframe.h); the body has no barrier
    const int lane = threadIdx.x & 63, wave = wave_in >= 0 ? wave_in : int(threadIdx.x >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int grp = by;
    // implicit-GEMM convs re-read halo rows across row tiles: XCD-aware tile order keeps those re-reads inside one L2
    const unsigned rt = p.conv_k > 1 ? xcd_block(bx, nbx) : bx;
    const long row0 = long(rt) * (64 * P) + wave * (16 * P);           // first row (within the group) of this wave
    const T* X = static_cast<const T*>(p.X);
    const uint4* Wf = reinterpret_cast<const uint4*>(static_cast<const T*>(p.W) + long(grp) * p.w_group_stride);
    const float* bias = p.bias + long(grp) * p.bias_group_stride;

    // per sub-tile: this lane's row, its validity and the element offset of its first channel
    long xoff[P];
    bool valid[P];
    long mrow[P];
    int iy0[P], ix0[P];                       // conv mode: top-left input coordinate of the receptive field
    ACH_UNROLL
    for (int q = 0; q < P; ++q) {
        const long mloc = row0 + q * 16 + px;
        valid[q] = mloc < p.M_per_group;
        const long m = long(grp) * p.M_per_group + (valid[q] ? mloc : 0);
        mrow[q] = m;
        iy0[q] = 0; ix0[q] = 0;
        if (p.conv_k > 0) {
            const long b = m / (long(p.Ho) * p.Wo);
            const int rem = int(m - b * long(p.Ho) * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            iy0[q] = oy * p.conv_s - p.conv_p;
            ix0[q] = ox * p.conv_s - p.conv_p;
            xoff[q] = b * p.Hin * long(p.Win);            // pixel index of the sample's first input pixel
        } else {
            xoff[q] = m * p.ldx;
        }
    }

    auto load_x = [&](int q, int s) -> uint4 {
        const int k0 = s * KC + g * VEC;
        if (!valid[q] || k0 >= p.K) return make_uint4(0u, 0u, 0u, 0u);
        long off;
        if (p.conv_k > 0) {
            const int tap = k0 / p.Cin, c = k0 - tap * p.Cin;
            const int ty = tap / p.conv_k;
            const int iy = iy0[q] + ty, ix = ix0[q] + (tap - ty * p.conv_k);
            if (iy < 0 || iy >= p.Hin || ix < 0 || ix >= p.Win) return make_uint4(0u, 0u, 0u, 0u);
            off = (xoff[q] + long(iy) * p.Win + ix) * p.ldx + c;
        } else {
            off = xoff[q] + k0;
        }
        return *reinterpret_cast<const uint4*>(X + off);
    };

    // ---- optional LayerNorm prologue: mean / variance over the K channels of each row (one sweep: sum and sum of squares
    //      in fp32; the rows are O(1) activations, so E[x^2] - mean^2 loses nothing that matters at eps = 1e-6)
    float mean[P], rstd[P];
    float tmean[P][4], trstd[P][4];                // LNTAP: per (row, tap)
    auto tap_of = [&](int s) { return (s * KC + g * VEC) / p.Cin; };
    if constexpr (LNTAP) {
        ACH_UNROLL
        for (int q = 0; q < P; ++q) {
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < p.ksteps; ++s) {
                float v[8];
                frag_unpack<T>(load_x(q, s), v);                   // a lane's VEC channels belong to ONE tap (Cin is a multiple of VEC)
                float a1 = 0.f, a2 = 0.f;
                ACH_UNROLL
                for (int j = 0; j < VEC; ++j) { a1 += v[j]; a2 += v[j] * v[j]; }
                const int tp = tap_of(s);
                ACH_UNROLL
                for (int t = 0; t < 4; ++t) { s1[t] += tp == t ? a1 : 0.f; s2[t] += tp == t ? a2 : 0.f; }
            }
            ACH_UNROLL
            for (int t = 0; t < 4; ++t) {
                s1[t] += __shfl_xor(s1[t], 16); s2[t] += __shfl_xor(s2[t], 16);
                s1[t] += __shfl_xor(s1[t], 32); s2[t] += __shfl_xor(s2[t], 32);
                const float mu = s1[t] / float(p.Cin);
                float var = s2[t] / float(p.Cin) - mu * mu;
                var = var > 0.f ? var : 0.f;
                tmean[q][t] = mu;
                trstd[q][t] = ln_rstd(var + p.ln_eps);
            }
        }
    }
    auto ln_tap = [&](int q, int s, uint4& xf) {       // x -> (x - mean[tap]) * rstd[tap]
        const int tp = tap_of(s);
        const float mu = tp == 0 ? tmean[q][0] : (tp == 1 ? tmean[q][1] : (tp == 2 ? tmean[q][2] : tmean[q][3]));
        const float rs = tp == 0 ? trstd[q][0] : (tp == 1 ? trstd[q][1] : (tp == 2 ? trstd[q][2] : trstd[q][3]));
        float v[8];
        frag_unpack<T>(xf, v);
        ACH_UNROLL
        for (int j = 0; j < VEC; ++j) v[j] = (v[j] - mu) * rs;
        xf = frag_pack<T>(v);
    };
    if (!LNTAP && p.ln) {
        ACH_UNROLL
        for (int q = 0; q < P; ++q) {
            float s1 = 0.f, s2 = 0.f;
            for (int s = 0; s < p.ksteps; ++s) {
                float v[8];
                frag_unpack<T>(load_x(q, s), v);                   // channels >= K load as 0
                ACH_UNROLL
                for (int j = 0; j < VEC; ++j) { s1 += v[j]; s2 += v[j] * v[j]; }
            }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            const float mu = s1 / float(p.K);
            float var = s2 / float(p.K) - mu * mu;
            var = var > 0.f ? var : 0.f;
            mean[q] = mu;
            rstd[q] = ln_rstd(var + p.ln_eps);
        }
    }

    const int c_begin = int(bz) * p.chunks_per_block;
    const int c_end = (c_begin + p.chunks_per_block < p.nchunks) ? c_begin + p.chunks_per_block : p.nchunks;
    for (int c = c_begin; c < c_end; ++c) {
        f32x4 acc[P][NT];
        ACH_UNROLL
        for (int q = 0; q < P; ++q)
            ACH_UNROLL
            for (int t = 0; t < NT; ++t) { acc[q][t][0] = 0.f; acc[q][t][1] = 0.f; acc[q][t][2] = 0.f; acc[q][t][3] = 0.f; }

        // the chunk's bias, requested before the k-loop (it used to be 4 NT predicated dword loads AFTER it, their latency exposed): channel
        // chunk_channel(NT, t, g, r) = one 16-byte piece per tile; the vector is followed by zeros (EngineBase::up_f32)
        const int cbase = c * (16 * NT);
        float bv[4 * NT];
        ACH_UNROLL
        for (int t = 0; t < NT; ++t) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + cbase + chunk_channel(NT, t, g, 0));
            bv[t * 4] = b4[0]; bv[t * 4 + 1] = b4[1]; bv[t * 4 + 2] = b4[2]; bv[t * 4 + 3] = b4[3];
        }
        if constexpr (DEEP) {
            // k-loop over a ring of GEMM_DEPTH operand slots: the operands of steps s+1 .. s+DEPTH-1 are in flight while step s runs, and a
            // slot is refilled with step s+DEPTH right after its MFMAs are issued.  On these shapes a wave's lifetime is a chain of L2 round
            // trips (4 NT MFMAs = 64 cycles of work per 500+ cycle fetch); one step ahead — with the compiler copying "next" into "current"
            // registers, i.e. waiting for the fetch right behind the MFMAs — hid a tenth of it.  The loop is unrolled over the ring so that the
            // slots are plain registers.  DEEP is chosen per launch for K >= GEMM_DEEP_KSTEPS k-steps (launch_gemm): on the short k-loops of the
            // headline config the ring only costs registers and branches (-2 % frames/s when applied everywhere), on EN-S2 / MV-S2 it pays.
            constexpr int DEPTH = GEMM_DEPTH;
            uint4 xb[DEPTH][P], wb[DEPTH][NT];
            const uint4* wchunk = Wf + (long(c) * p.ksteps) * NT * 64 + lane;
            auto fill = [&](int d, int s) {
                ACH_UNROLL
                for (int q = 0; q < P; ++q) xb[d][q] = load_x(q, s);
                ACH_UNROLL
                for (int t = 0; t < NT; ++t) wb[d][t] = wchunk[(long(s) * NT + t) * 64];
            };
            ACH_UNROLL
            for (int d = 0; d < DEPTH; ++d)
                if (d < p.ksteps) fill(d, d);
            for (int s0 = 0; s0 < p.ksteps; s0 += DEPTH) {
                ACH_UNROLL
                for (int d = 0; d < DEPTH; ++d) {
                    const int s = s0 + d;
                    if (s >= p.ksteps) break;
                    uint4 xf[P];
                    ACH_UNROLL
                    for (int q = 0; q < P; ++q) xf[q] = xb[d][q];
                    if constexpr (LNTAP) {
                        ACH_UNROLL
                        for (int q = 0; q < P; ++q) ln_tap(q, s, xf[q]);
                    } else
                    if (p.ln) {
                        ACH_UNROLL
                        for (int q = 0; q < P; ++q) {
                            float v[8];
                            frag_unpack<T>(xf[q], v);
                            ACH_UNROLL
                            for (int j = 0; j < VEC; ++j) v[j] = (v[j] - mean[q]) * rstd[q];
                            xf[q] = frag_pack<T>(v);
                        }
                    }
                    ACH_UNROLL
                    for (int t = 0; t < NT; ++t)
                        ACH_UNROLL
                        for (int q = 0; q < P; ++q) mfma16<T>(wb[d][t], xf[q], acc[q][t]);
                    if (s + DEPTH < p.ksteps) fill(d, s + DEPTH);
                }
            }
        } else {
            // the operands of step s+1 requested before the MFMAs of step s are issued
            uint4 xn[P], wn[NT];
            ACH_UNROLL
            for (int q = 0; q < P; ++q) xn[q] = load_x(q, 0);
            {
                const uint4* wrow = Wf + (long(c) * p.ksteps) * NT * 64 + lane;
                ACH_UNROLL
                for (int t = 0; t < NT; ++t) wn[t] = wrow[t * 64];
            }
            for (int s = 0; s < p.ksteps; ++s) {
                uint4 xf[P], wf[NT];
                ACH_UNROLL
                for (int q = 0; q < P; ++q) xf[q] = xn[q];
                ACH_UNROLL
                for (int t = 0; t < NT; ++t) wf[t] = wn[t];
                if (s + 1 < p.ksteps) {
                    ACH_UNROLL
                    for (int q = 0; q < P; ++q) xn[q] = load_x(q, s + 1);
                    const uint4* wrow = Wf + (long(c) * p.ksteps + s + 1) * NT * 64 + lane;
                    ACH_UNROLL
                    for (int t = 0; t < NT; ++t) wn[t] = wrow[t * 64];
                }
                if constexpr (LNTAP) {
                    ACH_UNROLL
                    for (int q = 0; q < P; ++q) ln_tap(q, s, xf[q]);
                } else
                if (p.ln) {
                    ACH_UNROLL
                    for (int q = 0; q < P; ++q) {
                        float v[8];
                        frag_unpack<T>(xf[q], v);
                        ACH_UNROLL
                        for (int j = 0; j < VEC; ++j) v[j] = (v[j] - mean[q]) * rstd[q];
                        xf[q] = frag_pack<T>(v);
                    }
                }
                ACH_UNROLL
                for (int t = 0; t < NT; ++t)
                    ACH_UNROLL
                    for (int q = 0; q < P; ++q) mfma16<T>(wf[t], xf[q], acc[q][t]);
            }
        }

        // ---- epilogue: lane holds, for pixel px of every sub-tile, the channels chunk_channel(NT, t, g, r) of this chunk

        ACH_UNROLL
        for (int q = 0; q < P; ++q) {
            if (!valid[q]) continue;
            float o[4 * NT];
            ACH_UNROLL
            for (int t = 0; t < NT; ++t)
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) o[t * 4 + r] = acc[q][t][r] + bv[t * 4 + r];
            apply_act_n<T, 4 * NT>(o, p.act);
            const long m = mrow[q];
            if (p.out_nchw) {
                T* Y = static_cast<T*>(p.Y);
                const long b = m / p.HW, pix = m - b * p.HW;
                ACH_UNROLL
                for (int t = 0; t < NT; ++t)
                    ACH_UNROLL
                    for (int r = 0; r < 4; ++r) {
                        const int n = cbase + chunk_channel(NT, t, g, r);
                        if (n < p.N) st_user<T>(Y, (b * p.Ctot + p.coff + n) * p.HW + pix, o[t * 4 + r], p.out_nchw == 2);
                    }
                continue;
            }
            T* yrow = static_cast<T*>(p.Y) + m * p.ldy;
            const T* rrow = p.R ? static_cast<const T*>(p.R) + m * p.ldr : nullptr;
            if (NT >= 2) {
                ACH_UNROLL
                for (int j = 0; j < NT / 2; ++j) {
                    const int nb = cbase + j * 32 + g * 8;                  // 8 consecutive channels nb .. nb+7
                    if (nb >= p.N) continue;
                    float v8[8];
                    ACH_UNROLL
                    for (int i = 0; i < 8; ++i) v8[i] = o[j * 8 + i];       // tiles 2j (first 4) and 2j+1 (last 4)
                    if (p.vec_store && nb + 8 <= p.N) {
                        if (rrow) { float r8[8]; Store<T>::ld8(rrow + nb, r8); ACH_UNROLL for (int i = 0; i < 8; ++i) v8[i] += r8[i]; }
                        Store<T>::st8(yrow + nb, v8);
                    } else {
                        for (int i = 0; i < 8; ++i)
                            if (nb + i < p.N) {
                                float v = v8[i];
                                if (rrow) v += Store<T>::ld(rrow + nb + i);
                                Store<T>::st(yrow + nb + i, v);
                            }
                    }
                }
            } else {
                const int nb = cbase + g * 4;                               // 4 consecutive channels
                if (nb < p.N) {
                    float v4[4] = {o[0], o[1], o[2], o[3]};
                    if (p.vec_store && nb + 4 <= p.N) {
                        if (rrow) { float r4[4]; Store<T>::ld4(rrow + nb, r4); v4[0] += r4[0]; v4[1] += r4[1]; v4[2] += r4[2]; v4[3] += r4[3]; }
                        Store<T>::st4(yrow + nb, v4);
                    } else {
                        for (int i = 0; i < 4; ++i)
                            if (nb + i < p.N) {
                                float v = v4[i];
                                if (rrow) v += Store<T>::ld(rrow + nb + i);
                                Store<T>::st(yrow + nb + i, v);
                            }
                    }
                }
            }
        }
    }
}

#ifndef ACH_GEMM_WAVES
#define ACH_GEMM_WAVES 0
#endif
#if ACH_GEMM_WAVES > 0
#define ACH_GEMM_BOUNDS __launch_bounds__(256, ACH_GEMM_WAVES)
#else
#define ACH_GEMM_BOUNDS __launch_bounds__(256)
#endif
template <class T, int NT, int P, bool DEEP = false, bool LNTAP = false>
__global__ ACH_GEMM_BOUNDS void gemm_kernel(const GemmParams p) { f16_sat_mode<T>(); gemm_body<T, NT, P, DEEP, LNTAP>(p, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z); }

// Up to three independent GEMMs of the same tile shape in one launch (blockIdx.y = job): the three pyramid levels of the
// detection head run the same layer on maps of 1600 / 400 / 100 pixels — the small levels ride in the big level's launch
// instead of costing a latency-bound launch each.  Jobs have groups == 1.
struct GemmJobs { GemmParams p[3]; unsigned nbx[3], nbz[3]; int n; };
template <class T, int NT>
__global__ __launch_bounds__(256) void gemm_multi_kernel(const GemmJobs m) { f16_sat_mode<T>();
    const unsigned j = blockIdx.y;
    if (blockIdx.x >= m.nbx[j] || blockIdx.z >= m.nbz[j]) return;
    gemm_body<T, NT, 1>(m.p[j], blockIdx.x, m.nbx[j], 0u, blockIdx.z);
}

// ---- shared MLP + max over the rows of a group (PointNet: conv1d -> BN -> [ReLU] -> max over the N points).
// One workgroup owns (group, chunk of 16*NT output channels): it sweeps ALL rows of the group, 64 at a time (4 waves x 16),
// keeps the running maxima in registers, reduces the 16 pixels of a wave with DPP row operations and the 4 waves through
// LDS, and writes the result once — no atomics, no initialisation pass, no decode pass.  The [rows, N] activation (134 MB
// in fp32 for the 128->1024 STN layers at batch 64) is never materialised.
struct GemmMaxParams {
    const void* X; long ldx;
    const void* W; const float* bias;
    void* Y; long ldy;                  // [groups, N]
    int M_per_group, groups, K, N, nchunks, ksteps, act;
};
// KH > 0: the chunk's KH x NT weight fragments are loaded once and stay in registers for all the row tiles a wave walks
// (K <= 128: the 128 -> 1024 convs of the PointNet transforms, where re-reading 16 fragments per 16 rows made the kernel
// L1-bound at 190 TFLOP/s).
#ifndef ACH_COLMAX_WAVES
#define ACH_COLMAX_WAVES 0
#endif
#if ACH_COLMAX_WAVES > 0
#define ACH_COLMAX_BOUNDS __launch_bounds__(256, ACH_COLMAX_WAVES)
#else
#define ACH_COLMAX_BOUNDS __launch_bounds__(256)
#endif
template <class T, int NT, int KH>
__global__ ACH_COLMAX_BOUNDS void gemm_colmax_kernel(const GemmMaxParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC;
    constexpr int KC = 4 * VEC;
    __shared__ float red[4][16 * NT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    // flat grid (groups can exceed 65535), XCD-aware: the nchunks workgroups of a group all read the group's rows, so a group's chunks stay
    // in ONE XCD's L2 (round 3: in plain dispatch order a sample's 16 chunks went to all eight XCDs and the 128 -> 1024 convs of PointNet
    // fetched their activations 7.7x: profiles/r02_traffic_en_s0.json)
    const unsigned wgid = xcd_block(blockIdx.x, gridDim.x);
    const int c = int(wgid % unsigned(p.nchunks)), grp = int(wgid / unsigned(p.nchunks));
    const T* X = static_cast<const T*>(p.X) + long(grp) * p.M_per_group * p.ldx;
    const uint4* Wf = reinterpret_cast<const uint4*>(p.W) + long(c) * p.ksteps * NT * 64 + lane;
    float bv[4 * NT], cm[4 * NT];
    ACH_UNROLL
    for (int t = 0; t < NT; ++t)
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { const int n = c * (16 * NT) + chunk_channel(NT, t, g, r); bv[t * 4 + r] = n < p.N ? p.bias[n] : 0.f; cm[t * 4 + r] = -3.0e38f; }
    uint4 wh[KH > 0 ? KH : 1][NT];
    if (KH > 0) {
        ACH_UNROLL
        for (int s = 0; s < KH; ++s)
            ACH_UNROLL
            for (int t = 0; t < NT; ++t) wh[s][t] = Wf[(s * NT + t) * 64];
    }
    uint4 xn[KH > 0 ? KH : 1];                                    // KH > 0: rows of the NEXT tile, requested one tile ahead
    auto fetch = [&](int r0) {
        const int row = r0 + px;
        ACH_UNROLL
        for (int s = 0; s < (KH > 0 ? KH : 1); ++s) {
            const int k0 = s * KC + g * VEC;
            xn[s] = (row < p.M_per_group && k0 < p.K) ? *reinterpret_cast<const uint4*>(X + long(row) * p.ldx + k0) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if (KH > 0 && wave * 16 < p.M_per_group) fetch(wave * 16);
    for (int r0 = wave * 16; r0 < p.M_per_group; r0 += 64) {
        const int row = r0 + px;
        const bool valid = row < p.M_per_group;
        f32x4 acc[NT];
        ACH_UNROLL
        for (int t = 0; t < NT; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; acc[t][3] = 0.f; }
        if (KH > 0) {
            uint4 xf[KH > 0 ? KH : 1];
            ACH_UNROLL
            for (int s = 0; s < KH; ++s) xf[s] = xn[s];
            if (r0 + 64 < p.M_per_group) fetch(r0 + 64);
            ACH_UNROLL
            for (int s = 0; s < KH; ++s)
                ACH_UNROLL
                for (int t = 0; t < NT; ++t) mfma16<T>(wh[s][t], xf[s], acc[t]);
        } else
        for (int s = 0; s < p.ksteps; ++s) {
            const int k0 = s * KC + g * VEC;
            uint4 xf = make_uint4(0u, 0u, 0u, 0u);
            if (valid && k0 < p.K) xf = *reinterpret_cast<const uint4*>(X + long(row) * p.ldx + k0);
            ACH_UNROLL
            for (int t = 0; t < NT; ++t) mfma16<T>(Wf[(s * NT + t) * 64], xf, acc[t]);
        }
        if (valid) {
            float av[4 * NT];
            ACH_UNROLL
            for (int t = 0; t < NT; ++t)
                ACH_UNROLL
                for (int r = 0; r < 4; ++r) av[t * 4 + r] = acc[t][r] + bv[t * 4 + r];
            apply_act_n<float, 4 * NT>(av, p.act);
            ACH_UNROLL
            for (int e = 0; e < 4 * NT; ++e) cm[e] = fmaxf(cm[e], av[e]);
        }
    }
    ACH_UNROLL
    for (int t = 0; t < NT; ++t)
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) {
            const float m = row16_max(cm[t * 4 + r]);
            if (px == 0) red[wave][chunk_channel(NT, t, g, r)] = m;
        }
    __syncthreads();
    if (threadIdx.x < 16 * NT) {
        const int n = c * (16 * NT) + threadIdx.x;
        const float m = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
        if (n < p.N) Store<T>::st(static_cast<T*>(p.Y) + long(grp) * p.ldy + n, m);
    }
}

// The same layer for MANY SMALL groups (PointNet++: the max over the 32 samples of a ball, 16 384 balls at the first level of a batch of 64).  gemm_colmax_kernel gives every
// (group, chunk) a four-wave workgroup — two of the four waves have no rows, every workgroup fetches the chunk's weights and meets at a barrier for 32 rows of work: 56 us for a
// 2 GFLOP layer (round 5: profiles/r05_ops_en_s0_pn2.json), launch-rate bound.  Here a WAVE owns whole groups — GMAX_PER_WAVE consecutive groups of one chunk, the chunk's
// weight fragments fetched once per wave (registers for KH k-steps, L1 otherwise), a group's row tiles walked in order, the 16 rows of a tile reduced by DPP row operations,
// nothing through LDS, no barrier.  Per row the same MFMA sequence, bias and activation as gemm_colmax_kernel and a maximum is order-independent: bit-identical outputs.
constexpr int GMAX_PER_WAVE = 4;
template <class T, int NT, int KH>
__global__ __launch_bounds__(256) void gemm_groupmax_kernel(const GemmMaxParams p) { f16_sat_mode<T>();
    constexpr int VEC = Store<T>::VEC;
    constexpr int KC = 4 * VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, g = lane >> 4;
    const unsigned wgid = blockIdx.x;
    const int c = int(wgid % unsigned(p.nchunks));
    const long grp0 = (long(wgid / unsigned(p.nchunks)) * 4 + wave) * GMAX_PER_WAVE;
    if (grp0 >= p.groups) return;                                     // whole waves leave; no barrier below
    const uint4* Wf = reinterpret_cast<const uint4*>(p.W) + long(c) * p.ksteps * NT * 64 + lane;
    float bv[4 * NT];
    ACH_UNROLL
    for (int t = 0; t < NT; ++t)
        ACH_UNROLL
        for (int r = 0; r < 4; ++r) { const int n = c * (16 * NT) + chunk_channel(NT, t, g, r); bv[t * 4 + r] = n < p.N ? p.bias[n] : 0.f; }
    uint4 wh[KH > 0 ? KH : 1][NT];
    if (KH > 0) {
        ACH_UNROLL
        for (int s = 0; s < KH; ++s)
            ACH_UNROLL
            for (int t = 0; t < NT; ++t) wh[s][t] = Wf[(s * NT + t) * 64];
    }
    for (int gi = 0; gi < GMAX_PER_WAVE; ++gi) {
        const long grp = grp0 + gi;
        if (grp >= p.groups) break;
        const T* X = static_cast<const T*>(p.X) + grp * p.M_per_group * p.ldx;
        float cm[4 * NT];
        ACH_UNROLL
        for (int e = 0; e < 4 * NT; ++e) cm[e] = -3.0e38f;
        for (int r0 = 0; r0 < p.M_per_group; r0 += 16) {
            const int row = r0 + px;
            const bool valid = row < p.M_per_group;
            f32x4 acc[NT];
            ACH_UNROLL
            for (int t = 0; t < NT; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; acc[t][3] = 0.f; }
            if (KH > 0) {
                uint4 xf[KH > 0 ? KH : 1];
                ACH_UNROLL
                for (int s = 0; s < KH; ++s) {
                    const int k0 = s * KC + g * VEC;
                    xf[s] = (valid && k0 < p.K) ? *reinterpret_cast<const uint4*>(X + long(row) * p.ldx + k0) : make_uint4(0u, 0u, 0u, 0u);
                }
                ACH_UNROLL
                for (int s = 0; s < KH; ++s)
                    ACH_UNROLL
                    for (int t = 0; t < NT; ++t) mfma16<T>(wh[s][t], xf[s], acc[t]);
            } else
            for (int s = 0; s < p.ksteps; ++s) {
                const int k0 = s * KC + g * VEC;
                uint4 xf = make_uint4(0u, 0u, 0u, 0u);
                if (valid && k0 < p.K) xf = *reinterpret_cast<const uint4*>(X + long(row) * p.ldx + k0);
                ACH_UNROLL
                for (int t = 0; t < NT; ++t) mfma16<T>(Wf[(s * NT + t) * 64], xf, acc[t]);
            }
            if (valid) {
                float av[4 * NT];
                ACH_UNROLL
                for (int t = 0; t < NT; ++t)
                    ACH_UNROLL
                    for (int r = 0; r < 4; ++r) av[t * 4 + r] = acc[t][r] + bv[t * 4 + r];
                apply_act_n<float, 4 * NT>(av, p.act);
                ACH_UNROLL
                for (int e = 0; e < 4 * NT; ++e) cm[e] = fmaxf(cm[e], av[e]);
            }
        }
        T* Y = static_cast<T*>(p.Y) + grp * p.ldy + c * (16 * NT);
        ACH_UNROLL
        for (int t = 0; t < NT; ++t)
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) {
                const float m = row16_max(cm[t * 4 + r]);
                const int n = chunk_channel(NT, t, g, r);
                if (px == 0 && c * (16 * NT) + n < p.N) Store<T>::st(Y + n, m);
            }
    }
}

template <class T>
inline void launch_gemm(const GemmParams& p, int NT, int P, hipStream_t stream) {
    const dim3 grid(unsigned(cdivl(p.M_per_group, 64L * P)), unsigned(p.groups), unsigned(cdiv(p.nchunks, p.chunks_per_block))), block(256);
    if (p.ln == 2) {           // per-tap LayerNorm of a 2x2 patchify conv (the engine only asks for it with NT = 4, P = 1)
        if (p.ksteps >= GEMM_DEEP_KSTEPS) ACH_LAUNCH((gemm_kernel<T, 4, 1, true, true>), grid, block, stream, p);
        else ACH_LAUNCH((gemm_kernel<T, 4, 1, false, true>), grid, block, stream, p);
        return;
    }
    if (P == 1 && p.ksteps >= GEMM_DEEP_KSTEPS) {
        if (NT == 1) { ACH_LAUNCH((gemm_kernel<T, 1, 1, true>), grid, block, stream, p); return; }
        if (NT == 2) { ACH_LAUNCH((gemm_kernel<T, 2, 1, true>), grid, block, stream, p); return; }
        if (NT == 4) { ACH_LAUNCH((gemm_kernel<T, 4, 1, true>), grid, block, stream, p); return; }
    }
#define ACH_GEMM_CASE(nt, pp) if (NT == nt && P == pp) { ACH_LAUNCH((gemm_kernel<T, nt, pp>), grid, block, stream, p); return; }
    ACH_GEMM_CASE(1, 1) ACH_GEMM_CASE(1, 2) ACH_GEMM_CASE(1, 4)
    ACH_GEMM_CASE(2, 1) ACH_GEMM_CASE(2, 2) ACH_GEMM_CASE(2, 4)
    ACH_GEMM_CASE(4, 1) ACH_GEMM_CASE(4, 2) ACH_GEMM_CASE(4, 4)
#undef ACH_GEMM_CASE
}

}  // namespace ach
