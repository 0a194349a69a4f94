// k_train.h — first training-mode kernels: the PointNet "shared MLP" layer
//     y = relu(BatchNorm1d(Conv1d_k1(x)))        x [B, Cin, N] -> y [B, Cout, N]      (pointnet_utils.py:29-31, 69-71, 124-127;
// pointnet_sem_seg.py:31-33) in TRAINING mode — batch statistics over (B, N), running statistics updated — forward and backward.
// The reference trains it through ATen autograd (utils/utils_fit.py:37-166); here the same arithmetic is four kernels in fp32:
//   train_gemm      C[b] (+)= op(A[b]) op(B[b]) on fp32 MFMA (v_mfma_f32_16x16x4_f32) or with operands rounded to bf16 (ach_train_set_gemm_precision),
//                   LDS-staged 64 x 64 x 32 tiles, any transposition,
//                   optional reduction over the batch (the weight gradient sums over samples) and row bias
//   bn_stats        per channel mean / biased variance over (B, N), two passes (mean, then centred squares)
//   bn_relu_fwd     y = [relu](gamma * (z - mean) * rstd + beta)
//   bn_relu_bwd_*   g = dy * (y > 0); dbeta = sum g, dgamma = sum g * xhat;  dz = gamma * rstd * (g - dbeta / M - xhat * dgamma / M)
// Inference keeps its own folded / fused path; these exist so that `.train()` can be built block by block.
#pragma once
#include "ach_platform.h"

namespace ach {

struct TrainGemmParams {
    const float* A; const float* B; float* C; const float* bias;   // bias: per ROW of C (output channel) or nullptr
    int M, N, K;                      // C is M x N
    long lda, ldb, ldc;               // leading dimensions of the STORED matrices (row-major)
    long sA, sB, sC;                  // batch strides (0: shared)
    int transA, transB;               // stored A is K x M / stored B is N x K
    int batch, reduce_batch;          // reduce_batch: C = sum_b op(A[b]) op(B[b])  (one output, grid.z = 1)
    int accumulate;                   // C += instead of C =
    int ksplit; float* ws;            // ksplit > 1 (reduce_batch only): workgroup z reduces its share of the (batch, k-tile) range into ws[z][M][N];
                                      // train_gemm_reduce_kernel then sums the partials in order (deterministic) and applies bias / accumulate.
                                      // A weight gradient reduces over B x H x W (819 200 at batch 8, 320 x 320) into a 32 x 48 tile: without the split one
                                      // workgroup did all of it (72 ms per call, 90 % of a training step)
};

// 256 threads = 4 waves; wave w owns the 32 x 32 quadrant (w >> 1, w & 1) of the 64 x 64 block tile: 2 x 2 MFMA tiles.
// Round 5: the operands' element type on the matrix cores is a template parameter —
//   T = float : v_mfma_f32_16x16x4_f32 on the fp32 values (the exact-arithmetic mode, the one the float64 reference-step test holds to 5e-3)
//   T = bf16_t: the fp32 values are rounded to bf16 WHILE THEY ARE STAGED into LDS (no 16-bit copy of anything exists in memory), products on
//               v_mfma_f32_16x16x16_bf16 pairs, accumulation, bias and output in fp32 — what the reference's default loop asks for with
//               torch.cuda.amp.autocast (utils/utils_fit.py:120-166), minus its 16-bit activations.  bf16, not fp16: GradScaler multiplies the gradients
//               by 2^16, which an fp16 operand would turn into infinities; bf16 keeps fp32's exponent range.
// and the staging is 16-byte loads along whichever index the stored matrix is contiguous in (scalar loads only where leading dimension, batch
// stride or base address are not multiples of four floats, or a tile crosses the matrix edge), a k-tile of 32, two LDS buffers (ONE barrier per
// k-tile, the next tile's global loads in flight while the matrix cores work on this one).  LDS tiles are [row][k] for both operands, so every
// fragment is one 16-byte LDS read (k = 4 lk + j for fp32 — any bijection is legal as long as A and B agree — and k = 8 lk + j for bf16).
constexpr int TRAIN_GEMM_TK = 32;

static __device__ __forceinline__ f32x4 tg_load4(const float* ptr, int nvalid, bool vec) {        // four consecutive floats, zeros beyond `nvalid`
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (nvalid >= 4 && vec) v = *reinterpret_cast<const f32x4*>(ptr);
    else {
        if (nvalid > 0) v[0] = ptr[0];
        if (nvalid > 1) v[1] = ptr[1];
        if (nvalid > 2) v[2] = ptr[2];
        if (nvalid > 3) v[3] = ptr[3];
    }
    return v;
}
template <class T> static __device__ __forceinline__ void tg_store_pair(T* dst, float a, float b) {  // elements k, k + 1 of one row (k even)
    if constexpr (is_h16<T>::value) *reinterpret_cast<uint32_t*>(dst) = H16<T>::pack(a, b);
    else { dst[0] = a; dst[1] = b; }
}

// one operand's ROWS x 32 tile (ROWS = 64 or 32): registers <- global (`fetch`), LDS <- registers (`stash`).  `kcontig`: the stored matrix is contiguous
// along k (A as stored M x K, B as stored N x K); otherwise along the row index (A stored K x M, B stored K x N).
//   kcontig            : thread = (row tid >> 3 [+ 32 when ROWS = 64], k quad tid & 7)  -> one 8- / 16-byte LDS store per vector
//   else, ROWS = 64    : thread = (k pair tid >> 4, row quad tid & 15), rows k and k + 1 -> four (k, k + 1) pair stores
//   else, ROWS = 32    : thread = (k tid >> 3, row quad tid & 7)                         -> four single-element stores
struct TgOperand { const float* base; long ld; int rows, r0; bool kcontig, vec; };
template <int ROWS> static __device__ __forceinline__ void tg_fetch(const TgOperand& o, const float* mat, int K, int k0, int tid, f32x4& v0, f32x4& v1) {
    if (o.kcontig) {
        const int kq = tid & 7, k = k0 + 4 * kq, r = o.r0 + (tid >> 3);
        v0 = tg_load4(mat + long(r) * o.ld + k, r < o.rows ? K - k : 0, o.vec);
        if constexpr (ROWS == 64) v1 = tg_load4(mat + long(r + 32) * o.ld + k, r + 32 < o.rows ? K - k : 0, o.vec);
    } else if constexpr (ROWS == 64) {
        const int k = k0 + 2 * (tid >> 4), r = o.r0 + 4 * (tid & 15);
        v0 = tg_load4(mat + long(k) * o.ld + r, k < K ? o.rows - r : 0, o.vec);
        v1 = tg_load4(mat + long(k + 1) * o.ld + r, k + 1 < K ? o.rows - r : 0, o.vec);
    } else {
        const int k = k0 + (tid >> 3), r = o.r0 + 4 * (tid & 7);
        v0 = tg_load4(mat + long(k) * o.ld + r, k < K ? o.rows - r : 0, o.vec);
    }
}
template <class T, int ROWS, int PITCH> static __device__ __forceinline__ void tg_stash(const TgOperand& o, int tid, const f32x4 v0, const f32x4 v1, T (*S)[PITCH]) {
    if (o.kcontig) {
        const float f0[4] = {v0[0], v0[1], v0[2], v0[3]};
        Store<T>::st4(&S[tid >> 3][4 * (tid & 7)], f0);
        if constexpr (ROWS == 64) { const float f1[4] = {v1[0], v1[1], v1[2], v1[3]}; Store<T>::st4(&S[(tid >> 3) + 32][4 * (tid & 7)], f1); }
    } else if constexpr (ROWS == 64) {
        const int kp = tid >> 4, r = 4 * (tid & 15);
        tg_store_pair<T>(&S[r][2 * kp], v0[0], v1[0]);
        tg_store_pair<T>(&S[r + 1][2 * kp], v0[1], v1[1]);
        tg_store_pair<T>(&S[r + 2][2 * kp], v0[2], v1[2]);
        tg_store_pair<T>(&S[r + 3][2 * kp], v0[3], v1[3]);
    } else {
        const int k = tid >> 3, r = 4 * (tid & 7);
        Store<T>::st(&S[r][k], v0[0]); Store<T>::st(&S[r + 1][k], v0[1]); Store<T>::st(&S[r + 2][k], v0[2]); Store<T>::st(&S[r + 3][k], v0[3]);
    }
}

// TM x TN block tile (each 64 or 32; 2 x 2 waves, a wave owns TM/2 x TN/2).  The 32-row forms are for the STREAMING shapes of a training step — a weight
// gradient is a 16 x 32 output reduced over 32 x 102 400 positions, a decoder's 1x1 convolution a 16 x 102 400 output with K = 32 — where what matters is
// bytes in flight per compute unit: a 64-row tile leaves most of the staging threads without a row to load and its LDS allows four workgroups per compute unit.
template <class T, int TM, int TN>
static __global__ __launch_bounds__(256) void train_gemm_kernel(const TrainGemmParams p) {
    constexpr int TK = TRAIN_GEMM_TK, VEC = Store<T>::VEC, PITCH = TK + VEC, KCH = TK / (4 * VEC), MI = TM / 32, NJ = TN / 32;
    __shared__ __attribute__((aligned(16))) T As[2][TM][PITCH];
    __shared__ __attribute__((aligned(16))) T Bs[2][TN][PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc[MI][NJ];
    ACH_UNROLL
    for (int i = 0; i < MI; ++i)
        ACH_UNROLL
        for (int j = 0; j < NJ; ++j) { acc[i][j][0] = acc[i][j][1] = acc[i][j][2] = acc[i][j][3] = 0.f; }
    // the reduction range as (batch, k-tile) pairs, linearised: everything for one output of a batched product, the whole batch for a
    // batch-reduced one, or this workgroup's share of it when split
    const long ktiles = (p.K + TK - 1) / TK;
    long t_lo, t_hi;
    if (!p.reduce_batch) { t_lo = long(blockIdx.z) * ktiles; t_hi = t_lo + ktiles; }
    else if (p.ksplit > 1) { const long T2 = long(p.batch) * ktiles, per = (T2 + p.ksplit - 1) / p.ksplit; t_lo = long(blockIdx.z) * per; t_hi = t_lo + per < T2 ? t_lo + per : T2; }
    else { t_lo = 0; t_hi = long(p.batch) * ktiles; }
    const auto aligned4 = [](const float* q, long ld, long stride) { return ((reinterpret_cast<uintptr_t>(q) & 15u) == 0) && (ld % 4 == 0) && (stride % 4 == 0); };
    const TgOperand oa{p.A, p.lda, p.M, m0, !p.transA, aligned4(p.A, p.lda, p.sA)};
    const TgOperand ob{p.B, p.ldb, p.N, n0, p.transB != 0, aligned4(p.B, p.ldb, p.sB)};
    f32x4 ra0, ra1, rb0, rb1;
    ra1 = rb1 = f32x4{0.f, 0.f, 0.f, 0.f};
#define ACH_TG_FETCH(t)  { const long t_ = (t); const int b_ = int(t_ / ktiles), k0_ = int(t_ - long(b_) * ktiles) * TK; \
                           tg_fetch<TM>(oa, p.A + long(b_) * p.sA, p.K, k0_, tid, ra0, ra1); tg_fetch<TN>(ob, p.B + long(b_) * p.sB, p.K, k0_, tid, rb0, rb1); }
    if (t_lo < t_hi) {
        ACH_TG_FETCH(t_lo);
        tg_stash<T, TM, PITCH>(oa, tid, ra0, ra1, As[0]);
        tg_stash<T, TN, PITCH>(ob, tid, rb0, rb1, Bs[0]);
    }
    __syncthreads();
    int cur = 0;
    for (long t = t_lo; t < t_hi; ++t) {
        const bool more = t + 1 < t_hi;
        if (more) ACH_TG_FETCH(t + 1);
        ACH_UNROLL
        for (int kc = 0; kc < KCH; ++kc) {
            uint4 a[MI], bb[NJ];
            ACH_UNROLL
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const uint4*>(&As[cur][wm + 16 * i + li][kc * 4 * VEC + lk * VEC]);
            ACH_UNROLL
            for (int j = 0; j < NJ; ++j) bb[j] = *reinterpret_cast<const uint4*>(&Bs[cur][wn + 16 * j + li][kc * 4 * VEC + lk * VEC]);
            ACH_UNROLL
            for (int i = 0; i < MI; ++i)
                ACH_UNROLL
                for (int j = 0; j < NJ; ++j) mfma16_pair<T>(a[i], bb[j], acc[i][j]);        // (long-lived matrix-instruction waves: the pair form)
        }
        if (more) {
            tg_stash<T, TM, PITCH>(oa, tid, ra0, ra1, As[cur ^ 1]);
            tg_stash<T, TN, PITCH>(ob, tid, rb0, rb1, Bs[cur ^ 1]);
        }
        __syncthreads();
        cur ^= 1;
    }
    const bool partial = p.reduce_batch && p.ksplit > 1;
    float* C = partial ? p.ws + long(blockIdx.z) * p.M * p.N : p.C + (p.reduce_batch ? 0L : long(blockIdx.z) * p.sC);
    const long ldc = partial ? long(p.N) : p.ldc;
    ACH_UNROLL
    for (int i = 0; i < MI; ++i)
        ACH_UNROLL
        for (int j = 0; j < NJ; ++j)
            ACH_UNROLL
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + 16 * i + lk * 4 + r, gn = n0 + wn + 16 * j + li;
                if (gm < p.M && gn < p.N) {
                    float* c = C + long(gm) * ldc + gn;
                    if (partial) { *c = acc[i][j][r]; continue; }
                    const float v = acc[i][j][r] + (p.bias ? p.bias[gm] : 0.f);
                    *c = p.accumulate ? *c + v : v;
                }
            }
#undef ACH_TG_FETCH
}
// C = [C +] bias + sum_z ws[z].  One 16-lane group per output element: lane j sums z = j, j + 16, ... (ascending), then the group
// folds 16 -> 1 in a fixed butterfly: a deterministic order, and 16 partial loads in flight per element instead of one thread walking
// up to 512 strided partials (that version took 35-120 us per weight gradient: 10 % of a training step).
// (Round 5: the 16 partial-sum walkers of an element used to be the 16 LANES of a group — 16 gathers of 4 bytes a step; now thread (j, e) of the workgroup walks z = j, j + 16, ... for
//  element i0 + e, so a 16-lane group reads 64 contiguous bytes, and the 16 walkers of an element meet in LDS.  The same sums in the same order: bit-identical.)
static __global__ __launch_bounds__(256) void train_gemm_reduce_kernel(const TrainGemmParams p) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, j = threadIdx.x >> 4;
    const long i = long(blockIdx.x) * 16 + e;
    const long MN = long(p.M) * p.N;
    float v = 0.f;
    if (i < MN)
        for (int z = j; z < p.ksplit; z += 16) v += p.ws[long(z) * MN + i];
    red[j][e] = v;
    __syncthreads();
    if (j != 0 || i >= MN) return;
    float a[8];
    ACH_UNROLL
    for (int q = 0; q < 8; ++q) a[q] = red[q][e] + red[q + 8][e];                     // the butterfly's xor-8 step ...
    v = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));            // ... and its xor-4, xor-2, xor-1 steps as lane 0 saw them
    const int gm = int(i / p.N), gn = int(i - long(gm) * p.N);
    v += p.bias ? p.bias[gm] : 0.f;
    float* c = p.C + long(gm) * p.ldc + gn;
    *c = p.accumulate ? *c + v : v;
}

// Walking the (B, N) positions of ONE channel of a [B, C, N] tensor with a stride of 256: position i = b N + n.  The kernels below used to form
// ((i / N) * C + c) * N + i % N per element — two 64-bit divisions (dozens of instructions each on this target, which has no integer divider, even through the compiler's 32-bit bypass) for one to three loads;
// the reductions were 16 % of a batch-32 training step.  One division at the start, then n += 256 with a carry into b.
// positions per slice of a channel's (B, N) range cut into S slices: whole quads when N is a multiple of four (the quad walks below); trailing slices may then be empty
__host__ __device__ __forceinline__ long bn_slice_len(long total, int S, int N) { const long per = (total + S - 1) / S; return (N & 3) == 0 ? ((per + 3) & ~3L) : per; }
struct BnWalk {
    long b; int n, N;
    __device__ __forceinline__ BnWalk(long i, int N_) : b(i / N_), n(int(i - (i / N_) * N_)), N(N_) {}
    __device__ __forceinline__ long offset(int C, int c) const { return (b * C + c) * long(N) + n; }
    __device__ __forceinline__ void step() {
        n += 256;
        if (N >= 256) { if (n >= N) { n -= N; ++b; } }
        else { const int q = n / N; b += q; n -= q * N; }
    }
};

// ---- BatchNorm over (B, N) of z [B, C, N]: one workgroup per channel
struct BnStatsParams { const float* Z; float* mean; float* var; int B, C, N; };        // var: biased
static __global__ __launch_bounds__(256) void train_bn_stats_kernel(const BnStatsParams p) {
    __shared__ float red[256];
    __shared__ float s_mean;
    const int c = blockIdx.x;
    const long total = long(p.B) * p.N;
    float s = 0.f;
    const bool quad = (p.N & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Z) & 15u) == 0;      // four positions per step, 16-byte loads from a 16-byte aligned tensor (see train_bn_relu_bwd_reduce_kernel)
    if (quad) { BnWalk w(threadIdx.x, p.N >> 2); for (long i = threadIdx.x; i < (total >> 2); i += 256, w.step()) { const float4 z = *reinterpret_cast<const float4*>(p.Z + (w.b * p.C + c) * long(p.N) + 4L * w.n); s += (z.x + z.y) + (z.z + z.w); } }
    else { BnWalk w(threadIdx.x, p.N); for (long i = threadIdx.x; i < total; i += 256, w.step()) s += p.Z[w.offset(p.C, c)]; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if (int(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) s_mean = red[0] / float(total);
    __syncthreads();
    const float mean = s_mean;
    float q = 0.f;
    if (quad) {
        BnWalk w(threadIdx.x, p.N >> 2);
        for (long i = threadIdx.x; i < (total >> 2); i += 256, w.step()) {
            const float4 z = *reinterpret_cast<const float4*>(p.Z + (w.b * p.C + c) * long(p.N) + 4L * w.n);
            const float d0 = z.x - mean, d1 = z.y - mean, d2 = z.z - mean, d3 = z.w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    } else { BnWalk w(threadIdx.x, p.N); for (long i = threadIdx.x; i < total; i += 256, w.step()) { const float d = p.Z[w.offset(p.C, c)] - mean; q += d * d; } }
    __syncthreads();
    red[threadIdx.x] = q;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if (int(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) { p.mean[c] = mean; p.var[c] = red[0] / float(total); }
}

// The same statistics with the (B, N) range of a channel cut into S slices (grid C x S) when it is long — at batch 8 a 320 x 320 map is
// 819 200 values per channel for one workgroup (2.5 ms per launch; with the backward reduction 2/3 of a training step once the weight
// gradients were split).  pass 0: partial sums; pass 1: partial centred squares around the mean that every workgroup re-derives from the
// S partial sums (same order everywhere: deterministic); finalize: mean, biased variance.
struct BnSliceParams { const float* Z; float* ws; float* mean; float* var; int B, C, N, S; };       // ws: [2][C][S]
template <int PASS>
static __global__ __launch_bounds__(256) void train_bn_slice_kernel(const BnSliceParams p) {
    __shared__ float red[256];
    const int c = blockIdx.x, sl = blockIdx.y;
    const long total = long(p.B) * p.N, per = (total + p.S - 1) / p.S;
    const long lo = long(sl) * per, hi = lo + per < total ? lo + per : total;
    float mean = 0.f;
    if (PASS == 1) { for (int j = 0; j < p.S; ++j) mean += p.ws[long(c) * p.S + j]; mean /= float(total); }
    float a = 0.f;
    BnWalk w(lo + threadIdx.x, p.N);
    for (long i = lo + threadIdx.x; i < hi; i += 256, w.step()) {
        const float v = p.Z[w.offset(p.C, c)] - mean;
        a += PASS == 0 ? v : v * v;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if (int(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) p.ws[(long(PASS) * p.C + c) * p.S + sl] = red[0];
}
// Round 5: the two passes of a slice in ONE launch.  A slice is ~16 K values (64 KB): the workgroup sums it, derives ITS OWN mean, and walks it again for the squares centred on that
// mean — the second walk comes out of the L2, the tensor is read from HBM once — and the finalize kernel combines the slices exactly (Chan et al.): M2 = sum_s [M2_s + n_s (mean_s - mean)^2].
// Numerically a two-pass variance, like the two launches above, which it replaces.  Measured: 50.5 us per call against 31.2 + 23.1 — the second walk is NOT free (other
// workgroups' slices push it out of the L2 at these sizes); what it saves is 61 launches per step (batch 8 is host-bound).
static __global__ __launch_bounds__(256) void train_bn_slice2_kernel(const BnSliceParams p) {
    __shared__ float red[256];
    __shared__ float s_mean;
    const int c = blockIdx.x, sl = blockIdx.y;
    const long total = long(p.B) * p.N, per = bn_slice_len(total, p.S, p.N);
    const long lo = long(sl) * per < total ? long(sl) * per : total, hi = lo + per < total ? lo + per : total;
    float a = 0.f;
    const bool quad = (p.N & 3) == 0 && (lo & 3) == 0 && (hi & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Z) & 15u) == 0;          // four positions per step, 16-byte loads (see train_bn_relu_bwd_reduce_kernel)
    if (quad) {
        BnWalk w((lo >> 2) + threadIdx.x, p.N >> 2);
        for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += 256, w.step()) { const float4 z = *reinterpret_cast<const float4*>(p.Z + (w.b * p.C + c) * long(p.N) + 4L * w.n); a += (z.x + z.y) + (z.z + z.w); }
    } else { BnWalk w(lo + threadIdx.x, p.N); for (long i = lo + threadIdx.x; i < hi; i += 256, w.step()) a += p.Z[w.offset(p.C, c)]; }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if (int(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) { s_mean = hi > lo ? red[0] / float(hi - lo) : 0.f; p.ws[long(c) * p.S + sl] = red[0]; }
    __syncthreads();
    const float m = s_mean;
    float q = 0.f;
    if (quad) {
        BnWalk w((lo >> 2) + threadIdx.x, p.N >> 2);
        for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += 256, w.step()) {
            const float4 z = *reinterpret_cast<const float4*>(p.Z + (w.b * p.C + c) * long(p.N) + 4L * w.n);
            const float d0 = z.x - m, d1 = z.y - m, d2 = z.z - m, d3 = z.w - m;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    } else { BnWalk w(lo + threadIdx.x, p.N); for (long i = lo + threadIdx.x; i < hi; i += 256, w.step()) { const float d = p.Z[w.offset(p.C, c)] - m; q += d * d; } }
    __syncthreads();
    red[threadIdx.x] = q;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if (int(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) p.ws[(long(p.C) + c) * p.S + sl] = red[0];
}
static __global__ __launch_bounds__(256) void train_bn_slice2_finalize_kernel(const BnSliceParams p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    const long total = long(p.B) * p.N, per = bn_slice_len(total, p.S, p.N);
    float sum = 0.f;
    for (int j = 0; j < p.S; ++j) sum += p.ws[long(c) * p.S + j];
    const float mean = sum / float(total);
    float m2 = 0.f;
    for (int j = 0; j < p.S; ++j) {
        const long lo = long(j) * per < total ? long(j) * per : total, hi = lo + per < total ? lo + per : total;
        if (hi <= lo) continue;
        const float nj = float(hi - lo), dm = p.ws[long(c) * p.S + j] / nj - mean;
        m2 += p.ws[(long(p.C) + c) * p.S + j] + nj * dm * dm;
    }
    p.mean[c] = mean; p.var[c] = m2 / float(total);
}
static __global__ __launch_bounds__(256) void train_bn_slice_finalize_kernel(const BnSliceParams p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    const float total = float(long(p.B) * p.N);
    float m = 0.f, q = 0.f;
    for (int j = 0; j < p.S; ++j) { m += p.ws[long(c) * p.S + j]; q += p.ws[(long(p.C) + c) * p.S + j]; }
    p.mean[c] = m / total; p.var[c] = q / total;
}

// nn.BatchNorm's running estimates after a training forward: running <- (1 - m) running + m batch (the variance unbiased: `unbias` = M / (M - 1)).  One launch in place of the four
// element-wise torch launches per layer (mul_, add_, mul_, add_: 320 launches of a training step's ~2 000, which at batch 32 is bound by the HOST's launch rate since round 5).
struct BnRunningParams { const float* mean; const float* var; float* running_mean; float* running_var; int C; float momentum, unbias; };
static __global__ __launch_bounds__(256) void train_bn_running_kernel(const BnRunningParams p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * p.mean[c];
    p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + (p.momentum * p.unbias) * p.var[c];
}

struct BnReluFwdParams { const float* Z; const float* mean; const float* var; const float* gamma; const float* beta; float* Y; int B, C, N; float eps; int relu; };
static __global__ __launch_bounds__(256) void train_bn_relu_fwd_kernel(const BnReluFwdParams p) {
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * p.C * p.N) return;
    const int c = long(p.B) * p.C * p.N < (1L << 31) ? int((unsigned(idx) / unsigned(p.N)) % unsigned(p.C)) : int((idx / p.N) % p.C);      // (64-bit divisions only where they are needed)
    const float v = p.gamma[c] * ((p.Z[idx] - p.mean[c]) * (1.0f / sqrtf(p.var[c] + p.eps))) + p.beta[c];
    p.Y[idx] = (p.relu && v < 0.f) ? 0.f : v;
}

// the same with one (sample, channel) plane per blockIdx.y and four positions per thread (N a multiple of four): no index division, the channel's constants formed once per thread
// from uniform loads, 16-byte loads and stores
static __global__ __launch_bounds__(256) void train_bn_relu_fwd4_kernel(const BnReluFwdParams p) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= unsigned(p.N >> 2)) return;
    const long plane = blockIdx.y;
    const int c = int(plane % p.C);
    const float mean = p.mean[c], beta = p.beta[c], g = p.gamma[c], rstd = 1.0f / sqrtf(p.var[c] + p.eps);
    const long off = plane * long(p.N) + 4L * q;
    const float4 z = *reinterpret_cast<const float4*>(p.Z + off);
    const float zi[4] = {z.x, z.y, z.z, z.w};
    float o[4];
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) { const float v = g * ((zi[i] - mean) * rstd) + beta; o[i] = (p.relu && v < 0.f) ? 0.f : v; }          // (the per-element kernel's expression, term for term)
    *reinterpret_cast<float4*>(p.Y + off) = make_float4(o[0], o[1], o[2], o[3]);
}

// backward, step 1: per channel  dbeta = sum g,  dgamma = sum g * xhat   with g = dy * (y > 0 | no relu)
struct BnReluBwdParams {
    const float* Z; const float* Y; const float* dY; const float* mean; const float* var; const float* gamma;
    float* dgamma; float* dbeta; float* dZ; int B, C, N; float eps; int relu;
    int S; float* ws;                 // S > 1: grid C x S, slice partials into ws [2][C][S], summed in order by train_bn_relu_bwd_finalize_kernel
};
static __global__ __launch_bounds__(256) void train_bn_relu_bwd_reduce_kernel(const BnReluBwdParams p) {
    __shared__ float r0[256], r1[256];
    const int c = blockIdx.x;
    const long total = long(p.B) * p.N;
    const long per = p.S > 1 ? bn_slice_len(total, p.S, p.N) : total, lo0 = p.S > 1 ? long(blockIdx.y) * per : 0, lo = lo0 < total ? lo0 : total, hi = lo + per < total ? lo + per : total;
    const float mean = p.mean[c], rstd = 1.0f / sqrtf(p.var[c] + p.eps);
    float sb = 0.f, sg = 0.f;
    if ((p.N & 3) == 0 && (lo & 3) == 0 && (hi & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.Z) | reinterpret_cast<uintptr_t>(p.Y) | reinterpret_cast<uintptr_t>(p.dY)) & 15u) == 0) {      // (a view at an odd element offset takes the scalar walk)
        // four positions per step with 16-byte loads (N and the slice bounds are multiples of four: a quad never straddles a sample): a quarter of the load instructions, four
        // times the bytes in flight per thread — this reduction reads three tensors and was at 2.6 TB/s with scalar loads
        BnWalk w((lo >> 2) + threadIdx.x, p.N >> 2);
        for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += 256, w.step()) {
            const long o = (w.b * p.C + c) * long(p.N) + 4L * w.n;
            const float4 z = *reinterpret_cast<const float4*>(p.Z + o), y = *reinterpret_cast<const float4*>(p.Y + o), d = *reinterpret_cast<const float4*>(p.dY + o);
            const float zi[4] = {z.x, z.y, z.z, z.w}, yi[4] = {y.x, y.y, y.z, y.w}, di[4] = {d.x, d.y, d.z, d.w};
            ACH_UNROLL
            for (int q = 0; q < 4; ++q) { const float g = (p.relu && !(yi[q] > 0.f)) ? 0.f : di[q]; sb += g; sg += g * ((zi[q] - mean) * rstd); }
        }
    } else {
    BnWalk w(lo + threadIdx.x, p.N);
    for (long i = lo + threadIdx.x; i < hi; i += 256, w.step()) {
        const long o = w.offset(p.C, c);
        const float g = (p.relu && !(p.Y[o] > 0.f)) ? 0.f : p.dY[o];
        sb += g; sg += g * ((p.Z[o] - mean) * rstd);
    }
    }
    r0[threadIdx.x] = sb; r1[threadIdx.x] = sg;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if (int(threadIdx.x) < st) { r0[threadIdx.x] += r0[threadIdx.x + st]; r1[threadIdx.x] += r1[threadIdx.x + st]; } __syncthreads(); }
    if (threadIdx.x == 0) {
        if (p.S > 1) { p.ws[long(c) * p.S + blockIdx.y] = r0[0]; p.ws[(long(p.C) + c) * p.S + blockIdx.y] = r1[0]; }
        else { p.dbeta[c] = r0[0]; p.dgamma[c] = r1[0]; }
    }
}
static __global__ __launch_bounds__(256) void train_bn_relu_bwd_finalize_kernel(const BnReluBwdParams p) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    float sb = 0.f, sg = 0.f;
    for (int j = 0; j < p.S; ++j) { sb += p.ws[long(c) * p.S + j]; sg += p.ws[(long(p.C) + c) * p.S + j]; }
    p.dbeta[c] = sb; p.dgamma[c] = sg;
}
// step 2: dz = gamma * rstd * (g - dbeta / M - xhat * dgamma / M)
static __global__ __launch_bounds__(256) void train_bn_relu_bwd_apply_kernel(const BnReluBwdParams p) {
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * p.C * p.N) return;
    const int c = long(p.B) * p.C * p.N < (1L << 31) ? int((unsigned(idx) / unsigned(p.N)) % unsigned(p.C)) : int((idx / p.N) % p.C);
    const float rstd = 1.0f / sqrtf(p.var[c] + p.eps), xhat = (p.Z[idx] - p.mean[c]) * rstd;
    const float g = (p.relu && !(p.Y[idx] > 0.f)) ? 0.f : p.dY[idx];
    const float invM = 1.0f / float(long(p.B) * p.N);
    p.dZ[idx] = p.gamma[c] * rstd * (g - p.dbeta[c] * invM - xhat * p.dgamma[c] * invM);
}

static __global__ __launch_bounds__(256) void train_bn_relu_bwd_apply4_kernel(const BnReluBwdParams p) {      // (plane per blockIdx.y, four positions per thread: see train_bn_relu_fwd4_kernel)
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= unsigned(p.N >> 2)) return;
    const long plane = blockIdx.y;
    const int c = int(plane % p.C);
    const float rstd = 1.0f / sqrtf(p.var[c] + p.eps), mean = p.mean[c], gamma = p.gamma[c], db = p.dbeta[c], dg = p.dgamma[c];
    const float invM = 1.0f / float(long(p.B) * p.N);
    const long off = plane * long(p.N) + 4L * q;
    const float4 z = *reinterpret_cast<const float4*>(p.Z + off), y = *reinterpret_cast<const float4*>(p.Y + off), dy = *reinterpret_cast<const float4*>(p.dY + off);
    const float zi[4] = {z.x, z.y, z.z, z.w}, yi[4] = {y.x, y.y, y.z, y.w}, di[4] = {dy.x, dy.y, dy.z, dy.w};
    float o[4];
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) {
        const float xhat = (zi[i] - mean) * rstd;
        const float g = (p.relu && !(yi[i] > 0.f)) ? 0.f : di[i];
        o[i] = gamma * rstd * (g - db * invM - xhat * dg * invM);
    }
    *reinterpret_cast<float4*>(p.dZ + off) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- depthwise 3x3, stride 1, pad 1 on [B, C, H, W] (GhostModule's cheap operation, the GhostBottleneck shortcut: ghost_conv.py:19-23,47-56)
//   forward            y = w (*) x                      (flip = 0)
//   input gradient     dx = flipped(w) (*) dz           (flip = 1: the same kernel on the gradient with the taps mirrored)
//   weight gradient    dw[c][t] = sum_{b,y,x} dz[b,c,y,x] * x[b,c,y+ty-1,x+tx-1]    one workgroup per channel
struct DwTrainParams { const float* X; const float* W; float* Y; int B, C, H, Wd, flip; };
static __global__ __launch_bounds__(256) void train_dw3x3_kernel(const DwTrainParams p) {
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    const long HW = long(p.H) * p.Wd;
    if (idx >= long(p.B) * p.C * HW) return;
    const int x = int(idx % p.Wd), y = int((idx / p.Wd) % p.H), c = int((idx / HW) % p.C);
    const float* plane = p.X + (idx / HW) * HW;
    const float* w = p.W + c * 9;
    float acc = 0.f;
    ACH_UNROLL
    for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd) acc += w[p.flip ? 8 - t : t] * plane[long(iy) * p.Wd + ix];
    }
    p.Y[idx] = acc;
}
struct DwWgradParams { const float* X; const float* dZ; float* dW; int B, C, H, Wd; };
static __global__ __launch_bounds__(256) void train_dw3x3_wgrad_kernel(const DwWgradParams p) {
    __shared__ float red[9][256];
    const int c = blockIdx.x;
    const long HW = long(p.H) * p.Wd, total = long(p.B) * HW;
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long i = threadIdx.x; i < total; i += 256) {
        const long b = i / HW, pix = i % HW;
        const int y = int(pix / p.Wd), x = int(pix % p.Wd);
        const float g = p.dZ[(b * p.C + c) * HW + pix];
        const float* plane = p.X + (b * p.C + c) * HW;
        ACH_UNROLL
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.Wd) acc[t] += g * plane[long(iy) * p.Wd + ix];
        }
    }
    ACH_UNROLL
    for (int t = 0; t < 9; ++t) red[t][threadIdx.x] = acc[t];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (int(threadIdx.x) < st) { ACH_UNROLL for (int t = 0; t < 9; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x < 9) p.dW[c * 9 + threadIdx.x] = red[threadIdx.x][0];
}

// ---- PointNet glue in training mode (pointnet_utils.py:32, 127; pointnet_sem_seg.py:34-37)
// max over the N points of [B, C, N] with the arg-max kept for the backward (ties: lowest index, as torch.max)
struct MaxPtsParams { const float* X; float* Y; int* idx; long rows; int N; };          // rows = B * C
static __global__ __launch_bounds__(256) void train_max_points_fwd_kernel(const MaxPtsParams p) {
    const int lane = threadIdx.x & 63;
    const long row = long(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* x = p.X + row * p.N;
    float best = -3.0e38f; int bi = 0x7fffffff;
    for (int n = lane; n < p.N; n += 64) { const float v = x[n]; if (v > best) { best = v; bi = n; } }
    ACH_UNROLL
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { p.Y[row] = best; p.idx[row] = bi; }
}
struct MaxPtsBwdParams { const float* dY; const int* idx; float* dX; long rows; int N; };
static __global__ __launch_bounds__(256) void train_max_points_bwd_kernel(const MaxPtsBwdParams p) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= p.rows * p.N) return;
    const long row = i / p.N;
    p.dX[i] = (int(i - row * p.N) == p.idx[row]) ? p.dY[row] : 0.f;
}
// log_softmax over the classes of z [B, K, N] written as y [B, N, K] (the transpose + view of pointnet_sem_seg.py:34-36), and its
// backward  dz[b,k,n] = dy[b,n,k] - exp(y[b,n,k]) * sum_k' dy[b,n,k']
struct LsmTrainParams { const float* Z; float* Y; const float* dY; float* dZ; int B, K, N; };
static __global__ __launch_bounds__(256) void train_log_softmax_fwd_kernel(const LsmTrainParams p) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= long(p.B) * p.N) return;
    const long b = i / p.N; const int n = int(i - b * p.N);
    const float* z = p.Z + b * p.K * p.N + n;
    float m = z[0];
    for (int k = 1; k < p.K; ++k) m = fmaxf(m, z[long(k) * p.N]);
    float s = 0.f;
    for (int k = 0; k < p.K; ++k) s += expf(z[long(k) * p.N] - m);
    const float lse = m + logf(s);
    for (int k = 0; k < p.K; ++k) p.Y[i * p.K + k] = z[long(k) * p.N] - lse;
}
static __global__ __launch_bounds__(256) void train_log_softmax_bwd_kernel(const LsmTrainParams p) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= long(p.B) * p.N) return;
    const long b = i / p.N; const int n = int(i - b * p.N);
    float s = 0.f;
    for (int k = 0; k < p.K; ++k) s += p.dY[i * p.K + k];
    for (int k = 0; k < p.K; ++k) p.dZ[(b * p.K + k) * p.N + n] = p.dY[i * p.K + k] - expf(p.Y[i * p.K + k]) * s;
}

}  // namespace ach
