// k_train3.h — training mode of the PointNet++ branch (`pc_seg='pn2'`; our own specification, DESIGN.md section 5b: the reference snapshot has no PointNet++ code,
// nets/Achelous.py:31-32).  The FORWARD geometry is the inference engine's own kernels at T = float (k_pn2.h: farthest-point sampling, ball query + grouping, 3-NN
// interpolation + skip concatenation — the index selection is integer work, bit-exact against oracle/pointnet2_oracle.py); the shared MLPs run on the training
// primitives of k_train.h with the points as the BatchNorm axis.  What is here are the two ADJOINTS, both scatter-adds of rows:
//   grouping        dfeats[b, group_idx[b,s,k], c] += dgrouped[(b,s,k), 3 + c]          (the xyz - centroid columns belong to the input cloud: no gradient)
//   interpolation   dskip[pt, c] = dout[pt, c];   dsparse[b, nn_j(pt), c] += w_j(pt) * dout[pt, C1 + c],  j < 3 — neighbours and weights re-derived from the coordinates
//                   exactly as the forward kernel derives them (same code, same tie rules)
// with fp32 atomics at the L2 (train_atomic_add, k_train2.h): the summation order over a point's contributions is unspecified, as in every scatter-form backward.
#pragma once
#include "ach_platform.h"
#include "k_pn2.h"
#include "k_train2.h"

namespace ach {

struct Pn2GroupBwdParams { const int* group_idx; const float* dgrouped; long ldg; float* dfeats; int C; int B, n, S, nsample; };
static __global__ __launch_bounds__(256) void train_pn2_group_bwd_kernel(const Pn2GroupBwdParams p) {     // one thread per (b, s, k, c)
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    const long rows = long(p.B) * p.S * p.nsample;
    if (i >= rows * p.C) return;
    const long row = tdiv(i, p.C);
    const int c = int(i - row * p.C);
    const long b = tdiv(row, long(p.S) * p.nsample);
    const int src = p.group_idx[row];
    train_atomic_add(p.dfeats + (b * p.n + src) * long(p.C) + c, p.dgrouped[row * p.ldg + 3 + c]);
}

// one wave per dense point, as pn2_interp_kernel: the three neighbours and their normalised inverse-distance weights, then the row's gradient
static __global__ __launch_bounds__(256) void train_pn2_interp_bwd_kernel(const InterpParams p, const float* __restrict__ dout, float* dp1, float* dp2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pt = long(blockIdx.x) * 4 + wave;
    if (pt >= long(p.B) * p.n) return;
    const long b = pt / p.n;
    const float ax = p.xyz1[pt * 3], ay = p.xyz1[pt * 3 + 1], az = p.xyz1[pt * 3 + 2];
    const float* xyz2 = p.xyz2 + b * p.s * 3;
    const float INF = 3.0e38f;
    float d[PN2_INTERP_SPL];
    ACH_UNROLL
    for (int k = 0; k < PN2_INTERP_SPL; ++k) {
        const int i = lane + 64 * k;
        d[k] = i < p.s ? pn2_sqdist(ax, ay, az, xyz2[i * 3], xyz2[i * 3 + 1], xyz2[i * 3 + 2]) : INF;
    }
    int nn[3]; float w[3];
    ACH_UNROLL
    for (int r = 0; r < 3; ++r) {
        float best = INF; int besti = 0x7fffffff;
        ACH_UNROLL
        for (int k = 0; k < PN2_INTERP_SPL; ++k)
            if (d[k] < best) { best = d[k]; besti = lane + 64 * k; }
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m); const int oi = __shfl_xor(besti, m);
            if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        nn[r] = besti;
        w[r] = 1.0f / (best + 1e-8f);
        ACH_UNROLL
        for (int k = 0; k < PN2_INTERP_SPL; ++k)
            if (lane + 64 * k == besti) d[k] = INF;
    }
    const float norm = (w[0] + w[1]) + w[2];
    w[0] /= norm; w[1] /= norm; w[2] /= norm;
    const float* g = dout + pt * p.ldo;
    for (int col = lane; col < p.C1 + p.C2; col += 64) {
        const float v = g[col];
        if (col < p.C1) dp1[pt * p.C1 + col] = v;
        else {
            const int c = col - p.C1;
            float* q = dp2 + b * p.s * long(p.C2) + c;
            train_atomic_add(q + long(nn[0]) * p.C2, w[0] * v);
            train_atomic_add(q + long(nn[1]) * p.C2, w[1] * v);
            train_atomic_add(q + long(nn[2]) * p.C2, w[2] * v);
        }
    }
}

}  // namespace ach
