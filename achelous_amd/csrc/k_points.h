// k_points.h — glue kernels of the PointNet branch (nets/pointcloudseg/pointnet2/pointnet_utils.py:10-133,
// pointnet_sem_seg.py:26-37).  The shared MLPs (conv1d k=1) and the FC stacks run on the MFMA GEMM with points as
// rows ([B, N, C] "NWC"); the max over points is fused into that GEMM's epilogue (wavefront shuffles + one atomic
// per column per wave).  What remains here: layout change of the input cloud, the 3x3 input transform, packing of
// the per-sample 32x32 feature transform into MFMA fragment order, the global-feature broadcast and log-softmax.
#pragma once
#include "ach_platform.h"
#include "k_gemm.h"

namespace ach {

// [B, D, N] (reference layout, utils/dataloader.py:546-547) -> rows [B*N, ld] with channels D..ld-1 zero
struct PcPrepParams { const void* X; void* Y; int B, D, N; long ld; };
template <class T, class IO = T>
__global__ void pc_prep_kernel(const PcPrepParams p) { f16_sat_mode<T>();
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * p.N * p.ld) return;
    const int c = int(idx % p.ld);
    const long row = idx / p.ld;
    const long b = row / p.N, n = row - b * p.N;
    const float v = c < p.D ? Store<IO>::ld(static_cast<const IO*>(p.X) + (b * p.D + c) * p.N + n) : 0.f;
    Store<T>::st(static_cast<T*>(p.Y) + idx, v);
}

// x' = [xyz @ (fc3 + I3), extra features]   (pointnet_utils.py:39-44,106-112)
struct PcT3Params { const void* X; long ldx; const void* t9; long ldt; void* Y; long ldy; int B, N, D; };
template <class T>
__global__ void pc_apply_t3_kernel(const PcT3Params p) { f16_sat_mode<T>();
    const long row = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= long(p.B) * p.N) return;
    const long b = row / p.N;
    const T* x = static_cast<const T*>(p.X) + row * p.ldx;
    const T* t = static_cast<const T*>(p.t9) + b * p.ldt;
    T* y = static_cast<T*>(p.Y) + row * p.ldy;
    const float x0 = Store<T>::ld(x), x1 = Store<T>::ld(x + 1), x2 = Store<T>::ld(x + 2);
    for (int j = 0; j < 3; ++j) {
        const float t0 = Store<T>::ld(t + j) + (j == 0 ? 1.f : 0.f);
        const float t1 = Store<T>::ld(t + 3 + j) + (j == 1 ? 1.f : 0.f);
        const float t2 = Store<T>::ld(t + 6 + j) + (j == 2 ? 1.f : 0.f);
        Store<T>::st(y + j, x0 * t0 + x1 * t1 + x2 * t2);
    }
    for (int j = 3; j < p.D; ++j) Store<T>::st(y + j, Store<T>::ld(x + j));
}

// per-sample feature transform (fc3 + I_k) -> packed MFMA weight fragments: out[n=j][k=i] = Tf[i][j]
// (bmm(x[N,k], Tf[k,k]), pointnet_utils.py:79-84,116-120)
struct PcPackParams { const void* t; long ldt; void* Wp; long group_stride; int B, k, NT, ksteps; };
template <class T>
__global__ void pc_pack_transform_kernel(const PcPackParams p) { f16_sat_mode<T>();
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * p.k * p.k) return;
    const int j = int(idx % p.k);
    const int i = int((idx / p.k) % p.k);
    const long b = idx / (long(p.k) * p.k);
    const float v = Store<T>::ld(static_cast<const T*>(p.t) + b * p.ldt + i * p.k + j) + (i == j ? 1.f : 0.f);
    Store<T>::st(static_cast<T*>(p.Wp) + b * p.group_stride + wfrag_offset(j, i, p.NT, p.ksteps, Store<T>::VEC), v);
}

// rows [B*N, ldy]: channels [0,G) = global feature of the sample, [G, G+F) = per-point feature
struct PcConcatParams { const void* g; long ldg; const void* f; long ldf; void* Y; long ldy; int B, N, G, F; };
template <class T>
__global__ void pc_concat_kernel(const PcConcatParams p) { f16_sat_mode<T>();
    const int C4 = (p.G + p.F) >> 2;                       // one thread = 4 channels (G and F are multiples of 4: 128 + 32)
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * p.N * C4) return;
    const int c = int(idx % C4) * 4;
    const long row = idx / C4;
    const long b = row / p.N;
    float v[4];
    if (c < p.G) Store<T>::ld4(static_cast<const T*>(p.g) + b * p.ldg + c, v);
    else Store<T>::ld4(static_cast<const T*>(p.f) + row * p.ldf + (c - p.G), v);
    Store<T>::st4(static_cast<T*>(p.Y) + row * p.ldy + c, v);
}

// log_softmax over the class axis; output dense [rows, K]
struct LsmParams { const void* X; long ldx; void* Y; long rows; int K; };
template <class T, class IO = T>
__global__ void log_softmax_kernel(const LsmParams p) { f16_sat_mode<T>();
    const long row = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= p.rows) return;
    const T* x = static_cast<const T*>(p.X) + row * p.ldx;
    float mx = -3.0e38f;
    for (int k = 0; k < p.K; ++k) mx = fmaxf(mx, Store<T>::ld(x + k));
    float s = 0.f;
    for (int k = 0; k < p.K; ++k) s += expf(Store<T>::ld(x + k) - mx);
    const float lse = mx + logf(s);
    for (int k = 0; k < p.K; ++k) Store<IO>::st(static_cast<IO*>(p.Y) + row * p.K + k, Store<T>::ld(x + k) - lse);
}

}  // namespace ach

