// k_prepost.h — the steps immediately either side of the forward, which the reference does on the host in
// NumPy / sklearn per frame: radar-map min-max scaling (utils/utils.py:51-54), point-cloud column L2 normalisation + [N,D]->[D,N]
// (achelous.py:240-243), image /255 - mean / std + HWC->CHW (utils/utils.py:44-48, achelous.py:205), and the per-pixel class of a
// segmentation output (achelous.py:283-318: argmax of the softmax == argmax of the logits at network resolution).
#pragma once
#include "ach_platform.h"

namespace ach {

// ---- radar map: (x - min) / (max - min) + 1e-13 with min / max over the WHOLE frame (all channels)
struct MinMaxParams { const float* X; float* partial; long per_frame; int S; };     // partial [B][S][2]
static __global__ __launch_bounds__(256) void frame_minmax_kernel(const MinMaxParams p) {
    __shared__ float smin[256], smax[256];
    const long b = blockIdx.x;
    const int s = blockIdx.y;
    const float* x = p.X + b * p.per_frame;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (long i = long(s) * 256 + threadIdx.x; i < p.per_frame; i += long(p.S) * 256) { const float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (int(threadIdx.x) < st) { smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + st]); smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { p.partial[(b * p.S + s) * 2] = smin[0]; p.partial[(b * p.S + s) * 2 + 1] = smax[0]; }
}
struct RadarScaleParams { const float* X; const float* partial; void* Y; long per_frame; int S; int B; };
template <class T>
__global__ __launch_bounds__(256) void radar_scale_kernel(const RadarScaleParams p) {
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.per_frame * p.B) return;
    const long b = idx / p.per_frame;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int s = 0; s < p.S; ++s) { mn = fminf(mn, p.partial[(b * p.S + s) * 2]); mx = fmaxf(mx, p.partial[(b * p.S + s) * 2 + 1]); }
    Store<T>::st(static_cast<T*>(p.Y) + idx, (p.X[idx] - mn) / (mx - mn) + 1e-13f);
}

// ---- points: sklearn.preprocessing.normalize(X[N,D], axis=0) (zero columns are left unchanged) and [N,D] -> [D,N]
struct PointNormParams { const float* X; void* Y; int B, N, D; };
template <class T>
__global__ __launch_bounds__(256) void point_norm_kernel(const PointNormParams p) {
    __shared__ float red[256];
    const long b = blockIdx.x / p.D;
    const int d = blockIdx.x % p.D;
    const float* x = p.X + b * p.N * long(p.D) + d;
    float s = 0.f;
    for (int n = threadIdx.x; n < p.N; n += 256) { const float v = x[long(n) * p.D]; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if (int(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    float nrm = sqrtf(red[0]);
    if (nrm == 0.f) nrm = 1.f;
    T* y = static_cast<T*>(p.Y) + (b * p.D + d) * long(p.N);
    for (int n = threadIdx.x; n < p.N; n += 256) Store<T>::st(y + n, x[long(n) * p.D] / nrm);
}

// ---- image: uint8 HWC (already letterboxed by the host) -> ((v / 255) - mean) / std, CHW
struct ImagePrepParams { const unsigned char* X; void* Y; int B, H, Wd; };
template <class T>
__global__ __launch_bounds__(256) void image_prep_kernel(const ImagePrepParams p) {
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    const long HW = long(p.H) * p.Wd;
    if (idx >= HW * p.B) return;
    const long b = idx / HW, pix = idx - b * HW;
    const unsigned char* x = p.X + idx * 3;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    ACH_UNROLL
    for (int c = 0; c < 3; ++c) Store<T>::st(static_cast<T*>(p.Y) + (b * 3 + c) * HW + pix, (float(x[c]) / 255.0f - mean[c]) / stdv[c]);
}

// ---- segmentation: class index per pixel (first maximum, as numpy / torch argmax)
struct SegArgmaxParams { const void* X; unsigned char* Y; int B, C; long HW; };
template <class T>
__global__ __launch_bounds__(256) void seg_argmax_kernel(const SegArgmaxParams p) {
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.HW * p.B) return;
    const long b = idx / p.HW, pix = idx - b * p.HW;
    const T* x = static_cast<const T*>(p.X) + b * p.C * p.HW + pix;
    float best = Store<T>::ld(x);
    int bi = 0;
    for (int c = 1; c < p.C; ++c) { const float v = Store<T>::ld(x + c * p.HW); if (v > best) { best = v; bi = c; } }
    p.Y[idx] = (unsigned char)bi;
}

}  // namespace ach
