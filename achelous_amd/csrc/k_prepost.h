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
__global__ __launch_bounds__(256) void radar_scale_kernel(const RadarScaleParams p) { f16_sat_mode<T>();
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
__global__ __launch_bounds__(256) void point_norm_kernel(const PointNormParams p) { f16_sat_mode<T>();
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
__global__ __launch_bounds__(256) void image_prep_kernel(const ImagePrepParams p) { f16_sat_mode<T>();
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
__global__ __launch_bounds__(256) void seg_argmax_kernel(const SegArgmaxParams p) { f16_sat_mode<T>();
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.HW * p.B) return;
    const long b = idx / p.HW, pix = idx - b * p.HW;
    const T* x = static_cast<const T*>(p.X) + b * p.C * p.HW + pix;
    float best = Store<T>::ld(x);
    int bi = 0;
    for (int c = 1; c < p.C; ++c) { const float v = Store<T>::ld(x + c * p.HW); if (v > best) { best = v; bi = c; } }
    p.Y[idx] = (unsigned char)bi;
}

// ---- segmentation output -> class map at the ORIGINAL image size, as the reference's detect_image does it (achelous.py:283-318):
// softmax over the classes at network resolution, crop the letterbox's grey bars, cv2.resize(..., INTER_LINEAR) to the original size,
// argmax.  (Interpolating probabilities and THEN taking the argmax is not the argmax of the network-resolution map.)
// Step 1: probabilities [B, C, R, R] fp32 into a workspace; step 2: one thread per output pixel blends the four source pixels of
// every class and keeps the first maximum.  INTER_LINEAR for float images, restated from OpenCV's resize (imgproc/src/resize.cpp:
// half-pixel centres, fx = (dx + 0.5) * (src / dst) - 0.5, sx = floor(fx); sx < 0 -> (0, weight 0); sx >= src - 1 -> (src - 1,
// weight 0); rows blended horizontally first, then vertically, in fp32).  OpenCV is not installable in this image: parity unpinned.
struct SegSoftmaxParams { const void* X; float* P; int B, C; long HW; };
template <class T>
__global__ __launch_bounds__(256) void seg_softmax_kernel(const SegSoftmaxParams p) { f16_sat_mode<T>();
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.HW * p.B) return;
    const long b = idx / p.HW, pix = idx - b * p.HW;
    const T* x = static_cast<const T*>(p.X) + b * p.C * p.HW + pix;
    float* o = p.P + b * p.C * p.HW + pix;
    float mx = Store<T>::ld(x);
    for (int c = 1; c < p.C; ++c) mx = fmaxf(mx, Store<T>::ld(x + c * p.HW));
    float sum = 0.f;
    for (int c = 0; c < p.C; ++c) { const float e = expf(Store<T>::ld(x + c * p.HW) - mx); o[c * p.HW] = e; sum += e; }
    for (int c = 0; c < p.C; ++c) o[c * p.HW] = o[c * p.HW] / sum;
}
struct SegResizeParams {
    const float* P; unsigned char* Y;
    int B, C, R;                  // probabilities [B, C, R, R]
    int y0, x0, nh, nw;           // the cropped (un-letterboxed) window of the network-resolution map
    int oh, ow;                   // output size
    double sy, sx;                // nh / oh, nw / ow
};
static __global__ __launch_bounds__(256) void seg_resize_argmax_kernel(const SegResizeParams p) {
    const long per = long(p.oh) * p.ow;
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= per * p.B) return;
    const long b = idx / per;
    const int dy = int((idx - b * per) / p.ow), dx = int((idx - b * per) % p.ow);
    float fy = float((double(dy) + 0.5) * p.sy - 0.5), fx = float((double(dx) + 0.5) * p.sx - 0.5);
    int iy = int(floorf(fy)), ix = int(floorf(fx));
    fy -= float(iy); fx -= float(ix);
    if (iy < 0) { iy = 0; fy = 0.f; }
    if (iy >= p.nh - 1) { iy = p.nh - 1; fy = 0.f; }
    if (ix < 0) { ix = 0; fx = 0.f; }
    if (ix >= p.nw - 1) { ix = p.nw - 1; fx = 0.f; }
    const int iy1 = iy + 1 < p.nh ? iy + 1 : iy, ix1 = ix + 1 < p.nw ? ix + 1 : ix;
    const long HW = long(p.R) * p.R;
    const float* base = p.P + b * p.C * HW;
    const long r0 = long(p.y0 + iy) * p.R + p.x0, r1 = long(p.y0 + iy1) * p.R + p.x0;
    const float ax0 = 1.f - fx, ay0 = 1.f - fy;
    float best = -1.f;
    int bi = 0;
    for (int c = 0; c < p.C; ++c) {
        const float* q = base + c * HW;
        const float top = __fadd_rn(__fmul_rn(q[r0 + ix], ax0), __fmul_rn(q[r0 + ix1], fx));
        const float bot = __fadd_rn(__fmul_rn(q[r1 + ix], ax0), __fmul_rn(q[r1 + ix1], fx));
        const float v = __fadd_rn(__fmul_rn(top, ay0), __fmul_rn(bot, fy));
        if (v > best) { best = v; bi = c; }
    }
    p.Y[idx] = (unsigned char)bi;
}

// ---- kept boxes -> image pixels: yolo_correct_boxes (utils/utils_bbox.py:5-30, called from non_max_suppression :177-180).
// rows [B, max_det, 7] = x1, y1, x2, y2 (normalised corners in the letterboxed network input), obj, cls_conf, cls_id  ->
// out  [B, max_det, 7] = y1, x1, y2, x2 in pixels of the ORIGINAL image, other columns copied.  The reference does this in numpy
// with float64 intermediates (a float32 array combined with float64 shape arrays) and stores float32: same here, in double.
struct BoxCorrectParams { const float* rows; const int* count; float* out; int B, max_det; double in_h, in_w, img_h, img_w; int letterbox; };
static __global__ __launch_bounds__(256) void correct_boxes_kernel(const BoxCorrectParams p) {
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * p.max_det) return;
    const int b = int(idx / p.max_det), k = int(idx % p.max_det);
    const float* r = p.rows + idx * 7;
    float* o = p.out + idx * 7;
    if (k >= p.count[b]) { for (int c = 0; c < 7; ++c) o[c] = 0.f; return; }
    // box_xy = (x1y1 + x2y2) / 2, box_wh = x2y2 - x1y1 in float32 (utils_bbox.py:178), then reversed to (y, x)
    const float cx = __fdiv_rn(__fadd_rn(r[0], r[2]), 2.f), cy = __fdiv_rn(__fadd_rn(r[1], r[3]), 2.f);
    float bw = __fsub_rn(r[2], r[0]), bh = __fsub_rn(r[3], r[1]);
    double yy = double(cy), xx = double(cx);
    if (p.letterbox) {
        const double m = fmin(p.in_h / p.img_h, p.in_w / p.img_w);
        const double nh = rint(p.img_h * m), nw = rint(p.img_w * m);                  // np.round: half to even, as rint
        const double off_y = (p.in_h - nh) / 2. / p.in_h, off_x = (p.in_w - nw) / 2. / p.in_w;
        const double sc_y = p.in_h / nh, sc_x = p.in_w / nw;
        yy = (yy - off_y) * sc_y; xx = (xx - off_x) * sc_x;
        bh = float(double(bh) * sc_y); bw = float(double(bw) * sc_x);                 // `box_hw *= scale` is an in-place float32 update
    }
    const float hy = __fdiv_rn(bh, 2.f), hx = __fdiv_rn(bw, 2.f);                    // float32 array / python float stays float32
    double y_lo, x_lo, y_hi, x_hi;
    if (p.letterbox) { y_lo = yy - double(hy); x_lo = xx - double(hx); y_hi = yy + double(hy); x_hi = xx + double(hx); }   // float64 - float32
    else { y_lo = double(__fsub_rn(cy, hy)); x_lo = double(__fsub_rn(cx, hx)); y_hi = double(__fadd_rn(cy, hy)); x_hi = double(__fadd_rn(cx, hx)); }   // all float32
    o[0] = float(y_lo * p.img_h); o[1] = float(x_lo * p.img_w);                        // `boxes *= image_shape`: computed in float64, stored float32
    o[2] = float(y_hi * p.img_h); o[3] = float(x_hi * p.img_w);
    o[4] = r[4]; o[5] = r[5]; o[6] = r[6];
}

// One pass of Pillow's ImagingResample for 8-bit pixels (the reference letterboxes with PIL: utils/utils.py:20-33, Image.BICUBIC): every output
// sample is an integer dot product of up to `ksize` source samples along ONE axis with 22-bit fixed-point coefficients (computed on the
// host in double precision exactly as Resample.c's precompute_coeffs / normalize_coeffs_8bpc do), rounded and clipped to 0..255.  A resize is a
// horizontal pass into an 8-bit intermediate followed by a vertical pass, as in Pillow — the intermediate's rounding is part of the result.
// Images are HWC (PIL's layout); `dst_pitch` lets the vertical pass write straight into the letterbox canvas.
struct ResamplePassParams {
    const uint8_t* src; uint8_t* dst; const int* bounds; const int* kk;      // bounds [n_out][2] = (first source index, count); kk [n_out][ksize]
    int ksize, H_in, W_in, H_out, W_out, C, vertical;
    long src_pitch, dst_pitch;                                                // bytes per row
};
static __global__ __launch_bounds__(256) void resample_pass_kernel(const ResamplePassParams p) {
    const long i = long(blockIdx.x) * 256 + threadIdx.x;
    if (i >= long(p.H_out) * p.W_out * p.C) return;
    const int c = int(i % p.C), x = int((i / p.C) % p.W_out), y = int(i / (long(p.C) * p.W_out));
    const int o = p.vertical ? y : x;
    const int first = p.bounds[2 * o], count = p.bounds[2 * o + 1];
    const int* k = p.kk + long(o) * p.ksize;
    int ss = 1 << 21;                                                         // 1 << (PRECISION_BITS - 1), PRECISION_BITS = 32 - 8 - 2
    if (p.vertical) {
        const uint8_t* s = p.src + long(first) * p.src_pitch + long(x) * p.C + c;
        for (int t = 0; t < count; ++t) ss += int(s[long(t) * p.src_pitch]) * k[t];
    } else {
        const uint8_t* s = p.src + long(y) * p.src_pitch + long(first) * p.C + c;
        for (int t = 0; t < count; ++t) ss += int(s[long(t) * p.C]) * k[t];
    }
    const int v = ss >> 22;                                                   // arithmetic shift, then clip8
    p.dst[long(y) * p.dst_pitch + long(x) * p.C + c] = uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v));
}


// ---- fp16 storage: elements of the plan's activation tensors that are saturated (|x| = 65504, what an overflowing conversion produces under MODE.FP16_OVFL,
// ach_platform.h f16_sat_mode), infinite or NaN — i.e. halves whose magnitude bits are >= 0x7bff.  blockIdx.y = tensor, grid-stride over its dwords.
struct SatRegion { const uint32_t* p; unsigned long long dwords; };
static __global__ __launch_bounds__(256) void sat_count_kernel(const SatRegion* __restrict__ regions, unsigned long long* __restrict__ count) {
    const SatRegion r = regions[blockIdx.y];
    unsigned n = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < r.dwords; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t w = r.p[i];
        n += ((w & 0x7fffu) >= 0x7bffu) + (((w >> 16) & 0x7fffu) >= 0x7bffu);
    }
    if (n) atomicAdd(count, (unsigned long long)n);
}

}  // namespace ach
