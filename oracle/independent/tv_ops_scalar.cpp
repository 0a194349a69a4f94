// tv_ops_scalar.cpp — SECOND, independently written CPU statement of the two third-party operators on the Achelous hot path.
//
// TEST INFRASTRUCTURE (only tests/ may load the library built from this file).  PARITY UNPINNED: torchvision==0.12.0
// (requirements.txt:17 of the reference) is un-vendored and absent from this image, so neither this file nor
// oracle/deform_conv.py / oracle/nms.py can be checked against the real binary here.
//
// Purpose: oracle/deform_conv.py (vectorised torch gathers) and oracle/nms.py (numpy) were written first and are what the golden
// fixtures were generated with.  This file states the same published algorithms a second time in the most literal form — scalar
// loops over one output element at a time, in the loop order of torchvision's CPU kernels
//   ops/cpu/deform_conv2d_kernel.cpp : bilinear_interpolate, deformable_im2col_kernel, then weight x columns
//   ops/cpu/nms_kernel.cpp           : nms_kernel_impl
//   ops/boxes.py                     : _batched_nms_coordinate_trick
// — sharing no code, helper or data layout trick with the first statement.  tests/test_independent_ops.py fuzzes one against the
// other (far offsets, offsets landing exactly on -1 / H, zero-area and duplicated boxes, ties), so a misreading of the published
// semantics has to be made twice, in two styles, to survive.
//
// Reference call sites these operators serve: backbone/conv_utils/dcn.py:56 (deform_conv2d, 3x3, stride 1, pad 1, one offset
// group, modulated) and utils/utils_bbox.py:125-130 (batched_nms).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace {

// value of a single-channel H x W plane at the real-valued point (h, w); zero outside, corners outside the plane contribute 0
float sample_plane(const float* plane, int H, int W, float h, float w) {
    if (h <= -1.0f || h >= float(H) || w <= -1.0f || w >= float(W)) return 0.0f;
    const int h0 = int(std::floor(h));
    const int w0 = int(std::floor(w));
    const int h1 = h0 + 1;
    const int w1 = w0 + 1;
    const float fh = h - float(h0);
    const float fw = w - float(w0);
    const float gh = 1.0f - fh;
    const float gw = 1.0f - fw;
    float tl = 0.0f, tr = 0.0f, bl = 0.0f, br = 0.0f;
    if (h0 >= 0 && w0 >= 0) tl = plane[h0 * W + w0];
    if (h0 >= 0 && w1 <= W - 1) tr = plane[h0 * W + w1];
    if (h1 <= H - 1 && w0 >= 0) bl = plane[h1 * W + w0];
    if (h1 <= H - 1 && w1 <= W - 1) br = plane[h1 * W + w1];
    const float a = gh * gw, b = gh * fw, c = fh * gw, d = fh * fw;
    // one rounding per operation, in the order the published kernel writes the sum
    volatile float s = a * tl;
    volatile float t = b * tr;
    s = s + t;
    t = c * bl;
    s = s + t;
    t = d * br;
    s = s + t;
    return s;
}

}  // namespace

extern "C" {

// Modulated deformable convolution, one offset group, groups = 1.
//   input  [B, Ci, H, W]    offset [B, 2*kh*kw, Ho, Wo] (channel 2k = dy, 2k+1 = dx of tap k = i*kw + j)
//   mask   [B, kh*kw, Ho, Wo] or nullptr    weight [Co, Ci, kh, kw]    bias [Co] or nullptr    out [B, Co, Ho, Wo]
// The contraction over (ci, i, j) is accumulated in double and rounded once: the checker's job is the sampling semantics, and a
// tolerance-level comparison (1e-5) of the contraction is what the test asks for.
int tv_deform_conv2d(const float* input, const float* offset, const float* mask, const float* weight, const float* bias, float* out,
                     int B, int Ci, int H, int W, int Co, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                     int dil_h, int dil_w) {
    const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
    const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
    if (Ho <= 0 || Wo <= 0) return -1;
    const long plane_out = long(Ho) * Wo;
    std::vector<float> col(size_t(Ci) * kh * kw);
    for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                // the column of this output position: one sampled, modulated value per (input channel, tap)
                for (int ci = 0; ci < Ci; ++ci) {
                    const float* plane = input + (long(b) * Ci + ci) * H * W;
                    for (int i = 0; i < kh; ++i)
                        for (int j = 0; j < kw; ++j) {
                            const int k = i * kw + j;
                            const float dy = offset[((long(b) * 2 * kh * kw) + 2 * k) * plane_out + long(oy) * Wo + ox];
                            const float dx = offset[((long(b) * 2 * kh * kw) + 2 * k + 1) * plane_out + long(oy) * Wo + ox];
                            const float m = mask ? mask[((long(b) * kh * kw) + k) * plane_out + long(oy) * Wo + ox] : 1.0f;
                            const float y = float(oy * stride_h - pad_h) + float(i * dil_h) + dy;
                            const float x = float(ox * stride_w - pad_w) + float(j * dil_w) + dx;
                            col[(size_t(ci) * kh + i) * kw + j] = m * sample_plane(plane, H, W, y, x);
                        }
                }
                for (int co = 0; co < Co; ++co) {
                    double acc = bias ? double(bias[co]) : 0.0;
                    const float* wrow = weight + size_t(co) * Ci * kh * kw;
                    for (size_t q = 0; q < col.size(); ++q) acc += double(wrow[q]) * double(col[q]);
                    out[(long(b) * Co + co) * plane_out + long(oy) * Wo + ox] = float(acc);
                }
            }
    return 0;
}

// Greedy NMS over n boxes (x1, y1, x2, y2).  `order` = candidate indices by descending score, ties by ascending index (the
// spec decision both statements share; torchvision's own tie order is unspecified).  Writes kept indices, returns their count.
long tv_nms(const float* boxes, const float* scores, long n, float iou_threshold, int64_t* keep) {
    std::vector<long> order(static_cast<size_t>(n));
    for (long i = 0; i < n; ++i) order[size_t(i)] = i;
    // insertion sort: stable by construction, no library comparator whose tie behaviour would have to be trusted
    for (long a = 1; a < n; ++a) {
        const long cur = order[size_t(a)];
        long pos = a;
        while (pos > 0 && scores[order[size_t(pos - 1)]] < scores[cur]) { order[size_t(pos)] = order[size_t(pos - 1)]; --pos; }
        order[size_t(pos)] = cur;
    }
    std::vector<unsigned char> dead(static_cast<size_t>(n), 0);
    std::vector<float> area(static_cast<size_t>(n));
    for (long i = 0; i < n; ++i) {
        volatile float w = boxes[4 * i + 2] - boxes[4 * i + 0];
        volatile float h = boxes[4 * i + 3] - boxes[4 * i + 1];
        volatile float a = w * h;
        area[size_t(i)] = a;
    }
    long kept = 0;
    for (long p = 0; p < n; ++p) {
        const long i = order[size_t(p)];
        if (dead[size_t(i)]) continue;
        keep[kept++] = i;
        for (long q = p + 1; q < n; ++q) {
            const long j = order[size_t(q)];
            if (dead[size_t(j)]) continue;
            const float lx = boxes[4 * i + 0] > boxes[4 * j + 0] ? boxes[4 * i + 0] : boxes[4 * j + 0];
            const float ly = boxes[4 * i + 1] > boxes[4 * j + 1] ? boxes[4 * i + 1] : boxes[4 * j + 1];
            const float rx = boxes[4 * i + 2] < boxes[4 * j + 2] ? boxes[4 * i + 2] : boxes[4 * j + 2];
            const float ry = boxes[4 * i + 3] < boxes[4 * j + 3] ? boxes[4 * i + 3] : boxes[4 * j + 3];
            volatile float ww = rx - lx;
            volatile float hh = ry - ly;
            if (ww < 0.0f) ww = 0.0f;
            if (hh < 0.0f) hh = 0.0f;
            volatile float inter = ww * hh;
            volatile float uni = area[size_t(i)] + area[size_t(j)];
            uni = uni - inter;
            volatile float iou = inter / uni;              // 0 / 0 = NaN compares false: a zero-area pair never suppresses
            if (iou > iou_threshold) dead[size_t(j)] = 1;
        }
    }
    return kept;
}

// batched_nms, coordinate-trick variant: every box is shifted by class_id * (max coordinate + 1) so that boxes of different
// classes never overlap, then one plain NMS.  class ids arrive as floats (utils_bbox.py:122 builds them with .float()).
long tv_batched_nms(const float* boxes, const float* scores, const float* class_ids, long n, float iou_threshold, int64_t* keep) {
    if (n == 0) return 0;
    float top = boxes[0];
    for (long q = 1; q < 4 * n; ++q) if (boxes[q] > top) top = boxes[q];
    volatile float span = top + 1.0f;
    std::vector<float> moved(size_t(4) * size_t(n));
    for (long i = 0; i < n; ++i) {
        volatile float off = class_ids[i] * span;
        for (int c = 0; c < 4; ++c) { volatile float v = boxes[4 * i + c] + off; moved[size_t(4 * i + c)] = v; }
    }
    return tv_nms(moved.data(), scores, n, iou_threshold, keep);
}

}  // extern "C"
