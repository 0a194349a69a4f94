// k_detect.h — box decode and class-aware NMS on the device (utils/utils_bbox.py:33-132).
//
// decode: the three raw head maps [B,5+C,h,w] -> [B, A, 5+C] fp32, A = sum(h*w): sigmoid on obj/cls, grid + stride
//         decode, exp on w/h, normalise by the input size.
// nms:    one workgroup per image.  xywh->xyxy, class max, confidence filter (obj*cls >= conf), torchvision 0.12.0
//         `batched_nms` coordinate trick (boxes + cls * (max_coord + 1)), stable descending sort by score (bitonic,
//         in LDS, ties -> lower anchor index), greedy suppression IoU > thr.  Every fp32 operation is issued as a
//         single correctly-rounded op in the oracle's order (no FMA contraction): identical decoded inputs give
//         bit-identical kept-index sequences.
#pragma once
#include "ach_platform.h"
#include "k_gemm.h"   // order_encode

namespace ach {

#if defined(ACH_HOSTEMU)
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
#endif

struct DecodeParams {
    const void* det[3]; int h[3], w[3];
    float* out;                 // [B, A, NC5]
    int B, NC5, A; float in_h, in_w;
};
template <class T>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeParams p) { f16_sat_mode<T>();
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= long(p.B) * p.A) return;
    const long b = idx / p.A;
    int a = int(idx - b * p.A);
    int lvl = 0;
    while (lvl < 2 && a >= p.h[lvl] * p.w[lvl]) { a -= p.h[lvl] * p.w[lvl]; ++lvl; }
    const int hw = p.h[lvl] * p.w[lvl];
    const int gy = a / p.w[lvl], gx = a - gy * p.w[lvl];
    const float stride = p.in_h / float(p.h[lvl]);
    const T* src = static_cast<const T*>(p.det[lvl]) + b * p.NC5 * hw + a;
    float* o = p.out + idx * p.NC5;
    const float v0 = Store<T>::ld(src), v1 = Store<T>::ld(src + hw), v2 = Store<T>::ld(src + 2 * hw), v3 = Store<T>::ld(src + 3 * hw);
    o[0] = __fdiv_rn(__fmul_rn(__fadd_rn(v0, float(gx)), stride), p.in_w);
    o[1] = __fdiv_rn(__fmul_rn(__fadd_rn(v1, float(gy)), stride), p.in_h);
    o[2] = __fdiv_rn(__fmul_rn(expf(v2), stride), p.in_w);
    o[3] = __fdiv_rn(__fmul_rn(expf(v3), stride), p.in_h);
    for (int c = 4; c < p.NC5; ++c) o[c] = sigmoidf_(Store<T>::ld(src + long(c) * hw));
}

struct NmsParams {
    const float* dec;           // [B, A, NC5] decoded predictions (cx, cy, w, h, obj, cls...)
    float* scratch;             // [B, A, 8] workspace: candidate (x1,y1,x2,y2 offset boxes, score, cls_conf, obj, cls_id)
    int* scratch_idx;           // [B, A] candidate -> anchor index
    float* scratch_boxes;       // [B, A, 4] sorted candidate boxes, used instead of LDS when more than NMS_LDS_BOXES pass the filter
    float* rows; int* kept; int* count;   // [B, max_det, 7], [B, max_det], [B]
    int B, A, NC5, num_classes, max_det; float conf, iou;
};

// IoU(bi, bj) > thr with bi the earlier (higher-score) box; every operation is a single correctly rounded fp32 op in the
// oracle's order (oracle/nms.py), so the kept-index sequence is bit-exact
__device__ __forceinline__ bool nms_overlaps(const float4& bi, float ai, const float4& bj, float thr) {
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float aj = __fmul_rn(__fsub_rn(bj.z, bj.x), __fsub_rn(bj.w, bj.y));
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, aj), inter));
    return ovr > thr;
}

constexpr int NMS_THREADS = 1024;
constexpr int NMS_MAXA = 4096;
constexpr int NMS_LDS_BOXES = 2112;      // sorted boxes that fit in the LDS space of the (consumed) sort keys

static __global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const NmsParams p) {
    __shared__ unsigned long long keybuf[NMS_MAXA + 128];       // sort keys, later aliased by the sorted boxes (float4[A])
    __shared__ unsigned short order[NMS_MAXA];
    __shared__ unsigned char supp[NMS_MAXA];
    __shared__ int scan[NMS_THREADS];
    __shared__ float redf[NMS_THREADS];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float* dec = p.dec + long(b) * p.A * p.NC5;
    float* sc = p.scratch + long(b) * p.A * 8;
    int* sidx = p.scratch_idx + long(b) * p.A;
    constexpr int PER = NMS_MAXA / NMS_THREADS;                 // anchors per thread (contiguous block -> stable compaction)

    // ---- 1. per-anchor box / score / filter
    float bx[PER][4], bscore[PER], bconf[PER], bobj[PER];
    int bcls[PER], flag[PER];
    int mine = 0;
    float mymax = -3.0e38f;
    ACH_UNROLL
    for (int e = 0; e < PER; ++e) {
        const int a = tid * PER + e;
        flag[e] = 0;
        if (a < p.A) {
            const float* r = dec + long(a) * p.NC5;
            const float hw = __fdiv_rn(r[2], 2.0f), hh = __fdiv_rn(r[3], 2.0f);
            bx[e][0] = __fsub_rn(r[0], hw); bx[e][1] = __fsub_rn(r[1], hh);
            bx[e][2] = __fadd_rn(r[0], hw); bx[e][3] = __fadd_rn(r[1], hh);
            float best = r[5]; int bi = 0;
            // torch.max over the classes (utils_bbox.py:109) propagates NaN: the first NaN wins and stays, the score becomes NaN and
            // the `>= conf_thres` filter below drops the anchor
            for (int c = 1; c < p.num_classes; ++c) if (best == best && !(r[5 + c] <= best)) { best = r[5 + c]; bi = c; }
            bconf[e] = best; bcls[e] = bi; bobj[e] = r[4];
            bscore[e] = __fmul_rn(r[4], best);
            flag[e] = bscore[e] >= p.conf ? 1 : 0;
            if (flag[e]) { mine++; mymax = fmaxf(mymax, fmaxf(fmaxf(bx[e][0], bx[e][1]), fmaxf(bx[e][2], bx[e][3]))); }
        }
    }
    // ---- 2. exclusive scan of candidate counts (stable compaction) + max coordinate
    scan[tid] = mine;
    redf[tid] = mymax;
    __syncthreads();
    for (int off = 1; off < NMS_THREADS; off <<= 1) {
        const int v = tid >= off ? scan[tid - off] : 0;
        const float m = tid >= off ? redf[tid - off] : -3.0e38f;
        __syncthreads();
        scan[tid] += v;
        redf[tid] = fmaxf(redf[tid], m);
        __syncthreads();
    }
    const int n = scan[NMS_THREADS - 1];
    const float max1 = __fadd_rn(redf[NMS_THREADS - 1], 1.0f);
    int pos = scan[tid] - mine;
    for (int i = tid; i < NMS_MAXA; i += NMS_THREADS) keybuf[i] = ~0ull;
    __syncthreads();
    ACH_UNROLL
    for (int e = 0; e < PER; ++e) {
        if (!flag[e]) continue;
        const float off = __fmul_rn(float(bcls[e]), max1);
        float* q = sc + long(pos) * 8;
        q[0] = __fadd_rn(bx[e][0], off); q[1] = __fadd_rn(bx[e][1], off);
        q[2] = __fadd_rn(bx[e][2], off); q[3] = __fadd_rn(bx[e][3], off);
        q[4] = bscore[e]; q[5] = bconf[e]; q[6] = bobj[e]; q[7] = float(bcls[e]);
        sidx[pos] = tid * PER + e;
        keybuf[pos] = ((unsigned long long)(~order_encode(bscore[e])) << 32) | unsigned(pos);
        ++pos;
    }
    __syncthreads();
    // ---- 3. bitonic sort of the keys (ascending: descending score, ascending candidate index)
    int npow = 1;
    while (npow < n) npow <<= 1;
    for (int k = 2; k <= npow; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow; i += NMS_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keybuf[i], y = keybuf[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keybuf[i] = y; keybuf[ixj] = x; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += NMS_THREADS) order[i] = (unsigned short)(keybuf[i] & 0xffffull);
    __syncthreads();
    // ---- 4. sorted boxes into LDS (aliasing the key buffer; every key has been consumed above)
    // (more candidates than that — only possible above 320x320 — go through a global scratch copy instead; same arithmetic)
    float4* sbox = n <= NMS_LDS_BOXES ? reinterpret_cast<float4*>(keybuf) : reinterpret_cast<float4*>(p.scratch_boxes) + long(b) * p.A;
    for (int i = tid; i < n; i += NMS_THREADS) {
        const float* q = sc + long(order[i]) * 8;
        supp[i] = 0;
        sbox[i] = make_float4(q[0], q[1], q[2], q[3]);
    }
    __syncthreads();
    // ---- 5. greedy suppression in sorted order, 64 candidates per round (three barriers per round instead of one per candidate):
    //   a. all threads: the 64x64 "row suppresses column" bits inside the chunk (4 pairs per thread)
    //   b. wave 0: lane l assembles row l's mask; the greedy pass over the chunk runs in registers (masks read lane by lane);
    //      kept candidates write their output rows
    //   c. all threads: every later, still-alive candidate is tested against the boxes the chunk kept
    // A candidate is kept iff no EARLIER KEPT candidate overlaps it by more than the threshold — exactly the sequential rule.
    __shared__ unsigned char pm[64][16];            // 4 pair bits per entry
    __shared__ unsigned long long s_keep;
    int kept_total = 0;                             // same value in every thread
    for (int c0 = 0; c0 < n && kept_total < p.max_det; c0 += 64) {
        {   // a.
            const int row = tid >> 4, cb = (tid & 15) * 4;
            const int i = c0 + row;
            unsigned bits = 0;
            if (i < n && supp[i] == 0) {
                const float4 bi = sbox[i];
                const float ai = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
                ACH_UNROLL
                for (int e = 0; e < 4; ++e) {
                    const int col = cb + e, j = c0 + col;
                    if (col > row && j < n && nms_overlaps(bi, ai, sbox[j], p.iou)) bits |= 1u << e;
                }
            }
            pm[row][tid & 15] = (unsigned char)bits;
        }
        __syncthreads();
        if (tid < 64) {   // b.
            const int i = c0 + tid;
            const bool alive = i < n && supp[i] == 0;
            unsigned long long m = 0;
            ACH_UNROLL
            for (int q = 0; q < 16; ++q) m |= (unsigned long long)(pm[tid][q]) << (4 * q);
            const unsigned long long alive_bits = wave_ballot64(alive);
            unsigned long long removed = ~alive_bits, keep = 0;
            for (int l = 0; l < 64; ++l) {
                const unsigned long long ml = wave_read64(m, l);
                if (!((removed >> l) & 1ull)) { keep |= 1ull << l; removed |= ml; }
            }
            if (tid == 0) s_keep = keep;
            if ((keep >> tid) & 1ull) {
                const int slot = kept_total + __popcll(keep & ((1ull << tid) - 1ull));
                if (slot < p.max_det) {
                    const int cand = order[i];
                    const float* q = sc + long(cand) * 8;
                    float* r = p.rows + (long(b) * p.max_det + slot) * 7;
                    // original (un-offset) corners are recomputed from the decoded row, as the reference gathers them
                    const int a = sidx[cand];
                    const float* d = dec + long(a) * p.NC5;
                    const float hw = __fdiv_rn(d[2], 2.0f), hh = __fdiv_rn(d[3], 2.0f);
                    r[0] = __fsub_rn(d[0], hw); r[1] = __fsub_rn(d[1], hh); r[2] = __fadd_rn(d[0], hw); r[3] = __fadd_rn(d[1], hh);
                    r[4] = q[6]; r[5] = q[5]; r[6] = q[7];
                    p.kept[long(b) * p.max_det + slot] = a;
                }
            }
        }
        __syncthreads();
        const unsigned long long keep = s_keep;
        kept_total += __popcll(keep);
        for (int j = c0 + 64 + tid; j < n; j += NMS_THREADS) {   // c.
            if (supp[j]) continue;
            const float4 bj = sbox[j];
            unsigned long long k = keep;
            while (k) {
                const int l = __ffsll((long long)k) - 1;
                k &= k - 1ull;
                const float4 bi = sbox[c0 + l];
                const float ai = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
                if (nms_overlaps(bi, ai, bj, p.iou)) { supp[j] = 1; break; }
            }
        }
        __syncthreads();
    }
    // unused output slots: zero rows, index -1 (the caller hands over uninitialised buffers; kept_total is uniform)
    for (int slot = (kept_total < p.max_det ? kept_total : p.max_det) + tid; slot < p.max_det; slot += NMS_THREADS) {
        float* r = p.rows + (long(b) * p.max_det + slot) * 7;
        ACH_UNROLL
        for (int c = 0; c < 7; ++c) r[c] = 0.f;
        p.kept[long(b) * p.max_det + slot] = -1;
    }
    if (tid == 0) p.count[b] = kept_total < p.max_det ? kept_total : p.max_det;
}

}  // namespace ach
