// engine_impl.h — Engine<T>: the launch plan of the whole path for one storage type (see engine.h).  Included by
// engine_f32.cpp and engine_bf16.cpp, which instantiate it once each so that the two halves compile in parallel.
#pragma once
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "k_dechead.h"
#include "k_csphead.h"
#include "k_upchain.h"
#include "k_detect.h"
#include "k_conv3.h"
#include "k_gemm.h"
#include "k_ghost.h"
#include "k_headdw.h"
#include "k_mlp.h"
#include "k_mlpband.h"
#include "k_mv2.h"
#include "k_mvit.h"
#include "k_nhwc.h"
#include "k_points.h"
#include "k_pn2.h"
#include "k_prepost.h"
#include "k_radar.h"
#include <functional>
#include "k_sdta.h"
#include "k_xca.h"
#include "k_xcaframe.h"

namespace ach {

#define ACH_HIP_CHECK(expr)                                                                               \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) throw AchError{ACH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)}; \
    } while (0)

static inline long round_up(long v, long m) { return (v + m - 1) / m * m; }

// ================================================================================================ Engine<T>
template <class T>
class Engine final : public EngineBase {
    using S = Store<T>;
    static constexpr int VEC = S::VEC;
    static constexpr int KC = 4 * VEC;
    static constexpr bool H16E = is_h16<T>::value;                  // bf16 or fp16 storage: the production kernels (k_dechead.h, k_mlpband.h, k_headdw.h, k_upchain.h)
    // the alternative type of the CALLER's tensors: the fp16-storage engine also takes / returns bf16 (option "io_bf16"); for the other two it is T itself
    using IOB = typename std::conditional<std::is_same<T, f16_t>::value, bf16_t, T>::type;
    bool io_alt() const { return std::is_same<T, f16_t>::value && io_bf16; }

public:
    explicit Engine(const ach_config& c) : EngineBase(c) {}

    struct A {              // NHWC activation view
        T* p = nullptr; int B = 0, H = 0, W = 0, C = 0; long ld = 0;
        long rows() const { return long(B) * H * W; }
        A slice(int c0, int c) const { A s = *this; s.p = p + c0; s.C = c; return s; }
    };
    struct Pl {             // planar NCHW tensor (radar branch)
        T* p = nullptr; int B = 0, C = 0, H = 0, W = 0;
    };
    struct Lin { std::vector<float> w, b; int N = 0, K = 0; };      // w[n*K + k]

    // ------------------------------------------------------------------------------------------ helpers
    A alloc(int B, int H, int W, int C) {
        A a; a.B = B; a.H = H; a.W = W; a.C = C; a.ld = round_up(C, 8);
        a.p = static_cast<T*>(aalloc(size_t(a.rows()) * a.ld * sizeof(T)));
        note_region(a.p, size_t(a.rows()) * a.ld * sizeof(T));
        return a;
    }
    A alloc_ld(int B, int H, int W, int C, long ld) {          // explicit pixel pitch (the 4-channel maps of the first RCBlock)
        A a; a.B = B; a.H = H; a.W = W; a.C = C; a.ld = ld;
        a.p = static_cast<T*>(aalloc(size_t(a.rows()) * a.ld * sizeof(T)));
        note_region(a.p, size_t(a.rows()) * a.ld * sizeof(T));
        return a;
    }
    Pl alloc_pl(int B, int C, int H, int W) {
        Pl a; a.B = B; a.C = C; a.H = H; a.W = W;
        a.p = static_cast<T*>(aalloc(size_t(B) * C * H * W * sizeof(T)));
        note_region(a.p, size_t(B) * C * H * W * sizeof(T));
        return a;
    }
    float* alloc_f32(size_t n) { return static_cast<float*>(aalloc(n * sizeof(float))); }

    T* up_T(const std::vector<float>& v) {
        T* d = static_cast<T*>(walloc(v.size() * sizeof(T)));
        if (!measuring) {
            std::vector<T> h(v.size());
            for (size_t i = 0; i < v.size(); ++i) S::st(&h[i], v[i]);
            ACH_HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
        }
        return d;
    }
    void tap(const std::string& name, const A& a) {
        TapInfo t; t.ptr = a.p; t.kind = 0; t.B = a.B; t.H = a.H; t.W = a.W; t.C = a.C; t.ld = a.ld; add_tap(name, t);
    }
    void tap(const std::string& name, const Pl& a) {
        TapInfo t; t.ptr = a.p; t.kind = 1; t.B = a.B; t.H = a.H; t.W = a.W; t.C = a.C; add_tap(name, t);
    }

    // ---- linear-layer algebra on the host
    Lin lin(const std::string& wkey, const std::string& bkey) const {
        const HostTensor& w = W(wkey);
        if (w.shape.size() < 2) throw AchError{ACH_ERR_MISSING_KEY, "not a matrix: " + wkey};
        Lin l; l.N = int(w.shape[0]); l.K = int(w.numel() / w.shape[0]);
        l.w = w.data;
        if (!bkey.empty() && hasW(bkey)) { l.b = W(bkey).data; if (int(l.b.size()) != l.N) throw AchError{ACH_ERR_MISSING_KEY, "bias shape: " + bkey}; }
        else l.b.assign(size_t(l.N), 0.f);
        return l;
    }
    void bn_coeffs(const std::string& pfx, double eps, std::vector<float>& scale, std::vector<float>& shift) const {
        const auto& g = W(pfx + ".weight").data; const auto& b = W(pfx + ".bias").data;
        const auto& m = W(pfx + ".running_mean").data; const auto& v = W(pfx + ".running_var").data;
        scale.resize(g.size()); shift.resize(g.size());
        for (size_t i = 0; i < g.size(); ++i) {
            const double s = double(g[i]) / std::sqrt(double(v[i]) + eps);
            scale[i] = float(s); shift[i] = float(double(b[i]) - double(m[i]) * s);
        }
    }
    void fold_bn(Lin& l, const std::string& pfx, double eps) const {            // y = bn(Wx + b)
        std::vector<float> sc, sh; bn_coeffs(pfx, eps, sc, sh);
        if (int(sc.size()) != l.N) throw AchError{ACH_ERR_MISSING_KEY, "BatchNorm width mismatch at " + pfx};
        for (int n = 0; n < l.N; ++n) {
            for (int k = 0; k < l.K; ++k) l.w[size_t(n) * l.K + k] *= sc[n];
            l.b[n] = l.b[n] * sc[n] + sh[n];
        }
    }
    void fold_ln_in(Lin& l, const std::string& pfx) const {                      // y = W(lnw * xhat + lnb) + b
        const auto& lw = W(pfx + ".weight").data; const auto& lb = W(pfx + ".bias").data;
        if (int(lw.size()) != l.K) throw AchError{ACH_ERR_MISSING_KEY, "LayerNorm width mismatch at " + pfx};
        for (int n = 0; n < l.N; ++n) {
            double acc = l.b[n];
            for (int k = 0; k < l.K; ++k) { acc += double(l.w[size_t(n) * l.K + k]) * lb[k]; l.w[size_t(n) * l.K + k] *= lw[k]; }
            l.b[n] = float(acc);
        }
    }
    void fold_scale_out(Lin& l, const std::vector<float>& g) const {             // y = g * (Wx + b)
        for (int n = 0; n < l.N; ++n) { for (int k = 0; k < l.K; ++k) l.w[size_t(n) * l.K + k] *= g[n]; l.b[n] *= g[n]; }
    }

    std::vector<std::function<void()>> post_plan;            // run once when the plan is built, after the arena has been zeroed
    struct Packed { T* w = nullptr; float* b = nullptr; int N = 0, K = 0, NT = 1, nchunks = 0, ksteps = 0; long group_elems = 0; };
    static int pick_nt(int N) { return N <= 16 ? 1 : (N <= 32 ? 2 : 4); }
    Packed pack_shape(int N, int K) const {
        Packed p; p.N = N; p.K = K; p.NT = pick_nt(N);
        p.nchunks = cdiv(N, 16 * p.NT); p.ksteps = cdiv(K, KC);
        p.group_elems = long(p.nchunks) * p.ksteps * p.NT * 64 * VEC;
        return p;
    }
    Packed pack(const Lin& l) {
        Packed p = pack_shape(l.N, l.K);
        std::vector<float> blob(size_t(p.group_elems), 0.f);
        if (!measuring)
            for (int n = 0; n < l.N; ++n)
                for (int k = 0; k < l.K; ++k) blob[size_t(wfrag_offset(n, k, p.NT, p.ksteps, VEC))] = l.w[size_t(n) * l.K + k];
        p.w = up_T(blob);
        p.b = up_f32(l.b);
        return p;
    }

    struct GemmOpt {
        int act = ACT_NONE; bool ln = false; float ln_eps = 0.f;
        bool ln_tap = false;                                                      // conv mode, 2x2: LayerNorm of every tap's input pixel fused in (k_gemm.h LNTAP)
        const A* residual = nullptr;
        int conv_k = 0, conv_s = 1, conv_p = 0, Hin = 0, Win = 0, Cin = 0, Ho = 0, Wo = 0;   // implicit-GEMM conv over NHWC
        int Creal = 0;                                                            // conv mode: real channels per pixel (Cin is the stored pitch) for the byte accounting
        void** ydyn = nullptr; int out_nchw = 0, HW = 0, Ctot = 0, coff = 0;     // NCHW scatter into a user buffer
        int groups = 1; long w_group_stride = 0;
        const T* w_override = nullptr;                                            // data-dependent packed weights
    };
    // rows view: X.p, X.ld, rows = M ; K = pk.K
    void gemm(const std::string& name, const T* Xp, long ldx, long M, const Packed& pk, T* Yp, long ldy, const GemmOpt& o) {
        GemmParams g;
        std::memset(&g, 0, sizeof(g));
        g.X = Xp; g.ldx = ldx; g.conv_k = o.conv_k; g.conv_s = o.conv_s; g.conv_p = o.conv_p; g.Hin = o.Hin; g.Win = o.Win; g.Cin = o.Cin; g.Ho = o.Ho; g.Wo = o.Wo;
        g.W = o.w_override ? o.w_override : pk.w; g.w_group_stride = o.w_group_stride;
        g.bias = pk.b; g.bias_group_stride = 0;
        g.Y = Yp; g.ldy = ldy;
        g.R = o.residual ? o.residual->p : nullptr; g.ldr = o.residual ? o.residual->ld : 0;
        g.groups = o.groups; g.M_per_group = int(M / o.groups);
        g.K = pk.K; g.N = pk.N; g.nchunks = pk.nchunks; g.ksteps = pk.ksteps;
        g.act = o.act; g.ln = o.ln_tap ? 2 : (o.ln ? 1 : 0); g.ln_eps = o.ln_eps;
        if (o.ln_tap && (o.conv_k != 2 || pk.NT != 4 || o.groups != 1 || batching || o.Cin % VEC != 0)) throw AchError{ACH_ERR_INVALID, name + ": per-tap LayerNorm needs a 2x2 conv with more than 32 outputs"};
        g.out_nchw = (o.out_nchw && io_alt()) ? 2 : o.out_nchw; g.HW = o.HW; g.Ctot = o.Ctot; g.coff = o.coff;
        g.vec_store = (ldy % 8 == 0 && (!o.residual || o.residual->ld % 8 == 0)) ? 1 : 0;
        if (ldx % VEC != 0) throw AchError{ACH_ERR_INVALID, name + ": activation row stride not 16-byte aligned"};
        if (o.conv_k > 0 && (o.Cin % VEC != 0 || pk.K != o.conv_k * o.conv_k * o.Cin)) throw AchError{ACH_ERR_UNSUPPORTED, name + ": conv channel count not 16-byte aligned"};
        // launch geometry: >= ~4 workgroups per CU.  Rows first (P sub-tiles of 16 rows per wave), then split the
        // N-chunks over blockIdx.z when the row count alone cannot fill 256 CUs (10x10 / 20x20 maps, FC layers).
        // (measured on MI355X, tests/gpu_gemm_bench.py: one 16-row sub-tile per wave beats 2 or 4 at every shape of this
        //  network — the kernel is latency/bandwidth bound and lives on occupancy, not on weight-fragment reuse)
        const long kTargetBlocks = gemm_blocks > 0 ? gemm_blocks : 1024;
        // ... except the dense 3x3 convs of the MobileViT blocks (K = 9 x 2C = 2592 .. 4320): every 16-row tile re-reads the whole packed
        // weight (0.5-0.9 MB) from L2, so two / four sub-tiles per wave halve / quarter that stream (option `gemm_rows`: sub-tiles per wave for K >= 1024)
        int P = (!batching && pk.K >= 1024 && o.groups == 1 && gemm_rows > 1) ? gemm_rows : 1;
        while (P > 1 && cdivl(g.M_per_group, 64L * P) * pk.nchunks < 512) P >>= 1;          // keep >= two workgroups per CU
        const long row_blocks = cdivl(g.M_per_group, 64L * P) * g.groups;
        int zsplit = int(std::min<long>(pk.nchunks, std::max<long>(1, cdivl(kTargetBlocks, row_blocks))));
        g.chunks_per_block = cdiv(pk.nchunks, zsplit);
        const int NT = pk.NT;
        void** ydyn = o.ydyn;
        const double esz = double(sizeof(T));
        const double px_in = o.conv_k > 0 ? double(M) / (double(o.Ho) * o.Wo) * o.Hin * o.Win : 0.0;
        const double in_bytes = o.conv_k > 0 ? px_in * (o.Creal ? o.Creal : o.Cin) * esz : double(M) * pk.K * esz;   // inputs read ONCE, real channels
        const double rest = double(M) * pk.N * esz + (o.residual ? double(M) * pk.N * esz : 0.0)
                            + double(pk.group_elems) * esz * (o.w_group_stride ? o.groups : 1);
        const double bytes = in_bytes + rest;
        const double lbytes = (o.conv_k > 0 ? px_in * o.Cin * esz : in_bytes) + rest;
        if (batching) {           // collected now, emitted by flush_batch() as one launch per layer across the pyramid levels
            if (g.groups != 1 || P != 1) throw AchError{ACH_ERR_INVALID, name + ": batched GEMMs must be ungrouped"};
            BatchJob j; j.kind = 0; j.name = name; j.g = g; j.NT = NT; j.ydyn = ydyn; j.bytes = bytes; j.flops = 2.0 * double(M) * (o.conv_k > 0 && o.Creal ? double(o.conv_k * o.conv_k * o.Creal) : double(pk.K)) * pk.N;
            batch_jobs.push_back(j);
            return;
        }
        add_op(name, [g, NT, P, ydyn](hipStream_t s) mutable {
            if (ydyn) g.Y = *ydyn;
            launch_gemm<T>(g, NT, P, s);
        }, bytes, 2.0 * double(M) * (o.conv_k > 0 && o.Creal ? double(o.conv_k * o.conv_k * o.Creal) : double(pk.K)) * pk.N, lbytes);
    }
    // ---- batching of the same layer over the detection head's pyramid levels
    struct BatchJob { int kind = 0; std::string name; GemmParams g; int NT = 1; void** ydyn = nullptr; DwParams d; int ks = 0; double bytes = 0, flops = 0; HeadDwJob hj; int shared_in = 0; };
    bool batching = false;
    std::vector<BatchJob> batch_jobs;
    // `levels` chains of `per_level` jobs each were recorded level by level; emit stage s of all levels as one launch
    void flush_batch(int levels, int per_level) {
        batching = false;
        if (int(batch_jobs.size()) != levels * per_level || levels > 3) throw AchError{ACH_ERR_INVALID, "head batching: unexpected job count"};
        for (int st = 0; st < per_level; ++st) {
            const BatchJob& j0 = batch_jobs[size_t(st)];
            double bytes = 0, flops = 0;
            const std::string name = j0.name + "+levels";
            if (j0.kind == 0) {
                GemmJobs m;
                std::memset(&m, 0, sizeof(m));
                m.n = levels;
                unsigned gx = 1, gz = 1;
                void** yd[3] = {nullptr, nullptr, nullptr};
                for (int l = 0; l < levels; ++l) {
                    const BatchJob& j = batch_jobs[size_t(l * per_level + st)];
                    if (j.kind != 0 || j.NT != j0.NT) throw AchError{ACH_ERR_INVALID, "head batching: layer shapes differ across levels"};
                    m.p[l] = j.g;
                    m.nbx[l] = unsigned(cdivl(j.g.M_per_group, 64L));
                    m.nbz[l] = unsigned(cdiv(j.g.nchunks, j.g.chunks_per_block));
                    gx = std::max(gx, m.nbx[l]); gz = std::max(gz, m.nbz[l]);
                    yd[l] = j.ydyn; bytes += j.bytes; flops += j.flops;
                }
                const dim3 grid(gx, unsigned(levels), gz), block(256);
                const int NT = j0.NT;
                void** y0 = yd[0]; void** y1 = yd[1]; void** y2 = yd[2];
                add_op(name, [m, grid, block, NT, y0, y1, y2](hipStream_t s) mutable {
                    if (y0) m.p[0].Y = *y0;
                    if (y1) m.p[1].Y = *y1;
                    if (y2) m.p[2].Y = *y2;
                    if (NT == 1) ACH_LAUNCH((gemm_multi_kernel<T, 1>), grid, block, s, m);
                    else if (NT == 2) ACH_LAUNCH((gemm_multi_kernel<T, 2>), grid, block, s, m);
                    else ACH_LAUNCH((gemm_multi_kernel<T, 4>), grid, block, s, m);
                }, bytes, flops);
            } else if (j0.kind == 2) {              // fused depthwise 5x5 + pointwise layer of the head towers (k_headdw.h), one job per level
                HeadDwParams hp;
                std::memset(&hp, 0, sizeof(hp));
                hp.njobs = levels; hp.shared_in = j0.shared_in; hp.dbg = head_fuse_dbg;
                int wg = 0;
                for (int l = 0; l < levels; ++l) {
                    const BatchJob& j = batch_jobs[size_t(l * per_level + st)];
                    if (j.kind != 2 || j.shared_in != j0.shared_in) throw AchError{ACH_ERR_INVALID, "head batching: fused layers differ across levels"};
                    hp.job[l] = j.hj; hp.job[l].wg0 = wg;
                    hp.B = j.d.B;
                    wg += 2 * j.hj.bands * j.d.B;
                    bytes += j.bytes; flops += j.flops;
                }
                const dim3 grid(static_cast<unsigned>(wg)), block(HDW_THREADS);
                // (if constexpr: instantiated by the bf16 engine's translation unit ONLY — see fused_mlp_lin)
                if constexpr (H16E)
                    add_op(name, [hp, grid, block](hipStream_t s) { ACH_LAUNCH((headdw_kernel<T>), grid, block, s, hp); }, bytes, flops);
                else throw AchError{ACH_ERR_INVALID, "head batching: the fused layer exists for 16-bit storage only"};
            } else {
                DwJobs m;
                std::memset(&m, 0, sizeof(m));
                m.n = levels;
                unsigned gx = 1;
                for (int l = 0; l < levels; ++l) {
                    const BatchJob& j = batch_jobs[size_t(l * per_level + st)];
                    if (j.kind != 1 || j.ks != 5 || j.d.stride != 1) throw AchError{ACH_ERR_INVALID, "head batching: depthwise shapes differ across levels"};
                    m.p[l] = j.d;
                    m.nbx[l] = unsigned(cdivl(long(j.d.B) * j.d.Ho * cdiv(j.d.Wo, 4) * (j.d.C / 4), 256));
                    gx = std::max(gx, m.nbx[l]);
                    bytes += j.bytes;
                }
                const dim3 grid(gx, unsigned(levels)), block(256);
                add_op(name, [m, grid, block](hipStream_t s) { ACH_LAUNCH((dwconv_strip_multi_kernel<T, 5, 4>), grid, block, s, m); }, bytes, 0);
            }
        }
        batch_jobs.clear();
    }
    void gemm(const std::string& name, const A& X, const Packed& pk, const A& Y, const GemmOpt& o = GemmOpt()) {
        gemm(name, X.p, X.ld, X.rows(), pk, Y.p, Y.ld, o);
    }

    // depthwise conv (+ folded BN) on NHWC views
    void dwconv(const std::string& name, const A& X, const A* X2, const std::string& wkey, const std::string& bkey,
                const std::string& bnpfx, double bneps, int ks, int stride, int act, const A& Y) {
        const HostTensor& w = W(wkey);
        const int C = X.C;
        if (w.shape[0] != C || w.numel() != long(C) * ks * ks) throw AchError{ACH_ERR_MISSING_KEY, "depthwise weight shape: " + wkey};
        if (C % 4) throw AchError{ACH_ERR_UNSUPPORTED, name + ": depthwise channel count must be a multiple of 4"};
        std::vector<float> wt(size_t(ks) * ks * C), bias(size_t(C), 0.f), sc(size_t(C), 1.f), sh(size_t(C), 0.f);
        if (!bkey.empty()) bias = W(bkey).data;
        if (!bnpfx.empty()) bn_coeffs(bnpfx, bneps, sc, sh);
        for (int c = 0; c < C; ++c) {
            for (int t = 0; t < ks * ks; ++t) wt[size_t(t) * C + c] = w.data[size_t(c) * ks * ks + t] * sc[c];
            bias[c] = bias[c] * sc[c] + sh[c];
        }
        if (double(X.rows()) * double(std::max(X.ld, X2 ? X2->ld : 0L)) * sizeof(T) >= 2147483648.0)
            throw AchError{ACH_ERR_UNSUPPORTED, name + ": depthwise input of 2 GiB or more (batch too large for one plan)"};
        DwParams p;
        std::memset(&p, 0, sizeof(p));
        p.X = X.p; p.ldx = X.ld; p.X2 = X2 ? X2->p : nullptr; p.ldx2 = X2 ? X2->ld : 0;
        p.W = up_f32(wt); p.bias = up_f32(bias); p.Y = Y.p; p.ldy = Y.ld;
        p.B = X.B; p.H = X.H; p.Wd = X.W; p.C = C; p.Ho = Y.H; p.Wo = Y.W; p.stride = stride; p.act = act;
        p.tile = (dw_tile && X.H * X.W <= 144) ? 1 : 0;      // measured: wins on 10x10 maps (2x), loses from 20x20 up (LDS issue bound)
        const double px_in = double(X.B) * X.H * X.W * C * sizeof(T), px_out = double(Y.B) * Y.H * Y.W * C * sizeof(T);
        add_op(name, [p, ks](hipStream_t s) { launch_dwconv<T>(p, ks, s); }, px_in * (X2 ? 2 : 1) + px_out, 0);
    }

    template <class K, class Pm>
    void ew(const std::string& name, K kern, const Pm& p, long total, double bytes = 0, double layout_bytes = -1) {      // 256-thread element-wise launch
        const dim3 grid(unsigned(cdivl(total, 256))), block(256);
        add_op(name, [kern, p, grid, block](hipStream_t s) { ACH_LAUNCH(kern, grid, block, s, p); }, bytes, 0, layout_bytes);
    }
    void add(const std::string& name, const A& a, const A& b, const A& y) {
        AddParams p{a.p, a.ld, b.p, b.ld, y.p, y.ld, a.rows(), a.C};
        ew(name, add_kernel<T>, p, a.rows() * (a.C / 4), 3.0 * a.rows() * a.C * sizeof(T));
    }
    void copy(const std::string& name, const A& x, const A& y, const float* posenc = nullptr) {
        CopyParams p{x.p, x.ld, y.p, y.ld, x.rows(), x.C, posenc, x.H * x.W};
        ew(name, copy_kernel<T>, p, x.rows() * (x.C / 4), 2.0 * x.rows() * x.C * sizeof(T));
    }
    // per-(sample, channel) sums over H*W -> partial [B][S][2][C] ; returns S
    int stats(const std::string& name, const A& x, float*& partial) {
        const int HW = x.H * x.W;
        const int S = HW >= 1024 ? 8 : (HW >= 256 ? 4 : 1);
        partial = alloc_f32(size_t(x.B) * S * 2 * x.C);
        StatParams p{x.p, x.ld, partial, HW, x.C, S};
        const dim3 grid(unsigned(x.B), unsigned(S)), block(256);
        add_op(name, [p, grid, block](hipStream_t s) { ACH_LAUNCH(chan_stats_kernel<T>, grid, block, s, p); });
        return S;
    }

    // ------------------------------------------------------------------------------------------ EdgeNeXt (a2-a5)
    struct EnCfg { int depths[4]; int dims[4]; int heads; int scales[4]; int ks[4]; };
    EnCfg en_cfg() const {
        switch (cfg.phi) {
            case ACH_PHI_S0: return {{2, 2, 6, 2}, {32, 48, 96, 176}, 4, {2, 2, 3, 4}, {3, 5, 7, 9}};
            case ACH_PHI_S1: return {{3, 3, 9, 3}, {32, 48, 120, 224}, 4, {2, 2, 3, 4}, {3, 5, 7, 9}};
            default: return {{3, 3, 9, 3}, {32, 64, 144, 288}, 8, {2, 2, 3, 4}, {3, 5, 7, 9}};
        }
    }
    // constant-folded Fourier positional encoding [HW][C] (edgenext_modules/layers.py:38-59)
    float* posenc_table(const std::string& pfx, int H, int Wd, int C) {
        const int hidden = 32;
        const HostTensor& w = W(pfx + ".token_projection.weight");
        const HostTensor& b = W(pfx + ".token_projection.bias");
        std::vector<float> tab(size_t(H) * Wd * C);
        if (!measuring) {
            std::vector<float> feat(2 * hidden);
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < Wd; ++x) {
                    const float ye = float(y + 1) / (float(H) + 1e-6f) * float(2.0 * M_PI);
                    const float xe = float(x + 1) / (float(Wd) + 1e-6f) * float(2.0 * M_PI);
                    for (int i = 0; i < hidden; ++i) {
                        const float dim_t = std::pow(10000.0f, float(2 * (i / 2)) / float(hidden));
                        const float py = ye / dim_t, px = xe / dim_t;
                        feat[i] = (i % 2 == 0) ? std::sin(py) : std::cos(py);
                        feat[hidden + i] = (i % 2 == 0) ? std::sin(px) : std::cos(px);
                    }
                    for (int c = 0; c < C; ++c) {
                        double acc = b.data[c];
                        for (int k = 0; k < 2 * hidden; ++k) acc += double(w.data[size_t(c) * 2 * hidden + k]) * feat[k];
                        tab[(size_t(y) * Wd + x) * C + c] = float(acc);
                    }
                }
        }
        return up_f32(tab);
    }

    A pw_mlp(const std::string& pfx, const A& xin, const A& resid) {     // LN -> Linear -> GELU -> Linear -> gamma -> + resid
        const int C = xin.C;
        Lin l1 = lin(pfx + ".pwconv1.weight", pfx + ".pwconv1.bias");
        fold_ln_in(l1, pfx + ".norm");
        Lin l2 = lin(pfx + ".pwconv2.weight", pfx + ".pwconv2.bias");
        fold_scale_out(l2, W(pfx + ".gamma").data);
        A h = alloc(xin.B, xin.H, xin.W, 4 * C);
        GemmOpt o1; o1.act = ACT_GELU; o1.ln = true; o1.ln_eps = 1e-6f;
        gemm(pfx + ".pwconv1", xin, pack(l1), h, o1);
        A y = block_out(xin);
        GemmOpt o2; o2.residual = &resid;
        gemm(pfx + ".pwconv2", h, pack(l2), y, o2);
        return y;
    }
    // One launch for [depthwise kxk ->] LN -> Linear -> act -> Linear -> scale -> + resid (k_mlp.h).  Returns false when the
    // width is outside the kernel's instantiations (the caller then runs the layer-wise path).
    // ConvEncoder blocks that took the band kernel, in plan order (index of their launch): merge_band_runs turns each run of consecutive ones — block i + 1 reads what
    // block i wrote, same geometry, nothing else of the plan in between — into ONE persistent launch (k_mlpband.h mlp_band_run_kernel; option "mlp_band_run")
    struct BandOp { size_t op; MlpBandParams bp; int nb, shape; };
    std::vector<BandOp> band_ops;
    void merge_band_runs() {
#if !defined(ACH_HOSTEMU)
        if constexpr (H16E) {
            if (measuring || !mlp_band_run || use_graph) { band_ops.clear(); return; }
            for (size_t a = 0; a < band_ops.size();) {
                size_t e = a + 1;
                while (e < band_ops.size() && e - a < size_t(MLPB_RUN_MAX) && band_ops[e].op == band_ops[e - 1].op + 1 && band_ops[e].shape == band_ops[a].shape && band_ops[e].nb == band_ops[a].nb &&
                       band_ops[e].bp.m.X == band_ops[e - 1].bp.m.Y && band_ops[e].bp.bands == band_ops[a].bp.bands && band_ops[e].bp.rb == band_ops[a].bp.rb &&
                       ops[band_ops[e].op].stream == ops[band_ops[a].op].stream && ops[band_ops[e].op].wait_ev < 0 && ops[band_ops[e].op].wait_ev2 < 0 && !ops[band_ops[e].op].xwait &&
                       !ops[band_ops[e].op].xwait2 && !ops[band_ops[e].op].xwait3 && !ops[band_ops[e - 1].op].xsignal3 && ops[band_ops[e - 1].op].signal_ev < 0 && !ops[band_ops[e - 1].op].xsignal && !ops[band_ops[e - 1].op].xsignal2) ++e;
                const size_t n = e - a;
                // (the bands of a frame must share an XCD — xcd_block: the launch's workgroups in eight equal chunks of whole frames — because the blocks hand their rows over through that L2)
                const long nwg = long(band_ops[a].bp.bands) * band_ops[a].nb;
                if (n >= 2 && mlp_band_run_shape(band_ops[a].shape) && nwg % 8 == 0 && (nwg / 8) % band_ops[a].bp.bands == 0) {
                    MlpBandRunParams rp;
                    std::memset(&rp, 0, sizeof(rp));
                    for (size_t k = 0; k < n; ++k) rp.blk[k] = band_ops[a + k].bp;
                    rp.n = int(n);
                    std::vector<unsigned> zeros(size_t(band_ops[a].nb), 0u);
                    rp.sync = static_cast<unsigned*>(up_raw(zeros.data(), zeros.size() * sizeof(unsigned)));
                    const int nb = band_ops[a].nb, shape = band_ops[a].shape;
                    const size_t first = band_ops[a].op, last = band_ops[e - 1].op;
                    Op& op = ops[first];
                    for (size_t k = first + 1; k <= last; ++k) { op.bytes += ops[k].bytes; op.layout_bytes += ops[k].layout_bytes; op.flops += ops[k].flops; }
                    op.signal_ev = ops[last].signal_ev; op.xsignal = ops[last].xsignal; op.xsignal2 = ops[last].xsignal2;
                    op.name += "..+" + std::to_string(n - 1);
                    op.fn = [rp, nb, shape](hipStream_t s) mutable { launch_mlp_band_run<T>(rp, shape, nb, s); ++rp.epoch; };
                    ops.erase(ops.begin() + long(first) + 1, ops.begin() + long(last) + 1);
                    for (size_t k = e; k < band_ops.size(); ++k) band_ops[k].op -= n - 1;
                }
                a = e;
            }
        }
#endif
        band_ops.clear();
    }
    bool fused_mlp(const std::string& pfx, const A& xin, const A& resid, int dw_ks, A& y) {
        if (!fuse_mlp || mlp_pick_dt(xin.C) == 0) return false;
        Lin l1 = lin(pfx + ".pwconv1.weight", pfx + ".pwconv1.bias");
        fold_ln_in(l1, pfx + ".norm");
        Lin l2 = lin(pfx + ".pwconv2.weight", pfx + ".pwconv2.bias");
        fold_scale_out(l2, W(pfx + ".gamma").data);
        return fused_mlp_lin(pfx + (dw_ks ? ".block" : ".mlp"), pfx, xin, resid, dw_ks, l1, l2, ACT_GELU, 1e-6f, y);
    }
    // generic form: l1 has the LayerNorm affine folded in, l2 any output scale; `pfx`.dwconv.{weight,bias} when dw_ks > 0
    // When set, the next block output is written there instead of a fresh allocation (a channel slice of a concat buffer of the
    // neck: torch.cat((upsampled, backbone feature), 1) then needs no copy).  Consumed by the first block that produces an output.
    A preset_out; bool has_preset = false;
    bool neck_on_side = false;        // this plan runs the neck on stream 2 (option dec_fork = 3; decided in build())
    A block_out(const A& like) {
        if (has_preset) {
            has_preset = false;
            if (neck_on_side) mark_xwait3_next();      // (the launch that writes this slice of the neck's concat buffer: the previous forward's neck, on stream 2, may still read it)
            if (preset_out.B != like.B || preset_out.H != like.H || preset_out.W != like.W || preset_out.C != like.C)
                throw AchError{ACH_ERR_INVALID, "preset block output does not match the block"};
            return preset_out;
        }
        return alloc(like.B, like.H, like.W, like.C);
    }
    bool fused_mlp_lin(const std::string& name, const std::string& pfx, const A& xin, const A& resid, int dw_ks, const Lin& l1, const Lin& l2,
                       int act, float ln_eps, A& y) {
        if (!fuse_mlp) return false;
        const int C = xin.C, DT = mlp_pick_dt(C);
        if (DT == 0 || (dw_ks != 0 && dw_ks != 3 && dw_ks != 5 && dw_ks != 7 && dw_ks != 9)) return false;
        const int hidden = l1.N, k1 = cdiv(C, KC), J = cdiv(hidden, 32), hstep = 8 / VEC, ks2 = J * hstep;
        if (l1.K != C || l2.N != C || l2.K != hidden) throw AchError{ACH_ERR_MISSING_KEY, "MLP shapes at " + pfx};
        std::vector<float> w1(size_t(J) * k1 * 2 * 64 * VEC, 0.f), b1(size_t(J) * 32, 0.f);
        std::vector<float> w2(size_t(ks2) * DT * 64 * VEC, 0.f), b2(size_t(DT) * 16, 0.f);
        if (!measuring) {
            for (int n = 0; n < hidden; ++n) {
                b1[n] = l1.b[n];
                for (int k = 0; k < C; ++k) w1[size_t(wfrag_offset(n, k, 2, k1, VEC))] = l1.w[size_t(n) * C + k];
            }
            for (int n = 0; n < C; ++n) {
                b2[n] = l2.b[n];
                for (int kap = 0; kap < 32 * J; ++kap) {
                    const int ch = mlp_hidden_channel(kap, VEC);
                    if (ch < hidden) w2[size_t(wfrag_offset(n, kap, DT, ks2, VEC))] = l2.w[size_t(n) * hidden + ch];
                }
            }
        }
        MlpParams mp;
        std::memset(&mp, 0, sizeof(mp));
        mp.X = xin.p; mp.ldx = xin.ld; mp.R = resid.p; mp.ldr = resid.ld;
        y = block_out(xin);
        mp.Y = y.p; mp.ldy = y.ld;
        mp.dw_k = dw_ks; mp.H = xin.H; mp.W = xin.W;
        if (dw_ks) {
            const double xb = double(xin.rows()) * double(xin.ld) * sizeof(T);
            if (xb >= 2147483648.0) throw AchError{ACH_ERR_UNSUPPORTED, "depthwise input of 2 GiB or more (batch too large for one plan)"};
            mp.xbytes = unsigned(xb);
        }
        if (dw_ks) {
            const HostTensor& w = W(pfx + ".dwconv.weight");
            const std::vector<float>& b = W(pfx + ".dwconv.bias").data;
            const int ldc = k1 * KC, kk = dw_ks * dw_ks;
            if (size_t(w.numel()) != size_t(C) * size_t(kk)) throw AchError{ACH_ERR_MISSING_KEY, "depthwise shape at " + pfx};
            std::vector<float> wt(size_t(kk) * ldc, 0.f), bt(size_t(ldc), 0.f);
            for (int c = 0; c < C; ++c) { bt[c] = b[c]; for (int t = 0; t < kk; ++t) wt[size_t(t) * ldc + c] = w.data[size_t(c) * kk + t]; }
            mp.Wdw = up_f32(wt); mp.bdw = up_f32(bt);
        }
        mp.W1 = up_T(w1); mp.b1 = up_f32(b1); mp.W2 = up_T(w2); mp.b2 = up_f32(b2);
        mp.M = xin.rows(); mp.C = C; mp.k1 = k1; mp.J = J; mp.act = act; mp.ln_eps = ln_eps; mp.ln = 1; mp.Cout = C;
        // per-sample map size, not batch, decides the geometry: a frame's result must not depend on the batch it is in
        const bool split = mlp_split < 0 ? xin.H * xin.W <= mlp_split_hw : mlp_split != 0;
        if (split && dw_ks && dw_even && mlp_even_dt(DT)) {               // the instantiated widths (launch_mlp)
            // deal the k1 * k tap rows to the four waves in contiguous shares when that lowers the slowest wave's count and no share spans
            // more than two k-steps (the exchange buffer holds two slots per wave)
            const int U = k1 * dw_ks, chunk = cdiv(U, 4);
            bool ok = chunk < cdiv(k1, 4) * dw_ks;
            for (int w = 0; w < 4 && ok; ++w) {
                const int u0 = w * chunk, u1 = std::min(U, u0 + chunk);
                if (u0 < u1 && (u1 - 1) / dw_ks > u0 / dw_ks + 1) ok = false;
            }
            mp.dw_even = ok ? 1 : 0;
        }
        const double bytes = double(mp.M) * C * sizeof(T) * (xin.p == resid.p ? 2.0 : 3.0);
        const double flops = 4.0 * double(mp.M) * C * hidden + (dw_ks ? 2.0 * double(mp.M) * C * dw_ks * dw_ks : 0.0);
        // small maps, bf16: a band of rows per workgroup — taps from an LDS halo tile, MLP weights fetched once per band (k_mlpband.h)
        // (if constexpr: kernels that only the bf16 engine launches must not be instantiated by the fp32 engine's translation unit as well —
        //  the two code objects would carry the same symbol and the runtime registers one of them for the shared host stub)
        if constexpr (H16E) if (mlp_band && (split || mlp_band > 1) && dw_ks && act == ACT_GELU && xin.p == resid.p && mlp_band_supported(k1, DT, dw_ks, xin.H, xin.W)) {
            MlpBandParams bp;
            bp.m = mp; bp.rb = mlp_band_rows(k1, DT, dw_ks, xin.H, xin.W);
            if (band_rows_s3 > 0 && mlp_band_shape(k1, DT, dw_ks, xin.W) == 2) bp.rb = std::min(std::min(band_rows_s3, 5), xin.H);     // (the 10 x 10 maps: 5-row bands are 128 workgroups at batch 64)
            bp.bands = cdiv(xin.H, bp.rb); bp.dbg = mlp_band_dbg;
            int shape = mlp_band_shape(k1, DT, dw_ks, xin.W);
            if (shape == 1 && mlp_band_lean) shape = 11;
            const int nb = xin.B;
            if (!measuring) band_ops.push_back(BandOp{ops.size(), bp, nb, shape});          // (merge_band_runs: consecutive blocks of a stage as one launch)
            add_op(name, [bp, nb, shape](hipStream_t s) { launch_mlp_band<T>(bp, shape, nb, s); }, bytes, flops);
            return true;
        }
        // plain rows, wide blocks (MobileViT's feed-forward layers, d = 144 / 192): two tiles per wave — half the weight traffic (k_mlp.h ffn2_kernel); bit-identical
        // (large maps only — the per-sample size decides, as for `split`: on the 20 x 20 maps 800 two-tile waves are too few, 54 -> 118 us measured)
        if constexpr (H16E) if (ffn_rows2 && !split && !dw_ks && (DT == 10 || DT == 12) && mp.M >= long(ffn_rows2_min)) {
            add_op(name, [mp, DT](hipStream_t s) { launch_ffn2<T>(mp, DT, s); }, bytes, flops);
            return true;
        }
        add_op(name, [mp, DT, split](hipStream_t s) { launch_mlp<T>(mp, DT, split, s); }, bytes, flops);
        return true;
    }
    // packed weights of a two-layer 1x1 chain (mlp_kernel's layouts with DT output tiles) into `mp`
    void chain_weights(MlpParams& mp, int Cin, int DT, const Lin& l1, int act, const Lin& l2) {
        const int hidden = l1.N, Cout = l2.N;
        const int k1 = cdiv(Cin, KC), J = cdiv(hidden, 32), hstep = 8 / VEC, ks2 = J * hstep;
        std::vector<float> w1(size_t(J) * k1 * 2 * 64 * VEC, 0.f), b1(size_t(J) * 32, 0.f);
        std::vector<float> w2(size_t(ks2) * DT * 64 * VEC, 0.f), b2(size_t(DT) * 16, 0.f);
        if (!measuring) {
            for (int n = 0; n < hidden; ++n) {
                b1[n] = l1.b[n];
                for (int k = 0; k < Cin; ++k) w1[size_t(wfrag_offset(n, k, 2, k1, VEC))] = l1.w[size_t(n) * Cin + k];
            }
            for (int n = 0; n < Cout; ++n) {
                b2[n] = l2.b[n];
                for (int kap = 0; kap < 32 * J; ++kap) {
                    const int ch = mlp_hidden_channel(kap, VEC);
                    if (ch < hidden) w2[size_t(wfrag_offset(n, kap, DT, ks2, VEC))] = l2.w[size_t(n) * hidden + ch];
                }
            }
        }
        mp.W1 = up_T(w1); mp.b1 = up_f32(b1); mp.W2 = up_T(w2); mp.b2 = up_f32(b2);
        mp.C = Cin; mp.k1 = k1; mp.J = J; mp.act = act; mp.ln = 0; mp.Cout = Cout;
    }
    // y = l2(act(l1(x))) as one launch (k_mlp.h without LayerNorm / residual); false when the widths are not instantiated
    // (`allow_split` = false: the caller's "map" is not a frame — pc_pair presents B*N points as one map, whose size depends on the BATCH; the four-waves-per-tile
    //  mode sums in another fp32 order than the one-wave mode, and a frame's result must not depend on the batch it is in)
    bool chain2(const std::string& name, const A& x, const Lin& l1, int act, const Lin& l2, A& y, bool* planar = nullptr, bool allow_split = true) {
        if (!fuse_mlp) return false;
        const int Cin = x.C, hidden = l1.N, Cout = l2.N;
        const int k1 = cdiv(Cin, KC), J = cdiv(hidden, 32);
        const bool split = allow_split && (mlp_split < 0 ? x.H * x.W <= mlp_split_hw : mlp_split != 0);
        // narrow layers on maps that are not latency-bound: every weight fragment in registers, several tiles per wave (chain_kernel)
        const bool small = Cout <= 32 && k1 <= 3 && J <= 2 && !split;
        const int DT = small ? 2 : mlp_pick_dt(std::max(Cin, Cout));
        if (DT == 0 || l1.K != Cin || l2.K != hidden) return false;
        y = alloc(x.B, x.H, x.W, Cout);
        MlpParams mp;
        std::memset(&mp, 0, sizeof(mp));
        mp.X = x.p; mp.ldx = x.ld; mp.Y = y.p; mp.ldy = y.ld;
        chain_weights(mp, Cin, DT, l1, act, l2);
        mp.M = x.rows();
        if (planar) {                            // channel-planar rows for the MFMA bilinear phase of the fused last decoder level
            *planar = *planar && small && Cout == 16;
            if (*planar) mp.planar_w = x.W;
        }
        add_op(name, [mp, DT, split, small](hipStream_t s) { if (small) launch_chain<T>(mp, s); else launch_mlp<T>(mp, DT, split, s); },
               double(mp.M) * (Cin + Cout) * sizeof(T), 2.0 * double(mp.M) * hidden * (Cin + Cout));
        return true;
    }
    // the same pair with BOTH outputs kept, into existing (slices of) tensors: hdst = act(l1 x), ydst = l2 hdst — chain_kernel only (narrow layers, hidden a multiple of 32)
    bool chain2_into(const std::string& name, const A& x, const Lin& l1, int act, const Lin& l2, const A& ydst, const A& hdst) {
        if (!fuse_mlp || VEC != 8) return false;
        const int Cin = x.C, hidden = l1.N, Cout = l2.N;
        const int k1 = cdiv(Cin, KC), J = cdiv(hidden, 32);
        if (!(Cout <= 32 && k1 <= 3 && J <= 2) || hidden % 32 || l1.K != Cin || l2.K != hidden || hdst.C != hidden || ydst.C != Cout || hdst.ld % 8 || ydst.ld % 8) return false;
        MlpParams mp;
        std::memset(&mp, 0, sizeof(mp));
        mp.X = x.p; mp.ldx = x.ld; mp.Y = ydst.p; mp.ldy = ydst.ld; mp.Hout = hdst.p; mp.ldh = hdst.ld;
        chain_weights(mp, Cin, 2, l1, act, l2);
        mp.M = x.rows();
        add_op(name, [mp](hipStream_t s) { launch_chain<T>(mp, s); }, double(mp.M) * (Cin + hidden + Cout) * sizeof(T), 2.0 * double(mp.M) * hidden * (Cin + Cout));
        return true;
    }
    A conv_encoder(const std::string& pfx, const A& x, int ks) {          // conv_encoder.py:19-32
        A fy;
        if (fused_mlp(pfx, x, x, ks, fy)) return fy;
        A d = alloc(x.B, x.H, x.W, x.C);
        dwconv(pfx + ".dwconv", x, nullptr, pfx + ".dwconv.weight", pfx + ".dwconv.bias", "", 0, ks, 1, ACT_NONE, d);
        return pw_mlp(pfx, d, x);
    }
    A sdta_encoder(const std::string& pfx, const A& x, int scales, int heads) {   // sdta_encoder.py:39-74
        const int C = x.C;
        const int width = std::max((C + scales - 1) / scales, C / scales);
        const int nums = scales - 1;
        if ((C / heads) > 64) throw AchError{ACH_ERR_UNSUPPORTED, "XCA head dimension > 64"};
        if (width % 4) throw AchError{ACH_ERR_UNSUPPORTED, "SDTA split width must be a multiple of 4"};
        A y = alloc(x.B, x.H, x.W, C);
        const int HW = x.H * x.W, tailc = C - nums * width, Q = sdta_pre_quads(x.H, x.W, width / 4);
        const bool has_pos = hasW(pfx + ".pos_embd.token_projection.weight");
        if (sdta_fuse && (sdta_fuse > 1 || HW <= 400) && Q > 0 && nums >= 1 && tailc >= 0 && tailc % 4 == 0) {
            // cascade of depthwise 3x3 convs + tail copy + positional encoding as one launch (k_sdta.h)
            std::vector<float> wt(size_t(nums) * 9 * width), bs(size_t(nums) * width);
            for (int i = 0; i < nums; ++i) {
                const HostTensor& w = W(pfx + ".convs." + std::to_string(i) + ".weight");
                const HostTensor& bi = W(pfx + ".convs." + std::to_string(i) + ".bias");
                if (w.numel() != long(width) * 9 || bi.numel() != width) throw AchError{ACH_ERR_MISSING_KEY, "SDTA conv shapes at " + pfx};
                for (int c = 0; c < width; ++c) { bs[size_t(i) * width + c] = bi.data[c]; for (int t = 0; t < 9; ++t) wt[(size_t(i) * 9 + t) * width + c] = w.data[size_t(c) * 9 + t]; }
            }
            SdtaPreParams sp{x.p, x.ld, y.p, y.ld, up_f32(wt), up_f32(bs), has_pos ? posenc_table(pfx + ".pos_embd", x.H, x.W, C) : nullptr,
                             x.B, x.H, x.W, C, width, nums, Q, cdiv(width / 4, Q), cdiv(tailc / 4, Q)};
            const dim3 grid(unsigned(x.B) * unsigned(sp.conv_wgs + sp.tail_wgs)), block(SDTA_THREADS);
            add_op(pfx + ".sdta_pre", [sp, grid, block](hipStream_t s) { ACH_LAUNCH(sdta_pre_kernel<T>, grid, block, s, sp); },
                   2.0 * double(x.rows()) * C * sizeof(T));
        } else {
            for (int i = 0; i < nums; ++i) {
                A xi = x.slice(i * width, width), yi = y.slice(i * width, width);
                A prev = i > 0 ? y.slice((i - 1) * width, width) : A();
                const std::string c = pfx + ".convs." + std::to_string(i);
                dwconv(c, xi, i > 0 ? &prev : nullptr, c + ".weight", c + ".bias", "", 0, 3, 1, ACT_NONE, yi);
            }
            copy(pfx + ".split_tail", x.slice(nums * width, C - nums * width), y.slice(nums * width, C - nums * width));
            if (has_pos) copy(pfx + ".pos_embd", y, y, posenc_table(pfx + ".pos_embd", x.H, x.W, C));
        }
        // XCA
        Lin lq = lin(pfx + ".xca.qkv.weight", pfx + ".xca.qkv.bias");
        fold_ln_in(lq, pfx + ".norm_xca");
        A qkv = alloc(x.B, x.H, x.W, 3 * C);
        const Packed pq = pack(lq);
        const int d = C / heads, N = x.H * x.W;
        // heads per workgroup of the MFMA Gram kernel: the largest divisor of `heads` whose channels fit the 64-channel staging tile and
        // whose tiles are at most six per wave (k_xca.h)
        const int tmx = cdiv(d, 16), per_head = tmx * tmx + 2 * tmx;
        int hg = 0;
        for (int c = 1; c <= heads; ++c) if (heads % c == 0 && c * d <= 64 && c * per_head <= 24) hg = c;
        Packed pe = pack_shape(C, C);
        T* weff = static_cast<T*>(aalloc(size_t(x.B) * pe.group_elems * sizeof(T)));
        Lin lp = lin(pfx + ".xca.proj.weight", pfx + ".xca.proj.bias");
        const std::vector<float>& gx = W(pfx + ".gamma_xca").data;
        std::vector<float> bproj(static_cast<size_t>(C), 0.f);
        for (int c = 0; c < C; ++c) bproj[c] = lp.b[c] * gx[c];
        pe.b = up_f32(bproj);
        A t2 = alloc(x.B, x.H, x.W, C);
        // ---- the attention in two launches (k_xcaframe.h; 16-bit engines; option `xca_frame`: 2 = front + back kernels, 1 = ONE launch with a workgroup per frame
        //      (measured slower), 0 = the four launches of rounds 1-5)
        bool framed = false;
        if constexpr (H16E) {
            const int KSf = cdiv(d, KC), CT = cdiv(C, 16), DR = tmx * 16, KP = KSf * KC + VEC;
            const bool tiny = C * (d + 3) <= XCAF_TINY_AFL && heads * DR * KP <= XCAF_TINY_PEL;
            const bool small = d <= 48 && C * (d + 3) <= XCAF_SMALL_AFL && heads * DR * KP <= XCAF_SMALL_PEL;
            const bool big = d <= 64 && C * (d + 3) <= XCAF_BIG_AFL && heads * DR * KP <= XCAF_BIG_PEL;
            if ((xca_frame == 1 || xca_frame == 2) && !full_taps && xca_mfma && d % 2 == 0 && hg > 0 && (small || big) && pq.NT == 4 && pe.NT == 4 &&
                qkv.ld % 8 == 0 && y.ld % 8 == 0 && t2.ld % 8 == 0) {
                const bool two = xca_frame == 2;
                // token slices of the front kernel: 64 tokens (128 on the 40 x 40 maps: 13 partials per frame to sum instead of 25); one slice in the one-launch form
                const int per = !two ? ((N + 15) / 16) * 16 : (xca_slice > 0 ? ((xca_slice + 15) / 16) * 16 : (N >= 1024 ? 128 : 64));
                const int S = cdiv(N, per);
                GemmParams g1;
                std::memset(&g1, 0, sizeof(g1));
                g1.X = y.p; g1.ldx = y.ld; g1.W = pq.w; g1.bias = pq.b; g1.Y = qkv.p; g1.ldy = qkv.ld;
                g1.groups = x.B; g1.M_per_group = N; g1.K = pq.K; g1.N = pq.N; g1.nchunks = pq.nchunks; g1.ksteps = pq.ksteps; g1.chunks_per_block = 1;
                g1.act = ACT_NONE; g1.ln = 1; g1.ln_eps = 1e-6f; g1.vec_store = 1;
                GemmParams g2;
                std::memset(&g2, 0, sizeof(g2));
                g2.X = qkv.p + 2 * C; g2.ldx = qkv.ld; g2.W = weff; g2.w_group_stride = pe.group_elems; g2.bias = pe.b; g2.Y = t2.p; g2.ldy = t2.ld;
                g2.R = y.p; g2.ldr = y.ld; g2.groups = x.B; g2.M_per_group = N; g2.K = C; g2.N = C; g2.nchunks = pe.nchunks; g2.ksteps = pe.ksteps; g2.chunks_per_block = 1;
                g2.act = ACT_NONE; g2.vec_store = 1;
                XcaGramParams gp{qkv.p, qkv.ld, alloc_f32(size_t(x.B) * heads * S * (d * d + 2 * d)), x.B, N, C, heads, S, hg};
                gp.per = per;
                std::vector<float> wpg(size_t(heads) * CT * KSf * 64 * VEC, 0.f);
                for (int h = 0; h < heads; ++h)
                    for (int ct = 0; ct < CT; ++ct)
                        for (int s2 = 0; s2 < KSf; ++s2)
                            for (int l = 0; l < 64; ++l)
                                for (int jj = 0; jj < VEC; ++jj) {
                                    const int n = ct * 16 + (l & 15), k = s2 * KC + (l >> 4) * VEC + jj;
                                    if (n < C && k < d) wpg[((size_t(h) * CT + ct) * KSf + s2) * 64 * VEC + size_t(l) * VEC + jj] = gx[n] * lp.w[size_t(n) * C + h * d + k];
                                }
                XcaFoldParams fo{gp.partial, S, up_f32(W(pfx + ".xca.temperature").data), up_T(wpg), nullptr, C, heads, d, KSf, CT};
                const double wbytes = double(pq.group_elems + heads * CT * KSf * 64 * VEC) * sizeof(T);
                const double flops = 2.0 * double(x.rows()) * C * (4.0 * C + 2.0 * d) + 2.0 * double(x.B) * C * d * d;
                if (!two) {
                    XcaFrameParams fp;
                    std::memset(&fp, 0, sizeof(fp));
                    fp.qkv = g1; fp.proj = g2; fp.gram = gp; fp.fold = fo; fp.N = N; fp.heads = heads;
                    const dim3 grid(static_cast<unsigned>(x.B));
                    add_op(pfx + ".xca.frame", [fp, grid, small](hipStream_t s) {
                               if (small) ACH_LAUNCH((xca_frame_kernel<T, 48, XCAF_SMALL_AFL, XCAF_SMALL_PEL, 16>), grid, dim3(1024), s, fp);
                               else ACH_LAUNCH((xca_frame_kernel<T, 64, XCAF_BIG_AFL, XCAF_BIG_PEL, 16>), grid, dim3(1024), s, fp); },
                           2.0 * double(x.rows()) * C * sizeof(T) + wbytes, flops);
                } else {
                    XcaFrontParams fr;
                    std::memset(&fr, 0, sizeof(fr));
                    fr.qkv = g1; fr.gram = gp; fr.N = N; fr.S = S;
                    // waves per workgroup: enough for a slice's (tile, chunk) units in about two rounds
                    const int funits = (per / 16) * pq.nchunks;
                    const int fw = xca_front_waves > 0 ? xca_front_waves : (funits >= 24 ? 16 : (funits >= 12 ? 8 : 4));
                    const dim3 gridf(static_cast<unsigned>(x.B * S));
                    const bool gsmall = hg * d <= 48;
                    add_op(pfx + ".xca.qkv+gram", [fr, gridf, gsmall, fw](hipStream_t s) {
#define ACH_XCAF_FRONT(dm, nw) ACH_LAUNCH((xca_front_kernel<T, dm, nw>), gridf, dim3(64 * nw), s, fr)
                               if (gsmall) { if (fw == 16) ACH_XCAF_FRONT(48, 16); else if (fw == 8) ACH_XCAF_FRONT(48, 8); else ACH_XCAF_FRONT(48, 4); }
                               else { if (fw == 16) ACH_XCAF_FRONT(64, 16); else if (fw == 8) ACH_XCAF_FRONT(64, 8); else ACH_XCAF_FRONT(64, 4); }
#undef ACH_XCAF_FRONT
                           }, double(x.rows()) * C * sizeof(T) + double(pq.group_elems) * sizeof(T), 2.0 * double(x.rows()) * C * (3.0 * C + 2.0 * d));
                    XcaBackParams bk;
                    std::memset(&bk, 0, sizeof(bk));
                    bk.proj = g2; bk.fold = fo; bk.N = N; bk.RB = cdiv(N, 64);
                    const int bw = xca_back_waves > 0 ? xca_back_waves : (heads * tmx * CT >= 64 ? 16 : (heads * tmx * CT >= 24 ? 8 : 4));
                    const dim3 gridb(static_cast<unsigned>(x.B * bk.RB));
                    add_op(pfx + ".xca.fold+proj", [bk, gridb, tiny, small, bw](hipStream_t s) {
#define ACH_XCAF_BACK(af, pl, nw) ACH_LAUNCH((xca_back_kernel<T, af, pl, nw>), gridb, dim3(64 * nw), s, bk)
                               if (tiny) { if (bw == 16) ACH_XCAF_BACK(XCAF_TINY_AFL, XCAF_TINY_PEL, 16); else if (bw == 8) ACH_XCAF_BACK(XCAF_TINY_AFL, XCAF_TINY_PEL, 8); else ACH_XCAF_BACK(XCAF_TINY_AFL, XCAF_TINY_PEL, 4); }
                               else if (small) { if (bw == 16) ACH_XCAF_BACK(XCAF_SMALL_AFL, XCAF_SMALL_PEL, 16); else if (bw == 8) ACH_XCAF_BACK(XCAF_SMALL_AFL, XCAF_SMALL_PEL, 8); else ACH_XCAF_BACK(XCAF_SMALL_AFL, XCAF_SMALL_PEL, 4); }
                               else { if (bw == 16) ACH_XCAF_BACK(XCAF_BIG_AFL, XCAF_BIG_PEL, 16); else if (bw == 8) ACH_XCAF_BACK(XCAF_BIG_AFL, XCAF_BIG_PEL, 8); else ACH_XCAF_BACK(XCAF_BIG_AFL, XCAF_BIG_PEL, 4); }
#undef ACH_XCAF_BACK
                           }, double(x.rows()) * C * sizeof(T) + double(heads * CT * KSf * 64 * VEC) * sizeof(T), 2.0 * double(x.rows()) * C * C + 2.0 * double(x.B) * C * d * d);
                }
                framed = true;
            }
        }
        if (!framed) {
        GemmOpt oq; oq.ln = true; oq.ln_eps = 1e-6f;
        gemm(pfx + ".xca.qkv", y, pq, qkv, oq);
        // Gram matrices over token slices, then softmax + fold into per-sample projection weights (see k_xca.h)
        const int S = N >= 1024 ? 8 : (N >= 256 ? 4 : 1);
        float* partial = alloc_f32(size_t(x.B) * heads * S * (d * d + 2 * d));
        XcaGramParams pg{qkv.p, qkv.ld, partial, x.B, N, C, heads, S, hg};
        {
            const dim3 grid(unsigned(x.B * heads), unsigned(S)), block(256);
            // channel pairs are fetched as one 4-byte (bf16) / 8-byte (fp32) load; where a group's channels are not whole 16-byte pieces
            // (d = 18, 30 in bf16) and the Gram matrix is small, the VALU kernel is faster (measured: EN-S2 stage 2, 30.7 vs 35.4 us)
            const bool mf = xca_mfma && d % 2 == 0 && hg > 0 && ((hg * d) % VEC == 0 || tmx >= 3);
            const dim3 gridm(unsigned(x.B * (mf ? heads / hg : 1)), unsigned(S));
            const bool small = hg * d <= 48;
            add_op(pfx + ".xca.gram", [pg, grid, gridm, block, d, mf, small](hipStream_t s) {
                       if (mf) { if (small) ACH_LAUNCH((xca_gram_mfma_kernel<T, 48>), gridm, block, s, pg); else ACH_LAUNCH((xca_gram_mfma_kernel<T, 64>), gridm, block, s, pg); }
                       else if (d <= 48) ACH_LAUNCH((xca_gram_kernel<T, 48>), grid, block, s, pg); else ACH_LAUNCH((xca_gram_kernel<T, 64>), grid, block, s, pg); },
                   2.0 * x.rows() * C * sizeof(T));
        }
        XcaFinalParams pf{partial, S, up_f32(W(pfx + ".xca.temperature").data), up_f32(lp.w), up_f32(gx), nullptr, weff, pe.group_elems,
                          x.B, C, heads, pe.NT, pe.ksteps};
        bool fold_done = false;
        if constexpr (H16E) {
            // the same launch with the fold on the matrix cores, one workgroup per (frame, head) (k_xcaframe.h xca_fold_body; option xca_fold_mfma)
            const int KSf = cdiv(d, KC), CT = cdiv(C, 16), DR = tmx * 16, KP = KSf * KC + VEC;
            const bool tiny = C * (d + 3) <= XCAF_TINY_AFL && heads * DR * KP <= XCAF_TINY_PEL;
            const bool small = d <= 48 && C * (d + 3) <= XCAF_SMALL_AFL && heads * DR * KP <= XCAF_SMALL_PEL;
            const bool big = d <= 64 && C * (d + 3) <= XCAF_BIG_AFL && heads * DR * KP <= XCAF_BIG_PEL;
            if (xca_fold_mfma && !full_taps && (tiny || small || big) && pe.NT == 4) {
                std::vector<float> wpg(size_t(heads) * CT * KSf * 64 * VEC, 0.f);
                for (int h = 0; h < heads; ++h)
                    for (int ct = 0; ct < CT; ++ct)
                        for (int s2 = 0; s2 < KSf; ++s2)
                            for (int l = 0; l < 64; ++l)
                                for (int jj = 0; jj < VEC; ++jj) {
                                    const int n = ct * 16 + (l & 15), k = s2 * KC + (l >> 4) * VEC + jj;
                                    if (n < C && k < d) wpg[((size_t(h) * CT + ct) * KSf + s2) * 64 * VEC + size_t(l) * VEC + jj] = gx[n] * lp.w[size_t(n) * C + h * d + k];
                                }
                XcaFinalMfmaParams fm;
                std::memset(&fm, 0, sizeof(fm));
                fm.proj.W = weff; fm.proj.w_group_stride = pe.group_elems; fm.proj.ksteps = pe.ksteps;
                fm.fold = XcaFoldParams{partial, S, pf.temperature, up_T(wpg), nullptr, C, heads, d, KSf, CT};
                const dim3 grid(unsigned(x.B * heads)), block(256);
                add_op(pfx + ".xca.finalize", [fm, grid, block, tiny, small](hipStream_t s) {
                           if (tiny) ACH_LAUNCH((xca_finalize_mfma_kernel<T, XCAF_TINY_AFL, XCAF_TINY_PEL>), grid, block, s, fm);
                           else if (small) ACH_LAUNCH((xca_finalize_mfma_kernel<T, XCAF_SMALL_AFL, XCAF_SMALL_PEL>), grid, block, s, fm);
                           else ACH_LAUNCH((xca_finalize_mfma_kernel<T, XCAF_BIG_AFL, XCAF_BIG_PEL>), grid, block, s, fm); });
                fold_done = true;
            }
        }
        if (!fold_done) {
            const dim3 grid(unsigned(x.B * heads), unsigned(cdiv(C, XCA_CT))), block(256);
            add_op(pfx + ".xca.finalize", [pf, grid, block, d](hipStream_t s) { if (d <= 48) ACH_LAUNCH((xca_finalize_kernel<T, 48>), grid, block, s, pf); else ACH_LAUNCH((xca_finalize_kernel<T, 64>), grid, block, s, pf); });
        }
        // t2 = y + gamma_xca * proj(attn @ v): one GEMM over v (channel slice [2C,3C) of qkv) with per-sample weights
        {
            GemmOpt op; op.residual = &y; op.groups = x.B; op.w_group_stride = pe.group_elems; op.w_override = weff;
            gemm(pfx + ".xca.proj", qkv.p + 2 * C, qkv.ld, qkv.rows(), pe, t2.p, t2.ld, op);
        }
        }
        A fy;
        if (fused_mlp(pfx, t2, x, 0, fy)) return fy;
        return pw_mlp(pfx, t2, x);
    }
    // `dst[i]` (optional, may be null): where the output of stage i is to be written
    void edgenext(const std::string& pfx, A feats[4], const A* const dst[4]) {                   // edgenext.py:73-86
        const EnCfg ec = en_cfg();
        const int B = batch, R = cfg.resolution;
        A x;
        for (int i = 0; i < 4; ++i) {
            const std::string d = pfx + ".downsample_layers." + std::to_string(i);
            if (i == 0) {
                const HostTensor& w = W(d + ".0.weight");
                if (w.shape.size() != 4 || w.shape[0] != 32 || w.shape[1] != 3 || w.shape[2] != 4) throw AchError{ACH_ERR_UNSUPPORTED, "stem shape"};
                std::vector<float> wt(48 * 32);
                for (int o = 0; o < 32; ++o)
                    for (int k = 0; k < 48; ++k) wt[size_t(k) * 32 + o] = w.data[size_t(o) * 48 + k];
                x = alloc(B, R / 4, R / 4, 32);
                const void** img = &io.image;
                Lin ls; ls.N = 32; ls.K = 48; ls.w = w.data; ls.b = W(d + ".0.bias").data;            // k = c*16 + dy*4 + dx: the conv weight's own order
                Packed pk = pack(ls);
                if (stem_mfma && pk.NT == 2 && pk.nchunks == 1 && x.ld == 32) {
                    StemMfmaParams sp{nullptr, x.p, pk.w, pk.b, up_f32(W(d + ".1.weight").data), up_f32(W(d + ".1.bias").data), B, R, R, pk.ksteps, 1e-6f};
                    const dim3 grid(unsigned(cdivl(x.rows(), 64))), block(256);
                    const bool alt = io_alt();
                    add_op(d, [sp, grid, block, img, alt](hipStream_t s) mutable { sp.X = *img; if (alt) ACH_LAUNCH((stem_mfma_kernel<T, IOB>), grid, block, s, sp); else ACH_LAUNCH((stem_mfma_kernel<T, T>), grid, block, s, sp); },
                           double(B) * 3 * R * R * sizeof(T) + double(x.rows()) * 32 * sizeof(T), 2.0 * double(x.rows()) * 48 * 32);
                } else {
                    StemParams sp{nullptr, x.p, up_f32(wt), up_f32(W(d + ".0.bias").data), up_f32(W(d + ".1.weight").data),
                                  up_f32(W(d + ".1.bias").data), B, R, R, 1e-6f};
                    const dim3 grid(unsigned(cdivl(x.rows(), 256))), block(256);
                    const bool alt = io_alt();
                    add_op(d, [sp, grid, block, img, alt](hipStream_t s) mutable { sp.X = *img; if (alt) ACH_LAUNCH((stem_kernel<T, IOB>), grid, block, s, sp); else ACH_LAUNCH((stem_kernel<T, T>), grid, block, s, sp); });
                }
            } else {
                // conv 2x2 stride 2: k = (dy, dx, c) over two contiguous NHWC segments
                const HostTensor& w = W(d + ".1.weight");
                const int Co = int(w.shape[0]), Ci = int(w.shape[1]);
                if (Ci != x.C || Ci % 8 != 0) throw AchError{ACH_ERR_UNSUPPORTED, "downsample conv shape"};
                Lin l; l.N = Co; l.K = 4 * Ci; l.w.resize(size_t(Co) * 4 * Ci); l.b = W(d + ".1.bias").data;
                for (int o = 0; o < Co; ++o)
                    for (int c = 0; c < Ci; ++c)
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dx = 0; dx < 2; ++dx)
                                l.w[size_t(o) * 4 * Ci + (dy * 2 + dx) * Ci + c] = w.data[((size_t(o) * Ci + c) * 2 + dy) * 2 + dx];
                A y = alloc(x.B, x.H / 2, x.W / 2, Co);
                GemmOpt o; o.conv_k = 2; o.conv_s = 2; o.conv_p = 0; o.Hin = x.H; o.Win = x.W; o.Cin = Ci; o.Ho = x.H / 2; o.Wo = x.W / 2;
                if (ds_fuse && !full_taps && pick_nt(Co) == 4) {
                    // the channels-first LayerNorm in front of the conv (edgenext.py:29-34) inside the conv's k-loop: statistics per input pixel = per tap,
                    // the affine part folded into the weights (k_gemm.h LNTAP) — one launch and one tensor less per down-sampling layer
                    const auto& lw = W(d + ".0.weight").data; const auto& lb = W(d + ".0.bias").data;
                    if (int(lw.size()) != Ci) throw AchError{ACH_ERR_MISSING_KEY, "LayerNorm width mismatch at " + d};
                    for (int n = 0; n < Co; ++n) {
                        double acc = l.b[n];
                        for (int k = 0; k < 4 * Ci; ++k) { acc += double(l.w[size_t(n) * 4 * Ci + k]) * lb[k % Ci]; l.w[size_t(n) * 4 * Ci + k] *= lw[k % Ci]; }
                        l.b[n] = float(acc);
                    }
                    o.ln_tap = true; o.ln_eps = 1e-6f;
                    gemm(d + ".ln+conv", x.p, x.ld, y.rows(), pack(l), y.p, y.ld, o);
                } else {
                    A t = alloc(x.B, x.H, x.W, x.C);
                    if (t.ld != t.C) throw AchError{ACH_ERR_UNSUPPORTED, "downsample conv shape"};
                    int G = 1;
                    while (G < x.C / 4 && G < 64) G <<= 1;
                    LnParams lp{x.p, x.ld, t.p, t.ld, up_f32(W(d + ".0.weight").data), up_f32(W(d + ".0.bias").data), x.rows(), x.C, 1e-6f, G};
                    const dim3 grid(unsigned(cdivl(x.rows(), 256 / G))), block(256);
                    add_op(d + ".0", [lp, grid, block](hipStream_t s) { ACH_LAUNCH(layernorm_kernel<T>, grid, block, s, lp); });
                    gemm(d + ".1", t.p, t.ld, y.rows(), pack(l), y.p, y.ld, o);
                    tap("backbone.ds" + std::to_string(i) + ".ln", t);
                }
                tap("backbone.ds" + std::to_string(i), y);
                x = y;
            }
            for (int j = 0; j < ec.depths[i]; ++j) {
                const std::string b = pfx + ".stages." + std::to_string(i) + "." + std::to_string(j);
                if (j == ec.depths[i] - 1 && dst && dst[i]) { preset_out = *dst[i]; has_preset = true; }
                if (i > 0 && j == ec.depths[i] - 1) x = sdta_encoder(b, x, ec.scales[i], ec.heads);
                else x = conv_encoder(b, x, ec.ks[i]);
                tap("backbone.s" + std::to_string(i) + ".b" + std::to_string(j), x);
            }
            if (has_preset) throw AchError{ACH_ERR_INVALID, "stage output destination was not consumed"};
            feats[i] = x;
            merge_band_runs();
            if (i == radar_start_eff()) signal_after_last(0);
        }
    }

    // ------------------------------------------------------------------------------------------ MobileViT (a6)
    struct MvCfg { int dims[3]; int ch[11]; int exp; };
    MvCfg mv_cfg() const {
        switch (cfg.phi) {
            case ACH_PHI_S0: return {{64, 80, 96}, {16, 16, 32, 32, 48, 48, 96, 96, 96, 96, 176}, 2};
            case ACH_PHI_S1: return {{96, 120, 144}, {16, 32, 32, 32, 48, 48, 120, 120, 120, 120, 224}, 4};
            default: return {{144, 192, 240}, {16, 32, 32, 32, 64, 64, 144, 144, 144, 144, 288}, 4};
        }
    }
    // nn.Sequential(conv kxk (no bias), BatchNorm(1e-5), SiLU) as an (implicit-)GEMM   (mobilevit.py:6-19)
    A mv_conv(const std::string& pfx, const A& x, int k, int stride) {
        const HostTensor& w = W(pfx + ".0.weight");
        Lin l = (k == 1) ? lin(pfx + ".0.weight", "") : conv_lin(pfx + ".0.weight", "", int(w.shape[1]), int(x.ld), k);
        if (k == 1 && l.K != x.C) throw AchError{ACH_ERR_MISSING_KEY, "conv width at " + pfx};
        fold_bn(l, pfx + ".1", 1e-5);
        if (k == 1) { A y = alloc(x.B, x.H, x.W, l.N); GemmOpt o; o.act = ACT_SILU; gemm(pfx, x, pack(l), y, o); return y; }
        return conv_gemm(pfx, x, l, k, stride, ACT_SILU);
    }
    // the whole block as one launch (k_mv2.h): hidden widths 64 / 128 (the 160x160 .. 40x40 blocks, where the expanded map is the traffic)
    bool fused_mv2(const std::string& pfx, const A& x, int stride, int oup, const Lin& l1, const Lin& l2, A& y) {
        const int hid = l1.N, Cin = x.C, k1 = cdiv(Cin, KC);
        if (!fuse_mv2 || !mv2_supported(stride, hid, oup) || x.ld % VEC != 0 || l1.K != Cin || l2.K != hid) return false;
        const HostTensor& wd = W(pfx + ".conv.3.weight");
        if (wd.numel() != long(hid) * 9) return false;
        std::vector<float> sc, sh; bn_coeffs(pfx + ".conv.4", 1e-5, sc, sh);
        const int ks2 = hid / KC, nt1 = hid / 16, nt2 = cdiv(oup, 16);
        std::vector<float> w1(size_t(nt1) * k1 * 64 * VEC, 0.f), w2(size_t(nt2) * ks2 * 64 * VEC, 0.f), wdw(size_t(9) * hid), bdw(static_cast<size_t>(hid)), b2(size_t(nt2) * 16, 0.f);
        if (!measuring) {
            for (int n = 0; n < hid; ++n) for (int k = 0; k < Cin; ++k) w1[size_t(mv2_frag_offset(n, k, k1, VEC))] = l1.w[size_t(n) * Cin + k];
            for (int n = 0; n < oup; ++n) { b2[n] = l2.b[n]; for (int k = 0; k < hid; ++k) w2[size_t(mv2_frag_offset(n, k, ks2, VEC))] = l2.w[size_t(n) * hid + k]; }
            for (int c = 0; c < hid; ++c) { bdw[c] = sh[c]; for (int t = 0; t < 9; ++t) wdw[size_t(t) * hid + c] = wd.data[size_t(c) * 9 + t] * sc[c]; }
        }
        const int Ho = (x.H + 2 - 3) / stride + 1, Wo = (x.W + 2 - 3) / stride + 1;
        y = alloc(x.B, Ho, Wo, oup);
        const bool res = stride == 1 && x.C == oup;
        Mv2Params mp{x.p, x.ld, y.p, y.ld, up_T(w1), up_f32(l1.b), up_f32(wdw), up_f32(bdw), up_T(w2), up_f32(b2), res ? x.p : nullptr, res ? x.ld : 0,
                     x.B, x.H, x.W, Ho, Wo, hid, oup, k1};
        const double bytes = (double(x.rows()) * Cin * (res ? 2 : 1) + double(y.rows()) * oup) * sizeof(T);
        const double flops = 2.0 * double(x.rows()) * Cin * hid + 2.0 * double(y.rows()) * hid * (9.0 + oup);
        add_op(pfx + ".block", [mp, stride](hipStream_t s) { launch_mv2<T>(mp, stride, s); }, bytes, flops);
        return true;
    }
    A mv2block(const std::string& pfx, const A& x, int stride, int oup) {         // mobilevit.py:93-131 (expansion != 1)
        Lin l1 = lin(pfx + ".conv.0.weight", ""); fold_bn(l1, pfx + ".conv.1", 1e-5);
        {
            Lin l2f = lin(pfx + ".conv.6.weight", ""); fold_bn(l2f, pfx + ".conv.7", 1e-5);
            if (l2f.N != oup) throw AchError{ACH_ERR_MISSING_KEY, "MV2 width at " + pfx};
            A fy;
            if (fused_mv2(pfx, x, stride, oup, l1, l2f, fy)) return fy;
        }
        A hdn = alloc(x.B, x.H, x.W, l1.N);
        { GemmOpt o; o.act = ACT_SILU; gemm(pfx + ".pw1", x, pack(l1), hdn, o); }
        const int Ho = (x.H + 2 - 3) / stride + 1, Wo = (x.W + 2 - 3) / stride + 1;
        A d = alloc(x.B, Ho, Wo, l1.N);
        dwconv(pfx + ".dw", hdn, nullptr, pfx + ".conv.3.weight", "", pfx + ".conv.4", 1e-5, 3, stride, ACT_SILU, d);
        Lin l2 = lin(pfx + ".conv.6.weight", ""); fold_bn(l2, pfx + ".conv.7", 1e-5);
        if (l2.N != oup) throw AchError{ACH_ERR_MISSING_KEY, "MV2 width at " + pfx};
        A y = alloc(x.B, Ho, Wo, oup);
        GemmOpt o; if (stride == 1 && x.C == oup) o.residual = &x;
        gemm(pfx + ".pw2", d, pack(l2), y, o);
        return y;
    }
    A mvit_block(const std::string& pfx, const A& x, int depth) {                   // mobilevit.py:147-165
        if ((x.H & 1) || (x.W & 1) || (x.H / 2) * (x.W / 2) > MVIT_NMAX) throw AchError{ACH_ERR_UNSUPPORTED, "MobileViT token count"};
        A t = mv_conv(pfx + ".conv1", x, 3, 1);
        t = mv_conv(pfx + ".conv2", t, 1, 1);
        const int D = t.C;
        for (int l = 0; l < depth; ++l) {
            const std::string a = pfx + ".transformer.layers." + std::to_string(l) + ".0", f = pfx + ".transformer.layers." + std::to_string(l) + ".1";
            Lin lq = lin(a + ".fn.to_qkv.weight", ""); fold_ln_in(lq, a + ".norm");
            if (lq.N != 96) throw AchError{ACH_ERR_UNSUPPORTED, "MobileViT attention is built for 4 heads x 8"};
            A qkv = alloc(t.B, t.H, t.W, 96);
            { GemmOpt o; o.ln = true; o.ln_eps = 1e-5f; gemm(a + ".qkv", t, pack(lq), qkv, o); }
            A ao = alloc(t.B, t.H, t.W, 32);
            MvitAttnParams ap{qkv.p, qkv.ld, ao.p, ao.ld, t.B, t.H, t.W, 4, 0.35355339059327373f};
            const int N = (t.H / 2) * (t.W / 2);
            const dim3 grid(unsigned(t.B * 4 * 4), unsigned(cdiv(N, 256))), block(256);
            if (N > MVIT_NMAX) throw AchError{ACH_ERR_UNSUPPORTED, "MobileViT attention: more than 1600 tokens per group"};
            if (N <= 400 && attn_mfma) {                    // scores and P.V on the matrix cores (k_mvit.h), up to 256 queries per workgroup
                add_op(a + ".attn", [ap, grid, block](hipStream_t s) { ACH_LAUNCH((mvit_attn_mfma_kernel<T, 400>), grid, block, s, ap); },
                       double(t.rows()) * 128 * sizeof(T), 4.0 * double(t.rows()) * N * 8);
            } else
            if (N <= 400) add_op(a + ".attn", [ap, grid, block](hipStream_t s) { ACH_LAUNCH((mvit_attn_kernel<T, 400>), grid, block, s, ap); }, double(t.rows()) * 128 * sizeof(T));
            else add_op(a + ".attn", [ap, grid, block](hipStream_t s) { ACH_LAUNCH((mvit_attn_kernel<T, MVIT_NMAX>), grid, block, s, ap); }, double(t.rows()) * 128 * sizeof(T));
            A t1 = alloc(t.B, t.H, t.W, D);
            { GemmOpt o; o.residual = &t; gemm(a + ".to_out", ao, pack(lin(a + ".fn.to_out.0.weight", a + ".fn.to_out.0.bias")), t1, o); }
            Lin l1 = lin(f + ".fn.net.0.weight", f + ".fn.net.0.bias"); fold_ln_in(l1, f + ".norm");
            Lin l2 = lin(f + ".fn.net.3.weight", f + ".fn.net.3.bias");
            A t2;
            if (!fused_mlp_lin(f + ".ffn", f, t1, t1, 0, l1, l2, ACT_SILU, 1e-5f, t2)) {      // LN -> fc1 -> SiLU -> fc2 -> + input (k_mlp.h)
                A hdn = alloc(t.B, t.H, t.W, l1.N);
                { GemmOpt o; o.ln = true; o.ln_eps = 1e-5f; o.act = ACT_SILU; gemm(f + ".ff1", t1, pack(l1), hdn, o); }
                t2 = alloc(t.B, t.H, t.W, D);
                { GemmOpt o; o.residual = &t1; gemm(f + ".ff2", hdn, pack(l2), t2, o); }
            }
            t = t2;
        }
        // conv3 (1x1 D->C) written next to the block input: cat((conv3(x), y), 1) -> conv4 3x3
        Lin l3 = lin(pfx + ".conv3.0.weight", ""); fold_bn(l3, pfx + ".conv3.1", 1e-5);
        if (l3.N != x.C) throw AchError{ACH_ERR_MISSING_KEY, "MobileViT conv3 width at " + pfx};
        A cat = alloc(x.B, x.H, x.W, 2 * x.C);
        { GemmOpt o; o.act = ACT_SILU; gemm(pfx + ".conv3", t, pack(l3), cat.slice(0, x.C), o); }
        copy(pfx + ".cat", x, cat.slice(x.C, x.C));
        return mv_conv(pfx + ".conv4", cat, 3, 1);
    }
    void mobilevit(const std::string& pfx, A feats[4]) {                            // mobilevit.py:198-222
        const MvCfg mc = mv_cfg();
        const int B = batch, R = cfg.resolution;
        A x;
        bool stem_done = false;
        if constexpr (H16E) {
            // conv1 gathered from the NCHW image (k_nhwc.h, mvstem_kernel): no NHWC copy of the image
            const HostTensor& w = W(pfx + ".conv1.0.weight");
            if (mv_stem && R % 2 == 0 && w.shape.size() == 4 && w.shape[0] == 16 && w.shape[1] == 3 && w.shape[2] == 3 && w.shape[3] == 3) {
                Lin l; l.N = 16; l.K = 64; l.w.assign(size_t(16) * 64, 0.f); l.b.assign(16, 0.f);
                for (int n = 0; n < 16; ++n)
                    for (int c = 0; c < 3; ++c)
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) l.w[size_t(n) * 64 + (c * 3 + ky) * 4 + kx] = w.data[((size_t(n) * 3 + c) * 3 + ky) * 3 + kx];
                fold_bn(l, pfx + ".conv1.1", 1e-5);
                Packed pk = pack(l);
                if (pk.NT == 1 && pk.nchunks == 1 && pk.ksteps == 2) {
                    x = alloc(B, R / 2, R / 2, 16);
                    MvStemParams sp{nullptr, x.p, pk.w, up_f32(l.b), B, R, R};
                    const dim3 grid(unsigned(cdivl(x.rows(), 64 * MVSTEM_TPW))), block(256);
                    const void** in = &io.image;
                    const bool alt = io_alt();
                    add_op(pfx + ".conv1", [sp, grid, block, in, alt](hipStream_t s) mutable { sp.X = *in; if (alt) ACH_LAUNCH((mvstem_kernel<T, IOB>), grid, block, s, sp); else ACH_LAUNCH((mvstem_kernel<T, T>), grid, block, s, sp); },
                           (double(B) * 3 * R * R + double(x.rows()) * 16) * sizeof(T), 2.0 * double(x.rows()) * 27 * 16);
                    stem_done = true;
                }
            }
        }
        if (!stem_done) {
        A img = alloc(B, R, R, 3);
        {
            ToNhwcParams tp{nullptr, img.p, B, 3, R, R, img.ld};
            const dim3 grid(unsigned(cdivl(img.rows(), 256))), block(256);
            const void** in = &io.image;
            const bool alt = io_alt();
            add_op(pfx + ".to_nhwc", [tp, grid, block, in, alt](hipStream_t s) mutable { tp.X = *in; if (alt) ACH_LAUNCH((nchw_to_nhwc_kernel<T, IOB>), grid, block, s, tp); else ACH_LAUNCH((nchw_to_nhwc_kernel<T, T>), grid, block, s, tp); },
                   double(img.rows()) * (3 + 3) * sizeof(T), 0, double(img.rows()) * (3 + img.ld) * sizeof(T));
        }
        x = mv_conv(pfx + ".conv1", img, 3, 2);
        }
        x = mv2block(pfx + ".mv2.0", x, 1, mc.ch[1]);
        x = mv2block(pfx + ".mv2.1", x, 2, mc.ch[2]);
        x = mv2block(pfx + ".mv2.2", x, 1, mc.ch[3]);
        x = mv2block(pfx + ".mv2.3", x, 1, mc.ch[3]);
        feats[0] = x;
        if (radar_start_eff() == 0) signal_after_last(0);
        x = mv2block(pfx + ".mv2.4", x, 2, mc.ch[4]);
        x = mvit_block(pfx + ".mvit.0", x, 2);
        feats[1] = x;
        if (radar_start_eff() == 1) signal_after_last(0);
        x = mv2block(pfx + ".mv2.5", x, 2, mc.ch[6]);
        x = mvit_block(pfx + ".mvit.1", x, 4);
        feats[2] = x;
        if (radar_start_eff() >= 2) signal_after_last(0);
        x = mv2block(pfx + ".mv2.6", x, 2, mc.ch[8]);
        x = mvit_block(pfx + ".mvit.2", x, 3);
        feats[3] = mv_conv(pfx + ".conv2", x, 1, 1);
    }

    // ------------------------------------------------------------------------------------------ neck (a7-a13)
    Lin conv_bn(const std::string& conv, const std::string& bn, double eps) const { Lin l = lin(conv + ".weight", conv + ".bias"); fold_bn(l, bn, eps); return l; }

    // GhostModule (ghost_conv.py:6-29) on NHWC: primary 1x1 -> channels [0,init), cheap dw3x3 -> [init, 2*init)
    // NT = 2 fragments of a 1x1 conv for the band kernels of k_ghost.h: [chunk of 32 outputs][k-step][2][64 lanes], bias padded to whole chunks
    struct BandW { const uint4* w = nullptr; const float* b = nullptr; int k1 = 0, chunks = 0; };
    BandW pack_band(const Lin& l) {
        BandW bw; bw.k1 = cdiv(l.K, KC); bw.chunks = cdiv(l.N, 32);
        std::vector<float> blob(size_t(bw.chunks) * bw.k1 * 2 * 64 * VEC, 0.f), bias(size_t(bw.chunks) * 32, 0.f);
        if (!measuring)
            for (int n = 0; n < l.N; ++n) {
                bias[n] = l.b[n];
                for (int k = 0; k < l.K; ++k) blob[size_t(wfrag_offset(n, k, 2, bw.k1, VEC))] = l.w[size_t(n) * l.K + k];
            }
        bw.w = reinterpret_cast<const uint4*>(up_T(blob)); bw.b = up_f32(bias);
        return bw;
    }
    void dw_fold9(const std::string& wkey, const std::string& bnpfx, int C, std::vector<float>& wt, std::vector<float>& bias) const {
        const HostTensor& w = W(wkey);
        if (w.numel() != long(C) * 9) throw AchError{ACH_ERR_MISSING_KEY, "depthwise weight shape: " + wkey};
        std::vector<float> sc, sh; bn_coeffs(bnpfx, 1e-5, sc, sh);
        wt.assign(size_t(9) * C, 0.f); bias.assign(size_t(C), 0.f);
        for (int c = 0; c < C; ++c) { bias[c] = sh[c]; for (int t = 0; t < 9; ++t) wt[size_t(t) * C + c] = w.data[size_t(c) * 9 + t] * sc[c]; }
    }
    A ghost(const std::string& pfx, const A& x, int oup, bool relu) {
        const int init = (oup + 1) / 2;
        if (init % 4 || oup != 2 * init) throw AchError{ACH_ERR_UNSUPPORTED, pfx + ": NHWC GhostModule needs an even output width divisible by 8"};
        A y = alloc(x.B, x.H, x.W, oup);
        if constexpr (H16E) {
            // primary 1x1 + cheap depthwise 3x3 as ONE band kernel (k_ghost.h): the primary output's halo rows are recomputed, nothing is read back
            const int rb = ghost_band_rows(x.H, x.W, init, ghost_rb);
            if (ghost_fuse && !full_taps && rb > 0 && init % 8 == 0 && x.C % 8 == 0 && x.C <= 320 && x.ld % 8 == 0 && y.ld % 8 == 0) {
                Lin lp = conv_bn(pfx + ".primary_conv.0", pfx + ".primary_conv.1", 1e-5);
                if (lp.N != init || lp.K != x.C) throw AchError{ACH_ERR_MISSING_KEY, "GhostModule widths at " + pfx};
                BandW bw = pack_band(lp);
                std::vector<float> wt, bs; dw_fold9(pfx + ".cheap_operation.0.weight", pfx + ".cheap_operation.1", init, wt, bs);
                GhostParams gp{x.p, x.ld, y.p, y.ld, bw.w, bw.b, up_f32(wt), up_f32(bs), x.B, x.H, x.W, x.C, bw.k1, init, bw.chunks,
                               relu ? int(ACT_RELU) : int(ACT_NONE), rb, cdiv(x.H, rb)};
                const dim3 grid(unsigned(gp.bands) * unsigned(x.B)), block(GH_THREADS);
                add_op(pfx + ".ghost", [gp, grid, block](hipStream_t s) { ACH_BAND_LAUNCH(ghost_kernel, gp.k1, grid, block, s, gp); },
                       double(x.rows()) * (x.C + oup) * sizeof(T), 2.0 * double(x.rows()) * x.C * init);
                return y;
            }
        }
        GemmOpt o; o.act = relu ? ACT_RELU : ACT_NONE;
        gemm(pfx + ".primary", x, pack(conv_bn(pfx + ".primary_conv.0", pfx + ".primary_conv.1", 1e-5)), y.slice(0, init), o);
        dwconv(pfx + ".cheap", y.slice(0, init), nullptr, pfx + ".cheap_operation.0.weight", "", pfx + ".cheap_operation.1", 1e-5, 3, 1,
               relu ? ACT_RELU : ACT_NONE, y.slice(init, init));
        return y;
    }
    A ghost_bottleneck(const std::string& pfx, const A& x, int out_chs) {       // ghost_conv.py:58-70, stride 1, in != out
        A g1 = ghost(pfx + ".ghost1", x, x.C, true);
        A g2 = ghost(pfx + ".ghost2", g1, out_chs, false);
        if constexpr (H16E) {
            // shortcut: depthwise 3x3 + BN -> 1x1 + BN, + ghost2's output, as ONE band kernel (k_ghost.h dwpw_kernel)
            const int rb = dwpw_band_rows(x.H, x.W, x.C, ghost_rb);
            if (ghost_fuse && !full_taps && rb > 0 && x.C % 8 == 0 && x.C <= 320 && out_chs % 8 == 0 && x.ld % 8 == 0 && g2.ld % 8 == 0) {   // (ACH_BAND_LAUNCH: k1 <= 10)
                Lin lp = conv_bn(pfx + ".shortcut.2", pfx + ".shortcut.3", 1e-5);
                if (lp.N != out_chs || lp.K != x.C) throw AchError{ACH_ERR_MISSING_KEY, "GhostBottleneck shortcut widths at " + pfx};
                BandW bw = pack_band(lp);
                std::vector<float> wt, bs; dw_fold9(pfx + ".shortcut.0.weight", pfx + ".shortcut.1", x.C, wt, bs);
                A y = alloc(x.B, x.H, x.W, out_chs);
                DwPwParams dp{x.p, x.ld, g2.p, g2.ld, y.p, y.ld, up_f32(wt), up_f32(bs), bw.w, bw.b, x.B, x.H, x.W, x.C, bw.k1, out_chs, bw.chunks, rb, cdiv(x.H, rb)};
                const dim3 grid(unsigned(dp.bands) * unsigned(x.B)), block(GH_THREADS);
                add_op(pfx + ".shortcut", [dp, grid, block](hipStream_t s) { ACH_BAND_LAUNCH(dwpw_kernel, dp.k1, grid, block, s, dp); },
                       double(x.rows()) * (x.C + 2.0 * out_chs) * sizeof(T), 2.0 * double(x.rows()) * x.C * out_chs);
                return y;
            }
        }
        A sd = alloc(x.B, x.H, x.W, x.C);
        dwconv(pfx + ".shortcut.dw", x, nullptr, pfx + ".shortcut.0.weight", "", pfx + ".shortcut.1", 1e-5, 3, 1, ACT_NONE, sd);
        A y = alloc(x.B, x.H, x.W, out_chs);
        GemmOpt o; o.residual = &g2;
        gemm(pfx + ".shortcut.pw", sd, pack(conv_bn(pfx + ".shortcut.2", pfx + ".shortcut.3", 1e-5)), y, o);
        return y;
    }
    // ---- CSP-Dual-FPN blocks (neck/cspdualfpn.py:42-78; BaseConv = conv + BN(1e-3) + act, normal_conv.py:36-47)
    Lin base_conv3(const std::string& pfx, int Ci, int Cp) const {                // 3x3 BaseConv as an implicit-GEMM matrix, BN folded
        Lin l = conv_lin(pfx + ".conv.weight", "", Ci, Cp, 3);
        fold_bn(l, pfx + ".bn", 1e-3);
        return l;
    }
    // Bottleneck: 1x1 (SiLU) -> 3x3 (ReLU, BaseConv's default act) [+ x when in == out].  Writes into `dst` (may be a channel slice, may
    // alias x for the in-place use inside CSPLayer) or, with `user_out`, scatters to the caller's NCHW buffer.
    void csp_bottleneck(const std::string& pfx, const A& x, int cout, const A* dst, void** user_out) {
        Lin l1 = conv_bn(pfx + ".conv1.conv", pfx + ".conv1.bn", 1e-3);
        A t = alloc(x.B, x.H, x.W, l1.N);
        { GemmOpt o; o.act = ACT_SILU; gemm(pfx + ".conv1", x, pack(l1), t, o); }
        Lin l2 = base_conv3(pfx + ".conv2", l1.N, int(t.ld));
        if (l2.N != cout) throw AchError{ACH_ERR_MISSING_KEY, "bottleneck width at " + pfx};
        // 25..32 outputs from <= 32 inputs (the full-resolution decoder levels): row-walking kernel on the dense map (k_conv3.h)
        {
            const int cv = int(t.ld) / VEC, ks = cdiv(9 * cv, 4);
            Packed pk = pack(l2);
            if (row_conv && !user_out && dst && dst->ld == 32 && t.ld % VEC == 0 && (ks == 3 || ks == 5 || ks == 9) && pk.NT == 2 && pk.nchunks == 1 && pk.ksteps == ks &&
                (x.C != cout || x.ld % 8 == 0)) {
                std::vector<float> b32(32, 0.f);
                for (int n = 0; n < l2.N; ++n) b32[n] = l2.b[n];
                const bool res = x.C == cout;
                Conv3Params cp{t.p, t.ld, long(t.W) * t.ld, long(t.H) * t.W * t.ld, dst->p, dst->ld, pk.w, up_f32(b32), t.B, t.H, t.W, cv, ACT_RELU, 1,
                               res ? x.p : nullptr, res ? x.ld : 0, (radar_rows4 == 2 && t.H % 4 == 0) ? 1 : 0};
                const double bytes = double(t.rows()) * (t.ld + dst->ld * (res ? 2 : 1)) * sizeof(T);
                add_op(pfx + ".conv2", [cp, ks](hipStream_t s) { launch_conv3<T>(cp, ks, 2, 1, s); }, bytes, 2.0 * double(t.rows()) * l2.K * l2.N);
                return;
            }
        }
        GemmOpt o; o.act = ACT_RELU; o.residual = (x.C == cout) ? &x : nullptr;
        o.conv_k = 3; o.conv_s = 1; o.conv_p = 1; o.Hin = x.H; o.Win = x.W; o.Cin = int(t.ld); o.Ho = x.H; o.Wo = x.W;
        if (user_out) {
            o.ydyn = user_out; o.out_nchw = 1; o.HW = x.H * x.W; o.Ctot = cout; o.coff = 0;
            gemm(pfx + ".conv2", t.p, t.ld, t.rows(), pack(l2), nullptr, 0, o);
        } else {
            gemm(pfx + ".conv2", t.p, t.ld, t.rows(), pack(l2), dst->p, dst->ld, o);
        }
    }
    // CSPLayer, n = 1: conv1 and conv2 share their input -> one GEMM writes [x1 | x2]; the bottleneck rewrites x1 in place; conv3
    A csp_layer(const std::string& pfx, const A& x, int cout) {
        Lin l1 = conv_bn(pfx + ".conv1.conv", pfx + ".conv1.bn", 1e-3), l2 = conv_bn(pfx + ".conv2.conv", pfx + ".conv2.bn", 1e-3);
        const int hidden = l1.N;
        if (l2.N != hidden || l1.K != x.C || hidden % 8) throw AchError{ACH_ERR_UNSUPPORTED, pfx + ": CSPLayer widths"};
        Lin l12; l12.N = 2 * hidden; l12.K = l1.K; l12.w = l1.w; l12.w.insert(l12.w.end(), l2.w.begin(), l2.w.end());
        l12.b = l1.b; l12.b.insert(l12.b.end(), l2.b.begin(), l2.b.end());
        A cat = alloc(x.B, x.H, x.W, 2 * hidden);
        { GemmOpt o; o.act = ACT_SILU; gemm(pfx + ".conv12", x, pack(l12), cat, o); }
        const A x1 = cat.slice(0, hidden);
        csp_bottleneck(pfx + ".m.0", x1, hidden, &x1, nullptr);
        A y = alloc(x.B, x.H, x.W, cout);
        { GemmOpt o; o.act = ACT_SILU; gemm(pfx + ".conv3", cat, pack(conv_bn(pfx + ".conv3.conv", pfx + ".conv3.bn", 1e-3)), y, o); }
        return y;
    }
    // CSP-Dual-FPN, a decoder level (Upsample + Bottleneck) as ONE row-walking launch (k_csphead.h) behind two low-resolution 1x1 GEMMs: u = relu(BN(conv_up(y))),
    // v = BN(conv1(u)) (conv1's SiLU moves behind the interpolation it commutes with) — with `head_pfx` the level is the last one and the segmentation head (a
    // Bottleneck, num_class outputs, NCHW into the caller's tensor) rides in the same launch; otherwise the level's output y (NHWC, 32 channels) is written to `dst`.
    // false = the widths are not the fused kernel's (32 -> 16 -> 32 channels, head hidden <= 4, <= 16 classes) or the plan wants the level's taps: layer-wise launches.
    bool csp_level_fused(const std::string& up_pfx, const std::string& bn_pfx, const std::string& head_pfx, const A& y, int oup, void** out, const A* dst) {
        if constexpr (!H16E) return false;
        else {
        const bool head = !head_pfx.empty();
        if (!csp_fuse || full_taps || (!head && csp_fuse < 2)) return false;
        Lin lu = conv_bn(up_pfx + ".upsample.0.conv", up_pfx + ".upsample.0.bn", 1e-3);
        Lin l1 = conv_bn(bn_pfx + ".conv1.conv", bn_pfx + ".conv1.bn", 1e-3);
        if (lu.N != 32 || lu.K != y.C || l1.N != 16 || l1.K != 32) return false;
        Lin lh1, lh2;
        int hid = 0;
        if (head) {
            lh1 = conv_bn(head_pfx + ".conv1.conv", head_pfx + ".conv1.bn", 1e-3);
            hid = lh1.N;
            if (lh1.K != 32 || hid < 1 || hid > 4 || oup < 1 || oup > 16) return false;
            lh2 = base_conv3(head_pfx + ".conv2", hid, 4);
            if (lh2.N != oup) return false;
        } else if (!dst || dst->C != 32 || dst->ld % 4) return false;
        Lin l2 = base_conv3(bn_pfx + ".conv2", 16, 16);
        if (l2.N != 32) return false;
        const int H2 = 2 * y.H, W2 = 2 * y.W;
        if (double(y.H) * y.W * 48 * sizeof(T) >= 2147483648.0 || double(H2) * W2 * (head ? oup : int(dst->ld)) * sizeof(T) >= 2147483648.0) return false;
        A uv = alloc(y.B, y.H, y.W, 48);
        const A u = uv.slice(0, 32), v = uv.slice(32, 16);
        if (!chain2_into(up_pfx + ".conv+conv1_lowres", y, lu, ACT_RELU, l1, v, u)) {            // one launch for both (u is the chain's hidden layer), or two GEMMs
            { GemmOpt o; o.act = ACT_RELU; gemm(up_pfx + ".conv", y, pack(lu), u, o); }
            gemm(bn_pfx + ".conv1_lowres", u, pack(l1), v);
        }
        // A fragments (lane l: row i = l & 15, k group kg = l >> 4, elements j < 8), see k_csphead.h
        std::vector<uint16_t> f2(size_t(10) * 64 * 8, 0), f1(size_t(64) * 8, 0), f3(size_t(2) * 64 * 8, 0);
        for (int l = 0; l < 64; ++l) {
            const int i = l & 15, kg = l >> 4;
            for (int j = 0; j < 8; ++j) {
                for (int s = 0; s < 5; ++s)
                    for (int t = 0; t < 2; ++t) {
                        const int tp = 2 * s + (kg >> 1), ch = 8 * (kg & 1) + j;
                        if (tp < 9) f2[((size_t(s) * 2 + t) * 64 + l) * 8 + j] = H16<T>::bits(l2.w[size_t(16 * t + i) * l2.K + size_t(tp) * 16 + ch]);
                    }
                if (!head) continue;
                const int r = i & 3, c = j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4);
                if (r < hid) f1[size_t(l) * 8 + j] = H16<T>::bits(lh1.w[size_t(r) * 32 + c]);
                for (int s = 0; s < 2; ++s) {
                    const int tp = 8 * s + 2 * kg + (j >> 2), ch = j & 3;
                    if (tp < 9 && ch < hid && i < oup) f3[(size_t(s) * 64 + l) * 8 + j] = H16<T>::bits(lh2.w[size_t(i) * lh2.K + size_t(tp) * 4 + ch]);
                }
            }
        }
        std::vector<float> b2(32, 0.f), bh1(4, 0.f), bh2(16, 0.f);
        for (int n = 0; n < 32; ++n) b2[size_t(n)] = l2.b[size_t(n)];
        for (int n = 0; n < hid; ++n) bh1[size_t(n)] = lh1.b[size_t(n)];
        for (int n = 0; n < (head ? oup : 0); ++n) bh2[size_t(n)] = lh2.b[size_t(n)];
        const int band = std::max(8, std::min(csp_band, H2));
        CspHeadParams cp{uv.p, uv.ld, head ? nullptr : static_cast<void*>(dst->p), head ? 0 : dst->ld,
                         static_cast<const uint4*>(up_raw(f2.data(), f2.size() * 2)), up_f32(b2), static_cast<const uint4*>(up_raw(f1.data(), f1.size() * 2)), up_f32(bh1),
                         static_cast<const uint4*>(up_raw(f3.data(), f3.size() * 2)), up_f32(bh2), y.B, y.H, y.W, hid, oup,
                         y.H > 0 ? float(y.H - 1) / float(H2 - 1) : 0.f, y.W > 0 ? float(y.W - 1) / float(W2 - 1) : 0.f, band, cdiv(H2, band), cdiv(W2, head ? CSPH_VALID : CSPL_VALID)};
        if (double(cp.strips) * cp.bands * y.B >= 4294967296.0) return false;
        std::vector<DecHeadRow> rg(static_cast<size_t>(H2) + 4);
        for (int i = 0; i < H2 + 4; ++i) {                       // the float arithmetic of upsample2x_kernel / torch (align_corners)
            const float fy = cp.sy * float(i < H2 ? i : H2 - 1);
            int y0 = int(fy);
            if (y0 > y.H - 1) y0 = y.H - 1;
            rg[size_t(i)] = DecHeadRow{y0, y0 < y.H - 1 ? fy - float(y0) : 0.f};
        }
        const DecHeadRow* rows = static_cast<const DecHeadRow*>(up_raw(rg.data(), rg.size() * sizeof(DecHeadRow)));
        const dim3 grid(unsigned(cp.strips) * unsigned(cp.bands) * unsigned(y.B)), block(64);
        const bool alt = io_alt();
        const double px = double(y.B) * H2 * W2;
        if (head)
            add_op(head_pfx + ".csp_level+head", [cp, grid, block, out, rows, alt](hipStream_t s) mutable {
                cp.out = *out;
                if (alt) { if constexpr (std::is_same<T, f16_t>::value) ACH_LAUNCH((csp_head_rows_kernel<T, bf16_t, true>), grid, block, s, cp, rows); }
                else ACH_LAUNCH((csp_head_rows_kernel<T, T, true>), grid, block, s, cp, rows);
            }, double(uv.rows()) * 48 * sizeof(T) + px * oup * sizeof(T), 2.0 * px * (144.0 * 32 + 32.0 * hid + 36.0 * oup));
        else
            add_op(bn_pfx + ".csp_level", [cp, grid, block, rows](hipStream_t s) { ACH_LAUNCH((csp_head_rows_kernel<T, T, false>), grid, block, s, cp, rows); },
                   double(uv.rows()) * 48 * sizeof(T) + px * 32 * sizeof(T), 2.0 * px * 144.0 * 32);
        return true;
        }
    }
    // Upsample = BaseConv 1x1 + BN(1e-3) + ReLU, bilinear x2 align_corners (ghostdualfpn.py:28-39); writes into `dst`
    void upsample(const std::string& pfx, const A& x, const A& dst) {
        Lin l = conv_bn(pfx + ".upsample.0.conv", pfx + ".upsample.0.bn", 1e-3);
        if constexpr (H16E) {
            // conv + BN + ReLU + bilinear x2 as ONE band kernel (k_ghost.h upconv_kernel): the low-resolution tensor stays in LDS
            const int rb = upconv_band_rows(x.H, x.W, l.N, 2 * ghost_rb - 2);
            if (ghost_fuse && !full_taps && rb > 0 && l.N % 8 == 0 && x.C % 8 == 0 && x.C <= 320 && x.ld % 8 == 0 && dst.ld % 4 == 0 && l.K == x.C) {
                BandW bw = pack_band(l);
                UpConvParams up{x.p, x.ld, dst.p, dst.ld, bw.w, bw.b, x.B, x.H, x.W, x.C, bw.k1, l.N, bw.chunks, rb, cdiv(2 * x.H, rb)};
                const dim3 grid(unsigned(up.bands) * unsigned(x.B)), block(GH_THREADS);
                add_op(pfx + ".conv+bilinear", [up, grid, block](hipStream_t s) { ACH_BAND_LAUNCH(upconv_kernel, up.k1, grid, block, s, up); },
                       double(x.rows()) * (x.C + 4.0 * l.N) * sizeof(T), 2.0 * double(x.rows()) * x.C * l.N);
                return;
            }
        }
        A t = alloc(x.B, x.H, x.W, l.N);
        GemmOpt o; o.act = ACT_RELU;
        gemm(pfx + ".conv", x, pack(l), t, o);
        UpParams p{t.p, t.ld, dst.p, dst.ld, x.B, x.H, x.W, l.N};
        ew(pfx + ".bilinear", upsample2x_kernel<T>, p, long(x.B) * x.H * 2 * x.W * 2 * (l.N / 4), 5.0 * x.rows() * l.N * sizeof(T));
    }
    // the two ShuffleAttention modules that open the decoders, on their common input (shuffle_attention.py:48-72, G = 4)
    void shuffle_attention_pair(const std::string& pfx0, const std::string& pfx1, const A& x, A& y0, A& y1) {
        float* partial = nullptr;
        const int S = stats(pfx0 + ".stats", x, partial);
        float* coef = alloc_f32(size_t(2) * x.B * x.C * 2);
        SaCoefParams pc;
        std::memset(&pc, 0, sizeof(pc));
        pc.partial = partial; pc.S = S; pc.coef = coef; pc.B = x.B; pc.C = x.C; pc.G = 4; pc.HW = x.H * x.W; pc.eps = 1e-5f;
        const std::string* pf[2] = {&pfx0, &pfx1};
        for (int m = 0; m < 2; ++m)
            pc.w[m] = SaWeights{up_f32(W(*pf[m] + ".cweight").data), up_f32(W(*pf[m] + ".cbias").data), up_f32(W(*pf[m] + ".sweight").data),
                                up_f32(W(*pf[m] + ".sbias").data), up_f32(W(*pf[m] + ".gn.weight").data), up_f32(W(*pf[m] + ".gn.bias").data)};
        y0 = alloc(x.B, x.H, x.W, x.C);
        y1 = alloc(x.B, x.H, x.W, x.C);
        SaApplyParams pa{x.p, x.ld, y0.p, y1.p, y0.ld, coef, x.B, x.H * x.W, x.C};
        if (sa_fuse && x.C <= 256 && (long(x.H) * x.W * x.C) % 256 == 0) {       // the coefficients in the apply launch (k_nhwc.h): bit-identical, one launch fewer
            SaFusedParams pf{pc, pa};
            ew(pfx0 + ".coef+apply", sa_apply_fused_kernel<T>, pf, x.rows() * x.C, 3.0 * x.rows() * x.C * sizeof(T));
            return;
        }
        ew(pfx0 + ".coef", sa_coef_kernel, pc, long(2) * x.B * x.C);
        if (x.C % 8 == 0 && x.ld % 4 == 0 && y0.ld % 8 == 0) ew(pfx0 + ".apply", sa_apply8_kernel<T>, pa, x.rows() * (x.C / 8), 3.0 * x.rows() * x.C * sizeof(T));
        else ew(pfx0 + ".apply", sa_apply_kernel<T>, pa, x.rows() * x.C, 3.0 * x.rows() * x.C * sizeof(T));
    }
    // one decoder level: Upsample (1x1+BN+ReLU, bilinear x2) + GhostModule, restructured (see upghost_kernel):
    // both 1x1 convs at low resolution on MFMA, then one fused full-resolution kernel.
    // the level's two 1x1 convs at LOW resolution (they commute with the bilinear interpolation): t = Wp relu(Wu x + bu) + bp
    A decoder_pair(const std::string& up_pfx, const std::string& ghost_pfx, const A& x, int cout) {
        Lin lu = conv_bn(up_pfx + ".upsample.0.conv", up_pfx + ".upsample.0.bn", 1e-3);
        Lin lp = conv_bn(ghost_pfx + ".primary_conv.0", ghost_pfx + ".primary_conv.1", 1e-5);
        const int Cg = lp.N;
        if (2 * Cg != cout || Cg % 4 || Cg > UPG_CMAX || lp.K != lu.N) throw AchError{ACH_ERR_UNSUPPORTED, ghost_pfx + ": decoder level widths"};
        A t;
        if (!chain2(ghost_pfx + ".lowres_pair", x, lu, ACT_RELU, lp, t)) {
            A u = alloc(x.B, x.H, x.W, lu.N);
            { GemmOpt o; o.act = ACT_RELU; gemm(up_pfx + ".conv", x, pack(lu), u, o); }
            t = alloc(x.B, x.H, x.W, Cg);
            gemm(ghost_pfx + ".primary_lowres", u, pack(lp), t);
        }
        return t;
    }
    // can this level's full-resolution kernel also apply the NEXT level's conv pair (k_upchain.h)?  bf16 production plans only: the level's
    // output y is a tap of the parity plans
    bool level_chains(const std::string& ghost_pfx, const std::string& next_up, const std::string& next_ghost) const {
        if (!level_chain || full_taps || !fuse_mlp || !H16E) return false;
        const int Cg = int(W(ghost_pfx + ".primary_conv.0.weight").shape[0]);
        const HostTensor& wu = W(next_up + ".upsample.0.conv.weight");
        const HostTensor& wp = W(next_ghost + ".primary_conv.0.weight");
        return (Cg == 16 || Cg == 24 || Cg == 32) && wu.shape[0] == 32 && wu.shape[1] == 2 * Cg && wp.shape[0] == 16 && wp.shape[1] == 32;
    }
    // full-resolution part of a level on t (low resolution, Cg channels) + the next level's pair: returns the next level's t
    A decoder_level_chained(const std::string& ghost_pfx, const A& t, const std::string& next_up, const std::string& next_ghost) {
        const int Cg = t.C;
        Lin lu = conv_bn(next_up + ".upsample.0.conv", next_up + ".upsample.0.bn", 1e-3);
        Lin lp = conv_bn(next_ghost + ".primary_conv.0", next_ghost + ".primary_conv.1", 1e-5);
        const HostTensor& w = W(ghost_pfx + ".cheap_operation.0.weight");
        std::vector<float> sc, sh; bn_coeffs(ghost_pfx + ".cheap_operation.1", 1e-5, sc, sh);
        std::vector<float> wt(size_t(9) * Cg);
        for (int c = 0; c < Cg; ++c) for (int k = 0; k < 9; ++k) wt[size_t(k) * Cg + c] = w.data[size_t(c) * 9 + k] * sc[c];
        A tn = alloc(t.B, 2 * t.H, 2 * t.W, lp.N);
        MlpParams mp;
        std::memset(&mp, 0, sizeof(mp));
        chain_weights(mp, 2 * Cg, 2, lu, ACT_RELU, lp);
        UpGhostChainParams cp{UpGhostParams{t.p, t.ld, nullptr, 0, up_f32(wt), up_f32(sh), t.B, t.H, t.W, Cg}, tn.p, tn.ld, mp.W1, mp.b1, mp.W2, mp.b2, lp.N};
        const dim3 grid(unsigned(cdiv(2 * t.W, UPG_TS)) * unsigned(cdiv(2 * t.H, UPG_TS)) * unsigned(t.B)), block(unsigned(16 * Cg));
        const double bytes = double(t.rows()) * Cg * sizeof(T) + double(tn.rows()) * lp.N * sizeof(T);
        if (lu.N != 32 || lp.N != 16 || lu.K != 2 * Cg || lp.K != 32) throw AchError{ACH_ERR_INVALID, ghost_pfx + ": chained level expects a 2Cg -> 32 -> 16 pair"};
        if constexpr (!H16E) throw AchError{ACH_ERR_INVALID, ghost_pfx + ": chained decoder levels exist for 16-bit storage only"};
        if constexpr (H16E)
            add_op(ghost_pfx + ".upghost+pair", [cp, grid, block, Cg](hipStream_t s) {
                if (Cg == 16) ACH_LAUNCH((upghost_chain_kernel<T, 16>), grid, block, s, cp);
                else if (Cg == 24) ACH_LAUNCH((upghost_chain_kernel<T, 24>), grid, block, s, cp);
                else ACH_LAUNCH((upghost_chain_kernel<T, 32>), grid, block, s, cp);
            }, bytes, 2.0 * double(tn.rows()) * (2.0 * Cg * 32 + 32.0 * lp.N));
        return tn;
    }
    A decoder_level(const std::string& up_pfx, const std::string& ghost_pfx, const A& x, int cout, const A* t_in = nullptr) {
        const A t = t_in ? *t_in : decoder_pair(up_pfx, ghost_pfx, x, cout);
        const int Cg = t.C;
        const HostTensor& w = W(ghost_pfx + ".cheap_operation.0.weight");
        std::vector<float> sc, sh; bn_coeffs(ghost_pfx + ".cheap_operation.1", 1e-5, sc, sh);
        std::vector<float> wt(size_t(9) * Cg);
        for (int c = 0; c < Cg; ++c) for (int k = 0; k < 9; ++k) wt[size_t(k) * Cg + c] = w.data[size_t(c) * 9 + k] * sc[c];
        A y = alloc(x.B, 2 * x.H, 2 * x.W, cout);
        UpGhostParams p{t.p, t.ld, y.p, y.ld, up_f32(wt), up_f32(sh), x.B, x.H, x.W, Cg};
        const dim3 grid(unsigned(cdiv(2 * x.W, UPG_TS)) * unsigned(cdiv(2 * x.H, UPG_TS)) * unsigned(x.B)), block(unsigned(16 * Cg));
        const double bytes = double(t.rows()) * Cg * sizeof(T) + double(y.rows()) * cout * sizeof(T);
        if constexpr (H16E) if (level_rows && (Cg == 16 || Cg == 24 || Cg == 32) && t.ld % 2 == 0 && y.ld % 2 == 0 &&
            double(y.rows()) * y.ld * sizeof(T) < 2147483648.0) {
            // row-walking form (k_dechead.h; option level_rows, OFF: measured slower than the LDS tile on these write-bound levels — 34 / 60 us against 25 / 48)
            const int H2 = 2 * x.H, band = std::max(8, std::min(head_band, H2));
            UpGhostRowsParams rp{t.p, t.ld, y.p, y.ld, p.Wdw, p.bdw, x.B, x.H, x.W, Cg, x.W > 0 ? float(x.W - 1) / float(2 * x.W - 1) : 0.f,
                                 band, cdiv(H2, band), cdiv(2 * x.W, UGR_VALID)};
            const float sy = x.H > 0 ? float(x.H - 1) / float(2 * x.H - 1) : 0.f;
            std::vector<DecHeadRow> rg(static_cast<size_t>(H2) + 4);
            for (int i = 0; i < H2 + 4; ++i) {
                const float fy = sy * float(i < H2 ? i : H2 - 1);
                int y0 = int(fy);
                if (y0 > x.H - 1) y0 = x.H - 1;
                rg[size_t(i)] = DecHeadRow{y0, y0 < x.H - 1 ? fy - float(y0) : 0.f};
            }
            const DecHeadRow* rows = static_cast<const DecHeadRow*>(up_raw(rg.data(), rg.size() * sizeof(DecHeadRow)));
            const dim3 rgrid(unsigned(rp.strips) * unsigned(rp.bands) * unsigned(x.B)), rblock(64);
            const int np = Cg / 8;
            add_op(ghost_pfx + ".upghost", [rp, rgrid, rblock, rows, np](hipStream_t s) {
                if (np == 2) ACH_LAUNCH((upghost_rows_kernel<T, 2>), rgrid, rblock, s, rp, rows);
                else if (np == 3) ACH_LAUNCH((upghost_rows_kernel<T, 3>), rgrid, rblock, s, rp, rows);
                else ACH_LAUNCH((upghost_rows_kernel<T, 4>), rgrid, rblock, s, rp, rows);
            }, bytes);
            return y;
        }
        if (Cg == 16) add_op(ghost_pfx + ".upghost", [p, grid, block](hipStream_t s) { ACH_LAUNCH((upghost_kernel<T, 16>), grid, block, s, p); }, bytes);
        else if (Cg == 24) add_op(ghost_pfx + ".upghost", [p, grid, block](hipStream_t s) { ACH_LAUNCH((upghost_kernel<T, 24>), grid, block, s, p); }, bytes);
        else if (Cg == 32) add_op(ghost_pfx + ".upghost", [p, grid, block](hipStream_t s) { ACH_LAUNCH((upghost_kernel<T, 32>), grid, block, s, p); }, bytes);
        else throw AchError{ACH_ERR_UNSUPPORTED, ghost_pfx + ": Ghost half-width must be 16, 24 or 32"};
        return y;
    }
    // last decoder level (1_to_0) + segmentation head in one full-resolution kernel (upghost_head_kernel)
    // (`t_in`: the level's low-resolution t, already produced by the previous level's kernel — decoder_level_chained; x is then only a shape)
    void decoder_last_level(const std::string& up_pfx, const std::string& ghost_pfx, const std::string& head_pfx, const std::string& tap_name,
                            const A& x, int cout, int oup, void** out, const A* t_in = nullptr) {
        Lin lu = conv_bn(up_pfx + ".upsample.0.conv", up_pfx + ".upsample.0.bn", 1e-3);
        Lin lp = conv_bn(ghost_pfx + ".primary_conv.0", ghost_pfx + ".primary_conv.1", 1e-5);
        const int Cg = lp.N, init = (oup + 1) / 2, nch = oup - init;
        Lin lh = conv_bn(head_pfx + ".primary_conv.0", head_pfx + ".primary_conv.1", 1e-5);
        if (Cg != UGH_CG || 2 * Cg != cout || lh.K != cout || lh.N != init || init > UGH_IMAX) throw AchError{ACH_ERR_UNSUPPORTED, head_pfx + ": fused last level expects 16+16 channels"};
        A t;
        // bf16: bilinear phase on the matrix cores, t channel-planar (upghost_head_mfma_kernel)
        bool planar = !t_in && head_mfma && std::is_same<T, bf16_t>::value && x.W % 4 == 0 && size_t(x.B) * x.H * x.W * Cg * sizeof(T) < (size_t(1) << 31);
        if (t_in) t = *t_in;
        else if (!chain2(ghost_pfx + ".lowres_pair", x, lu, ACT_RELU, lp, t, &planar)) {
            planar = false;
            A u = alloc(x.B, x.H, x.W, lu.N);
            { GemmOpt o; o.act = ACT_RELU; gemm(up_pfx + ".conv", x, pack(lu), u, o); }
            t = alloc(x.B, x.H, x.W, Cg);
            gemm(ghost_pfx + ".primary_lowres", u, pack(lp), t);
        }
        if (planar) aalloc(256);                  // the MFMA head's 16-byte window loads may start in the last row's last 16 bytes
        auto dw_fold = [&](const std::string& pfx, int n, std::vector<float>& wt, std::vector<float>& bias) {
            const HostTensor& w = W(pfx + ".cheap_operation.0.weight");
            std::vector<float> sc, sh; bn_coeffs(pfx + ".cheap_operation.1", 1e-5, sc, sh);
            wt.assign(size_t(9) * std::max(n, 1), 0.f); bias.assign(static_cast<size_t>(std::max(n, 1)), 0.f);
            for (int c = 0; c < n; ++c) { for (int k = 0; k < 9; ++k) wt[size_t(k) * n + c] = w.data[size_t(c) * 9 + k] * sc[c]; bias[c] = sh[c]; }
        };
        std::vector<float> wl, bl, wh2, bh2;
        dw_fold(ghost_pfx, Cg, wl, bl);
        dw_fold(head_pfx, nch, wh2, bh2);
        A f;
        if (full_taps) { f = alloc(x.B, 2 * x.H, 2 * x.W, cout); tap(tap_name, f); }
        UpGhostHeadParams p{t.p, t.ld, full_taps ? f.p : nullptr, full_taps ? f.ld : 0, nullptr, up_f32(wl), up_f32(bl), up_f32(lh.w), up_f32(lh.b),
                            up_f32(wh2), up_f32(bh2), x.B, x.H, x.W, init, nch, oup,
                            x.H > 0 ? float(x.H - 1) / float(2 * x.H - 1) : 0.f, x.W > 0 ? float(x.W - 1) / float(2 * x.W - 1) : 0.f};
        p.out_bf16 = io_alt() ? 1 : 0;
        const double bytes = double(t.rows()) * Cg * sizeof(T) + 4.0 * double(t.rows()) * oup * sizeof(T);
        // (the row walk addresses one SAMPLE at a time — a 64-bit base per frame, 32-bit byte offsets inside it — so only a frame's own tensors must stay below
        //  2 GiB, whatever the batch: round 4's test was on the whole batch and dropped plans above ~145 frames back to the LDS-tile head)
        if constexpr (H16E) if (head_rows && !planar && t.ld % 4 == 0 && double(x.H) * x.W * t.ld * sizeof(T) < 2147483648.0 &&
                                4.0 * double(x.H) * x.W * oup * sizeof(T) < 2147483648.0 && double(x.B) * cdiv(2 * x.H, 8) * cdiv(2 * x.W, DH_VALID) < 4294967296.0) {
            // row-walking kernel (k_dechead.h): head 1x1 as the A fragment of v_mfma_f32_16x16x32_bf16 — D row 4g + r = head channel g + 4r,
            // k = 8g + j = channel 4g + j of x1 (j < 4) or of x2 (j >= 4) — biases and the head's depthwise filters indexed by head channel
            std::vector<uint16_t> af(size_t(64) * 8, 0);
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, kg = l >> 4, jj = (i / 4) + 4 * (i % 4);
                for (int j = 0; j < 8; ++j) {
                    const int orig = j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4);
                    af[size_t(l) * 8 + j] = (i % 4 < 2 && jj < init) ? H16<T>::bits(lh.w[size_t(jj) * cout + orig]) : uint16_t(0);
                }
            }
            std::vector<float> bh8(8, 0.f), wd8(72, 0.f), bd8(8, 0.f);
            for (int jj = 0; jj < init; ++jj) bh8[jj] = lh.b[jj];
            for (int jj = 0; jj < nch; ++jj) { bd8[jj] = bh2[jj]; for (int k = 0; k < 9; ++k) wd8[size_t(k) * 8 + jj] = wh2[size_t(k) * nch + jj]; }
            const int H2 = 2 * x.H, band = std::max(8, std::min(head_band, H2));
            DecHeadParams dp{t.p, t.ld, full_taps ? f.p : nullptr, full_taps ? f.ld : 0, nullptr, p.Wdw, p.bdw, static_cast<const uint4*>(up_raw(af.data(), af.size() * 2)),
                             up_f32(bh8), up_f32(wd8), up_f32(bd8), x.B, x.H, x.W, init, nch, oup, p.sy, p.sx, band, cdiv(H2, band), cdiv(2 * x.W, DH_VALID)};
            const bool two = head_rows >= 2;              // two columns per lane: strips of 28 valid columns
            if (two) dp.strips = cdiv(2 * x.W, DH2_VALID);
            const int wgw = two ? ACH_DH_WG_WAVES : 1;
            const dim3 grid(unsigned(cdiv(dp.strips * dp.bands * x.B, wgw))), block(unsigned(64 * wgw));
            // per-output-row interpolation geometry, in the float arithmetic of the tile kernel / torch (k_dechead.h)
            std::vector<DecHeadRow> rg(static_cast<size_t>(H2) + 4);
            for (int i = 0; i < H2 + 4; ++i) {
                const float fy = p.sy * float(i < H2 ? i : H2 - 1);
                int y0 = int(fy);
                if (y0 > x.H - 1) y0 = x.H - 1;
                rg[size_t(i)] = DecHeadRow{y0, y0 < x.H - 1 ? fy - float(y0) : 0.f};
            }
            const DecHeadRow* rows = static_cast<const DecHeadRow*>(up_raw(rg.data(), rg.size() * sizeof(DecHeadRow)));
            const int dbg = head_debug;
            const bool dw2 = nch > 4, tapf = full_taps;
            const bool alt = io_alt();
            const unsigned pad = unsigned(head_lds_pad);          // occupancy cap of the row-walking head (option "head_lds_pad", bytes of unused dynamic LDS per single-wave workgroup)
            add_op(head_pfx + ".upghost_head", [dp, grid, block, out, dbg, dw2, tapf, rows, two, alt, pad](hipStream_t s) mutable {
                dp.out = *out;
                auto go = [&](auto io_tag) {             // IO: the type of the caller's output tensor
                    using IO = decltype(io_tag);
                    if (two) {
                        if (tapf) { if (dw2) ACH_LAUNCH((dechead_rows2_kernel<T, IO, true, true>), grid, block, s, dp, rows); else ACH_LAUNCH((dechead_rows2_kernel<T, IO, false, true>), grid, block, s, dp, rows); }
                        else { if (dw2) ACH_LAUNCH_LDS((dechead_rows2_kernel<T, IO, true, false>), grid, block, pad, s, dp, rows); else ACH_LAUNCH_LDS((dechead_rows2_kernel<T, IO, false, false>), grid, block, pad, s, dp, rows); }
                        return;
                    }
                    if (tapf) {                      // parity-test plans: the variant that also writes [x1 | x2]
                        if (dw2) ACH_LAUNCH((dechead_rows_kernel<T, IO, true, true, 0>), grid, block, s, dp, rows); else ACH_LAUNCH((dechead_rows_kernel<T, IO, false, true, 0>), grid, block, s, dp, rows);
                        return;
                    }
#define ACH_DH_CASE(D) case D: if (dw2) ACH_LAUNCH((dechead_rows_kernel<T, IO, true, false, D>), grid, block, s, dp, rows); else ACH_LAUNCH((dechead_rows_kernel<T, IO, false, false, D>), grid, block, s, dp, rows); break;
#if defined(ACH_HEAD_DEBUG)
                    switch (dbg) { ACH_DH_CASE(1) ACH_DH_CASE(2) ACH_DH_CASE(3) ACH_DH_CASE(4) ACH_DH_CASE(7) ACH_DH_CASE(8) ACH_DH_CASE(15) default: ACH_DH_CASE(0) }
#else
                    switch (dbg) { default: ACH_DH_CASE(0) }
#endif
#undef ACH_DH_CASE
                };
                if (alt) go(IOB{}); else go(T{});
            }, bytes, 2.0 * double(t.rows()) * 4.0 * cout * init);
            return;
        }
        const dim3 block(UGH_THREADS);
        if constexpr (std::is_same<T, bf16_t>::value) if (planar) {
            const int tw = 16 * UGM_NSEG - 4;
            const unsigned tiles = unsigned(cdiv(2 * x.W, tw)) * unsigned(cdiv(2 * x.H, UGM_TH)) * unsigned(x.B);
            const unsigned cap = head_grid > 0 ? unsigned(head_grid) : (UGM_GRID > 0 ? unsigned(UGM_GRID) : tiles);
            const dim3 grid(tiles < cap ? tiles : cap);
            const int dbg = head_debug;
            add_op(head_pfx + ".upghost_head", [p, grid, block, out, dbg, tiles](hipStream_t s) mutable {
                p.out = *out;
#define ACH_UGM_CASE(D) case D: if (grid.x < tiles) ACH_LAUNCH((upghost_head_mfma_kernel<UGM_NSEG, UGM_TH, true, D>), grid, block, s, p, p.Wdw, p.bdw, p.Wh, p.bh, p.Wdh, p.bdh); \
                             else ACH_LAUNCH((upghost_head_mfma_kernel<UGM_NSEG, UGM_TH, false, D>), grid, block, s, p, p.Wdw, p.bdw, p.Wh, p.bh, p.Wdh, p.bdh); break;
#if defined(ACH_HEAD_DEBUG)
                switch (dbg) { ACH_UGM_CASE(1) ACH_UGM_CASE(3) ACH_UGM_CASE(7) ACH_UGM_CASE(14) ACH_UGM_CASE(15) default: ACH_UGM_CASE(0) }
#else
                switch (dbg) { default: ACH_UGM_CASE(0) }
#endif
#undef ACH_UGM_CASE
            }, bytes);
            return;
        }
        const dim3 grid(unsigned(cdiv(2 * x.W, UGH_TW)) * unsigned(cdiv(2 * x.H, UGH_TH)) * unsigned(x.B));
        const int dbg = head_debug;
        add_op(head_pfx + ".upghost_head", [p, grid, block, out, dbg](hipStream_t s) mutable {
            p.out = *out;
#define ACH_UGH_CASE(D) case D: ACH_LAUNCH((upghost_head_kernel<T, D>), grid, block, s, p, p.Wdw, p.bdw, p.Wh, p.bh, p.Wdh, p.bdh); break;
#if defined(ACH_HEAD_DEBUG)
            switch (dbg) { ACH_UGH_CASE(1) ACH_UGH_CASE(3) ACH_UGH_CASE(7) ACH_UGH_CASE(14) ACH_UGH_CASE(15) default: ACH_UGH_CASE(0) }
#else
            switch (dbg) { default: ACH_UGH_CASE(0) }
#endif
#undef ACH_UGH_CASE
        }, bytes);
    }

    A cat_buf[2];                     // concat buffers of the top-down path when the backbone writes its features into them
    void neck(A m[4], A q[3]) {                                                  // ghostdualfpn.py:156-200
        const std::string f = "image_radar_encoder.fpn";
        const int* w = widths();
        A m3 = m[1], m4 = m[2], m5 = m[3];
        tap("map2", m[0]); tap("map3", m3); tap("map4", m4); tap("map5", m5);
        // SPP (spp.py:41-67)
        const int c_ = w[3] / 2;
        if (c_ % 4) throw AchError{ACH_ERR_UNSUPPORTED, "SPP hidden width must be a multiple of 4"};
        if (neck_on_side) {              // dec_fork = 3: everything from here on runs on stream 2, behind the backbone's last launch
            if (measuring || ops.empty() || ops.back().signal_ev >= 0) { if (!measuring) throw AchError{ACH_ERR_INVALID, "dec_fork = 3: the backbone's last launch already signals an event"}; }
            else ops.back().signal_ev = 2;
            cur_stream = 2;
            wait_before_next(2);
        }
        mark_xwait_next();               // pipelined forwards: the neck rewrites what the previous forward's stream 2 reads (engine.cpp)
        mark_xwait2_next();              // ... and what its decoders read
        A p5;
        bool spp_done = false;
        if constexpr (H16E) if (ghost_fuse && !full_taps && spp_fused_supported(m5.H, m5.W, w[3], c_) && m5.ld % 8 == 0) {
            // cv1 -> pools -> cv2 as ONE launch (k_ghost.h spp_fused_kernel): the concat buffer never exists
            Lin l1 = conv_bn(f + ".spp.cv1.conv", f + ".spp.cv1.bn", 1e-3), l2 = conv_bn(f + ".spp.cv2.conv", f + ".spp.cv2.bn", 1e-3);
            if (l1.N != c_ || l1.K != w[3] || l2.N != w[3] || l2.K != 4 * c_) throw AchError{ACH_ERR_MISSING_KEY, "SPP widths"};
            BandW b1 = pack_band(l1), b2 = pack_band(l2);
            p5 = alloc(m5.B, m5.H, m5.W, w[3]);
            const int split = spp_split > 0 ? std::min(spp_split, b2.chunks) : (b2.chunks >= 4 ? 2 : 1);       // workgroups per frame: each recomputes cv1 + the pools and takes a share of cv2's output chunks
            SppFusedParams sp{m5.p, m5.ld, p5.p, p5.ld, b1.w, b1.b, b2.w, b2.b, m5.B, m5.H, m5.W, w[3], c_, b1.k1, b1.chunks, b2.k1, b2.chunks, split};
            const dim3 grid(unsigned(m5.B) * unsigned(split)), block(GH_THREADS);
            add_op(f + ".spp", [sp, grid, block](hipStream_t s) { ACH_LAUNCH((spp_fused_kernel<T>), grid, block, s, sp); },
                   2.0 * double(m5.rows()) * w[3] * sizeof(T), 2.0 * double(m5.rows()) * (double(w[3]) * c_ + 4.0 * c_ * w[3]));
            spp_done = true;
        }
        if (!spp_done) {
        A cat5 = alloc(m5.B, m5.H, m5.W, 4 * c_);
        { GemmOpt o; o.act = ACT_SILU; gemm(f + ".spp.cv1", m5, pack(conv_bn(f + ".spp.cv1.conv", f + ".spp.cv1.bn", 1e-3)), cat5.slice(0, c_), o); }
        {
            const int hw = m5.H * m5.W, cq = c_ / 4;
            if (hw > SPP_TILE) throw AchError{ACH_ERR_UNSUPPORTED, "SPP map larger than the pooling tile"};
            const int cqb = std::max(1, std::min(cq, SPP_TILE / hw));
            SppParams sp{cat5.p, cat5.ld, m5.B, m5.H, m5.W, c_, cqb};
            const dim3 grid(unsigned(m5.B) * unsigned(cdiv(cq, cqb))), block(256);
            add_op(f + ".spp.pool", [sp, grid, block](hipStream_t s) { ACH_LAUNCH(spp_pool_kernel<T>, grid, block, s, sp); }, 5.0 * double(m5.rows()) * c_ * sizeof(T));
        }
        p5 = alloc(m5.B, m5.H, m5.W, w[3]);
        { GemmOpt o; o.act = ACT_SILU; gemm(f + ".spp.cv2", cat5, pack(conv_bn(f + ".spp.cv2.conv", f + ".spp.cv2.bn", 1e-3)), p5, o); }
        }
        tap("spp", p5);
        // top-down
        A c4 = cat_buf[1].p ? cat_buf[1] : alloc(m4.B, m4.H, m4.W, 2 * w[2]);
        upsample(f + ".upsample_5_to_4", p5, c4.slice(0, w[2]));
        if (m4.p != c4.slice(w[2], w[2]).p) copy(f + ".cat4", m4, c4.slice(w[2], w[2]));      // else: the backbone wrote it in place
        const bool csp = cfg.neck == ACH_NECK_CDF;                              // cspdualfpn.py:193-237: same graph, CSP blocks
        A p4 = csp ? csp_layer(f + ".ghost_5_to_4", c4, w[2]) : ghost_bottleneck(f + ".ghost_5_to_4", c4, w[2]);
        A c3 = cat_buf[0].p ? cat_buf[0] : alloc(m3.B, m3.H, m3.W, 2 * w[1]);
        upsample(f + ".upsample_4_to_3", p4, c3.slice(0, w[1]));
        if (m3.p != c3.slice(w[1], w[1]).p) copy(f + ".cat3", m3, c3.slice(w[1], w[1]));
        A p3 = csp ? csp_layer(f + ".ghost_4_to_3", c3, w[1]) : ghost_bottleneck(f + ".ghost_4_to_3", c3, w[1]);
        tap("fpn4", p4); tap("fpn3", p3);
        const bool split_dec = split_decoders != 0;
        const bool piped = pipeline && multi_stream && !split_dec;
        // option "dec_fork": where the pipelined plan's decoders leave the caller's stream — 0: behind the shared ShuffleAttention stage (round 3), 1: in front of it
        // (the stage's three launches move to stream 2: the caller's stream is the pipelined loop's bottleneck), 2: as soon as p3 exists (the residual
        // adds stay on the caller's stream and run beside the attention stage)
        const int fork = neck_on_side ? 3 : (piped ? std::min(dec_fork, 2) : 0);
        if (fork == 2) signal_after_last(2);
        // residual FPN outputs (ghostdualfpn.py:200) — computed BEFORE the decoders so that the detection branch (fusion + head,
        // on the radar stream) can start while the two heavy decoders still run on this stream
        q[0] = alloc(p3.B, p3.H, p3.W, p3.C);
        q[1] = alloc(p4.B, p4.H, p4.W, p4.C);
        q[2] = alloc(p5.B, p5.H, p5.W, p5.C);
        {   // the three adds as one launch (they were 6-10 us launch floors on the caller's stream)
            const A* pa[3] = {&p3, &p4, &p5}; const A* pb[3] = {&m3, &m4, &m5};
            AddJobs aj; aj.n = 3;
            long mx = 0; double bytes = 0;
            for (int k = 0; k < 3; ++k) {
                aj.j[k] = AddParams{pa[k]->p, pa[k]->ld, pb[k]->p, pb[k]->ld, q[k].p, q[k].ld, pa[k]->rows(), pa[k]->C};
                mx = std::max(mx, pa[k]->rows() * (pa[k]->C / 4));
                bytes += 3.0 * pa[k]->rows() * pa[k]->C * sizeof(T);
            }
            const dim3 grid(unsigned(cdivl(mx, 256)), 3), block(256);
            add_op(f + ".q3+q4+q5", [aj, grid, block](hipStream_t s) { ACH_LAUNCH(add_multi_kernel<T>, grid, block, s, aj); }, bytes);
        }
        signal_after_last(1);
        if (fork == 3) mark_xsignal3_last();        // the residual adds are the neck's last reader of the backbone's feature maps
        // two segmentation decoders
        const char* names[2] = {"lane", "se"};
        const char* sa[2] = {"stage_3_lane_seg", "stage_3_semantic_seg"};
        const int oups[2] = {2, cfg.num_seg};
        void** outs[2] = {&io.lane, &io.se};
        A ysa[2];
        if (fork == 1) { cur_stream = 2; wait_before_next(1); }                        // (event 1 = the residual adds, the caller's last launch of this forward)
        if (fork == 2) { cur_stream = 2; wait_before_next(2); }
        shuffle_attention_pair(f + "." + sa[0], f + "." + sa[1], p3, ysa[0], ysa[1]);
        if (piped && fork == 0) { signal_after_last(2); cur_stream = 2; wait_before_next(2); }     // both decoders leave the caller's stream
        if (split_dec) signal_after_last(2);   // the semantic decoder may start on its own stream
        for (int d = 0; d < 2; ++d) {
            // past the shared attention stage the two decoders are independent: water-line decoder on the caller's stream, semantic
            // decoder on stream 3 when the option is 1; option 2: the (shorter) water-line decoder joins side stream 1 behind the radar
            // branch instead — no extra stream — and the semantic decoder keeps the caller's
            if (split_decoders == 1 && d == 1) { cur_stream = 3; wait_before_next(2); }
            if (split_decoders == 2) { cur_stream = d == 0 ? 1 : 0; if (d == 0) wait_before_next(2); }
            const std::string n = names[d];
            A y = ysa[d];
            tap(n + ".sa", y);
            const char* lv[3] = {"3_to_2", "2_to_1", "1_to_0"};
            const int cw[3] = {w[1], w[0], w[0]};
            if (csp) {          // Upsample + Bottleneck per level, Bottleneck head: layer-wise on the generic kernels, the full-resolution level + head fused (k_csphead.h)
                for (int l = 0; l < 3; ++l) {
                    if (l == 2 && csp_level_fused(f + "." + n + "_seg_" + lv[l], f + "." + n + "_seg_ghost_" + lv[l], f + "." + n + "_seg_head", y, oups[d], outs[d], nullptr)) { y = A(); break; }
                    if (l < 2 && cw[l] == 32 && csp_fuse >= 2 && !full_taps && H16E) {
                        A v = alloc(y.B, 2 * y.H, 2 * y.W, cw[l]);
                        if (csp_level_fused(f + "." + n + "_seg_" + lv[l], f + "." + n + "_seg_ghost_" + lv[l], "", y, 0, nullptr, &v)) { y = v; continue; }
                    }
                    A u = alloc(y.B, 2 * y.H, 2 * y.W, cw[l]);
                    upsample(f + "." + n + "_seg_" + lv[l], y, u);
                    A v = alloc(u.B, u.H, u.W, cw[l]);
                    csp_bottleneck(f + "." + n + "_seg_ghost_" + lv[l], u, cw[l], &v, nullptr);
                    tap(n + "." + lv[l], v);
                    y = v;
                }
                if (y.p) csp_bottleneck(f + "." + n + "_seg_head", y, oups[d], nullptr, outs[d]);
                continue;
            }
            // level l's full-resolution kernel also applies level l+1's low-resolution conv pair where it can (k_upchain.h): the level's
            // output then never exists in HBM and the pair's launch disappears
            auto up_of = [&](int l) { return f + "." + n + "_seg_" + lv[l]; };
            auto gh_of = [&](int l) { return f + "." + n + "_seg_ghost_" + lv[l]; };
            A t = decoder_pair(up_of(0), gh_of(0), y, cw[0]);
            bool have_t = true;
            for (int l = 0; l < 2; ++l) {
                if (level_chains(gh_of(l), up_of(l + 1), gh_of(l + 1))) { t = decoder_level_chained(gh_of(l), t, up_of(l + 1), gh_of(l + 1)); have_t = true; continue; }
                y = decoder_level(up_of(l), gh_of(l), t, cw[l], &t);
                tap(n + "." + lv[l], y);
                have_t = l < 1;
                if (have_t) t = decoder_pair(up_of(l + 1), gh_of(l + 1), y, cw[l + 1]);
            }
            decoder_last_level(up_of(2), gh_of(2), f + "." + n + "_seg_head", n + "." + lv[2], have_t ? t : y, cw[2], oups[d], outs[d], have_t ? &t : nullptr);
        }
        if (piped) mark_xsignal2_last();    // the decoders' last launch: the next forward's neck may rewrite the attention maps after it
        cur_stream = 0;
    }
    const int* widths() const {
        static const int w0[4] = {32, 48, 96, 176}, w1[4] = {32, 48, 120, 224}, w2[4] = {32, 64, 144, 288};
        return cfg.phi == ACH_PHI_S0 ? w0 : (cfg.phi == ACH_PHI_S1 ? w1 : w2);
    }

    // ------------------------------------------------------------------------------------------ radar (a14-a15)
    // dense k x k conv weight [Co][Ci][k][k] -> implicit-GEMM matrix with K ordered (tap, channel) over Cp channels per tap
    Lin conv_lin(const std::string& wkey, const std::string& bkey, int Ci, int Cp, int k) const {
        const HostTensor& w = W(wkey);
        const int Co = int(w.shape[0]);
        if (w.numel() != long(Co) * Ci * k * k) throw AchError{ACH_ERR_MISSING_KEY, "conv weight shape: " + wkey};
        Lin l; l.N = Co; l.K = k * k * Cp; l.w.assign(size_t(Co) * l.K, 0.f);
        for (int o = 0; o < Co; ++o)
            for (int c = 0; c < Ci; ++c)
                for (int t = 0; t < k * k; ++t) l.w[size_t(o) * l.K + size_t(t) * Cp + c] = w.data[(size_t(o) * Ci + c) * k * k + t];
        if (!bkey.empty() && hasW(bkey)) l.b = W(bkey).data; else l.b.assign(size_t(Co), 0.f);
        return l;
    }
    A conv_gemm(const std::string& name, const A& x, const Lin& l, int k, int stride, int act, const A* residual = nullptr) {
        const int pad = k / 2;
        const int Ho = (x.H + 2 * pad - k) / stride + 1, Wo = (x.W + 2 * pad - k) / stride + 1;
        A y = alloc(x.B, Ho, Wo, l.N);
        GemmOpt o; o.act = act; o.residual = residual;
        o.conv_k = k; o.conv_s = stride; o.conv_p = pad; o.Hin = x.H; o.Win = x.W; o.Cin = int(x.ld); o.Creal = x.C; o.Ho = Ho; o.Wo = Wo;
        gemm(name, x.p, x.ld, y.rows(), pack(l), y.p, y.ld, o);
        return y;
    }
    // NHWC tensor with a one-pixel zero border around every sample (the arena is zeroed when the plan is built and the border is
    // never written): p0 = pixel (0,0) of sample 0
    struct Bordered { T* p0 = nullptr; T* base = nullptr; int B = 0, H = 0, W = 0, C = 0; long ld = 0, row = 0, img = 0; };
    Bordered alloc_bordered(int B, int H, int W, int C, long ld = 0) {
        Bordered t; t.B = B; t.H = H; t.W = W; t.C = C; t.ld = ld ? ld : round_up(C, 8);
        t.row = long(W + 2) * t.ld; t.img = long(H + 2) * t.row;
        t.base = static_cast<T*>(aalloc(size_t(B) * t.img * sizeof(T)));
        t.p0 = t.base + t.row + t.ld;
        return t;
    }
    // 3x3 / pad 1 conv (stride 1 or 2) of a bordered tensor with up to 32 outputs: the offset + modulator convs and the
    // stride-2 weight_conv2 of the RCBlocks (k_conv3.h); anything else goes to the generic implicit GEMM
    A conv3_bordered(const std::string& name, const Bordered& x, const Lin& l, int act, int stride = 1) {
        const int Ho = (x.H - 1) / stride + 1, Wo = (x.W - 1) / stride + 1;
        A y = alloc(x.B, Ho, Wo, l.N);
        const bool half = 2 * x.ld == VEC;                       // 8-byte pixels: one k-slot per tap, its upper half zero (k_conv3.h)
        const int cv = half ? 1 : int(x.ld) / VEC, ks = cdiv(9 * cv, 4);
        Packed pk = pack(l);
        const double bytes = double(x.B) * x.H * x.W * x.C * sizeof(T) + double(y.rows()) * y.C * sizeof(T);
        const double lbytes = double(x.B) * x.H * x.W * x.ld * sizeof(T) + double(y.rows()) * y.ld * sizeof(T);
        const bool shape_ok = (pk.NT == 2 && y.ld == 32 && !half) || (pk.NT == 1 && stride == 2 && y.ld <= 16);
        if (row_conv && shape_ok && (ks == 3 || ks == 5 || ks == 9) && pk.nchunks == 1 && pk.ksteps == ks) {
            std::vector<float> b32(32, 0.f);
            for (int n = 0; n < l.N; ++n) b32[n] = l.b[n];
            Conv3Params cp{x.p0, x.ld, x.row, x.img, y.p, y.ld, pk.w, up_f32(b32), x.B, Ho, Wo, cv, act, 0, nullptr, 0, (radar_rows4 == 2 && Ho % 4 == 0) ? 1 : 0};
            const int NT = pk.NT;
            add_op(name, [cp, ks, NT, stride, half](hipStream_t s) { launch_conv3<T>(cp, ks, NT, stride, s, half); }, bytes, 2.0 * double(y.rows()) * l.K * l.N, lbytes);
            return y;
        }
        if (half) throw AchError{ACH_ERR_INVALID, name + ": 8-byte pixels are only read by the row-walking conv"};
        // generic implicit GEMM: the bordered buffer is a dense [B, H+2, W+2, ld] tensor convolved without padding
        GemmOpt o; o.act = act;
        o.conv_k = 3; o.conv_s = stride; o.conv_p = 0; o.Hin = x.H + 2; o.Win = x.W + 2; o.Cin = int(x.ld); o.Creal = x.C; o.Ho = Ho; o.Wo = Wo;
        gemm(name, x.base, x.ld, y.rows(), pk, y.p, y.ld, o);
        return y;
    }
    void rcnet(A outs[3]) {                                                      // RadarEncoder.py:38-109
        const int* w = widths();
        const int chans[9] = {3, w[0] / 4, w[0] / 4, w[0] / 4, w[1] / 4, w[1] / 4, w[2] / 4, w[2] / 4, w[3] / 4};
        const bool down[8] = {true, true, false, true, false, true, false, true};
        const int B = batch, R = cfg.resolution;
        // The 3-channel maps of the first RCBlock (network input, pooled map, block output: 6.5 M pixels each at batch 64) are carried
        // as 4-channel pixels — 8 B in bf16, 16 B in fp32 — instead of the 8-channel padding of the generic NHWC layout: they were 2.7x
        // their real bytes in HBM traffic.  Only the fused front + row-walking conv read that layout; the layer-wise fallback keeps 8.
        const bool narrow0 = fuse_rc && row_conv && chans[0] <= 4 && (R * R) % 4 == 0;
        // round 4 (16-bit engines): no NHWC copy of the map at all — the first pool and the first front kernel's residual read the caller's NCHW planes
        // (option "radar_direct"; needs the strip pool and rows of whole 16-pixel segments)
        bool direct0 = false;
        if constexpr (H16E) direct0 = radar_direct && narrow0 && pool_strip > 0 && R % 16 == 0 && R % POOLN_ROWS == 0;
        A x;
        if (direct0) { x.B = B; x.H = R; x.W = R; x.C = 3; x.ld = 4; x.p = nullptr; }     // shape only
        else x = narrow0 ? alloc_ld(B, R, R, 3, 4) : alloc(B, R, R, 3);
        if (!direct0) {
            ToNhwcParams tp{nullptr, x.p, B, 3, R, R, x.ld};
            const void** rin = &io.radar;
            mark_xwait_next();           // pipelined forwards: this branch rewrites the radar pyramid the previous forward's fusion reads
            const double bytes = double(x.rows()) * (3 + 3) * sizeof(T), lbytes = double(x.rows()) * (3 + x.ld) * sizeof(T);
            const bool alt = io_alt();
            if (narrow0) {
                const dim3 grid(unsigned(cdivl(x.rows() / 4, 256))), block(256);
                add_op("image_radar_encoder.radar_encoder.to_nhwc", [tp, grid, block, rin, alt](hipStream_t s) mutable { tp.X = *rin; if (alt) ACH_LAUNCH((nchw3_to_nhwc4_kernel<T, IOB>), grid, block, s, tp); else ACH_LAUNCH((nchw3_to_nhwc4_kernel<T, T>), grid, block, s, tp); }, bytes, 0, lbytes);
            } else {
                const dim3 grid(unsigned(cdivl(x.rows(), 256))), block(256);
                add_op("image_radar_encoder.radar_encoder.to_nhwc", [tp, grid, block, rin, alt](hipStream_t s) mutable { tp.X = *rin; if (alt) ACH_LAUNCH((nchw_to_nhwc_kernel<T, IOB>), grid, block, s, tp); else ACH_LAUNCH((nchw_to_nhwc_kernel<T, T>), grid, block, s, tp); }, bytes, 0, lbytes);
            }
        }
        for (int i = 0; i < 8; ++i) {
            const std::string pfx = "image_radar_encoder.radar_encoder.rc_blocks." + std::to_string(i);
            const std::string d = pfx + ".radar_conv.deformable_conv";
            const bool narrow = i == 0 && narrow0;
            const bool half = narrow && 2 * x.ld == VEC;             // 8-byte pixels (bf16): a k-slot is one tap x [4 channels | 4 zeros]
            const int C = chans[i], Cp = half ? VEC : int(x.ld);     // k-elements per tap in the packed conv matrices
            // AvgPool2d(3,1,1)
            Bordered pooled = alloc_bordered(B, x.H, x.W, C, narrow ? 4 : 0);
            // block 0 reads the raw radar map (> 99 % zeros): occupancy masks (one bit per column) per 16-pixel row segment of the pooled map let rc_front
            // take its closed-form shortcut on empty segments (k_conv3.h); the reach follows from the offset conv's bias
            unsigned short* occ = nullptr;
            int occ_r = 0;
            const bool strip0 = pool_strip > 0 && (C >= 16 || narrow || (pool_strip > 1 && C >= 8));
            if (i == 0 && radar_skip && fuse_rc && strip0 && C <= 4 && x.W % 16 == 0 && x.W / 16 <= 32) {
                const HostTensor& ob = W(d + ".offset_conv.bias");
                float mx = 0.f;
                bool finite = true;
                for (float v : ob.data) { finite = finite && std::isfinite(v); mx = std::max(mx, std::fabs(v)); }
                // reach of a pixel whose 3x3 window of P is zero (offsets = the bias b exactly): tap row ty samples at oy + (ty - 1) + b, |b| <= mx, so the bilinear corners lie in
                // [oy - 1 - ceil(mx), oy + 2 + floor(mx)]: 2 + floor(mx) rows / columns (round 6; it was 2 + ceil(mx), one more than needed for every non-integer mx — 7 x 7 instead of
                // 5 x 5 pixels of full path around each occupied cell of the bench's maps).  At an integer mx the outermost corner has weight exactly 0 on a finite value.
                if (finite && mx <= 13.f) { occ_r = 2 + int(std::floor(mx)); occ = static_cast<unsigned short*>(aalloc(size_t(B) * x.H * (x.W / 16) * sizeof(unsigned short))); }
            }
            if (i == 0 && direct0) {
                if constexpr (H16E) {
                PoolNchwParams pp{nullptr, pooled.p0, pooled.ld, B, x.H, x.W, pooled.row, pooled.img, occ};
                pp.sparse = (radar_pool_sparse && occ) ? 1 : 0;
                const dim3 grid(unsigned(cdivl(long(B) * (x.H / POOLN_ROWS) * (x.W / 4), 256))), block(256);
                const void** rin = &io.radar;
                const bool alt = io_alt();
                mark_xwait_next();           // pipelined forwards: this branch rewrites the radar pyramid the previous forward's fusion reads
                add_op(pfx + ".avgpool", [pp, grid, block, rin, alt](hipStream_t s) mutable { pp.X = *rin; if (alt) ACH_LAUNCH((avgpool3x3_nchw3_kernel<T, IOB>), grid, block, s, pp); else ACH_LAUNCH((avgpool3x3_nchw3_kernel<T, T>), grid, block, s, pp); },
                       2.0 * x.rows() * C * sizeof(T), 0, double(x.rows()) * (3 + pooled.ld) * sizeof(T));
                }
            } else
            { PoolParams pp{x.p, x.ld, pooled.p0, pooled.ld, B, x.H, x.W, C, pooled.row, pooled.img, occ};
              const double bytes = 2.0 * x.rows() * C * sizeof(T), lbytes = double(x.rows()) * (x.ld + pooled.ld) * sizeof(T);
              // strips of 4 output pixels per thread where a strip is contiguous enough for the loads to coalesce: >= 16 channels, or
              // the 4-channel pixels of block 0 (option "pool_strip": 0 never, 1 as described, 2 also the 8-channel blocks)
              const bool strip = pool_strip > 0 && (C >= 16 || narrow || (pool_strip > 1 && C >= 8));
              if (strip) ew(pfx + ".avgpool", avgpool3x3_kernel<T, 4>, pp, long(B) * x.H * cdiv(x.W, 4) * ((C + 3) / 4), bytes, lbytes);
              else ew(pfx + ".avgpool", avgpool3x3_kernel<T, 1>, pp, x.rows() * ((C + 3) / 4), bytes, lbytes); }
            // offset_conv (18) + modulator_conv (9) as one implicit GEMM
            Lin lo = conv_lin(d + ".offset_conv.weight", d + ".offset_conv.bias", C, Cp, 3);
            Lin lm = conv_lin(d + ".modulator_conv.weight", d + ".modulator_conv.bias", C, Cp, 3);
            if (lo.N != 18 || lm.N != 9) throw AchError{ACH_ERR_MISSING_KEY, "deformable conv shapes at " + d};
            Lin lom; lom.N = 27; lom.K = lo.K; lom.w = lo.w; lom.w.insert(lom.w.end(), lm.w.begin(), lm.w.end()); lom.b = lo.b; lom.b.insert(lom.b.end(), lm.b.begin(), lm.b.end());
            const int cvp = half ? 1 : int(pooled.ld) / VEC, ksp = cdiv(9 * cvp, 4);
            const bool fused_front = fuse_rc && C <= 16 && (ksp == 3 || ksp == 5 || ksp == 9);
            A om;
            if (!fused_front) om = conv3_bordered(pfx + ".offmask", pooled, lom, ACT_NONE);
            // regular_conv (no bias) folded with weight_conv1 (bias) and BatchNorm:  Wf[co][k][c] = sum_m W1'[co][m] Wd3[m][c][k]
            std::vector<float> sc, sh; bn_coeffs(pfx + ".norm", 1e-5, sc, sh);
            const HostTensor& w1 = W(pfx + ".weight_conv1.weight"); const HostTensor& b1 = W(pfx + ".weight_conv1.bias");
            const HostTensor& w3 = W(d + ".regular_conv.weight");
            if (w3.numel() != long(C) * C * 9 || w1.numel() != long(C) * C) throw AchError{ACH_ERR_MISSING_KEY, "radar block shapes at " + pfx};
            Lin lf; lf.N = C; lf.K = 9 * Cp; lf.w.assign(size_t(C) * lf.K, 0.f); lf.b.assign(size_t(C), 0.f);
            for (int co = 0; co < C; ++co) {
                lf.b[co] = b1.data[co] * sc[co] + sh[co];
                for (int c = 0; c < C; ++c)
                    for (int k = 0; k < 9; ++k) {
                        double acc = 0;
                        for (int m = 0; m < C; ++m) acc += double(w1.data[size_t(co) * C + m]) * sc[co] * w3.data[(size_t(m) * C + c) * 9 + k];
                        lf.w[size_t(co) * lf.K + size_t(k) * Cp + c] = float(acc);
                    }
            }
            A y;
            Bordered yb;
            bool y_bordered = false;
            if (fused_front) {           // conv + sampling + folded contraction + ReLU + residual as one launch (k_conv3.h)
                Lin lf2 = lf;            // the kernel multiplies the samples by sigmoid(logit); the factor 2 of the modulation (dcn.py:59) is exact here
                for (float& v : lf2.w) v *= 2.0f;
                Packed pkom = pack(lom), pkf = pack(lf2);
                if (pkom.NT != 2 || pkom.nchunks != 1 || pkom.ksteps != ksp || pkf.NT != 1 || pkf.nchunks != 1 || pkf.ksteps != ksp)
                    throw AchError{ACH_ERR_UNSUPPORTED, "radar block packing at " + pfx};
                std::vector<float> b32(32, 0.f), b16(16, 0.f);
                for (int n = 0; n < 27; ++n) b32[n] = lom.b[n];
                for (int n = 0; n < C; ++n) b16[n] = lf.b[n];
                if (down[i] && row_conv) {            // the stride-2 weight_conv2 reads it through the row-walking kernel: zero border
                    yb = alloc_bordered(B, x.H, x.W, C, narrow ? 4 : 0);
                    y_bordered = true;
                } else {
                    y = alloc(B, x.H, x.W, C);
                }
                const long yld = y_bordered ? yb.ld : y.ld;
                RcFrontParams rp{pooled.p0, pooled.ld, pooled.row, pooled.img, pkom.w, up_f32(b32), pkf.w, up_f32(b16), x.p, x.ld,
                                 y_bordered ? yb.p0 : y.p, yld, y_bordered ? yb.row : long(x.W) * y.ld, y_bordered ? yb.img : long(x.H) * x.W * y.ld,
                                 B, x.H, x.W, cvp, C, occ, occ_r, radar_compact ? 1 : 0, (((occ && radar_rows4 == 1) || radar_rows4 == 2) && x.H % 4 == 0) ? 1 : 0, nullptr, 0};
                const bool rdirect = i == 0 && direct0;
                if constexpr (H16E) {
                    // background mode (k_conv3.h, round 6): the conditions are the kernel's own for its compact path, plus the NCHW pool (its occupancy masks include non-zero raw inputs)
                    if (radar_bg && rdirect && occ && y_bordered && narrow && rp.compact && rp.rows4 && yld <= 4 && x.W <= 512) {
                        rp.prev = static_cast<unsigned short*>(aalloc(size_t(B) * x.H * (x.W / 16) * sizeof(unsigned short)));
                        rp.bg = 1;
                        uint32_t h[4];
                        for (int c = 0; c < 4; ++c) { const float r = c < C ? lf.b[c] : 0.f; h[c] = H16<T>::bits(r > 0.f ? r : 0.f); }
                        FillPx8Params fq{yb.p0, yb.row, yb.img, B, x.H, x.W, make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16))};
                        if (!measuring) post_plan.push_back([fq]() { ACH_LAUNCH(fill_px8_kernel<T>, dim3(unsigned(cdivl(long(fq.B) * fq.H * fq.Wd, 256))), dim3(256), hipStream_t(nullptr), fq); });
                    }
                }
                const void** rres = &io.radar;
                const int rbf = io_alt() ? 1 : 0;
                // algorithmic bytes: pooled map + residual read once, output written once, REAL channels; the layout figure
                // counts the pixel pitches the kernel actually moves
                const double bytes = double(x.rows()) * 3.0 * C * sizeof(T), lbytes = double(x.rows()) * (pooled.ld + x.ld + yld) * sizeof(T);
                add_op(pfx + ".front", [rp, ksp, rdirect, rres, rbf](hipStream_t s) mutable { if (rdirect) { rp.Rn = *rres; rp.rn_bf16 = rbf; } launch_rc_front<T>(rp, ksp, s); }, bytes,
                       2.0 * double(x.rows()) * 9.0 * C * (27 + C), lbytes);
            } else {
            DeformParams dp;
            std::memset(&dp, 0, sizeof(dp));
            dp.pooled = pooled.p0; dp.ldp = pooled.ld; dp.prow = pooled.row; dp.pimg = pooled.img; dp.om = om.p; dp.ldo = om.ld; dp.res = x.p; dp.ldr = x.ld;
            dp.B = B; dp.H = x.H; dp.Wd = x.W; dp.Cp = Cp;
            if (Cp == 8 && (C == 3 || C == 8)) {
                y = alloc(B, x.H, x.W, C);
                dp.Y = y.p; dp.ldy = y.ld; dp.Wf = up_f32(lf.w); dp.bf = up_f32(lf.b);
                const dim3 grid(unsigned(cdivl(x.rows(), 256))), block(256);
                const double bytes = double(x.rows()) * (3.0 * C + 27) * sizeof(T), lbytes = double(x.rows()) * (3.0 * Cp + 32) * sizeof(T);
                if (C == 3) add_op(pfx + ".deform", [dp, grid, block](hipStream_t s) { ACH_LAUNCH((deform_fused_kernel<T, 3, 4>), grid, block, s, dp, dp.Wf, dp.bf); }, bytes, 0, lbytes);   // 3 channels: one 4-vector per corner
                else add_op(pfx + ".deform", [dp, grid, block](hipStream_t s) { ACH_LAUNCH((deform_fused_kernel<T, 8, 8>), grid, block, s, dp, dp.Wf, dp.bf); }, bytes, 0, lbytes);
            } else {
                A col = alloc(B, x.H, x.W, 9 * Cp);
                dp.Y = col.p; dp.ldy = col.ld;
                ew(pfx + ".deform.sample", deform_sample_kernel<T>, dp, x.rows() * 9 * (Cp / 4), double(x.rows()) * (10.0 * C + 27) * sizeof(T), double(x.rows()) * (10.0 * Cp + 32) * sizeof(T));
                y = alloc(B, x.H, x.W, C);
                GemmOpt o; o.act = ACT_RELU; o.residual = &x;            // epilogue order: act, then + residual
                gemm(pfx + ".deform.contract", col, pack(lf), y, o);
            }
            }
            // weight_conv2: 1x1, or 3x3 stride 2
            const int k = down[i] ? 3 : 1;
            if (y_bordered) x = conv3_bordered(pfx + ".conv2", yb, conv_lin(pfx + ".weight_conv2.weight", pfx + ".weight_conv2.bias", C, half ? VEC : int(yb.ld), 3), ACT_NONE, 2);
            else x = conv_gemm(pfx + ".conv2", y, conv_lin(pfx + ".weight_conv2.weight", pfx + ".weight_conv2.bias", C, int(y.ld), k), k, down[i] ? 2 : 1, ACT_NONE);
            if (x.C != chans[i + 1]) throw AchError{ACH_ERR_MISSING_KEY, "radar width mismatch at " + pfx};
            tap("radar.b" + std::to_string(i), x);
            if (i == 3) outs[0] = x;
            if (i == 5) outs[1] = x;
            if (i == 7) outs[2] = x;
        }
        tap("r3", outs[0]); tap("r4", outs[1]); tap("r5", outs[2]);
    }

    // ------------------------------------------------------------------------------------------ fusion (a16)
    // ECA channel attention + BatchNorm + ReLU on cat(image level, radar level) for the three pyramid levels (IREncoder.py:79-89).
    // The six (level, source) jobs are independent: three launches in total — statistics, ECA scales, scaled write — each
    // covering all six (Multi6 in k_nhwc.h), instead of eighteen small ones.
    void fuse_all(const A img[3], const A rad[3], A out[3]) {
        const std::string e = "image_radar_encoder";
        Multi6<StatParams> ms; Multi6<EcaParams> me; Multi6<FuseParams> mf;
        std::memset(&ms, 0, sizeof(ms)); std::memset(&me, 0, sizeof(me)); std::memset(&mf, 0, sizeof(mf));
        int n = 0, smax = 1;
        long eca_max = 0, fuse_max = 0;
        double bytes = 0;
        for (int l = 0; l < 3; ++l) {
            const std::string st = std::to_string(3 + l);
            const int Ci = img[l].C, Cr = rad[l].C, HW = img[l].H * img[l].W;
            std::vector<float> sc, sh; bn_coeffs(e + ".norm_stage" + st, 1e-5, sc, sh);
            if (int(sc.size()) != Ci + Cr) throw AchError{ACH_ERR_MISSING_KEY, "fusion norm width"};
            out[l] = alloc(img[l].B, img[l].H, img[l].W, Ci + Cr);
            const A srcs[2] = {img[l], rad[l]};
            int coff = 0;
            for (int h = 0; h < 2; ++h, ++n) {
                const A& x = srcs[h];
                const int S = HW >= 1024 ? 8 : (HW >= 256 ? 4 : 1);
                float* part = alloc_f32(size_t(x.B) * S * 2 * x.C);
                ms.j[n] = StatParams{x.p, x.ld, part, HW, x.C, S};
                smax = std::max(smax, S);
                const HostTensor& wk = W(e + ".channel_attn_stage" + st + "." + std::to_string(h) + ".conv.weight");
                float* scl = alloc_f32(size_t(x.B) * x.C);
                me.j[n] = EcaParams{part, S, up_f32(wk.data), int(wk.numel()), up_f32(std::vector<float>(sc.begin() + coff, sc.begin() + coff + x.C)), scl, x.B, x.C, HW};
                eca_max = std::max(eca_max, long(x.B) * x.C);
                A ys = out[l].slice(coff, x.C);
                mf.j[n] = FuseParams{x.p, x.ld, 0, ys.p, ys.ld, scl, up_f32(std::vector<float>(sh.begin() + coff, sh.begin() + coff + x.C)), x.B, HW, x.C};
                fuse_max = std::max(fuse_max, cdivl(x.rows() * x.C, 4));        // 4 channels per thread
                bytes += 2.0 * x.rows() * x.C * sizeof(T);
                coff += x.C;
            }
            tap("p" + st, out[l]);
        }
        ms.n = me.n = mf.n = n;
        {
            const dim3 grid(unsigned(img[0].B), unsigned(smax), unsigned(n)), block(256);
            add_op(e + ".fusion.stats", [ms, grid, block](hipStream_t s) { ACH_LAUNCH(chan_stats_multi_kernel<T>, grid, block, s, ms); }, bytes / 2);
        }
        {
            const dim3 grid(unsigned(cdivl(eca_max, 256)), unsigned(n)), block(256);
            add_op(e + ".fusion.eca", [me, grid, block](hipStream_t s) { ACH_LAUNCH(eca_scale_multi_kernel, grid, block, s, me); });
        }
        {
            const dim3 grid(unsigned(cdivl(fuse_max, 256)), unsigned(n)), block(256);
            add_op(e + ".fusion.apply", [mf, grid, block](hipStream_t s) { ACH_LAUNCH(fuse_scale_multi_kernel<T>, grid, block, s, mf); }, bytes);
            mark_xsignal_last();         // last reader of the FPN outputs / radar pyramid (and, in stream order, after the decoders' reads)
        }
    }

    // ------------------------------------------------------------------------------------------ head (a17)
    // The cls and reg branches (2 x [dw5x5 -> pw1x1+BN+ReLU]) read the same stem output and never interact, so they are run as
    // ONE 128-channel branch: depthwise filters of both banks side by side (the first over the shared 64-channel input), the
    // pointwise convs as block-diagonal 128x128 GEMMs, and the three prediction convs as one 12 x 128 GEMM scattered straight
    // into the NCHW output map [reg 4 | obj 1 | cls num_det].  6 launches per level instead of 11, twice the work per launch on
    // the small maps where the head is latency-bound.
    void head(A p[3]) {                                                          // decouplehead.py:58-103
        const int NC5 = 5 + cfg.num_det;
        batching = head_batch;          // the same six layers on three maps: one launch per layer for all levels (flush_batch)
        batch_jobs.clear();
        // the fused depthwise + pointwise layer (k_headdw.h) is chosen for ALL three levels or for none: flush_batch needs the same number of
        // jobs on every level (a level-0 map wider than 66 columns — resolution >= 544 — has no band that fits the kernel's LDS tile)
        bool fuse_layers = head_fuse && batching && H16E;
        for (int k = 0; k < 3 && fuse_layers; ++k)
            fuse_layers = int(W("det_head.stems." + std::to_string(k) + ".conv.weight").shape[0]) == HDW_C && headdw_band_rows(p[k].H, p[k].W) > 0;
        for (int k = 0; k < 3; ++k) {
            const std::string ks = std::to_string(k);
            const A& x = p[k];
            const int HW = x.H * x.W;
            Lin ls = conv_bn("det_head.stems." + ks + ".conv", "det_head.stems." + ks + ".bn", 1e-3);
            const int base = ls.N;
            A cur = alloc(x.B, x.H, x.W, base);
            { GemmOpt o; o.act = ACT_RELU; gemm("det_head.stems." + ks, x, pack(ls), cur, o); }
            for (int j = 0; j < 2; ++j) {
                const std::string js = std::to_string(j);
                const std::string c = "det_head.cls_convs." + ks + "." + js, r = "det_head.reg_convs." + ks + "." + js;
                // depthwise 5x5, both banks: out[0:base] = cls filters, out[base:2base] = reg filters
                const HostTensor& wc = W(c + ".conv.dconv.weight"); const HostTensor& wr = W(r + ".conv.dconv.weight");
                if (wc.numel() != long(base) * 25 || wr.numel() != long(base) * 25) throw AchError{ACH_ERR_MISSING_KEY, "head depthwise shape"};
                std::vector<float> wt(size_t(25) * 2 * base), bias(size_t(2) * base, 0.f);
                for (int ch = 0; ch < base; ++ch)
                    for (int t = 0; t < 25; ++t) { wt[size_t(t) * 2 * base + ch] = wc.data[size_t(ch) * 25 + t]; wt[size_t(t) * 2 * base + base + ch] = wr.data[size_t(ch) * 25 + t]; }
                // fused layer (k_headdw.h): bf16, 64-wide towers, one launch for the three levels
                const int rbh = headdw_band_rows(x.H, x.W);
                if constexpr (H16E) if (fuse_layers) {
                    if (cur.ld % 8 != 0 || rbh <= 0) throw AchError{ACH_ERR_INVALID, "head: fused layer chosen for a level it does not fit"};
                    Lin lc = conv_bn(c + ".conv.pconv", c + ".bn", 1e-3), lr = conv_bn(r + ".conv.pconv", r + ".bn", 1e-3);
                    std::vector<uint16_t> wp(size_t(2) * 2 * 4 * 64 * 8, 0);
                    std::vector<float> pb(size_t(2) * base, 0.f);
                    for (int br = 0; br < 2; ++br) {
                        const Lin& l = br == 0 ? lc : lr;
                        for (int n = 0; n < base; ++n) pb[size_t(br) * base + n] = l.b[n];
                        for (int s2 = 0; s2 < 2; ++s2)
                            for (int t = 0; t < 4; ++t)
                                for (int ln = 0; ln < 64; ++ln) {
                                    const int i = ln & 15, kg = ln >> 4, n = (t / 2) * 32 + (i / 4) * 8 + (t % 2) * 4 + (i % 4);
                                    for (int e = 0; e < 8; ++e)
                                        wp[(((size_t(br) * 2 + s2) * 4 + t) * 64 + ln) * 8 + e] = H16<T>::bits(l.w[size_t(n) * base + s2 * 32 + kg * 8 + e]);
                                }
                    }
                    A y = alloc(x.B, x.H, x.W, 2 * base);
                    BatchJob bj; bj.kind = 2; bj.name = "det_head.convs." + ks + "." + js + ".dwpw"; bj.shared_in = (j == 0) ? 1 : 0;
                    std::memset(&bj.hj, 0, sizeof(bj.hj));
                    bj.hj.X = cur.p; bj.hj.Y = y.p; bj.hj.ldx = cur.ld; bj.hj.ldy = y.ld; bj.hj.Wdw = up_f32(wt);
                    bj.hj.Wp = static_cast<const uint4*>(up_raw(wp.data(), wp.size() * 2)); bj.hj.bias = up_f32(pb);
                    bj.hj.H = x.H; bj.hj.W = x.W; bj.hj.rb = rbh; bj.hj.bands = cdiv(x.H, rbh);
                    std::memset(&bj.d, 0, sizeof(bj.d)); bj.d.B = x.B;
                    bj.bytes = double(cur.rows()) * cur.C * sizeof(T) + double(y.rows()) * y.C * sizeof(T);
                    bj.flops = 2.0 * double(y.rows()) * 2 * base * base;
                    batch_jobs.push_back(bj);
                    cur = y;
                    continue;
                }
                A d = alloc(x.B, x.H, x.W, 2 * base);
                DwParams dp;
                std::memset(&dp, 0, sizeof(dp));
                dp.X = cur.p; dp.ldx = cur.ld; dp.W = up_f32(wt); dp.bias = up_f32(bias); dp.Y = d.p; dp.ldy = d.ld;
                dp.B = x.B; dp.H = x.H; dp.Wd = x.W; dp.C = 2 * base; dp.Ho = x.H; dp.Wo = x.W; dp.stride = 1; dp.act = ACT_NONE;
                dp.cin_mod = (j == 0) ? base : 0;
                dp.tile = (dw_tile && x.H * x.W <= 144) ? 1 : 0;
                if (batching) {
                    BatchJob bj; bj.kind = 1; bj.name = "det_head.convs." + ks + "." + js + ".dconv"; bj.d = dp; bj.ks = 5;
                    bj.bytes = double(cur.rows()) * cur.C * sizeof(T) + double(d.rows()) * d.C * sizeof(T);
                    batch_jobs.push_back(bj);
                } else
                add_op("det_head.convs." + ks + "." + js + ".dconv", [dp](hipStream_t s) { launch_dwconv<T>(dp, 5, s); },
                       double(cur.rows()) * cur.C * sizeof(T) + double(d.rows()) * d.C * sizeof(T));
                // pointwise, block diagonal
                Lin lc = conv_bn(c + ".conv.pconv", c + ".bn", 1e-3), lr = conv_bn(r + ".conv.pconv", r + ".bn", 1e-3);
                Lin lb; lb.N = 2 * base; lb.K = 2 * base; lb.w.assign(size_t(lb.N) * lb.K, 0.f); lb.b = lc.b; lb.b.insert(lb.b.end(), lr.b.begin(), lr.b.end());
                for (int n = 0; n < base; ++n)
                    for (int kk = 0; kk < base; ++kk) { lb.w[size_t(n) * lb.K + kk] = lc.w[size_t(n) * base + kk]; lb.w[size_t(base + n) * lb.K + base + kk] = lr.w[size_t(n) * base + kk]; }
                A y = alloc(x.B, x.H, x.W, 2 * base);
                GemmOpt o; o.act = ACT_RELU;
                gemm("det_head.convs." + ks + "." + js + ".pconv", d, pack(lb), y, o);
                cur = y;
            }
            // predictions: rows [reg 4 | obj 1] read the reg half, rows [cls] read the cls half
            Lin lreg = lin("det_head.reg_preds." + ks + ".weight", "det_head.reg_preds." + ks + ".bias");
            Lin lobj = lin("det_head.obj_preds." + ks + ".weight", "det_head.obj_preds." + ks + ".bias");
            Lin lcls = lin("det_head.cls_preds." + ks + ".weight", "det_head.cls_preds." + ks + ".bias");
            Lin lp; lp.N = NC5; lp.K = 2 * base; lp.w.assign(size_t(NC5) * lp.K, 0.f); lp.b.assign(size_t(NC5), 0.f);
            for (int n = 0; n < 4; ++n) { lp.b[n] = lreg.b[n]; for (int kk = 0; kk < base; ++kk) lp.w[size_t(n) * lp.K + base + kk] = lreg.w[size_t(n) * base + kk]; }
            lp.b[4] = lobj.b[0];
            for (int kk = 0; kk < base; ++kk) lp.w[size_t(4) * lp.K + base + kk] = lobj.w[kk];
            for (int n = 0; n < cfg.num_det; ++n) { lp.b[5 + n] = lcls.b[n]; for (int kk = 0; kk < base; ++kk) lp.w[size_t(5 + n) * lp.K + kk] = lcls.w[size_t(n) * base + kk]; }
            GemmOpt o1; o1.ydyn = &io.det[k]; o1.out_nchw = 1; o1.HW = HW; o1.Ctot = NC5; o1.coff = 0;
            gemm("det_head.preds." + ks, cur.p, cur.ld, cur.rows(), pack(lp), nullptr, 0, o1);
        }
        if (batching) flush_batch(3, int(batch_jobs.size()) / 3);
    }

    // ------------------------------------------------------------------------------------------ PointNet (a18)
    struct Rows { T* p = nullptr; long rows = 0; int C = 0; long ld = 0; };
    Rows alloc_rows(long rows, int C) { Rows r; r.rows = rows; r.C = C; r.ld = round_up(C, 8); r.p = static_cast<T*>(aalloc(size_t(rows) * r.ld * sizeof(T))); note_region(r.p, size_t(rows) * r.ld * sizeof(T)); return r; }
    Lin lin_bn1d(const std::string& conv, const std::string& bn) const { Lin l = lin(conv + ".weight", conv + ".bias"); if (!bn.empty()) fold_bn(l, bn, 1e-5); return l; }
    Rows pc_layer(const std::string& name, const Rows& x, const Lin& l, int act) {
        Rows y = alloc_rows(x.rows, l.N);
        GemmOpt o; o.act = act;
        gemm(name, x.p, x.ld, x.rows, pack(l), y.p, y.ld, o);
        return y;
    }
    // two consecutive shared-MLP layers as one launch (chain2: the hidden layer never leaves the registers; same rounding points as the two GEMMs);
    // the points are presented as a one-frame map so that the one-tile-per-wave mode is chosen (32 768 rows)
    Rows pc_pair(const std::string& n1, const std::string& n2, const Rows& x, const Lin& l1, int act1, const Lin& l2, int act2) {
        if (pc_chain && act2 == ACT_NONE && x.rows % 16 == 0) {
            A xa; xa.p = x.p; xa.B = 1; xa.H = int(x.rows / 16); xa.W = 16; xa.C = x.C; xa.ld = x.ld;
            A ya;
            if (chain2(n1 + "+" + n2.substr(n2.rfind('.') + 1), xa, l1, act1, l2, ya, nullptr, false)) { Rows y; y.p = ya.p; y.rows = x.rows; y.C = ya.C; y.ld = ya.ld; return y; }
        }
        return pc_layer(n2, pc_layer(n1, x, l1, act1), l2, act2);
    }
    // shared MLP + max over the N points of every sample -> [B, C]   (gemm_colmax_kernel: no atomics, nothing materialised)
    Rows pc_layer_max(const std::string& name, const Rows& x, const Lin& l, int act, int B, const Rows* dst = nullptr, int coff = 0) {
        Packed pk = pack(l);
        Rows y;
        if (dst) { y = *dst; y.p += coff; y.C = l.N; }         // channels [coff, coff + N) of a wider [B, .] tensor (PointNet++ multi-scale: the scales' maxima side by side)
        else y = alloc_rows(B, l.N);
        GemmMaxParams g{x.p, x.ld, pk.w, pk.b, y.p, y.ld, int(x.rows / B), B, pk.K, pk.N, pk.nchunks, pk.ksteps, act};
        const dim3 grid(unsigned(pk.nchunks) * unsigned(B)), block(256);
        const int NT = pk.NT;
        const double bytes = double(x.rows) * pk.K * sizeof(T) + double(pk.group_elems) * sizeof(T) + double(B) * pk.N * sizeof(T);
        // many small groups (PointNet++'s balls): a wave per group instead of a workgroup per group (k_gemm.h gemm_groupmax_kernel; option "group_max", bit-identical)
        if (group_max > 0 && NT == 4 && g.M_per_group <= 64 && B >= group_max) {
            const dim3 gridg(unsigned(pk.nchunks) * unsigned(cdiv(B, 4 * GMAX_PER_WAVE)));
            add_op(name, [g, gridg, block](hipStream_t s) {
                if (g.ksteps == 1) ACH_LAUNCH((gemm_groupmax_kernel<T, 4, 1>), gridg, block, s, g);
                else if (g.ksteps == 2) ACH_LAUNCH((gemm_groupmax_kernel<T, 4, 2>), gridg, block, s, g);
                else if (g.ksteps == 4) ACH_LAUNCH((gemm_groupmax_kernel<T, 4, 4>), gridg, block, s, g);
                else ACH_LAUNCH((gemm_groupmax_kernel<T, 4, 0>), gridg, block, s, g);
            }, bytes, 2.0 * double(x.rows) * pk.K * pk.N);
            return y;
        }
        add_op(name, [g, grid, block, NT](hipStream_t s) {
            if (NT == 1) ACH_LAUNCH((gemm_colmax_kernel<T, 1, 0>), grid, block, s, g);
            else if (NT == 2) ACH_LAUNCH((gemm_colmax_kernel<T, 2, 0>), grid, block, s, g);
            else if (g.ksteps == 4) ACH_LAUNCH((gemm_colmax_kernel<T, 4, 4>), grid, block, s, g);     // weights register-resident
            else if (g.ksteps == 8) ACH_LAUNCH((gemm_colmax_kernel<T, 4, 8>), grid, block, s, g);
            else ACH_LAUNCH((gemm_colmax_kernel<T, 4, 0>), grid, block, s, g);
        }, bytes, 2.0 * double(x.rows) * pk.K * pk.N);
        return y;
    }
    Rows stn(const std::string& pfx, const Rows& x, int B) {                     // pointnet_utils.py:27-45,67-85 (without + I)
        Rows h = pc_layer(pfx + ".conv1", x, lin_bn1d(pfx + ".conv1", pfx + ".bn1"), ACT_RELU);
        h = pc_layer(pfx + ".conv2", h, lin_bn1d(pfx + ".conv2", pfx + ".bn2"), ACT_RELU);
        Rows g = pc_layer_max(pfx + ".conv3", h, lin_bn1d(pfx + ".conv3", pfx + ".bn3"), ACT_RELU, B);
        // (a one-workgroup-per-sample fusion of the three FC layers was measured 4-6x SLOWER than three small GEMM launches:
        //  64 workgroups cannot hide the weight-row latency; the N-chunk split of the GEMM spreads each layer over the chip)
        g = pc_layer(pfx + ".fc1", g, lin_bn1d(pfx + ".fc1", pfx + ".bn4"), ACT_RELU);
        g = pc_layer(pfx + ".fc2", g, lin_bn1d(pfx + ".fc2", pfx + ".bn5"), ACT_RELU);
        return pc_layer(pfx + ".fc3", g, lin_bn1d(pfx + ".fc3", ""), ACT_NONE);
    }
    void pointnet() {                                                            // pointnet_sem_seg.py:26-37
        const std::string p = "pc_seg_model";
        const int B = batch, N = cfg.num_points, D = cfg.pc_channels;
        if (N % 16) throw AchError{ACH_ERR_UNSUPPORTED, "num_points must be a multiple of 16"};
        Rows x0 = alloc_rows(long(B) * N, D);
        {
            PcPrepParams pp{nullptr, x0.p, B, D, N, x0.ld};
            const dim3 grid(unsigned(cdivl(long(B) * N * x0.ld, 256))), block(256);
            const void** pin = &io.points;
            const bool alt = io_alt();
            add_op(p + ".prep", [pp, grid, block, pin, alt](hipStream_t s) mutable { pp.X = *pin; if (alt) ACH_LAUNCH((pc_prep_kernel<T, IOB>), grid, block, s, pp); else ACH_LAUNCH((pc_prep_kernel<T, T>), grid, block, s, pp); });
        }
        Rows t9 = stn(p + ".feat.stn", x0, B);
        { TapInfo t; t.ptr = t9.p; t.kind = 2; t.B = B; t.H = 1; t.W = 1; t.C = 9; t.ld = t9.ld; t.add_eye = 3; add_tap("pc.trans", t); }
        Rows x1 = alloc_rows(long(B) * N, D);
        { PcT3Params q{x0.p, x0.ld, t9.p, t9.ld, x1.p, x1.ld, B, N, D}; ew(p + ".feat.apply_t3", pc_apply_t3_kernel<T>, q, long(B) * N); }
        Rows f1 = pc_layer(p + ".feat.conv1", x1, lin_bn1d(p + ".feat.conv1", p + ".feat.bn1"), ACT_RELU);
        const int kf = f1.C;                                                     // 32
        Rows tf = stn(p + ".feat.fstn", f1, B);
        { TapInfo t; t.ptr = tf.p; t.kind = 2; t.B = B; t.H = 1; t.W = 1; t.C = kf * kf; t.ld = tf.ld; t.add_eye = kf; add_tap("pc.trans_feat", t); }
        // per-sample kf x kf transform as packed MFMA weights
        Packed pk = pack_shape(kf, kf);
        T* wp = static_cast<T*>(aalloc(size_t(B) * pk.group_elems * sizeof(T)));
        pk.b = up_f32(std::vector<float>(static_cast<size_t>(kf), 0.f));
        { PcPackParams q{tf.p, tf.ld, wp, pk.group_elems, B, kf, pk.NT, pk.ksteps}; ew(p + ".feat.pack_tf", pc_pack_transform_kernel<T>, q, long(B) * kf * kf); }
        Rows pf = alloc_rows(long(B) * N, kf);
        { GemmOpt o; o.groups = B; o.w_group_stride = pk.group_elems; o.w_override = wp; gemm(p + ".feat.bmm_tf", f1.p, f1.ld, f1.rows, pk, pf.p, pf.ld, o); }
        Rows h = pc_layer(p + ".feat.conv2", pf, lin_bn1d(p + ".feat.conv2", p + ".feat.bn2"), ACT_RELU);
        Rows g = pc_layer_max(p + ".feat.conv3", h, lin_bn1d(p + ".feat.conv3", p + ".feat.bn3"), ACT_NONE, B);
        { TapInfo t; t.ptr = g.p; t.kind = 2; t.B = B; t.H = 1; t.W = 1; t.C = g.C; t.ld = g.ld; add_tap("pc.global", t); }
        Rows cat = alloc_rows(long(B) * N, g.C + kf);
        { PcConcatParams q{g.p, g.ld, pf.p, pf.ld, cat.p, cat.ld, B, N, g.C, kf}; if ((g.C | kf) & 3) throw AchError{ACH_ERR_UNSUPPORTED, "PointNet feature widths must be multiples of 4"}; ew(p + ".concat", pc_concat_kernel<T>, q, long(B) * N * ((g.C + kf) / 4)); }
        Rows y = pc_layer(p + ".conv1", cat, lin_bn1d(p + ".conv1", p + ".bn1"), ACT_RELU);
        y = pc_layer(p + ".conv2", y, lin_bn1d(p + ".conv2", p + ".bn2"), ACT_RELU);
        y = pc_pair(p + ".conv3", p + ".conv4", y, lin_bn1d(p + ".conv3", p + ".bn3"), ACT_RELU, lin_bn1d(p + ".conv4", ""), ACT_NONE);
        {
            LsmParams q{y.p, y.ld, nullptr, long(B) * N, cfg.pc_classes};
            const dim3 grid(unsigned(cdivl(long(B) * N, 256))), block(256);
            void** out = &io.pc;
            const bool alt = io_alt();
            add_op(p + ".log_softmax", [q, grid, block, out, alt](hipStream_t s) mutable { q.Y = *out; if (alt) ACH_LAUNCH((log_softmax_kernel<T, IOB>), grid, block, s, q); else ACH_LAUNCH((log_softmax_kernel<T, T>), grid, block, s, q); });
        }
    }

    // ------------------------------------------------------------------------------------------ PointNet++ (config 4)
    // OUR OWN specification (spec.py::PN2): the reference snapshot has no PointNet++ code.  Geometry in
    // k_pn2.h (fp32 coordinates, bit-exact index selection); every shared MLP on the MFMA GEMM with (centroid, sample) pairs or
    // points as rows; the max over a ball in the last layer's epilogue.
    struct Pn2Level { float* xyz = nullptr; int n = 0; Rows f; };
    void pointnet2() {
        const std::string p = "pc_seg_model";
        const int B = batch, N = cfg.num_points, D = cfg.pc_channels;
        static const int kDiv[4] = {2, 8, 32, 128};
        static const double kRadius[4] = {0.03, 0.06, 0.12, 0.24};
        static const int kFpLayers[4] = {2, 2, 2, 3};                             // fp4, fp3, fp2, fp1
        // spec.py::PN2 — one (radius, 32 samples) stack per level, keys sa{k}.mlp_convs.<i> — or PN2_MSG (round 6) — two stacks per level on the same centroids, radii [r, 2r],
        // 16 / 32 samples, keys sa{k}.conv_blocks.<scale>.<i>, their maxima concatenated
        const bool msg = cfg.pc_seg == ACH_PCSEG_PN2_MSG;
        const int nscales = msg ? 2 : 1;
        const int kNsS[2] = {msg ? 16 : 32, 32};
        if (N % 128 || N > 64 * PN2_FPS_MAX_PPT) throw AchError{ACH_ERR_UNSUPPORTED, "pn2: num_points must be a multiple of 128, at most 1024"};
        if (D < 3) throw AchError{ACH_ERR_UNSUPPORTED, "pn2: pc_channels must be at least 3 (xyz first)"};
        Pn2Level lv[5];
        int* fidx_all[4] = {nullptr, nullptr, nullptr, nullptr};
        lv[0].n = N; lv[0].f = alloc_rows(long(B) * N, D);
        {
            PcPrepParams pp{nullptr, lv[0].f.p, B, D, N, lv[0].f.ld};
            const dim3 grid(unsigned(cdivl(long(B) * N * lv[0].f.ld, 256))), block(256);
            const void** pin = &io.points;
            const bool alt = io_alt();
            add_op(p + ".prep", [pp, grid, block, pin, alt](hipStream_t s) mutable { pp.X = *pin; if (alt) ACH_LAUNCH((pc_prep_kernel<T, IOB>), grid, block, s, pp); else ACH_LAUNCH((pc_prep_kernel<T, T>), grid, block, s, pp); });
        }
        lv[0].xyz = static_cast<float*>(aalloc(size_t(B) * N * 3 * sizeof(float)));
        { Pn2XyzParams q{lv[0].f.p, lv[0].f.ld, lv[0].xyz, long(B) * N}; ew(p + ".xyz", pn2_xyz_kernel<T>, q, long(B) * N * 3); }
        if (pn2_fps_all) {               // the four levels' farthest-point sampling as ONE launch (k_pn2.h): it depends on the input cloud only
            FpsAllParams fa;
            std::memset(&fa, 0, sizeof(fa));
            fa.levels = 4;
            double bytes = 0;
            for (int k = 0; k < 4; ++k) {
                const int S = N / kDiv[k];
                lv[k + 1].n = S;
                lv[k + 1].xyz = static_cast<float*>(aalloc(size_t(B) * S * 3 * sizeof(float)));
                fidx_all[k] = static_cast<int*>(aalloc(size_t(B) * S * sizeof(int)));
                fa.lv[k] = FpsParams{lv[k].xyz, lv[k].n, S, fidx_all[k], lv[k + 1].xyz};
                bytes += double(B) * (lv[k].n + S) * 12.0;
            }
            const dim3 grid{unsigned(B)}, block(64);
            add_op(p + ".fps_all", [fa, grid, block](hipStream_t s) { ACH_LAUNCH(pn2_fps_all_kernel, grid, block, s, fa); }, bytes);
        }
        for (int k = 0; k < 4; ++k) {
            const std::string sa = p + ".sa" + std::to_string(k + 1);
            const Pn2Level& src = lv[k];
            Pn2Level& dst = lv[k + 1];
            const int S = N / kDiv[k];
            dst.n = S;
            int* fidx = fidx_all[k];
            if (!pn2_fps_all) {
                dst.xyz = static_cast<float*>(aalloc(size_t(B) * S * 3 * sizeof(float)));
                fidx = static_cast<int*>(aalloc(size_t(B) * S * sizeof(int)));
                FpsParams q{src.xyz, src.n, S, fidx, dst.xyz};
                add_op(sa + ".fps", [q, B](hipStream_t s) { launch_pn2_fps(q, B, s); },
                       double(B) * (src.n + S) * 12.0);
            }
            { TapInfo t; t.ptr = dst.xyz; t.kind = 2; t.is_f32 = 1; t.B = B * S; t.H = 1; t.W = 1; t.C = 3; t.ld = 3; add_tap("pc.sa" + std::to_string(k + 1) + ".xyz", t); }
            { TapInfo t; t.ptr = fidx; t.kind = 2; t.is_i32 = 1; t.B = B; t.H = 1; t.W = 1; t.C = S; t.ld = S; add_tap("pc.sa" + std::to_string(k + 1) + ".fps", t); }
            int wtot = 0, wsc[2] = {0, 0};
            for (int j = 0; j < nscales; ++j) {
                const std::string last = msg ? sa + ".conv_blocks." + std::to_string(j) + ".2.weight" : sa + ".mlp_convs.2.weight";
                wsc[j] = int(W(last).shape[0]);
                wtot += wsc[j];
            }
            if (msg) dst.f = alloc_rows(long(B) * S, wtot);
            int coff = 0;
            for (int j = 0; j < nscales; ++j) {
                const int kNs = kNsS[j];
                const double radius = kRadius[k] * (j == 0 ? 1.0 : 2.0);
                const std::string sfx = msg ? "." + std::to_string(j) : "";
                auto cname = [&](int i) { return msg ? sa + ".conv_blocks." + std::to_string(j) + "." + std::to_string(i) : sa + ".mlp_convs." + std::to_string(i); };
                auto bname = [&](int i) { return msg ? sa + ".bn_blocks." + std::to_string(j) + "." + std::to_string(i) : sa + ".mlp_bns." + std::to_string(i); };
                Rows g = alloc_rows(long(B) * S * kNs, 3 + src.f.C);
                int* gidx = nullptr;
                if (full_taps) {                                                      // parity hook: the ball-query selections themselves
                    gidx = static_cast<int*>(aalloc(size_t(B) * S * kNs * sizeof(int)));
                    TapInfo t; t.ptr = gidx; t.kind = 2; t.is_i32 = 1; t.B = B * S; t.H = 1; t.W = 1; t.C = kNs; t.ld = kNs; add_tap("pc.sa" + std::to_string(k + 1) + ".group_idx" + sfx, t);
                }
                {
                    const int wpc = (group_wpc > 0 && long(B) * S <= group_wpc) ? 4 : 1;          // few centroids: a workgroup per centroid (k_pn2.h)
                    GroupParams q{src.xyz, dst.xyz, src.f.p, src.f.ld, src.f.C, g.p, g.ld, gidx, B, src.n, S, kNs, float(radius * radius), wpc};
                    const dim3 grid(unsigned(wpc == 4 ? long(B) * S : cdivl(long(B) * S, 4))), block(256);
                    add_op(sa + ".group" + sfx, [q, grid, block](hipStream_t s) { ACH_LAUNCH(pn2_group_kernel<T>, grid, block, s, q); },
                           double(g.rows) * g.C * sizeof(T) * 2.0);
                }
                Rows h = pc_layer(sa + ".mlp" + sfx + ".0", g, lin_bn1d(cname(0), bname(0)), ACT_RELU);
                h = pc_layer(sa + ".mlp" + sfx + ".1", h, lin_bn1d(cname(1), bname(1)), ACT_RELU);
                if (msg) pc_layer_max(sa + ".mlp" + sfx + ".2", h, lin_bn1d(cname(2), bname(2)), ACT_RELU, B * S, &dst.f, coff);
                else dst.f = pc_layer_max(sa + ".mlp.2", h, lin_bn1d(cname(2), bname(2)), ACT_RELU, B * S);
                coff += wsc[j];
            }
            { TapInfo t; t.ptr = dst.f.p; t.kind = 2; t.B = B * S; t.H = 1; t.W = 1; t.C = dst.f.C; t.ld = dst.f.ld; add_tap("pc.sa" + std::to_string(k + 1) + ".feat", t); }
        }
        Rows cur = lv[4].f;
        for (int j = 0; j < 4; ++j) {
            const int lvl = 3 - j;                                                // dense level
            const std::string fp = p + ".fp" + std::to_string(lvl + 1);
            const Pn2Level& dn = lv[lvl];
            const Pn2Level& sp = lv[lvl + 1];
            if (sp.n < 3 || sp.n > 64 * PN2_INTERP_SPL) throw AchError{ACH_ERR_UNSUPPORTED, "pn2: level size outside the interpolation kernel's range"};
            const int C1 = lvl > 0 ? dn.f.C : 0;
            Rows cat = alloc_rows(long(B) * dn.n, C1 + cur.C);
            {
                InterpParams q{dn.xyz, sp.xyz, dn.f.p, dn.f.ld, C1, cur.p, cur.ld, cur.C, cat.p, cat.ld, B, dn.n, sp.n};
                const dim3 grid(unsigned(cdivl(long(B) * dn.n, 4))), block(256);
                add_op(fp + ".interp", [q, grid, block](hipStream_t s) { ACH_LAUNCH(pn2_interp_kernel<T>, grid, block, s, q); },
                       double(cat.rows) * cat.C * sizeof(T) * 2.0);
            }
            cur = cat;
            for (int i = 0; i < kFpLayers[j]; ++i)
                cur = pc_layer(fp + ".mlp." + std::to_string(i), cur, lin_bn1d(fp + ".mlp_convs." + std::to_string(i), fp + ".mlp_bns." + std::to_string(i)), ACT_RELU);
            { TapInfo t; t.ptr = cur.p; t.kind = 2; t.B = B * dn.n; t.H = 1; t.W = 1; t.C = cur.C; t.ld = cur.ld; add_tap("pc.fp" + std::to_string(lvl + 1), t); }
        }
        Rows y = pc_layer(p + ".conv1", cur, lin_bn1d(p + ".conv1", p + ".bn1"), ACT_RELU);
        y = pc_layer(p + ".conv2", y, lin_bn1d(p + ".conv2", ""), ACT_NONE);
        {
            LsmParams q{y.p, y.ld, nullptr, long(B) * N, cfg.pc_classes};
            const dim3 grid(unsigned(cdivl(long(B) * N, 256))), block(256);
            void** out = &io.pc;
            const bool alt = io_alt();
            add_op(p + ".log_softmax", [q, grid, block, out, alt](hipStream_t s) mutable { q.Y = *out; if (alt) ACH_LAUNCH((log_softmax_kernel<T, IOB>), grid, block, s, q); else ACH_LAUNCH((log_softmax_kernel<T, T>), grid, block, s, q); });
        }
    }

    // ------------------------------------------------------------------------------------------ plan (a1)
    void build() {
        if (cfg.resolution % 32 || cfg.resolution < 64) throw AchError{ACH_ERR_INVALID, "resolution must be a multiple of 32"};
        // enqueue order = plan order: the two side branches first, so they are already running while the (longest) image
        // path is being enqueued on the caller's stream
        A r[3];
        // "head_stream": PointNet queues behind the radar branch on the low-priority stream; the (four times longer) PointNet++
        // branch opens stream 2 instead, ahead of fusion + head, which only start once the neck is done (measured +4.8 %; PointNet: neutral)
        // (auto: stream 2 for PointNet++, and in the pipelined plan, where it shares the stream with the decoders: 27.7 k against 26.4 k
        //  frames/s; the plain plan keeps PointNet on stream 1 ahead of the radar branch — two streams in all: 26.35 k against 25.95 k)
        const bool point2 = point_on_head_stream < 0 ? (cfg.pc_seg == ACH_PCSEG_PN2 || cfg.pc_seg == ACH_PCSEG_PN2_MSG || (!head_stream && pipeline)) : point_on_head_stream != 0;
        const bool radar_late = radar_start_eff() >= 0 && multi_stream;
        auto points = [&] { cur_stream = point_on_head_stream == 3 ? 3 : (point2 ? 2 : 1); if (cfg.pc_seg == ACH_PCSEG_PN2 || cfg.pc_seg == ACH_PCSEG_PN2_MSG) pointnet2(); else if (cfg.pc_seg == ACH_PCSEG_PN) pointnet(); };   // ACH_PCSEG_NONE: Achelous3T
        if (radar_late) points();         // the point branch (small launches) fills the window before the radar branch is released
        cur_stream = 1;
        if (radar_late) wait_before_next(0);
        rcnet(r);
        signal_after_last(3);             // radar pyramid ready
        if (!radar_late) points();
        cur_stream = 0;
        A m[4];
        cat_buf[0] = A(); cat_buf[1] = A();
        // the neck on stream 2 needs the stage outputs' launches to be identifiable (the preset concat slices of the EdgeNeXt plans) and the radar branch released before stage 3
        neck_on_side = pipeline && multi_stream && split_decoders == 0 && dec_fork == 3 && cfg.backbone == ACH_BACKBONE_EDGENEXT && radar_start_eff() < 3;
        if (cfg.backbone == ACH_BACKBONE_EDGENEXT) {
            // stage 1 / stage 2 outputs go straight into the second half of the neck's concat buffers (no cat copy)
            const int* wd = widths();
            const int R = cfg.resolution;
            cat_buf[0] = alloc(batch, R / 8, R / 8, 2 * wd[1]);
            cat_buf[1] = alloc(batch, R / 16, R / 16, 2 * wd[2]);
            const A d1 = cat_buf[0].slice(wd[1], wd[1]), d2 = cat_buf[1].slice(wd[2], wd[2]);
            const A* dst[4] = {nullptr, &d1, &d2, nullptr};
            edgenext("image_radar_encoder.fpn.backbone", m, dst);
        }
        else mobilevit("image_radar_encoder.fpn.backbone", m);
        A q[3];
        neck(m, q);
        tap("q3", q[0]); tap("q4", q[1]); tap("q5", q[2]);
        // detection branch: fusion + head continue on the RADAR stream (in order after the radar taps), gated only on the FPN
        // outputs (event 1) — they overlap with the segmentation decoders that keep the caller's stream busy
        // ("head_stream": on stream 2 at the caller's priority, gated on both the FPN outputs and the radar pyramid)
        cur_stream = head_stream ? 2 : 1;
        detect_stream = cur_stream;
        wait_before_next(1);
        if (head_stream) wait_before_next2(3);
        A p[3];
        fuse_all(q, r, p);
        head(p);
        cur_stream = 0;
        interleave_streams();
    }

    // Host enqueue order = plan order.  Built branch by branch, the plan would enqueue all ~65 radar launches before the first
    // image-path launch, delaying the critical path by the host cost of those launches (measured: the image path started 1.5 ms
    // late under rocprofv3).  Merge the per-stream sequences proportionally instead, keeping each stream's own order and never
    // placing a wait before the launch that signals its event.
    void interleave_streams() {
        if (measuring || ops.size() < 2) return;
        std::vector<std::vector<Op>> seq(1 + kSideStreams);
        for (auto& op : ops) seq[size_t(op.stream)].push_back(std::move(op));
        const size_t total = ops.size();
        ops.clear();
        std::vector<size_t> pos(seq.size(), 0);
        bool signalled[kJoinEvents] = {false, false, false, false};
        while (ops.size() < total) {
            // pick the stream that is furthest behind its proportional share and whose next launch is not blocked
            int best = -1; double best_lag = -1e30;
            for (size_t k = 0; k < seq.size(); ++k) {
                if (pos[k] >= seq[k].size()) continue;
                const Op& nx = seq[k][pos[k]];
                if (nx.wait_ev >= 0 && !signalled[nx.wait_ev]) continue;
                if (nx.wait_ev2 >= 0 && !signalled[nx.wait_ev2]) continue;
                const double lag = double(ops.size() + 1) * double(seq[k].size()) / double(total) - double(pos[k]);
                if (lag > best_lag) { best_lag = lag; best = int(k); }
            }
            if (best < 0) throw AchError{ACH_ERR_INVALID, "plan interleave: unsatisfiable event order"};
            Op& op = seq[size_t(best)][pos[size_t(best)]++];
            if (op.signal_ev >= 0) signalled[op.signal_ev] = true;
            ops.push_back(std::move(op));
        }
    }

    void plan(int B) override {
        if (B <= 0) throw AchError{ACH_ERR_INVALID, "batch must be positive"};
        if (weights.empty()) throw AchError{ACH_ERR_INVALID, "ach_load_weights must precede ach_plan"};
        // several kernels address an activation tensor with 32-bit byte offsets (range-checked buffer resources): the largest NHWC
        // tensor of a plan (full resolution x 32 channels) must stay below 2 GiB — 327 frames at 320x320 in bf16
        if (double(B) * cfg.resolution * cfg.resolution * 32.0 * sizeof(T) >= 2147483648.0)
            throw AchError{ACH_ERR_UNSUPPORTED, "batch too large for one plan (activation tensors of 2 GiB or more): split the batch"};
        // the arenas are rewritten below with synchronous copies on the null stream, which does not order against non-blocking
        // streams: a forward still in flight on the caller's or the side streams must have drained first
        ACH_HIP_CHECK(hipDeviceSynchronize());
        batch = B;
        reset_plan();
        measuring = true;
        build();
        const size_t wneed = warena_used + (1 << 20), aneed = aarena_used + (1 << 20);
        measuring = false;
        if (wneed > warena_cap) { if (warena) (void)hipFree(warena); warena = nullptr; ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&warena), wneed)); warena_cap = wneed; }
        if (aneed > aarena_cap) { if (aarena) (void)hipFree(aarena); aarena = nullptr; ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&aarena), aneed)); aarena_cap = aneed; }
        reset_plan();
        post_plan.clear();
        build();
        ACH_HIP_CHECK(hipMemset(aarena, 0, aarena_used));      // channel padding lanes stay zero for the lifetime of the plan
        for (auto& f : post_plan) f();                          // tensors that start from something other than zero (the first RCBlock's background, radar_bg)
        ACH_HIP_CHECK(hipDeviceSynchronize());
    }

    // ---- pre / post-processing
    void preprocess_radar(int B, int C, const float* in, void* out, hipStream_t s) override {
        const int S = 64;
        const size_t need = size_t(B) * S * 2 * sizeof(float);
        if (need > prepost_scratch_bytes) {
            if (prepost_scratch) (void)hipFree(prepost_scratch);
            ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&prepost_scratch), need));
            prepost_scratch_bytes = need;
        }
        const long per = long(C) * cfg.resolution * cfg.resolution;
        MinMaxParams pm{in, prepost_scratch, per, S};
        ACH_LAUNCH(frame_minmax_kernel, dim3(unsigned(B), unsigned(S)), dim3(256), s, pm);
        RadarScaleParams ps{in, prepost_scratch, out, per, S, B};
        if (io_alt()) ACH_LAUNCH(radar_scale_kernel<IOB>, dim3(unsigned(cdivl(per * B, 256))), dim3(256), s, ps);
        else ACH_LAUNCH(radar_scale_kernel<T>, dim3(unsigned(cdivl(per * B, 256))), dim3(256), s, ps);
    }
    void normalize_points(int B, int N, int D, const float* in, void* out, hipStream_t s) override {
        PointNormParams pp{in, out, B, N, D};
        if (io_alt()) ACH_LAUNCH(point_norm_kernel<IOB>, dim3(unsigned(B * D)), dim3(256), s, pp);
        else ACH_LAUNCH(point_norm_kernel<T>, dim3(unsigned(B * D)), dim3(256), s, pp);
    }
    void preprocess_image(int B, const unsigned char* in, void* out, hipStream_t s) override {
        ImagePrepParams pp{in, out, B, cfg.resolution, cfg.resolution};
        if (io_alt()) ACH_LAUNCH(image_prep_kernel<IOB>, dim3(unsigned(cdivl(long(B) * cfg.resolution * cfg.resolution, 256))), dim3(256), s, pp);
        else ACH_LAUNCH(image_prep_kernel<T>, dim3(unsigned(cdivl(long(B) * cfg.resolution * cfg.resolution, 256))), dim3(256), s, pp);
    }
    void seg_argmax(int B, int C, const void* seg, unsigned char* out, hipStream_t s) override {
        SegArgmaxParams pp{seg, out, B, C, long(cfg.resolution) * cfg.resolution};
        if (io_alt()) ACH_LAUNCH(seg_argmax_kernel<IOB>, dim3(unsigned(cdivl(pp.HW * B, 256))), dim3(256), s, pp);
        else ACH_LAUNCH(seg_argmax_kernel<T>, dim3(unsigned(cdivl(pp.HW * B, 256))), dim3(256), s, pp);
    }

    // achelous.py:283-318: softmax -> crop the letterbox bars -> INTER_LINEAR resize to the original size -> argmax (k_prepost.h)
    void seg_resize_argmax(int B, int C, const void* seg, int out_h, int out_w, float* prob_ws, unsigned char* out, hipStream_t s) override {
        const int R = cfg.resolution;
        const long HW = long(R) * R;
        SegSoftmaxParams sp{seg, prob_ws, B, C, HW};
        if (io_alt()) ACH_LAUNCH(seg_softmax_kernel<IOB>, dim3(unsigned(cdivl(HW * B, 256))), dim3(256), s, sp);
        else ACH_LAUNCH(seg_softmax_kernel<T>, dim3(unsigned(cdivl(HW * B, 256))), dim3(256), s, sp);
        // utils_seg/utils.py:19-31 (resize_image): scale = min(w / iw, h / ih), nw = int(iw * scale), nh = int(ih * scale), centred
        const double scale = std::min(double(R) / double(out_w), double(R) / double(out_h));
        const int nw = std::max(1, int(double(out_w) * scale)), nh = std::max(1, int(double(out_h) * scale));
        SegResizeParams rp{prob_ws, out, B, C, R, (R - nh) / 2, (R - nw) / 2, nh, nw, out_h, out_w, double(nh) / double(out_h), double(nw) / double(out_w)};
        ACH_LAUNCH(seg_resize_argmax_kernel, dim3(unsigned(cdivl(long(out_h) * out_w * B, 256))), dim3(256), s, rp);
    }

    float bench_gemm(int M, int K, int N, int act, int ln, int residual, int Pforce, int iters, hipStream_t s) override {
        Packed pk = pack_shape(N, K);
        const long ldx = round_up(K, 8), ldy = round_up(N, 8);
        T *X = nullptr, *Y = nullptr, *R = nullptr, *Wp = nullptr; float* bias = nullptr;
        ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&X), size_t(M) * ldx * sizeof(T)));
        ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&Y), size_t(M) * ldy * sizeof(T)));
        ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&R), size_t(M) * ldy * sizeof(T)));
        ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&Wp), size_t(pk.group_elems) * sizeof(T)));
        ACH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&bias), (size_t(N) + 64) * sizeof(float)));      // GemmParams.bias: 64 zero floats behind the last channel (gemm_body's 16-byte loads)
        ACH_HIP_CHECK(hipMemset(X, 0x3c, size_t(M) * ldx * sizeof(T)));     // small finite values
        ACH_HIP_CHECK(hipMemset(R, 0x3c, size_t(M) * ldy * sizeof(T)));
        ACH_HIP_CHECK(hipMemset(Wp, 0x3c, size_t(pk.group_elems) * sizeof(T)));
        ACH_HIP_CHECK(hipMemset(bias, 0, (size_t(N) + 64) * sizeof(float)));
        GemmParams g;
        std::memset(&g, 0, sizeof(g));
        g.X = X; g.ldx = ldx; g.W = Wp; g.bias = bias; g.Y = Y; g.ldy = ldy; g.R = residual ? R : nullptr; g.ldr = ldy;
        g.groups = 1; g.M_per_group = M; g.K = K; g.N = N; g.nchunks = pk.nchunks; g.ksteps = pk.ksteps;
        g.act = act; g.ln = ln; g.ln_eps = 1e-6f; g.vec_store = 1;
        int P = 1;
        if (Pforce > 0) P = Pforce;
        const long row_blocks = cdivl(M, 64L * P);
        const int zsplit = int(std::min<long>(pk.nchunks, std::max<long>(1, cdivl(1024, row_blocks))));
        g.chunks_per_block = cdiv(pk.nchunks, zsplit);
        hipEvent_t e0, e1;
        ACH_HIP_CHECK(hipEventCreate(&e0)); ACH_HIP_CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) launch_gemm<T>(g, pk.NT, P, s);
        ACH_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) launch_gemm<T>(g, pk.NT, P, s);
        ACH_HIP_CHECK(hipEventRecord(e1, s));
        ACH_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        ACH_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(X); (void)hipFree(Y); (void)hipFree(R); (void)hipFree(Wp); (void)hipFree(bias);
        return ms / float(iters);
    }

    unsigned long long count_saturated(hipStream_t s) override {
        if constexpr (!std::is_same<T, f16_t>::value) { (void)s; return 0ull; }
        else {
            if (ops.empty()) throw AchError{ACH_ERR_INVALID, "ach_plan must precede ach_count_saturated"};
            const size_t n = t_regions.size();
            if (n == 0) return 0ull;
            if (!sat_dev || sat_dev_regions != n) {
                if (sat_dev) (void)hipFree(sat_dev);
                sat_dev = nullptr;
                ACH_HIP_CHECK(hipMalloc(&sat_dev, 256 + n * sizeof(SatRegion)));
                std::vector<SatRegion> tab(n);
                for (size_t i = 0; i < n; ++i) tab[i] = SatRegion{static_cast<const uint32_t*>(t_regions[i].first), static_cast<unsigned long long>((t_regions[i].second + 3) / 4)};
                ACH_HIP_CHECK(hipMemcpy(static_cast<char*>(sat_dev) + 256, tab.data(), n * sizeof(SatRegion), hipMemcpyHostToDevice));
                sat_dev_regions = n;
            }
            ACH_HIP_CHECK(hipMemsetAsync(sat_dev, 0, 8, s));
            ACH_LAUNCH(sat_count_kernel, dim3(64, unsigned(n)), dim3(256), s, reinterpret_cast<const SatRegion*>(static_cast<char*>(sat_dev) + 256),
                       static_cast<unsigned long long*>(sat_dev));
            unsigned long long out = 0;
            ACH_HIP_CHECK(hipMemcpyAsync(&out, sat_dev, 8, hipMemcpyDeviceToHost, s));
            ACH_HIP_CHECK(hipStreamSynchronize(s));
            return out;
        }
    }

    void decode(int B, const void* d3, const void* d4, const void* d5, float* out, hipStream_t s) override {
        DecodeParams p;
        const int r = cfg.resolution;
        p.det[0] = d3; p.det[1] = d4; p.det[2] = d5;
        p.h[0] = p.w[0] = r / 8; p.h[1] = p.w[1] = r / 16; p.h[2] = p.w[2] = r / 32;
        p.out = out; p.B = B; p.NC5 = 5 + cfg.num_det; p.A = num_anchors(); p.in_h = float(r); p.in_w = float(r);
        if (io_alt()) ACH_LAUNCH(decode_kernel<IOB>, dim3(unsigned(cdivl(long(B) * p.A, 256))), dim3(256), s, p);
        else ACH_LAUNCH(decode_kernel<T>, dim3(unsigned(cdivl(long(B) * p.A, 256))), dim3(256), s, p);
    }
};

}  // namespace ach
