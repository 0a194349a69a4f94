// api.cpp — the extern "C" boundary of libachelous_hip.so (declared in include/achelous.h).
// No exception crosses it: every failure becomes a negative code + a message retrievable with ach_last_error().
#include <cstring>
#include <functional>
#include <string>

#include "engine.h"
#include "k_train.h"

struct ach_handle {
    ach::EngineBase* eng = nullptr;
    std::string err;
};

static thread_local std::string g_create_error;

template <class F>
static int guarded(ach_handle* h, F&& f) {
    if (!h || !h->eng) return ACH_ERR_INVALID;
    try {
        f();
        return ACH_OK;
    } catch (const ach::AchError& e) {
        h->err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        h->err = e.what();
        return ACH_ERR_INVALID;
    } catch (...) {
        h->err = "unknown error";
        return ACH_ERR_INVALID;
    }
}

extern "C" {

int ach_create(const ach_config* cfg, ach_handle** out) {
    if (!cfg || !out) return ACH_ERR_INVALID;
    try {
        if (cfg->backbone != ACH_BACKBONE_EDGENEXT && cfg->backbone != ACH_BACKBONE_MOBILEVIT)
            throw ach::AchError{ACH_ERR_UNSUPPORTED, "backbone must be 'en' or 'mv'"};
        if (cfg->phi < ACH_PHI_S0 || cfg->phi > ACH_PHI_S2) throw ach::AchError{ACH_ERR_UNSUPPORTED, "phi must be S0, S1 or S2"};
        if (cfg->neck != ACH_NECK_GDF && cfg->neck != ACH_NECK_CDF) throw ach::AchError{ACH_ERR_UNSUPPORTED, "neck must be 'gdf' or 'cdf'"};
        if (cfg->pc_seg != ACH_PCSEG_PN && cfg->pc_seg != ACH_PCSEG_PN2) throw ach::AchError{ACH_ERR_UNSUPPORTED, "pc_seg must be 'pn' or 'pn2'"};
        if (!cfg->nano_head) throw ach::AchError{ACH_ERR_UNSUPPORTED, "only nano_head=True is built"};
        if (cfg->num_det < 1 || cfg->num_det > 59 || cfg->num_seg < 1 || cfg->pc_classes < 1 || cfg->pc_channels < 3)
            throw ach::AchError{ACH_ERR_INVALID, "bad class / channel counts"};
        ach_handle* h = new ach_handle();
        h->eng = ach::make_engine(*cfg);
        *out = h;
        return ACH_OK;
    } catch (const ach::AchError& e) {
        g_create_error = e.msg;
        return e.code;
    } catch (...) {
        g_create_error = "allocation failure";
        return ACH_ERR_NOMEM;
    }
}

void ach_destroy(ach_handle* h) {
    if (!h) return;
    delete h->eng;
    delete h;
}

const char* ach_last_error(const ach_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ach_load_weights(ach_handle* h, const ach_tensor_desc* tensors, size_t n) {
    return guarded(h, [&] { h->eng->load(tensors, n); });
}

int ach_set_option(ach_handle* h, const char* key, int32_t value) {
    return guarded(h, [&] {
        if (!key) throw ach::AchError{ACH_ERR_INVALID, "null option"};
        if (std::string(key) == "full_taps") h->eng->full_taps = value != 0;
        else if (std::string(key) == "streams") h->eng->multi_stream = value != 0;
        else if (std::string(key) == "graph") h->eng->use_graph = value != 0;
        else if (std::string(key) == "fused_mlp") h->eng->fuse_mlp = value != 0;
        else if (std::string(key) == "mlp_split") h->eng->mlp_split = value;
        else if (std::string(key) == "row_conv") h->eng->row_conv = value != 0;
        else if (std::string(key) == "fused_rc") h->eng->fuse_rc = value != 0;
        else if (std::string(key) == "dw_tile") h->eng->dw_tile = value != 0;
        else if (std::string(key) == "stem_mfma") h->eng->stem_mfma = value != 0;
        else if (std::string(key) == "point_stream2") h->eng->point_on_head_stream = value;
        else if (std::string(key) == "head_batch") h->eng->head_batch = value != 0;
        else if (std::string(key) == "side_priority") h->eng->side_low_priority = value;
        else if (std::string(key) == "head_stream") h->eng->head_stream = value != 0;
        else if (std::string(key) == "split_decoders") h->eng->split_decoders = value;
        else if (std::string(key) == "radar_start") h->eng->radar_start = value;
        else if (std::string(key) == "pipeline") h->eng->pipeline = value != 0;
        else if (std::string(key) == "pool_strip") h->eng->pool_strip = value;
        else if (std::string(key) == "fused_mv2") h->eng->fuse_mv2 = value != 0;
        else if (std::string(key) == "radar_skip") h->eng->radar_skip = value != 0;
        else if (std::string(key) == "head_mfma") h->eng->head_mfma = value != 0;
        else if (std::string(key) == "head_grid") h->eng->head_grid = value;
        else if (std::string(key) == "head_debug") h->eng->head_debug = value;
        else if (std::string(key) == "attn_mfma") h->eng->attn_mfma = value != 0;
        else throw ach::AchError{ACH_ERR_INVALID, std::string("unknown option: ") + key};
    });
}

int ach_plan(ach_handle* h, int32_t batch) {
    return guarded(h, [&] { h->eng->plan(batch); });
}

size_t ach_arena_bytes(const ach_handle* h) { return (h && h->eng) ? h->eng->aarena_used + h->eng->warena_used : 0; }

int ach_forward(ach_handle* h, const void* image, const void* radar, const void* points, void* det3, void* det4, void* det5,
                void* se_seg, void* lane_seg, void* pc_seg, void* stream) {
    return guarded(h, [&] {
        if (h->eng->ops.empty()) throw ach::AchError{ACH_ERR_INVALID, "ach_plan must precede ach_forward"};
        if (!image || !radar || !points || !det3 || !det4 || !det5 || !se_seg || !lane_seg || !pc_seg)
            throw ach::AchError{ACH_ERR_INVALID, "null input/output pointer"};
        ach::IoPtrs& io = h->eng->io;
        io.image = image; io.radar = radar; io.points = points;
        io.det[0] = det3; io.det[1] = det4; io.det[2] = det5; io.se = se_seg; io.lane = lane_seg; io.pc = pc_seg;
        h->eng->run(static_cast<hipStream_t>(stream));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) throw ach::AchError{ACH_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e)};
    });
}

int ach_forward_detect(ach_handle* h, const void* image, const void* radar, const void* points, void* det3, void* det4, void* det5,
                       void* se_seg, void* lane_seg, void* pc_seg, float* decoded, float conf_thres, float nms_thres, int32_t max_det,
                       float* out_rows, int32_t* out_idx, int32_t* out_count, void* workspace, void* stream) {
    return guarded(h, [&] {
        if (h->eng->ops.empty()) throw ach::AchError{ACH_ERR_INVALID, "ach_plan must precede ach_forward_detect"};
        if (!image || !radar || !points || !det3 || !det4 || !det5 || !se_seg || !lane_seg || !pc_seg)
            throw ach::AchError{ACH_ERR_INVALID, "null input/output pointer"};
        if (max_det <= 0 || !decoded || !out_rows || !out_idx || !out_count || !workspace)
            throw ach::AchError{ACH_ERR_INVALID, "bad detect arguments"};
        ach::IoPtrs& io = h->eng->io;
        io.image = image; io.radar = radar; io.points = points;
        io.det[0] = det3; io.det[1] = det4; io.det[2] = det5; io.se = se_seg; io.lane = lane_seg; io.pc = pc_seg;
        ach::EngineBase* e = h->eng;
        const int B = e->batch;
        e->detect_tail = [=](hipStream_t st) {
            e->decode(B, det3, det4, det5, decoded, st);
            e->nms(B, decoded, conf_thres, nms_thres, max_det, out_rows, out_idx, out_count, workspace, st);
        };
        struct Clear { ach::EngineBase* e; ~Clear() { e->detect_tail = nullptr; } } clear{e};
        e->run_eager(static_cast<hipStream_t>(stream));          // the tail carries per-call arguments: never replayed from a graph
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) throw ach::AchError{ACH_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(err)};
    });
}

int ach_join(ach_handle* h, void* stream) { return guarded(h, [&] { h->eng->join(static_cast<hipStream_t>(stream)); }); }
int ach_forwards_in_flight(const ach_handle* h) { return (h && h->eng) ? int(h->eng->forwards_in_flight()) : 0; }

int ach_decode(ach_handle* h, int32_t batch, const void* det3, const void* det4, const void* det5, float* decoded, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || !det3 || !det4 || !det5 || !decoded) throw ach::AchError{ACH_ERR_INVALID, "bad decode arguments"};
        h->eng->decode(batch, det3, det4, det5, decoded, static_cast<hipStream_t>(stream));
    });
}

size_t ach_nms_workspace_bytes(const ach_handle* h, int32_t batch) { return (h && h->eng) ? h->eng->nms_workspace_bytes(batch) : 0; }

int ach_nms(ach_handle* h, int32_t batch, const float* decoded, float conf_thres, float nms_thres, int32_t max_det,
            float* out_rows, int32_t* out_idx, int32_t* out_count, void* workspace, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || max_det <= 0 || !decoded || !out_rows || !out_idx || !out_count || !workspace)
            throw ach::AchError{ACH_ERR_INVALID, "bad nms arguments"};
        h->eng->nms(batch, decoded, conf_thres, nms_thres, max_det, out_rows, out_idx, out_count, workspace, static_cast<hipStream_t>(stream));
    });
}

int ach_preprocess_radar(ach_handle* h, int32_t batch, int32_t channels, const float* in, void* out, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || channels <= 0 || !in || !out) throw ach::AchError{ACH_ERR_INVALID, "bad preprocess_radar arguments"};
        h->eng->preprocess_radar(batch, channels, in, out, static_cast<hipStream_t>(stream));
    });
}
int ach_normalize_points(ach_handle* h, int32_t batch, int32_t n, int32_t d, const float* in, void* out, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || n <= 0 || d <= 0 || !in || !out) throw ach::AchError{ACH_ERR_INVALID, "bad normalize_points arguments"};
        h->eng->normalize_points(batch, n, d, in, out, static_cast<hipStream_t>(stream));
    });
}
int ach_preprocess_image(ach_handle* h, int32_t batch, const uint8_t* in, void* out, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || !in || !out) throw ach::AchError{ACH_ERR_INVALID, "bad preprocess_image arguments"};
        h->eng->preprocess_image(batch, in, out, static_cast<hipStream_t>(stream));
    });
}
int ach_seg_argmax(ach_handle* h, int32_t batch, int32_t channels, const void* seg, uint8_t* out, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || channels <= 0 || channels > 255 || !seg || !out) throw ach::AchError{ACH_ERR_INVALID, "bad seg_argmax arguments"};
        h->eng->seg_argmax(batch, channels, seg, out, static_cast<hipStream_t>(stream));
    });
}

int ach_seg_resize_argmax(ach_handle* h, int32_t batch, int32_t channels, const void* seg, int32_t out_h, int32_t out_w, float* prob_workspace,
                          uint8_t* out, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || channels <= 0 || channels > 255 || out_h <= 0 || out_w <= 0 || !seg || !prob_workspace || !out)
            throw ach::AchError{ACH_ERR_INVALID, "bad seg_resize_argmax arguments"};
        h->eng->seg_resize_argmax(batch, channels, seg, out_h, out_w, prob_workspace, out, static_cast<hipStream_t>(stream));
    });
}
int ach_correct_boxes(ach_handle* h, int32_t batch, int32_t max_det, const float* rows, const int32_t* count, int32_t image_h, int32_t image_w,
                      int32_t letterbox, float* out_rows, void* stream) {
    return guarded(h, [&] {
        if (batch <= 0 || max_det <= 0 || image_h <= 0 || image_w <= 0 || !rows || !count || !out_rows) throw ach::AchError{ACH_ERR_INVALID, "bad correct_boxes arguments"};
        h->eng->correct_boxes(batch, max_det, rows, count, image_h, image_w, letterbox, out_rows, static_cast<hipStream_t>(stream));
    });
}

// ---- training-mode kernels (k_train.h): stateless, fp32, no handle
static int train_guard(const std::function<void()>& fn) {
    try { fn(); hipError_t e = hipGetLastError(); if (e != hipSuccess) { g_create_error = std::string("kernel launch: ") + hipGetErrorString(e); return ACH_ERR_DEVICE; } return ACH_OK; }
    catch (const ach::AchError& e) { g_create_error = e.msg; return e.code; }
    catch (const std::exception& e) { g_create_error = e.what(); return ACH_ERR_INVALID; }
}
int ach_train_gemm(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                   int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t trans_a, int32_t trans_b, int32_t batch, int32_t reduce_batch,
                   int32_t accumulate, void* stream) {
    return train_guard([&] {
        if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_gemm arguments"};
        ach::TrainGemmParams p{A, B, C, bias, M, N, K, long(lda), long(ldb), long(ldc), long(stride_a), long(stride_b), long(stride_c), trans_a, trans_b, batch, reduce_batch, accumulate};
        const dim3 grid(unsigned((N + 63) / 64), unsigned((M + 63) / 64), unsigned(reduce_batch ? 1 : batch));
        ACH_LAUNCH(ach::train_gemm_kernel, grid, dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_bn_stats(const float* z, float* mean, float* var, int32_t B, int32_t C, int32_t N, void* stream) {
    return train_guard([&] {
        if (!z || !mean || !var || B <= 0 || C <= 0 || N <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_bn_stats arguments"};
        ach::BnStatsParams p{z, mean, var, B, C, N};
        ACH_LAUNCH(ach::train_bn_stats_kernel, dim3(unsigned(C)), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_bn_relu_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta, float* y, int32_t B, int32_t C,
                          int32_t N, float eps, int32_t relu, void* stream) {
    return train_guard([&] {
        if (!z || !mean || !var || !gamma || !beta || !y || B <= 0 || C <= 0 || N <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_bn_relu_fwd arguments"};
        ach::BnReluFwdParams p{z, mean, var, gamma, beta, y, B, C, N, eps, relu};
        ACH_LAUNCH(ach::train_bn_relu_fwd_kernel, dim3(unsigned(ach::cdivl(long(B) * C * N, 256))), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_bn_relu_bwd(const float* z, const float* y, const float* dy, const float* mean, const float* var, const float* gamma, float* dgamma,
                          float* dbeta, float* dz, int32_t B, int32_t C, int32_t N, float eps, int32_t relu, void* stream) {
    return train_guard([&] {
        if (!z || !y || !dy || !mean || !var || !gamma || !dgamma || !dbeta || !dz || B <= 0 || C <= 0 || N <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_bn_relu_bwd arguments"};
        ach::BnReluBwdParams p{z, y, dy, mean, var, gamma, dgamma, dbeta, dz, B, C, N, eps, relu};
        ACH_LAUNCH(ach::train_bn_relu_bwd_reduce_kernel, dim3(unsigned(C)), dim3(256), static_cast<hipStream_t>(stream), p);
        ACH_LAUNCH(ach::train_bn_relu_bwd_apply_kernel, dim3(unsigned(ach::cdivl(long(B) * C * N, 256))), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}

int ach_train_dw3x3(const float* x, const float* w, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t flip, void* stream) {
    return train_guard([&] {
        if (!x || !w || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_dw3x3 arguments"};
        ach::DwTrainParams p{x, w, y, B, C, H, W, flip};
        ACH_LAUNCH(ach::train_dw3x3_kernel, dim3(unsigned(ach::cdivl(long(B) * C * H * W, 256))), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_dw3x3_wgrad(const float* x, const float* dz, float* dw, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
    return train_guard([&] {
        if (!x || !dz || !dw || B <= 0 || C <= 0 || H <= 0 || W <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_dw3x3_wgrad arguments"};
        ach::DwWgradParams p{x, dz, dw, B, C, H, W};
        ACH_LAUNCH(ach::train_dw3x3_wgrad_kernel, dim3(unsigned(C)), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}

int ach_train_max_points(const float* x, float* y, int32_t* idx, const float* dy, float* dx, int64_t rows, int32_t N, void* stream) {
    return train_guard([&] {                                 /* forward when dy == NULL, backward otherwise */
        if (rows <= 0 || N <= 0 || !idx) throw ach::AchError{ACH_ERR_INVALID, "bad train_max_points arguments"};
        if (!dy) {
            if (!x || !y) throw ach::AchError{ACH_ERR_INVALID, "bad train_max_points arguments"};
            ach::MaxPtsParams p{x, y, idx, long(rows), N};
            ACH_LAUNCH(ach::train_max_points_fwd_kernel, dim3(unsigned(ach::cdivl(long(rows), 4))), dim3(256), static_cast<hipStream_t>(stream), p);
        } else {
            if (!dx) throw ach::AchError{ACH_ERR_INVALID, "bad train_max_points arguments"};
            ach::MaxPtsBwdParams p{dy, idx, dx, long(rows), N};
            ACH_LAUNCH(ach::train_max_points_bwd_kernel, dim3(unsigned(ach::cdivl(long(rows) * N, 256))), dim3(256), static_cast<hipStream_t>(stream), p);
        }
    });
}
int ach_train_log_softmax(const float* z, float* y, const float* dy, float* dz, int32_t B, int32_t K, int32_t N, void* stream) {
    return train_guard([&] {                                 /* forward when dy == NULL (z -> y [B,N,K]); backward: y, dy -> dz [B,K,N] */
        if (B <= 0 || K <= 0 || N <= 0 || !y) throw ach::AchError{ACH_ERR_INVALID, "bad train_log_softmax arguments"};
        ach::LsmTrainParams p{z, y, dy, dz, B, K, N};
        const dim3 grid(unsigned(ach::cdivl(long(B) * N, 256)));
        if (!dy) { if (!z) throw ach::AchError{ACH_ERR_INVALID, "bad train_log_softmax arguments"}; ACH_LAUNCH(ach::train_log_softmax_fwd_kernel, grid, dim3(256), static_cast<hipStream_t>(stream), p); }
        else { if (!dz) throw ach::AchError{ACH_ERR_INVALID, "bad train_log_softmax arguments"}; ACH_LAUNCH(ach::train_log_softmax_bwd_kernel, grid, dim3(256), static_cast<hipStream_t>(stream), p); }
    });
}

int ach_tap_count(const ach_handle* h) { return (h && h->eng) ? int(h->eng->tap_order.size()) : 0; }
const char* ach_tap_name(const ach_handle* h, int i) {
    if (!h || !h->eng || i < 0 || i >= int(h->eng->tap_order.size())) return nullptr;
    return h->eng->tap_order[size_t(i)].c_str();
}
int ach_tap_shape(const ach_handle* hc, const char* name, int64_t shape[4], int32_t* ndim) {
    ach_handle* h = const_cast<ach_handle*>(hc);
    return guarded(h, [&] {
        if (!name || !shape || !ndim) throw ach::AchError{ACH_ERR_INVALID, "bad tap arguments"};
        std::vector<long> s = h->eng->tap_shape(name);
        *ndim = int32_t(s.size());
        for (size_t i = 0; i < s.size() && i < 4; ++i) shape[i] = s[i];
    });
}
int ach_read_tap(ach_handle* h, const char* name, float* host_out, size_t capacity_elems) {
    return guarded(h, [&] {
        if (!name || !host_out) throw ach::AchError{ACH_ERR_INVALID, "bad tap arguments"};
        h->eng->read_tap(name, host_out, capacity_elems);
    });
}
int ach_plan_launches(const ach_handle* h) { return (h && h->eng) ? int(h->eng->ops.size()) : 0; }
const char* ach_op_name(const ach_handle* h, int i) {
    if (!h || !h->eng || i < 0 || i >= int(h->eng->ops.size())) return nullptr;
    return h->eng->ops[size_t(i)].name.c_str();
}
double ach_op_bytes(const ach_handle* h, int i) {
    if (!h || !h->eng || i < 0 || i >= int(h->eng->ops.size())) return 0;
    return h->eng->ops[size_t(i)].bytes;
}
double ach_op_layout_bytes(const ach_handle* h, int i) {
    if (!h || !h->eng || i < 0 || i >= int(h->eng->ops.size())) return 0;
    return h->eng->ops[size_t(i)].layout_bytes;
}
int ach_op_stream(const ach_handle* h, int i) {
    if (!h || !h->eng || i < 0 || i >= int(h->eng->ops.size())) return -1;
    return h->eng->ops[size_t(i)].stream;
}
double ach_op_flops(const ach_handle* h, int i) {
    if (!h || !h->eng || i < 0 || i >= int(h->eng->ops.size())) return 0;
    return h->eng->ops[size_t(i)].flops;
}
int ach_forward_profiled(ach_handle* h, const void* image, const void* radar, const void* points, void* det3, void* det4,
                         void* det5, void* se_seg, void* lane_seg, void* pc_seg, void* stream, float* op_ms, size_t capacity) {
    return guarded(h, [&] {
        if (h->eng->ops.empty()) throw ach::AchError{ACH_ERR_INVALID, "ach_plan must precede ach_forward"};
        if (!image || !radar || !points || !det3 || !det4 || !det5 || !se_seg || !lane_seg || !pc_seg || !op_ms)
            throw ach::AchError{ACH_ERR_INVALID, "null pointer"};
        ach::IoPtrs& io = h->eng->io;
        io.image = image; io.radar = radar; io.points = points;
        io.det[0] = det3; io.det[1] = det4; io.det[2] = det5; io.se = se_seg; io.lane = lane_seg; io.pc = pc_seg;
        h->eng->run_profiled(static_cast<hipStream_t>(stream), op_ms, capacity);
    });
}
int ach_bench_gemm(ach_handle* h, int M, int K, int N, int act, int ln, int residual, int P, int iters, void* stream, float* ms) {
    return guarded(h, [&] {
        if (M <= 0 || K <= 0 || N <= 0 || iters <= 0 || !ms) throw ach::AchError{ACH_ERR_INVALID, "bad bench arguments"};
        *ms = h->eng->bench_gemm(M, K, N, act, ln, residual, P, iters, static_cast<hipStream_t>(stream));
    });
}
int ach_set_probe(ach_handle* h, int op_index) { return guarded(h, [&] { h->eng->set_probe(op_index); }); }
int ach_set_probe_range(ach_handle* h, int slot, int first, int last) { return guarded(h, [&] { h->eng->set_probe_range(slot, first, last); }); }
int ach_read_probe_slot(ach_handle* h, int slot, float* avg_ms, int* samples) {
    return guarded(h, [&] {
        if (!avg_ms || !samples) throw ach::AchError{ACH_ERR_INVALID, "null probe outputs"};
        h->eng->read_probe_slot(slot, avg_ms, samples);
    });
}
int ach_read_probe(ach_handle* h, float* avg_ms, int* samples) {
    return guarded(h, [&] {
        if (!avg_ms || !samples) throw ach::AchError{ACH_ERR_INVALID, "null pointer"};
        h->eng->read_probe(avg_ms, samples);
    });
}

}  // extern "C"
