// api.cpp — the extern "C" boundary of libachelous_hip.so (declared in include/achelous.h).
// No exception crosses it: every failure becomes a negative code + a message retrievable with ach_last_error().
#include <atomic>
#include <cstring>
#include <functional>
#include <string>

#include "engine.h"
#include <map>
#include <mutex>
#include <algorithm>
#include "k_detect.h"
#include "k_prepost.h"
#include "k_train.h"
#include "k_train2.h"
#include "k_train3.h"

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
        if (cfg->pc_seg != ACH_PCSEG_PN && cfg->pc_seg != ACH_PCSEG_PN2 && cfg->pc_seg != ACH_PCSEG_NONE && cfg->pc_seg != ACH_PCSEG_PN2_MSG)
            throw ach::AchError{ACH_ERR_UNSUPPORTED, "pc_seg must be 'pn', 'pn2', 'pn2_msg' or none (Achelous3T)"};
        if (cfg->num_det < 1 || cfg->num_det > 59 || cfg->num_seg < 1 || (cfg->pc_seg != ACH_PCSEG_NONE && (cfg->pc_classes < 1 || cfg->pc_channels < 3)))
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
        else if (std::string(key) == "head_fuse") h->eng->head_fuse = value != 0;
        else if (std::string(key) == "head_fuse_dbg") h->eng->head_fuse_dbg = value;
        else if (std::string(key) == "side_priority") h->eng->side_low_priority = value;
        else if (std::string(key) == "head_lds_pad") h->eng->head_lds_pad = value < 0 ? 0 : (value > 65536 ? 65536 : value);
        else if (std::string(key) == "head_stream") h->eng->head_stream = value != 0;
        else if (std::string(key) == "split_decoders") h->eng->split_decoders = value;
        else if (std::string(key) == "group_wpc") h->eng->group_wpc = value < 0 ? 0 : value;
        else if (std::string(key) == "group_max") h->eng->group_max = value < 0 ? 0 : value;
        else if (std::string(key) == "dec_fork") h->eng->dec_fork = value < 0 ? 0 : (value > 3 ? 3 : value);
        else if (std::string(key) == "radar_start") h->eng->radar_start = value;
        else if (std::string(key) == "pipeline") h->eng->pipeline = value != 0;
        else if (std::string(key) == "pool_strip") h->eng->pool_strip = value;
        else if (std::string(key) == "fused_mv2") h->eng->fuse_mv2 = value != 0;
        else if (std::string(key) == "dw_even") h->eng->dw_even = value != 0;
        else if (std::string(key) == "xca_mfma") h->eng->xca_mfma = value != 0;
        else if (std::string(key) == "xca_frame") h->eng->xca_frame = value;
        else if (std::string(key) == "xca_fold_mfma") h->eng->xca_fold_mfma = value != 0;
        else if (std::string(key) == "xca_slice") h->eng->xca_slice = value;
        else if (std::string(key) == "xca_front_waves") h->eng->xca_front_waves = value;
        else if (std::string(key) == "xca_back_waves") h->eng->xca_back_waves = value;
        else if (std::string(key) == "gemm_rows") h->eng->gemm_rows = (value == 2 || value == 4) ? value : 1;
        else if (std::string(key) == "radar_rows4") h->eng->radar_rows4 = value;
        else if (std::string(key) == "radar_skip") h->eng->radar_skip = value != 0;
        else if (std::string(key) == "radar_bg") h->eng->radar_bg = value != 0;
        else if (std::string(key) == "radar_pool_sparse") h->eng->radar_pool_sparse = value != 0;
        else if (std::string(key) == "radar_compact") h->eng->radar_compact = value != 0;
        else if (std::string(key) == "head_mfma") h->eng->head_mfma = value != 0;
        else if (std::string(key) == "head_rows") h->eng->head_rows = value;
        else if (std::string(key) == "xwait2_op") h->eng->dbg_xwait2_op = value;
        else if (std::string(key) == "gemm_blocks") h->eng->gemm_blocks = value;
        else if (std::string(key) == "sdta_fuse") h->eng->sdta_fuse = value;
        else if (std::string(key) == "level_chain") h->eng->level_chain = value != 0;
        else if (std::string(key) == "level_rows") h->eng->level_rows = value != 0;
        else if (std::string(key) == "mlp_band") h->eng->mlp_band = value;
        else if (std::string(key) == "mlp_band_run") h->eng->mlp_band_run = value != 0;
        else if (std::string(key) == "mlp_band_dbg") h->eng->mlp_band_dbg = value;
        else if (std::string(key) == "mlp_band_lean") h->eng->mlp_band_lean = value;
        else if (std::string(key) == "head_band") h->eng->head_band = value > 0 ? value : 40;
        else if (std::string(key) == "head_grid") h->eng->head_grid = value;
        else if (std::string(key) == "head_debug") h->eng->head_debug = value;
        else if (std::string(key) == "attn_mfma") h->eng->attn_mfma = value != 0;
        else if (std::string(key) == "ds_fuse") h->eng->ds_fuse = value != 0;
        else if (std::string(key) == "sa_fuse") h->eng->sa_fuse = value != 0;
        else if (std::string(key) == "pn2_fps_all") h->eng->pn2_fps_all = value != 0;
        else if (std::string(key) == "ghost_rb") h->eng->ghost_rb = value > 0 ? value : 5;
        else if (std::string(key) == "pc_chain") h->eng->pc_chain = value != 0;
        else if (std::string(key) == "mlp_split_hw") h->eng->mlp_split_hw = value;
        else if (std::string(key) == "radar_direct") h->eng->radar_direct = value != 0;
        else if (std::string(key) == "mv_stem") h->eng->mv_stem = value != 0;
        else if (std::string(key) == "csp_fuse") h->eng->csp_fuse = value < 0 ? 0 : (value > 2 ? 2 : value);
        else if (std::string(key) == "band_rows_s3") h->eng->band_rows_s3 = value;
        else if (std::string(key) == "spp_split") h->eng->spp_split = value;
        else if (std::string(key) == "ffn_rows2") h->eng->ffn_rows2 = value != 0;
        else if (std::string(key) == "csp_band") h->eng->csp_band = value > 0 ? value : 40;
        else if (std::string(key) == "ghost_fuse") h->eng->ghost_fuse = value != 0;
        else if (std::string(key) == "io_bf16") {
            if (value != 0 && h->eng->cfg.dtype != ACH_DTYPE_F16) throw ach::AchError{ACH_ERR_INVALID, "io_bf16 applies to the fp16-storage engine (ACH_DTYPE_F16) only"};
            h->eng->io_bf16 = value != 0;
        }
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
        if (!image || !radar || !det3 || !det4 || !det5 || !se_seg || !lane_seg || (h->eng->cfg.pc_seg != ACH_PCSEG_NONE && (!points || !pc_seg)))
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
        if (!image || !radar || !det3 || !det4 || !det5 || !se_seg || !lane_seg || (h->eng->cfg.pc_seg != ACH_PCSEG_NONE && (!points || !pc_seg)))
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

// ---- multi-GPU: the one exchange step of the path — an all-gather of the shards' fixed-size detection records over RCCL.
// RCCL is resolved at first use (dlopen of librccl.so; ACH_RCCL_LIBRARY overrides the name): the library itself does not link against it, so a
// single-GPU consumer needs no RCCL at all.  `comm` is an ncclComm_t the caller created with RCCL's own API (one rank per GPU).
#if !defined(ACH_HOSTEMU)
#include <dlfcn.h>
#endif
size_t ach_record_words(int32_t batch, int32_t max_det) { return (batch > 0 && max_det > 0) ? size_t(batch) * (size_t(max_det) * 8 + 1) : 0; }
int ach_all_gather_records(ach_handle* h, void* comm, const int32_t* send_record, int32_t* recv_records, int32_t batch, int32_t max_det, void* stream) {
    return guarded(h, [&] {
        if (!comm || !send_record || !recv_records || batch <= 0 || max_det <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad all-gather arguments"};
#if defined(ACH_HOSTEMU)
        throw ach::AchError{ACH_ERR_UNSUPPORTED, "the CPU emulation has no RCCL"};
#else
        typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
        static allgather_fn fn = nullptr;
        if (!fn) {
            const char* name = std::getenv("ACH_RCCL_LIBRARY");
            void* lib = dlopen(name ? name : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib && !name) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) throw ach::AchError{ACH_ERR_DEVICE, std::string("RCCL library not found: ") + dlerror()};
            fn = reinterpret_cast<allgather_fn>(dlsym(lib, "ncclAllGather"));
            if (!fn) throw ach::AchError{ACH_ERR_DEVICE, "ncclAllGather not found in the RCCL library"};
        }
        const int rc = fn(send_record, recv_records, ach_record_words(batch, max_det), /* ncclInt32 */ 2, comm, static_cast<hipStream_t>(stream));
        if (rc != 0) throw ach::AchError{ACH_ERR_DEVICE, "ncclAllGather failed with code " + std::to_string(rc)};
#endif
    });
}

// ---- one pass of the 8-bit PIL resample (k_prepost.h): stateless
int ach_resample_pass_u8(const uint8_t* src, uint8_t* dst, const int32_t* bounds, const int32_t* coeffs, int32_t ksize, int32_t h_in, int32_t w_in,
                         int32_t h_out, int32_t w_out, int32_t channels, int32_t vertical, int64_t src_pitch, int64_t dst_pitch, void* stream) {
    try {
        if (!src || !dst || !bounds || !coeffs || ksize <= 0 || h_in <= 0 || w_in <= 0 || h_out <= 0 || w_out <= 0 || channels <= 0 ||
            (vertical ? w_out != w_in : h_out != h_in))
            throw ach::AchError{ACH_ERR_INVALID, "bad resample_pass arguments"};
        ach::ResamplePassParams p{src, dst, bounds, coeffs, ksize, h_in, w_in, h_out, w_out, channels, vertical, long(src_pitch), long(dst_pitch)};
        ACH_LAUNCH(ach::resample_pass_kernel, dim3(unsigned(ach::cdivl(long(h_out) * w_out * channels, 256))), dim3(256), static_cast<hipStream_t>(stream), p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { g_create_error = std::string("kernel launch: ") + hipGetErrorString(e); return ACH_ERR_DEVICE; }
        return ACH_OK;
    } catch (const ach::AchError& e) { g_create_error = e.msg; return e.code; }
}

// ---- training-mode kernels (k_train.h): stateless, fp32, no handle
static int train_guard(const std::function<void()>& fn) {
    try { fn(); hipError_t e = hipGetLastError(); if (e != hipSuccess) { g_create_error = std::string("kernel launch: ") + hipGetErrorString(e); return ACH_ERR_DEVICE; } return ACH_OK; }
    catch (const ach::AchError& e) { g_create_error = e.msg; return e.code; }
    catch (const std::exception& e) { g_create_error = e.what(); return ACH_ERR_INVALID; }
}
// scratch for split reductions: one buffer per DEVICE, grown on demand.  Its users are ordered on the caller's stream; training runs on one.
static float* train_workspace(size_t bytes) {
    static std::mutex mu;
    static std::map<int, std::pair<float*, size_t>> pool;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto& ent = pool[dev];
    if (bytes > ent.second) {
        if (ent.first) { (void)hipDeviceSynchronize(); (void)hipFree(ent.first); ent.first = nullptr; ent.second = 0; }
        const size_t want = std::max(bytes, size_t(8) << 20);
        if (hipMalloc(reinterpret_cast<void**>(&ent.first), want) != hipSuccess) { ent.first = nullptr; throw ach::AchError{ACH_ERR_NOMEM, "training workspace"}; }
        ent.second = want;
    }
    return ent.first;
}
// slices of a per-channel reduction over `total` values: ~16 K values per workgroup, at most ~2048 workgroups in all
static int train_slices(long total, int C) {
    const long want = (total + 16383) / 16384, cap = std::max<long>(1, 2048 / std::max(C, 1));
    return int(std::max<long>(1, std::min<long>(std::min<long>(want, cap), 64)));
}
// element type of the training GEMMs' matrix-instruction operands, process-wide: 0 = fp32 (default), 1 = bf16 operands with fp32 accumulation (k_train.h)
static std::atomic<int> g_train_gemm_precision{0};
int ach_train_set_gemm_precision(int32_t precision) {
    if (precision != 0 && precision != 1) return ACH_ERR_INVALID;
    g_train_gemm_precision.store(precision, std::memory_order_relaxed);
    return ACH_OK;
}
int ach_train_get_gemm_precision(void) { return g_train_gemm_precision.load(std::memory_order_relaxed); }
int ach_train_gemm(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                   int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t trans_a, int32_t trans_b, int32_t batch, int32_t reduce_batch,
                   int32_t accumulate, void* stream) {
    return ach_train_gemm_p(A, B, C, bias, M, N, K, lda, ldb, ldc, stride_a, stride_b, stride_c, trans_a, trans_b, batch, reduce_batch, accumulate, -1, stream);
}
int ach_train_gemm_p(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                     int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t trans_a, int32_t trans_b, int32_t batch, int32_t reduce_batch,
                     int32_t accumulate, int32_t precision, void* stream) {
    return train_guard([&] {
        if (precision < -1 || precision > 1) throw ach::AchError{ACH_ERR_INVALID, "train_gemm precision must be -1 (the process-wide setting), 0 (fp32) or 1 (bf16 operands)"};
        if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_gemm arguments"};
        ach::TrainGemmParams p{A, B, C, bias, M, N, K, long(lda), long(ldb), long(ldc), long(stride_a), long(stride_b), long(stride_c), trans_a, trans_b, batch, reduce_batch, accumulate, 1, nullptr};
        // block tile: 32 rows / columns where the output has no more (k_train.h: the streaming shapes)
        const int tm = M <= 32 ? 32 : 64, tn = N <= 32 ? 32 : 64;
        dim3 grid(unsigned((N + tn - 1) / tn), unsigned((M + tm - 1) / tm), unsigned(reduce_batch ? 1 : batch));
        if (reduce_batch) {           // few output tiles, long reduction (weight gradients): split it over ~1024 workgroups, partial sums in a workspace
            const long T = long(batch) * ((K + ach::TRAIN_GEMM_TK - 1) / ach::TRAIN_GEMM_TK), tiles = long(grid.x) * grid.y;
            long split = std::min<long>(std::min<long>(2048 / tiles, T / 4), 2048);
            if (split > 1) {
                p.ksplit = int(split);
                p.ws = train_workspace(size_t(split) * M * N * sizeof(float));
                grid.z = unsigned(split);
            }
        }
        const hipStream_t st = static_cast<hipStream_t>(stream);
        const bool h16 = (precision >= 0 ? precision : g_train_gemm_precision.load(std::memory_order_relaxed)) == 1;
#define ACH_TG_LAUNCH(TM, TN) { if (h16) ACH_LAUNCH((ach::train_gemm_kernel<ach::bf16_t, TM, TN>), grid, dim3(256), st, p); else ACH_LAUNCH((ach::train_gemm_kernel<float, TM, TN>), grid, dim3(256), st, p); }
        if (tm == 32 && tn == 32) ACH_TG_LAUNCH(32, 32)
        else if (tm == 32) ACH_TG_LAUNCH(32, 64)
        else if (tn == 32) ACH_TG_LAUNCH(64, 32)
        else ACH_TG_LAUNCH(64, 64)
#undef ACH_TG_LAUNCH
        if (p.ksplit > 1) ACH_LAUNCH(ach::train_gemm_reduce_kernel, dim3(unsigned(ach::cdivl(long(M) * N, 16))), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_bn_stats(const float* z, float* mean, float* var, int32_t B, int32_t C, int32_t N, void* stream) {
    return train_guard([&] {
        if (!z || !mean || !var || B <= 0 || C <= 0 || N <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_bn_stats arguments"};
        const int S = train_slices(long(B) * N, C);
        if (S > 1) {
            ach::BnSliceParams q{z, train_workspace(size_t(2) * C * S * sizeof(float)), mean, var, B, C, N, S};
            ACH_LAUNCH(ach::train_bn_slice2_kernel, dim3(unsigned(C), unsigned(S)), dim3(256), static_cast<hipStream_t>(stream), q);          // both passes of a slice, the second out of the L2
            ACH_LAUNCH(ach::train_bn_slice2_finalize_kernel, dim3(unsigned((C + 255) / 256)), dim3(256), static_cast<hipStream_t>(stream), q);
            return;
        }
        ach::BnStatsParams p{z, mean, var, B, C, N};
        ACH_LAUNCH(ach::train_bn_stats_kernel, dim3(unsigned(C)), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_bn_running(const float* mean, const float* var, float* running_mean, float* running_var, int32_t C, float momentum, float unbias, void* stream) {
    return train_guard([&] {
        if (!mean || !var || !running_mean || !running_var || C <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_bn_running arguments"};
        ach::BnRunningParams p{mean, var, running_mean, running_var, C, momentum, unbias};
        ACH_LAUNCH(ach::train_bn_running_kernel, dim3(unsigned((C + 255) / 256)), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_bn_relu_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta, float* y, int32_t B, int32_t C,
                          int32_t N, float eps, int32_t relu, void* stream) {
    return train_guard([&] {
        if (!z || !mean || !var || !gamma || !beta || !y || B <= 0 || C <= 0 || N <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_bn_relu_fwd arguments"};
        ach::BnReluFwdParams p{z, mean, var, gamma, beta, y, B, C, N, eps, relu};
        const bool quad = (N & 3) == 0 && long(B) * C <= 65535 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;      // a plane per blockIdx.y, four positions per thread
        if (quad) ACH_LAUNCH(ach::train_bn_relu_fwd4_kernel, dim3(unsigned(ach::cdivl(long(N) / 4, 256)), unsigned(long(B) * C)), dim3(256), static_cast<hipStream_t>(stream), p);
        else
        ACH_LAUNCH(ach::train_bn_relu_fwd_kernel, dim3(unsigned(ach::cdivl(long(B) * C * N, 256))), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_bn_relu_bwd(const float* z, const float* y, const float* dy, const float* mean, const float* var, const float* gamma, float* dgamma,
                          float* dbeta, float* dz, int32_t B, int32_t C, int32_t N, float eps, int32_t relu, void* stream) {
    return train_guard([&] {
        if (!z || !y || !dy || !mean || !var || !gamma || !dgamma || !dbeta || !dz || B <= 0 || C <= 0 || N <= 0) throw ach::AchError{ACH_ERR_INVALID, "bad train_bn_relu_bwd arguments"};
        ach::BnReluBwdParams p{z, y, dy, mean, var, gamma, dgamma, dbeta, dz, B, C, N, eps, relu, 1, nullptr};
        p.S = train_slices(long(B) * N, C);
        if (p.S > 1) p.ws = train_workspace(size_t(2) * C * p.S * sizeof(float));
        ACH_LAUNCH(ach::train_bn_relu_bwd_reduce_kernel, dim3(unsigned(C), unsigned(p.S)), dim3(256), static_cast<hipStream_t>(stream), p);
        if (p.S > 1) ACH_LAUNCH(ach::train_bn_relu_bwd_finalize_kernel, dim3(unsigned((C + 255) / 256)), dim3(256), static_cast<hipStream_t>(stream), p);
        const bool quad = (N & 3) == 0 && long(B) * C <= 65535 &&
                          ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz)) & 15u) == 0;
        if (quad) ACH_LAUNCH(ach::train_bn_relu_bwd_apply4_kernel, dim3(unsigned(ach::cdivl(long(N) / 4, 256)), unsigned(long(B) * C)), dim3(256), static_cast<hipStream_t>(stream), p);
        else
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

// ---- training-mode primitives of k_train2.h
#define ACH_TRAIN_1D(kern, p, total) ACH_LAUNCH(kern, dim3(unsigned(ach::cdivl((total), 256))), dim3(256), static_cast<hipStream_t>(stream), p)
#define ACH_TRAIN_ROWS(kern, p, nblocks) ACH_LAUNCH(kern, dim3(unsigned(nblocks)), dim3(256), static_cast<hipStream_t>(stream), p)
static void train_need(bool ok, const char* what) { if (!ok) throw ach::AchError{ACH_ERR_INVALID, std::string("bad arguments: ") + what}; }
int ach_train_act(const float* x, const float* dy, float* out, int64_t n, int32_t kind, void* stream) {
    return train_guard([&] {
        train_need(x && out && n > 0 && kind >= 0 && kind <= 3, "ach_train_act");
        ach::TrainActParams p{x, dy, out, long(n), kind};
        const bool quad = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
        if (quad) ACH_TRAIN_1D(ach::train_act4_kernel, p, long(n) / 4);
        else ACH_TRAIN_1D(ach::train_act_kernel, p, long(n));
    });
}
int ach_train_mul(const float* a, const float* b, float* out, int64_t n, void* stream) {
    return train_guard([&] {
        train_need(a && b && out && n > 0, "ach_train_mul");
        ach::TrainMulParams p{a, b, out, long(n)};
        ACH_TRAIN_1D(ach::train_mul_kernel, p, long(n));
    });
}
int ach_train_layernorm(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int32_t C, int64_t inner,
                        float eps, void* stream) {
    return train_guard([&] {
        train_need(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0 && inner > 0, "ach_train_layernorm");
        ach::TrainLnParams p{x, gamma, beta, y, mean, rstd, long(rows), C, long(inner), eps};
        const bool quad = (inner & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15u) == 0;
        if (quad) ACH_TRAIN_1D(ach::train_ln_fwd4_kernel, p, long(rows) * (inner / 4));
        else ACH_TRAIN_1D(ach::train_ln_fwd_kernel, p, long(rows) * inner);
    });
}
int ach_train_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                            int64_t rows, int32_t C, int64_t inner, void* stream) {
    return train_guard([&] {
        train_need(x && dy && gamma && mean && rstd && dx && dgamma && dbeta && rows > 0 && C > 0 && inner > 0, "ach_train_layernorm_bwd");
        ach::TrainLnBwdParams p{x, dy, gamma, mean, rstd, dx, dgamma, dbeta, long(rows), C, long(inner), 1, nullptr};
        const bool quad = (inner & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15u) == 0;
        if (quad) ACH_TRAIN_1D(ach::train_ln_bwd_dx4_kernel, p, long(rows) * (inner / 4));
        else ACH_TRAIN_1D(ach::train_ln_bwd_dx_kernel, p, long(rows) * inner);
        p.S = train_slices(long(rows) * inner, C);
        if (p.S > 1) p.ws = train_workspace(size_t(2) * C * p.S * sizeof(float));
        ACH_LAUNCH(ach::train_ln_bwd_param_kernel, dim3(unsigned(C), unsigned(p.S)), dim3(256), static_cast<hipStream_t>(stream), p);
        if (p.S > 1) ACH_LAUNCH(ach::train_ln_bwd_param_finalize_kernel, dim3(unsigned((C + 255) / 256)), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_dwconv(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t k, int32_t flip, void* stream) {
    return train_guard([&] {
        train_need(x && w && y && B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && (k & 1), "ach_train_dwconv");
        ach::TrainDwParams p{x, w, bias, y, B, C, H, W, k, flip};
        const bool quad = (W & 3) == 0 && long(B) * C <= 65535 && (reinterpret_cast<uintptr_t>(y) & 15u) == 0;      // four outputs of a row per thread (k_train2.h)
        const dim3 gq(unsigned(ach::cdivl(long(H) * (W / 4), 256)), unsigned(long(B) * C)), bq(256);
        const hipStream_t st = static_cast<hipStream_t>(stream);
        if (quad && k == 3) ACH_LAUNCH(ach::train_dwconv4_kernel<3>, gq, bq, st, p);
        else if (quad && k == 5) ACH_LAUNCH(ach::train_dwconv4_kernel<5>, gq, bq, st, p);
        else if (quad && k == 7) ACH_LAUNCH(ach::train_dwconv4_kernel<7>, gq, bq, st, p);
        else if (quad && k == 9) ACH_LAUNCH(ach::train_dwconv4_kernel<9>, gq, bq, st, p);
        else
        if (k == 3) ACH_TRAIN_1D(ach::train_dwconv_kernel<3>, p, long(B) * C * H * W);
        else if (k == 5) ACH_TRAIN_1D(ach::train_dwconv_kernel<5>, p, long(B) * C * H * W);
        else if (k == 7) ACH_TRAIN_1D(ach::train_dwconv_kernel<7>, p, long(B) * C * H * W);
        else if (k == 9) ACH_TRAIN_1D(ach::train_dwconv_kernel<9>, p, long(B) * C * H * W);
        else ACH_TRAIN_1D(ach::train_dwconv_kernel<0>, p, long(B) * C * H * W);
    });
}
int ach_train_dwconv_wgrad(const float* x, const float* dz, float* dw, int32_t B, int32_t C, int32_t H, int32_t W, int32_t k, void* stream) {
    return train_guard([&] {
        train_need(x && dz && dw && B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && (k & 1), "ach_train_dwconv_wgrad");
        ach::TrainDwWgradParams p{x, dz, dw, B, C, H, W, k, 1, nullptr};
        const bool taps = (k == 3 || k == 5 || k == 7 || k == 9) && long(B) * H * W < (1L << 31);      // all taps of a channel in one pass (k_train2.h)
        p.S = train_slices(long(B) * H * W, taps ? C : C * k * k);
        if (p.S > 1) p.ws = train_workspace(size_t(C) * k * k * p.S * sizeof(float));
        const hipStream_t st = static_cast<hipStream_t>(stream);
        const dim3 gt(unsigned(C), unsigned(p.S));
        if (taps && k == 3) ACH_LAUNCH(ach::train_dwconv_wgrad_taps_kernel<3>, gt, dim3(256), st, p);
        else if (taps && k == 5) ACH_LAUNCH(ach::train_dwconv_wgrad_taps_kernel<5>, gt, dim3(256), st, p);
        else if (taps && k == 7) ACH_LAUNCH(ach::train_dwconv_wgrad_taps_kernel<7>, gt, dim3(256), st, p);
        else if (taps) ACH_LAUNCH(ach::train_dwconv_wgrad_taps_kernel<9>, gt, dim3(256), st, p);
        else
        ACH_LAUNCH(ach::train_dwconv_wgrad_kernel, dim3(unsigned(C * k * k), unsigned(p.S)), dim3(256), static_cast<hipStream_t>(stream), p);
        if (p.S > 1) ACH_LAUNCH(ach::train_dwconv_wgrad_finalize_kernel, dim3(unsigned((C * k * k + 255) / 256)), dim3(256), static_cast<hipStream_t>(stream), p);
    });
}
int ach_train_im2col(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw,
                     int32_t Ho, int32_t Wo, int32_t backward, void* stream) {
    return train_guard([&] {
        train_need(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && Ho > 0 && Wo > 0, "ach_train_im2col");
        ach::TrainColParams p{src, dst, B, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo};
        if (!backward) ACH_TRAIN_1D(ach::train_im2col_kernel, p, long(B) * C * kh * kw * Ho * Wo);
        else ACH_TRAIN_1D(ach::train_col2im_kernel, p, long(B) * C * H * W);
    });
}
int ach_train_softmax(const float* x, float* y, const float* dy, float* dx, int64_t rows, int32_t d, void* stream) {
    return train_guard([&] {
        train_need(y && rows > 0 && d > 0 && (dy ? dx != nullptr : x != nullptr), "ach_train_softmax");
        ach::TrainSoftmaxParams p{x, y, dy, dx, long(rows), d};
        ACH_TRAIN_1D(ach::train_softmax_kernel, p, long(rows));
    });
}
int ach_train_upsample2x(const float* src, float* dst, int64_t planes, int32_t h, int32_t w, int32_t backward, void* stream) {
    return train_guard([&] {
        train_need(src && dst && planes > 0 && h > 0 && w > 0, "ach_train_upsample2x");
        ach::TrainUpParams p{src, dst, long(planes), h, w, h > 1 ? float(h - 1) / float(2 * h - 1) : 0.f, w > 1 ? float(w - 1) / float(2 * w - 1) : 0.f};
        if (!backward) ACH_TRAIN_1D(ach::train_up2_fwd_kernel, p, long(planes) * 4 * h * w);
        else ACH_TRAIN_1D(ach::train_up2_bwd_kernel, p, long(planes) * h * w);
    });
}
int ach_train_maxpool(const float* x, float* y, int32_t* idx, const float* dy, float* dx, int64_t planes, int32_t H, int32_t W, int32_t k, void* stream) {
    return train_guard([&] {
        train_need(idx && planes > 0 && H > 0 && W > 0 && k > 0 && (k & 1) && (dy ? dx != nullptr : (x && y)), "ach_train_maxpool");
        ach::TrainPoolParams p{x, y, idx, dy, dx, long(planes), H, W, k};
        ACH_TRAIN_1D(ach::train_maxpool_kernel, p, long(planes) * H * W);
    });
}
int ach_train_avgpool3(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream) {
    return train_guard([&] {
        train_need(x && y && planes > 0 && H > 0 && W > 0, "ach_train_avgpool3");
        ach::TrainPoolParams p{x, y, nullptr, nullptr, nullptr, long(planes), H, W, 3};
        ACH_TRAIN_1D(ach::train_avgpool3_kernel, p, long(planes) * H * W);
    });
}
int ach_train_row_reduce(const float* a, const float* b, float* out, int64_t rows, int64_t n, float scale, void* stream) {
    return train_guard([&] {
        train_need(a && out && rows > 0 && n > 0, "ach_train_row_reduce");
        ach::TrainRowParams p{a, b, out, long(rows), long(n), 1, scale};
        ACH_TRAIN_ROWS(ach::train_row_reduce_kernel, p, rows);
    });
}
int ach_train_row_scale(const float* x, const float* s, float* out, int64_t rows, int64_t n, int64_t period, void* stream) {
    return train_guard([&] {
        train_need(s && out && rows > 0 && n > 0 && period > 0, "ach_train_row_scale");
        ach::TrainRowParams p{x, s, out, long(rows), long(n), long(period), 1.f};
        ACH_TRAIN_1D(ach::train_row_scale_kernel, p, long(rows) * n);
    });
}
int ach_train_col_reduce(const float* a, const float* b, float* out, int64_t rows, int64_t cols, float scale, void* stream) {
    return train_guard([&] {
        train_need(a && out && rows > 0 && cols > 0, "ach_train_col_reduce");
        ach::TrainRowParams p{a, b, out, long(rows), long(cols), 1, scale};
        ACH_TRAIN_ROWS(ach::train_col_reduce_kernel, p, cols);
    });
}
int ach_train_col_scale(const float* x, const float* g, float* out, int64_t rows, int64_t cols, void* stream) {
    return train_guard([&] {
        train_need(x && g && out && rows > 0 && cols > 0, "ach_train_col_scale");
        ach::TrainRowParams p{x, g, out, long(rows), long(cols), 1, 1.f};
        ACH_TRAIN_1D(ach::train_col_scale_kernel, p, long(rows) * cols);
    });
}
int ach_train_instnorm(const float* x, const float* dy, const float* gamma, const float* beta, float* y, float* mean, float* rstd, float* dx, float* dgamma_rows,
                       float* dbeta_rows, int64_t rows, int64_t n, int32_t C, float eps, void* stream) {
    return train_guard([&] {
        train_need(x && gamma && mean && rstd && rows > 0 && n > 0 && C > 0 && (dy ? (dx && dgamma_rows && dbeta_rows) : (y && beta)), "ach_train_instnorm");
        ach::TrainInParams p{x, dy, gamma, beta, y, mean, rstd, dx, dgamma_rows, dbeta_rows, long(rows), long(n), C, eps};
        ACH_TRAIN_ROWS(ach::train_instnorm_kernel, p, rows);
    });
}
int ach_train_l2norm(const float* x, float* y, float* norm, const float* dy, float* dx, int64_t rows, int64_t n, float eps, void* stream) {
    return train_guard([&] {
        train_need(x && norm && rows > 0 && n > 0 && (dy ? dx != nullptr : y != nullptr), "ach_train_l2norm");
        ach::TrainL2Params p{x, y, norm, dy, dx, long(rows), long(n), eps};
        ACH_TRAIN_ROWS(ach::train_l2norm_kernel, p, rows);
    });
}
int ach_train_deform_im2col(const float* x, const float* offset, const float* mask, float* col, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                            int32_t stride, int32_t pad, void* stream) {
    return train_guard([&] {
        train_need(x && offset && mask && col && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && stride > 0, "ach_train_deform_im2col");
        ach::TrainDeformParams p{x, offset, mask, col, nullptr, nullptr, nullptr, nullptr, B, C, H, W, Ho, Wo, stride, pad};
        ACH_TRAIN_1D(ach::train_deform_im2col_kernel, p, long(B) * C * 9 * Ho * Wo);
    });
}
int ach_train_deform_bwd(const float* x, const float* offset, const float* mask, const float* dcol, float* dx_zeroed, float* doffset, float* dmask, int32_t B,
                         int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t stride, int32_t pad, void* stream) {
    return train_guard([&] {
        train_need(x && offset && mask && dcol && doffset && dmask && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && stride > 0, "ach_train_deform_bwd");
        ach::TrainDeformParams p{x, offset, mask, nullptr, dcol, dx_zeroed, doffset, dmask, B, C, H, W, Ho, Wo, stride, pad};
        ACH_TRAIN_1D(ach::train_deform_bwd_coord_kernel, p, long(B) * 9 * Ho * Wo);
        // (dx_zeroed = NULL: the input needs no gradient — the first RCBlock reads the pooled radar map.  An LDS-combining form of this scatter — a workgroup per 16 x 16 tile of
        //  positions, LDS atomics, one L2 atomic per touched input pixel instead of 36 per position — was measured 4x SLOWER end to end (batch-32 step 64.5 -> 86.7 ms) and is not kept.)
        if (dx_zeroed) ACH_TRAIN_1D(ach::train_deform_bwd_input_kernel, p, long(B) * C * 9 * Ho * Wo);
    });
}
// ---- PointNet++ in training mode (k_train3.h): the inference engine's geometry kernels at fp32 + the two scatter-form adjoints
int ach_train_pn2_fps(const float* xyz, int32_t B, int32_t n, int32_t npoint, int32_t* idx, float* new_xyz, void* stream) {
    return train_guard([&] {
        train_need(xyz && idx && new_xyz && B > 0 && n > 0 && npoint > 0 && npoint <= n && n <= 64 * ach::PN2_FPS_MAX_PPT, "ach_train_pn2_fps");
        ach::launch_pn2_fps(ach::FpsParams{xyz, n, npoint, idx, new_xyz}, B, static_cast<hipStream_t>(stream));
    });
}
int ach_train_pn2_group(const float* xyz, const float* new_xyz, const float* feats, int32_t C, int32_t B, int32_t n, int32_t S, int32_t nsample, float radius2,
                        float* grouped, int32_t* group_idx, void* stream) {
    return train_guard([&] {
        train_need(xyz && new_xyz && (feats || C == 0) && grouped && group_idx && C >= 0 && B > 0 && n > 0 && S > 0 && nsample > 0 && nsample <= ach::PN2_MAX_NSAMPLE, "ach_train_pn2_group");
        ach::GroupParams q{xyz, new_xyz, feats, long(C), C, grouped, long(3 + C), group_idx, B, n, S, nsample, radius2, 1};
        ACH_LAUNCH(ach::pn2_group_kernel<float>, dim3(unsigned(ach::cdivl(long(B) * S, 4))), dim3(256), static_cast<hipStream_t>(stream), q);
    });
}
int ach_train_pn2_group_bwd(const int32_t* group_idx, const float* dgrouped, float* dfeats_zeroed, int32_t C, int32_t B, int32_t n, int32_t S, int32_t nsample, void* stream) {
    return train_guard([&] {
        train_need(group_idx && dgrouped && dfeats_zeroed && C > 0 && B > 0 && n > 0 && S > 0 && nsample > 0, "ach_train_pn2_group_bwd");
        ach::Pn2GroupBwdParams q{group_idx, dgrouped, long(3 + C), dfeats_zeroed, C, B, n, S, nsample};
        ACH_TRAIN_1D(ach::train_pn2_group_bwd_kernel, q, long(B) * S * nsample * C);
    });
}
int ach_train_pn2_interp(const float* xyz1, const float* xyz2, const float* skip, int32_t C1, const float* sparse, int32_t C2, float* out, float* dskip, float* dsparse_zeroed,
                         const float* dout, int32_t B, int32_t n, int32_t s, void* stream) {
    return train_guard([&] {
        train_need(xyz1 && xyz2 && C1 >= 0 && C2 > 0 && B > 0 && n > 0 && s >= 3 && s <= 64 * ach::PN2_INTERP_SPL, "ach_train_pn2_interp");
        const dim3 grid(unsigned(ach::cdivl(long(B) * n, 4))), block(256);
        if (!dout) {
            train_need((skip || C1 == 0) && sparse && out, "ach_train_pn2_interp (forward)");
            ach::InterpParams q{xyz1, xyz2, skip, long(C1), C1, sparse, long(C2), C2, out, long(C1 + C2), B, n, s};
            ACH_LAUNCH(ach::pn2_interp_kernel<float>, grid, block, static_cast<hipStream_t>(stream), q);
        } else {
            train_need((dskip || C1 == 0) && dsparse_zeroed, "ach_train_pn2_interp (backward)");
            ach::InterpParams q{xyz1, xyz2, nullptr, long(C1), C1, nullptr, long(C2), C2, nullptr, long(C1 + C2), B, n, s};
            ACH_LAUNCH(ach::train_pn2_interp_bwd_kernel, grid, block, static_cast<hipStream_t>(stream), q, dout, dskip, dsparse_zeroed);
        }
    });
}
#undef ACH_TRAIN_1D
#undef ACH_TRAIN_ROWS

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
int ach_count_saturated(ach_handle* h, void* stream, uint64_t* count) {
    return guarded(h, [&] {
        if (!count) throw ach::AchError{ACH_ERR_INVALID, "null count"};
        *count = uint64_t(h->eng->count_saturated(static_cast<hipStream_t>(stream)));
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
        if (!image || !radar || !det3 || !det4 || !det5 || !se_seg || !lane_seg || !op_ms || (h->eng->cfg.pc_seg != ACH_PCSEG_NONE && (!points || !pc_seg)))
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
