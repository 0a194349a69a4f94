// engine.h — host side of the engine: state-dict ingestion, constant folding, weight packing, activation arena
// and the launch plan.  Compiled by hipcc into libachelous_hip.so (and by g++ against tests/hostemu for the CPU
// emulation used by the unit tests).  See include/achelous.h for the C ABI and DESIGN.md for the data layout.
#pragma once
#include <cstdlib>
#include <cstdio>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/achelous.h"
#include "ach_platform.h"

namespace ach {

struct AchError {
    int code;
    std::string msg;
};

struct HostTensor {
    std::vector<float> data;
    std::vector<long> shape;
    long numel() const { long n = 1; for (long s : shape) n *= s; return n; }
};

struct TapInfo {
    const void* ptr = nullptr;
    int kind = 0;                 // 0: NHWC activation [B,H,W,C] (ld) -> NCHW ; 1: NCHW dense ; 2: rows [R, C] (ld) ; 3: user output (not readable)
    int B = 0, H = 0, W = 0, C = 0;
    long ld = 0;
    int is_f32 = 0;               // buffer holds fp32 regardless of the engine dtype
    int is_i32 = 0;               // buffer holds int32 (index selections); read back as exact floats
    int add_eye = 0;              // rows kind: add identity of this size (PointNet transforms are stored without +I)
};

struct Op {
    std::string name;
    std::function<void(hipStream_t)> fn;
    double bytes = 0;             // algorithmic HBM bytes of one launch: inputs read once + outputs written once + weights, REAL channels
    double layout_bytes = 0;      // the same count over the pixel pitches actually stored (channel padding included): what the launch must move
    double flops = 0;             // 2 * MACs of one launch (dense contractions only)
    int stream = 0;               // 0: caller's stream (image path) ; 1, 2: engine-owned side streams (radar / point branches)
    int wait_ev = -1, wait_ev2 = -1;   // join: wait for these events before the launch
    int signal_ev = -1;           // record this event after the launch
    bool xwait = false;           // pipelined forwards: first launch of its stream to overwrite buffers that the PREVIOUS forward's
                                  //   detection stream still reads -> waits for that forward's `xsignal` launch
    bool xsignal = false;         // pipelined forwards: last launch that reads buffers another stream rewrites in the next forward
    bool xwait2 = false;          // the same pair for the DECODERS' reads of the attention maps: since fusion + head moved to the radar stream
    bool xsignal2 = false;        //   (head_stream = 0) the fusion launch no longer follows the decoders in stream order, so `xsignal` alone would
                                  //   let the next forward's neck overwrite what this forward's decoders still read
    bool xwait3 = false;          // option dec_fork = 3 (the NECK on stream 2): the backbone launch that first writes a feature map the previous forward's neck
    bool xsignal3 = false;        //   reads (stage 1's output, a slice of the neck's concat buffer) waits for that neck's last reader of the backbone's maps
};

struct IoPtrs {
    const void* image = nullptr; const void* radar = nullptr; const void* points = nullptr;
    void* det[3] = {nullptr, nullptr, nullptr};
    void* se = nullptr; void* lane = nullptr; void* pc = nullptr;
};

class EngineBase {
public:
    explicit EngineBase(const ach_config& c) : cfg(c) {}
    virtual ~EngineBase();
    ach_config cfg;
    std::map<std::string, HostTensor> weights;
    std::string last_error;

    // device memory owned by the engine
    char* warena = nullptr; size_t warena_cap = 0, warena_used = 0;      // packed weights / constants
    char* aarena = nullptr; size_t aarena_cap = 0, aarena_used = 0;      // activations
    bool measuring = false;
    // option "graph": replay the plan as a hipGraph (captured per distinct set of I/O pointers).  OFF by default — measured on
    // MI355X the interleaved eager launches on three streams are as fast at batch 64 (4.3 ms both) and faster at batch 1
    // (1.36 ms vs 1.66 ms): HIP's graph executor serialises more of the three-branch DAG than the streams do.
    bool use_graph = false;
    bool io_bf16 = false;             // option "io_bf16" (fp16-storage engine only): the caller's input / output tensors are bf16; converted in the first / last kernels
    int csp_fuse = 2;                 // option "csp_fuse": 16-bit engines, CSP-Dual-FPN — 1: the full-resolution decoder level and the segmentation head as one row-walking launch (k_csphead.h) instead of five layer-wise ones; 2 (default): the 32-channel level below it as well; 0: layer-wise
    int band_rows_s3 = 0;             // option "band_rows_s3": rows per band of the stage-3 band kernel (0 = 5)
    int spp_split = 0;                // option "spp_split": workgroups per frame of the fused SPP launch (0 = auto: 2)
    bool ffn_rows2 = true;            // option "ffn_rows2": 16-bit engines — plain-row fused MLPs of 144..192 channels (MobileViT's feed-forward layers) with two 16-row tiles per wave (k_mlp.h ffn2_kernel); bit-identical
    int ffn_rows2_min = 0;            // ... for launches of at least this many rows (0: always — a frame must not depend on the batch it is in, and the four-waves-per-tile kernel sums in another order)
    int csp_band = 40;                // option "csp_band": rows per band of those launches (a band pays 2-4 rows of run-in)
    bool ghost_fuse = true;           // option "ghost_fuse": 16-bit engines — the neck's GhostModules (primary 1x1 + cheap depthwise 3x3) and the bottlenecks' shortcuts (depthwise 3x3 + 1x1 + residual) as band kernels (k_ghost.h): 3 launches per bottleneck instead of 6
    bool mv_stem = true;              // option "mv_stem": 16-bit engines — MobileViT's conv1 gathered from the NCHW image (no NHWC copy of the image; k_nhwc.h mvstem_kernel)
    bool radar_direct = true;         // option "radar_direct": 16-bit engines — the first RCBlock reads the caller's NCHW radar map itself (pool + residual): no NHWC copy, one launch fewer
    bool pc_chain = true;             // option "pc_chain": PointNet's conv3 + conv4 (256 -> 128 -> classes) as one two-layer chain launch (k_mlp.h)
    int ghost_rb = 5;                 // option "ghost_rb": rows per band of the neck's band kernels (upper bound; the LDS tiles may force fewer)
    bool pn2_fps_all = true;          // option "pn2_fps_all": PointNet++ — the four levels' farthest-point sampling as one launch (k_pn2.h pn2_fps_all_kernel; identical selections)
    bool sa_fuse = false;             // option "sa_fuse": ShuffleAttention's coefficient launch inside the apply launch (k_nhwc.h sa_apply_fused_kernel; bit-identical).  OFF: measured 38.3 k against 38.9 k frames/s
    bool ds_fuse = true;              // option "ds_fuse": the LayerNorm in front of EdgeNeXt's three 2x2 / stride-2 convs inside the conv's k-loop (k_gemm.h LNTAP; bit-level: same arithmetic, the
                                      // normalised pixel is rounded to the storage type once, as the separate launch's output was)
    bool multi_stream = true;         // option "streams": run the independent radar / point branches on side streams
    int gemm_rows = 1;                // option "gemm_rows": 16-row sub-tiles per wave (1 / 2 / 4) for GEMMs with K >= 1024 (the dense 3x3 convs of MobileViT)
    int xca_frame = 0;                // option "xca_frame" (16-bit engines, k_xcaframe.h): 2 = an XCA as TWO launches (qkv + Gram partials per token slice; softmax + fold + projection per 64 tokens), 1 = ONE launch with a workgroup per frame (measured slower), 0 = the four launches of rounds 1-5
    bool xca_fold_mfma = true;        // option "xca_fold_mfma" (16-bit engines): the finalize launch of the four-launch XCA folds softmax(attn) into the projection weights on the matrix cores, one workgroup per (frame, head) (k_xcaframe.h); 0 = round 5's fp32 VALU fold per (frame, head, 32 output channels)
    int xca_slice = 0;                // option "xca_slice": tokens per workgroup of the front kernel (0: 64, 128 on maps of 1024 tokens or more)
    int xca_front_waves = 0, xca_back_waves = 0;   // options: waves per workgroup of the two kernels (4 / 8 / 16; 0: by the number of work units)
    bool xca_mfma = true;             // option "xca_mfma": XCA Gram matrices on the matrix cores (xca_gram_mfma_kernel, k_xca.h); 0 = the VALU kernel
    bool dw_even = true;              // option "dw_even": SPLIT mlp_kernel deals depthwise tap ROWS, not whole k-steps, to its four waves (k_mlp.h)
    int radar_rows4 = 2;              // option "radar_rows4": a workgroup of rc_front owns four rows, one per wave (1: block 0 when radar_skip is on; 2: every
                                      // fused block — 29.1 k against 27.7 k frames/s: the per-workgroup weight staging and tables were a fifth of these kernels)
    bool radar_compact = true;        // option "radar_compact": first RCBlock — the active PIXELS of a row are compacted into dense tiles (k_conv3.h; needs radar_skip and four-row workgroups)
    bool radar_bg = true;             // option "radar_bg" (round 6): first RCBlock — its output map keeps the background value relu(bias) at every pixel that is not active, rc_front neither reads nor writes unoccupied pixels (k_conv3.h background mode; needs radar_skip, radar_compact, radar_direct)
    bool radar_pool_sparse = true;    // option "radar_pool_sparse" (round 6): first RCBlock's pool stores a pixel only where the pooled map is, or was after the previous forward, non-zero (k_radar.h; needs radar_skip's occupancy masks and radar_direct)
    bool radar_skip = true;           // option "radar_skip": first RCBlock — closed-form shortcut on 16-pixel segments whose neighbourhood of the radar map is empty (k_conv3.h)
    int gemm_blocks = 0;          // option "gemm_blocks": workgroups a GEMM launch aims for when the rows alone do not fill the chip (0 = 1024: four per CU)
    int sdta_fuse = 1;            // option "sdta_fuse": an SDTA encoder's conv cascade + tail copy + positional encoding as one launch (k_sdta.h): 1 = on maps of at most 20 x 20, 2 = every map that fits, 0 = never
    bool level_chain = true;      // option "level_chain": bf16 production plans — a decoder level's kernel also applies the next level's low-resolution conv pair (k_upchain.h)
    int dbg_xwait2_op = -1;       // option "xwait2_op" (debugging a pipelined race): the launch index that waits for the previous forward's decoders instead of the planned one
    int head_rows = 2;            // option "head_rows": bf16 — fused last decoder level + head as the row-walking kernel (k_dechead.h: no LDS, DPP row shifts, head 1x1 on MFMA); 0 = the LDS tile kernel (k_nhwc.h)
    bool level_rows = false;          // option "level_rows": the other two decoder levels through upghost_rows_kernel too (k_dechead.h).  OFF: their 32- / 48-channel NHWC rows are
                                      // write-bound, and the 16-column strips write them in 64-byte pieces: 3_to_2 25 -> 34 us, 2_to_1 48 -> 60 us, 32.6 k -> 31.6 k frames/s
    int head_band = 40;               // option "head_band": rows per band of the row-walking kernel (round 3, after the other kernels had settled: 40 rows 37.2 k frames/s, 80: 37.4 k, 160: 36.7 k, 320: 34.4 k;
                                      // round 4, once the kernel's waits were exact: 16 rows 39.5 k, 20: 39.7 k, 32: 39.9 k, 40: 40.0 k, 80: 39.7 k — two alternating passes each)
    bool head_mfma = false;           // option "head_mfma": bf16 — bilinear phase of the fused last decoder level on MFMA over a channel-planar t (k_nhwc.h)
    int head_grid = 0;                // option "head_grid": persistent workgroups of the MFMA head kernel (0 = UGM_GRID)
    int head_debug = 0;               // option "head_debug": timing experiments on the fused last decoder level (skips phases: results are wrong)
    bool attn_mfma = true;            // option "attn_mfma": MobileViT attention scores / P.V on MFMA (k_mvit.h) instead of one query per thread
    bool fuse_mv2 = true;             // option "fused_mv2": MobileViT's MV2 blocks (1x1 -> dw3x3 -> 1x1) as one launch (k_mv2.h)
    bool mlp_band_run = false;        // option "mlp_band_run" (default 0: measured level — 41.7 k against 41.9 k frames/s on EN-S0, plain loop of EN-S2 +3 %): consecutive band-kernel ConvEncoder blocks of a stage as ONE persistent launch with per-frame barriers between the blocks (k_mlpband.h mlp_band_run_kernel); bit-identical
    int mlp_band_lean = 0;            // option "mlp_band_lean" (experiment, round 6): the d = 96 band kernel with a 16-bit halo tile and no weight-prefetch register set (98 KB of LDS instead of 153)
    int mlp_band_dbg = 0;             // option "mlp_band_dbg": phase-kill timing experiments on the band kernel (results are wrong)
    int mlp_band = 1;                 // option "mlp_band" (2: also the large maps of stages 0 / 1): bf16 — ConvEncoder blocks on the small maps as the band kernel (k_mlpband.h: LDS halo tile, weights once per band); 0 = mlp_kernel's SPLIT mode
    bool fuse_mlp = true;             // option "fused_mlp": EdgeNeXt blocks as one kernel (k_mlp.h) instead of dw / pw1 / pw2 launches
    int group_wpc = 4096;             // option "group_wpc" (round 5; 0 = never): PointNet++ grouping with a WORKGROUP per centroid (four waves share a group's rows) on levels with at most this many centroids; bit-identical
    int group_max = 1024;             // option "group_max" (round 5; 0 = never): PointNet++'s shared-MLP + max-over-the-ball layers with a WAVE per ball (k_gemm.h gemm_groupmax_kernel) instead of a workgroup
                                      // per ball, for layers with at least this many balls (batch 64: 56 / 35 / 30 us -> 27 / 18 / 18 us for 16 384 / 4 096 / 1 024 balls; the last level's 256 balls: 36 -> 43 us, kept on the workgroup form); bit-identical
    int dec_fork = 1;                 // option "dec_fork" (pipelined plan): 0 = the decoders leave the caller's stream behind the shared ShuffleAttention stage (round 3), 1 = in front of it
                                      // (default, round 5: +0.9 %), 2 = as soon as p3 exists, 3 = the whole NECK on stream 2 (EdgeNeXt plans; the caller's stream carries the backbone only) — engine_impl.h neck()
    int split_decoders = 0;           // option "split_decoders": semantic decoder on side stream 3.  OFF: with the other branches at low
                                      // priority it no longer pays (23.8 k vs 22.9 k frames/s), and the process must stay at <= 4 ACTIVE
                                      // streams — caller + 2 here leaves one for a collective (RCCL) stream; a fifth costs 28 %
    bool head_stream = false;         // option "head_stream" = 1: radar and point branches share low-priority stream 1; fusion + detection head
                                      // (+ decode + NMS) get stream 2 at the caller's priority.  Was the default (+1.1 % in round 1) until the
                                      // first RCBlock's shortcut shortened the radar branch: with the head queued BEHIND the radar branch on
                                      // stream 1 the plan is now +1.2 % faster (26.1 k vs 25.8 k frames/s), and — caller + ONE side stream —
                                      // it leaves room for RCCL's stream: all-gather overhead at world size 1 9 % -> 1.5-4.5 %
    int head_lds_pad = 0;             // option "head_lds_pad": bytes of unused dynamic LDS per workgroup of the row-walking decoder head = an occupancy cap (160 KB / pad workgroups per CU)
    int side_low_priority = 3;        // option "side_priority" (with head_stream = 0): bit k set = side stream k+1 is created at the
                                      // lowest stream priority.  (2 = only the decoders' stream low is +0.8 % without a collective and -19 % WITH RCCL's stream
                                      // beside the engine's: 30.9 k against 37.7 k frames/s with the all-gather forced at world size 1 — both streams stay low.)
    bool pipeline = false;            // option "pipeline": consecutive forwards overlap.  The segmentation decoders move from the caller's stream
                                      // to side stream 2 (ahead of fusion + head), the caller's stream is done after the neck, and NOTHING is
                                      // joined at the end of ach_forward: the caller enqueues the next forward first and then calls ach_join
                                      // (at most two forwards in flight; see run_eager).  Results are identical; only the schedule differs.
    int radar_start = -2;             // option "radar_start": -1 = the radar branch starts with the forward; k = 0..3: only once backbone stage k is done (round 2: 1 measured
                                      // +1 %, the block-0 front kernel 0.58 -> 0.38 ms in-step); -2 (default) = 2 in the pipelined plan — with round 3's kernels, one box, alternating,
                                      // three runs: stage 1: 36.31 k frames/s, stage 2: 37.33 k, stage 3: 36.62 k — and 1 in the plain plan (35.77 k / 35.59 k / 33.57 k)
                                      // (released by event 0, with the point branch ahead of it on the same stream — the first RCBlocks are
                                      //  throughput-bound like backbone stages 0 / 1 and halve each other's speed when they overlap)
    int radar_start_eff() const { return radar_start == -2 ? ((pipeline && multi_stream) ? 2 : 1) : radar_start; }
    int pool_strip = 2;               // option "pool_strip": which RCBlock average pools use the 4-pixel strip kernel (engine_impl.h, rcnet)
    int head_fuse_dbg = 0;            // option "head_fuse_dbg": phase-kill timing experiments on the fused head layer (results are wrong)
    bool head_fuse = true;            // option "head_fuse": bf16, 64-wide towers — a head layer's depthwise 5x5 + pointwise conv as one launch (k_headdw.h); needs head_batch
    bool head_batch = true;           // option "head_batch": each detection-head layer as one launch for the three pyramid levels
    int point_on_head_stream = -1;    // option "point_stream2" (-1 auto / 0 / 1 / 3): the point branch opens stream 2 (ahead of fusion + head) instead of queueing behind the radar branch;
                                      // 3 (round 5): a stream of its own — the process's FOURTH active stream, i.e. none left for a collective's (section 4.10): single-GPU serving only
    bool stem_mfma = true;            // option "stem_mfma": the 4x4/s4 stem conv as an MFMA GEMM gathered from the NCHW image
    bool dw_tile = true;              // option "dw_tile": LDS-tiled depthwise kernel on the 10x10 maps
    bool fuse_rc = true;              // option "fused_rc": RCBlock conv + deformable sampling + contraction as one launch (k_conv3.h)
    bool row_conv = true;             // option "row_conv": narrow 3x3 convs through k_conv3.h instead of the generic implicit GEMM
    int mlp_split_hw = 1024;          // option "mlp_split_hw": with mlp_split = -1, maps of at most this many pixels run four waves per tile
    int mlp_split = -1;               // option "mlp_split": -1 auto (by tile count), 0 one tile per wave, 1 four waves per tile
    bool full_taps = false;           // option "full_taps": also materialise boundaries that production plans keep on-chip
    int batch = 0;
    std::vector<Op> ops;
    std::map<std::string, TapInfo> taps;
    std::vector<std::string> tap_order;
    IoPtrs io;

    void load(const ach_tensor_desc* t, size_t n);
    virtual void plan(int B) = 0;
    void run(hipStream_t s);            // graph replay when possible, else eager launches
    void run_eager(hipStream_t s);
    void join(hipStream_t s);           // pipelined mode: `s` waits for the oldest forward that has not been joined yet (no-op when none)
    long forwards_in_flight() const { return issued - joined; }
    // transient: extra launches enqueued behind the last op of the detection branch (stream 1) by ach_forward_detect
    std::function<void(hipStream_t)> detect_tail;
    void run_profiled(hipStream_t s, float* op_ms, size_t cap);
    // live probes: HIP events around ONE op of the plan (slot 0: set_probe) or around a RUN of ops first..last that sit on one stream
    // (slot 1: set_probe_range — the neck + decoder sub-path of the caller's stream) on every run() (bench.py's roofline leg)
    void set_probe(int op_index) { set_probe_range(0, op_index, op_index); }
    void read_probe(float* avg_ms, int* samples) { read_probe_slot(0, avg_ms, samples); }
    void set_probe_range(int slot, int first, int last);
    void read_probe_slot(int slot, float* avg_ms, int* samples);
    virtual void decode(int B, const void* d3, const void* d4, const void* d5, float* out, hipStream_t s) = 0;
    // fp16 storage: how many elements of the plan's activation tensors are saturated (+-65504: every kernel of the fp16 engine runs with MODE.FP16_OVFL, so an
    // overflowing conversion clamps instead of becoming infinity) or non-finite after the forward(s) enqueued on `s` so far; synchronises `s`.  0 for the other engines.
    virtual unsigned long long count_saturated(hipStream_t s) = 0;
    std::vector<std::pair<const void*, size_t>> t_regions;       // activation tensors held in the storage type: (device pointer, bytes), filled by plan()
    void* sat_dev = nullptr; size_t sat_dev_regions = 0;         // device copy of t_regions + the counter (count_saturated)
    void note_region(const void* p, size_t bytes) { if (!measuring) t_regions.emplace_back(p, bytes); }
    void nms(int B, const float* decoded, float conf, float iou, int max_det, float* rows, int* idx, int* count,
             void* workspace, hipStream_t s);
    size_t nms_workspace_bytes(int B) const;
    int num_anchors() const;
    void read_tap(const std::string& name, float* out, size_t cap);
    std::vector<long> tap_shape(const std::string& name) const;
    // pre / post-processing around the forward (k_prepost.h)
    virtual void preprocess_radar(int B, int C, const float* in, void* out, hipStream_t s) = 0;
    virtual void normalize_points(int B, int N, int D, const float* in, void* out, hipStream_t s) = 0;
    virtual void preprocess_image(int B, const unsigned char* in, void* out, hipStream_t s) = 0;
    virtual void seg_argmax(int B, int C, const void* seg, unsigned char* out, hipStream_t s) = 0;
    virtual void seg_resize_argmax(int B, int C, const void* seg, int out_h, int out_w, float* prob_ws, unsigned char* out, hipStream_t s) = 0;
    void correct_boxes(int B, int max_det, const float* rows, const int* count, int img_h, int img_w, int letterbox, float* out, hipStream_t s);
    float* prepost_scratch = nullptr; size_t prepost_scratch_bytes = 0;
    // micro-benchmark hook: time the MFMA GEMM kernel alone on scratch buffers (ms per launch)
    virtual float bench_gemm(int M, int K, int N, int act, int ln, int residual, int P, int iters, hipStream_t s) = 0;

protected:
    const HostTensor& W(const std::string& key) const;
    bool hasW(const std::string& key) const { return weights.count(key) != 0; }
    void* walloc(size_t bytes);
    void* aalloc(size_t bytes);
    float* up_f32(const std::vector<float>& v);
    void* up_raw(const void* src, size_t bytes);        // opaque constants (pre-packed MFMA fragments)
    bool skip_warned = false;
    void add_op(const std::string& name, std::function<void(hipStream_t)> fn, double bytes = 0, double flops = 0, double layout_bytes = -1) {
        if (measuring) return;
        // timing experiments only (profiles/scripts/skip_ops.sh): ACH_DEBUG_SKIP="substr,substr" turns the matching launches into no-ops
        // (events and stream order stay) to read off what a kernel group costs END TO END; the outputs are garbage then.  Compiled ONLY into
        // the variant libraries that profiles/scripts/build_variant.sh builds with -DACH_TIMING_HOOKS: the shipped library never reads the variable.
#if defined(ACH_TIMING_HOOKS)
        if (const char* skip = std::getenv("ACH_DEBUG_SKIP")) {
            std::string all(skip);
            for (size_t a = 0; a < all.size();) {
                size_t b = all.find(',', a); if (b == std::string::npos) b = all.size();
                if (b > a && name.find(all.substr(a, b - a)) != std::string::npos) {
                    fn = [](hipStream_t) {};
                    if (!skip_warned) { skip_warned = true; std::fprintf(stderr, "achelous: ACH_DEBUG_SKIP is set: matching launches are no-ops, outputs are GARBAGE (timing experiments only)\n"); }
                    break;
                }
                a = b + 1;
            }
        }
        // ACH_DEBUG_ONLY="substr,substr": the complement — every launch that matches NONE of the substrings becomes a no-op.  tests/test_gpu_coresidency.py builds
        // its AGGRESSOR engines this way (a plan reduced to the row-walking heads, or the radar front kernels, or the band kernels, looping on its own stream
        // beside a victim forward of the SHIPPED library); an aggressor's outputs are never looked at.
        if (const char* only = std::getenv("ACH_DEBUG_ONLY")) {
            std::string all(only);
            bool keep = false;
            for (size_t a = 0; a < all.size() && !keep;) {
                size_t b = all.find(',', a); if (b == std::string::npos) b = all.size();
                if (b > a && name.find(all.substr(a, b - a)) != std::string::npos) keep = true;
                a = b + 1;
            }
            if (!keep && !all.empty()) fn = [](hipStream_t) {};
        }
#endif
        Op op{name, std::move(fn), bytes, layout_bytes < 0 ? bytes : layout_bytes, flops};
        op.stream = cur_stream;
        op.wait_ev = pending_wait; op.wait_ev2 = pending_wait2;
        pending_wait = -1; pending_wait2 = -1;
        op.xwait = pending_xwait; pending_xwait = false;
        op.xwait2 = pending_xwait2; pending_xwait2 = false;
        op.xwait3 = pending_xwait3; pending_xwait3 = false;
        ops.push_back(std::move(op));
    }
    // branch bookkeeping while the plan is built
    int cur_stream = 0, pending_wait = -1, pending_wait2 = -1;
    int detect_stream = 1;            // the stream the detection head ends on (decode + NMS of ach_forward_detect follow it there)
    void signal_after_last(int ev) { if (!measuring && !ops.empty()) ops.back().signal_ev = ev; }
    void wait_before_next(int ev) { pending_wait = ev; }
    void wait_before_next2(int ev) { pending_wait2 = ev; }
    static constexpr int kSideStreams = 3, kJoinEvents = 4;
    hipStream_t side_stream[kSideStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kJoinEvents] = {nullptr, nullptr, nullptr, nullptr}, ev_end[kSideStreams] = {nullptr, nullptr, nullptr};
    bool streams_ready = false;
    void ensure_streams();
    // pipelined mode: two alternating event sets (forward k uses set k & 1)
    hipEvent_t ev_x2[2] = {nullptr, nullptr};
    bool x2_recorded[2] = {false, false};
    hipEvent_t ev_x[2] = {nullptr, nullptr}, ev_done[kSideStreams][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    bool done_used[2][kSideStreams] = {{false, false, false}, {false, false, false}};
    long issued = 0, joined = 0;
    void mark_xwait_next() { pending_xwait = true; }
    void mark_xsignal_last() { if (!measuring && !ops.empty()) ops.back().xsignal = true; }
    bool pending_xwait = false, pending_xwait2 = false;
    void mark_xwait2_next() { pending_xwait2 = true; }
    void mark_xsignal2_last() { if (!measuring && !ops.empty()) ops.back().xsignal2 = true; }
    bool pending_xwait3 = false;
    void mark_xwait3_next() { pending_xwait3 = true; }
    void mark_xsignal3_last() { if (!measuring && !ops.empty()) ops.back().xsignal3 = true; }
    hipEvent_t ev_x3[2] = {nullptr, nullptr};
    bool x3_recorded[2] = {false, false};
#if !defined(ACH_HOSTEMU)
    struct GraphEntry { IoPtrs io; hipGraphExec_t exec; unsigned long stamp; };
    std::vector<GraphEntry> graphs;
    hipStream_t capture_stream = nullptr;
    unsigned long graph_clock = 0;
    bool graph_failed = false;
    void drop_graphs();
#endif
    static constexpr int kProbeEvents = 512, kProbeSlots = 3;
    struct Probe { int first = -1, last = -1; std::vector<hipEvent_t> ev0, ev1; long count = 0; };
    Probe probes[kProbeSlots];
    bool probing() const { for (const auto& pr : probes) if (pr.first >= 0) return true; return false; }
    void add_tap(const std::string& name, const TapInfo& t) { if (!measuring) { if (!taps.count(name)) tap_order.push_back(name); taps[name] = t; } }
    void reset_plan();
};

EngineBase* make_engine(const ach_config& cfg);

}  // namespace ach
