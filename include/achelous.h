/* achelous.h — C ABI of the MI355X-native Achelous forward engine (libachelous_hip.so).
 *
 * The reference (GuanRunwei/Achelous @ 2024-08-07) is 100 % Python: there is no FFI, plugin registry or native
 * operator table in it.  Its "operator API" for the hot path is one nn.Module and two free functions, and each
 * entry point below replaces exactly one of them (file:line relative to the reference tree):
 *
 *   ach_create / ach_destroy      <- nets/Achelous.py:26-47      Achelous.__init__ (ctor arguments -> ach_config)
 *   ach_load_weights              <- achelous.py:171             net.load_state_dict(torch.load(path))  (reference key names)
 *   ach_plan                      <- (implicit in eager PyTorch)  shape specialisation for one batch size
 *   ach_forward                   <- nets/Achelous.py:49-53      Achelous.forward(x, x_radar, x_point_clouds)
 *   ach_decode                    <- utils/utils_bbox.py:33-85   decode_outputs(outputs, input_shape)
 *   ach_nms                       <- utils/utils_bbox.py:87-132  non_max_suppression(...) up to the host-side un-letterbox
 *   ach_forward_detect            <- achelous.py:196-262         the three calls above as the reference's detect_image chains them
 *   ach_all_gather_records        <- achelous.py:176             nn.DataParallel's gather of the replicas' outputs (here: RCCL all-gather of detection records)
 *   ach_read_tap / ach_tap_*      <- (test hook) intermediate tensors at the SURVEY.md §8(a) boundaries
 *
 * Conventions: plain pointers and sizes only (no torch / HIP C++ types in the signatures; `stream` is a
 * hipStream_t passed as void*).  Every function returns 0 on success and a negative code on failure and never
 * throws; ach_last_error() returns a message for the most recent failure on that handle (or a global one when the
 * handle itself could not be created).  The caller owns inputs and outputs (device pointers, dtype = config dtype,
 * layouts exactly those of the reference: NCHW images / maps, [B, C, N] point clouds, [B, N, classes] point
 * log-probabilities).  The engine owns its packed weights and activation arena (allocated in ach_load_weights /
 * ach_plan; nothing is allocated inside ach_forward, which is asynchronous on `stream` and performs no host sync).
 * One handle per (device, stream); calls on one handle must not overlap.
 */
#ifndef ACHELOUS_H_
#define ACHELOUS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ach_handle ach_handle;

enum { ACH_DTYPE_F32 = 0, ACH_DTYPE_BF16 = 1,
       ACH_DTYPE_F16 = 2    /* fp16 activations and MFMA operands, fp32 accumulation: the type the reference's mixed-precision mode computes in
                             * (utils/utils_fit.py:120-121, train.py:37 --fp16).  Inputs / outputs are fp16, or bf16 with option "io_bf16" */ };
enum { ACH_BACKBONE_EDGENEXT = 0, ACH_BACKBONE_MOBILEVIT = 1 };
enum { ACH_PHI_S0 = 0, ACH_PHI_S1 = 1, ACH_PHI_S2 = 2 };

enum {
    ACH_OK = 0,
    ACH_ERR_INVALID = -1,       /* bad argument / shape / state */
    ACH_ERR_UNSUPPORTED = -2,   /* configuration outside the built scope (mirrors NotImplementedError) */
    ACH_ERR_MISSING_KEY = -3,   /* state-dict key absent or of the wrong shape */
    ACH_ERR_DEVICE = -4,        /* HIP runtime error */
    ACH_ERR_NOMEM = -5
};

enum { ACH_NECK_GDF = 0, ACH_NECK_CDF = 1 };     /* Ghost-Dual-FPN (neck/ghostdualfpn.py) / CSP-Dual-FPN (neck/cspdualfpn.py) */
/* point-cloud branch: PointNet (nets/pointcloudseg/pointnet2/pointnet_sem_seg.py) / PointNet++.  The reference snapshot has no
 * PointNet++ code (nets/Achelous.py:31-32 builds only 'pn'): ACH_PCSEG_PN2 runs OUR OWN specification of that branch
 * (state-dict keys pc_seg_model.sa{1-4} / fp{4-1} / conv1 / bn1 / conv2); num_points must be a multiple
 * of 128 and at most 1024. */
enum { ACH_PCSEG_PN = 0, ACH_PCSEG_PN2 = 1,
       ACH_PCSEG_NONE = 2,  /* nets/Achelous.py:56-76 (Achelous3T): no point-cloud stream; `points` / `pc_seg` arguments are ignored (may be NULL) */
       ACH_PCSEG_PN2_MSG = 3 /* round 6: the MULTI-scale-grouping PointNet++ (spec.py::PN2_MSG: two radii and two shared-MLP stacks per level, keys sa{k}.conv_blocks.<scale>.<layer>):
                                the variant the reference's only PointNet++ datum (README.md:81,83: ~2.0 M parameters) points at; own specification, parity unpinned */ };

typedef struct ach_config {
    int32_t num_det;        /* detection classes           (Achelous.__init__ num_det)      */
    int32_t num_seg;        /* semantic classes            (num_seg)                        */
    int32_t phi;            /* ACH_PHI_*                   (phi)                            */
    int32_t backbone;       /* ACH_BACKBONE_*              (backbone in {'en','mv'})        */
    int32_t resolution;     /* square input size           (resolution)                     */
    int32_t pc_channels;    /* point features              (pc_channels)                    */
    int32_t pc_classes;     /* point classes               (pc_classes)                     */
    int32_t num_points;     /* points per cloud N                                           */
    int32_t nano_head;      /* 1: 64-channel head          (nano_head)                      */
    int32_t spp;            /* 1: SPP, 0: SPPF             (spp)                            */
    int32_t dtype;          /* ACH_DTYPE_*: storage type of activations, inputs and outputs */
    int32_t neck;           /* ACH_NECK_*                  (neck in {'gdf','cdf'})           */
    int32_t pc_seg;         /* ACH_PCSEG_*                 (pc_seg in {'pn','pn2'}; NONE = Achelous3T) */
} ach_config;

/* one entry of a reference-keyed state_dict; `data` is HOST memory, fp32, contiguous, reference shape */
typedef struct ach_tensor_desc {
    const char* name;
    const float* data;
    int32_t ndim;
    int64_t shape[4];
} ach_tensor_desc;

int ach_create(const ach_config* cfg, ach_handle** out);
void ach_destroy(ach_handle* h);
const char* ach_last_error(const ach_handle* h);

/* Copies, folds (eval-mode BatchNorm, LayerNorm affine, layer scale, constant positional encoding) and repacks
 * the weights into kernel-native layouts on the device.  Integer buffers (num_batches_tracked) may be omitted. */
int ach_load_weights(ach_handle* h, const ach_tensor_desc* tensors, size_t n);

/* Options, set before ach_plan.  "full_taps" = 1: also write to HBM the SURVEY §8(a) boundaries that production plans keep
 * on-chip (the 32-channel full-resolution decoder tensors), so that parity tests can read them back.
 * "streams" = 0: launch everything on the caller's stream (default 1: the independent radar and point branches run on two
 * engine-owned side streams, forked from / joined into the caller's stream with events).
 * "graph" = 1: capture the plan into a hipGraph per distinct set of I/O pointers and replay it (default 0: measured no faster
 * than the interleaved eager launches on the side streams).
 * Kernel-selection switches, all default 1, kept so that each fused kernel can be A/B-measured against the layer-wise path it
 * replaced (results agree to rounding): "fused_mlp" (EdgeNeXt blocks and decoder conv pairs through k_mlp.h), "mlp_split"
 * (-1 auto / 0 / 1: four waves per pixel tile in k_mlp.h), "row_conv" (offset/modulator convs through k_conv3.h),
 * "split_decoders" (semantic decoder on its own stream), "fused_rc" (RCBlock conv + deformable sampling + contraction as one
 * kernel), "dw_tile" (LDS-tiled depthwise kernel on 10x10 maps), "head_batch" (detection-head layers batched over the levels),
 * "head_stream" (1: radar + point branches share low-priority stream 1, fusion + head + NMS run on stream 2 at the caller's priority; default 0: they queue behind the radar branch on stream 1),
 * "side_priority" (with head_stream = 0: bit mask of the side streams created at the lowest stream priority),
 * "point_stream2" (-1 auto / 0 / 1: the point branch opens stream 2 ahead of fusion + head; auto = PointNet++ only),
 * "stem_mfma" (the 4x4/s4 stem conv as an MFMA GEMM gathered from the NCHW image; 0: scalar-FMA kernel),
 * "radar_skip" (closed-form shortcut for empty 16-pixel segments of the first RCBlock, bit-identical), "radar_rows4" (0 / 1 / 2: a
 * workgroup owns four rows in the first / in all fused RCBlocks and the row-walking conv), "radar_start" (the radar branch is released
 * after backbone stage k = 0..3, -1 = at once; default -2 = stage 2 in the pipelined plan, stage 1 in the plain plan), "dw_even" (even deal of depthwise tap rows over the four SPLIT waves, d = 144), "xca_mfma" (XCA Gram
 * matrices on the matrix cores; 0: VALU kernel), "head_mfma" (default 0: bilinear phase of the fused segmentation head on MFMA —
 * measured slower), "gemm_rows" (default 1: 16-row sub-tiles per wave for GEMMs with K >= 1024 — 2 / 4 measured slower), "gemm_blocks" (workgroups a GEMM
 * launch aims for when its rows alone cannot fill the chip; 0 = 1024),
 * round 4: "io_bf16" (ACH_DTYPE_F16 only: the caller's input / output tensors are bf16, converted in the first / last kernels);
 * "ghost_fuse" (default 1: the neck's Ghost bottlenecks, shortcuts, Upsample modules and the SPP block as band kernels — k_ghost.h; bit-identical), "ghost_rb" (rows
 * per band of those kernels, default 5), "ds_fuse" (default 1: the channels-first LayerNorm of a downsample layer inside its 2x2 / stride-2 conv), "sa_fuse" (default 0:
 * ShuffleAttention's coefficient launch folded into the apply launch — measured slower), "pn2_fps_all" (default 1: PointNet++'s four farthest-point samplings as one
 * launch), "pc_chain" (default 1: PointNet's conv3 + conv4 as one two-layer chain launch), "mv_stem" (default 1, 16-bit engines: MobileViT's conv1 gathered straight
 * from the NCHW image instead of an NHWC copy + implicit GEMM), "radar_direct" (default 1, 16-bit engines: the first RCBlock pools and adds its residual straight
 * from the NCHW radar map; bit-identical), "mlp_split_hw" (default 1024: maps of at most this many pixels run the fused blocks with four waves per 16-pixel tile);
 * round 3 (16-bit engines): "head_rows" (2: last decoder level + segmentation head as the row-walking two-columns-per-lane kernel, 1: one
 * column, 0: the LDS tile kernel), "head_band" (rows per workgroup band of that kernel, default 40), "mlp_band" (1: EdgeNeXt blocks of
 * the instantiated shapes — d = 96 / 144 on maps up to 20 wide, d = 176 up to 10 wide — as the band kernel, 2: also stages 0 / 1, 0: never), "head_fuse" (a detection-head layer's depthwise + pointwise convs of both
 * towers and all levels as one launch), "radar_compact" (first RCBlock: per-pixel activity, active pixels compacted into dense tiles;
 * bit-identical), "level_chain" (a decoder level's kernel also applies the next level's low-resolution conv pair; bit-identical),
 * "sdta_fuse" (1: an SDTA encoder's conv cascade + tail copy + positional encoding as one launch on maps up to 20 x 20, 2: every map,
 * 0: never; bit-identical), "level_rows" (default 0: decoder levels as row-walking kernels — measured slower);  debugging only:
 * "xwait2_op" (pipelined mode: the launch index that waits for the previous forward's decoders), "head_debug" / "mlp_band_dbg" /
 * "head_fuse_dbg" (phase-kill timing variants, WRONG results);
 * round 5 (16-bit engines): "csp_fuse" (CSP-Dual-FPN decoders — 2, default: the two 32-channel levels (160 x 160 and full resolution) as row-walking launches
 * (k_csphead.h), the last one with the segmentation head in it; 1: the last level + head only; 0: layer-wise), "csp_band" (rows per band of those launches, default
 * 40), "head_lds_pad" (default 0: bytes of unused dynamic LDS per workgroup of the Ghost-FPN row-walking head = an occupancy cap; every cap measured slower);
 * "ffn_rows2" (default 1: MobileViT's feed-forward layers on the large maps with two 16-row tiles per wave; bit-identical), "mlp_band_run" (default 0: a stage's
 * band-kernel ConvEncoder blocks as one persistent launch with per-frame barriers; bit-identical, measured no faster), "dec_fork" (pipelined plan: where the
 * segmentation decoders leave the caller's stream — 1, default: in front of the shared ShuffleAttention stage (+0.9 %), 0: behind it (round 3), 2: as soon as the
 * neck's p3 exists, 3: the whole neck on stream 2, ordered against the next forward's backbone by a third cross-forward event (EdgeNeXt plans; level with 1);
 * bit-identical), "group_max" (default 1024; 0 = never: PointNet++'s shared-MLP + max-over-the-ball layers with a wave per ball for layers of at least this many balls;
 * bit-identical), "group_wpc" (default 4096; 0 = never: PointNet++ grouping with a workgroup per centroid on levels with at most this many centroids; bit-identical), "point_stream2" = 3 (the point branch on a stream of its own — the process's fourth: measured -1 %, and it leaves no stream for a collective);
 * round 6 (16-bit engines): "xca_fold_mfma" (default 1: the XCA finalize launch folds softmax(attn) into the projection weights on the matrix cores, one workgroup per (frame, head);
 * 0: round 5's fp32 VALU fold), "xca_frame" (default 0: four launches per attention; 2: two launches — qkv + Gram partials per token slice, softmax + fold + projection per 64 tokens;
 * 1: one launch with a workgroup per frame; both measured slower) with "xca_slice", "xca_front_waves", "xca_back_waves"; "radar_pool_sparse" (default 1: the first
 * RCBlock's pool stores a pixel only where the pooled map is, or was after the previous forward, non-zero) and "radar_bg" (default 1: the block's output map keeps relu(bias) at every
 * unoccupied pixel; rc_front neither reads nor writes such pixels) — both bit-identical, both carry masks from one forward to the next inside the plan's arena;
 * "mlp_band_lean" (default 0: the d = 96 band kernel with a 16-bit halo tile, 98 KB of LDS: measured neutral);
 * "pipeline" (see ach_join).  DESIGN.md §4 and profiles/NOTES_r0*.md have the measurement behind every default. */
int ach_set_option(ach_handle* h, const char* key, int32_t value);

/* Builds the launch plan and the activation arena for batch size B (re-plan to change B). */
int ach_plan(ach_handle* h, int32_t batch);
size_t ach_arena_bytes(const ach_handle* h);

/* image [B,3,R,R], radar [B,3,R,R], points [B,pc_channels,N]  ->
 * det3 [B,5+num_det,R/8,R/8], det4 [..,R/16,R/16], det5 [..,R/32,R/32], se_seg [B,num_seg,R,R],
 * lane_seg [B,2,R,R], pc_seg [B,N,pc_classes] (log-probabilities).  All device pointers of the config dtype. */
int ach_forward(ach_handle* h, const void* image, const void* radar, const void* points,
                void* det3, void* det4, void* det5, void* se_seg, void* lane_seg, void* pc_seg, void* stream);

/* ach_forward + ach_decode + ach_nms in one call (achelous.py:246-262 runs net -> decode_outputs -> non_max_suppression back to
 * back).  Same outputs as the three separate calls, bit for bit; the difference is scheduling: decode and NMS are enqueued on the
 * engine's detection-branch stream right behind the detection head, so they overlap with the segmentation decoders instead of
 * running alone at the end of the step.  `decoded` [B,A,5+num_det] fp32 and `workspace` (>= ach_nms_workspace_bytes) are scratch
 * owned by the caller. */
int ach_forward_detect(ach_handle* h, const void* image, const void* radar, const void* points,
                       void* det3, void* det4, void* det5, void* se_seg, void* lane_seg, void* pc_seg,
                       float* decoded, float conf_thres, float nms_thres, int32_t max_det,
                       float* out_rows, int32_t* out_idx, int32_t* out_count, void* workspace, void* stream);

/* Pipelined serving (ach_set_option(h, "pipeline", 1) before ach_plan): ach_forward / ach_forward_detect return with the work of
 * the engine's side streams NOT yet joined into `stream`, so that the caller can enqueue the next forward first; ach_join makes
 * `stream` wait for the oldest un-joined forward (its outputs are complete in stream order after that).  At most two forwards may
 * be un-joined (a third call joins the oldest itself); inputs, outputs and the detect workspace of a forward must stay allocated
 * and untouched until it has been joined.  Results are bit-identical to the plain mode.  The reference's counterpart is the
 * python loop around `net(...)` in achelous.py:246-262 / utils/callbacks.py:184, which serialises frames. */
int ach_join(ach_handle* h, void* stream);
int ach_forwards_in_flight(const ach_handle* h);

/* det maps (config dtype) -> decoded [B, A, 5+num_det] fp32, A = (R/8)^2 + (R/16)^2 + (R/32)^2 */
int ach_decode(ach_handle* h, int32_t batch, const void* det3, const void* det4, const void* det5,
               float* decoded, void* stream);

/* decoded [B,A,5+num_det] fp32 -> per image: rows [max_det,7] = x1,y1,x2,y2,obj,cls_conf,cls_id (normalised
 * corners), kept anchor indices [max_det] (descending score), count.  Every slot of the outputs is written: slots past `count`
 * get zero rows and index -1 (the caller need not clear them).  `workspace` >= ach_nms_workspace_bytes(). */
size_t ach_nms_workspace_bytes(const ach_handle* h, int32_t batch);
int ach_nms(ach_handle* h, int32_t batch, const float* decoded, float conf_thres, float nms_thres, int32_t max_det,
            float* out_rows, int32_t* out_idx, int32_t* out_count, void* workspace, void* stream);

/* Multi-GPU (the reference's nn.DataParallel gather, achelous.py:176): frames shard by contiguous batch slices with NO data-path
 * collective; the one exchange step is an all-gather of the shards' fixed-size detection records.  A shard's record is ONE flat int32 buffer of
 * ach_record_words(batch, max_det) words, planar: rows [batch, max_det, 7] (fp32 bits) | kept anchor indices [batch, max_det] | counts [batch] — pass
 * pointers into it as out_rows / out_idx / out_count of ach_nms / ach_forward_detect (the kernel fills every slot).  ach_all_gather_records enqueues
 * ncclAllGather(send, recv, words, ncclInt32, comm, stream): `comm` is an ncclComm_t the caller created with RCCL (one rank per GPU), recv_records holds
 * world x words.  RCCL is dlopen'ed at first use (librccl.so, or $ACH_RCCL_LIBRARY); the library does not link against it. */
size_t ach_record_words(int32_t batch, int32_t max_det);
int ach_all_gather_records(ach_handle* h, void* nccl_comm, const int32_t* send_record, int32_t* recv_records, int32_t batch, int32_t max_det, void* stream);

/* Pre / post-processing either side of the forward (the reference does these per frame on the host):
 *   ach_preprocess_radar  <- utils/utils.py:51-54   preprocess_input_radar: (x - min) / (max - min) + 1e-13, min/max per frame;
 *                                                   in fp32 [B,C,R,R] -> out [B,C,R,R] (config dtype)
 *   ach_normalize_points  <- achelous.py:240-243    sklearn normalize(X[N,D], axis=0) + permute; in fp32 [B,N,D] -> out [B,D,N]
 *   ach_preprocess_image  <- utils/utils.py:44-48   preprocess_input + HWC->CHW (achelous.py:205); in uint8 [B,R,R,3] (already
 *                                                   letterboxed) -> out [B,3,R,R]
 *   ach_seg_argmax        <- achelous.py:283-318    per-pixel class at network resolution; in [B,C,R,R] -> out uint8 [B,R,R] */
int ach_preprocess_radar(ach_handle* h, int32_t batch, int32_t channels, const float* in, void* out, void* stream);
int ach_normalize_points(ach_handle* h, int32_t batch, int32_t n, int32_t d, const float* in, void* out, void* stream);
int ach_preprocess_image(ach_handle* h, int32_t batch, const uint8_t* in, void* out, void* stream);
int ach_seg_argmax(ach_handle* h, int32_t batch, int32_t channels, const void* seg, uint8_t* out, void* stream);
/*   ach_seg_resize_argmax <- achelous.py:283-318    the class map at the ORIGINAL image size as detect_image produces it: softmax ->
 *                                                   crop the letterbox bars (utils_seg/utils.py:19-31) -> cv2.resize INTER_LINEAR ->
 *                                                   argmax; seg [B,C,R,R] (config dtype) -> out uint8 [B,out_h,out_w];
 *                                                   prob_workspace: batch * channels * R * R floats
 *   ach_correct_boxes     <- utils_bbox.py:5-30,177-180  kept rows (normalised x1,y1,x2,y2 in the network input) -> (y1,x1,y2,x2) in pixels
 *                                                   of the original image (letterbox undone), rows past count[b] zeroed; fp64 as numpy */
int ach_seg_resize_argmax(ach_handle* h, int32_t batch, int32_t channels, const void* seg, int32_t out_h, int32_t out_w,
                          float* prob_workspace, uint8_t* out, void* stream);
int ach_correct_boxes(ach_handle* h, int32_t batch, int32_t max_det, const float* rows, const int32_t* count, int32_t image_h, int32_t image_w,
                      int32_t letterbox, float* out_rows, void* stream);

/* Range guard of the fp16-storage engine (ACH_DTYPE_F16).  The reference runs fp32 or, under autocast, fp16 WITH torch's GradScaler / inf checks
 * (utils/utils_fit.py:120-166); bf16 callers (BASELINE configs[1]) are served by fp16 storage inside, which overflows at 65504 where bf16 does not.
 * Every kernel of that engine therefore runs with MODE.FP16_OVFL (overflowing conversions clamp to +-65504, never infinity), and this entry counts the
 * elements of the plan's activation tensors that are saturated or non-finite after the forward(s) enqueued on `stream` so far (it synchronises the
 * stream; ~1 ms at batch 64).  0 = the forward stayed inside the fp16 range.  Always 0 for the fp32 / bf16 engines.  The Python module checks a
 * model's first forward with it and falls back to bf16 storage when it is not 0 (achelous_amd/nets.py, `f16_guard`). */
int ach_count_saturated(ach_handle* h, void* stream, uint64_t* count);

/* test hooks: intermediate tensors of the last ach_forward, converted to fp32 NCHW (or [rows, C]) on the host */
int ach_tap_count(const ach_handle* h);
const char* ach_tap_name(const ach_handle* h, int i);
int ach_tap_shape(const ach_handle* h, const char* name, int64_t shape[4], int32_t* ndim);
int ach_read_tap(ach_handle* h, const char* name, float* host_out, size_t capacity_elems);

/* measurement hooks (bench.py): the plan's launch list with the ALGORITHMIC bytes / flops of each launch, a pass
 * that brackets every launch with HIP events, and a live probe that brackets ONE launch on every ach_forward. */
int ach_plan_launches(const ach_handle* h);
const char* ach_op_name(const ach_handle* h, int i);
double ach_op_bytes(const ach_handle* h, int i);          /* inputs read once + outputs written once + weights, REAL channel counts */
double ach_op_layout_bytes(const ach_handle* h, int i);   /* the same over the stored pixel pitches (channel padding included) */
double ach_op_flops(const ach_handle* h, int i);          /* 2 x MACs of the dense contractions of the launch */
int ach_op_stream(const ach_handle* h, int i);            /* 0: caller's stream, 1..: engine side streams */
int ach_forward_profiled(ach_handle* h, const void* image, const void* radar, const void* points,
                         void* det3, void* det4, void* det5, void* se_seg, void* lane_seg, void* pc_seg, void* stream,
                         float* op_ms, size_t capacity);
int ach_set_probe(ach_handle* h, int op_index);          /* -1 disables */
/* micro-benchmark of the MFMA GEMM kernel alone on scratch buffers: ms per launch of Y[M,N] = epi(X[M,K] W^T) */
int ach_bench_gemm(ach_handle* h, int M, int K, int N, int act, int ln, int residual, int P, int iters, void* stream, float* ms);
int ach_read_probe(ach_handle* h, float* avg_ms, int* samples);
/* second probe form: events before launch `first` and after launch `last` of the plan (both on the same stream: a run of the
 * caller's stream such as the neck + decoder sub-path), slot 0, 1 or 2 (ach_set_probe uses slot 0); first = -1 disables */
int ach_set_probe_range(ach_handle* h, int slot, int first, int last);
int ach_read_probe_slot(ach_handle* h, int slot, float* avg_ms, int* samples);

/* ---- training mode, first block (utils/utils_fit.py:37-166 trains through ATen autograd).
 * Stateless fp32 kernels (no handle; errors through ach_last_error(NULL)) for the PointNet shared-MLP layer
 * relu(BatchNorm1d(Conv1d_k1(x))) on [B, C, N] tensors in TRAINING mode — batch statistics — forward and backward:
 *   ach_train_gemm         C[b] (+)= op(A[b]) op(B[b]) [+ bias per row]: row-major, leading dimensions and batch strides in elements,
 *                          trans_x = the stored matrix is the transpose; reduce_batch sums the products over the batch into ONE C
 *   ach_train_bn_stats     mean / biased variance per channel over (B, N)
 *   ach_train_bn_relu_fwd  y = [relu](gamma * (z - mean) / sqrt(var + eps) + beta)
 *   ach_train_bn_relu_bwd  dgamma, dbeta and dz from dy (z, y, mean, var of the forward) */
int ach_train_gemm(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                   int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t trans_a, int32_t trans_b, int32_t batch, int32_t reduce_batch,
                   int32_t accumulate, void* stream);
/*   ach_train_set_gemm_precision   element type of ach_train_gemm's matrix-instruction operands, process-wide: 0 = fp32 (default; the arithmetic the
 *                          float64 reference-step test holds to 5e-3), 1 = the fp32 operands are rounded to bf16 while they are staged into LDS, products
 *                          accumulated in fp32, fp32 output (the reference's default loop runs under torch.cuda.amp.autocast, utils/utils_fit.py:120-166,
 *                          train.py:37 `fp16 = True`; achelous_amd.Achelous.train_precision = 'bf16' selects it).  Every other training kernel stays fp32.
 *                          Measured no faster on this network (its training GEMMs are HBM-bound streams of fp32 activations): opt-in.
 *                          Returns ACH_ERR_INVALID for another value; ach_train_get_gemm_precision returns the current one. */
int ach_train_set_gemm_precision(int32_t precision);
int ach_train_get_gemm_precision(void);
/*   ach_train_gemm_p       ach_train_gemm with the operand type of THIS call (0 / 1 as above; -1 = the process-wide setting): an autograd node records the type its
 *                          forward ran with and passes it to its backward launches, so two modules (or threads) with different `train_precision` never run a backward
 *                          with the other's choice */
int ach_train_gemm_p(const float* A, const float* B, float* C, const float* bias, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                     int64_t stride_a, int64_t stride_b, int64_t stride_c, int32_t trans_a, int32_t trans_b, int32_t batch, int32_t reduce_batch,
                     int32_t accumulate, int32_t precision, void* stream);
int ach_train_bn_stats(const float* z, float* mean, float* var, int32_t B, int32_t C, int32_t N, void* stream);
/*   ach_train_bn_running    nn.BatchNorm's running estimates after a training forward: running <- (1 - momentum) running + momentum batch; `unbias` = M / (M - 1) on the variance */
int ach_train_bn_running(const float* mean, const float* var, float* running_mean, float* running_var, int32_t C, float momentum, float unbias, void* stream);
int ach_train_bn_relu_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta, float* y, int32_t B, int32_t C,
                          int32_t N, float eps, int32_t relu, void* stream);
int ach_train_bn_relu_bwd(const float* z, const float* y, const float* dy, const float* mean, const float* var, const float* gamma, float* dgamma,
                          float* dbeta, float* dz, int32_t B, int32_t C, int32_t N, float eps, int32_t relu, void* stream);
/*   ach_train_dw3x3        depthwise 3x3 / stride 1 / pad 1 on [B,C,H,W], w [C,9]; flip = 1 mirrors the taps (the input gradient)
 *   ach_train_dw3x3_wgrad  its weight gradient dw [C,9] = sum over (B,H,W) of dz x shifted x   (GhostModule cheap operation, ghost_conv.py:19-23) */
int ach_train_dw3x3(const float* x, const float* w, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t flip, void* stream);
int ach_train_dw3x3_wgrad(const float* x, const float* dz, float* dw, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
/*   ach_train_max_points   max over the N points of rows = B*C rows with the arg-max (dy == NULL: forward x -> y, idx; else backward dy, idx -> dx)
 *   ach_train_log_softmax  log-softmax over the classes of z [B,K,N] written as [B,N,K] (dy == NULL: forward; else y, dy -> dz [B,K,N])
 *                          (pointnet_utils.py:32,127; pointnet_sem_seg.py:34-37) */
int ach_train_max_points(const float* x, float* y, int32_t* idx, const float* dy, float* dx, int64_t rows, int32_t N, void* stream);
int ach_train_log_softmax(const float* z, float* y, const float* dy, float* dz, int32_t B, int32_t K, int32_t N, void* stream);

/* One pass of Pillow's 8-bit ImagingResample (the reference letterboxes its input with PIL BICUBIC, utils/utils.py:20-33): integer dot products
 * with 22-bit fixed-point coefficients computed on the host exactly as Resample.c does (achelous_amd/prepost.py).  HWC uint8 images; a resize
 * is a horizontal pass into an 8-bit intermediate and a vertical pass (dst_pitch lets it write into the letterbox canvas).  Stateless. */
int ach_resample_pass_u8(const uint8_t* src, uint8_t* dst, const int32_t* bounds, const int32_t* coeffs, int32_t ksize, int32_t h_in, int32_t w_in,
                         int32_t h_out, int32_t w_out, int32_t channels, int32_t vertical, int64_t src_pitch, int64_t dst_pitch, void* stream);

/* Training-mode primitives, second set (achelous_amd/csrc/k_train2.h): with the functions above, every arithmetic operation of
 * Achelous.forward in .train() — forward and backward (utils/utils_fit.py:37-166 runs them through ATen autograd).  fp32, NCHW-contiguous.
 * A function with a `dy` argument runs FORWARD when dy == NULL and BACKWARD otherwise.
 *   ach_train_act            kind 0 ReLU, 1 SiLU, 2 GELU (erf), 3 sigmoid: out = f(x) | dy * f'(x)
 *   ach_train_layernorm      over C of elements (r, c, i) at (r*C + c)*inner + i (channels-last rows: inner 1; channels_first maps: inner H*W)
 *   ach_train_dwconv         depthwise k x k, stride 1, pad k/2, w [C,k*k]; flip = 1: the input gradient;  _wgrad: dw [C,k*k]
 *   ach_train_im2col         col [B][C*kh*kw][Ho*Wo] <- x (backward = 1: dx <- dcol), dense convolutions run through ach_train_gemm
 *   ach_train_softmax        over the last dimension d of [rows, d]
 *   ach_train_upsample2x     bilinear x2, align_corners=True: [planes,h,w] -> [planes,2h,2w] (backward = 1: the adjoint)
 *   ach_train_maxpool        k x k, stride 1, pad k/2 with arg-max;  ach_train_avgpool3: 3x3 / 9 (self-adjoint)
 *   ach_train_row_reduce     out[r] = scale * sum_i a[r,i] * (b ? b[r,i] : 1);     ach_train_row_scale  out[r,i] = (x ? x[r,i] : 1) * s[r % period]
 *   ach_train_col_reduce     out[c] = scale * sum_r a[r,c] * (b ? b[r,c] : 1);     ach_train_col_scale  out[r,c] = x[r,c] * g[c]
 *   ach_train_instnorm       per-row statistics (GroupNorm with one channel per group), gamma / beta index r % C; backward also returns per-row dgamma / dbeta
 *   ach_train_l2norm         y = x / max(||x||, eps) per row (F.normalize)
 *   ach_train_deform_im2col  modulated deformable 3x3 sampling (torchvision 0.12 deform_conv2d semantics): col [B][C*9][Ho*Wo]
 *   ach_train_deform_bwd     from dcol: doffset, dmask, and dx (scattered with fp32 atomics into a buffer the caller zeroed; dx_zeroed = NULL skips it: an input without a gradient) */
int ach_train_act(const float* x, const float* dy, float* out, int64_t n, int32_t kind, void* stream);
int ach_train_mul(const float* a, const float* b, float* out, int64_t n, void* stream);      /* out = a * b element-wise */
int ach_train_layernorm(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int32_t C, int64_t inner,
                        float eps, void* stream);
int ach_train_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                            int64_t rows, int32_t C, int64_t inner, void* stream);
int ach_train_dwconv(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t k, int32_t flip, void* stream);
int ach_train_dwconv_wgrad(const float* x, const float* dz, float* dw, int32_t B, int32_t C, int32_t H, int32_t W, int32_t k, void* stream);
int ach_train_im2col(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw,
                     int32_t Ho, int32_t Wo, int32_t backward, void* stream);
int ach_train_softmax(const float* x, float* y, const float* dy, float* dx, int64_t rows, int32_t d, void* stream);
int ach_train_upsample2x(const float* src, float* dst, int64_t planes, int32_t h, int32_t w, int32_t backward, void* stream);
int ach_train_maxpool(const float* x, float* y, int32_t* idx, const float* dy, float* dx, int64_t planes, int32_t H, int32_t W, int32_t k, void* stream);
int ach_train_avgpool3(const float* x, float* y, int64_t planes, int32_t H, int32_t W, void* stream);
int ach_train_row_reduce(const float* a, const float* b, float* out, int64_t rows, int64_t n, float scale, void* stream);
int ach_train_row_scale(const float* x, const float* s, float* out, int64_t rows, int64_t n, int64_t period, void* stream);
int ach_train_col_reduce(const float* a, const float* b, float* out, int64_t rows, int64_t cols, float scale, void* stream);
int ach_train_col_scale(const float* x, const float* g, float* out, int64_t rows, int64_t cols, void* stream);
int ach_train_instnorm(const float* x, const float* dy, const float* gamma, const float* beta, float* y, float* mean, float* rstd, float* dx, float* dgamma_rows,
                       float* dbeta_rows, int64_t rows, int64_t n, int32_t C, float eps, void* stream);
int ach_train_l2norm(const float* x, float* y, float* norm, const float* dy, float* dx, int64_t rows, int64_t n, float eps, void* stream);
int ach_train_deform_im2col(const float* x, const float* offset, const float* mask, float* col, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                            int32_t stride, int32_t pad, void* stream);
int ach_train_deform_bwd(const float* x, const float* offset, const float* mask, const float* dcol, float* dx_zeroed, float* doffset, float* dmask, int32_t B,
                         int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t stride, int32_t pad, void* stream);

/* PointNet++ in training mode (achelous_amd/csrc/k_train3.h; OUR OWN specification of `pc_seg='pn2'`, DESIGN.md 5b — the reference snapshot builds only 'pn',
 * nets/Achelous.py:31-32, and trains whatever it builds through ATen autograd, utils/utils_fit.py:37-166).  Point tensors are ROWS: xyz [B, n, 3], features [B, n, C].
 *   ach_train_pn2_fps        farthest-point sampling (start at point 0, ties to the lowest index): idx [B, npoint], new_xyz [B, npoint, 3]
 *   ach_train_pn2_group      ball query + grouping: grouped [B*S*nsample, 3 + C] = [xyz - centroid | features], group_idx [B, S, nsample] (first nsample in-ball
 *                            points in index order, padded with the first)
 *   ach_train_pn2_group_bwd  dfeats[b, group_idx, c] += dgrouped[.., 3 + c] into a buffer the caller zeroed (fp32 atomics; the xyz columns carry no gradient)
 *   ach_train_pn2_interp     dout == NULL: out [B*n, C1 + C2] = [skip | sum_j w_j sparse[nn_j]] (3-NN, w = 1 / (d + 1e-8) normalised);
 *                            else the adjoint: dskip [B*n, C1] and dsparse [B, s, C2] (zeroed by the caller, fp32 atomics) from dout */
int ach_train_pn2_fps(const float* xyz, int32_t B, int32_t n, int32_t npoint, int32_t* idx, float* new_xyz, void* stream);
int ach_train_pn2_group(const float* xyz, const float* new_xyz, const float* feats, int32_t C, int32_t B, int32_t n, int32_t S, int32_t nsample, float radius2,
                        float* grouped, int32_t* group_idx, void* stream);
int ach_train_pn2_group_bwd(const int32_t* group_idx, const float* dgrouped, float* dfeats_zeroed, int32_t C, int32_t B, int32_t n, int32_t S, int32_t nsample, void* stream);
int ach_train_pn2_interp(const float* xyz1, const float* xyz2, const float* skip, int32_t C1, const float* sparse, int32_t C2, float* out, float* dskip, float* dsparse_zeroed,
                         const float* dout, int32_t B, int32_t n, int32_t s, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACHELOUS_H_ */
