// engine.cpp — weight folding / packing and the launch plan of the Achelous forward path.
//
// The plan is a flat list of kernel launches over a bump-allocated activation arena (288 GB of HBM: no buffer is
// reused inside a forward, so any intermediate can be tapped by the parity tests; channel padding lanes are zeroed
// once at plan time and never written).  Image path: NHWC.  Radar path: planar NCHW.  Point path: [B*N, C] rows.
// Everything input-independent is folded on the host when the plan is built: eval-mode BatchNorm into the
// preceding conv, LayerNorm affine into the following linear, layer scale into the preceding linear, the Fourier
// positional encoding into a constant [HW, C] table.
#include "engine.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <mutex>

#include "k_detect.h"
#include "k_prepost.h"

namespace ach {

#define ACH_HIP_CHECK(expr)                                                                               \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) throw AchError{ACH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)}; \
    } while (0)

static inline long round_up(long v, long m) { return (v + m - 1) / m * m; }

// ================================================================================================ EngineBase
EngineBase::~EngineBase() {
    if (warena) (void)hipFree(warena);
    if (aarena) (void)hipFree(aarena);
    if (prepost_scratch) (void)hipFree(prepost_scratch);
    if (sat_dev) (void)hipFree(sat_dev);
#if !defined(ACH_HOSTEMU)
    drop_graphs();
    if (capture_stream) (void)hipStreamDestroy(capture_stream);
#endif
    if (streams_ready) {
        for (int k = 0; k < kSideStreams; ++k) (void)hipEventDestroy(ev_end[k]);          // (the streams belong to the process-wide pool)
        (void)hipEventDestroy(ev_fork);
        for (int k = 0; k < kJoinEvents; ++k) (void)hipEventDestroy(ev_join[k]);
        for (int q = 0; q < 2; ++q) { (void)hipEventDestroy(ev_x[q]); (void)hipEventDestroy(ev_x2[q]); (void)hipEventDestroy(ev_x3[q]); for (int k = 0; k < kSideStreams; ++k) (void)hipEventDestroy(ev_done[k][q]); }
    }
    for (auto& pr : probes) { for (auto e : pr.ev0) (void)hipEventDestroy(e); for (auto e : pr.ev1) (void)hipEventDestroy(e); }
}

void EngineBase::load(const ach_tensor_desc* t, size_t n) {
    (void)hipDeviceSynchronize();          // a forward in flight still reads the packed weights the next plan() will overwrite
    weights.clear();
    for (size_t i = 0; i < n; ++i) {
        if (!t[i].name || !t[i].data || t[i].ndim < 0 || t[i].ndim > 4) throw AchError{ACH_ERR_INVALID, "bad tensor descriptor"};
        HostTensor h;
        long numel = 1;
        for (int d = 0; d < t[i].ndim; ++d) { h.shape.push_back(long(t[i].shape[d])); numel *= long(t[i].shape[d]); }
        h.data.assign(t[i].data, t[i].data + numel);
        weights[t[i].name] = std::move(h);
    }
}

const HostTensor& EngineBase::W(const std::string& key) const {
    auto it = weights.find(key);
    if (it == weights.end()) throw AchError{ACH_ERR_MISSING_KEY, "state_dict key missing: " + key};
    return it->second;
}

void* EngineBase::walloc(size_t bytes) {
    bytes = size_t(round_up(long(bytes), 256));
    if (measuring) { warena_used += bytes; return reinterpret_cast<void*>(uintptr_t(0x10000)); }
    if (warena_used + bytes > warena_cap) throw AchError{ACH_ERR_NOMEM, "weight arena exhausted"};
    void* p = warena + warena_used;
    warena_used += bytes;
    return p;
}
void* EngineBase::aalloc(size_t bytes) {
    bytes = size_t(round_up(long(bytes), 256));
    void* p = measuring ? reinterpret_cast<void*>(uintptr_t(0x10000) + aarena_used) : static_cast<void*>(aarena + aarena_used);
    if (!measuring && aarena_used + bytes > aarena_cap) throw AchError{ACH_ERR_NOMEM, "activation arena exhausted"};
    aarena_used += bytes;
    return p;
}
// Every fp32 vector is followed by 64 zeros: the GEMM epilogues fetch the bias of a whole 16 NT-channel chunk with unconditional 16-byte
// loads (channels >= N of the last chunk read the zeros; their results are never stored).
float* EngineBase::up_f32(const std::vector<float>& v) {
    constexpr size_t kTail = 64;
    float* d = static_cast<float*>(walloc((v.size() + kTail) * sizeof(float)));
    if (!measuring) {
        std::vector<float> padded(v.size() + kTail, 0.f);
        std::copy(v.begin(), v.end(), padded.begin());
        ACH_HIP_CHECK(hipMemcpy(d, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return d;
}
void* EngineBase::up_raw(const void* src, size_t bytes) {
    void* d = walloc(bytes);
    if (!measuring) ACH_HIP_CHECK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    return d;
}
void EngineBase::reset_plan() {
#if !defined(ACH_HOSTEMU)
    drop_graphs();
#endif
    for (auto& pr : probes) pr.first = pr.last = -1;
    cur_stream = 0; pending_wait = -1; pending_wait2 = -1;
    ops.clear(); taps.clear(); tap_order.clear(); t_regions.clear(); sat_dev_regions = 0;
    warena_used = 0; aarena_used = 0;
}
// Side streams are shared by every engine of the process on a device (one set per priority pattern) and live as long as the process:
// the runtime maps streams onto a handful of hardware queues, and a process with more than four ACTIVE streams loses a quarter of its
// throughput — a module that holds an fp32 and a bf16 engine, or a plain and a pipelined one, must not multiply them.
// Sharing is safe: a stream is an in-order queue, and every forward joins (or, pipelined, is joined by its caller) before reuse.
static std::mutex g_pool_mutex;
static std::map<std::pair<int, int>, std::array<hipStream_t, 3>> g_stream_pool;
void EngineBase::ensure_streams() {
    if (streams_ready) return;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        int dev = 0;
        (void)hipGetDevice(&dev);
        const int mask = head_stream ? 1 : side_low_priority;
        auto it = g_stream_pool.find({dev, mask});
        if (it == g_stream_pool.end()) {
            std::array<hipStream_t, 3> st{};
            // side branches have slack, the caller's stream carries the critical path (backbone -> neck -> decoder): the masked ones
            // are created at the lowest priority so that the dispatcher favours the caller's stream under contention
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            for (int k = 0; k < kSideStreams; ++k) {
                if ((mask >> k) & 1) ACH_HIP_CHECK(hipStreamCreateWithPriority(&st[size_t(k)], hipStreamNonBlocking, lo));
                else ACH_HIP_CHECK(hipStreamCreateWithFlags(&st[size_t(k)], hipStreamNonBlocking));
            }
            it = g_stream_pool.emplace(std::make_pair(dev, mask), st).first;
        }
        for (int k = 0; k < kSideStreams; ++k) side_stream[k] = it->second[size_t(k)];
    }
    for (int k = 0; k < kSideStreams; ++k) ACH_HIP_CHECK(hipEventCreate(&ev_end[k]));
    ACH_HIP_CHECK(hipEventCreate(&ev_fork));
    for (int k = 0; k < kJoinEvents; ++k) ACH_HIP_CHECK(hipEventCreate(&ev_join[k]));
    for (int q = 0; q < 2; ++q) {
        ACH_HIP_CHECK(hipEventCreate(&ev_x[q]));
        ACH_HIP_CHECK(hipEventCreate(&ev_x2[q]));
        ACH_HIP_CHECK(hipEventCreate(&ev_x3[q]));
        for (int k = 0; k < kSideStreams; ++k) ACH_HIP_CHECK(hipEventCreate(&ev_done[k][q]));
    }
    streams_ready = true;
}
// The radar and point branches do not depend on the image path until the fusion stage, and most of their kernels are
// latency-bound on small maps: they run on two engine-owned side streams, forked from and joined back into the caller's
// stream with events, so their launches fill the CUs the image path leaves idle.
//
// Pipelined mode (option "pipeline"): the end-of-forward join is left to join(), so that the caller can enqueue forward k+1 before
// forward k's side streams have finished — the caller's stream is free after the neck and starts the next backbone while stream 2
// still runs forward k's decoders, fusion, head, decode and NMS.  Buffers are NOT double-buffered; the two places where forward k+1
// overwrites something forward k's stream 2 still reads are ordered by a cross-forward event instead: the FPN outputs / attention
// maps (written by the neck on the caller's stream) and the radar pyramid (written by stream 1) are last read by the fusion launch
// of forward k (`xsignal`), and the first neck launch / first radar launch of forward k+1 (`xwait`) wait for it.  Everything else a
// stream rewrites is only read by itself (in order) or by launches that the in-forward join events already order.
// At most two forwards may be un-joined: the third call joins the oldest on its own stream first.
void EngineBase::join(hipStream_t s) {
    if (joined >= issued) return;
    const int q = int(joined & 1);
    for (int k = 0; k < kSideStreams; ++k) if (done_used[q][k]) (void)hipStreamWaitEvent(s, ev_done[k][q], 0);
    ++joined;
}
void EngineBase::run_eager(hipStream_t s) {
    bool used[kSideStreams] = {false, false, false};
    const bool multi = multi_stream;
    const bool piped = pipeline && multi;
    if (piped) { ensure_streams(); while (issued - joined >= 2) join(s); }
    else while (joined < issued) join(s);                        // a plain forward after pipelined ones: drain them first
    const int par = int(issued & 1);
    if (multi) {
        ensure_streams();
        (void)hipEventRecord(ev_fork, s);
        // only the side streams that carry launches of this plan are touched: an idle-but-waiting stream still occupies one of the
        // runtime's hardware queues (measured: a fifth active stream in the process costs 28 % of the throughput)
        bool has[kSideStreams] = {false, false, false};
        for (const Op& op : ops) if (op.stream > 0) has[op.stream - 1] = true;
        for (int k = 0; k < kSideStreams; ++k) if (has[k]) (void)hipStreamWaitEvent(side_stream[k], ev_fork, 0);
    }
    for (size_t i = 0; i < ops.size(); ++i) {
        Op& op = ops[i];
        hipStream_t st = s;
        if (multi && op.stream > 0) { st = side_stream[op.stream - 1]; used[op.stream - 1] = true; }
        if (multi && op.wait_ev >= 0) (void)hipStreamWaitEvent(st, ev_join[op.wait_ev], 0);
        if (multi && op.wait_ev2 >= 0) (void)hipStreamWaitEvent(st, ev_join[op.wait_ev2], 0);
        if (piped && op.xwait && issued > 0) (void)hipStreamWaitEvent(st, ev_x[par ^ 1], 0);
        if (piped && (dbg_xwait2_op >= 0 ? int(i) == dbg_xwait2_op : op.xwait2) && issued > 0 && x2_recorded[par ^ 1]) (void)hipStreamWaitEvent(st, ev_x2[par ^ 1], 0);
        if (piped && op.xwait3 && issued > 0 && x3_recorded[par ^ 1]) (void)hipStreamWaitEvent(st, ev_x3[par ^ 1], 0);
        for (auto& pr : probes) if (int(i) == pr.first) (void)hipEventRecord(pr.ev0[size_t(pr.count % kProbeEvents)], st);
        op.fn(st);
        for (auto& pr : probes) if (int(i) == pr.last) { (void)hipEventRecord(pr.ev1[size_t(pr.count % kProbeEvents)], st); ++pr.count; }
        if (multi && op.signal_ev >= 0) (void)hipEventRecord(ev_join[op.signal_ev], st);
        if (piped && op.xsignal) (void)hipEventRecord(ev_x[par], st);
        if (piped && op.xsignal2) { (void)hipEventRecord(ev_x2[par], st); x2_recorded[par] = true; }
        if (piped && op.xsignal3) { (void)hipEventRecord(ev_x3[par], st); x3_recorded[par] = true; }
    }
    if (detect_tail) detect_tail((multi && detect_stream > 0 && used[detect_stream - 1]) ? side_stream[detect_stream - 1] : s);   // det maps are final on that stream
    if (piped) {
        for (int k = 0; k < kSideStreams; ++k) { done_used[par][k] = used[k]; if (used[k]) (void)hipEventRecord(ev_done[k][par], side_stream[k]); }
        ++issued;
    } else if (multi)
        for (int k = 0; k < kSideStreams; ++k)
            if (used[k]) { (void)hipEventRecord(ev_end[k], side_stream[k]); (void)hipStreamWaitEvent(s, ev_end[k], 0); }
}
#if defined(ACH_HOSTEMU)
void EngineBase::run(hipStream_t s) { run_eager(s); }
#else
static bool same_io(const IoPtrs& a, const IoPtrs& b) {
    return a.image == b.image && a.radar == b.radar && a.points == b.points && a.det[0] == b.det[0] && a.det[1] == b.det[1] &&
           a.det[2] == b.det[2] && a.se == b.se && a.lane == b.lane && a.pc == b.pc;
}
void EngineBase::drop_graphs() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
    graphs.clear();
}
// The plan is ~200 launches; at small batch the forward is bound by host launch cost, not by the kernels.  The whole plan
// (including the fork / join of the side streams) is therefore captured once per distinct set of I/O pointers into a hipGraph
// and replayed with one hipGraphLaunch on the caller's stream.  Capture happens on an engine-owned stream (the caller's may be
// the legacy default stream, which cannot be captured).  Any failure falls back to eager launches of the SAME kernels.
void EngineBase::run(hipStream_t s) {
    if (!use_graph || graph_failed || probing() || pipeline) { run_eager(s); return; }
    for (auto& g : graphs)
        if (same_io(g.io, io)) {
            g.stamp = ++graph_clock;
            if (hipGraphLaunch(g.exec, s) == hipSuccess) return;
            graph_failed = true; run_eager(s); return;
        }
    if (!capture_stream && hipStreamCreateWithFlags(&capture_stream, hipStreamNonBlocking) != hipSuccess) { graph_failed = true; run_eager(s); return; }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { graph_failed = true; run_eager(s); return; }
    run_eager(capture_stream);
    if (hipStreamEndCapture(capture_stream, &graph) != hipSuccess || !graph) { (void)hipGetLastError(); graph_failed = true; run_eager(s); return; }
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess || !exec) { (void)hipGetLastError(); graph_failed = true; run_eager(s); return; }
    if (graphs.size() >= 8) {                                  // evict the least recently used
        size_t lru = 0;
        for (size_t i = 1; i < graphs.size(); ++i) if (graphs[i].stamp < graphs[lru].stamp) lru = i;
        (void)hipGraphExecDestroy(graphs[lru].exec);
        graphs.erase(graphs.begin() + long(lru));
    }
    graphs.push_back(GraphEntry{io, exec, ++graph_clock});
    if (hipGraphLaunch(exec, s) != hipSuccess) { graph_failed = true; run_eager(s); }
}
#endif
void EngineBase::run_profiled(hipStream_t s, float* op_ms, size_t cap) {
    if (cap < ops.size()) throw AchError{ACH_ERR_INVALID, "profile buffer too small"};
    std::vector<hipEvent_t> ev(ops.size() + 1);
    for (auto& e : ev) ACH_HIP_CHECK(hipEventCreate(&e));
    ACH_HIP_CHECK(hipEventRecord(ev[0], s));
    for (size_t i = 0; i < ops.size(); ++i) { ops[i].fn(s); ACH_HIP_CHECK(hipEventRecord(ev[i + 1], s)); }
    ACH_HIP_CHECK(hipEventSynchronize(ev[ops.size()]));
    for (size_t i = 0; i < ops.size(); ++i) ACH_HIP_CHECK(hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]));
    for (auto& e : ev) (void)hipEventDestroy(e);
}
void EngineBase::set_probe_range(int slot, int first, int last) {
    if (slot < 0 || slot >= kProbeSlots) throw AchError{ACH_ERR_INVALID, "probe slot out of range"};
    if (first >= int(ops.size()) || last >= int(ops.size()) || (first >= 0 && last < first)) throw AchError{ACH_ERR_INVALID, "probe index out of range"};
    if (first >= 0 && ops[size_t(first)].stream != ops[size_t(last)].stream) throw AchError{ACH_ERR_INVALID, "probe range must start and end on one stream"};
    Probe& pr = probes[slot];
    if (pr.ev0.empty()) {
        pr.ev0.resize(kProbeEvents); pr.ev1.resize(kProbeEvents);
        for (int i = 0; i < kProbeEvents; ++i) { ACH_HIP_CHECK(hipEventCreate(&pr.ev0[size_t(i)])); ACH_HIP_CHECK(hipEventCreate(&pr.ev1[size_t(i)])); }
    }
    pr.first = first < 0 ? -1 : first; pr.last = first < 0 ? -1 : last;
    pr.count = 0;
}
void EngineBase::read_probe_slot(int slot, float* avg_ms, int* samples) {
    if (slot < 0 || slot >= kProbeSlots) throw AchError{ACH_ERR_INVALID, "probe slot out of range"};
    Probe& pr = probes[slot];
    const long n = pr.count < kProbeEvents ? pr.count : kProbeEvents;
    double tot = 0;
    for (long i = 0; i < n; ++i) {
        float ms = 0.f;
        ACH_HIP_CHECK(hipEventSynchronize(pr.ev1[size_t(i)]));
        ACH_HIP_CHECK(hipEventElapsedTime(&ms, pr.ev0[size_t(i)], pr.ev1[size_t(i)]));
        tot += ms;
    }
    *avg_ms = n ? float(tot / double(n)) : 0.f;
    *samples = int(n);
}
int EngineBase::num_anchors() const {
    const int r = cfg.resolution;
    return (r / 8) * (r / 8) + (r / 16) * (r / 16) + (r / 32) * (r / 32);
}
size_t EngineBase::nms_workspace_bytes(int B) const { return size_t(B) * num_anchors() * (8 * sizeof(float) + sizeof(int) + 4 * sizeof(float)) + 512; }

void EngineBase::nms(int B, const float* decoded, float conf, float iou, int max_det, float* rows, int* idx, int* count,
                     void* workspace, hipStream_t s) {
    const int A = num_anchors();
    if (A > NMS_MAXA) throw AchError{ACH_ERR_UNSUPPORTED, "device NMS supports up to 4096 anchors per image (resolution <= 416)"};
    NmsParams p;
    p.dec = decoded;
    p.scratch = static_cast<float*>(workspace);
    p.scratch_idx = reinterpret_cast<int*>(static_cast<char*>(workspace) + size_t(round_up(long(B) * A * 8 * long(sizeof(float)), 256)));
    p.scratch_boxes = reinterpret_cast<float*>(reinterpret_cast<char*>(p.scratch_idx) + size_t(round_up(long(B) * A * long(sizeof(int)), 256)));
    p.rows = rows; p.kept = idx; p.count = count;
    p.B = B; p.A = A; p.NC5 = 5 + cfg.num_det; p.num_classes = cfg.num_det; p.max_det = max_det; p.conf = conf; p.iou = iou;
    ACH_LAUNCH(nms_kernel, dim3(unsigned(B)), dim3(NMS_THREADS), s, p);
}

void EngineBase::correct_boxes(int B, int max_det, const float* rows, const int* count, int img_h, int img_w, int letterbox, float* out, hipStream_t s) {
    BoxCorrectParams p{rows, count, out, B, max_det, double(cfg.resolution), double(cfg.resolution), double(img_h), double(img_w), letterbox};
    ACH_LAUNCH(correct_boxes_kernel, dim3(unsigned(cdivl(long(B) * max_det, 256))), dim3(256), s, p);
}

std::vector<long> EngineBase::tap_shape(const std::string& name) const {
    auto it = taps.find(name);
    if (it == taps.end()) throw AchError{ACH_ERR_INVALID, "unknown tap: " + name};
    const TapInfo& t = it->second;
    if (t.kind == 2) return t.add_eye ? std::vector<long>{t.B, t.add_eye, t.add_eye} : std::vector<long>{long(t.B) * t.H * t.W, t.C};
    return {t.B, t.C, t.H, t.W};
}


// ================================================================================================ taps
void EngineBase::read_tap(const std::string& name, float* out, size_t cap) {
    auto it = taps.find(name);
    if (it == taps.end()) throw AchError{ACH_ERR_INVALID, "unknown tap: " + name};
    const TapInfo& t = it->second;
    const bool bf = cfg.dtype == ACH_DTYPE_BF16 && !t.is_f32 && !t.is_i32, hf = cfg.dtype == ACH_DTYPE_F16 && !t.is_f32 && !t.is_i32;
    const size_t esz = (bf || hf) ? 2 : 4;
    ACH_HIP_CHECK(hipDeviceSynchronize());
    auto fetch = [&](size_t elems) {
        std::vector<unsigned char> raw(elems * esz);
        ACH_HIP_CHECK(hipMemcpy(raw.data(), t.ptr, raw.size(), hipMemcpyDeviceToHost));
        std::vector<float> f(elems);
        if (bf) for (size_t i = 0; i < elems; ++i) { uint16_t b; std::memcpy(&b, &raw[i * 2], 2); f[i] = bf16_to_f32(b); }
        else if (hf) for (size_t i = 0; i < elems; ++i) { uint16_t b; std::memcpy(&b, &raw[i * 2], 2); f[i] = f16_bits_to_f32(b); }
        else if (t.is_i32) for (size_t i = 0; i < elems; ++i) { int32_t v; std::memcpy(&v, &raw[i * 4], 4); f[i] = float(v); }
        else std::memcpy(f.data(), raw.data(), elems * 4);
        return f;
    };
    if (t.kind == 0) {
        const size_t rows = size_t(t.B) * t.H * t.W, HW = size_t(t.H) * t.W;
        if (cap < rows * t.C) throw AchError{ACH_ERR_INVALID, "tap buffer too small"};
        // a view may start inside a wider row: fetch up to the last element it covers
        std::vector<float> f = fetch((rows - 1) * size_t(t.ld) + t.C);
        for (size_t b = 0; b < size_t(t.B); ++b)
            for (size_t p = 0; p < HW; ++p)
                for (int c = 0; c < t.C; ++c) out[(b * t.C + c) * HW + p] = f[(b * HW + p) * t.ld + c];
    } else if (t.kind == 1) {
        const size_t n = size_t(t.B) * t.C * t.H * t.W;
        if (cap < n) throw AchError{ACH_ERR_INVALID, "tap buffer too small"};
        std::vector<float> f = fetch(n);
        std::memcpy(out, f.data(), n * 4);
    } else if (t.kind == 2) {
        const size_t rows = size_t(t.B) * t.H * t.W;
        if (cap < rows * t.C) throw AchError{ACH_ERR_INVALID, "tap buffer too small"};
        std::vector<float> f = fetch((rows - 1) * size_t(t.ld) + t.C);
        for (size_t r = 0; r < rows; ++r)
            for (int c = 0; c < t.C; ++c) {
                float v = f[r * t.ld + c];
                if (t.add_eye && (c / t.add_eye) == (c % t.add_eye)) v += 1.0f;
                out[r * t.C + c] = v;
            }
    } else throw AchError{ACH_ERR_INVALID, "tap not readable"};
}

EngineBase* make_engine_f32(const ach_config& cfg);       // engine_f32.cpp
EngineBase* make_engine_bf16(const ach_config& cfg);      // engine_bf16.cpp
EngineBase* make_engine_f16(const ach_config& cfg);       // engine_f16.cpp
EngineBase* make_engine(const ach_config& cfg) {
    if (cfg.dtype == ACH_DTYPE_F32) return make_engine_f32(cfg);
    if (cfg.dtype == ACH_DTYPE_BF16) return make_engine_bf16(cfg);
    if (cfg.dtype == ACH_DTYPE_F16) return make_engine_f16(cfg);
    throw AchError{ACH_ERR_INVALID, "unknown dtype"};
}

}  // namespace ach
