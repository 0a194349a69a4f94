// k_pn2.h — geometry kernels of the PointNet++ branch (`pc_seg='pn2'`, BASELINE.json config 4).
//
// The reference snapshot has no PointNet++ code (nets/Achelous.py:31-32 builds only 'pn'; SURVEY.md top): these kernels
// implement OUR OWN specification of that branch (achelous_amd/spec.py::PN2), whose checker is
// oracle/pointnet2_oracle.py.  All index selection is integer work and is bit-exact against that checker, which is why every
// squared distance below is ((dx*dx + dy*dy) + dz*dz) in fp32 with contraction switched off: one rounding per operation, as
// numpy evaluates it.  The shared MLPs of the set-abstraction / feature-propagation levels are plain rows x channels GEMMs and
// run on k_gemm.h (the max over a ball is fused into the last layer's epilogue, gemm_colmax_kernel); what is here is the
// gather-shaped part: farthest-point sampling, ball query + grouping, 3-NN inverse-distance interpolation + skip concatenation.
#pragma once
#include "ach_platform.h"

namespace ach {

__device__ __forceinline__ float pn2_sqdist(float ax, float ay, float az, float bx, float by, float bz) {
#ifndef ACH_HOSTEMU
#pragma clang fp contract(off)
#endif
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

// rows [B*N, ld] (storage type) -> xyz fp32 [B*N, 3]
struct Pn2XyzParams { const void* X; long ldx; float* xyz; long rows; };
template <class T>
__global__ void pn2_xyz_kernel(const Pn2XyzParams p) { f16_sat_mode<T>();
    const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= p.rows * 3) return;
    const long row = idx / 3;
    const int c = int(idx - row * 3);
    p.xyz[idx] = Store<T>::ld(static_cast<const T*>(p.X) + row * p.ldx + c);
}

// ---- farthest-point sampling: ONE WAVE per cloud, no LDS and no barrier.  Lane l keeps points l, l + 64, ... and their running
// minimum distance in registers.  One pick = the distance update, a wave max of the per-lane maxima (DPP within rows of 16 lanes,
// four read-lanes across them), then the lowest index holding that maximum from PPT ballots - scalar work - and a read-lane
// broadcast of the new centroid.  The picks are inherently serial (npoint dependent steps), so a cloud cannot use more than
// one wave usefully: at B = 64 this kernel is latency-bound on 64 waves by construction and the lever is the length of one step
// (measured: 0.96 us per pick with a 256-thread workgroup and two LDS exchanges, ~0.3 us this way).
// Start at point 0; ties to the lowest index.
constexpr int PN2_FPS_MAX_PPT = 16;             // n <= 1024
struct FpsParams { const float* xyz; int n, npoint; int* idx; float* new_xyz; };
// one level for one cloud (blockIdx.x), executed by one wave
template <int PPT>
__device__ __forceinline__ void pn2_fps_level(const FpsParams& p) {
    const int lane = threadIdx.x;
    const float* xyz = p.xyz + long(blockIdx.x) * p.n * 3;
    float px[PPT], py[PPT], pz[PPT], dist[PPT];
    ACH_UNROLL
    for (int k = 0; k < PPT; ++k) {
        const int i = lane + 64 * k;
        const bool in = i < p.n;
        px[k] = in ? xyz[i * 3] : 0.f; py[k] = in ? xyz[i * 3 + 1] : 0.f; pz[k] = in ? xyz[i * 3 + 2] : 0.f;
        dist[k] = in ? 1e10f : -1.f;                                  // distances are >= 0: a padding slot never wins
    }
    int far = 0;
    for (int it = 0; it < p.npoint; ++it) {
        const int fl = far & 63, fk = far >> 6;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        ACH_UNROLL
        for (int k = 0; k < PPT; ++k)
            if (k == fk) { cx = wave_lane_f32(px[k], fl); cy = wave_lane_f32(py[k], fl); cz = wave_lane_f32(pz[k], fl); }
        if (lane == 0) {
            p.idx[long(blockIdx.x) * p.npoint + it] = far;
            float* o = p.new_xyz + (long(blockIdx.x) * p.npoint + it) * 3;
            o[0] = cx; o[1] = cy; o[2] = cz;
        }
        float m = -1.f;
        ACH_UNROLL
        for (int k = 0; k < PPT; ++k) {
            dist[k] = fminf(dist[k], pn2_sqdist(px[k], py[k], pz[k], cx, cy, cz));
            m = fmaxf(m, dist[k]);
        }
        const float smax = wave_max_f32(m);
        int nxt = -1;
        ACH_UNROLL
        for (int k = 0; k < PPT; ++k) {                               // index = lane + 64 k: lowest k first, then lowest lane
            const unsigned long long mask = wave_ballot64(dist[k] == smax);
            if (nxt < 0 && mask) nxt = 64 * k + __ffsll((long long)mask) - 1;
        }
        far = nxt;
    }
}
template <int PPT>
__global__ __launch_bounds__(64) void pn2_fps_kernel(const FpsParams p) { pn2_fps_level<PPT>(p); }

// Every level's sampling in ONE launch (round 4): level k + 1 samples the centroids level k picked, so the whole chain depends on the input cloud
// only.  The wave runs the levels back to back; a level reads the centroids its predecessor wrote (lane 0's stores) after a device-scope fence.
// Replaces four launches that sat between the levels' shared MLPs (the three small ones are pure latency: 64 / 16 / 4 dependent picks).
struct FpsAllParams { FpsParams lv[4]; int levels; };
__device__ __forceinline__ void pn2_fps_dispatch(const FpsParams& q) {
    const int ppt = (q.n + 63) / 64;
    if (ppt <= 1) pn2_fps_level<1>(q);
    else if (ppt <= 4) pn2_fps_level<4>(q);
    else if (ppt <= 8) pn2_fps_level<8>(q);
    else pn2_fps_level<PN2_FPS_MAX_PPT>(q);
}
static __global__ __launch_bounds__(64) void pn2_fps_all_kernel(const FpsAllParams p) {
    for (int l = 0; l < p.levels; ++l) {
        if (l > 0) {
#if defined(ACH_HOSTEMU)
            (void)__shfl(0, 0);
#else
            __threadfence();                      // lane 0's centroid stores of the previous level are visible to the whole wave's loads
#endif
        }
        pn2_fps_dispatch(p.lv[l]);
    }
}
inline void launch_pn2_fps(const FpsParams& p, int B, hipStream_t s) {
    const dim3 grid{unsigned(B)}, block(64);
    const int ppt = cdiv(p.n, 64);
    if (ppt <= 1) ACH_LAUNCH(pn2_fps_kernel<1>, grid, block, s, p);
    else if (ppt <= 4) ACH_LAUNCH(pn2_fps_kernel<4>, grid, block, s, p);
    else if (ppt <= 8) ACH_LAUNCH(pn2_fps_kernel<8>, grid, block, s, p);
    else ACH_LAUNCH(pn2_fps_kernel<PN2_FPS_MAX_PPT>, grid, block, s, p);
}

// ---- ball query + grouping: one wave per centroid.  The wave scans the cloud 64 points at a time in index order; a ballot and
// a prefix pop-count give every in-ball point its slot, so the group is the first `nsample` in-ball indices (padded with the
// first).  The rows of the grouped matrix [ (centroid, sample), 3 + C ] = [xyz - centroid | features] are then written with the
// channel index on the lanes.
constexpr int PN2_MAX_NSAMPLE = 64;
struct GroupParams {
    const float* xyz; const float* new_xyz; const void* feats; long ldf; int C;
    void* out; long ldo; int* group_idx;
    int B, n, S, nsample; float r2;
    int wpc;                          // waves per centroid: 1 (a workgroup = four centroids) or 4 (round 5: a workgroup = ONE centroid, every wave repeats the (cheap) ball query and copies a
};                                    // quarter of the group's rows — the deeper levels have 256 - 4 096 centroids per batch of 64 and were a latency chain on as many waves: 32 - 42 us for 8 - 35 MB)
template <class T>
__global__ __launch_bounds__(256) void pn2_group_kernel(const GroupParams p) { f16_sat_mode<T>();
    __shared__ int s_idx[4][PN2_MAX_NSAMPLE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool shared = p.wpc == 4;
    const long cent = shared ? long(blockIdx.x) : long(blockIdx.x) * 4 + wave;
    if (cent >= long(p.B) * p.S) return;                      // whole waves leave; no workgroup barrier below
    const int jlo = shared ? wave * ((p.nsample + 3) / 4) : 0, jhi_ = shared ? jlo + (p.nsample + 3) / 4 : p.nsample, jhi = jhi_ < p.nsample ? jhi_ : p.nsample;      // this wave's rows of the group
    const long b = cent / p.S;
    const float cx = p.new_xyz[cent * 3], cy = p.new_xyz[cent * 3 + 1], cz = p.new_xyz[cent * 3 + 2];
    const float* xyz = p.xyz + b * p.n * 3;
    int cnt = 0;
    for (int base = 0; base < p.n && cnt < p.nsample; base += 64) {
        const int i = base + lane;
        const bool in = i < p.n && pn2_sqdist(xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], cx, cy, cz) <= p.r2;
        const unsigned long long mask = wave_ballot64(in);
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (in && pos < p.nsample) s_idx[wave][pos] = i;
        cnt += __popcll(mask);
    }
    if (cnt > p.nsample) cnt = p.nsample;
    wave_sync();
    if (lane < p.nsample && (!shared || wave == 0)) {
        const int src = s_idx[wave][lane < cnt ? lane : 0];
        if (p.group_idx) p.group_idx[cent * p.nsample + lane] = src;
    }
    const T* feats = static_cast<const T*>(p.feats) + b * p.n * p.ldf;
    T* out = static_cast<T*>(p.out) + cent * p.nsample * p.ldo;
    const int ldo = int(p.ldo), total = p.nsample * ldo;         // the group's rows are contiguous: (sample, channel) flattened over the lanes
    if (ldo >= 64 && ((jhi - jlo) & 3) == 0) {
        // wide rows (the deeper levels: 67 - 259 columns): FOUR rows at a time with the column index on the lanes — four independent loads in flight per lane and no division;
        // the flattened walk below is a chain of LDS read -> load -> store per element (132 dependent rounds for a 32 x 264 group: 41 us for the last level's 8.5 MB)
        for (int j0 = jlo; j0 < jhi; j0 += 4) {
            int src[4];
            ACH_UNROLL
            for (int q = 0; q < 4; ++q) src[q] = s_idx[wave][(j0 + q) < cnt ? j0 + q : 0];
            for (int col = lane; col < ldo; col += 64) {
                float v[4];
                ACH_UNROLL
                for (int q = 0; q < 4; ++q) {
                    v[q] = 0.f;
                    if (col < 3) v[q] = xyz[src[q] * 3 + col] - (col == 0 ? cx : col == 1 ? cy : cz);
                    else if (col < 3 + p.C) v[q] = Store<T>::ld(feats + long(src[q]) * p.ldf + (col - 3));
                }
                ACH_UNROLL
                for (int q = 0; q < 4; ++q) Store<T>::st(out + (j0 + q) * ldo + col, v[q]);
            }
        }
        return;
    }
    for (int e = jlo * ldo + lane; e < jhi * ldo; e += 64) {
        const int j = e / ldo, col = e - j * ldo;
        const int src = s_idx[wave][j < cnt ? j : 0];
        float v = 0.f;
        if (col < 3) v = xyz[src * 3 + col] - (col == 0 ? cx : col == 1 ? cy : cz);
        else if (col < 3 + p.C) v = Store<T>::ld(feats + long(src) * p.ldf + (col - 3));
        Store<T>::st(out + e, v);
    }
}

// ---- feature propagation input: one wave per dense point.  Distances to the (<= 64 * PN2_INTERP_SPL) sparse points sit in
// registers, three rounds of wave arg-min (ties to the lowest index) give the neighbours, then the row
// [skip features | sum_k w_k * sparse features] is written with the channel index on the lanes.  w = 1 / (d + 1e-8), normalised.
constexpr int PN2_INTERP_SPL = 8;               // s <= 512
struct InterpParams {
    const float* xyz1; const float* xyz2; const void* p1; long ld1; int C1; const void* p2; long ld2; int C2;
    void* out; long ldo; int B, n, s;
};
template <class T>
__global__ __launch_bounds__(256) void pn2_interp_kernel(const InterpParams p) { f16_sat_mode<T>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pt = long(blockIdx.x) * 4 + wave;
    if (pt >= long(p.B) * p.n) return;
    const long b = pt / p.n;
    const float ax = p.xyz1[pt * 3], ay = p.xyz1[pt * 3 + 1], az = p.xyz1[pt * 3 + 2];
    const float* xyz2 = p.xyz2 + b * p.s * 3;
    const float INF = 3.0e38f;
    float d[PN2_INTERP_SPL];
    ACH_UNROLL
    for (int k = 0; k < PN2_INTERP_SPL; ++k) {
        const int i = lane + 64 * k;
        d[k] = i < p.s ? pn2_sqdist(ax, ay, az, xyz2[i * 3], xyz2[i * 3 + 1], xyz2[i * 3 + 2]) : INF;
    }
    int nn[3]; float w[3];
    ACH_UNROLL
    for (int r = 0; r < 3; ++r) {
        float best = INF; int besti = 0x7fffffff;
        ACH_UNROLL
        for (int k = 0; k < PN2_INTERP_SPL; ++k)
            if (d[k] < best) { best = d[k]; besti = lane + 64 * k; }
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m); const int oi = __shfl_xor(besti, m);
            if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        nn[r] = besti;
        w[r] = 1.0f / (best + 1e-8f);
        ACH_UNROLL
        for (int k = 0; k < PN2_INTERP_SPL; ++k)
            if (lane + 64 * k == besti) d[k] = INF;
    }
    const float norm = (w[0] + w[1]) + w[2];
    w[0] /= norm; w[1] /= norm; w[2] /= norm;
    const T* p1 = static_cast<const T*>(p.p1) + pt * p.ld1;
    const T* q = static_cast<const T*>(p.p2) + b * p.s * p.ld2;
    const T* q0 = q + long(nn[0]) * p.ld2; const T* q1 = q + long(nn[1]) * p.ld2; const T* q2 = q + long(nn[2]) * p.ld2;
    T* out = static_cast<T*>(p.out) + pt * p.ldo;
    for (int col = lane; col < int(p.ldo); col += 64) {
        float v = 0.f;
        if (col < p.C1) v = Store<T>::ld(p1 + col);
        else if (col < p.C1 + p.C2) {
            const int c = col - p.C1;
            v = (w[0] * Store<T>::ld(q0 + c) + w[1] * Store<T>::ld(q1 + c)) + w[2] * Store<T>::ld(q2 + c);
        }
        Store<T>::st(out + col, v);
    }
}

}  // namespace ach
