// hostemu.h — a tiny HIP-on-CPU emulator.  TEST INFRASTRUCTURE ONLY.
//
// Purpose: let the *same* kernel sources under achelous_amd/csrc/ be compiled with g++ and executed on
// the CPU of the build container (which has no GPU), so that kernel indexing, the weight packer and the
// engine's plan can be checked against the oracle before a (scarce) GPU slot is spent.  It is never
// linked into libachelous_hip.so and is not reachable from the achelous_amd package: the product path
// fails loudly without a GPU.  The library built from it (tests/hostemu/libachelous_emu.so) is loaded
// only by tests/.
//
// Model: every thread of a workgroup is a fiber (hand-rolled x86-64 context switch); a workgroup runs
// on one OS thread; workgroups are distributed over OS threads.  `__syncthreads()` and the wavefront
// collectives (shuffles, MFMA) are rendezvous points at which fibers yield to a small scheduler.
// Wavefront = 64 lanes, MFMA fragment layouts as documented for gfx950
// (/opt/skills/guides/cdna_hip_programming.md §3).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

namespace hostemu {
struct ThreadCtx {
    dim3 tid, bid, bdim, gdim;
};
extern thread_local ThreadCtx g_ctx;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void block_sync();
// wave collectives: deposit `bytes` (<= 64) at this lane's slot, rendezvous, then read other lanes' slots.
void wave_deposit(const void* src, int bytes);
const void* wave_slot(int lane);
void wave_release();
int lane_id();
}  // namespace hostemu

#define threadIdx (hostemu::g_ctx.tid)
#define blockIdx (hostemu::g_ctx.bid)
#define blockDim (hostemu::g_ctx.bdim)
#define gridDim (hostemu::g_ctx.gdim)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
static inline void __syncthreads() { hostemu::block_sync(); }

template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    hostemu::wave_deposit(&v, sizeof(T));
    T r;
    std::memcpy(&r, hostemu::wave_slot(hostemu::lane_id() ^ mask), sizeof(T));
    hostemu::wave_release();
    return r;
}
template <class T>
static inline T __shfl_down(T v, int delta, int width = 64) {
    (void)width;
    hostemu::wave_deposit(&v, sizeof(T));
    int src = hostemu::lane_id() + delta;
    if (src > 63) src = hostemu::lane_id();
    T r;
    std::memcpy(&r, hostemu::wave_slot(src), sizeof(T));
    hostemu::wave_release();
    return r;
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    hostemu::wave_deposit(&v, sizeof(T));
    T r;
    std::memcpy(&r, hostemu::wave_slot(src & 63), sizeof(T));
    hostemu::wave_release();
    return r;
}

static inline float atomicAdd(float* p, float v) {
    auto* a = reinterpret_cast<std::atomic<float>*>(p);
    float old = a->load(std::memory_order_relaxed);
    while (!a->compare_exchange_weak(old, old + v, std::memory_order_relaxed)) {}
    return old;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }

// ---- the slice of the HIP runtime API the engine uses ------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { std::free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hostemu"; }
typedef void* hipEvent_t;
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return 0; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, void*, unsigned) { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
