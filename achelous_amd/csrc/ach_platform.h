// ach_platform.h — the only place that knows whether the sources are being compiled by hipcc for gfx950
// (the product: libachelous_hip.so) or by g++ against tests/hostemu (a CPU emulation used by the
// `-m "not gpu"` tests to check kernel indexing and the engine plan; never shipped, never a fallback).
#pragma once
#include <cstdint>
#include <cstddef>
#include <type_traits>

#if defined(ACH_HOSTEMU)
#include "hostemu.h"
#define ACH_LAUNCH(kern, grid, block, stream, ...) \
    do { (void)(stream); hostemu::launch((grid), (block), [=]() { kern(__VA_ARGS__); }); } while (0)
#define ACH_LAUNCH_LDS(kern, grid, block, lds_bytes, stream, ...) ACH_LAUNCH(kern, grid, block, stream, __VA_ARGS__)
#define ACH_UNROLL
#define ACH_NO_UNROLL
namespace ach {
struct f32x4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
}  // namespace ach
#else
#include <hip/hip_runtime.h>
#define ACH_LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL(kern, (grid), (block), 0, (stream), __VA_ARGS__)
// the same launch with `lds_bytes` of (unused) dynamic LDS per workgroup: an OCCUPANCY CAP — a compute unit holds at most 160 KB / lds_bytes workgroups of the
// kernel, whatever its registers would allow.  For long-lived, register-heavy kernels on the side streams: with 3 x 152 VGPRs per SIMD the row-walking
// head leaves the caller's stream no room on ANY compute unit for as long as it runs.
#define ACH_LAUNCH_LDS(kern, grid, block, lds_bytes, stream, ...) hipLaunchKernelGGL(kern, (grid), (block), (lds_bytes), (stream), __VA_ARGS__)
#define ACH_UNROLL _Pragma("unroll")
#define ACH_NO_UNROLL _Pragma("unroll 1")
namespace ach {
typedef float f32x4 __attribute__((ext_vector_type(4)));
}  // namespace ach
#endif

namespace ach {

// two packed fp32 lanes: `a * b + c` on these becomes one v_pk_fma_f32 (GCC/clang generic vector, also fine on the host)
typedef float f32x2 __attribute__((vector_size(8)));

// ---------------------------------------------------------------------------------------- storage types
struct bf16_t { uint16_t bits; };

__host__ __device__ __forceinline__ float bf16_to_f32(uint16_t b) {
    union { uint32_t u; float f; } c;
    c.u = uint32_t(b) << 16;
    return c.f;
}
// round to nearest even.  On the device this is gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR of values; the integer
// sequence below costs five per value and bf16 packing sits in every kernel's epilogue); the host packers and the CPU
// emulation of the kernels use the integer form, which rounds identically.
__host__ __device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return uint16_t(u >> 16);
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f)); }
#else
__host__ __device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return uint32_t(f32_to_bf16_bits(lo)) | (uint32_t(f32_to_bf16_bits(hi)) << 16);
}
__host__ __device__ __forceinline__ uint16_t f32_to_bf16(float f) { return f32_to_bf16_bits(f); }
#endif


// fp16 storage (round 4): same bytes and the same MFMA rate as bf16 with an 8x finer mantissa (11 bits against 8); it is also the type the
// reference's own mixed-precision mode computes in (torch.cuda.amp.autocast, utils/utils_fit.py:120-121; train.py:37 --fp16).  Round to
// nearest even; the host form below is what the weight packers and the CPU emulation use, the device uses v_cvt_pk_f16_f32 / v_cvt_f32_f16.
struct f16_t { uint16_t bits; };

__host__ __device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    const uint32_t sign = (c.u >> 16) & 0x8000u;
    const uint32_t x = c.u & 0x7fffffffu;
    if (x >= 0x7f800000u) return uint16_t(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    const uint32_t e = x >> 23;
    if (e >= 143u) return uint16_t(sign | 0x7bffu);                        // >= 65536: the largest finite half (MODE.FP16_OVFL, see f16_sat_mode below)
    if (e <= 112u) {                                                       // below 2^-14: subnormal or zero, unit 2^-24
        if (x < 0x33000000u) return uint16_t(sign);                        // < 2^-25 (a tie at 2^-25 rounds to even = 0)
        const uint32_t m = (x & 0x7fffffu) | 0x800000u, s = 126u - e;      // value = m * 2^(e - 126) units
        uint32_t r = m >> s;
        const uint32_t rem = m & ((1u << s) - 1u), half = 1u << (s - 1u);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return uint16_t(sign | r);
    }
    uint32_t r = ((e - 112u) << 10) | ((x & 0x7fffffu) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;                // a carry into the exponent is the right answer ...
    if (r >= 0x7c00u) r = 0x7bffu;                                         // ... except into infinity: a finite value saturates
    return uint16_t(sign | r);
}
__host__ __device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
    union { uint32_t u; float f; } c;
    const uint32_t sign = uint32_t(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0x1fu) c.u = sign | 0x7f800000u | (m << 13);
    else if (e != 0u) c.u = sign | ((e + 112u) << 23) | (m << 13);
    else if (m == 0u) c.u = sign;
    else {                                                                 // subnormal: m * 2^-24, exactly representable
        c.f = float(m) * 5.9604644775390625e-08f;
        c.u |= sign;
    }
    return c.f;
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_hw));              // v_cvt_pk_f16_f32
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f)); }
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return float(__builtin_bit_cast(_Float16, h)); }
__device__ __forceinline__ float f16lo_to_f32(uint32_t w) { return float(__builtin_bit_cast(f16x2_hw, w).x); }     // v_cvt_f32_f16
__device__ __forceinline__ float f16hi_to_f32(uint32_t w) { return float(__builtin_bit_cast(f16x2_hw, w).y); }     // v_cvt_f32_f16_sdwa WORD_1
#else
__host__ __device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    return uint32_t(f32_to_f16_bits(lo)) | (uint32_t(f32_to_f16_bits(hi)) << 16);
}
__host__ __device__ __forceinline__ uint16_t f32_to_f16(float f) { return f32_to_f16_bits(f); }
__host__ __device__ __forceinline__ float f16_to_f32(uint16_t h) { return f16_bits_to_f32(h); }
__host__ __device__ __forceinline__ float f16lo_to_f32(uint32_t w) { return f16_bits_to_f32(uint16_t(w & 0xffffu)); }
__host__ __device__ __forceinline__ float f16hi_to_f32(uint32_t w) { return f16_bits_to_f32(uint16_t(w >> 16)); }
#endif

// fp16 storage overflows at 65504 where bf16 does not.  Every kernel of the fp16 engine sets MODE.FP16_OVFL (bit 23 of HW_REG_MODE) as its first instruction:
// an fp16 result that overflows (v_cvt_pk_f16_f32, v_cvt_f16_f32, the packed-half arithmetic of k_conv3.h) then clamps to +-65504 instead of becoming infinity
// (measured on the MI355X, profiles/scripts/ubench/f16_ovfl.hip; infinities and NaNs that come IN stay what they are).  A saturated activation is still a wrong
// one — what the mode buys is that it stays finite and DETECTABLE: ach_count_saturated (api.cpp) counts the +-65504 / non-finite elements of a forward's
// activation tensors, and achelous_amd.Achelous falls back to bf16 storage when a model's first forward shows any (nets.py, `f16_guard`).  The host converter
// above saturates the same way, so the CPU emulation and the weight packers agree with the device.
template <class T> __device__ __forceinline__ void f16_sat_mode() {
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(ACH_NO_F16_OVFL)                                                                                  // (experiments only: the mode left clear)
    if constexpr (std::is_same<T, f16_t>::value) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = 1
#endif
#endif
}

// the two 16-bit storage types behind one interface: a dword holds two consecutive elements (low half first)
template <class T> struct is_h16 { static constexpr bool value = false; };
template <> struct is_h16<bf16_t> { static constexpr bool value = true; };
template <> struct is_h16<f16_t> { static constexpr bool value = true; };
template <class T> struct H16;
template <> struct H16<bf16_t> {
    __host__ __device__ static __forceinline__ uint16_t bits(float v) { return f32_to_bf16_bits(v); }         // host packers (RNE, identical to the device's)
    __host__ __device__ static __forceinline__ float lo(uint32_t w) { union { uint32_t u; float f; } c; c.u = w << 16; return c.f; }
    __host__ __device__ static __forceinline__ float hi(uint32_t w) { union { uint32_t u; float f; } c; c.u = w & 0xffff0000u; return c.f; }
    __host__ __device__ static __forceinline__ uint32_t pack(float a, float b) { return pack_bf16x2(a, b); }
    __host__ __device__ static __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};
template <> struct H16<f16_t> {
    __host__ __device__ static __forceinline__ uint16_t bits(float v) { return f32_to_f16_bits(v); }
    __host__ __device__ static __forceinline__ float lo(uint32_t w) { return f16lo_to_f32(w); }
    __host__ __device__ static __forceinline__ float hi(uint32_t w) { return f16hi_to_f32(w); }
    __host__ __device__ static __forceinline__ uint32_t pack(float a, float b) { return pack_f16x2(a, b); }
    __host__ __device__ static __forceinline__ float round(float v) { return f16_to_f32(f32_to_f16(v)); }
};

// a dword of two IO elements as a dword of two T elements (identity when the types agree)
template <class IO, class T> __host__ __device__ __forceinline__ uint32_t h16_recast(uint32_t w) {
    if constexpr (std::is_same<IO, T>::value || !is_h16<IO>::value || !is_h16<T>::value) return w;
    else return H16<T>::pack(H16<IO>::lo(w), H16<IO>::hi(w));
}

template <class T> struct Store;
template <> struct Store<float> {
    static constexpr int VEC = 4;          // elements per 16 bytes
    __host__ __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __host__ __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    // 4 consecutive elements (16 B aligned)
    __device__ static __forceinline__ void ld4(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    __device__ static __forceinline__ void st4(float* p, const float (&i)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(i[0], i[1], i[2], i[3]);
    }
    // 8 consecutive elements (16 B aligned)
    __device__ static __forceinline__ void ld8(const float* p, float (&o)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    __device__ static __forceinline__ void st8(float* p, const float (&i)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(i[0], i[1], i[2], i[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(i[4], i[5], i[6], i[7]);
    }
};
template <> struct Store<bf16_t> {
    static constexpr int VEC = 8;
    __host__ __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->bits); }
    __host__ __device__ static __forceinline__ void st(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }
    // 4 consecutive elements (8 B aligned)
    __device__ static __forceinline__ void ld4(const bf16_t* p, float (&o)[4]) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = bf16_to_f32(uint16_t(v.x & 0xffffu)); o[1] = bf16_to_f32(uint16_t(v.x >> 16));
        o[2] = bf16_to_f32(uint16_t(v.y & 0xffffu)); o[3] = bf16_to_f32(uint16_t(v.y >> 16));
    }
    __device__ static __forceinline__ void st4(bf16_t* p, const float (&i)[4]) {
        uint2 v;
        v.x = pack_bf16x2(i[0], i[1]);
        v.y = pack_bf16x2(i[2], i[3]);
        *reinterpret_cast<uint2*>(p) = v;
    }
    // 8 consecutive elements (16 B aligned)
    __device__ static __forceinline__ void ld8(const bf16_t* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; ++k) { o[2 * k] = bf16_to_f32(uint16_t(w[k] & 0xffffu)); o[2 * k + 1] = bf16_to_f32(uint16_t(w[k] >> 16)); }
    }
    __device__ static __forceinline__ void st8(bf16_t* p, const float (&i)[8]) {
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) w[k] = pack_bf16x2(i[2 * k], i[2 * k + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

template <> struct Store<f16_t> {
    static constexpr int VEC = 8;
    __host__ __device__ static __forceinline__ float ld(const f16_t* p) { return f16_to_f32(p->bits); }
    __host__ __device__ static __forceinline__ void st(f16_t* p, float v) { p->bits = f32_to_f16(v); }
    __device__ static __forceinline__ void ld4(const f16_t* p, float (&o)[4]) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = f16lo_to_f32(v.x); o[1] = f16hi_to_f32(v.x); o[2] = f16lo_to_f32(v.y); o[3] = f16hi_to_f32(v.y);
    }
    __device__ static __forceinline__ void st4(f16_t* p, const float (&i)[4]) {
        uint2 v;
        v.x = pack_f16x2(i[0], i[1]);
        v.y = pack_f16x2(i[2], i[3]);
        *reinterpret_cast<uint2*>(p) = v;
    }
    __device__ static __forceinline__ void ld8(const f16_t* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; ++k) { o[2 * k] = f16lo_to_f32(w[k]); o[2 * k + 1] = f16hi_to_f32(w[k]); }
    }
    __device__ static __forceinline__ void st8(f16_t* p, const float (&i)[8]) {
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) w[k] = pack_f16x2(i[2 * k], i[2 * k + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// element i of a tensor the CALLER owns (network outputs): the storage type, or — fp16-storage engine with bf16 inputs / outputs — bf16
template <class T> __device__ __forceinline__ void st_user(void* Y, long i, float v, bool alt_bf16) {
    if constexpr (std::is_same<T, f16_t>::value) { if (alt_bf16) { Store<bf16_t>::st(static_cast<bf16_t*>(Y) + i, v); return; } }
    Store<T>::st(static_cast<T*>(Y) + i, v);
}
template <class T> __device__ __forceinline__ float ld_user(const void* X, long i, bool alt_bf16) {
    if constexpr (std::is_same<T, f16_t>::value) { if (alt_bf16) return Store<bf16_t>::ld(static_cast<const bf16_t*>(X) + i); }
    return Store<T>::ld(static_cast<const T*>(X) + i);
}

// unpack one 16-byte fragment (4 f32 or 8 bf16) to floats / pack it back
template <class T> __device__ __forceinline__ void frag_unpack(const uint4& f, float* o);
template <> __device__ __forceinline__ void frag_unpack<float>(const uint4& f, float* o) {
    o[0] = __uint_as_float(f.x); o[1] = __uint_as_float(f.y); o[2] = __uint_as_float(f.z); o[3] = __uint_as_float(f.w);
}
template <> __device__ __forceinline__ void frag_unpack<bf16_t>(const uint4& f, float* o) {
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) { o[2 * i] = bf16_to_f32(uint16_t(w[i] & 0xffffu)); o[2 * i + 1] = bf16_to_f32(uint16_t(w[i] >> 16)); }
}
template <> __device__ __forceinline__ void frag_unpack<f16_t>(const uint4& f, float* o) {
    const uint32_t w[4] = {f.x, f.y, f.z, f.w};
    ACH_UNROLL
    for (int i = 0; i < 4; ++i) { o[2 * i] = f16lo_to_f32(w[i]); o[2 * i + 1] = f16hi_to_f32(w[i]); }
}
template <class T> __device__ __forceinline__ uint4 frag_pack(const float* i);
template <> __device__ __forceinline__ uint4 frag_pack<float>(const float* i) {
    return make_uint4(__float_as_uint(i[0]), __float_as_uint(i[1]), __float_as_uint(i[2]), __float_as_uint(i[3]));
}
template <> __device__ __forceinline__ uint4 frag_pack<bf16_t>(const float* i) {
    uint32_t w[4];
    ACH_UNROLL
    for (int k = 0; k < 4; ++k) w[k] = pack_bf16x2(i[2 * k], i[2 * k + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <> __device__ __forceinline__ uint4 frag_pack<f16_t>(const float* i) {
    uint32_t w[4];
    ACH_UNROLL
    for (int k = 0; k < 4; ++k) w[k] = pack_f16x2(i[2 * k], i[2 * k + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---------------------------------------------------------------------------------------- MFMA
// One "k-chunk" of the 16x16 MFMA family: every lane contributes 16 bytes of A and 16 bytes of B.
//   bf16: v_mfma_f32_16x16x32_bf16 — lane l holds A[i = l&15][k = (l>>4)*8 + j], B[k = (l>>4)*8 + j][n = l&15], j<8
//   f32 : 4 x v_mfma_f32_16x16x4_f32 — step j uses element j of the lane's float4:  A[i = l&15][k = l>>4]
//         (any bijection lane-group/element -> k is legal as long as A and B use the same one)
//   C/D : lane l holds C[row = (l>>4)*4 + r][col = l&15], r<4      (cdna_hip_programming.md §3)
template <class T> __device__ __forceinline__ void mfma16(const uint4& a, const uint4& b, f32x4& c);

#if defined(ACH_HOSTEMU)
template <class T> __device__ inline void mfma16(const uint4& a, const uint4& b, f32x4& c) {
    constexpr int VEC = Store<T>::VEC;
    struct { uint4 a, b; } mine{a, b};
    hostemu::wave_deposit(&mine, sizeof(mine));
    const int lane = hostemu::lane_id();
    const int col = lane & 15;
    float add[4] = {0, 0, 0, 0};
    for (int g = 0; g < 4; ++g) {
        uint4 fb;
        std::memcpy(&fb, static_cast<const unsigned char*>(hostemu::wave_slot(g * 16 + col)) + 16, 16);
        float bv[8];
        frag_unpack<T>(fb, bv);
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) * 4 + r;
            uint4 fa;
            std::memcpy(&fa, hostemu::wave_slot(g * 16 + row), 16);
            float av[8];
            frag_unpack<T>(fa, av);
            for (int j = 0; j < VEC; ++j) add[r] += av[j] * bv[j];
        }
    }
    hostemu::wave_release();
    for (int r = 0; r < 4; ++r) c[r] += add[r];
}
#else
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
// Round 6: the shipped library contains NO v_mfma_f32_16x16x32_{f16,bf16}: every 16-bit k-chunk is two 16x16x16 instructions (the same products, the same matrix-pipe time).
// profiles/r06_coresidency/ffn2_*_table.txt: with two waves of one SIMD interleaving matrix instructions, one of them the 16x16x32 form, whole 16-row tiles of EITHER wave come
// out wrong run to run — also when the victim only issues 16x16x16 pairs, also with 48 wait states behind every matrix instruction of the victim (so it is not a software hazard
// inside a wave), never with one wave per SIMD, never when no wave issues the 16x16x32 form.  The plans run three streams on the same compute units, so no kernel may issue it
// (tests/test_abi_and_host.py checks the ISA of both 16-bit engines).  0 = the one-instruction form (variant builds / measurements only; it measured +1.8 % on EN-S0 in round 5).
#ifndef ACH_MFMA16_SPLIT
#define ACH_MFMA16_SPLIT 1
#endif
// ACH_MFMA_NOP_AFTER (experiments, profiles/scripts/ffn2_bisect.sh): 48 wait states behind every matrix instruction issued through mfma16 / mfma16_pair — a whole
// 8-pass instruction has left the pipe before the wave issues anything else
#ifndef ACH_MFMA_NOP_AFTER
#define ACH_MFMA_NOP_AFTER 0
#endif
__device__ __forceinline__ void mfma_nop_after(f32x4& c) {
#if ACH_MFMA_NOP_AFTER
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(c));
#endif
}
template <> __device__ __forceinline__ void mfma16<bf16_t>(const uint4& a, const uint4& b, f32x4& c) {
#if ACH_MFMA16_SPLIT
    typedef short s16x4_hw __attribute__((ext_vector_type(4)));
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_hw, make_uint2(a.x, a.y)), __builtin_bit_cast(s16x4_hw, make_uint2(b.x, b.y)), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_hw, make_uint2(a.z, a.w)), __builtin_bit_cast(s16x4_hw, make_uint2(b.z, b.w)), c, 0, 0, 0);
#else
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
#endif
    mfma_nop_after(c);
}
typedef _Float16 f16x8_hw __attribute__((ext_vector_type(8)));
template <> __device__ __forceinline__ void mfma16<f16_t>(const uint4& a, const uint4& b, f32x4& c) {
#if ACH_MFMA16_SPLIT
    typedef _Float16 f16x4_hw __attribute__((ext_vector_type(4)));
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_hw, make_uint2(a.x, a.y)), __builtin_bit_cast(f16x4_hw, make_uint2(b.x, b.y)), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_hw, make_uint2(a.z, a.w)), __builtin_bit_cast(f16x4_hw, make_uint2(b.z, b.w)), c, 0, 0, 0);
#else
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), c, 0, 0, 0);
#endif
    mfma_nop_after(c);
}
template <> __device__ __forceinline__ void mfma16<float>(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}
#endif

// The same k-chunk as TWO v_mfma_f32_16x16x16 instructions (k = 0..15, then 16..31: the same products, the same accumulation order per half) — the form every kernel
// with LONG-LIVED matrix-instruction waves must use (row-walking kernels, multi-tile MLP waves): a co-resident wave that issues v_mfma_f32_16x16x32 changes the results of other
// waves' MFMA chunk loops on the MI355X, two 16x16x16 do not (tests/test_gpu_coresidency.py).  Same matrix-pipe passes (2 x 4 instead of 8).
#if defined(ACH_HOSTEMU)
template <class T> __device__ inline void mfma16_pair(const uint4& a, const uint4& b, f32x4& c) { mfma16<T>(a, b, c); }
#else
template <class T> __device__ __forceinline__ void mfma16_pair(const uint4& a, const uint4& b, f32x4& c);
template <> __device__ __forceinline__ void mfma16_pair<bf16_t>(const uint4& a, const uint4& b, f32x4& c) {
    typedef short s16x4_p __attribute__((ext_vector_type(4)));
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_p, make_uint2(a.x, a.y)), __builtin_bit_cast(s16x4_p, make_uint2(b.x, b.y)), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_p, make_uint2(a.z, a.w)), __builtin_bit_cast(s16x4_p, make_uint2(b.z, b.w)), c, 0, 0, 0);
    mfma_nop_after(c);
}
template <> __device__ __forceinline__ void mfma16_pair<f16_t>(const uint4& a, const uint4& b, f32x4& c) {
    typedef _Float16 f16x4_p __attribute__((ext_vector_type(4)));
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_p, make_uint2(a.x, a.y)), __builtin_bit_cast(f16x4_p, make_uint2(b.x, b.y)), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_p, make_uint2(a.z, a.w)), __builtin_bit_cast(f16x4_p, make_uint2(b.z, b.w)), c, 0, 0, 0);
    mfma_nop_after(c);
}
template <> __device__ __forceinline__ void mfma16_pair<float>(const uint4& a, const uint4& b, f32x4& c) { mfma16<float>(a, b, c); }
#endif

// wave-level helpers for 64-bit masks (the CPU emulation goes through its shuffle)
#if defined(ACH_HOSTEMU)
__device__ inline unsigned long long wave_read64(unsigned long long v, int lane) { return __shfl(v, lane); }
__device__ inline unsigned long long wave_ballot64(bool pred) {
    unsigned long long bits = 0;
    const int mine = pred ? 1 : 0;
    for (int l = 0; l < 64; ++l) bits |= (unsigned long long)(__shfl(mine, l)) << l;
    return bits;
}
__device__ inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
__device__ inline int __ffsll(long long v) { return __builtin_ffsll(v); }
#else
__device__ __forceinline__ unsigned long long wave_read64(unsigned long long v, int lane) {
    const unsigned lo = __builtin_amdgcn_readlane(unsigned(v), lane), hi = __builtin_amdgcn_readlane(unsigned(v >> 32), lane);
    return (unsigned long long)(hi) << 32 | lo;
}
__device__ __forceinline__ unsigned long long wave_ballot64(bool pred) { return __ballot(pred); }
#endif

// orders a wave's LDS writes before its own later LDS reads of other lanes' data (a wave's LDS operations execute in order; this
// only stops the compiler from moving them).  The CPU emulation needs a real rendezvous of the 64 fibers.
#if defined(ACH_HOSTEMU)
__device__ inline void wave_sync() { (void)__shfl(0, 0); }
#else
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

// max over each row of 16 lanes (all 16 lanes receive it): four DPP VALU ops instead of four LDS-routed shuffles.
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror pair up lanes / quads / halves of the row.
#if defined(ACH_HOSTEMU)
__device__ inline float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
#else
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
#endif

// Range-checked 16-byte loads through a buffer resource: an offset at or beyond the end of the buffer returns zeros in hardware
// (MUBUF, raw buffer, stride 0), which is how convolution taps that fall outside the map are handled without clamps or masks:
// the tap's byte offset is simply made huge.  `bytes` must be below 2^31 so that "in-range base + 0x80000000" never wraps.
#if defined(ACH_HOSTEMU)
struct BufRsrc { const char* base; unsigned bytes; };
inline BufRsrc make_buf(const void* p, unsigned bytes) { return BufRsrc{static_cast<const char*>(p), bytes}; }
inline uint4 buf_load16(const BufRsrc& r, unsigned off) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r.bytes >= 16u && off <= r.bytes - 16u) std::memcpy(&v, r.base + off, 16);
    return v;
}
inline uint4 buf_load16s(const BufRsrc& r, unsigned voff, unsigned soff) { return buf_load16(r, voff + soff); }
inline uint4 buf_load16_agent(const BufRsrc& r, unsigned off) { return buf_load16(r, off); }
#else
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned int buf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ BufRsrc make_buf(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, int(bytes), 0x00020000);      // gfx9 raw buffer, 32-bit data format
}
__device__ __forceinline__ uint4 buf_load16(BufRsrc r, unsigned off) {
    const buf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, int(off), 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// the same load at AGENT scope (sc1: it must not be served from this compute unit's L1): data another workgroup of the running kernel wrote (k_mlpband.h run kernel)
__device__ __forceinline__ uint4 buf_load16_agent(BufRsrc r, unsigned off) {
    const buf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, int(off), 0, 0x10);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// per-lane offset + wave-uniform offset (the MUBUF soffset operand: no VALU add); the range check applies to their sum
__device__ __forceinline__ uint4 buf_load16s(BufRsrc r, unsigned voff, unsigned soff) {
    const buf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, int(voff), int(soff), 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
#endif
constexpr unsigned BUF_OOB = 0x80000000u;
// 4-byte store through a buffer resource: a per-lane offset at or beyond the end of the buffer (BUF_OOB) is DROPPED in hardware — a store under a per-lane
// condition without an exec mask or a skip branch around it.  Straight-line code is what lets the compiler count the memory operations between a load and its
// use exactly (s_waitcnt vmcnt(N): loads and stores share the counter on gfx9 and retire in order; behind a conditional store it must assume N = 0, i.e.
// wait for every store issued so far to be acknowledged).  `soff` is the wave-uniform part of the address (MUBUF soffset, an SGPR).
#if defined(ACH_HOSTEMU)
inline void buf_store4(const BufRsrc& r, unsigned voff, unsigned soff, uint32_t v) {
    if (voff < r.bytes && r.bytes >= 4u && size_t(voff) + soff <= size_t(r.bytes) - 4u) std::memcpy(const_cast<char*>(r.base) + voff + soff, &v, 4);
}
#else
__device__ __forceinline__ void buf_store4(BufRsrc r, unsigned voff, unsigned soff, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, int(voff), int(soff), 0); }
#endif
// the same for 2 bytes (the low half of `v`) and for 8 bytes
#if defined(ACH_HOSTEMU)
inline void buf_store2(const BufRsrc& r, unsigned voff, unsigned soff, uint32_t v) {
    const uint16_t h = uint16_t(v);
    if (voff < r.bytes && r.bytes >= 2u && size_t(voff) + soff <= size_t(r.bytes) - 2u) std::memcpy(const_cast<char*>(r.base) + voff + soff, &h, 2);
}
inline void buf_store8(const BufRsrc& r, unsigned voff, unsigned soff, uint32_t a, uint32_t b) {
    const uint32_t q[2] = {a, b};
    if (voff < r.bytes && r.bytes >= 8u && size_t(voff) + soff <= size_t(r.bytes) - 8u) std::memcpy(const_cast<char*>(r.base) + voff + soff, q, 8);
}
#else
__device__ __forceinline__ void buf_store2(BufRsrc r, unsigned voff, unsigned soff, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b16(short(v), r, int(voff), int(soff), 0); }
typedef unsigned int buf_st_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void buf_store8(BufRsrc r, unsigned voff, unsigned soff, uint32_t a, uint32_t b) { __builtin_amdgcn_raw_buffer_store_b64(buf_st_u32x2{a, b}, r, int(voff), int(soff), 0); }
#endif
// four consecutive elements of the storage type through a buffer resource (8 bytes of bf16 / 16 bytes of fp32), as floats
#if defined(ACH_HOSTEMU)
inline void buf_load_raw(const BufRsrc& r, unsigned off, void* dst, unsigned n) {
    std::memset(dst, 0, n);
    if (r.bytes >= n && off <= r.bytes - n) std::memcpy(dst, r.base + off, n);
}
template <class T> inline void buf_ld4(const BufRsrc& r, unsigned off, float (&o)[4]);
template <> inline void buf_ld4<float>(const BufRsrc& r, unsigned off, float (&o)[4]) { buf_load_raw(r, off, o, 16); }
template <> inline void buf_ld4<bf16_t>(const BufRsrc& r, unsigned off, float (&o)[4]) {
    uint16_t h[4]; buf_load_raw(r, off, h, 8);
    for (int i = 0; i < 4; ++i) o[i] = bf16_to_f32(h[i]);
}
template <> inline void buf_ld4<f16_t>(const BufRsrc& r, unsigned off, float (&o)[4]) {
    uint16_t h[4]; buf_load_raw(r, off, h, 8);
    for (int i = 0; i < 4; ++i) o[i] = f16_bits_to_f32(h[i]);
}
#else
typedef unsigned int buf_u32x2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ void buf_ld4(BufRsrc r, unsigned off, float (&o)[4]);
template <> __device__ __forceinline__ void buf_ld4<float>(BufRsrc r, unsigned off, float (&o)[4]) {
    const buf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, int(off), 0, 0);
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void buf_ld4<bf16_t>(BufRsrc r, unsigned off, float (&o)[4]) {
    const buf_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, int(off), 0, 0);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void buf_ld4<f16_t>(BufRsrc r, unsigned off, float (&o)[4]) {
    const buf_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, int(off), 0, 0);
    o[0] = f16lo_to_f32(v.x); o[1] = f16hi_to_f32(v.x); o[2] = f16lo_to_f32(v.y); o[3] = f16hi_to_f32(v.y);
}
#endif

// tell the compiler that a value is the same in every lane of the wave (threadIdx.x >> 6 and what is derived from it): branches on it
// become scalar branches and the arithmetic moves to the scalar unit
#if defined(ACH_HOSTEMU)
__device__ inline int wave_uniform(int v) { return v; }
#else
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// A 32-bit per-lane byte offset the optimiser must re-materialise where it is used.  Instruction selection works one basic block at a time:
// `global_load v, v_off, s[base:base+1]` (SGPR base + zero-extended 32-bit VGPR offset) is only matched when the zero-extension sits in the same
// block as the access.  Hoisted out of a loop, the offset becomes a 64-bit VGPR pair and every access a v_lshl_add_u64 plus a 64-bit address pair.
#if defined(ACH_HOSTEMU)
__device__ inline unsigned local_offset(unsigned& v) { return v; }
#else
__device__ __forceinline__ unsigned local_offset(unsigned& v) { asm volatile("" : "+v"(v)); return v; }        // in place: no copy, the variable itself is "redefined" here
#endif

// value of lane `lane` (a compile-time or wave-uniform index) as a scalar
#if defined(ACH_HOSTEMU)
__device__ inline int wave_lane_i32(int v, int lane) { return __shfl(v, lane); }
#else
__device__ __forceinline__ int wave_lane_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
#endif

// wave-uniform max of a float / broadcast of one lane's float (lane must be wave-uniform)
#if defined(ACH_HOSTEMU)
__device__ inline float wave_max_f32(float v) { v = row16_max(v); v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ inline float wave_lane_f32(float v, int lane) { return __shfl(v, lane); }
#else
__device__ __forceinline__ float wave_lane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(lane)));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(wave_lane_f32(v, 0), wave_lane_f32(v, 16)), fmaxf(wave_lane_f32(v, 32), wave_lane_f32(v, 48)));
}
#endif

// ---------------------------------------------------------------------------------------- activations
// v_rcp_f32 / v_exp_f32 based (1 ulp): the IEEE-exact division and expf expansions cost ~10 VALU instructions each and sit
// in the epilogue of bandwidth-bound kernels (measured: +28 us on a 47 us GEMM for GELU with the exact forms)
#if defined(ACH_HOSTEMU)
__device__ inline float fast_rcp(float x) { return 1.0f / x; }
__device__ inline float ln_rstd(float v) { return 1.0f / sqrtf(v); }
__device__ inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ inline float fast_exp(float x) { return expf(x); }
__device__ inline float fast_exp2(float x) { return exp2f(x); }
#else
// ACH_TRANS_PAD (experiments): 8 wait states behind every transcendental instruction issued through these wrappers — the stand-alone reproducer
// (profiles/scripts/ubench/ffn2_coresidency.hip) stops differing with them, i.e. what a co-resident 8-pass matrix instruction disturbs is the hand-over of a
// transcendental result to the instruction that uses it (the compiler inserts ONE wait state there on gfx940 / gfx950: "trans forwarding hazard")
#ifndef ACH_TRANS_PAD
#define ACH_TRANS_PAD 0
#endif
__device__ __forceinline__ float trans_pad(float r) {
#if ACH_TRANS_PAD
    asm volatile("s_nop 7" : "+v"(r));
#endif
    return r;
}
__device__ __forceinline__ float fast_rcp(float x) { return trans_pad(__builtin_amdgcn_rcpf(x)); }
// 1 / sqrt(v) of a LayerNorm / norm statistic: the IEEE form by default; with ACH_TRANS_PAD the hardware reciprocal square root behind the padding
__device__ __forceinline__ float ln_rstd(float v) {
#if ACH_TRANS_PAD
    return trans_pad(__builtin_amdgcn_rsqf(v));
#else
    return 1.0f / sqrtf(v);
#endif
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }      // v_med3_f32
__device__ __forceinline__ float fast_exp(float x) { return trans_pad(__expf(x)); }
__device__ __forceinline__ float fast_exp2(float x) { return trans_pad(__builtin_amdgcn_exp2f(x)); }
#endif
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GELU = 3, ACT_SIGMOID = 4 };

#ifndef ACH_SIGMOID_POLY
#define ACH_SIGMOID_POLY 0           // experiments only (profiles/scripts/ubench/ffn2_coresidency.hip): 1 = a polynomial WITHOUT transcendental instructions (wrong values), 2 = the
#endif                               // real form with 8 wait states behind each transcendental instruction
#if ACH_SIGMOID_POLY == 1
__device__ __forceinline__ float sigmoidf_(float x) { const float c = clampf(x, -3.f, 3.f); return 0.5f + c * (0.25f - c * c * 0.0138f); }
#elif ACH_SIGMOID_POLY == 2
__device__ __forceinline__ float sigmoidf_(float x) {
    float e = fast_exp(-x);
    asm volatile("s_nop 7" : "+v"(e));
    float r = fast_rcp(1.0f + e);
    asm volatile("s_nop 7" : "+v"(r));
    return r;
}
#elif ACH_SIGMOID_POLY == 3          // the same eight wait states IN FRONT of each transcendental instruction (the same slow-down, no protection of the result's first use)
__device__ __forceinline__ float sigmoidf_(float x) {
    float a = -x;
    asm volatile("s_nop 7" : "+v"(a));
    float e = 1.0f + fast_exp(a);
    asm volatile("s_nop 7" : "+v"(e));
    return fast_rcp(e);
}
#elif ACH_SIGMOID_POLY == 6          // NO wait states: empty asm statements that only keep the compiler from pairing the consumers of the transcendental results into packed (v_pk_*) instructions
__device__ __forceinline__ float sigmoidf_(float x) {
    float e = fast_exp(-x);
    asm volatile("" : "+v"(e));
    float r = fast_rcp(1.0f + e);
    asm volatile("" : "+v"(r));
    return r;
}
#elif ACH_SIGMOID_POLY == 4 || ACH_SIGMOID_POLY == 5          // one / two extra wait states behind each transcendental instruction
__device__ __forceinline__ float sigmoidf_(float x) {
    float e = fast_exp(-x);
#if ACH_SIGMOID_POLY == 4
    asm volatile("s_nop 0" : "+v"(e));
#else
    asm volatile("s_nop 1" : "+v"(e));
#endif
    float r = fast_rcp(1.0f + e);
#if ACH_SIGMOID_POLY == 4
    asm volatile("s_nop 0" : "+v"(r));
#else
    asm volatile("s_nop 1" : "+v"(r));
#endif
    return r;
}
#else
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
#endif
// erf by Abramowitz & Stegun 7.1.26: |error| <= 1.5e-7 (fp32 epsilon level), one exp + one reciprocal + 6 FMA
// (the library erff is ~3x the VALU work; GELU runs on every hidden unit of every EdgeNeXt MLP)
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = fast_rcp(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float r = 1.0f - poly * fast_exp(-ax * ax);
    return x < 0.f ? -r : r;
}
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_SILU: return x * sigmoidf_(x);
        case ACT_GELU: return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));   // erf GELU (nn.GELU default)
        case ACT_SIGMOID: return sigmoidf_(x);
        default: return x;
    }
}

// GELU for bf16 storage: x * sigmoid(a x + b x^3) with (a, b) fitted to the erf form — max |error| 2.7e-4 over all x, a
// fifteenth of a bf16 ulp at 1.0 — in 5 VALU ops + exp + rcp instead of 16 + exp + rcp.  The MLP kernels of EdgeNeXt are
// VALU-issue bound and evaluate this on every hidden unit.  fp32 storage (the parity engine) keeps the erf form.
__device__ __forceinline__ float gelu_sigmoid(float x) {
    const float x2 = x * x;
    const float u = x * (-0.10012562f * x2 - 2.30876570f);            // -(a x + b x^3) * log2(e)
    return x * fast_rcp(1.0f + fast_exp2(u));
}
template <class T> __device__ __forceinline__ float apply_act_t(float x, int act) { return apply_act(x, act); }
template <> __device__ __forceinline__ float apply_act_t<bf16_t>(float x, int act) { return act == ACT_GELU ? gelu_sigmoid(x) : apply_act(x, act); }
template <> __device__ __forceinline__ float apply_act_t<f16_t>(float x, int act) { return act == ACT_GELU ? gelu_sigmoid(x) : apply_act(x, act); }
// N values at once with the switch on the (launch-uniform) activation OUTSIDE the element loop.  Per element, the compiler kept a
// scalar branch ladder around every value: eight serialised (bias load -> wait -> ladder -> exp -> rcp) chains per hidden chunk of
// mlp_kernel, no two transcendentals ever in flight together.  Same formulas as apply_act_t, so results are bit-identical.
// ACT >= 0 fixes the activation at compile time (no branch at all: the surrounding loop stays one schedulable block).
template <class T, int N, int ACT = -1>
__device__ __forceinline__ void apply_act_n(float* v, int act) {
    const int a = ACT >= 0 ? ACT : act;
    switch (a) {
        case ACT_RELU: ACH_UNROLL for (int i = 0; i < N; ++i) v[i] = apply_act(v[i], ACT_RELU); break;
        case ACT_SILU: ACH_UNROLL for (int i = 0; i < N; ++i) v[i] = apply_act(v[i], ACT_SILU); break;
        case ACT_GELU:
            if constexpr (sizeof(T) == 2 && N % 2 == 0) {
                // bf16 storage: gelu_sigmoid on PAIRS — the five plain operations of a value as packed fp32 (v_pk_mul / v_pk_fma / v_pk_add: two
                // values per issue), only exp2 and rcp stay per value: 4.5 instead of 7 VALU issues per hidden unit.  Same operations in the
                // same order per element as gelu_sigmoid.
                ACH_UNROLL
                for (int i = 0; i < N; i += 2) {
                    const f32x2 x = {v[i], v[i + 1]};
                    const f32x2 x2 = x * x;
                    const f32x2 u = x * (f32x2{-0.10012562f, -0.10012562f} * x2 - f32x2{2.30876570f, 2.30876570f});
                    const f32x2 d = f32x2{1.0f, 1.0f} + f32x2{fast_exp2(u[0]), fast_exp2(u[1])};
                    const f32x2 y = x * f32x2{fast_rcp(d[0]), fast_rcp(d[1])};
                    v[i] = y[0]; v[i + 1] = y[1];
                }
            } else {
                ACH_UNROLL for (int i = 0; i < N; ++i) v[i] = apply_act_t<T>(v[i], ACT_GELU);
            }
            break;
        case ACT_SIGMOID: ACH_UNROLL for (int i = 0; i < N; ++i) v[i] = apply_act(v[i], ACT_SIGMOID); break;
        default: break;
    }
}

// XCD-aware workgroup order for kernels whose neighbouring workgroups share input lines (3x3 halos, bilinear corners):
// workgroup w is observed to run on XCD w % 8, each XCD with its own L2.  Re-numbering w -> (w % 8) * (n / 8) + w / 8 hands
// every XCD one contiguous run of tiles, so shared lines are fetched from HBM once instead of once per XCD.  Speed only:
// results never depend on placement.  (Measured with rocprofv3 FETCH_SIZE: 2.0x -> 1.0x algorithmic bytes on the decoder kernel.)
__device__ __forceinline__ unsigned xcd_block(unsigned w, unsigned n) { return (n % 8 == 0) ? (w % 8) * (n / 8) + w / 8 : w; }

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ long cdivl(long a, long b) { return (a + b - 1) / b; }

}  // namespace ach
