// mfma32_coresidency.hip — stand-alone two-kernel reproducer of DESIGN 4.15 (no engine, no torch): does a wave that issues v_mfma_f32_16x16x32_{f16,bf16} change the
// RESULTS of another kernel's waves on the same compute unit?
//   hipcc --offload-arch=gfx950 -O3 -o mfma32_coresidency mfma32_coresidency.hip && ./mfma32_coresidency
//
// AGGRESSOR (stream 1): `spin<FORM>` — single-wave workgroups that live for ~100 us and issue one matrix instruction per `gap` dependent VALU operations; no memory
//   traffic, no LDS, no DPP.  FORM 0: v_mfma_f32_16x16x32_f16 | 1: two v_mfma_f32_16x16x16_f16 (the same product) | 2: v_mfma_f32_16x16x32_bf16 | 3: no matrix instruction.
// VICTIM (stream 0): `victim<KIND>` — every wave computes the SAME small GEMM tile `reps` times from operands that depend on the lane only, and compares each
//   repetition bit for bit with its first one (a wave checks itself: no host reference, no rounding question).  KIND picks the instruction pattern:
//     0  CHAIN     klen dependent back-to-back v_mfma_f32_16x16x32_f16 on ONE accumulator (a k-loop with one output tile per wave)
//     1  INTERLV   the same k-loop over FOUR independent accumulators, interleaved (four output tiles per wave: dependent MFMAs are three instructions apart)
//     2  CHAIN16   the chain as v_mfma_f32_16x16x16_f16 pairs
//     3  LOADED    CHAIN with the A operand of every k-step loaded from global memory just in time (what a weight-streaming kernel does)
//     4  VALUGAP   CHAIN with ~40 VALU operations between the dependent MFMAs (what a fused GEMM1 -> activation -> GEMM2 kernel does)
//     5  SCRATCH   NO matrix instruction: 64 values per lane written to the wave's PRIVATE (scratch) memory and read back (what a kernel with register spills does)
//     6  SCRMFMA   CHAIN whose accumulators make a round trip through scratch between the k-steps
//     7  MLP       the hidden-chunk loop of the library's fused ConvEncoder kernel (k_mlp.h, the kernel the effect was first localised to): per chunk GEMM1 = 2 k-steps x 2
//                  accumulators with the weights loaded from global memory, + bias, ReLU, packed to fp16 = the B operand of GEMM2 into four accumulators — at the
//                  register budget of that kernel (6 workgroups per CU: accumulators in architectural VGPRs, no AGPRs)
//   The table printed at the end: per (victim kind, aggressor form) the number of victim waves with at least one repetition that differed, of how many.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int FORM>
__global__ __launch_bounds__(64, 2) void spin(int iters, int gap, float seed, float* sink) {
    const float l = float(threadIdx.x) * 0.001f + seed;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = _Float16(l + 0.01f * e); b[e] = _Float16(0.5f - 0.02f * e - l); }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    float v = l;
    for (int i = 0; i < iters; ++i) {
        for (int q = 0; q < gap; ++q) v = __builtin_fmaf(v, 0.999f, 0.001f);
        b[0] = _Float16(v);
        if (FORM == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        else if (FORM == 1) {
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h4{a[0], a[1], a[2], a[3]}, h4{b[0], b[1], b[2], b[3]}, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h4{a[4], a[5], a[6], a[7]}, h4{b[4], b[5], b[6], b[7]}, acc, 0, 0, 0);
        } else if (FORM == 2) {
            b8 ab, bb;
            for (int e = 0; e < 8; ++e) { ab[e] = __bf16(float(a[e])); bb[e] = __bf16(float(b[e])); }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc, 0, 0, 0);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] + v == 123456.789f) sink[0] = v;
}

constexpr int KLEN = 6;
__device__ __forceinline__ h8 operand(int lane, int k, int which) {
    h8 r;
    for (int e = 0; e < 8; ++e) r[e] = _Float16(float(((lane * 7 + k * 13 + e * 3 + which * 5) % 31) - 15) * 0.0625f);
    return r;
}

template <int KIND>
__global__ __launch_bounds__(256) void victim(int reps, const h8* __restrict__ wmem, unsigned* __restrict__ bad_waves, unsigned* __restrict__ bad_reps, float* __restrict__ first_out) {
    const int lane = int(threadIdx.x) & 63;
    const int wave = (int(blockIdx.x) * int(blockDim.x) + int(threadIdx.x)) >> 6;
    h8 A[KLEN], B[KLEN];
    for (int k = 0; k < KLEN; ++k) { A[k] = operand(lane, k, 0); B[k] = operand(lane, k, 1); }
    f4 ref[4];
    unsigned nbad = 0;
    for (int r = 0; r < reps; ++r) {
        f4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // (the repetition index enters through a value the compiler cannot fold, so the loop body is really executed `reps` times)
        float z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));        // an opaque zero (its consumers are compiler-visible VALU instructions, never an MFMA directly)
        const _Float16 tw = _Float16(z);
        if (KIND == 0) {
            for (int k = 0; k < KLEN; ++k) { h8 b = B[k]; b[0] += tw; acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[k], b, acc[0], 0, 0, 0); }
        } else if (KIND == 1) {
            for (int k = 0; k < KLEN; ++k) {
                h8 b = B[k]; b[0] += tw;
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(k + t) % KLEN], b, acc[t], 0, 0, 0);
            }
        } else if (KIND == 2) {
            for (int k = 0; k < KLEN; ++k) {
                h8 b = B[k]; b[0] += tw;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x16f16(h4{A[k][0], A[k][1], A[k][2], A[k][3]}, h4{b[0], b[1], b[2], b[3]}, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x16f16(h4{A[k][4], A[k][5], A[k][6], A[k][7]}, h4{b[4], b[5], b[6], b[7]}, acc[0], 0, 0, 0);
            }
        } else if (KIND == 3) {
            for (int k = 0; k < KLEN; ++k) {
                h8 b = B[k]; b[0] += tw;
                const h8 a = wmem[(k * 64 + lane) + 64 * KLEN * (r & 3)];           // four copies of the same fragments: a fresh address every repetition
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[0], 0, 0, 0);
            }
        } else if (KIND == 5 || KIND == 6) {
            // private array with lane-dependent dynamic indices: lives in scratch memory (ISA: scratch_store / scratch_load)
            float priv[64];
            const int rot = (lane * 5 + r) & 63;
            const float wv = float(wave & 1023) * 0.5f;                // wave-dependent data: two waves sharing a scratch region would see each other's values
            for (int i = 0; i < 64; ++i) priv[(i + rot) & 63] = float(i * 3 + lane) + wv + z;
            if (KIND == 6) {
                for (int k = 0; k < KLEN; ++k) {
                    h8 b = B[k]; b[0] += tw;
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[k], b, acc[0], 0, 0, 0);
                    for (int e = 0; e < 4; ++e) priv[(e + 8 * k + rot) & 63] += acc[0][e];
                    for (int e = 0; e < 4; ++e) acc[0][e] = priv[(e + 8 * k + rot) & 63] - (float((e + 8 * k) * 3 + lane) + wv);
                }
            }
            float sum = 0.f;
            for (int i = 0; i < 64; ++i) sum += priv[(i + rot) & 63] * float((i & 7) + 1);
            acc[1][0] = sum;
        } else {
            float v = float(tw);
            for (int k = 0; k < KLEN; ++k) {
                h8 b = B[k]; b[0] += tw;
                for (int q = 0; q < 40; ++q) v = __builtin_fmaf(v, 0.999f, 0.0f);
                b[1] += _Float16(v);                                                 // (v stays 0: the chain only has to be there)
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[k], b, acc[0], 0, 0, 0);
            }
        }
        if (r == 0) { for (int t = 0; t < 4; ++t) ref[t] = acc[t]; }
        else {
            bool same = true;
            for (int t = 0; t < 4; ++t) for (int e = 0; e < 4; ++e) same &= (__builtin_bit_cast(uint32_t, acc[t][e]) == __builtin_bit_cast(uint32_t, ref[t][e]));
            if (!__builtin_amdgcn_readfirstlane(int(__ballot(!same) == 0))) ++nbad;
        }
    }
    if (lane == 0 && nbad) { atomicAdd(bad_waves, 1u); atomicAdd(bad_reps, nbad); }
    if (wave == 0) for (int e = 0; e < 4; ++e) first_out[lane * 4 + e] = ref[0][e];
}

__global__ __launch_bounds__(256, 6) void victim_mlp(int reps, const h8* __restrict__ wmem, unsigned* __restrict__ bad_waves, unsigned* __restrict__ bad_reps, float* __restrict__ first_out) {
    constexpr int J = 6, DT = 4;
    const int lane = int(threadIdx.x) & 63;
    const int wave = (int(blockIdx.x) * int(blockDim.x) + int(threadIdx.x)) >> 6;
    const h8 x0 = operand(lane, 1, 1), x1 = operand(lane, 2, 1);
    const h8* W = wmem + lane;                                   // fragments [k][64 lanes]; reused cyclically (KLEN * 4 of them)
    f4 ref[DT];
    unsigned nbad = 0;
    for (int r = 0; r < reps; ++r) {
        float z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        h8 xa = x0, xb = x1; xa[0] += _Float16(z); xb[0] += _Float16(z);
        f4 acc2[DT] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int j = 0; j < J; ++j) {
            f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            const h8* w1 = W + ((j * 4) % (KLEN * 4 - 4)) * 64;
            const h8 wa0 = w1[0], wb0 = w1[64], wa1 = w1[128], wb1 = w1[192];
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa0, xa, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb0, xa, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa1, xb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1, xb, a1, 0, 0, 0);
            h8 hf;
            for (int e = 0; e < 4; ++e) {
                const float u = a0[e] * 0.01f + 0.125f * float(e - 1), v = a1[e] * 0.01f - 0.125f * float(e - 2);
                hf[e] = _Float16(u > 0.f ? u : 0.f); hf[4 + e] = _Float16(v > 0.f ? v : 0.f);
            }
            const h8* w2 = W + ((j * DT + 5) % (KLEN * 4 - DT)) * 64;
            for (int t = 0; t < DT; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[t * 64], hf, acc2[t], 0, 0, 0);
        }
        if (r == 0) { for (int t = 0; t < DT; ++t) ref[t] = acc2[t]; }
        else {
            bool same = true;
            for (int t = 0; t < DT; ++t) for (int e = 0; e < 4; ++e) same &= (__builtin_bit_cast(uint32_t, acc2[t][e]) == __builtin_bit_cast(uint32_t, ref[t][e]));
            if (!__builtin_amdgcn_readfirstlane(int(__ballot(!same) == 0))) ++nbad;
        }
    }
    if (lane == 0 && nbad) { atomicAdd(bad_waves, 1u); atomicAdd(bad_reps, nbad); }
    if (wave == 0) for (int e = 0; e < 4; ++e) first_out[lane * 4 + e] = ref[0][e];
}

template <int FORM> static void launch_spin(hipStream_t s, int wgs, int iters, int gap, float* sink) { hipLaunchKernelGGL(spin<FORM>, dim3(wgs), dim3(64), 0, s, iters, gap, 0.25f, sink); }
static void spin_form(int form, hipStream_t s, int wgs, int iters, int gap, float* sink) {
    if (form == 0) launch_spin<0>(s, wgs, iters, gap, sink); else if (form == 1) launch_spin<1>(s, wgs, iters, gap, sink);
    else if (form == 2) launch_spin<2>(s, wgs, iters, gap, sink); else launch_spin<3>(s, wgs, iters, gap, sink);
}
template <int KIND> static void launch_victim(hipStream_t s, int wgs, int reps, const h8* w, unsigned* bw, unsigned* br, float* fo) { hipLaunchKernelGGL(victim<KIND>, dim3(wgs), dim3(256), 0, s, reps, w, bw, br, fo); }
static void victim_kind(int kind, hipStream_t s, int wgs, int reps, const h8* w, unsigned* bw, unsigned* br, float* fo) {
    switch (kind) { case 0: launch_victim<0>(s, wgs, reps, w, bw, br, fo); break; case 1: launch_victim<1>(s, wgs, reps, w, bw, br, fo); break; case 2: launch_victim<2>(s, wgs, reps, w, bw, br, fo); break;
                    case 3: launch_victim<3>(s, wgs, reps, w, bw, br, fo); break; case 4: launch_victim<4>(s, wgs, reps, w, bw, br, fo); break;
                    case 5: launch_victim<5>(s, wgs, reps, w, bw, br, fo); break; case 6: launch_victim<6>(s, wgs, reps, w, bw, br, fo); break;
                    default: hipLaunchKernelGGL(victim_mlp, dim3(wgs), dim3(256), 0, s, reps, w, bw, br, fo); }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    float *sink, *first; unsigned *bw, *br; h8* wmem;
    CK(hipMalloc(&sink, 16)); CK(hipMalloc(&first, 64 * 4 * 4)); CK(hipMalloc(&bw, 4)); CK(hipMalloc(&br, 4));
    // A fragments in memory for KIND 3: wmem[k * 64 + lane] = operand(lane, k, 0), four copies
    std::vector<_Float16> hw(size_t(4) * KLEN * 64 * 8);
    for (int c = 0; c < 4; ++c) for (int k = 0; k < KLEN; ++k) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e)
        hw[((size_t(c) * KLEN + k) * 64 + l) * 8 + e] = _Float16(float(((l * 7 + k * 13 + e * 3) % 31) - 15) * 0.0625f);
    CK(hipMalloc(&wmem, hw.size() * 2)); CK(hipMemcpy(wmem, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const char* kinds[8] = {"CHAIN", "INTERLV", "CHAIN16", "LOADED", "VALUGAP", "SCRATCH", "SCRMFMA", "MLP"};
    const char* forms[5] = {"alone", "16x16x32_f16", "2x16x16x16_f16", "16x16x32_bf16", "no-mfma"};
    const int victim_wgs = 2048, reps = 400;                 // 8192 victim waves x 400 repetitions per launch
    printf("victim waves with a repetition that differed from the wave's own first one / victim waves (repetitions that differed), %d rounds per cell\n", rounds);
    printf("%-10s", "victim");
    for (int f = 0; f < 5; ++f) printf(" | %-26s", forms[f]);
    printf("\n");
    for (int kind = 0; kind < 8; ++kind) {
        printf("%-10s", kinds[kind]);
        for (int f = -1; f < 4; ++f) {
            unsigned long long waves_bad = 0, reps_bad = 0;
            for (int r = 0; r < rounds; ++r) {
                CK(hipMemsetAsync(bw, 0, 4, sv)); CK(hipMemsetAsync(br, 0, 4, sv));
                CK(hipStreamSynchronize(sv));
                if (f >= 0) for (int q = 0; q < 3; ++q) spin_form(f, sa, 4096, 1500, 20, sink);        // ~3 x 0.6 ms of aggressor waves beside one victim launch
                victim_kind(kind, sv, victim_wgs, reps, wmem, bw, br, first);
                CK(hipDeviceSynchronize());
                unsigned a = 0, b = 0;
                CK(hipMemcpy(&a, bw, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, br, 4, hipMemcpyDeviceToHost));
                waves_bad += a; reps_bad += b;
            }
            char cell[64];
            snprintf(cell, sizeof cell, "%llu / %llu (%llu)", waves_bad, (unsigned long long)rounds * victim_wgs * 4, reps_bad);
            printf(" | %-26s", cell);
            fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}
