// Micro-benchmark (MI355X): issue rate of v_fma_f32, v_pk_fma_f32, v_fmac_f32 with a DPP row shift on src0, v_mov_b32_dpp.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// Each kernel runs ITER x 64 independent-chain instructions per wave; one wave per SIMD x WAVES waves; reports cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void k(float* out, float s) {
    float a[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = float(threadIdx.x + i);
    for (int i = 0; i < 8; ++i) p[i] = f2{float(threadIdx.x + i), float(i)};
    const float w = s;
    const f2 w2 = {s, s + 1.f};
    long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(w2));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(w));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(w));
        }
    }
    long t1 = clock64();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = float(t1 - t0);
}

int main() {
    float* d; CHECK(hipMalloc(&d, 1 << 24));
    const char* names[5] = {"v_fma_f32 x16", "v_pk_fma_f32 x8 (16 FMA)", "v_fmac_f32_dpp row_shr:1 x16", "v_mov_b32_dpp x16", "v_fmac_f32_e32 x16"};
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
            const int blocks = 256 * (256 * waves_per_simd / threads);
            auto launch = [&] {
                switch (mode) { case 0: k<0><<<blocks, threads>>>(d, 1.0001f); break; case 1: k<1><<<blocks, threads>>>(d, 1.0001f); break;
                                case 2: k<2><<<blocks, threads>>>(d, 1.0001f); break; case 3: k<3><<<blocks, threads>>>(d, 1.0001f); break; default: k<4><<<blocks, threads>>>(d, 1.0001f); }
            };
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            float clk; CHECK(hipMemcpy(&clk, d, 4, hipMemcpyDeviceToHost));
            const int per_iter = mode == 1 ? 8 : 16;
            // wall-clock estimate: instructions per SIMD = waves_per_simd * ITER * per_iter
            printf("waves/SIMD %d  %-32s  %.2f ms  s_memtime ticks/instr (wave 0) %.2f   ns per wave-instr per SIMD %.3f\n", waves_per_simd, names[mode], ms,
                   clk / (float(ITER) * per_iter), ms * 1e6 / (double(waves_per_simd) * ITER * per_iter));
        }
    }
    return 0;
}
