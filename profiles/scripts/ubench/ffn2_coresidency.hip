// ffn2_coresidency.hip — stand-alone reproducer (no engine, no torch) of DESIGN 4.20's second instance of the 16x16x32 effect, with the library's own kernel as
// the victim: ffn2_kernel (achelous_amd/csrc/k_mlp.h: LayerNorm -> fc1 -> SiLU -> fc2 -> + input, two 16-row tiles per wave, 9 hidden chunks of ~40 matrix
// instructions per wave) on random rows and random weight fragments, run alone (the reference) and then beside an aggressor on a second stream, compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../../achelous_amd/csrc -DACH_FFN2_MFMA32=1 -o ffn2_mfma32 ffn2_coresidency.hip && ./ffn2_mfma32      (the rejected form)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../../achelous_amd/csrc                      -o ffn2_pair   ffn2_coresidency.hip && ./ffn2_pair        (the shipped form)
// Aggressors: none (the victim against itself, run to run) | `busy`: single-wave workgroups spinning on dependent VALU operations (no matrix instruction, no memory) |
// `mfma32`: the same with one v_mfma_f32_16x16x32_f16 per 20 operations.  Output: runs (of `rounds`) whose output differed from the reference, and in how many elements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "k_mlp.h"

using namespace ach;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
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
        if (FORM == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] + v == 123456.789f) sink[0] = v;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    // optional: rows, hidden chunks, LayerNorm on / off, activation (2 SiLU, 1 ReLU, 0 none), residual on / off — to see which part of the kernel the effect needs
    const int C = 144, hidden = 288, DT = 10, k1 = 5, J = argc > 3 ? atoi(argv[3]) : 9;
    const long M = argc > 2 ? atol(argv[2]) : 64L * 40 * 40;       // default: MV-S2's first transformer at batch 64
    const int use_ln = argc > 4 ? atoi(argv[4]) : 1, use_act = argc > 5 ? atoi(argv[5]) : int(ACT_SILU), use_res = argc > 6 ? atoi(argv[6]) : 1;
    std::vector<uint16_t> hx(size_t(M) * C), w1(size_t(J) * k1 * 2 * 64 * 8), w2(size_t(J) * DT * 64 * 8);
    std::vector<float> b1(size_t(J) * 32), b2(size_t(DT) * 16);
    srand(7);
    auto rh = [&](float s) { const float f = (float(rand()) / float(RAND_MAX) - 0.5f) * s; return f32_to_f16_bits(f); };
    for (auto& v : hx) v = rh(4.f);
    for (auto& v : w1) v = rh(0.3f);
    for (auto& v : w2) v = rh(0.2f);
    for (auto& v : b1) v = (float(rand()) / float(RAND_MAX) - 0.5f) * 0.2f;
    for (auto& v : b2) v = (float(rand()) / float(RAND_MAX) - 0.5f) * 0.2f;
    void *dx, *dy, *dw1, *dw2, *dref; float *db1, *db2, *sink;
    CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dy, hx.size() * 2)); CK(hipMalloc(&dref, hx.size() * 2));
    CK(hipMalloc(&dw1, w1.size() * 2)); CK(hipMalloc(&dw2, w2.size() * 2)); CK(hipMalloc(&db1, b1.size() * 4)); CK(hipMalloc(&db2, b2.size() * 4)); CK(hipMalloc(&sink, 16));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
    MlpParams p;
    memset(&p, 0, sizeof(p));
    p.X = dx; p.ldx = C; p.R = use_res ? dx : nullptr; p.ldr = C; p.Y = dy; p.ldy = C; p.W1 = dw1; p.b1 = db1; p.W2 = dw2; p.b2 = db2;
    p.M = M; p.C = C; p.k1 = k1; p.J = J; p.act = use_act; p.ln_eps = 1e-5f; p.ln = use_ln; p.Cout = C;
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    const long tiles = (M + 15) / 16;
    const dim3 grid(unsigned((tiles + 7) / 8)), block(256);
    auto victim = [&](void* out) { p.Y = out; hipLaunchKernelGGL((ffn2_kernel<f16_t, 10>), grid, block, 0, sv, p); };
    victim(dref);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> ref(hx.size()), got(hx.size());
    CK(hipMemcpy(ref.data(), dref, ref.size() * 2, hipMemcpyDeviceToHost));
    printf("ffn2_kernel<f16, 10> (ACH_FFN2_MFMA32 = %d), %ld rows x %d channels, J = %d, ln %d, act %d, residual %d, %d rounds per aggressor\n", ACH_FFN2_MFMA32, M, C, J, use_ln, use_act, use_res, rounds);
    const char* names[3] = {"none", "busy (no matrix instruction)", "mfma32 (one 16x16x32 per 20 VALU)"};
    for (int ag = 0; ag < 3; ++ag) {
        int bad_runs = 0; long bad_elems = 0;
        for (int r = 0; r < rounds; ++r) {
            CK(hipMemsetAsync(dy, 0, hx.size() * 2, sv));
            CK(hipStreamSynchronize(sv));
            if (ag == 1) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(spin<0>, dim3(4096), dim3(64), 0, sa, 1500, 20, 0.25f, sink);
            if (ag == 2) for (int q = 0; q < 2; ++q) hipLaunchKernelGGL(spin<1>, dim3(4096), dim3(64), 0, sa, 1500, 20, 0.25f, sink);
            victim(dy);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), dy, got.size() * 2, hipMemcpyDeviceToHost));
            long n = 0;
            for (size_t i = 0; i < got.size(); ++i) n += got[i] != ref[i];
            bad_runs += n != 0; bad_elems += n;
        }
        printf("  aggressor %-36s: %d of %d runs differ from the run alone (%ld elements in all)\n", names[ag], bad_runs, rounds, bad_elems);
    }
    return 0;
}
