// f16_ovfl.hip — what v_cvt_pk_f16_f32 does with values beyond the fp16 range, with and without MODE.FP16_OVFL (bit 23 of HW_REG_MODE).
//   hipcc --offload-arch=gfx950 -O3 -o f16_ovfl f16_ovfl.hip && ./f16_ovfl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void cvt(const float* a, uint32_t* o, int sat) {
    if (sat) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE = 1, offset 23, width 1) = 1
    const int i = threadIdx.x;
    const f2 v = {a[2 * i], a[2 * i + 1]};
    o[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}
int main() {
    const float h[8] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 70000.0f, 1e9f, -1e9f, __builtin_inff()};
    float* d; uint32_t* o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int sat = 0; sat < 2; ++sat) {
        hipLaunchKernelGGL(cvt, dim3(1), dim3(4), 0, 0, d, o, sat);
        uint32_t r[4]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", sat);
        for (int i = 0; i < 4; ++i) printf(" %04x %04x", r[i] & 0xffff, r[i] >> 16);
        printf("\n");
    }
    return 0;
}
