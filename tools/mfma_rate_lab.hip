// mfma_rate_lab: cycles per MFMA per SIMD of the forms the reference-order kernels could use (back-to-back issue, independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(l + i); b[i] = (_Float16)(l - i); }
    i32x4 ia = {l, l + 1, l + 2, l + 3}, ib = {l, l - 1, l - 2, l - 3};
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0 = {}, d1 = {};
    i32x4 e0 = {0, 0, 0, 0}, e1 = e0, e2 = e0, e3 = e0;
    i32x16 g0 = {}, g1 = {};
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {   // f32_16x16x32_f16
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        } else if (KIND == 1) {   // f32_32x32x16_f16
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
        } else if (KIND == 2) {   // i32_16x16x32_i8 (K = 32 form)
            const long la = ((long)ia.x << 32) | (unsigned)ia.y, lb = ((long)ib.x << 32) | (unsigned)ib.y;
            e0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, e1, 0, 0, 0);
            e2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, e2, 0, 0, 0); e3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, e3, 0, 0, 0);
        } else if (KIND == 3) {   // i32_16x16x64_i8
            e0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, e1, 0, 0, 0);
            e2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, e2, 0, 0, 0); e3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, e3, 0, 0, 0);
        } else {   // i32_32x32x32_i8
            g0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, g0, 0, 0, 0); g1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, g0, 0, 0, 0); g1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, g1, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float r = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + (float)(e0[0] + e1[1] + e2[2] + e3[3] + g0[0] + g1[7]);
    out[blockIdx.x * blockDim.x + l] = r;
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> int run(const char* name, int waves_per_simd) {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 8));
    const int iters = 4000, blocks = 256 * waves_per_simd;
    k<KIND><<<blocks, 256>>>(out, iters, cyc);
    k<KIND><<<blocks, 256>>>(out, iters, cyc);
    CK(hipDeviceSynchronize());
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-22s %d wave(s)/SIMD: %6.1f cycles per MFMA per wave, %6.1f per SIMD\n", name, waves_per_simd, (double)h / (iters * 4), (double)h / (iters * 4) / waves_per_simd);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}
int main() {
    for (int w = 1; w <= 2; w++) {
        run<0>("f32_16x16x32_f16", w); run<1>("f32_32x32x16_f16", w); run<2>("i32_16x16x32_i8", w); run<3>("i32_16x16x64_i8", w); run<4>("i32_32x32x32_i8", w);
    }
    return 0;
}
