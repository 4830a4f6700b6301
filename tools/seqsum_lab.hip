// tools/seqsum_lab.hip -- the sampler's two accumulation kernels (jh_kernels.h: sample_sum_kernel / sample_pick_kernel, scheme
// of jh_seqsum.h) on softmax-shaped values at V = 128256: time per launch against the elements per lane and iteration, checked
// against the plain loop on the host.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/seqsum_lab.hip -o tools/seqsum_lab
#include "../jlama_amd/csrc/jh_kernels.h"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
using namespace jh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
template <int T, int E>
static int run(const char* what, const std::vector<float>& v, float u, float* dv, float* dsum, float* du, DecodeState* dst, int* dtok) {
    const int V = (int)v.size();
    float sum = 0.0f;
    for (float x : v) sum += x;
    std::vector<float> y(V);
    float acc = 0.0f;
    int pick = V - 1;
    for (int i = 0; i < V; i++) y[i] = v[i] / sum;
    for (int i = 0; i < V; i++) { acc += y[i]; if (acc >= u) { pick = i; break; } }
    CK(hipMemcpy(du, &u, 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_sum_kernel<T, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_sample(T, E)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_pick_kernel<T, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_sample(T, E)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms_sum = 0, ms_pick = 0, got_sum = 0;
    int got_pick = -1;
    const int reps = 50;
    for (int pass = 0; pass < 2; pass++) {
        CK(hipMemcpy(dv, pass == 0 ? v.data() : y.data(), (size_t)V * 4, hipMemcpyHostToDevice));
        for (int r = 0; r < reps + 5; r++) {
            if (r == 5) CK(hipEventRecord(a, 0));
            if (pass == 0) hipLaunchKernelGGL((sample_sum_kernel<T, E>), dim3(1), dim3(T), lds_bytes_sample(T, E), 0, (const float*)dv, V, (const DecodeState*)dst, dsum);
            else hipLaunchKernelGGL((sample_pick_kernel<T, E>), dim3(1), dim3(T), lds_bytes_sample(T, E), 0, (const float*)dv, V, (const float*)du, (const DecodeState*)dst, dtok);
        }
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(pass == 0 ? &ms_sum : &ms_pick, a, b));
    }
    CK(hipMemcpy(&got_sum, dsum, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&got_pick, dtok, 4, hipMemcpyDeviceToHost));
    printf("%-28s T %4d E %2d: sum %7.2f us  pick %7.2f us   sum %s pick %s (pick at %d of %d)\n", what, T, E, ms_sum / reps * 1e3, ms_pick / reps * 1e3,
           seq_bits(got_sum) == seq_bits(sum) ? "ok" : "MISMATCH", got_pick == pick ? "ok" : "MISMATCH", pick, V);
    return 0;
}
int main() {
    const int V = 128256;
    float *dv, *dsum, *du;
    DecodeState* dst;
    int* dtok;
    CK(hipMalloc(&dv, (size_t)(V + 64) * 4)); CK(hipMalloc(&dsum, 64)); CK(hipMalloc(&du, 64)); CK(hipMalloc(&dst, sizeof(DecodeState))); CK(hipMalloc(&dtok, 64));
    CK(hipMemset(dst, 0, sizeof(DecodeState)));
    std::mt19937_64 rng(7);
    std::normal_distribution<double> N01(0.0, 1.0);
    for (double T : {0.8, 0.05, 5.0}) {
        std::vector<double> l(V);
        double mx = -1e300;
        for (auto& x : l) { x = 3.0 * N01(rng); mx = std::max(mx, x); }
        std::vector<float> v(V);
        for (int i = 0; i < V; i++) v[i] = (float)std::exp((l[i] - mx) / T);
        char what[64];
        for (float u : {0.97f}) {
            snprintf(what, sizeof what, "softmax T %.2f u %.2f", T, u);
            if (run<1024, 16>(what, v, u, dv, dsum, du, dst, dtok)) return 1;
            if (run<512, 16>(what, v, u, dv, dsum, du, dst, dtok)) return 1;
            if (run<512, 32>(what, v, u, dv, dsum, du, dst, dtok)) return 1;
            if (run<256, 32>(what, v, u, dv, dsum, du, dst, dtok)) return 1;
            if (run<256, 64>(what, v, u, dv, dsum, du, dst, dtok)) return 1;
            if (run<128, 64>(what, v, u, dv, dsum, du, dst, dtok)) return 1;
        }
    }
    std::vector<float> ones(V, 1.0f);
    if (run<1024, 16>("all equal", ones, 0.9f, dv, dsum, du, dst, dtok)) return 1;
    if (run<256, 32>("all equal", ones, 0.9f, dv, dsum, du, dst, dtok)) return 1;
    if (run<256, 64>("all equal", ones, 0.9f, dv, dsum, du, dst, dtok)) return 1;
    return 0;
}
