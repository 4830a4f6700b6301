// launch_lab: cost of a dependent kernel->kernel edge inside a hipGraph on this chip (what 162 launches/token pay).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void lds_user(int* p) { extern __shared__ int sm[]; sm[threadIdx.x] = threadIdx.x; __syncthreads(); if (p && sm[(threadIdx.x + 1) % blockDim.x] == -5) *p = 1; }
__global__ void rw(const float* in, float* out) {   // minimal dependent work: read 16 KB written by the predecessor, write 16 KB
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4096) out[i] = in[i] + 1.0f;
}
// round 6: does the kernarg fetch sit on the edge?  The same dependent step over 256 x 512 threads (every workgroup reads the
// predecessor's 16 KB, one writes) with its arguments in a by-value struct (the library's GemvParams shape: never preloaded) or as
// leading scalars (preloaded into SGPRs at wave launch when built with -mllvm -amdgpu-kernarg-preload-count=8)
struct RwArgs { const float* in; float* out; int pad[28]; };
__global__ void rw_struct(RwArgs a) {
    const int i = threadIdx.x;
    float v = 0.f;
    for (int k = i; k < 4096; k += 512) v += a.in[k];
    if (blockIdx.x == 0) for (int k = i; k < 4096; k += 512) a.out[k] = v * 0.0f + 1.0f;
}
__global__ void rw_scalars(const float* in, float* out, int n) {
    const int i = threadIdx.x;
    float v = 0.f;
    for (int k = i; k < n; k += 512) v += in[k];
    if (blockIdx.x == 0) for (int k = i; k < n; k += 512) out[k] = v * 0.0f + 1.0f;
}
int main() {
    hipStream_t A; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    float *x, *y; CK(hipMalloc(&x, 16384)); CK(hipMalloc(&y, 16384)); CK(hipMemset(x, 0, 16384));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int N = 400;
    for (int var = 0; var < 7; var++) {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; i++) {
            if (var == 0) empty<<<1, 64, 0, A>>>(nullptr);
            else if (var == 1) empty<<<256, 512, 0, A>>>(nullptr);
            else if (var == 2) lds_user<<<256, 512, 32768, A>>>(nullptr);
            else if (var == 3) lds_user<<<2048, 512, 32768, A>>>(nullptr);
            else if (var == 4) rw<<<16, 256, 0, A>>>((i & 1) ? y : x, (i & 1) ? x : y);
            else if (var == 5) { RwArgs a{(i & 1) ? y : x, (i & 1) ? x : y, {0}}; rw_struct<<<256, 512, 0, A>>>(a); }
            else rw_scalars<<<256, 512, 0, A>>>((i & 1) ? y : x, (i & 1) ? x : y, 4096);
        }
        CK(hipStreamEndCapture(A, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, A)); CK(hipStreamSynchronize(A));
        CK(hipEventRecord(t0, A));
        for (int it = 0; it < 5; it++) CK(hipGraphLaunch(exec, A));
        CK(hipEventRecord(t1, A)); CK(hipStreamSynchronize(A));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        const char* names[] = {"empty 1x64", "empty 256x512", "lds 32K 256x512", "lds 32K 2048x512", "rw 16KB dependent 16x256", "read 16KB/wg 256x512, struct args", "read 16KB/wg 256x512, scalar args"};
        printf("%-40s %6.2f us per node\n", names[var], ms * 1e3 / (5 * N));
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
