// Diagnostic: per-launch time of tiny kernels as a function of dynamic LDS size and MFMA / AGPR use (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void k_plain(float* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.f; }
__global__ __launch_bounds__(256) void k_lds(float* out) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sm[17];
}
__global__ __launch_bounds__(256) void k_mfma(float* out, int n) {
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.f; }
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = acc[3];
}
template <typename F> float timeit(F f, int reps = 200) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    float* out; hipMalloc(&out, 1 << 20);
    printf("plain            1 block   : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_plain, dim3(1), dim3(256), 0, 0, out); }));
    printf("plain          256 blocks  : %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, 0, out); }));
    for (int kb : {1, 32, 64, 72, 96, 128, 159}) {
        hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        printf("lds %3d KB       1 block   : %.2f us", kb, timeit([&] { hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), kb * 1024, 0, out); }));
        printf("   256 blocks: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), kb * 1024, 0, out); }));
    }
    for (int n : {1, 16, 64, 512}) {
        printf("mfma x%-4d       1 block   : %.2f us", n, timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(1), dim3(256), 0, 0, out, n); }));
        printf("   256 blocks: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, 0, out, n); }));
    }
    return 0;
}
