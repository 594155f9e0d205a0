// Shared device/host helpers for the audiossl gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- C-ABI error codes (include/audiossl_hip.h) ---------------------------------
#define ASSL_OK 0
#define ASSL_EINVAL (-1)      // bad shape / null pointer / unsupported configuration
#define ASSL_ELAUNCH (-2)     // hipGetLastError() after a launch was not hipSuccess
#define ASSL_EALIGN (-3)      // pointer or leading dimension not 16-byte aligned

#define ASSL_REQUIRE(cond) do { if (!(cond)) return ASSL_EINVAL; } while (0)
#define ASSL_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)
// Scratch zeroing.  Entry points that accumulate into caller-provided scratch zero it themselves - unless the caller has
// declared (audiossl_set_prezeroed) that every scratch pointer it passes is already zero: the fused training step takes all
// of them from one arena cleared by a single memset, which removes ~30 tiny memset nodes from the step's graph.
//
// The clear is a KERNEL, never hipMemsetAsync: a memset captured into a hipGraph becomes a memset node, and on ROCm 7.2 the
// replayed graph did not reliably order such nodes against the kernel nodes around them.  Round 1's two-rank graph step
// (both ranks on one GPU) produced a wrong update in 2 of 8 runs with ~30 memset nodes clearing recycled statistics scratch
// inside the phase graphs; the same build with every memset replaced by this kernel: 0 of 20 (tools/bisect_ddp.sh,
// DESIGN.md section 5).  All sizes cleared here are multiples of 4 bytes.
extern int g_assl_prezeroed;
static __global__ void assl_zero_kernel(unsigned int* __restrict__ p, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static inline hipError_t assl_zero_async(void* p, size_t bytes, hipStream_t s) {
    long n = (long)((bytes + 3) / 4);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(assl_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (unsigned int*)p, n);
    return hipGetLastError();
}
#define ASSL_ZERO(ptr, bytes, s) \
    do { if (!g_assl_prezeroed && assl_zero_async((ptr), (bytes), (s)) != hipSuccess) return ASSL_ELAUNCH; } while (0)
#define ASSL_ZERO_ALWAYS(ptr, bytes, s) \
    do { if (assl_zero_async((ptr), (bytes), (s)) != hipSuccess) return ASSL_ELAUNCH; } while (0)

// Name of the kernel the most recent GEMM-family entry point launched (the dispatch rules pick among several instantiations);
// read back through audiossl_last_kernel by bench.py so that its per-kernel table joins to rocprofv3's kernel names.
extern const char* g_assl_last_kernel;
#define ASSL_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return ASSL_ELAUNCH; return ASSL_OK; } while (0)

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- scalar conversions -----------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }   // RNE, NaN-preserving

// 8-element vector of T (16 B for bf16, 32 B for float)
template <typename T> struct Vec8;
template <> struct Vec8<bf16> {
    bf16x8 v;
    __device__ __forceinline__ static Vec8 load(const bf16* p) { Vec8 r; r.v = *reinterpret_cast<const bf16x8*>(p); return r; }
    __device__ __forceinline__ void store(bf16* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
    __device__ __forceinline__ static Vec8 zero() { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = (bf16)0.f; return r; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
    __device__ __forceinline__ bf16 raw(int i) const { return v[i]; }
    __device__ __forceinline__ void setraw(int i, bf16 x) { v[i] = x; }
};
template <> struct Vec8<float> {
    f32x4 lo, hi;
    __device__ __forceinline__ static Vec8 load(const float* p) {
        Vec8 r; r.lo = *reinterpret_cast<const f32x4*>(p); r.hi = *reinterpret_cast<const f32x4*>(p + 4); return r; }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<f32x4*>(p) = lo; *reinterpret_cast<f32x4*>(p + 4) = hi; }
    __device__ __forceinline__ static Vec8 zero() { Vec8 r; r.lo = f32x4{0, 0, 0, 0}; r.hi = f32x4{0, 0, 0, 0}; return r; }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? lo[i] : hi[i - 4]; }
    __device__ __forceinline__ void set(int i, float x) { if (i < 4) lo[i] = x; else hi[i - 4] = x; }
    __device__ __forceinline__ float raw(int i) const { return get(i); }
    __device__ __forceinline__ void setraw(int i, float x) { set(i, x); }
};

// ---- counter-based dropout keep bit: keep iff hash(seed, index) >= p * 2^32 (splitmix64 finaliser).  ONE definition for the mask
// kernel (heads.hip: dropout_mask_kernel) and for the GEMM epilogue that draws the mask of its own elements (gemm.hip): the two
// produce the same mask from the same (seed, counter).
__device__ __forceinline__ unsigned long long dropout_seed(unsigned long long seed, const long long* counter) {
    return counter ? (seed + (unsigned long long)counter[0]) & 0xFFFFFFFFFFFFull : seed;
}
__device__ __forceinline__ unsigned int dropout_threshold(float p) { return (unsigned int)fminf(p * 4294967296.f, 4294967295.f); }
__device__ __forceinline__ unsigned int dropout_keep(unsigned long long seed, long i, unsigned int thr) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return ((unsigned int)(z >> 32) >= thr) ? 1u : 0u;
}

// ---- torch.optim.SGD on one element: g += wd p; buf = first ? g : momentum buf + g; p -= lr buf -------------------------
// ONE definition with explicit fused multiply-adds for the flat SGD pass (heads.hip) and the weight-gradient GEMM epilogue that
// applies the update in place (gemm.hip): the two must agree bit for bit, whatever the compiler would contract on its own.
__device__ __forceinline__ void sgd_step(float& p, float g, float& buf, float lr, float mom, float wd, float gscale, bool first) {
    const float gg = __builtin_fmaf(wd, p, g * gscale);
    buf = first ? gg : __builtin_fmaf(mom, buf, gg);
    p = __builtin_fmaf(-lr, buf, p);
}

// ---- wave / block reductions ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over a block of up to 1024 threads; result valid in every thread.  `sh` needs 16 slots.
template <typename A>
__device__ __forceinline__ A block_sum(A v, A* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    A t = 0;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}
