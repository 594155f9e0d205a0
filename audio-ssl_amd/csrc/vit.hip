// Transformer-block plumbing of the AST / MAST encoder on gfx950 (the GEMMs are gemm.hip, attention is attention.hip):
//   layernorm_fwd / _bwd : nn.LayerNorm(768, eps=1e-6) of the pre-norm ViT block (`ast_work.py:70-81` builds timm's
//                          VisionTransformer; its blocks are LN -> MHA -> +res, LN -> MLP(GELU) -> +res)
//   gelu_fwd / _bwd      : exact (erf) GELU of the MLP
//   patch_unfold         : 16x16 patches with stride (fstride, tstride) of `ASTModel`'s patch embedding Conv2d
//                          (`ast_work.py:101`), written as GEMM rows
//   adamw                : torch.optim.AdamW of `Moco_v2.configure_optimizers` (`moco_model.py:373-379`), one launch over
//                          the flat parameter buffer, step count read from device memory (graph replay advances it)
// The residual stream is fp32; LayerNorm writes the bf16 MFMA operand; every gradient entering a LayerNorm backward is fp32.
#include "common.h"

namespace {

constexpr int LN_MAX = 16;       // columns per lane: C <= 1024

// one wave per row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ y,
                                                            float* __restrict__ y32, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int M, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (long)row * C;
    float v[LN_MAX];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i)
        if (lane + 64 * i < C) { v[i] = xr[lane + 64 * i]; s += v[i]; }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i)
        if (lane + 64 * i < C) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i)
        if (lane + 64 * i < C) {
            const int c = lane + 64 * i;
            const float o = (v[i] - mu) * rs * gamma[c] + beta[c];
            y[(long)row * C + c] = (bf16)o;
            if (y32) y32[(long)row * C + c] = o;
        }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// The same with 16-byte loads, 8-byte bf16 stores (C % 4 == 0) and two rows per wave in flight (the dword form stored 2 bytes per lane:
// 47.6 us per 27,648 x 768 launch, 2.7 TB/s)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, bf16* __restrict__ y,
                                                                float* __restrict__ y32, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int M, int C, float eps) {
    const int lane = threadIdx.x & 63, nch = C >> 2;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= M) return;
    f32x4 v[2][NV];
    const bool live1 = row0 + 1 < M;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = lane + 64 * i;
            const bool on = ch < nch && (r == 0 || live1);
            v[r][i] = on ? *reinterpret_cast<const f32x4*>(x + (long)(row0 + r) * C + ch * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 gm[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = lane + 64 * i;
        gm[i] = ch < nch ? *reinterpret_cast<const f32x4*>(gamma + ch * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        bt[i] = ch < nch ? *reinterpret_cast<const f32x4*>(beta + ch * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r == 1 && !live1) break;
        const int row = row0 + r;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += v[r][i][e];                       // chunks past the row are zeros
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[r][i][e] - mu; q += d * d; }
            }
        const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[r][i][e] - mu) * rs * gm[i][e] + bt[i][e];
                *reinterpret_cast<bf16x4*>(y + (long)row * C + ch * 4) = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                if (y32) *reinterpret_cast<f32x4*>(y32 + (long)row * C + ch * 4) = o;
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// dres[row] += dx,  dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma
// dgamma += sum_rows dy * xhat, dbeta += sum_rows dy   (per-workgroup partials in LDS, then one atomic per column)
constexpr int LN_ROWS = 64;      // rows per workgroup (16 per wave)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dres,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C) {
    __shared__ float sg[1024], sb[1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < C; c += 256) { sg[c] = 0.f; sb[c] = 0.f; }
    __syncthreads();
    float pg[LN_MAX], pb[LN_MAX], gm[LN_MAX];
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i) { pg[i] = 0.f; pb[i] = 0.f; gm[i] = lane + 64 * i < C ? gamma[lane + 64 * i] : 0.f; }
    for (int k = 0; k < LN_ROWS / 4; ++k) {
        const int row = blockIdx.x * LN_ROWS + wave * (LN_ROWS / 4) + k;
        if (row >= M) break;
        const float mu = mean[row], rs = rstd[row];
        float g[LN_MAX], xh[LN_MAX];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX; ++i)
            if (lane + 64 * i < C) {
                const long o = (long)row * C + lane + 64 * i;
                const float d = dy[o];
                xh[i] = (x[o] - mu) * rs;
                g[i] = d * gm[i];
                s1 += g[i];
                s2 += g[i] * xh[i];
                pg[i] += d * xh[i];
                pb[i] += d;
            }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < LN_MAX; ++i)
            if (lane + 64 * i < C) {
                const long o = (long)row * C + lane + 64 * i;
                dres[o] += rs * (g[i] - s1 - xh[i] * s2);
            }
    }
#pragma unroll
    for (int i = 0; i < LN_MAX; ++i)
        if (lane + 64 * i < C) { atomicAdd(&sg[lane + 64 * i], pg[i]); atomicAdd(&sb[lane + 64 * i], pb[i]); }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(dgamma + c, sg[c]); atomicAdd(dbeta + c, sb[c]); }
}

// The same with 16-byte accesses (C % 4 == 0: lane l owns the float4 chunks l, l + 64, ...) and TWO rows of a wave in flight: the
// one-dword-per-lane form above moved 1.6 TB/s on AST-base's 27,648 x 768 launches (212 us each, 12 % of the SS-MAST step) - every
// row was a load -> two wave reductions -> read-modify-write chain with nothing else outstanding.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, float* __restrict__ dres,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C) {
    __shared__ float sg[1024], sb[1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nch = C >> 2;                                          // float4 chunks per row
    for (int c = threadIdx.x; c < C; c += 256) { sg[c] = 0.f; sb[c] = 0.f; }
    __syncthreads();
    f32x4 gm[NV], pg[NV], pb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = lane + 64 * i;
        gm[i] = ch < nch ? *reinterpret_cast<const f32x4*>(gamma + ch * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        pg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int row0 = blockIdx.x * LN_ROWS + wave * (LN_ROWS / 4);
    for (int k = 0; k < LN_ROWS / 4; k += 2) {
        f32x4 d[2][NV], xv[2][NV], rv[2][NV];
        float mu[2], rs[2];
        bool live[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {                                // both rows' loads are issued before anything is reduced
            const int row = row0 + k + r;
            live[r] = row < M;
            mu[r] = live[r] ? mean[row] : 0.f;
            rs[r] = live[r] ? rstd[row] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = lane + 64 * i;
                const bool on = live[r] && ch < nch;
                const long o = (long)row * C + ch * 4;
                d[r][i] = on ? *reinterpret_cast<const f32x4*>(dy + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                xv[r][i] = on ? *reinterpret_cast<const f32x4*>(x + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                rv[r][i] = on ? *reinterpret_cast<const f32x4*>(dres + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (!live[r]) continue;                                  // wave-uniform
            float s1 = 0.f, s2 = 0.f;
            f32x4 g[NV], xh[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool on = lane + 64 * i < nch;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xe = on ? (xv[r][i][e] - mu[r]) * rs[r] : 0.f;
                    const float ge = d[r][i][e] * gm[i][e];
                    xh[i][e] = xe; g[i][e] = ge;
                    s1 += ge; s2 += ge * xe;
                    pg[i][e] += d[r][i][e] * xe;
                    pb[i][e] += d[r][i][e];
                }
            }
            s1 = wave_sum(s1) / (float)C;
            s2 = wave_sum(s2) / (float)C;
            const int row = row0 + k + r;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nch) {
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = rv[r][i][e] + rs[r] * (g[i][e] - s1 - xh[i][e] * s2);
                    *reinterpret_cast<f32x4*>(dres + (long)row * C + ch * 4) = o4;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { atomicAdd(&sg[ch * 4 + e], pg[i][e]); atomicAdd(&sb[ch * 4 + e], pb[i][e]); }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) { atomicAdd(dgamma + c, sg[c]); atomicAdd(dbeta + c, sb[c]); }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16* __restrict__ a, bf16* __restrict__ h, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const Vec8<bf16> v = Vec8<bf16>::load(a + i * 8);
    Vec8<bf16> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, gelu_f(v.get(k)));
    o.store(h + i * 8);
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ dh, bf16* __restrict__ da,
                                                       long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const Vec8<bf16> v = Vec8<bf16>::load(a + i * 8), g = Vec8<bf16>::load(dh + i * 8);
    Vec8<bf16> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, g.get(k) * gelu_d(v.get(k)));
    o.store(da + i * 8);
}

// x [B][F][T] fp32 -> rows [B*nf*nt][256] bf16, row = (b, pf, pt) (the order of conv_out.flatten(2)), col = kh*16 + kw
__global__ __launch_bounds__(256) void patch_unfold_kernel(const float* __restrict__ x, bf16* __restrict__ out, int B, int F, int T,
                                                           int nf, int nt, int fs, int ts) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // over rows * 32 (8 columns each)
    const long rows = (long)B * nf * nt;
    if (idx >= rows * 32) return;
    const int c8 = (int)(idx & 31);
    const long row = idx >> 5;
    const int pt = (int)(row % nt), pf = (int)((row / nt) % nf), b = (int)(row / ((long)nt * nf));
    const int kh = c8 >> 1, kw = (c8 & 1) * 8;
    const float* src = x + ((long)b * F + pf * fs + kh) * T + pt * ts + kw;
    Vec8<bf16> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.set(k, src[k]);
    o.store(out + row * 256 + c8 * 8);
}

// out[r][c] = scale * src[(r / group) % period][c]
//   group 1, period N : the learned position embedding tiled over the batch (start value of the residual stream)
//   group N, period B : backward of the mean over tokens (every token row gets dpooled / N)
__global__ __launch_bounds__(256) void tile_rows_kernel(const float* __restrict__ src, float* __restrict__ out, long rows, int period,
                                                        int group, float scale, int C4) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C4) return;
    const long r = idx / C4;
    const int c = (int)(idx % C4);
    f32x4 v = reinterpret_cast<const f32x4*>(src)[((r / group) % period) * C4 + c];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] *= scale;
    reinterpret_cast<f32x4*>(out)[idx] = v;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                                                    float gscale, const long long* __restrict__ step) {
    const float t = (float)step[0];
    const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + i), mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gg = gv[k] * gscale;
                pv[k] *= 1.f - lr * wd;
                mv[k] = b1 * mv[k] + (1.f - b1) * gg;
                vv[k] = b2 * vv[k] + (1.f - b2) * gg * gg;
                pv[k] -= step_size * mv[k] / (sqrtf(vv[k]) / bc2s + eps);
            }
            *reinterpret_cast<f32x4*>(p + i) = pv;
            *reinterpret_cast<f32x4*>(m + i) = mv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
        } else {
            for (long j = i; j < n; ++j) {
                const float gg = g[j] * gscale;
                float pj = p[j] * (1.f - lr * wd);
                m[j] = b1 * m[j] + (1.f - b1) * gg;
                v[j] = b2 * v[j] + (1.f - b2) * gg * gg;
                p[j] = pj - step_size * m[j] / (sqrtf(v[j]) / bc2s + eps);
            }
        }
    }
}

}  // namespace

#define S_(stream) static_cast<hipStream_t>(stream)

extern "C" int audiossl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* y32, float* mean, float* rstd,
                                      int M, int C, float eps, void* stream) {
    ASSL_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0 && C > 0 && C <= 64 * LN_MAX);
    const bool vec = C % 4 == 0 && ASSL_ALIGNED16(x) && ASSL_ALIGNED16(gamma) && ASSL_ALIGNED16(beta) &&
                     (reinterpret_cast<size_t>(y) & 7) == 0 && (!y32 || ASSL_ALIGNED16(y32));
    if (vec) {
        const int nv = ceil_div(C / 4, 64);
#define LNF(NV_) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<NV_>, dim3(ceil_div(M, 8)), dim3(256), 0, S_(stream), x, gamma, beta, \
                                    static_cast<bf16*>(y), y32, mean, rstd, M, C, eps)
        if (nv == 1) LNF(1); else if (nv == 2) LNF(2); else if (nv == 3) LNF(3); else LNF(4);
#undef LNF
        ASSL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, S_(stream), x, gamma, beta, static_cast<bf16*>(y),
                       y32, mean, rstd, M, C, eps);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                      float* dres, float* dgamma, float* dbeta, int M, int C, void* stream) {
    ASSL_REQUIRE(dy && x && mean && rstd && gamma && dres && dgamma && dbeta && M > 0 && C > 0 && C <= 64 * LN_MAX);
    const bool vec = C % 4 == 0 && ASSL_ALIGNED16(dy) && ASSL_ALIGNED16(x) && ASSL_ALIGNED16(dres) && ASSL_ALIGNED16(gamma);
    if (vec) {
        const int nv = ceil_div(C / 4, 64);                      // float4 chunks per lane: 1 ... 4
#define LNB(NV_) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<NV_>, dim3(ceil_div(M, LN_ROWS)), dim3(256), 0, S_(stream), dy, x, mean, rstd, \
                                    gamma, dres, dgamma, dbeta, M, C)
        if (nv == 1) LNB(1); else if (nv == 2) LNB(2); else if (nv == 3) LNB(3); else LNB(4);
#undef LNB
        ASSL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(ceil_div(M, LN_ROWS)), dim3(256), 0, S_(stream), dy, x, mean, rstd, gamma, dres,
                       dgamma, dbeta, M, C);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_gelu_fwd(const void* a, void* h, long n, void* stream) {
    ASSL_REQUIRE(a && h && n > 0 && (n % 8) == 0);
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ceil_div(n / 8, 256)), dim3(256), 0, S_(stream), static_cast<const bf16*>(a),
                       static_cast<bf16*>(h), n / 8);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_gelu_bwd(const void* a, const void* dh, void* da, long n, void* stream) {
    ASSL_REQUIRE(a && dh && da && n > 0 && (n % 8) == 0);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ceil_div(n / 8, 256)), dim3(256), 0, S_(stream), static_cast<const bf16*>(a),
                       static_cast<const bf16*>(dh), static_cast<bf16*>(da), n / 8);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_patch_unfold(const float* x, void* out, int B, int F, int T, int fstride, int tstride, void* stream) {
    ASSL_REQUIRE(x && out && B > 0 && F >= 16 && T >= 16 && fstride > 0 && tstride > 0);
    const int nf = (F - 16) / fstride + 1, nt = (T - 16) / tstride + 1;
    const long total = (long)B * nf * nt * 32;
    hipLaunchKernelGGL(patch_unfold_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, S_(stream), x, static_cast<bf16*>(out), B, F, T,
                       nf, nt, fstride, tstride);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_tile_rows(const float* src, float* out, long rows, int period, int group, float scale, int C, void* stream) {
    ASSL_REQUIRE(src && out && rows > 0 && period > 0 && group > 0 && C > 0 && (C % 4) == 0);
    if (!ASSL_ALIGNED16(src) || !ASSL_ALIGNED16(out)) return ASSL_EALIGN;
    hipLaunchKernelGGL(tile_rows_kernel, dim3(ceil_div(rows * (C / 4), 256)), dim3(256), 0, S_(stream), src, out, rows, period, group, scale, C / 4);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, const long long* step, void* stream) {
    ASSL_REQUIRE(p && g && m && v && step && n > 0);
    if (!ASSL_ALIGNED16(p) || !ASSL_ALIGNED16(g) || !ASSL_ALIGNED16(m) || !ASSL_ALIGNED16(v)) return ASSL_EALIGN;
    const int grid = (int)min((long)2048, (n + 1023) / 1024);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, S_(stream), p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       grad_scale, step);
    ASSL_LAUNCH_CHECK();
}
