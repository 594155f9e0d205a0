// Loss heads and optimiser pieces of the remaining SURVEY 8(a) rows:
//   a18  NT-Xent / ClusterLoss           `extras/slicer/contrastive_loss.py:6-92`
//   a19  DeepCluster-v2 spherical k-means `extras/decar-v2/utils.py:276-346`, prototype CE `extras/decar-v2/main.py:228-233`
//   a21  LARS                            `extras/delores-s/multi_proc.py:4-43`
// The similarity / dot-product matrices come from the MFMA GEMM (gemm.hip); these kernels are the row-wise
// soft-max / arg-max / scatter passes around it (HBM-bound, one read of the [rows][cols] fp32 matrix each).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------- NT-Xent
// sim [N][N] fp32 (already divided by the temperature), N = 2B, positives at column (r + B) mod N, self excluded.
// loss += sum_r (lse_r - sim[r][pos]) / N ;  lse over all columns except r.
__global__ __launch_bounds__(256) void ntxent_fwd_kernel(const float* __restrict__ sim, int N, int B, float* __restrict__ lse,
                                                         float* __restrict__ loss_out) {
    __shared__ float sh[16];
    const int r = blockIdx.x;
    const float* row = sim + (long)r * N;
    float m = -3.0e38f;
    for (int c = threadIdx.x; c < N; c += 256) if (c != r) m = fmaxf(m, row[c]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float s = 0.f;
    for (int c = threadIdx.x; c < N; c += 256) if (c != r) s += expf(row[c] - m);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        const float l = m + logf(s);
        lse[r] = l;
        atomicAdd(loss_out, (l - row[(r + B) % N]) / (float)N);
    }
}

// dsim[r][c] = (softmax_r[c] - [c == pos(r)]) * gscale for c != r, 0 on the diagonal   (gscale = 1 / (N * tau))
template <typename T_>
__global__ __launch_bounds__(256) void ntxent_bwd_kernel(const float* __restrict__ sim, const float* __restrict__ lse, int N,
                                                         int B, float gscale, T_* __restrict__ dsim) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float v = 0.f;
    if (c != r) v = (expf(sim[(long)r * N + c] - lse[r]) - (c == (r + B) % N ? 1.f : 0.f)) * gscale;
    dsim[(long)r * N + c] = from_f32<T_>(v);
}

// ---------------------------------------------------------------------------------------------------- k-means / CE rows
// assign[r] = argmax_c dot[r][c] (first maximum, like torch.max); optional best value
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ dot, long N, int K, long long* __restrict__ assign) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const long r = blockIdx.x;
    const float* row = dot + r * K;
    float bv = -3.0e38f;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < K; c += 256) { const float v = row[c]; if (v > bv) { bv = v; bi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        assign[r] = bi;
    }
}

// M step: sums[a][:] += x[r][:], counts[a] += 1   (fp32 atomics; one wave per row, 128-byte runs)
__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float* __restrict__ x, const long long* __restrict__ assign,
                                                                long N, int D, float* __restrict__ sums, int* __restrict__ counts) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int lane = threadIdx.x & 63;
    const long long a = assign[r];
    if (a < 0) return;
    for (int d = lane; d < D; d += 64) atomicAdd(&sums[a * D + d], x[r * D + d]);
    if (lane == 0) atomicAdd(&counts[a], 1);
}

// centroids[k] = normalize(counts[k] > 0 ? sums[k] / counts[k] : centroids[k])   (empty clusters keep their old value,
// `utils.py:313-318`; F.normalize eps 1e-12).  One wave per centroid.
__global__ __launch_bounds__(64) void kmeans_update_kernel(const float* __restrict__ sums, const int* __restrict__ counts, int D,
                                                           float* __restrict__ centroids) {
    const long k = blockIdx.x;
    const int cnt = counts[k];
    float ss = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) {
        const float v = cnt > 0 ? sums[k * D + d] / (float)cnt : centroids[k * D + d];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int d = threadIdx.x; d < D; d += 64) {
        const float v = cnt > 0 ? sums[k * D + d] / (float)cnt : centroids[k * D + d];
        centroids[k * D + d] = v * inv;
    }
}

// Lloyd (Euclidean) M step of the offline labeler (faiss.Clustering in extras/decar-v2/clustering.py:46-85): plain mean, empty
// clusters keep their centroid.  out[k] = scale * |x_k|^2 is the bias of the E step's argmin |x - c|^2 = argmax (x.c - |c|^2 / 2).
__global__ __launch_bounds__(64) void kmeans_update_mean_kernel(const float* __restrict__ sums, const int* __restrict__ counts, int D,
                                                                float* __restrict__ centroids) {
    const long k = blockIdx.x;
    const int cnt = counts[k];
    if (cnt <= 0) return;
    for (int d = threadIdx.x; d < D; d += 64) centroids[k * D + d] = sums[k * D + d] / (float)cnt;
}
__global__ __launch_bounds__(64) void row_sqnorm_kernel(const float* __restrict__ x, int D, float scale, float* __restrict__ out) {
    const long k = blockIdx.x;
    float ss = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) { const float v = x[k * D + d]; ss += v * v; }
    ss = wave_sum(ss);
    if (threadIdx.x == 0) out[k] = scale * ss;
}

// nn.CrossEntropyLoss(ignore_index): loss += sum_{valid r} (lse_r - logit[r][t_r]) / n_valid ; dlogits = (softmax - onehot)/n_valid
// n_valid is counted by a first tiny launch into cnt[0].
__global__ void count_valid_kernel(const long long* __restrict__ target, int B, int ignore_index, int* cnt) {
    int c = 0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) c += target[i] != ignore_index;
    c = (int)wave_sum((float)c);
    if ((threadIdx.x & 63) == 0) atomicAdd(cnt, c);
}

template <typename T_>
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const long long* __restrict__ target, int K,
                                                      int ignore_index, const int* __restrict__ cnt, float* __restrict__ loss_out,
                                                      T_* __restrict__ dlogits) {
    __shared__ float sh[16];
    const long r = blockIdx.x;
    const float* row = logits + r * K;
    const long long t = target[r];
    const bool valid = t != ignore_index;
    float m = -3.0e38f;
    for (int c = threadIdx.x; c < K; c += 256) m = fmaxf(m, row[c]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    float s = 0.f;
    for (int c = threadIdx.x; c < K; c += 256) s += expf(row[c] - m);
    s = block_sum(s, sh);
    const float l = m + logf(s);
    const float inv = cnt[0] > 0 ? 1.f / (float)cnt[0] : 0.f;
    if (threadIdx.x == 0 && valid) atomicAdd(loss_out, (l - row[t]) * inv);
    if (dlogits)
        for (int c = threadIdx.x; c < K; c += 256)
            dlogits[r * K + c] = from_f32<T_>(valid ? (expf(row[c] - l) - (c == t ? 1.f : 0.f)) * inv : 0.f);
}

// ---------------------------------------------------------------------------------------------------- row softmax
// nn.Softmax(dim=1) of SLICER's cluster projector (`src/upstream/slicer/upstream_encoder.py:19`): one wave per row.
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int C) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M) return;
    const float* row = x + (long)r * C;
    float m = -3.0e38f;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(row[c] - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < C; c += 64) y[(long)r * C + c] = expf(row[c] - m) * inv;
}

// dx = y * (gy - sum_c gy*y)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                               float* __restrict__ gx, int M, int C) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M) return;
    const float* yr = y + (long)r * C;
    const float* gr = gy + (long)r * C;
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d += yr[c] * gr[c];
    d = wave_sum(d);
    for (int c = lane; c < C; c += 64) gx[(long)r * C + c] = yr[c] * (gr[c] - d);
}

// ---------------------------------------------------------------------------------------------------- LARS
// seg[i] = {offset, numel, flags}: flags bit0 = apply weight decay, bit1 = apply the trust ratio
// norms[i] = {sum p^2, sum dp^2} with dp = g + wd*p (if bit0)
struct LarsSeg { long long offset; long long numel; int flags; int pad; };

__global__ __launch_bounds__(256) void lars_norms_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                         const LarsSeg* __restrict__ seg, float wd, float gscale,
                                                         double* __restrict__ norms) {
    __shared__ double sh[16];
    const LarsSeg s = seg[blockIdx.y];
    const float w = (s.flags & 1) ? wd : 0.f;
    double a = 0.0, b = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.numel; i += (long long)gridDim.x * 256) {
        const float pv = p[s.offset + i], dv = g[s.offset + i] * gscale + w * pv;
        a += (double)pv * pv;
        b += (double)dv * dv;
    }
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0 && (a != 0.0 || b != 0.0)) { atomicAdd(&norms[2 * blockIdx.y], a); atomicAdd(&norms[2 * blockIdx.y + 1], b); }
}

// dp = g + wd*p; dp *= q with q = (|p| > 0 && |dp| > 0) ? eta*|p|/|dp| : 1 ; mu = m*mu + dp ; p -= lr*mu
__global__ __launch_bounds__(256) void lars_update_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu,
                                                          const LarsSeg* __restrict__ seg, const double* __restrict__ norms,
                                                          const float* __restrict__ lr, float wd, float momentum, float eta,
                                                          float gscale) {
    const LarsSeg s = seg[blockIdx.y];
    const float w = (s.flags & 1) ? wd : 0.f;
    float q = 1.f;
    if (s.flags & 2) {
        const float pn = (float)sqrt(norms[2 * blockIdx.y]), un = (float)sqrt(norms[2 * blockIdx.y + 1]);
        if (pn > 0.f && un > 0.f) q = eta * pn / un;
    }
    const float l = lr[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.numel; i += (long long)gridDim.x * 256) {
        const long long o = s.offset + i;
        const float pv = p[o];
        const float dp = (g[o] * gscale + w * pv) * q;
        const float m = momentum * mu[o] + dp;
        mu[o] = m;
        p[o] = pv - l * m;
    }
}

// apex LARC (extras/decar-v2/main.py:111: LARC(SGD(momentum .9, wd), trust_coefficient, clip=False)) over the same segments:
// per tensor, when |p| and |g| are both non-zero:  alr = tc |p| / (|g| + wd |p| + eps)  [clip: min(alr / lr, 1)],
// g' = (g + wd p) alr;  otherwise g' = g (no decay: apex zeroes the group's weight_decay and re-applies it only in that
// branch); then torch.optim.SGD with weight_decay 0: buf = m buf + g', p -= lr buf.
__global__ __launch_bounds__(256) void larc_norms_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                         const LarsSeg* __restrict__ seg, float gscale, double* __restrict__ norms) {
    __shared__ double sh[16];
    const LarsSeg s = seg[blockIdx.y];
    double a = 0.0, b = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.numel; i += (long long)gridDim.x * 256) {
        const float pv = p[s.offset + i], gv = g[s.offset + i] * gscale;
        a += (double)pv * pv;
        b += (double)gv * gv;
    }
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0 && (a != 0.0 || b != 0.0)) { atomicAdd(&norms[2 * blockIdx.y], a); atomicAdd(&norms[2 * blockIdx.y + 1], b); }
}
__global__ __launch_bounds__(256) void larc_update_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu,
                                                          const LarsSeg* __restrict__ seg, const double* __restrict__ norms, float lr,
                                                          float wd, float momentum, float tc, float eps, int clip, float gscale) {
    const LarsSeg s = seg[blockIdx.y];
    if (s.flags & 4) return;                                 // tensor without a gradient this step (frozen prototypes): untouched
    const float pn = (float)sqrt(norms[2 * blockIdx.y]), gn = (float)sqrt(norms[2 * blockIdx.y + 1]);
    float alr = 1.f, w = 0.f;
    if (pn != 0.f && gn != 0.f) {
        alr = tc * pn / (gn + pn * wd + eps);
        if (clip) alr = fminf(alr / lr, 1.f);
        w = wd;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.numel; i += (long long)gridDim.x * 256) {
        const long long o = s.offset + i;
        const float pv = p[o];
        const float dp = (g[o] * gscale + w * pv) * alr;
        const float m = momentum * mu[o] + dp;
        mu[o] = m;
        p[o] = pv - lr * m;
    }
}

}  // namespace

#define S_(stream) static_cast<hipStream_t>(stream)

extern "C" int audiossl_ntxent_fwd(const float* sim, int N, int B, float* lse, float* loss_out, void* stream) {
    ASSL_REQUIRE(sim && lse && loss_out && N > 1 && B > 0 && B < N);
    hipLaunchKernelGGL(ntxent_fwd_kernel, dim3(N), dim3(256), 0, S_(stream), sim, N, B, lse, loss_out);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_ntxent_bwd(int dtype, const float* sim, const float* lse, int N, int B, float gscale, void* dsim,
                                   void* stream) {
    ASSL_REQUIRE(sim && lse && dsim && N > 1 && B > 0 && (dtype == 0 || dtype == 1));
    dim3 grid(ceil_div(N, 256), N);
    if (dtype == 0) hipLaunchKernelGGL(ntxent_bwd_kernel<float>, grid, dim3(256), 0, S_(stream), sim, lse, N, B, gscale, (float*)dsim);
    else            hipLaunchKernelGGL(ntxent_bwd_kernel<bf16>, grid, dim3(256), 0, S_(stream), sim, lse, N, B, gscale, (bf16*)dsim);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_row_argmax(const float* dot, long N, int K, long long* assign, void* stream) {
    ASSL_REQUIRE(dot && assign && N > 0 && K > 0 && N < 2147483647L);
    hipLaunchKernelGGL(row_argmax_kernel, dim3((unsigned)N), dim3(256), 0, S_(stream), dot, N, K, assign);
    ASSL_LAUNCH_CHECK();
}

// sums [K][D] and counts [K] are zeroed here
extern "C" int audiossl_kmeans_accumulate(const float* x, const long long* assign, long N, int K, int D, float* sums, int* counts,
                                          void* stream) {
    ASSL_REQUIRE(x && assign && sums && counts && N > 0 && K > 0 && D > 0);
    hipStream_t s = S_(stream);
    ASSL_ZERO_ALWAYS(sums, sizeof(float) * K * D, s);
    ASSL_ZERO_ALWAYS(counts, sizeof(int) * K, s);
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, s, x, assign, N, D, sums, counts);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_kmeans_update(const float* sums, const int* counts, int K, int D, float* centroids, void* stream) {
    ASSL_REQUIRE(sums && counts && centroids && K > 0 && D > 0);
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(K), dim3(64), 0, S_(stream), sums, counts, D, centroids);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_kmeans_update_mean(const float* sums, const int* counts, int K, int D, float* centroids, void* stream) {
    ASSL_REQUIRE(sums && counts && centroids && K > 0 && D > 0);
    hipLaunchKernelGGL(kmeans_update_mean_kernel, dim3(K), dim3(64), 0, S_(stream), sums, counts, D, centroids);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_row_sqnorm(const float* x, int rows, int D, float scale, float* out, void* stream) {
    ASSL_REQUIRE(x && out && rows > 0 && D > 0);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(rows), dim3(64), 0, S_(stream), x, D, scale, out);
    ASSL_LAUNCH_CHECK();
}

// cnt: 1 int of scratch.  dlogits may be NULL (forward only).
extern "C" int audiossl_ce_rows(int dtype, const float* logits, const long long* target, int B, int K, int ignore_index, int* cnt,
                                float* loss_out, void* dlogits, void* stream) {
    ASSL_REQUIRE(logits && target && cnt && loss_out && B > 0 && K > 0 && (dtype == 0 || dtype == 1));
    hipStream_t s = S_(stream);
    ASSL_ZERO_ALWAYS(cnt, sizeof(int), s);
    hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, s, target, B, ignore_index, cnt);
    if (dtype == 0) hipLaunchKernelGGL(ce_rows_kernel<float>, dim3(B), dim3(256), 0, s, logits, target, K, ignore_index, cnt, loss_out, (float*)dlogits);
    else            hipLaunchKernelGGL(ce_rows_kernel<bf16>, dim3(B), dim3(256), 0, s, logits, target, K, ignore_index, cnt, loss_out, (bf16*)dlogits);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_softmax_rows_fwd(const float* x, float* y, int M, int C, void* stream) {
    ASSL_REQUIRE(x && y && M > 0 && C > 0);
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, S_(stream), x, y, M, C);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_softmax_rows_bwd(const float* y, const float* gy, float* gx, int M, int C, void* stream) {
    ASSL_REQUIRE(y && gy && gx && M > 0 && C > 0);
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, S_(stream), y, gy, gx, M, C);
    ASSL_LAUNCH_CHECK();
}

// seg: n_seg x {int64 offset, int64 numel, int32 flags, int32 pad} (device); lr: n_seg floats (device);
// norms: 2*n_seg doubles of scratch (zeroed here).
extern "C" int audiossl_lars_step(float* p, const float* g, float* mu, const void* seg, int n_seg, const float* lr, float weight_decay,
                                  float momentum, float eta, float grad_scale, double* norms, void* stream) {
    ASSL_REQUIRE(p && g && mu && seg && lr && norms && n_seg > 0);
    hipStream_t s = S_(stream);
    ASSL_ZERO_ALWAYS(norms, sizeof(double) * 2 * n_seg, s);
    const LarsSeg* sg = static_cast<const LarsSeg*>(seg);
    hipLaunchKernelGGL(lars_norms_kernel, dim3(64, n_seg), dim3(256), 0, s, p, g, sg, weight_decay, grad_scale, norms);
    hipLaunchKernelGGL(lars_update_kernel, dim3(64, n_seg), dim3(256), 0, s, p, g, mu, sg, norms, lr, weight_decay, momentum, eta,
                       grad_scale);
    ASSL_LAUNCH_CHECK();
}

// seg as for lars_step; flag bit 2 (value 4) marks a tensor that has no gradient this step.  mu must start at zero.
extern "C" int audiossl_larc_step(float* p, const float* g, float* mu, const void* seg, int n_seg, float lr, float weight_decay,
                                  float momentum, float trust_coefficient, float eps, int clip, float grad_scale, double* norms,
                                  void* stream) {
    ASSL_REQUIRE(p && g && mu && seg && norms && n_seg > 0);
    hipStream_t s = S_(stream);
    ASSL_ZERO_ALWAYS(norms, sizeof(double) * 2 * n_seg, s);
    const LarsSeg* sg = static_cast<const LarsSeg*>(seg);
    hipLaunchKernelGGL(larc_norms_kernel, dim3(64, n_seg), dim3(256), 0, s, p, g, sg, grad_scale, norms);
    hipLaunchKernelGGL(larc_update_kernel, dim3(64, n_seg), dim3(256), 0, s, p, g, mu, sg, norms, lr, weight_decay, momentum,
                       trust_coefficient, eps, clip, grad_scale);
    ASSL_LAUNCH_CHECK();
}
