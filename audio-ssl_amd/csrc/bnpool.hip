// BatchNorm(train) statistics / finalize, fused BN+ReLU+MaxPool2 forward and backward on the channels-last
// activations of conv blocks 2 and 3 (`src/encoder/audiontt.py:52-60, 81-93`), the per-layer temporal means
// x_1..x_3 (`audiontt.py:76-79`), im2col and the conv weight (un)packers.
// Activation layout everywhere: [N][T][F][C=64] (time, mel, channel) = the reference's
// x.permute(0,3,2,1), so feature index d*64+c of the reference is contiguous memory here.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------- column statistics
// x [M][C] row-major (ld).  sum/sumsq [C] in fp64 (atomics, caller zeroes).  C % 64 == 0.
// A block owns a slab of 64 columns (8 lanes x 8 elements = one 128-byte line per row) and 32 rows per iteration;
// blockIdx.y picks the slab, blockIdx.x a chunk of rows, blockIdx.z the group (x is [G][M][C], outputs [G][C]).
template <typename T_, bool SQ>
__global__ __launch_bounds__(256) void colstats_kernel(const T_* __restrict__ x, long M, int C, long ld, int rows_per_block,
                                                       double* __restrict__ sum, double* __restrict__ sumsq) {
    __shared__ float red[256][17];
    const int cg = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    const int col0 = blockIdx.y * 64 + cg * 8;
    const long row_begin = (long)blockIdx.x * rows_per_block;
    const long row_end = min(M, row_begin + rows_per_block);
    x += (long)blockIdx.z * M * ld;
    sum += (long)blockIdx.z * C;
    if (SQ) sumsq += (long)blockIdx.z * C;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
    if (col0 < C)                                            // C % 8 == 0: the last slab may be partial (the MViT widths are 96 * 2^k)
        for (long r = row_begin + r0; r < row_end; r += 32) {
            const Vec8<T_> v = Vec8<T_>::load(x + r * ld + col0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float f = v.get(i); s[i] += f; if (SQ) q[i] += f * f; }
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][8 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int g = threadIdx.x >> 4, k = threadIdx.x & 15;
        if (SQ || k < 8) {
            double t = 0.0;
            for (int r = 0; r < 32; ++r) t += (double)red[r * 8 + g][k];
            const int c = blockIdx.y * 64 + g * 8 + (k & 7);
            if (c < C) {
                if (k < 8) atomicAdd(&sum[c], t);
                else       atomicAdd(&sumsq[c], t);
            }
        }
    }
}

// gamma/beta may be null (affine=False).  running_* may be null.
// Groups are independent batches normalised by the SAME layer (e.g. the two views of the projector): outputs are
// [G][C]; the running statistics are updated group after group, like the reference's successive module calls.
__global__ void bn_finalize_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq, int groups, double count,
                                   int C, const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* scale, float* shift,
                                   float* save_mean, float* save_rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
    for (int k = 0; k < groups; ++k) {
        const int o = k * C + c;
        const double mean = sum[o] / count;
        double var = sumsq[o] / count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        scale[o] = (float)(g * rstd);
        shift[o] = (float)(b - mean * g * rstd);
        save_mean[o] = (float)mean;
        save_rstd[o] = (float)rstd;
        rm = (1.f - momentum) * rm + momentum * (float)mean;
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rv = (1.f - momentum) * rv + momentum * (float)unb;
    }
    if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

// ------------------------------------------------------------------------------- BN + ReLU + MaxPool2
// Y [N][Ti][Fi][64] -> P [N][To][Fo][64]; one thread per (n,to,fo,8-channel group)
template <typename TY, typename T_>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const TY* __restrict__ Y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, T_* __restrict__ P,
                                                               int N, int Ti, int Fi) {
    const int To = Ti / 2, Fo = Fi / 2;
    const long total = (long)N * To * Fo * 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx & 7);
    long r = idx >> 3;
    const int fo = (int)(r % Fo); r /= Fo;
    const int to = (int)(r % To);
    const int n = (int)(r / To);
    float sc[8], sh[8], m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale[c8 * 8 + i]; sh[i] = shift[c8 * 8 + i]; m[i] = 0.f; }
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const Vec8<TY> v = Vec8<TY>::load(Y + ((((long)n * Ti + 2 * to + dt) * Fi + 2 * fo + df) * 64 + c8 * 8));
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], sc[i] * v.get(i) + sh[i]);      // m starts at 0 = ReLU
        }
    Vec8<T_> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.set(i, m[i]);
    o.store(P + idx * 8);
}

// The same with the BatchNorm finalisation folded in (train mode, 64 channels): every workgroup turns the fp64 column sums into
// scale / shift itself (64 lanes, a few flops each) instead of waiting for a one-workgroup bn_finalize launch between the
// convolution and the pooling - one graph node and one dependent-launch gap less per conv block; workgroup 0 also stores
// scale / shift / mean / rstd for the backward and updates the running buffers.
template <typename TY, typename T_>
__global__ __launch_bounds__(256) void bn_relu_pool_train_fwd_kernel(const TY* __restrict__ Y, const double* __restrict__ sum,
                                                                     const double* __restrict__ sumsq, int stat_rep, double count,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float* running_mean, float* running_var, float momentum, float eps,
                                                                     T_* __restrict__ P, float* __restrict__ scale,
                                                                     float* __restrict__ shift, float* __restrict__ save_mean,
                                                                     float* __restrict__ save_rstd, int N, int Ti, int Fi) {
    __shared__ float sc_s[64], sh_s[64];
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        const double g = (double)gamma[c], b = (double)beta[c];
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < stat_rep; ++r) { s1 += sum[r * 128 + c]; s2 += sumsq[r * 128 + c]; }      // fold the replicas
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const float sc = (float)(g * rstd), sh = (float)(b - mean * g * rstd);
        sc_s[c] = sc; sh_s[c] = sh;
        if (blockIdx.x == 0) {
            scale[c] = sc; shift[c] = sh; save_mean[c] = (float)mean; save_rstd[c] = (float)rstd;
            if (running_mean) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    }
    __syncthreads();
    const int To = Ti / 2, Fo = Fi / 2;
    const long total = (long)N * To * Fo * 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx & 7);
    long r = idx >> 3;
    const int fo = (int)(r % Fo); r /= Fo;
    const int to = (int)(r % To);
    const int n = (int)(r / To);
    float sc[8], sh[8], m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = sc_s[c8 * 8 + i]; sh[i] = sh_s[c8 * 8 + i]; m[i] = 0.f; }
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const Vec8<TY> v = Vec8<TY>::load(Y + ((((long)n * Ti + 2 * to + dt) * Fi + 2 * fo + df) * 64 + c8 * 8));
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], sc[i] * v.get(i) + sh[i]);
        }
    Vec8<T_> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.set(i, m[i]);
    o.store(P + idx * 8);
}

// xl[n][f*64+c] = mean_t P[n][t][f][c].  A workgroup owns one image and a strip of 64 vectors (512 features); its four
// waves take the time rows t = w, w + 4, ... (1 KB contiguous per wave and row) and fold through LDS.  (One thread per
// output vector walking all To rows serially ran at 2 TB/s: 8 waves per CU, one dependent 16-byte load in flight each.)
template <typename T_, typename TO>
__device__ __forceinline__ void tmean_body(const T_* __restrict__ P, TO* __restrict__ xl, int N, int To, int Fo);

template <typename T_, typename TO>
__global__ __launch_bounds__(256) void tmean_fwd_kernel(const T_* __restrict__ P, TO* __restrict__ xl, int N, int To, int Fo) {
    tmean_body<T_, TO>(P, xl, N, To, Fo);
}
// the three layer means of one encoder pass (x_1, x_2, x_3) in ONE launch: blockIdx.z = layer
struct Tmean3 { const void* P[3]; void* xl[3]; int To[3], Fo[3]; const float* parts; };
template <typename T_, typename TO>
__global__ __launch_bounds__(256) void tmean3_fwd_kernel(Tmean3 a, int N) {
    const int l = blockIdx.z;
    if ((int)blockIdx.x * 64 >= a.Fo[l] * 8) return;
    if (!a.P[l]) {                                               // x_1 from the parts the stem left (fixed order: reproducible)
        if (a.parts && l == 0) {
            const long row = (long)a.Fo[0] * 64, tot = (long)N * row;
            const long i = (long)blockIdx.y * row + blockIdx.x * 512 + threadIdx.x * 2;
            if (blockIdx.x * 512 + threadIdx.x * 2 < row) {
                float2 s = *reinterpret_cast<const float2*>(a.parts + i);
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const float2 v = *reinterpret_cast<const float2*>(a.parts + k * tot + i);
                    s.x += v.x; s.y += v.y;
                }
                *reinterpret_cast<float2*>(static_cast<float*>(a.xl[0]) + i) = s;
            }
        }
        return;
    }
    tmean_body<T_, TO>(static_cast<const T_*>(a.P[l]), static_cast<TO*>(a.xl[l]), N, a.To[l], a.Fo[l]);
}

template <typename T_, typename TO>
__device__ __forceinline__ void tmean_body(const T_* __restrict__ P, TO* __restrict__ xl, int N, int To, int Fo) {
    __shared__ float red[4][64][9];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nv = Fo * 8;                                        // vectors per time row
    const int v = blockIdx.x * 64 + lane;
    const int n = blockIdx.y;
    const bool live = v < nv;
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    if (live) {
        const T_* p = P + ((long)n * To * nv + v) * 8;
#pragma unroll 4
        for (int t = w; t < To; t += 4) {
            const Vec8<T_> x = Vec8<T_>::load(p + (long)t * nv * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += x.get(i);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[w][lane][i] = s[i];
    __syncthreads();
    if (w == 0 && live) {
        Vec8<TO> o;
        const float inv = 1.f / (float)To;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.set(i, (red[0][lane][i] + red[1][lane][i] + red[2][lane][i] + red[3][lane][i]) * inv);
        o.store(xl + ((long)n * nv + v) * 8);
    }
}

// Backward, pass 1: dbeta_c = sum routed grad, dgamma_c = sum routed grad * xhat   (fp32 atomics into stat[2][64])
// Backward, pass 2: dY = gamma*rstd * (dyhat_routed - dbeta/n - xhat*dgamma/n) at EVERY position (incl. the
// unpooled last time row when Ti is odd).
template <typename TY, bool APPLY, typename TG, typename T_>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_kernel(const TY* __restrict__ Y, const TG* __restrict__ dP,
                                                               const TG* __restrict__ dxl, float inv_To,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ stat, float inv_count,
                                                               T_* __restrict__ dY, int N, int Ti, int Fi, int fold,
                                                               float* dgamma, float* dbeta) {
    __shared__ float red[256][16];
    // APPLY with fold: the 32 replicas of the statistics pass are folded here, once per workgroup and in replica order (what the
    // stat_reduce launch between the two passes did), and workgroup 0 adds the sums to the parameter gradients (the add_stat
    // launch behind the apply pass): two small launches and their gaps off the serial chain of the convolution backward
    __shared__ float folded[128];
    if (APPLY && fold) {
        if (threadIdx.x < 128) {
            float t = 0.f;
            for (int r = 0; r < 32; ++r) t += stat[128 + r * 128 + threadIdx.x];
            folded[threadIdx.x] = t;
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < 64 && dgamma) {
            dbeta[threadIdx.x] += folded[threadIdx.x];
            dgamma[threadIdx.x] += folded[64 + threadIdx.x];
        }
    }
    const int To = Ti / 2, Fo = Fi / 2;
    const int Tq = (Ti + 1) / 2;                 // quads along time, the last one may be half outside the pooled area
    const long total = (long)N * Tq * Fo * 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < total;
    float db[8], dg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { db[i] = 0.f; dg[i] = 0.f; }
    if (live) {
        const int c8 = (int)(idx & 7);
        long r = idx >> 3;
        const int fo = (int)(r % Fo); r /= Fo;
        const int tq = (int)(r % Tq);
        const int n = (int)(r / Tq);
        const bool pooled = tq < To;
        float sc[8], sh[8], mu[8], rs[8], g[8], cb[8], cg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c8 * 8 + i;
            sc[i] = scale[c]; sh[i] = shift[c]; mu[i] = mean[c]; rs[i] = rstd[c]; g[i] = 0.f;
            if (APPLY) { cb[i] = (fold ? folded[c] : stat[c]) * inv_count; cg[i] = (fold ? folded[64 + c] : stat[64 + c]) * inv_count; }
        }
        if (pooled) {
            const Vec8<TG> gp = Vec8<TG>::load(dP + ((((long)n * To + tq) * Fo + fo) * 64 + c8 * 8));
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = gp.get(i);
            if (dxl) {
                const Vec8<TG> gx = Vec8<TG>::load(dxl + ((long)n * Fo * 64 + fo * 64 + c8 * 8));
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] += gx.get(i) * inv_To;
            }
        }
        Vec8<TY> y[4];
        int best[8];
        float bm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { best[i] = 0; bm[i] = 0.f; }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int df = p >> 1, dt = p & 1;
            const int t = 2 * tq + dt;
            if (t < Ti) y[p] = Vec8<TY>::load(Y + ((((long)n * Ti + t) * Fi + 2 * fo + df) * 64 + c8 * 8));
            else y[p] = Vec8<TY>::zero();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = sc[i] * y[p].get(i) + sh[i];
                if (p == 0 || a > bm[i]) { bm[i] = a; best[i] = p; }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float da = (pooled && bm[i] > 0.f) ? g[i] : 0.f;
            g[i] = da;
            if (!APPLY) {
                float yb = y[0].get(i);
#pragma unroll
                for (int p = 1; p < 4; ++p) yb = best[i] == p ? y[p].get(i) : yb;
                db[i] = da;
                dg[i] = da * (yb - mu[i]) * rs[i];
            }
        }
        if (APPLY) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int df = p >> 1, dt = p & 1;
                const int t = 2 * tq + dt;
                if (t >= Ti) continue;
                Vec8<T_> o;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xhat = (y[p].get(i) - mu[i]) * rs[i];
                    const float da = best[i] == p ? g[i] : 0.f;
                    o.set(i, sc[i] * (da - cb[i] - xhat * cg[i]));
                }
                o.store(dY + ((((long)n * Ti + t) * Fi + 2 * fo + df) * 64 + c8 * 8));
            }
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = db[i]; red[threadIdx.x][8 + i] = dg[i]; }
        __syncthreads();
        if (threadIdx.x < 128) {
            // thread -> (kind, channel): 64 channels x {dbeta, dgamma}; channel c lives in threads with (tid & 7) == c/8
            const int kind = threadIdx.x >> 6, c = threadIdx.x & 63;
            float t = 0.f;
            for (int r = (c >> 3); r < 256; r += 8) t += red[r][kind * 8 + (c & 7)];
            atomicAdd(&stat[128 + (blockIdx.x & 31) * 128 + kind * 64 + c], t);      // one of 32 replicas (contention)
        }
    }
}

// Backward, pass 1 from the POOLED forward output: the routed gradient is non-zero only at the window's arg-max and only where the
// pooled activation P = relu(max a) is positive, and there a = P, so xhat = ((P - shift) / scale - mean) * rstd needs no look at Y:
// 26 + 26 MB read instead of 105 + 26 (block 2, B = 512).  P carries one bf16 rounding of a, as Y carries one of y: the sums move
// by rounding noise only (tests/test_gpu_kernels.py compares the two).  A zero scale (gamma == 0: xhat is not recoverable) adds 0.
template <typename TP, typename TG>
__global__ __launch_bounds__(256) void bn_pool_bwd_stats_p_kernel(const TP* __restrict__ P, const TG* __restrict__ dP,
                                                                  const TG* __restrict__ dxl, float inv_To,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  float* __restrict__ stat, int N, int To, int Fo) {
    __shared__ float red[256][16];
    const long total = (long)N * To * Fo * 8;
    float db[8], dg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { db[i] = 0.f; dg[i] = 0.f; }
    // the stride is a multiple of 8: a thread keeps its channel chunk over the trips; xhat = a * k1 + k0 per channel
    const int c8 = threadIdx.x & 7;
    float k1[8], k0[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        const float sc = scale[c], inv = sc != 0.f ? 1.f / sc : 0.f;
        k1[i] = inv * rstd[c];
        k0[i] = sc != 0.f ? -(shift[c] * inv + mean[c]) * rstd[c] : 0.f;
    }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pos = idx >> 3;                                   // (n, to, fo)
        const Vec8<TP> pv = Vec8<TP>::load(P + pos * 64 + c8 * 8);
        const Vec8<TG> gp = Vec8<TG>::load(dP + pos * 64 + c8 * 8);
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = gp.get(i);
        if (dxl) {
            const long n = pos / ((long)To * Fo);
            const int fo = (int)(pos % Fo);
            const Vec8<TG> gx = Vec8<TG>::load(dxl + (n * Fo + fo) * 64 + c8 * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] += gx.get(i) * inv_To;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = pv.get(i);
            const float da = a > 0.f ? g[i] : 0.f;
            db[i] += da;
            dg[i] += da * (a * k1[i] + k0[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = db[i]; red[threadIdx.x][8 + i] = dg[i]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int kind = threadIdx.x >> 6, c = threadIdx.x & 63;
        float t = 0.f;
        for (int r = (c >> 3); r < 256; r += 8) t += red[r][kind * 8 + (c & 7)];
        atomicAdd(&stat[128 + (blockIdx.x & 31) * 128 + kind * 64 + c], t);
    }
}

// stat[0..127] = sum over the 32 replicas stat[128 + r*128 + ...]
__global__ void stat_reduce_kernel(float* stat) {
    const int i = threadIdx.x;
    if (i >= 128) return;
    float t = 0.f;
    for (int r = 0; r < 32; ++r) t += stat[128 + r * 128 + i];
    stat[i] = t;
}

// -------------------------------------------------------------------------------------------- im2col 3x3
// X [N][Ti][Fi][64] -> col [N*Ti*Fi][9*64], col[pix][tap*64+c] = X[n][t+kw-1][f+kh-1][c], tap = kh*3+kw
// (kh walks mel = torch H, kw walks time = torch W)
template <typename T_>
__global__ __launch_bounds__(256) void im2col_kernel(const T_* __restrict__ X, T_* __restrict__ col, int N, int Ti, int Fi) {
    const long total = (long)N * Ti * Fi * 72;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int k8 = (int)(idx % 72);
    long pix = idx / 72;
    const int tap = k8 >> 3, c8 = k8 & 7;
    const int f = (int)(pix % Fi);
    const long r = pix / Fi;
    const int t = (int)(r % Ti);
    const long n = r / Ti;
    const int ff = f + tap / 3 - 1, tt = t + tap % 3 - 1;
    Vec8<T_> v = Vec8<T_>::zero();
    if (ff >= 0 && ff < Fi && tt >= 0 && tt < Ti) v = Vec8<T_>::load(X + (((n * Ti + tt) * Fi + ff) * 64 + c8 * 8));
    v.store(col + idx * 8);
}

// W fp32 [co][ci][3][3] (torch) -> Wf [co][tap*64+ci] (forward) and Wd [ci][tap*64+co] with the taps flipped (dgrad)
template <typename T_>
__global__ void pack_conv_w_kernel(const float* __restrict__ W, T_* __restrict__ Wf, T_* __restrict__ Wd) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 64 * 576) return;
    const int row = idx / 576, k = idx % 576, tap = k / 64, c = k % 64;
    Wf[idx] = from_f32<T_>(W[(row * 64 + c) * 9 + tap]);                 // row = co, c = ci
    Wd[idx] = from_f32<T_>(W[(c * 64 + row) * 9 + (8 - tap)]);           // row = ci, c = co
}

// both 3x3 layers of an encoder in one launch (blockIdx.y = layer)
template <typename T_>
__global__ void pack_conv_w2_kernel(const float* __restrict__ Wa, T_* __restrict__ Wfa, T_* __restrict__ Wda,
                                    const float* __restrict__ Wb, T_* __restrict__ Wfb, T_* __restrict__ Wdb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 64 * 576) return;
    const float* W = blockIdx.y ? Wb : Wa;
    T_* Wf = blockIdx.y ? Wfb : Wfa;
    T_* Wd = blockIdx.y ? Wdb : Wda;
    const int row = idx / 576, k = idx % 576, tap = k / 64, c = k % 64;
    Wf[idx] = from_f32<T_>(W[(row * 64 + c) * 9 + tap]);
    Wd[idx] = from_f32<T_>(W[(c * 64 + row) * 9 + (8 - tap)]);
}

// dWp fp32 [co][tap*64+ci] -> dW [co][ci][3][3] +=
__global__ void unpack_conv_dw_kernel(const float* __restrict__ dWp, float* __restrict__ dW) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 64 * 576) return;
    const int co = idx / 576, k = idx % 576, tap = k / 64, ci = k % 64;
    dW[(co * 64 + ci) * 9 + tap] += dWp[idx];
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) do { if ((dtype) == 0) { CALL_F32; } else { CALL_BF16; } } while (0)

extern "C" int audiossl_colstats(int dtype, const void* x, int groups, long M, int C, long ld, int want_sq, double* sum,
                                 double* sumsq, void* stream) {
    ASSL_REQUIRE(x && sum && groups > 0 && M > 0 && C > 0 && (C % 8) == 0 && (ld % 8) == 0);
    ASSL_REQUIRE((dtype == 0 || dtype == 1) && (!want_sq || sumsq));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nb = sizeof(double) * C * groups;
    if (want_sq && sumsq == sum + (size_t)C * groups) {            // contiguous scratch: one memset node
        ASSL_ZERO(sum, 2 * nb, s);
    } else {
        ASSL_ZERO(sum, nb, s);
        if (want_sq) ASSL_ZERO(sumsq, nb, s);
    }
    const int slabs = (C + 63) / 64;
    long it = (M * slabs * groups + 32L * 2048 - 1) / (32L * 2048);  // aim for ~2048 blocks in total, 1..128 iterations each
    it = it < 1 ? 1 : (it > 128 ? 128 : it);
    const int rpb = 32 * (int)it;
    const dim3 grid(ceil_div(M, rpb), slabs, groups);
#define CS(TT, SQ) hipLaunchKernelGGL((colstats_kernel<TT, SQ>), grid, dim3(256), 0, s, static_cast<const TT*>(x), M, C, ld, rpb, sum, sumsq)
    if (dtype == 0) { if (want_sq) CS(float, true); else CS(float, false); }
    else            { if (want_sq) CS(bf16, true);  else CS(bf16, false); }
#undef CS
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_bn_finalize(const double* sum, const double* sumsq, int groups, double count, int C,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_rstd,
                                    void* stream) {
    ASSL_REQUIRE(sum && sumsq && scale && shift && save_mean && save_rstd && C > 0 && count > 0 && groups > 0);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sum, sumsq,
                       groups, count, C, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_rstd);
    ASSL_LAUNCH_CHECK();
}

// ydtype: storage type of the pre-BatchNorm conv output Y (0 = fp32 also on the bf16 path, see colbn_fwd)
extern "C" int audiossl_bn_relu_pool_fwd(int dtype, int ydtype, const void* Y, const float* scale, const float* shift, void* P,
                                         int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(Y && scale && shift && P && N > 0 && Ti >= 2 && Fi >= 2 && (Fi % 2) == 0 && (dtype == 0 || dtype == 1));
    ASSL_REQUIRE(ydtype == 0 || ydtype == dtype);
    const long total = (long)N * (Ti / 2) * (Fi / 2) * 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PF(TY, TO) hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<TY, TO>), dim3(ceil_div(total, 256)), dim3(256), 0, s, \
        static_cast<const TY*>(Y), scale, shift, static_cast<TO*>(P), N, Ti, Fi)
    if (dtype == 0) PF(float, float); else if (ydtype == 0) PF(float, bf16); else PF(bf16, bf16);
#undef PF
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_bn_relu_pool_train_fwd(int dtype, int ydtype, const void* Y, const double* sum, const double* sumsq,
                                               int stat_replicas, double count,
                                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                                               float momentum, float eps, void* P, float* scale, float* shift, float* save_mean,
                                               float* save_rstd, int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(Y && sum && sumsq && gamma && beta && P && scale && shift && save_mean && save_rstd && count > 0.0);
    ASSL_REQUIRE(stat_replicas >= 1 && (stat_replicas == 1 || sumsq == sum + 64));
    ASSL_REQUIRE(N > 0 && Ti >= 2 && Fi >= 2 && (Fi % 2) == 0 && (dtype == 0 || dtype == 1) && (ydtype == 0 || ydtype == dtype));
    const long total = (long)N * (Ti / 2) * (Fi / 2) * 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PF(TY, TO) hipLaunchKernelGGL((bn_relu_pool_train_fwd_kernel<TY, TO>), dim3(ceil_div(total, 256)), dim3(256), 0, s,          \
        static_cast<const TY*>(Y), sum, sumsq, stat_replicas, count, gamma, beta, running_mean, running_var, momentum, eps,      \
        static_cast<TO*>(P),                                                                                                      \
        scale, shift, save_mean, save_rstd, N, Ti, Fi)
    if (dtype == 0) PF(float, float); else if (ydtype == 0) PF(float, bf16); else PF(bf16, bf16);
#undef PF
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_tmean_fwd(int dtype, int out_f32, const void* P, void* xl, int N, int To, int Fo, void* stream) {
    ASSL_REQUIRE(P && xl && N > 0 && To > 0 && Fo > 0 && (dtype == 0 || dtype == 1));
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TM(TI, TO_) hipLaunchKernelGGL((tmean_fwd_kernel<TI, TO_>), dim3(ceil_div(Fo * 8, 64), N), dim3(256), 0, s, \
        static_cast<const TI*>(P), static_cast<TO_*>(xl), N, To, Fo)
    if (dtype == 0) TM(float, float); else if (out_f32) TM(bf16, float); else TM(bf16, bf16);
#undef TM
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_tmean3_fwd(int dtype, int out_f32, const void* P1, void* x1, const float* x1_parts, int To1, int Fo1,
                                   const void* P2, void* x2, int To2, int Fo2, const void* P3, void* x3, int To3, int Fo3, int N,
                                   void* stream) {
    ASSL_REQUIRE(x1 && P2 && x2 && P3 && x3 && N > 0 && To1 > 0 && To2 > 0 && To3 > 0 && Fo1 > 0 && Fo2 > 0 && Fo3 > 0);
    ASSL_REQUIRE((P1 != nullptr) != (x1_parts != nullptr) && (!x1_parts || (out_f32 && dtype == 1)));
    ASSL_REQUIRE(dtype == 0 || dtype == 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Tmean3 a{{P1, P2, P3}, {x1, x2, x3}, {To1, To2, To3}, {Fo1, Fo2, Fo3}, x1_parts};
    const int gx = ceil_div(max(Fo1, max(Fo2, Fo3)) * 8, 64);
#define TM3(TI, TO_) hipLaunchKernelGGL((tmean3_fwd_kernel<TI, TO_>), dim3(gx, N, 3), dim3(256), 0, s, a, N)
    if (dtype == 0) TM3(float, float); else if (out_f32) TM3(bf16, float); else TM3(bf16, bf16);
#undef TM3
    ASSL_LAUNCH_CHECK();
}

// stat: 33*128 floats scratch (zeroed here).  dgamma/dbeta accumulated (+=) by a tiny tail launch.
namespace {
__global__ void add_stat_kernel(const float* __restrict__ stat, float* dgamma, float* dbeta) {
    const int c = threadIdx.x;
    if (c < 64) { dbeta[c] += stat[c]; dgamma[c] += stat[64 + c]; }
}
}  // namespace

// phase: 0 = everything (statistics, fold, apply, parameter gradients); 1 = statistics + fold only (stat[0..127] = this rank's
// sum g / sum g xhat); 2 = apply with the sums in `gstat` over `count` elements (the all-reduced sums of SyncBatchNorm) and add
// this rank's own sums (stat) to dgamma / dbeta
static int bn_relu_pool_bwd_run(int phase, int dtype, int ydtype, int gdtype, const void* Y, const void* P, const void* dP, const void* dxl,
                                const float* scale, const float* shift, const float* mean, const float* rstd, float* stat,
                                const float* gstat, double count, void* dY, float* dgamma, float* dbeta, int N, int Ti, int Fi,
                                void* stream) {
    ASSL_REQUIRE(Y && dP && scale && shift && mean && rstd && stat);
    ASSL_REQUIRE(N > 0 && Ti >= 2 && Fi >= 2 && (Fi % 2) == 0 && (dtype == 0 || dtype == 1) && (gdtype == 0 || gdtype == dtype));
    ASSL_REQUIRE(ydtype == 0 || ydtype == dtype);
    ASSL_REQUIRE(phase == 1 || (dY && dgamma && dbeta));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (phase != 2) ASSL_ZERO(stat, sizeof(float) * 128 * 33, s);
    const long total = (long)N * ((Ti + 1) / 2) * (Fi / 2) * 8;
    const int grid = ceil_div(total, 256);
    const float inv_To = 1.f / (float)(Ti / 2);
    const float inv_count = (float)(1.0 / (phase == 2 ? count : (double)N * Ti * Fi));
    float* apply_stat = phase == 2 ? const_cast<float*>(gstat) : stat;
    const int fold = phase == 0;               // one rank: fold + parameter gradients inside the apply pass (no stat_reduce / add_stat launch)
#define BW(TY, AP, TG, TO, ST) hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<TY, AP, TG, TO>), dim3(grid), dim3(256), 0, s,              \
        static_cast<const TY*>(Y), static_cast<const TG*>(dP), static_cast<const TG*>(dxl), inv_To, scale, shift, mean, rstd, ST, \
        inv_count, static_cast<TO*>(dY), N, Ti, Fi, (AP) ? fold : 0, dgamma, dbeta)
#define BWP(TG, TO) hipLaunchKernelGGL((bn_pool_bwd_stats_p_kernel<TO, TG>), dim3(pgrid), dim3(256), 0, s, static_cast<const TO*>(P),          \
        static_cast<const TG*>(dP), static_cast<const TG*>(dxl), inv_To, scale, shift, mean, rstd, stat, N, Ti / 2, Fi / 2)
#define BW2(TY, TG, TO) do { if (phase != 2) { if (P) BWP(TG, TO); else BW(TY, false, TG, TO, stat);                                       \
                                               if (!fold) hipLaunchKernelGGL(stat_reduce_kernel, dim3(1), dim3(128), 0, s, stat); }          \
                             if (phase != 1) BW(TY, true, TG, TO, apply_stat); } while (0)
    const long ptotal = (long)N * (Ti / 2) * (Fi / 2) * 8;
    const int pgrid = (int)min((long)2048, (ptotal + 255) / 256);         // 2,048 x 256 threads: a multiple of 8, see the kernel
    if (dtype == 0) BW2(float, float, float);
    else if (ydtype == 0 && gdtype == 0) BW2(float, float, bf16);
    else if (ydtype == 0) BW2(float, bf16, bf16);
    else if (gdtype == 0) BW2(bf16, float, bf16);
    else BW2(bf16, bf16, bf16);
#undef BW2
#undef BWP
#undef BW
    if (phase == 2) hipLaunchKernelGGL(add_stat_kernel, dim3(1), dim3(64), 0, s, stat, dgamma, dbeta);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_bn_relu_pool_bwd(int dtype, int ydtype, int gdtype, const void* Y, const void* dP, const void* dxl, const float* scale,
                                         const float* shift, const float* mean, const float* rstd, float* stat, void* dY,
                                         float* dgamma, float* dbeta, int N, int Ti, int Fi, void* stream) {
    return bn_relu_pool_bwd_run(0, dtype, ydtype, gdtype, Y, nullptr, dP, dxl, scale, shift, mean, rstd, stat, nullptr, 0.0, dY, dgamma, dbeta, N,
                                Ti, Fi, stream);
}
extern "C" int audiossl_bn_relu_pool_bwd_p(int dtype, int ydtype, int gdtype, const void* Y, const void* P, const void* dP, const void* dxl,
                                           const float* scale, const float* shift, const float* mean, const float* rstd, float* stat,
                                           void* dY, float* dgamma, float* dbeta, int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(P);
    return bn_relu_pool_bwd_run(0, dtype, ydtype, gdtype, Y, P, dP, dxl, scale, shift, mean, rstd, stat, nullptr, 0.0, dY, dgamma, dbeta, N,
                                Ti, Fi, stream);
}
extern "C" int audiossl_bn_relu_pool_bwd_stats(int dtype, int ydtype, int gdtype, const void* Y, const void* dP, const void* dxl,
                                               const float* scale, const float* shift, const float* mean, const float* rstd,
                                               float* stat, int N, int Ti, int Fi, void* stream) {
    return bn_relu_pool_bwd_run(1, dtype, ydtype, gdtype, Y, nullptr, dP, dxl, scale, shift, mean, rstd, stat, nullptr, 0.0, nullptr, nullptr,
                                nullptr, N, Ti, Fi, stream);
}
extern "C" int audiossl_bn_relu_pool_bwd_apply(int dtype, int ydtype, int gdtype, const void* Y, const void* dP, const void* dxl,
                                               const float* scale, const float* shift, const float* mean, const float* rstd,
                                               float* stat, const float* gstat, double count_global, void* dY, float* dgamma,
                                               float* dbeta, int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(gstat && count_global > 1.0);
    return bn_relu_pool_bwd_run(2, dtype, ydtype, gdtype, Y, nullptr, dP, dxl, scale, shift, mean, rstd, stat, gstat, count_global, dY, dgamma,
                                dbeta, N, Ti, Fi, stream);
}

extern "C" int audiossl_im2col3x3(int dtype, const void* X, void* col, int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(X && col && N > 0 && Ti > 0 && Fi > 0 && (dtype == 0 || dtype == 1));
    const long total = (long)N * Ti * Fi * 72;
    ASSL_REQUIRE((total + 255) / 256 < 2147483647L);
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const float*>(X),
                           static_cast<float*>(col), N, Ti, Fi),
        hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const bf16*>(X),
                           static_cast<bf16*>(col), N, Ti, Fi));
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_pack_conv_w(int dtype, const float* W, void* Wf, void* Wd, void* stream) {
    ASSL_REQUIRE(W && Wf && Wd && (dtype == 0 || dtype == 1));
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(pack_conv_w_kernel<float>, dim3(144), dim3(256), 0, s, W, static_cast<float*>(Wf), static_cast<float*>(Wd)),
        hipLaunchKernelGGL(pack_conv_w_kernel<bf16>, dim3(144), dim3(256), 0, s, W, static_cast<bf16*>(Wf), static_cast<bf16*>(Wd)));
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_pack_conv_w2(int dtype, const float* Wa, void* Wfa, void* Wda, const float* Wb, void* Wfb, void* Wdb, void* stream) {
    ASSL_REQUIRE(Wa && Wfa && Wda && Wb && Wfb && Wdb && (dtype == 0 || dtype == 1));
    hipStream_t s = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(pack_conv_w2_kernel<float>, dim3(144, 2), dim3(256), 0, s, Wa, static_cast<float*>(Wfa), static_cast<float*>(Wda),
                           Wb, static_cast<float*>(Wfb), static_cast<float*>(Wdb)),
        hipLaunchKernelGGL(pack_conv_w2_kernel<bf16>, dim3(144, 2), dim3(256), 0, s, Wa, static_cast<bf16*>(Wfa), static_cast<bf16*>(Wda),
                           Wb, static_cast<bf16*>(Wfb), static_cast<bf16*>(Wdb)));
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_unpack_conv_dw(const float* dWp, float* dW, void* stream) {
    ASSL_REQUIRE(dWp && dW);
    hipLaunchKernelGGL(unpack_conv_dw_kernel, dim3(144), dim3(256), 0, static_cast<hipStream_t>(stream), dWp, dW);
    ASSL_LAUNCH_CHECK();
}
