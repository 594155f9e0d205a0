// Encoder tail, loss heads and optimiser kernels (everything that is not a GEMM or a conv block).
//   maxmean   : `upstream_encoder.py:26-28`   max_T(x) + mean_T(x) and its backward (ReLU-gated)
//   colbn_*   : BatchNorm1d(train) of the Barlow projector (`upstream_expert.py:15-32`), apply + backward
//   barlow_*  : cross-correlation loss and its gradient (`upstream_expert.py:36-45`, `utils.py:185-189`)
//   l2norm / moco_ce / enqueue : MoCo InfoNCE head (`delores_m/upstream_expert.py:156-172, 231-264`)
//   sgd / ema / cast / dropout_mask : optimiser + parameter plumbing on flat buffers
#include "common.h"

namespace {

// --------------------------------------------------------------------------------------------- max + mean over time
// H [N][Tt][D] -> y [N][D] = max_t + mean_t ; arg [N][D] = first argmax
template <typename T_, typename TO>
__global__ __launch_bounds__(256) void maxmean_fwd_kernel(const T_* __restrict__ H, TO* __restrict__ y, uint8_t* __restrict__ arg,
                                                          int N, int Tt, int D) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // over N * D/8
    const int D8 = D / 8;
    if (idx >= (long)N * D8) return;
    const int d8 = (int)(idx % D8);
    const long n = idx / D8;
    float mx[8], sm[8];
    int am[8];
    for (int t = 0; t < Tt; ++t) {
        const Vec8<T_> v = Vec8<T_>::load(H + ((n * Tt + t) * D + d8 * 8));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float f = v.get(i);
            if (t == 0) { mx[i] = f; sm[i] = f; am[i] = 0; }
            else { sm[i] += f; if (f > mx[i]) { mx[i] = f; am[i] = t; } }
        }
    }
    Vec8<TO> o;
    const float inv = 1.f / (float)Tt;
#pragma unroll
    for (int i = 0; i < 8; ++i) { o.set(i, mx[i] + sm[i] * inv); arg[n * D + d8 * 8 + i] = (uint8_t)am[i]; }
    o.store(y + n * D + d8 * 8);
}

// dA[n][t][d] = (dy[n][d]/Tt + dy[n][d]*[t==arg]) * (H[n][t][d] > 0)      (TG = type of the incoming gradient dy)
template <typename T_, typename TG>
__global__ __launch_bounds__(256) void maxmean_bwd_kernel(const TG* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                          const T_* __restrict__ H, T_* __restrict__ dA, int N, int Tt, int D) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // over N * Tt * D/8
    const int D8 = D / 8;
    if (idx >= (long)N * Tt * D8) return;
    const int d8 = (int)(idx % D8);
    const long r = idx / D8;
    const int t = (int)(r % Tt);
    const long n = r / Tt;
    const Vec8<TG> g = Vec8<TG>::load(dy + n * D + d8 * 8);
    const Vec8<T_> h = Vec8<T_>::load(H + idx * 8);
    Vec8<T_> o;
    const float inv = 1.f / (float)Tt;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float gi = g.get(i);
        const float v = gi * inv + (arg[n * D + d8 * 8 + i] == t ? gi : 0.f);
        o.set(i, h.get(i) > 0.f ? v : 0.f);
    }
    o.store(dA + idx * 8);
}

// --------------------------------------------------------------------------------------------- BatchNorm1d pieces
// h = act(scale_g*a + shift_g), elementwise over [G][M][C] with per-group [G][C] scale / shift
template <typename TA, typename T_>
__global__ __launch_bounds__(256) void colbn_fwd_kernel(const TA* __restrict__ a, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu, T_* __restrict__ h, long M, int C,
                                                        int groups) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int C8 = C / 8;
    if (idx >= (long)groups * M * C8) return;
    const int c8 = (int)(idx % C8);
    const int grp = (int)(idx / (M * C8));
    scale += (long)grp * C;
    shift += (long)grp * C;
    const Vec8<TA> v = Vec8<TA>::load(a + idx * 8);
    Vec8<T_> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float f = scale[c8 * 8 + i] * v.get(i) + shift[c8 * 8 + i];
        o.set(i, relu ? fmaxf(f, 0.f) : f);
    }
    o.store(h + idx * 8);
}

// stats: sg[c] = sum_b g, sgx[c] = sum_b g*xhat with g = dh * (act > 0 if relu), xhat = (a-mean)*rstd   (fp64 atomics)
// Same slab mapping as colstats: a block = 64 columns x 32 rows per iteration, blockIdx.y = slab.
template <typename TA, typename TG>
__global__ __launch_bounds__(256) void colbn_bwd_stats_kernel(const TA* __restrict__ a, const TG* __restrict__ dh,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              int relu, long M, int C, int rows_per_block,
                                                              double* __restrict__ sg, double* __restrict__ sgx) {
    __shared__ float red[256][17];
    const int cg = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    const int col0 = blockIdx.y * 64 + cg * 8;
    const long rb = (long)blockIdx.x * rows_per_block, re = min(M, rb + rows_per_block);
    {   // group offsets: activations [G][M][C], statistics [G][C]
        const long go = (long)blockIdx.z * C, ro = (long)blockIdx.z * M * C;
        a += ro; dh += ro; scale += go; shift += go; mean += go; rstd += go; sg += go; sgx += go;
    }
    float s[8], q[8], sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[i] = 0.f; q[i] = 0.f;
        sc[i] = scale[col0 + i]; sh[i] = shift[col0 + i]; mu[i] = mean[col0 + i]; rs[i] = rstd[col0 + i];
    }
    for (long r = rb + r0; r < re; r += 32) {
        const Vec8<TA> va = Vec8<TA>::load(a + r * C + col0);
        const Vec8<TG> vg = Vec8<TG>::load(dh + r * C + col0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = va.get(i);
            float g = vg.get(i);
            if (relu && !(sc[i] * x + sh[i] > 0.f)) g = 0.f;
            s[i] += g;
            q[i] += g * (x - mu[i]) * rs[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][8 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int g = threadIdx.x >> 4, k = threadIdx.x & 15;
        double t = 0.0;
        for (int r = 0; r < 32; ++r) t += (double)red[r * 8 + g][k];
        const int c = blockIdx.y * 64 + g * 8 + (k & 7);
        if (k < 8) atomicAdd(&sg[c], t); else atomicAdd(&sgx[c], t);
    }
}

// da = scale * (g - sg/M - xhat*sgx/M); rows handled by block 0 also accumulate the parameter grads.
template <typename TA, typename TG, typename T_>
__global__ __launch_bounds__(256) void colbn_bwd_apply_kernel(const TA* __restrict__ a, const TG* __restrict__ dh,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              int relu, long M, int C, int groups, const double* __restrict__ sg,
                                                              const double* __restrict__ sgx, T_* __restrict__ da,
                                                              float* dgamma, float* dbeta, float invM) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int C8 = C / 8;
    if (idx >= (long)groups * M * C8) return;
    const int c8 = (int)(idx % C8);
    const long grow = idx / C8;
    const int grp = (int)(grow / M);
    const long row = grow - (long)grp * M;
    {
        const long go = (long)grp * C;
        scale += go; shift += go; mean += go; rstd += go; sg += go; sgx += go;
    }
    const Vec8<TA> va = Vec8<TA>::load(a + idx * 8);
    const Vec8<TG> vg = Vec8<TG>::load(dh + idx * 8);
    // the eight per-column constants as vector loads (they were 6 scalar loads per element)
    const Vec8<float> vsc = Vec8<float>::load(scale + c8 * 8), vsh = Vec8<float>::load(shift + c8 * 8);
    const Vec8<float> vmu = Vec8<float>::load(mean + c8 * 8), vrs = Vec8<float>::load(rstd + c8 * 8);
    double dsg[8], dsx[8];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const double2 p = *reinterpret_cast<const double2*>(sg + c8 * 8 + i), q = *reinterpret_cast<const double2*>(sgx + c8 * 8 + i);
        dsg[i] = p.x; dsg[i + 1] = p.y; dsx[i] = q.x; dsx[i + 1] = q.y;
    }
    Vec8<T_> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        const float x = va.get(i);
        float g = vg.get(i);
        if (relu && !(vsc.get(i) * x + vsh.get(i) > 0.f)) g = 0.f;
        const float xhat = (x - vmu.get(i)) * vrs.get(i);
        const float mg = (float)dsg[i] * invM, mgx = (float)dsx[i] * invM;
        o.set(i, vsc.get(i) * (g - mg - xhat * mgx));
        if (row == 0 && dgamma) { atomicAdd(&dgamma[c], (float)sgx[c]); atomicAdd(&dbeta[c], (float)sg[c]); }
    }
    o.store(da + idx * 8);
}

// ------------------------------------------------------------------- fused BatchNorm1d(train) for short batches
// The projector batches are short (M = 512 rows per view): statistics, finalisation and normalisation as three launches
// cost three dispatch latencies of a serial chain for a few microseconds of work.  Here one workgroup owns a strip of 32
// columns (4 lanes x 8 columns per row) and keeps its M <= 64 * NR rows in registers between the statistics pass and the
// normalisation.  Column sums: fp32 per thread (NR rows), fp64 across the 64 row-threads - like colstats_kernel.
template <int NS>
__device__ __forceinline__ void strip_reduce(const float (&part)[NS][8], float (*red)[65], double* tot) {
    // part[k][i]: this thread's partial of statistic k, column (t & 3) * 8 + i  ->  tot[k * 32 + column]
    const int cl = threadIdx.x & 3, r0 = threadIdx.x >> 2;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) red[r0][cl * 8 + i] = part[k][i];
        __syncthreads();
        if (threadIdx.x < 32) {
            double t = 0.0;
            for (int r = 0; r < 64; ++r) t += (double)red[r][threadIdx.x];
            tot[k * 32 + threadIdx.x] = t;
        }
    }
    __syncthreads();
}

template <typename TA, typename T_, int NR>
__device__ __forceinline__ void colbn_train_fwd_body(const TA* __restrict__ a, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* running_mean,
                                                              float* running_var, float momentum, float eps, int relu, int groups,
                                                              int M, int C, T_* __restrict__ h, float* __restrict__ scale,
                                                              float* __restrict__ shift, float* __restrict__ save_mean,
                                                              float* __restrict__ save_rstd) {
    __shared__ float red[64][65];
    __shared__ double tot[64];
    __shared__ float sc_s[32], sh_s[32];
    const int cl = threadIdx.x & 3, r0 = threadIdx.x >> 2;
    const int col0 = blockIdx.x * 32 + cl * 8;
    const int c = blockIdx.x * 32 + threadIdx.x;             // threads 0..31: the column they finalise
    float rm = 0.f, rv = 0.f;
    if (threadIdx.x < 32 && running_mean) { rm = running_mean[c]; rv = running_var[c]; }
    // the rows of group g + 1 are loaded before group g is reduced, normalised and stored (the groups run one after the other inside
    // a workgroup because the running statistics are updated group after group): one memory round trip of the strip is hidden
    // behind the other's arithmetic - two views of 512 rows: 16 -> ~10 us per launch on the serial chain of the projector heads
    Vec8<TA> vn[NR];
    auto fetch = [&](int g, Vec8<TA> (&dst)[NR]) {
        const TA* ag = a + (long)g * M * C;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = r0 + 64 * j;
            if (r < M) dst[j] = Vec8<TA>::load(ag + (long)r * C + col0);
        }
    };
    fetch(0, vn);
    for (int g = 0; g < groups; ++g) {
        Vec8<TA> v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = vn[j];
        if (g + 1 < groups) fetch(g + 1, vn);
        float part[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { part[0][i] = 0.f; part[1][i] = 0.f; }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = r0 + 64 * j;
            if (r < M) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float f = v[j].get(i); part[0][i] += f; part[1][i] += f * f; }
            }
        }
        strip_reduce<2>(part, red, tot);
        if (threadIdx.x < 32) {
            const double gm = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
            const double mean = tot[threadIdx.x] / (double)M;
            double var = tot[32 + threadIdx.x] / (double)M - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            const float sc = (float)(gm * rstd), sh = (float)(bt - mean * gm * rstd);
            const long o = (long)g * C + c;
            scale[o] = sc; shift[o] = sh; save_mean[o] = (float)mean; save_rstd[o] = (float)rstd;
            sc_s[threadIdx.x] = sc; sh_s[threadIdx.x] = sh;
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            const double unb = M > 1 ? var * (double)M / ((double)M - 1.0) : var;
            rv = (1.f - momentum) * rv + momentum * (float)unb;
        }
        __syncthreads();
        float sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = sc_s[cl * 8 + i]; sh[i] = sh_s[cl * 8 + i]; }
        T_* hg = h + (long)g * M * C;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = r0 + 64 * j;
            if (r < M) {
                Vec8<T_> o;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float f = sc[i] * v[j].get(i) + sh[i]; o.set(i, relu ? fmaxf(f, 0.f) : f); }
                o.store(hg + (long)r * C + col0);
            }
        }
    }
    if (threadIdx.x < 32 && running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

template <typename TA, typename T_, int NR>
__global__ __launch_bounds__(256) void colbn_train_fwd_kernel(const TA* __restrict__ a, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* running_mean,
                                                              float* running_var, float momentum, float eps, int relu, int groups,
                                                              int M, int C, T_* __restrict__ h, float* __restrict__ scale,
                                                              float* __restrict__ shift, float* __restrict__ save_mean,
                                                              float* __restrict__ save_rstd) {
    colbn_train_fwd_body<TA, T_, NR>(a, gamma, beta, running_mean, running_var, momentum, eps, relu, groups, M, C, h, scale, shift,
                                     save_mean, save_rstd);
}

// Several layers of one shape in ONE launch (blockIdx.y / .z = problem): the BatchNorms of the three Barlow heads.
constexpr int MAXP = 4;
struct BnFwdMulti {
    const void* a[MAXP]; const float* gamma[MAXP]; const float* beta[MAXP]; float* rm[MAXP]; float* rv[MAXP];
    void* h[MAXP]; float* st[MAXP];                          // st: [4][G*C] = scale, shift, mean, rstd
};
template <typename TA, typename T_, int NR>
__global__ __launch_bounds__(256) void colbn_train_fwd_multi_kernel(BnFwdMulti m, float momentum, float eps, int relu, int groups,
                                                                    int M, int C) {
    const int p = blockIdx.y;
    const long GC = (long)groups * C;
    float* st = m.st[p];
    colbn_train_fwd_body<TA, T_, NR>(static_cast<const TA*>(m.a[p]), m.gamma[p], m.beta[p], m.rm[p], m.rv[p], momentum, eps, relu,
                                     groups, M, C, static_cast<T_*>(m.h[p]), st, st + GC, st + 2 * GC, st + 3 * GC);
}

// backward of the same: grid (C / 32, groups); sg / sgx as in colbn_bwd_stats_kernel, then da in the same launch
template <typename TA, typename TG, typename T_, int NR>
__device__ __forceinline__ void colbn_bwd_fused_body(const TA* __restrict__ a, const TG* __restrict__ dh,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     int relu, int M, int C, T_* __restrict__ da, float* dgamma, float* dbeta) {
    __shared__ float red[64][65];
    __shared__ double tot[64];
    const int cl = threadIdx.x & 3, r0 = threadIdx.x >> 2;
    const int col0 = blockIdx.x * 32 + cl * 8;
    const long go = (long)blockIdx.y * C, ro = (long)blockIdx.y * M * C;
    a += ro; dh += ro; da += ro;
    const Vec8<float> vsc = Vec8<float>::load(scale + go + col0), vsh = Vec8<float>::load(shift + go + col0);
    const Vec8<float> vmu = Vec8<float>::load(mean + go + col0), vrs = Vec8<float>::load(rstd + go + col0);
    float gq[NR][8], xh[NR][8];
    float part[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { part[0][i] = 0.f; part[1][i] = 0.f; }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + 64 * j;
        if (r < M) {
            const Vec8<TA> va = Vec8<TA>::load(a + (long)r * C + col0);
            const Vec8<TG> vg = Vec8<TG>::load(dh + (long)r * C + col0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float x = va.get(i);
                float g = vg.get(i);
                if (relu && !(vsc.get(i) * x + vsh.get(i) > 0.f)) g = 0.f;
                const float xhat = (x - vmu.get(i)) * vrs.get(i);
                gq[j][i] = g; xh[j][i] = xhat;
                part[0][i] += g; part[1][i] += g * xhat;
            }
        }
    }
    strip_reduce<2>(part, red, tot);
    if (threadIdx.x < 32 && dgamma) {
        atomicAdd(&dgamma[blockIdx.x * 32 + threadIdx.x], (float)tot[32 + threadIdx.x]);
        atomicAdd(&dbeta[blockIdx.x * 32 + threadIdx.x], (float)tot[threadIdx.x]);
    }
    const float invM = 1.f / (float)M;
    float mg[8], mgx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { mg[i] = (float)tot[cl * 8 + i] * invM; mgx[i] = (float)tot[32 + cl * 8 + i] * invM; }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + 64 * j;
        if (r < M) {
            Vec8<T_> o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.set(i, vsc.get(i) * (gq[j][i] - mg[i] - xh[j][i] * mgx[i]));
            o.store(da + (long)r * C + col0);
        }
    }
}

template <typename TA, typename TG, typename T_, int NR>
__global__ __launch_bounds__(256) void colbn_bwd_fused_kernel(const TA* __restrict__ a, const TG* __restrict__ dh,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              int relu, int M, int C, T_* __restrict__ da, float* dgamma,
                                                              float* dbeta) {
    colbn_bwd_fused_body<TA, TG, T_, NR>(a, dh, scale, shift, mean, rstd, relu, M, C, da, dgamma, dbeta);
}

struct BnBwdMulti {
    const void* a[MAXP]; const void* dh[MAXP]; const float* st[MAXP]; void* da[MAXP]; float* dgamma[MAXP]; float* dbeta[MAXP];
};
template <typename TA, typename TG, typename T_, int NR>
__global__ __launch_bounds__(256) void colbn_bwd_multi_kernel(BnBwdMulti m, int relu, int groups, int M, int C) {
    const int p = blockIdx.z;
    const long GC = (long)groups * C;
    const float* st = m.st[p];
    colbn_bwd_fused_body<TA, TG, T_, NR>(static_cast<const TA*>(m.a[p]), static_cast<const TG*>(m.dh[p]), st, st + GC, st + 2 * GC,
                                         st + 3 * GC, relu, M, C, static_cast<T_*>(m.da[p]), m.dgamma[p], m.dbeta[p]);
}

__global__ void add_d2f_kernel(const double* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += (float)src[i];
}

// --------------------------------------------------------------------------------------------- Barlow loss on c
// c [D][D] fp32 (already divided by the batch).  loss_out[0] += coef * sum (c - I)^2 ; dc = dscale * (c - I) as T_
template <typename T_>
__global__ __launch_bounds__(256) void barlow_loss_kernel(const float* __restrict__ c, int D, float coef, float dscale,
                                                          T_* __restrict__ dc, float* __restrict__ loss_out) {
    __shared__ float sh[16];
    const int total8 = D * (D / 8);                     // 8 consecutive columns of one row per thread and trip (D % 8 == 0)
    float acc = 0.f;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < total8; v += gridDim.x * 256) {
        const int i = v / (D / 8), j0 = (v - i * (D / 8)) * 8;
        const Vec8<float> x = Vec8<float>::load(c + (long)v * 8);
        Vec8<T_> o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d = x.get(k) - (i == j0 + k ? 1.f : 0.f);
            acc += d * d;
            o.set(k, dscale * d);
        }
        o.store(dc + (long)v * 8);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(loss_out, coef * acc);
}

struct BarlowMulti { const float* c[MAXP]; void* dc[MAXP]; float* loss[MAXP]; float coef[MAXP]; float dscale[MAXP]; };
template <typename T_>
__global__ __launch_bounds__(256) void barlow_loss_multi_kernel(BarlowMulti m, int D) {
    __shared__ float sh[16];
    const int p = blockIdx.y;
    const float* c = m.c[p];
    T_* dc = static_cast<T_*>(m.dc[p]);
    const float dscale = m.dscale[p];
    const int total8 = D * (D / 8);
    float acc = 0.f;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < total8; v += gridDim.x * 256) {
        const int i = v / (D / 8), j0 = (v - i * (D / 8)) * 8;
        const Vec8<float> x = Vec8<float>::load(c + (long)v * 8);
        Vec8<T_> o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d = x.get(k) - (i == j0 + k ? 1.f : 0.f);
            acc += d * d;
            o.set(k, dscale * d);
        }
        o.store(dc + (long)v * 8);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(m.loss[p], m.coef[p] * acc);
}

// --------------------------------------------------------------------------------------------- MoCo head
// one wave per row: qn = q / max(||q||, 1e-12)
template <typename T_>
__global__ __launch_bounds__(64) void l2norm_fwd_kernel(const float* __restrict__ q, int D, T_* __restrict__ qn,
                                                        float* __restrict__ qn32, float* __restrict__ inv_norm) {
    const long b = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) { const float v = q[b * D + d]; s += v * v; }
    s = wave_sum(s);
    const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int d = threadIdx.x; d < D; d += 64) {
        const float v = q[b * D + d] * inv;
        qn32[b * D + d] = v;
        qn[b * D + d] = from_f32<T_>(v);
    }
    if (threadIdx.x == 0) inv_norm[b] = inv;
}

// the MoCo head's prologue in one launch (`delores_m/upstream_expert.py:235-252`): both normalisations and the positive logit -
// the arithmetic of l2norm_fwd (twice) and rowdot, per row, in the same order
template <typename T_>
__global__ __launch_bounds__(64) void moco_prep_kernel(const float* __restrict__ q, const float* __restrict__ k, int D, float scale,
                                                       T_* __restrict__ qn, float* __restrict__ qn32, float* __restrict__ qinv,
                                                       T_* __restrict__ kn, float* __restrict__ kn32, float* __restrict__ kinv,
                                                       float* __restrict__ lpos) {
    const long b = blockIdx.x;
    float sq = 0.f, sk = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) {
        const float v = q[b * D + d], u = k[b * D + d];
        sq += v * v; sk += u * u;
    }
    sq = wave_sum(sq); sk = wave_sum(sk);
    const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) {
        const float v = q[b * D + d] * iq, u = k[b * D + d] * ik;
        qn32[b * D + d] = v; qn[b * D + d] = from_f32<T_>(v);
        kn32[b * D + d] = u; kn[b * D + d] = from_f32<T_>(u);
        s += v * u;
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) { qinv[b] = iq; kinv[b] = ik; lpos[b] = s * scale; }
}

// lpos[b] = <qn_b, kn_b> / temp
__global__ __launch_bounds__(64) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ bb, int D, float scale,
                                                    float* __restrict__ out) {
    const long b = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) s += a[b * D + d] * bb[b * D + d];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[b] = s * scale;
}

// logits row = [lpos, lneg[0..K)] (already divided by temp); CE with label 0, mean over the batch.
// ONE pass over the row (online softmax: running max + rescaled sum per thread, merged across the block).
__device__ __forceinline__ void lse_merge(float& m, float& s, float om, float os) {
    const float nm = fmaxf(m, om);
    s = s * __expf(m - nm) + os * __expf(om - nm);
    m = nm;
}
__global__ __launch_bounds__(256) void moco_ce_fwd_kernel(const float* __restrict__ lpos, const float* __restrict__ lneg, int K,
                                                          float inv_B, float* __restrict__ lse, float* __restrict__ loss_out) {
    __shared__ float shm[4], shs[4];
    const long b = blockIdx.x;
    const float* row = lneg + b * K;
    float m = -3.0e38f, s = 0.f;
    const int K4 = K & ~3;
    for (int k = threadIdx.x * 4; k < K4; k += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
        const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (vm > m) { s *= __expf(m - vm); m = vm; }
        s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
    }
    for (int k = K4 + threadIdx.x; k < K; k += 256) lse_merge(m, s, row[k], 1.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lse_merge(m, s, __shfl_xor(m, o, 64), __shfl_xor(s, o, 64));
    if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = m; shs[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) lse_merge(m, s, shm[w], shs[w]);
        lse_merge(m, s, lpos[b], 1.f);
        const float l = m + logf(s);
        lse[b] = l;
        atomicAdd(loss_out, (l - lpos[b]) * inv_B);
    }
}

// Merge of the per-slab (max, sum exp) partials written by the logits GEMM's epilogue (audiossl_moco_logits mode 1) with the
// positive logit: lse[b], loss += mean_b (lse_b - lpos_b), dlpos[b] = (softmax_pos - 1) * gscale.
__global__ __launch_bounds__(256) void moco_lse_merge_kernel(const float* __restrict__ lpos, const float* __restrict__ part,
                                                             int nslot, float inv_B, float gscale, float* __restrict__ lse,
                                                             float* __restrict__ loss_out, float* __restrict__ dlpos) {
    __shared__ float shm[4], shs[4];
    const long b = blockIdx.x;
    const float* row = part + b * nslot * 2;
    float m = -3.0e38f, s = 0.f;
    for (int k = threadIdx.x; k < nslot; k += 256) lse_merge(m, s, row[2 * k], row[2 * k + 1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lse_merge(m, s, __shfl_xor(m, o, 64), __shfl_xor(s, o, 64));
    if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = m; shs[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) lse_merge(m, s, shm[w], shs[w]);
        const float lp = lpos[b];
        lse_merge(m, s, lp, 1.f);
        const float l = m + logf(s);
        lse[b] = l;
        atomicAdd(loss_out, (l - lp) * inv_B);
        if (dlpos) dlpos[b] = (expf(lp - l) - 1.f) * gscale;
    }
}

// P[b][k] = softmax_neg * gscale (T_), dlpos[b] = (softmax_pos - 1) * gscale;  gscale = 1/(B*temp)
template <typename T_>
__global__ __launch_bounds__(256) void moco_ce_bwd_kernel(const float* __restrict__ lpos, const float* __restrict__ lneg,
                                                          const float* __restrict__ lse, int K, float gscale,
                                                          T_* __restrict__ P, float* __restrict__ dlpos) {
    // 8 consecutive columns per thread (two 16-byte loads, one 16-byte bf16 store) when the row length allows; the
    // one-element-per-thread form wrote 2 bytes per lane and ran at a third of the HBM rate
    const long b = blockIdx.y;
    const float l = lse[b];
    const long k0 = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if ((K & 7) == 0) {
        if (k0 < K) {
            const Vec8<float> v = Vec8<float>::load(lneg + b * K + k0);
            Vec8<T_> o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.set(i, expf(v.get(i) - l) * gscale);
            o.store(P + b * K + k0);
        }
    } else {
        for (long k = k0; k < k0 + 8 && k < K; ++k) P[b * K + k] = from_f32<T_>(expf(lneg[b * K + k] - l) * gscale);
    }
    if (k0 == 0) dlpos[b] = (expf(lpos[b] - l) - 1.f) * gscale;
}

// dq = (g - qn <qn, g>) * inv_norm with g = dqn + dlpos * kn
template <typename T_>
__global__ __launch_bounds__(64) void l2norm_bwd_kernel(const float* __restrict__ dqn, const float* __restrict__ dlpos,
                                                        const float* __restrict__ kn32, const float* __restrict__ qn32,
                                                        const float* __restrict__ inv_norm, int D, T_* __restrict__ dq) {
    const long b = blockIdx.x;
    const float dl = dlpos[b];
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) s += qn32[b * D + d] * (dqn[b * D + d] + dl * kn32[b * D + d]);
    s = wave_sum(s);
    const float inv = inv_norm[b];
    for (int d = threadIdx.x; d < D; d += 64) {
        const float g = dqn[b * D + d] + dl * kn32[b * D + d];
        dq[b * D + d] = from_f32<T_>((g - qn32[b * D + d] * s) * inv);
    }
}

// queue[:, ptr:ptr+B] = keys^T  (fp32 master [D][K] + T_ shadow)
template <typename T_>
__global__ __launch_bounds__(256) void enqueue_kernel(const float* __restrict__ keys, int B, int D, int K, int ptr,
                                                      const long long* __restrict__ ptr_dev, float* __restrict__ queue,
                                                      T_* __restrict__ shadow) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * D) return;
    if (ptr_dev) {                                      // write position kept on the device (hipGraph replay)
        const long long pd = ptr_dev[0];
        if (pd < 0 || pd + B > K) return;
        ptr = (int)pd;
    }
    const int d = idx / B, b = idx % B;
    const float v = keys[(long)b * D + d];
    queue[(long)d * K + ptr + b] = v;
    if (shadow) shadow[(long)d * K + ptr + b] = from_f32<T_>(v);
}

__global__ void advance_ptr_kernel(long long* ptr_dev, int B, int K) { ptr_dev[0] = (ptr_dev[0] + B) % K; }

// --------------------------------------------------------------------------------------------- flat-buffer plumbing
// torch.optim.SGD: g += wd*p; buf = first ? g : mom*buf + g; p -= lr*buf
// Optional tail work on the same pass (saves two full sweeps of the buffers at the start of the next step): `shadow` =
// bf16 copy of the updated parameters (the MFMA operands of the next step), `zero_grad` = clear g after it has been read.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                  long n, float lr, float mom, float wd, int first, float gscale_host,
                                                  const float* __restrict__ gscale_dev, bf16* __restrict__ shadow, int zero_grad) {
    const float gscale = gscale_dev ? gscale_host * gscale_dev[0] : gscale_host;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 bv = first ? f32x4{0, 0, 0, 0} : *reinterpret_cast<f32x4*>(buf + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = pv[k], bk = bv[k];
                sgd_step(pk, gv[k], bk, lr, mom, wd, gscale, first);
                pv[k] = pk; bv[k] = bk;
            }
            *reinterpret_cast<f32x4*>(buf + i) = bv;
            *reinterpret_cast<f32x4*>(p + i) = pv;
            if (shadow) *reinterpret_cast<bf16x4*>(shadow + i) = bf16x4{(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
            if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (long j = i; j < n; ++j) {
                float pj = p[j], b = first ? 0.f : buf[j];
                sgd_step(pj, g[j], b, lr, mom, wd, gscale, first);
                buf[j] = b;
                p[j] = pj;
                if (shadow) shadow[j] = (bf16)pj;
                if (zero_grad) g[j] = 0.f;
            }
        }
    }
}

// The same over a table of segments of the flat buffers (blockIdx.y = segment; offsets and lengths are multiples of 4 elements:
// FlatGroup aligns every tensor to 64): the tensors of a slice whose other tensors were already stepped by the weight-gradient
// GEMMs themselves (audiossl_gemm_multi_sgd)
__global__ __launch_bounds__(256) void sgd_segments_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                           const long* __restrict__ segs, float lr, float mom, float wd, int first,
                                                           float gscale_host, const float* __restrict__ gscale_dev,
                                                           bf16* __restrict__ shadow, int zero_grad) {
    const float gscale = gscale_dev ? gscale_host * gscale_dev[0] : gscale_host;
    const long off = segs[2 * blockIdx.y], n = segs[2 * blockIdx.y + 1];
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        const long e = off + i;
        if (i + 4 <= n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + e);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + e);
            f32x4 bv = first ? f32x4{0, 0, 0, 0} : *reinterpret_cast<f32x4*>(buf + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = pv[k], bk = bv[k];
                sgd_step(pk, gv[k], bk, lr, mom, wd, gscale, first);
                pv[k] = pk; bv[k] = bk;
            }
            *reinterpret_cast<f32x4*>(buf + e) = bv;
            *reinterpret_cast<f32x4*>(p + e) = pv;
            if (shadow) *reinterpret_cast<bf16x4*>(shadow + e) = bf16x4{(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
            if (zero_grad) *reinterpret_cast<f32x4*>(g + e) = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (long j = e; j < off + n; ++j) {
                float pj = p[j], b = first ? 0.f : buf[j];
                sgd_step(pj, g[j], b, lr, mom, wd, gscale, first);
                buf[j] = b;
                p[j] = pj;
                if (shadow) shadow[j] = (bf16)pj;
                if (zero_grad) g[j] = 0.f;
            }
        }
    }
}

// g[off_s .. off_s + n_s) = 0 for a table of segments (blockIdx.y = segment): the small tensors of a flat gradient whose big ones are
// overwritten (not accumulated into) by their only writer and therefore need no clearing
__global__ __launch_bounds__(256) void zero_segments_kernel(float* __restrict__ g, const long* __restrict__ segs) {
    const long off = segs[2 * blockIdx.y], n = segs[2 * blockIdx.y + 1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[off + i] = 0.f;
}

// pk = pk*m + pq*(1-m)
// shadow (optional): bf16 copy of the updated pk, written in the same pass (the key encoder's MFMA operands)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pk, const float* __restrict__ pq, long n, float m,
                                                  bf16* __restrict__ shadow) {
    const float om = 1.f - m;
    const bool vec = ((reinterpret_cast<size_t>(pk) | reinterpret_cast<size_t>(pq)) & 15) == 0 && (reinterpret_cast<size_t>(shadow) & 7) == 0;
    const long n4 = vec ? n / 4 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 a = *reinterpret_cast<const f32x4*>(pk + i * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pq + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = a[k] * m + b[k] * om;
        *reinterpret_cast<f32x4*>(pk + i * 4) = a;
        if (shadow) *reinterpret_cast<bf16x4*>(shadow + i * 4) = bf16x4{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3]};
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = pk[i] * m + pq[i] * om;
        pk[i] = v;
        if (shadow) shadow[i] = (bf16)v;
    }
}

// 8 elements per thread and trip (16-byte stores of bf16); `vec` = both pointers 16-byte aligned
template <typename T_>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T_* __restrict__ dst, long n, int vec) {
    const long n8 = vec ? n / 8 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const Vec8<float> v = Vec8<float>::load(src + i * 8);
        Vec8<T_> o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.set(k, v.get(k));
        o.store(dst + i * 8);
    }
    for (long i = n8 * 8 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = from_f32<T_>(src[i]);
}

template <typename T_>
__global__ __launch_bounds__(256) void cast_back_kernel(const T_* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = to_f32(src[i]);
}

// counter-based keep mask: keep iff hash(seed, index) >= p * 2^32  (splitmix64 finaliser)
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ keep, long n, unsigned long long seed, float p,
                                                           const long long* __restrict__ counter) {
    seed = dropout_seed(seed, counter);
    const unsigned int thr = dropout_threshold(p);
    auto bit = [&](long i) -> unsigned int { return dropout_keep(seed, i, thr); };
    // 16 mask bytes per thread and trip, one 16-byte store (one byte per lane: 64-byte wave stores, 5x slower)
    const long n16 = ((reinterpret_cast<size_t>(keep) & 15) == 0) ? n / 16 : 0;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n16; v += (long)gridDim.x * 256) {
        unsigned int w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long i = v * 16 + q * 4;
            w[q] = bit(i) | (bit(i + 1) << 8) | (bit(i + 2) << 16) | (bit(i + 3) << 24);
        }
        *reinterpret_cast<uint4*>(keep + v * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (long i = n16 * 16 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) keep[i] = (uint8_t)bit(i);
}

}  // namespace

int g_assl_prezeroed = 0;
extern "C" int audiossl_set_prezeroed(int on) { g_assl_prezeroed = on ? 1 : 0; return ASSL_OK; }
const char* g_assl_last_kernel = "";
extern "C" int audiossl_last_kernel(char* name, int capacity) {
    ASSL_REQUIRE(name && capacity > 0);
    int i = 0;
    for (; g_assl_last_kernel[i] && i < capacity - 1; ++i) name[i] = g_assl_last_kernel[i];
    name[i] = 0;
    return ASSL_OK;
}

#define S_(stream) static_cast<hipStream_t>(stream)
#define GRID1(n) dim3((unsigned)(((long)(n) + 255) / 256))

extern "C" int audiossl_maxmean_fwd(int dtype, int out_f32, const void* H, void* y, uint8_t* arg, int N, int Tt, int D, void* stream) {
    ASSL_REQUIRE(H && y && arg && N > 0 && Tt > 0 && Tt < 256 && D > 0 && (D % 8) == 0 && (dtype == 0 || dtype == 1));
    const long total = (long)N * D / 8;
    if (dtype == 0) hipLaunchKernelGGL((maxmean_fwd_kernel<float, float>), GRID1(total), dim3(256), 0, S_(stream), (const float*)H, (float*)y, arg, N, Tt, D);
    else if (out_f32) hipLaunchKernelGGL((maxmean_fwd_kernel<bf16, float>), GRID1(total), dim3(256), 0, S_(stream), (const bf16*)H, (float*)y, arg, N, Tt, D);
    else            hipLaunchKernelGGL((maxmean_fwd_kernel<bf16, bf16>), GRID1(total), dim3(256), 0, S_(stream), (const bf16*)H, (bf16*)y, arg, N, Tt, D);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_maxmean_bwd(int dtype, int gdtype, const void* dy, const uint8_t* arg, const void* H, void* dA, int N, int Tt,
                                    int D, void* stream) {
    ASSL_REQUIRE(dy && arg && H && dA && N > 0 && Tt > 0 && D > 0 && (D % 8) == 0 && (dtype == 0 || dtype == 1));
    ASSL_REQUIRE(gdtype == 0 || gdtype == dtype);
    const long total = (long)N * Tt * D / 8;
    if (dtype == 0) hipLaunchKernelGGL((maxmean_bwd_kernel<float, float>), GRID1(total), dim3(256), 0, S_(stream), (const float*)dy, arg, (const float*)H, (float*)dA, N, Tt, D);
    else if (gdtype == 0) hipLaunchKernelGGL((maxmean_bwd_kernel<bf16, float>), GRID1(total), dim3(256), 0, S_(stream), (const float*)dy, arg, (const bf16*)H, (bf16*)dA, N, Tt, D);
    else            hipLaunchKernelGGL((maxmean_bwd_kernel<bf16, bf16>), GRID1(total), dim3(256), 0, S_(stream), (const bf16*)dy, arg, (const bf16*)H, (bf16*)dA, N, Tt, D);
    ASSL_LAUNCH_CHECK();
}

// adtype: storage type of the BatchNorm input `a` (0 = fp32 even when the output dtype is bf16: pre-normalisation values
// can have |mean| >> std, which bf16's 8 significand bits cannot carry); dtype: type of the output (an MFMA operand).
extern "C" int audiossl_colbn_fwd(int dtype, int adtype, const void* a, const float* scale, const float* shift, int relu, void* h,
                                  int groups, long M, int C, void* stream) {
    ASSL_REQUIRE(a && scale && shift && h && groups > 0 && M > 0 && C > 0 && (C % 8) == 0 && (dtype == 0 || dtype == 1));
    ASSL_REQUIRE(adtype == 0 || adtype == dtype);
    const long total = groups * M * C / 8;
    if (dtype == 0) hipLaunchKernelGGL((colbn_fwd_kernel<float, float>), GRID1(total), dim3(256), 0, S_(stream), (const float*)a, scale, shift, relu, (float*)h, M, C, groups);
    else if (adtype == 0) hipLaunchKernelGGL((colbn_fwd_kernel<float, bf16>), GRID1(total), dim3(256), 0, S_(stream), (const float*)a, scale, shift, relu, (bf16*)h, M, C, groups);
    else hipLaunchKernelGGL((colbn_fwd_kernel<bf16, bf16>), GRID1(total), dim3(256), 0, S_(stream), (const bf16*)a, scale, shift, relu, (bf16*)h, M, C, groups);
    ASSL_LAUNCH_CHECK();
}

// Statistics + finalisation (running statistics included) + normalisation of `colstats` -> `bn_finalize` -> `colbn_fwd`
// in one launch; M <= 1024 rows per group, C % 32 == 0.  Returns ASSL_EINVAL for shapes outside that (callers then use
// the three separate calls).
extern "C" int audiossl_colbn_train_fwd(int dtype, int adtype, const void* a, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, float momentum, float eps, int relu,
                                        int groups, long M, int C, void* h, float* scale, float* shift, float* save_mean,
                                        float* save_rstd, void* stream) {
    ASSL_REQUIRE(a && h && scale && shift && save_mean && save_rstd && groups > 0 && M > 0 && M <= 1024 && C > 0 && (C % 32) == 0);
    ASSL_REQUIRE((dtype == 0 || dtype == 1) && (adtype == 0 || adtype == dtype));
    ASSL_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    hipStream_t s = S_(stream);
#define TF(TA, TO) do {                                                                                                        \
    if (M <= 512) hipLaunchKernelGGL((colbn_train_fwd_kernel<TA, TO, 8>), dim3(C / 32), dim3(256), 0, s, (const TA*)a, gamma, beta, \
                                     running_mean, running_var, momentum, eps, relu, groups, (int)M, C, (TO*)h, scale, shift,    \
                                     save_mean, save_rstd);                                                                     \
    else hipLaunchKernelGGL((colbn_train_fwd_kernel<TA, TO, 16>), dim3(C / 32), dim3(256), 0, s, (const TA*)a, gamma, beta,      \
                            running_mean, running_var, momentum, eps, relu, groups, (int)M, C, (TO*)h, scale, shift, save_mean,  \
                            save_rstd); } while (0)
    if (dtype == 0) TF(float, float);
    else if (adtype == 0) TF(float, bf16);
    else TF(bf16, bf16);
#undef TF
    ASSL_LAUNCH_CHECK();
}

// tmp: 2*G*C doubles of scratch.  dgamma/dbeta may be null (affine=False); otherwise accumulated (+=).
// adtype / gdtype: storage types of `a` and of the incoming gradient `dh` (0 = fp32); `da` is written in `dtype`.
extern "C" int audiossl_colbn_bwd(int dtype, int adtype, int gdtype, const void* a, const void* dh, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, int relu, int groups, long M, int C,
                                  double* tmp, void* da, float* dgamma, float* dbeta, void* stream) {
    ASSL_REQUIRE(a && dh && scale && shift && mean && rstd && tmp && da && groups > 0 && M > 0 && C > 0 && (C % 8) == 0);
    ASSL_REQUIRE((dtype == 0 || dtype == 1) && (C % 64) == 0 && (gdtype == 0 || gdtype == dtype) && (adtype == 0 || adtype == dtype));
    hipStream_t s = S_(stream);
    static const bool fused_ok = getenv("AUDIOSSL_BN_FUSED") ? atoi(getenv("AUDIOSSL_BN_FUSED")) != 0 : true;
    if (fused_ok && M <= 1024 && (C % 32) == 0) {             // short batches: statistics + apply in one launch
        dim3 fgrid(C / 32, groups);
#define CF(TA, TG, TO) do {                                                                                                    \
    if (M <= 512) hipLaunchKernelGGL((colbn_bwd_fused_kernel<TA, TG, TO, 8>), fgrid, dim3(256), 0, s, (const TA*)a, (const TG*)dh, \
                                     scale, shift, mean, rstd, relu, (int)M, C, (TO*)da, dgamma, dbeta);                        \
    else hipLaunchKernelGGL((colbn_bwd_fused_kernel<TA, TG, TO, 16>), fgrid, dim3(256), 0, s, (const TA*)a, (const TG*)dh,      \
                            scale, shift, mean, rstd, relu, (int)M, C, (TO*)da, dgamma, dbeta); } while (0)
        if (dtype == 0) CF(float, float, float);
        else if (adtype == 0 && gdtype == 0) CF(float, float, bf16);
        else if (adtype == 0) CF(float, bf16, bf16);
        else if (gdtype == 0) CF(bf16, float, bf16);
        else CF(bf16, bf16, bf16);
#undef CF
        ASSL_LAUNCH_CHECK();
    }
    ASSL_ZERO(tmp, sizeof(double) * 2 * C * groups, s);
    const int rpb = M >= 4096 ? 256 : 64;
    dim3 grid(ceil_div(M, rpb), C / 64, groups);
    const long total = groups * M * C / 8;
    const long GC = (long)groups * C;
#define CB(TA, TG, TO) do {                                                                                                           \
    hipLaunchKernelGGL((colbn_bwd_stats_kernel<TA, TG>), grid, dim3(256), 0, s, (const TA*)a, (const TG*)dh, scale, shift, mean, rstd,  \
                       relu, M, C, rpb, tmp, tmp + GC);                                                                               \
    hipLaunchKernelGGL((colbn_bwd_apply_kernel<TA, TG, TO>), GRID1(total), dim3(256), 0, s, (const TA*)a, (const TG*)dh, scale, shift,  \
                       mean, rstd, relu, M, C, groups, tmp, tmp + GC, (TO*)da, dgamma, dbeta, 1.f / (float)M); } while (0)
    if (dtype == 0) CB(float, float, float);
    else if (adtype == 0 && gdtype == 0) CB(float, float, bf16);
    else if (adtype == 0) CB(float, bf16, bf16);
    else if (gdtype == 0) CB(bf16, float, bf16);
    else CB(bf16, bf16, bf16);
#undef CB
    ASSL_LAUNCH_CHECK();
}

// The same backward in two halves for SyncBatchNorm (statistics exchanged between ranks in between): colbn_bwd_stats leaves this
// rank's sum g / sum g xhat in tmp [2][G*C] (fp64); colbn_bwd_apply computes da from (all-reduced) sums over `count` rows and takes
// no parameter gradients (the caller adds its own rank's sums to dgamma / dbeta with add_d2f).
extern "C" int audiossl_colbn_bwd_stats(int dtype, int adtype, int gdtype, const void* a, const void* dh, const float* scale,
                                        const float* shift, const float* mean, const float* rstd, int relu, int groups, long M, int C,
                                        double* tmp, void* stream) {
    ASSL_REQUIRE(a && dh && scale && shift && mean && rstd && tmp && groups > 0 && M > 0 && C > 0 && (C % 64) == 0);
    ASSL_REQUIRE((dtype == 0 || dtype == 1) && (gdtype == 0 || gdtype == dtype) && (adtype == 0 || adtype == dtype));
    hipStream_t s = S_(stream);
    ASSL_ZERO(tmp, sizeof(double) * 2 * C * groups, s);
    const int rpb = M >= 4096 ? 256 : 64;
    dim3 grid(ceil_div(M, rpb), C / 64, groups);
    const long GC = (long)groups * C;
#define CS(TA, TG) hipLaunchKernelGGL((colbn_bwd_stats_kernel<TA, TG>), grid, dim3(256), 0, s, (const TA*)a, (const TG*)dh, scale, shift, \
                                      mean, rstd, relu, M, C, rpb, tmp, tmp + GC)
    if (dtype == 0 || (adtype == 0 && gdtype == 0)) CS(float, float);
    else if (adtype == 0) CS(float, bf16);
    else if (gdtype == 0) CS(bf16, float);
    else CS(bf16, bf16);
#undef CS
    ASSL_LAUNCH_CHECK();
}
extern "C" int audiossl_colbn_bwd_apply(int dtype, int adtype, int gdtype, const void* a, const void* dh, const float* scale,
                                        const float* shift, const float* mean, const float* rstd, int relu, int groups, long M, int C,
                                        const double* sums, double count, void* da, void* stream) {
    ASSL_REQUIRE(a && dh && scale && shift && mean && rstd && sums && da && groups > 0 && M > 0 && C > 0 && (C % 64) == 0 && count > 1.0);
    ASSL_REQUIRE((dtype == 0 || dtype == 1) && (gdtype == 0 || gdtype == dtype) && (adtype == 0 || adtype == dtype));
    hipStream_t s = S_(stream);
    const long total = groups * M * C / 8;
    const long GC = (long)groups * C;
    const float inv = (float)(1.0 / count);
#define CA(TA, TG, TO) hipLaunchKernelGGL((colbn_bwd_apply_kernel<TA, TG, TO>), GRID1(total), dim3(256), 0, s, (const TA*)a, (const TG*)dh, \
                                          scale, shift, mean, rstd, relu, M, C, groups, sums, sums + GC, (TO*)da, (float*)nullptr,          \
                                          (float*)nullptr, inv)
    if (dtype == 0) CA(float, float, float);
    else if (adtype == 0 && gdtype == 0) CA(float, float, bf16);
    else if (adtype == 0) CA(float, bf16, bf16);
    else if (gdtype == 0) CA(bf16, float, bf16);
    else CA(bf16, bf16, bf16);
#undef CA
    ASSL_LAUNCH_CHECK();
}

// ---- multi-problem forms for the three Barlow heads (bf16 outputs): `count` layers of one shape per launch.
// stats[p]: [4][groups*C] fp32 = scale, shift, mean, rstd of problem p (written by the forward, read by the backward).
extern "C" int audiossl_colbn_train_fwd_multi(int count, int adtype, const void* const* a, const float* const* gamma,
                                              const float* const* beta, float* const* running_mean, float* const* running_var,
                                              float momentum, float eps, int relu, int groups, long M, int C, void* const* h,
                                              float* const* stats, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAXP && a && h && stats && groups > 0 && M > 0 && M <= 1024 && C > 0 && (C % 32) == 0);
    ASSL_REQUIRE(adtype == 0 || adtype == 1);
    BnFwdMulti m{};
    for (int p = 0; p < count; ++p) {
        ASSL_REQUIRE(a[p] && h[p] && stats[p]);
        m.a[p] = a[p]; m.h[p] = h[p]; m.st[p] = stats[p];
        m.gamma[p] = gamma ? gamma[p] : nullptr; m.beta[p] = beta ? beta[p] : nullptr;
        m.rm[p] = running_mean ? running_mean[p] : nullptr; m.rv[p] = running_var ? running_var[p] : nullptr;
        ASSL_REQUIRE((m.rm[p] == nullptr) == (m.rv[p] == nullptr));
    }
    hipStream_t s = S_(stream);
    dim3 grid(C / 32, count);
#define TFM(TA, NR_) hipLaunchKernelGGL((colbn_train_fwd_multi_kernel<TA, bf16, NR_>), grid, dim3(256), 0, s, m, momentum, eps, relu, \
                                        groups, (int)M, C)
    if (adtype == 0) { if (M <= 512) TFM(float, 8); else TFM(float, 16); }
    else             { if (M <= 512) TFM(bf16, 8); else TFM(bf16, 16); }
#undef TFM
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_colbn_bwd_multi(int count, int adtype, int gdtype, const void* const* a, const void* const* dh,
                                        const float* const* stats, int relu, int groups, long M, int C, void* const* da,
                                        float* const* dgamma, float* const* dbeta, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAXP && a && dh && stats && da && groups > 0 && M > 0 && M <= 1024 && C > 0 && (C % 32) == 0);
    ASSL_REQUIRE((adtype == 0 || adtype == 1) && (gdtype == 0 || gdtype == 1));
    BnBwdMulti m{};
    for (int p = 0; p < count; ++p) {
        ASSL_REQUIRE(a[p] && dh[p] && stats[p] && da[p]);
        m.a[p] = a[p]; m.dh[p] = dh[p]; m.st[p] = stats[p]; m.da[p] = da[p];
        m.dgamma[p] = dgamma ? dgamma[p] : nullptr; m.dbeta[p] = dbeta ? dbeta[p] : nullptr;
        ASSL_REQUIRE((m.dgamma[p] == nullptr) == (m.dbeta[p] == nullptr));
    }
    hipStream_t s = S_(stream);
    dim3 grid(C / 32, groups, count);
#define BWM(TA, TG, NR_) hipLaunchKernelGGL((colbn_bwd_multi_kernel<TA, TG, bf16, NR_>), grid, dim3(256), 0, s, m, relu, groups, (int)M, C)
#define BWM2(TA, TG) do { if (M <= 512) BWM(TA, TG, 8); else BWM(TA, TG, 16); } while (0)
    if (adtype == 0 && gdtype == 0) BWM2(float, float);
    else if (adtype == 0) BWM2(float, bf16);
    else if (gdtype == 0) BWM2(bf16, float);
    else BWM2(bf16, bf16);
#undef BWM2
#undef BWM
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_barlow_loss_multi(int count, const float* const* c, int D, const float* coef, const float* dscale,
                                          void* const* dc, float* const* loss_out, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAXP && c && coef && dscale && dc && loss_out && D > 0 && (D % 8) == 0 && D <= 16384);
    BarlowMulti m{};
    for (int p = 0; p < count; ++p) {
        ASSL_REQUIRE(c[p] && dc[p] && loss_out[p]);
        if (!ASSL_ALIGNED16(c[p]) || !ASSL_ALIGNED16(dc[p])) return ASSL_EALIGN;
        m.c[p] = c[p]; m.dc[p] = dc[p]; m.loss[p] = loss_out[p]; m.coef[p] = coef[p]; m.dscale[p] = dscale[p];
    }
    const int grid = min(512, ceil_div((long)D * (D / 8), 256));
    hipLaunchKernelGGL(barlow_loss_multi_kernel<bf16>, dim3(grid, count), dim3(256), 0, S_(stream), m, D);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_add_d2f(const double* src, float* dst, int n, void* stream) {
    ASSL_REQUIRE(src && dst && n > 0);
    hipLaunchKernelGGL(add_d2f_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, S_(stream), src, dst, n);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_barlow_loss(int dtype, const float* c, int D, float coef, float dscale, void* dc, float* loss_out,
                                    void* stream) {
    ASSL_REQUIRE(c && dc && loss_out && D > 0 && (D % 8) == 0 && D <= 16384 && (dtype == 0 || dtype == 1));
    if (!ASSL_ALIGNED16(c) || !ASSL_ALIGNED16(dc)) return ASSL_EALIGN;
    const int grid = min(512, ceil_div((long)D * (D / 8), 256));
    if (dtype == 0) hipLaunchKernelGGL(barlow_loss_kernel<float>, dim3(grid), dim3(256), 0, S_(stream), c, D, coef, dscale, (float*)dc, loss_out);
    else            hipLaunchKernelGGL(barlow_loss_kernel<bf16>, dim3(grid), dim3(256), 0, S_(stream), c, D, coef, dscale, (bf16*)dc, loss_out);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_l2norm_fwd(int dtype, const float* q, int B, int D, void* qn, float* qn32, float* inv_norm, void* stream) {
    ASSL_REQUIRE(q && qn && qn32 && inv_norm && B > 0 && D > 0 && (dtype == 0 || dtype == 1));
    if (dtype == 0) hipLaunchKernelGGL(l2norm_fwd_kernel<float>, dim3(B), dim3(64), 0, S_(stream), q, D, (float*)qn, qn32, inv_norm);
    else            hipLaunchKernelGGL(l2norm_fwd_kernel<bf16>, dim3(B), dim3(64), 0, S_(stream), q, D, (bf16*)qn, qn32, inv_norm);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_moco_prep(int dtype, const float* q, const float* k, int B, int D, float scale, void* qn, float* qn32, float* qinv,
                                  void* kn, float* kn32, float* kinv, float* lpos, void* stream) {
    ASSL_REQUIRE(q && k && qn && qn32 && qinv && kn && kn32 && kinv && lpos && B > 0 && D > 0 && (dtype == 0 || dtype == 1));
    if (dtype == 0) hipLaunchKernelGGL(moco_prep_kernel<float>, dim3(B), dim3(64), 0, S_(stream), q, k, D, scale, (float*)qn, qn32, qinv,
                                       (float*)kn, kn32, kinv, lpos);
    else            hipLaunchKernelGGL(moco_prep_kernel<bf16>, dim3(B), dim3(64), 0, S_(stream), q, k, D, scale, (bf16*)qn, qn32, qinv,
                                       (bf16*)kn, kn32, kinv, lpos);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_rowdot(const float* a, const float* b, int B, int D, float scale, float* out, void* stream) {
    ASSL_REQUIRE(a && b && out && B > 0 && D > 0);
    hipLaunchKernelGGL(rowdot_kernel, dim3(B), dim3(64), 0, S_(stream), a, b, D, scale, out);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_moco_ce_fwd(const float* lpos, const float* lneg, int B, int K, float* lse, float* loss_out, void* stream) {
    ASSL_REQUIRE(lpos && lneg && lse && loss_out && B > 0 && K > 0);
    hipLaunchKernelGGL(moco_ce_fwd_kernel, dim3(B), dim3(256), 0, S_(stream), lpos, lneg, K, 1.f / (float)B, lse, loss_out);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_moco_lse_merge(const float* lpos, const float* part, int B, int nslot, float gscale, float* lse,
                                       float* loss_out, float* dlpos, void* stream) {
    ASSL_REQUIRE(lpos && part && lse && loss_out && B > 0 && nslot > 0);
    hipLaunchKernelGGL(moco_lse_merge_kernel, dim3(B), dim3(256), 0, S_(stream), lpos, part, nslot, 1.f / (float)B, gscale, lse,
                       loss_out, dlpos);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_moco_ce_bwd(int dtype, const float* lpos, const float* lneg, const float* lse, int B, int K, float gscale,
                                    void* P, float* dlpos, void* stream) {
    ASSL_REQUIRE(lpos && lneg && lse && P && dlpos && B > 0 && K > 0 && (dtype == 0 || dtype == 1));
    dim3 grid(ceil_div(K, 256 * 8), B);
    if (dtype == 0) hipLaunchKernelGGL(moco_ce_bwd_kernel<float>, grid, dim3(256), 0, S_(stream), lpos, lneg, lse, K, gscale, (float*)P, dlpos);
    else            hipLaunchKernelGGL(moco_ce_bwd_kernel<bf16>, grid, dim3(256), 0, S_(stream), lpos, lneg, lse, K, gscale, (bf16*)P, dlpos);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_l2norm_bwd(int dtype, const float* dqn, const float* dlpos, const float* kn32, const float* qn32,
                                   const float* inv_norm, int B, int D, void* dq, void* stream) {
    ASSL_REQUIRE(dqn && dlpos && kn32 && qn32 && inv_norm && dq && B > 0 && D > 0 && (dtype == 0 || dtype == 1));
    if (dtype == 0) hipLaunchKernelGGL(l2norm_bwd_kernel<float>, dim3(B), dim3(64), 0, S_(stream), dqn, dlpos, kn32, qn32, inv_norm, D, (float*)dq);
    else            hipLaunchKernelGGL(l2norm_bwd_kernel<bf16>, dim3(B), dim3(64), 0, S_(stream), dqn, dlpos, kn32, qn32, inv_norm, D, (bf16*)dq);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_enqueue(int dtype, const float* keys, int B, int D, int K, int ptr, long long* ptr_dev, float* queue,
                                void* shadow, void* stream) {
    ASSL_REQUIRE(keys && queue && B > 0 && D > 0 && K > 0 && B <= K && (dtype == 0 || dtype == 1));
    ASSL_REQUIRE(ptr_dev || (ptr >= 0 && ptr + B <= K));
    const int grid = ceil_div((long)B * D, 256);
    if (dtype == 0) hipLaunchKernelGGL(enqueue_kernel<float>, dim3(grid), dim3(256), 0, S_(stream), keys, B, D, K, ptr, ptr_dev, queue, (float*)shadow);
    else            hipLaunchKernelGGL(enqueue_kernel<bf16>, dim3(grid), dim3(256), 0, S_(stream), keys, B, D, K, ptr, ptr_dev, queue, (bf16*)shadow);
    if (ptr_dev) hipLaunchKernelGGL(advance_ptr_kernel, dim3(1), dim3(1), 0, S_(stream), ptr_dev, B, K);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_sgd_momentum(float* p, float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                                     int first, float grad_scale, const float* grad_scale_dev, void* shadow_bf16, int zero_grad,
                                     void* stream) {
    ASSL_REQUIRE(p && g && buf && n > 0);
    if (!ASSL_ALIGNED16(p) || !ASSL_ALIGNED16(g) || !ASSL_ALIGNED16(buf)) return ASSL_EALIGN;
    if (shadow_bf16 && (reinterpret_cast<size_t>(shadow_bf16) & 7)) return ASSL_EALIGN;
    const int grid = (int)min((long)2048, (n + 1023) / 1024);
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, S_(stream), p, g, buf, n, lr, momentum, weight_decay, first, grad_scale,
                       grad_scale_dev, static_cast<bf16*>(shadow_bf16), zero_grad);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_sgd_momentum_segments(float* p, float* g, float* buf, const long* segs, int nseg, long max_n, float lr,
                                              float momentum, float weight_decay, int first, float grad_scale,
                                              const float* grad_scale_dev, void* shadow_bf16, int zero_grad, void* stream) {
    ASSL_REQUIRE(p && g && buf && segs && nseg > 0 && nseg <= 65535 && max_n > 0);
    if (!ASSL_ALIGNED16(p) || !ASSL_ALIGNED16(g) || !ASSL_ALIGNED16(buf)) return ASSL_EALIGN;
    if (shadow_bf16 && (reinterpret_cast<size_t>(shadow_bf16) & 7)) return ASSL_EALIGN;
    const int gx = (int)min((long)256, (max_n + 1023) / 1024);
    hipLaunchKernelGGL(sgd_segments_kernel, dim3(gx, nseg), dim3(256), 0, S_(stream), p, g, buf, segs, lr, momentum, weight_decay, first,
                       grad_scale, grad_scale_dev, static_cast<bf16*>(shadow_bf16), zero_grad);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_zero_segments(float* g, const long* segs, int nseg, long max_n, void* stream) {
    ASSL_REQUIRE(g && segs && nseg > 0 && nseg <= 65535 && max_n > 0);
    const int gx = (int)min((long)64, (max_n + 255) / 256);
    hipLaunchKernelGGL(zero_segments_kernel, dim3(gx, nseg), dim3(256), 0, S_(stream), g, segs);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_ema_update(float* pk, const float* pq, long n, float m, void* shadow, void* stream) {
    ASSL_REQUIRE(pk && pq && n > 0);
    const int grid = (int)min((long)2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(256), 0, S_(stream), pk, pq, n, m, static_cast<bf16*>(shadow));
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_cast(int dtype, const float* src, void* dst, long n, void* stream) {
    ASSL_REQUIRE(src && dst && n > 0 && (dtype == 0 || dtype == 1));
    const int grid = (int)min((long)4096, (n / 8 + 255) / 256 + 1);
    const int vec = ASSL_ALIGNED16(src) && ASSL_ALIGNED16(dst);
    if (dtype == 0) hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, S_(stream), src, (float*)dst, n, vec);
    else            hipLaunchKernelGGL(cast_kernel<bf16>, dim3(grid), dim3(256), 0, S_(stream), src, (bf16*)dst, n, vec);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_cast_back(int dtype, const void* src, float* dst, long n, void* stream) {
    ASSL_REQUIRE(src && dst && n > 0 && (dtype == 0 || dtype == 1));
    const int grid = (int)min((long)4096, (n + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(cast_back_kernel<float>, dim3(grid), dim3(256), 0, S_(stream), (const float*)src, dst, n);
    else            hipLaunchKernelGGL(cast_back_kernel<bf16>, dim3(grid), dim3(256), 0, S_(stream), (const bf16*)src, dst, n);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_dropout_mask(uint8_t* keep, long n, unsigned long long seed, float p, const long long* counter, void* stream) {
    ASSL_REQUIRE(keep && n > 0 && p >= 0.f && p < 1.f);
    const int grid = (int)min((long)4096, (n / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, S_(stream), keep, n, seed, p, counter);
    ASSL_LAUNCH_CHECK();
}

// ---- small extras -----------------------------------------------------------------------------------------------
namespace {
template <typename T_>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T_* __restrict__ g, const T_* __restrict__ h, T_* __restrict__ out, long n8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    const Vec8<T_> vg = Vec8<T_>::load(g + idx * 8), vh = Vec8<T_>::load(h + idx * 8);
    Vec8<T_> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.set(i, vh.get(i) > 0.f ? vg.get(i) : 0.f);
    o.store(out + idx * 8);
}
// eval-mode BatchNorm as an affine map: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                      float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float s = g / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = b - rm[c] * s;
}
// x (fp32) = hi + lo with hi = bf16(x), lo = bf16(x - hi): two MFMA operands that together carry ~16 significand bits
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ hi, bf16* __restrict__ lo, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        const bf16 h = (bf16)v;
        hi[i] = h;
        lo[i] = (bf16)(v - (float)h);
    }
}
// Centred cast of the time-pooled features that feed a projector (Linear without bias -> train-mode BatchNorm):
// yc[g][r][c] = bf16(y[g][r][c] - mean_r y[g][.][c]), cmean[g][c] = that mean.  Post-ReLU features averaged over time have
// |column mean| >> batch standard deviation, so a plain bf16 rounding costs 2^-9 * |mean| / std of the batch variation the
// BatchNorm keeps (10-20 % on the first projector layer).  The column shift itself is invisible to the
// layer: (y - 1 c^T) W^T = y W^T - 1 (W c)^T moves every output column by a constant, which train-mode BatchNorm removes;
// only its running mean sees it (shift_running_mean below), and the weight gradient da1^T y = da1^T yc because a BatchNorm
// input gradient sums to zero over the batch.  One workgroup = 32 columns of one group, rows in registers (M <= 64 * NR).
template <int NR>
__device__ __forceinline__ void center_cast_body(const float* __restrict__ y, bf16* __restrict__ yc, float* __restrict__ cmean, int M, int C);
template <int NR>
__global__ __launch_bounds__(256) void center_cast_kernel(const float* __restrict__ y, bf16* __restrict__ yc,
                                                          float* __restrict__ cmean, int M, int C) {
    center_cast_body<NR>(y, yc, cmean, M, C);
}
// the three Barlow heads of DeLoRes-M in one launch (blockIdx.z = head; their widths differ)
struct CenterMulti { const float* y[MAXP]; bf16* yc[MAXP]; float* cmean[MAXP]; int C[MAXP]; };
template <int NR>
__global__ __launch_bounds__(256) void center_cast_multi_kernel(CenterMulti m, int M) {
    const int p = blockIdx.z;
    if ((int)blockIdx.x * 32 >= m.C[p]) return;
    center_cast_body<NR>(m.y[p], m.yc[p], m.cmean[p], M, m.C[p]);
}
template <int NR>
__device__ __forceinline__ void center_cast_body(const float* __restrict__ y, bf16* __restrict__ yc, float* __restrict__ cmean, int M, int C) {
    __shared__ float red[64][65];
    __shared__ double tot[32];
    __shared__ float mu_s[32];
    const int cl = threadIdx.x & 3, r0 = threadIdx.x >> 2, g = blockIdx.y;
    const int col0 = blockIdx.x * 32 + cl * 8;
    const float* yg = y + (long)g * M * C;
    Vec8<float> v[NR];
    float part[1][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) part[0][i] = 0.f;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + 64 * j;
        if (r < M) {
            v[j] = Vec8<float>::load(yg + (long)r * C + col0);
#pragma unroll
            for (int i = 0; i < 8; ++i) part[0][i] += v[j].get(i);
        }
    }
    strip_reduce<1>(part, red, tot);
    if (threadIdx.x < 32) {
        const float mu = (float)(tot[threadIdx.x] / (double)M);
        mu_s[threadIdx.x] = mu;
        cmean[(long)g * C + blockIdx.x * 32 + threadIdx.x] = mu;
    }
    __syncthreads();
    float mu[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) mu[i] = mu_s[cl * 8 + i];
    bf16* og = yc + (long)g * M * C;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = r0 + 64 * j;
        if (r < M) {
            Vec8<bf16> o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.set(i, v[j].get(i) - mu[i]);
            o.store(og + (long)r * C + col0);
        }
    }
}
// running_mean[j] += sum_g momentum (1 - momentum)^(G-1-g) * (cmean[g] . W[j]): what the running mean of the BatchNorm behind
// W (bf16 [D][K], row-major) would have seen from the un-centred input.  One wave per output row j.
__device__ __forceinline__ void shift_running_mean_body(const bf16* __restrict__ W, const float* __restrict__ cmean,
                                                        float* __restrict__ running_mean, int D, int K, int groups, float momentum);
__global__ __launch_bounds__(256) void shift_running_mean_kernel(const bf16* __restrict__ W, const float* __restrict__ cmean,
                                                                 float* __restrict__ running_mean, int D, int K, int groups,
                                                                 float momentum) {
    shift_running_mean_body(W, cmean, running_mean, D, K, groups, momentum);
}
struct ShiftMulti { const bf16* W[MAXP]; const float* cmean[MAXP]; float* rm[MAXP]; int K[MAXP]; };
__global__ __launch_bounds__(256) void shift_running_mean_multi_kernel(ShiftMulti m, int D, int groups, float momentum) {
    const int p = blockIdx.y;
    shift_running_mean_body(m.W[p], m.cmean[p], m.rm[p], D, m.K[p], groups, momentum);
}
__device__ __forceinline__ void shift_running_mean_body(const bf16* __restrict__ W, const float* __restrict__ cmean,
                                                        float* __restrict__ running_mean, int D, int K, int groups, float momentum) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= D) return;
    float acc = 0.f, coef = momentum;
    for (int g = groups - 1; g >= 0; --g) {
        const float* cg = cmean + (long)g * K;
        float t = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            const Vec8<bf16> w = Vec8<bf16>::load(W + (long)j * K + k);
            const Vec8<float> cv = Vec8<float>::load(cg + k);
#pragma unroll
            for (int i = 0; i < 8; ++i) t += w.get(i) * cv.get(i);
        }
        acc += coef * t;
        coef *= (1.f - momentum);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) running_mean[j] += acc;
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}
}  // namespace

extern "C" int audiossl_relu_bwd(int dtype, const void* g, const void* h, void* out, long n, void* stream) {
    ASSL_REQUIRE(g && h && out && n > 0 && (n % 8) == 0 && (dtype == 0 || dtype == 1));
    if (dtype == 0) hipLaunchKernelGGL(relu_bwd_kernel<float>, GRID1(n / 8), dim3(256), 0, S_(stream), (const float*)g, (const float*)h, (float*)out, n / 8);
    else            hipLaunchKernelGGL(relu_bwd_kernel<bf16>, GRID1(n / 8), dim3(256), 0, S_(stream), (const bf16*)g, (const bf16*)h, (bf16*)out, n / 8);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                       float eps, int C, float* scale, float* shift, void* stream) {
    ASSL_REQUIRE(running_mean && running_var && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, S_(stream), gamma, beta, running_mean,
                       running_var, eps, C, scale, shift);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_split_bf16(const float* x, void* hi, void* lo, long n, void* stream) {
    ASSL_REQUIRE(x && hi && lo && n > 0);
    const int grid = (int)min((long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid), dim3(256), 0, S_(stream), x, (bf16*)hi, (bf16*)lo, n);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_center_cast(const float* y, void* yc, float* cmean, int groups, long M, int C, void* stream) {
    ASSL_REQUIRE(y && yc && cmean && groups > 0 && M > 0 && M <= 1024 && C > 0 && C % 32 == 0);
    ASSL_REQUIRE(ASSL_ALIGNED16(y) && ASSL_ALIGNED16(yc));
    const dim3 grid(C / 32, groups);
    if (M <= 512) hipLaunchKernelGGL(center_cast_kernel<8>, grid, dim3(256), 0, S_(stream), y, (bf16*)yc, cmean, (int)M, C);
    else          hipLaunchKernelGGL(center_cast_kernel<16>, grid, dim3(256), 0, S_(stream), y, (bf16*)yc, cmean, (int)M, C);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_center_cast_multi(int count, const float* const* y, void* const* yc, float* const* cmean, const int* C, int groups,
                                          long M, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAXP && y && yc && cmean && C && groups > 0 && M > 0 && M <= 1024);
    CenterMulti m;
    int cmax = 0;
    for (int i = 0; i < count; ++i) {
        ASSL_REQUIRE(y[i] && yc[i] && cmean[i] && C[i] > 0 && C[i] % 32 == 0 && ASSL_ALIGNED16(y[i]) && ASSL_ALIGNED16(yc[i]));
        m.y[i] = y[i]; m.yc[i] = static_cast<bf16*>(yc[i]); m.cmean[i] = cmean[i]; m.C[i] = C[i];
        cmax = max(cmax, C[i]);
    }
    const dim3 grid(cmax / 32, groups, count);
    if (M <= 512) hipLaunchKernelGGL(center_cast_multi_kernel<8>, grid, dim3(256), 0, S_(stream), m, (int)M);
    else          hipLaunchKernelGGL(center_cast_multi_kernel<16>, grid, dim3(256), 0, S_(stream), m, (int)M);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_shift_running_mean_multi(int count, const void* const* W, const float* const* cmean, float* const* running_mean,
                                                 int D, const int* K, int groups, float momentum, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAXP && W && cmean && running_mean && K && D > 0 && groups > 0);
    ShiftMulti m;
    for (int i = 0; i < count; ++i) {
        ASSL_REQUIRE(W[i] && cmean[i] && running_mean[i] && K[i] > 0 && K[i] % 8 == 0 && ASSL_ALIGNED16(W[i]) && ASSL_ALIGNED16(cmean[i]));
        m.W[i] = static_cast<const bf16*>(W[i]); m.cmean[i] = cmean[i]; m.rm[i] = running_mean[i]; m.K[i] = K[i];
    }
    hipLaunchKernelGGL(shift_running_mean_multi_kernel, dim3(ceil_div(D, 4), count), dim3(256), 0, S_(stream), m, D, groups, momentum);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_shift_running_mean(const void* W, const float* cmean, float* running_mean, int D, int K, int groups,
                                           float momentum, void* stream) {
    ASSL_REQUIRE(W && cmean && running_mean && D > 0 && K > 0 && K % 8 == 0 && groups > 0);
    ASSL_REQUIRE(ASSL_ALIGNED16(W) && ASSL_ALIGNED16(cmean));
    hipLaunchKernelGGL(shift_running_mean_kernel, dim3(ceil_div(D, 4)), dim3(256), 0, S_(stream), (const bf16*)W, cmean,
                       running_mean, D, K, groups, momentum);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_axpy(float* y, const float* x, float a, long n, void* stream) {
    ASSL_REQUIRE(y && x && n > 0);
    const int grid = (int)min((long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(256), 0, S_(stream), y, x, a, n);
    ASSL_LAUNCH_CHECK();
}
