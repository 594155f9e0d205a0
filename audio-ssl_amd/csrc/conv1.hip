// K6: the 1->64 channel stem of AudioNTT2020Task6 (`src/encoder/audiontt.py:46-50, 74`):
// Conv2d(1,64,3,pad 1) + BatchNorm2d(train) + ReLU + MaxPool2d(2), forward and backward, without ever
// materialising the full-resolution 64-channel tensor (1.65 MB per view in fp32).
//
// A 3x3 convolution of a single-channel image is linear in 9 shifted copies x_tap of that image, so the
// batch statistics of every output channel follow from the 9 first and 45 second moments of the taps:
//   mean_c = w_c . S1 / n + b_c,   var_c = w_c^T Cov w_c            (moments kernel + finalize, fp64)
// and the forward is ONE pass that recomputes the conv per pooled pixel and writes only the pooled map.
// The backward never needs dX (the input is data), and BN's correction terms are again linear in the
// same moments:
//   dW_c,tap = gamma rstd [ G_c,tap - dbeta_c/n S1_tap - dgamma_c/n * rstd (w_c . S2[:,tap] + (b_c-mean_c) S1_tap) ]
// with G = sum over pooled pixels of (routed, ReLU-gated) gradient x tap value, so it is ONE more pass.
// Layouts: image [N][F][T] fp32 (NCHW, C=1); pooled output [N][T/2][F/2][64] (time, mel, channel) in T_.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int NTAP = 9, NMOM = 9 + 45;
constexpr int MOM_REPL = 16;  // replicas of the moment accumulator (`mom` scratch = MOM_REPL * 54 doubles)

__device__ __forceinline__ int tri(int a, int b) {   // index of pair (a<=b) in the packed upper triangle
    return a * 9 - a * (a - 1) / 2 + (b - a);
}

// One workgroup per (image, chunk of MOM_TC time columns): the chunk (+ halo, zero outside the image) is staged in LDS once,
// each thread then walks pixels p = tid, tid+256, ... of the chunk reading its 9 taps from LDS (lanes run along time:
// conflict-free), 54 FMAs per pixel.  (A flat grid-stride loop with 9 global loads per pixel was latency-bound: 52 us at
// N = 512 for 13 MB of input.)
constexpr int MOM_TC = 128;
__global__ __launch_bounds__(256) void conv1_moments_kernel(const float* __restrict__ img, double* __restrict__ mom,
                                                            int N, int F, int T, int nchunk, int tcw) {
    extern __shared__ float tile[];                               // [(F+2)][tc+2]
    __shared__ float sh[4][NMOM];
    const int n = blockIdx.x / nchunk, t0 = (blockIdx.x - n * nchunk) * tcw;
    const int tc = min(tcw, T - t0), ld = tc + 2;
    const float* im = img + (long)n * F * T;
    {
        // eight independent loads in flight per thread and pass (a one-load-per-iteration loop with a division per element
        // spent ~40 us of this kernel waiting); (row, column) of element i advance incrementally: i += 256
        const int total = (F + 2) * ld, dr = 256 / ld, dc = 256 - dr * ld;
        int r = threadIdx.x / ld, cidx = threadIdx.x - r * ld;
        for (int i0 = threadIdx.x; i0 < total; i0 += 8 * 256) {
            float v[8];
            int rr = r, cc = cidx;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = rr - 1, t = t0 + cc - 1;
                v[k] = (i0 + k * 256 < total && f >= 0 && f < F && t >= 0 && t < T) ? im[f * T + t] : 0.f;
                cc += dc; rr += dr;
                if (cc >= ld) { cc -= ld; ++rr; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + k * 256 < total) tile[i0 + k * 256] = v[k];
            r = rr; cidx = cc;
        }
    }
    __syncthreads();
    float acc[NMOM];
#pragma unroll
    for (int i = 0; i < NMOM; ++i) acc[i] = 0.f;
    const int df = 256 / tc, dt = 256 - df * tc;                  // p += 256  ==  (f, t) += (df, dt) with carry
    int f = threadIdx.x / tc, t = threadIdx.x - f * tc;
    for (; f < F;) {
        const float* c = tile + f * ld + t;                       // top-left tap of pixel (f, t)
        float x[NTAP];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = c[kh * ld + kw];
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            acc[a] += x[a];
#pragma unroll
            for (int b = 0; b < 9; ++b)                           // (a <= b) pairs; the index is a constant after unrolling - a running
                if (b >= a) acc[9 + tri(a, b)] += x[a] * x[b];    // counter (acc[q++]) left the array in scratch memory: 46 MB of traffic
        }
        t += dt; f += df;
        if (t >= tc) { t -= tc; ++f; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NMOM; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) sh[w][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NMOM) {
        const double v = (double)sh[0][threadIdx.x] + (double)sh[1][threadIdx.x] + (double)sh[2][threadIdx.x] +
                         (double)sh[3][threadIdx.x];
        atomicAdd(&mom[(blockIdx.x & (MOM_REPL - 1)) * NMOM + threadIdx.x], v);     // replicated: same-address atomics serialise
    }
}

// 576 threads = (channel, tap a): the 81 covariances once per workgroup, per thread the row sum_b w_b cov(a, b), then the 9 rows of
// a channel folded in tap order by its first thread.  (One thread per channel walking all 81 products after a serial fold of the
// replicas: 9 us at the head of both encoder chains.)
__global__ __launch_bounds__(576) void conv1_finalize_kernel(double* __restrict__ momr, const float* __restrict__ w,
                                      const float* __restrict__ bias, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float* running_mean, float* running_var,
                                      float momentum, float eps, double count, float* scale, float* shift,
                                      float* save_mean, float* save_rstd, int repl) {
    __shared__ double tot[NMOM];
    __shared__ double cov[81];
    __shared__ double rowv[64][9], rowm[64][9];
    const int i = threadIdx.x;
    // fold the replicas into mom[0..53] (kept for the backward), then every channel reads the totals
    if (i < NMOM) {
        double t = 0.0;
        for (int r = 0; r < repl; ++r) t += momr[r * NMOM + i];
        tot[i] = t;
    }
    __syncthreads();
    if (i < NMOM) momr[i] = tot[i];
    if (i < 81) {
        const int a = i / 9, bb = i - a * 9;
        const double s2 = tot[9 + (a <= bb ? tri(a, bb) : tri(bb, a))] / count;
        cov[i] = s2 - (tot[a] / count) * (tot[bb] / count);
    }
    __syncthreads();
    const int c = i / 9, a = i - (i / 9) * 9;
    if (i < 576) {
        double r = 0.0;
        for (int bb = 0; bb < 9; ++bb) r += (double)w[c * 9 + bb] * cov[a * 9 + bb];
        rowv[c][a] = (double)w[c * 9 + a] * r;
        rowm[c][a] = (double)w[c * 9 + a] * tot[a];
    }
    __syncthreads();
    if (i < 576 && a == 0) {
        double mean = 0.0, var = 0.0;
        for (int k = 0; k < 9; ++k) { mean += rowm[c][k]; var += rowv[c][k]; }
        mean = mean / count;
        var = var < 0.0 ? 0.0 : var;
        mean += (double)bias[c];
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const float sc = (float)((double)gamma[c] * rstd);
        scale[c] = sc;
        shift[c] = (float)((double)beta[c] - mean * (double)gamma[c] * rstd);
        save_mean[c] = (float)mean;
        save_rstd[c] = (float)rstd;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * count / (count - 1.0));
        }
    }
}

// Stage the 4 time columns x (F+2) mel rows an (n, tp) item needs: patch[f+1][k] = img[n][f][2tp-1+k].
__device__ __forceinline__ void load_patch(float* patch, const float* __restrict__ img, int n, int tp, int F, int T) {
    const float* im = img + (long)n * F * T;
    for (int i = threadIdx.x; i < (F + 2) * 4; i += 256) {
        const int f = (i >> 2) - 1, t = 2 * tp - 1 + (i & 3);
        patch[i] = (f >= 0 && f < F && t >= 0 && t < T) ? im[f * T + t] : 0.f;
    }
}

// Register-staged variant: fetch the NEXT item's patch while the current one is being consumed.
struct PatchRegs {
    float v[2];
    __device__ __forceinline__ void fetch(const float* __restrict__ img, int n, int tp, int F, int T) {
        const float* im = img + (long)n * F * T;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + j * 256;
            const int f = (i >> 2) - 1, t = 2 * tp - 1 + (i & 3);
            v[j] = (i < (F + 2) * 4 && f >= 0 && f < F && t >= 0 && t < T) ? im[f * T + t] : 0.f;
        }
    }
    __device__ __forceinline__ void put(float* patch, int F) const {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < (F + 2) * 4) patch[i] = v[j];
        }
    }
};

// conv outputs of the 2x2 pooling window at pooled mel row fp: y[p], p = 2*df + dt (torch's scan order)
__device__ __forceinline__ void conv4(const float* patch, int fp, const float* w, float bias, float* y, float (*x)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                              // rows f = 2fp-1 .. 2fp+2, one 16-byte read each
        const f32x4 row = *reinterpret_cast<const f32x4*>(patch + (2 * fp + r) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[r][k] = row[k];
    }
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            float s = bias;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) s += w[kh * 3 + kw] * x[df + kh][dt + kw];
            y[2 * df + dt] = s;
        }
}

template <typename T_>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, T_* __restrict__ out,
                                                        int N, int F, int T) {
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int To = T / 2, Fo = F / 2;
    float wr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wr[i] = w[lane * 9 + i];
    const float b = bias[lane], sc = scale[lane], sh = shift[lane];
    PatchRegs pre;
    int item = blockIdx.x;
    if (item < N * To) { pre.fetch(img, item / To, item % To, F, T); pre.put(patch, F); }
    __syncthreads();
    for (; item < N * To; item += gridDim.x) {
        const int n = item / To, tp = item - n * To;
        const int nxt = item + gridDim.x;
        if (nxt < N * To) pre.fetch(img, nxt / To, nxt % To, F, T);
        for (int fp = wv; fp < Fo; fp += 4) {
            float y[4], x[4][4];
            conv4(patch, fp, wr, b, y, x);
            float m = fmaxf(sc * y[0] + sh, 0.f);
#pragma unroll
            for (int p = 1; p < 4; ++p) m = fmaxf(m, fmaxf(sc * y[p] + sh, 0.f));
            out[(((long)n * To + tp) * Fo + fp) * 64 + lane] = from_f32<T_>(m);
        }
        __syncthreads();
        if (nxt < N * To) pre.put(patch, F);
        __syncthreads();
    }
}

// acc[c][0..8] = G, acc[c][9] = dbeta, acc[c][10] = dgamma   (fp32 atomics, one set per block)
template <typename T_>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const T_* __restrict__ dP,
                                                        const T_* __restrict__ dxl, float inv_To, float* __restrict__ acc,
                                                        int N, int F, int T) {
    extern __shared__ __attribute__((aligned(16))) float patch[];
    __shared__ float red[4][64][11];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int To = T / 2, Fo = F / 2;
    float wr[9], G[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { wr[i] = w[lane * 9 + i]; G[i] = 0.f; }
    const float b = bias[lane], sc = scale[lane], sh = shift[lane], mu = mean[lane], rs = rstd[lane];
    float dbeta = 0.f, dgamma = 0.f;
    PatchRegs pre;
    int item = blockIdx.x;
    if (item < N * To) { pre.fetch(img, item / To, item % To, F, T); pre.put(patch, F); }
    __syncthreads();
    for (; item < N * To; item += gridDim.x) {
        const int n = item / To, tp = item - n * To;
        const int nxt = item + gridDim.x;
        if (nxt < N * To) pre.fetch(img, nxt / To, nxt % To, F, T);
        // all of this wave's pooled gradients of the item are fetched up front (<= 8 independent loads in flight): the
        // per-pixel dependent load was the kernel's critical path (latency-bound at 0.5 TB/s)
        float gin[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int fp = wv + 4 * i;
            gin[i] = 0.f;
            if (fp < Fo) {
                gin[i] = to_f32(dP[(((long)n * To + tp) * Fo + fp) * 64 + lane]);
                if (dxl) gin[i] += to_f32(dxl[(long)n * Fo * 64 + fp * 64 + lane]) * inv_To;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int fp = wv + 4 * i;
            if (fp >= Fo) break;
            float y[4], x[4][4];
            conv4(patch, fp, wr, b, y, x);
            int best = 0;
            float m = sc * y[0] + sh;
#pragma unroll
            for (int p = 1; p < 4; ++p) { const float a = sc * y[p] + sh; if (a > m) { m = a; best = p; } }
            const float g = gin[i];
            const float da = m > 0.f ? g : 0.f;
            const float ybest = best == 0 ? y[0] : best == 1 ? y[1] : best == 2 ? y[2] : y[3];
            dbeta += da;
            dgamma += da * (ybest - mu) * rs;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float dm = (best == p) ? da : 0.f;
                const int df = p >> 1, dt = p & 1;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) G[kh * 3 + kw] += dm * x[df + kh][dt + kw];      // registers, not LDS
            }
        }
        __syncthreads();
        if (nxt < N * To) pre.put(patch, F);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) red[wv][lane][i] = G[i];
    red[wv][lane][9] = dbeta;
    red[wv][lane][10] = dgamma;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 11; i += 256) {
        const int c = i / 11, k = i - c * 11;
        // 32 replicas of the accumulator (same-address float atomics from every workgroup serialise at the memory side)
        atomicAdd(&acc[(blockIdx.x & 31) * 704 + i], red[0][c][k] + red[1][c][k] + red[2][c][k] + red[3][c][k]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 MFMA formulation of the stem (the fp32 VALU kernels above are the validation path; they are VALU-bound: 7.5 GFLOP of
// fp32 FMAs per 512-clip step forward).  One wave owns 8 pooled mel rows of an (image, pooled time column) item = 32 conv
// pixels = one 32-row MFMA tile, ordered row = 4 * pooled_pixel + window_position, so that in the C/D map
// (row = (r & 3) + 8 * (r >> 2) + 4 * half) register r of a lane is window position r & 3 of pooled pixel 2 * (r >> 2) + half:
// BatchNorm + ReLU + the 2x2 max are in-register per lane (= channel).  K = 9 taps padded to 16: lanes of the lower half
// carry taps 0..7, the upper half tap 8 and zeros.
__device__ __forceinline__ Vec8<bf16> stem_taps(const float* patch, int f, int dt, int half) {
    Vec8<bf16> a;
    if (half == 0) {
        a.v[0] = (bf16)patch[(f + 0) * 4 + dt + 0]; a.v[1] = (bf16)patch[(f + 0) * 4 + dt + 1]; a.v[2] = (bf16)patch[(f + 0) * 4 + dt + 2];
        a.v[3] = (bf16)patch[(f + 1) * 4 + dt + 0]; a.v[4] = (bf16)patch[(f + 1) * 4 + dt + 1]; a.v[5] = (bf16)patch[(f + 1) * 4 + dt + 2];
        a.v[6] = (bf16)patch[(f + 2) * 4 + dt + 0]; a.v[7] = (bf16)patch[(f + 2) * 4 + dt + 1];
    } else {
        a = Vec8<bf16>::zero();
        a.v[0] = (bf16)patch[(f + 2) * 4 + dt + 2];
    }
    return a;
}
// weights as the B operand: B[k = tap][n = channel], lane = channel 32 j + (lane & 31), k = 8 half + jj
__device__ __forceinline__ Vec8<bf16> stem_weights(const float* w, int c, int half) {
    Vec8<bf16> b = Vec8<bf16>::zero();
    if (half == 0) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) b.v[jj] = (bf16)w[c * 9 + jj];
    } else {
        b.v[0] = (bf16)w[c * 9 + 8];
    }
    return b;
}

__global__ __launch_bounds__(256) void conv1_fwd_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, bf16* __restrict__ out,
                                                             float* __restrict__ xl, int XS, int N, int F, int T) {
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int To = T / 2, Fo = F / 2;
    // the item's pooled output [Fo][64] is collected in LDS and written as whole 16-byte pieces (it is one contiguous run of
    // Fo * 128 bytes): straight from the accumulator layout it was 2 bytes per lane - ~820 k store instructions per launch
    bf16* const stage = reinterpret_cast<bf16*>(patch + (((F + 2) * 4 + 3) & ~3));
    Vec8<bf16> wb[2];
    float b[2], sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = 32 * j + l31;
        wb[j] = stem_weights(w, c, half);
        b[j] = bias[c]; sc[j] = scale[c]; sh[j] = shift[c];
    }
    const int pp = l31 >> 2, df = (l31 >> 1) & 1, dt = l31 & 1;          // this lane's A row: pooled pixel pp, window (df, dt)
    PatchRegs pre;
    // items (image, pooled time column): strided over the grid - or, when the layer output x_l = mean over time of the pooled
    // map is wanted too (`audiontt.py:76-78`), XS workgroups per image walking a share of its columns each, so the column sums
    // stay in registers; each leaves its part in xl[share][N][Fo*64] and tmean3_fwd adds the parts in a fixed order (a separate
    // pass re-read the whole 105 MB map; one workgroup per image was no faster than that pass; atomics are not reproducible)
    int item, step, end;
    if (xl) {
        const int n = blockIdx.x / XS, s4 = blockIdx.x - n * XS, per = (To + XS - 1) / XS;
        item = n * To + s4 * per; step = 1; end = n * To + min(To, (s4 + 1) * per);
    } else {
        item = blockIdx.x; step = gridDim.x; end = N * To;
    }
    float xs[2][2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[u][j][q] = 0.f;
    if (item < end) { pre.fetch(img, item / To, item % To, F, T); pre.put(patch, F); }
    __syncthreads();
    for (; item < end; item += step) {
        const int n = item / To, tp = item - n * To;
        const int nxt = item + step;
        if (nxt < end) pre.fetch(img, nxt / To, nxt % To, F, T);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tile = wv + 4 * u;                          // Fo <= 63: at most two 8-row tiles per wave
            if (tile * 8 >= Fo) break;
            const int fp = min(tile * 8 + pp, Fo - 1);
            const Vec8<bf16> a = stem_taps(patch, 2 * fp + df, dt, half);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, wb[j].v, acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float m = 0.f;
#pragma unroll
                    for (int r3 = 0; r3 < 4; ++r3) m = fmaxf(m, sc[j] * (acc[4 * q + r3] + b[j]) + sh[j]);
                    const int fpo = tile * 8 + 2 * q + half;
                    if (fpo < Fo) {
                        const bf16 mb = (bf16)m;
                        stage[fpo * 64 + 32 * j + l31] = mb;
                        xs[u][j][q] += (float)mb;
                    }
                }
            }
        }
        __syncthreads();
        if (nxt < end) pre.put(patch, F);
        bf16* o = out + (((long)n * To + tp) * Fo) * 64;
        for (int v = threadIdx.x; v < Fo * 8; v += 256) Vec8<bf16>::load(stage + v * 8).store(o + v * 8);
        __syncthreads();
    }
    if (xl) {
        const int n = blockIdx.x / XS, s4 = blockIdx.x - n * XS;
        const float inv = 1.f / (float)To;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int fpo = (wv + 4 * u) * 8 + 2 * q + half;
                    if (fpo < Fo) {
                        xl[((long)s4 * N + n) * Fo * 64 + fpo * 64 + 32 * j + l31] = xs[u][j][q] * inv;   // this quarter's part
                    }
                }
    }
}

// Backward with the same MFMA recompute (so the arg-max of the pooling window is the forward's) and the 9 tap sums
// G[c][tap] = sum_pixels dz[pixel][c] * x_tap[pixel] as a second MFMA: A = x taps (row = tap, k = pixel), B = the routed
// gradient straight out of the accumulator registers (lane = channel, k = pixel in the register order - the A operand reads
// its pixels from the LDS patch in that same order).
// TP: type of the pooled gradient dP (fp32, or bf16 - half the bytes of the largest tensor this kernel reads); dxl is fp32
template <bool SLAB, typename TP>
__global__ __launch_bounds__(256, 3) void conv1_bwd_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const TP* __restrict__ dP,
                                                             const float* __restrict__ dxl, float inv_To, float* __restrict__ accg,
                                                             int N, int F, int T) {
    // SLAB (n_mels = 64): the item's 8 KB slab of pooled gradients [32 rows][64 ch] fp32 is DMA'd straight into LDS
    // (global_load_lds, no registers) one item ahead - with three workgroups per CU the per-lane loads it replaces left
    // ~0.6 TB/s of reads in flight.  Dynamic LDS: two slabs, then the input patch.
    extern __shared__ __attribute__((aligned(1024))) float dyn[];
    float* const patch = dyn + (SLAB ? 2 * 2048 : 0);
    __shared__ float red[4][64][11];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int To = T / 2, Fo = F / 2;
    Vec8<bf16> wb[2];
    float b[2], sc[2], sh[2], mu[2], rs[2], dbeta[2] = {0.f, 0.f}, dgamma[2] = {0.f, 0.f};
    f32x16 G[2];
    auto slab_dma = [&](int it, int slot) {
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const char* src = reinterpret_cast<const char*>(dP + (long)it * Fo * 64);   // (n, tp) items are contiguous [Fo][64] slabs
        char* dst = reinterpret_cast<char*>(dyn + slot * 2048);
#pragma unroll
        for (int k = 0; k < (int)sizeof(TP) / 2; ++k)                      // 4 KB (bf16) or 8 KB (fp32) per item: 1 KB per wave and trip
            __builtin_amdgcn_global_load_lds((gptr_t)(src + ((k * 4 + wv) * 64 + lane) * 16),
                                             (lptr_t)(dst + (k * 4 + wv) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = 32 * j + l31;
        wb[j] = stem_weights(w, c, half);
        b[j] = bias[c]; sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; rs[j] = rstd[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) G[j][r] = 0.f;
    }
    const int pp = l31 >> 2, df = (l31 >> 1) & 1, dt = l31 & 1;
    const int tap = l31, kh = tap / 3, kw = tap - 3 * kh;               // A2 row of this lane (taps >= 9: zero rows)
    // pooled gradients of this lane's 2 channels x 4 pooled pixels of one (item, tile)
    // dxl term (gradient of the layer-mean output, [N][Fo*64], shared by all time columns) of one (item, tile)
    const float dxs = dxl ? inv_To : 0.f;
    auto fetch_dxl = [&](int it, int tile, float (&gx)[2][4]) {
        const int n = it / To;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int fpo = min(tile * 8 + 2 * q + half, Fo - 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) gx[j][q] = dxl ? dxl[(long)n * Fo * 64 + fpo * 64 + 32 * j + l31] : 0.f;   // raw: scaled at use
        }
    };
    // pooled gradients of this lane's 2 channels x 4 pooled pixels of one (item, tile)
    auto fetch_gp = [&](int it, int tile, int slot, const float (&gx)[2][4], float (&gp)[2][4]) {
        const int n = it / To, tp = it - n * To;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int fpo = tile * 8 + 2 * q + half;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float g = 0.f;
                if (fpo < Fo)
                    g = (float)(SLAB ? reinterpret_cast<const TP*>(dyn + slot * 2048)[fpo * 64 + 32 * j + l31]
                                     : dP[(((long)n * To + tp) * Fo + fpo) * 64 + 32 * j + l31]) + gx[j][q] * dxs;
                gp[j][q] = g;
            }
        }
    };
    PatchRegs pre;
    int item = blockIdx.x;
    float gp[2][4], gx[2][4], gxn[2][4];
    if (item < N * To) {
        pre.fetch(img, item / To, item % To, F, T);
        pre.put(patch, F);
        if (SLAB) { fetch_dxl(item, wv, gx); slab_dma(item, 0); }
    }
    __syncthreads();                                                    // (drains the first slab's DMA as well)
    int slot = 0;
    for (; item < N * To; item += gridDim.x, slot ^= 1) {
        const int nxt = item + gridDim.x;
        if (nxt < N * To) {
            // everything for the NEXT item is requested here and consumed after the barrier that ends this one: no
            // register-destination load is waited for in between (hipcc drains vmcnt to 0 for those, DMA included)
            pre.fetch(img, nxt / To, nxt % To, F, T);
            if (SLAB) { fetch_dxl(nxt, wv, gxn); slab_dma(nxt, slot ^ 1); }
        }
        for (int tile = wv; tile * 8 < Fo; tile += 4) {
            if (!SLAB) fetch_dxl(item, tile, gx);
            fetch_gp(item, tile, slot, gx, gp);
            const int fp = min(tile * 8 + pp, Fo - 1);
            const Vec8<bf16> a = stem_taps(patch, 2 * fp + df, dt, half);
            // x taps as the A operand of the G product: element jj of k-step st is pixel 16 st + (jj & 3) + 8 (jj >> 2) + 4 half
            // (hi + lo splitting of both operands was tried: the weight-gradient error of this flavour comes from pooling
            // windows whose arg-max differs from the fp32 convolution's, not from the rounding of the tap sums)
            Vec8<bf16> xh[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                xh[st] = Vec8<bf16>::zero();
                if (tap < 9) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int pix = 16 * st + (jj & 3) + 8 * (jj >> 2) + 4 * half;
                        const int fpx = min(tile * 8 + (pix >> 2), Fo - 1), wp = pix & 3;
                        xh[st].v[jj] = (bf16)patch[(2 * fpx + (wp >> 1) + kh) * 4 + (wp & 1) + kw];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, wb[j].v, acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int best = 0;
                    float ybest = acc[4 * q] + b[j];
                    float m = sc[j] * ybest + sh[j];
#pragma unroll
                    for (int r3 = 1; r3 < 4; ++r3) {
                        const float y = acc[4 * q + r3] + b[j], z = sc[j] * y + sh[j];
                        if (z > m) { m = z; best = r3; ybest = y; }
                    }
                    const float da = m > 0.f ? gp[j][q] : 0.f;
                    dbeta[j] += da;
                    dgamma[j] += da * (ybest - mu[j]) * rs[j];
#pragma unroll
                    for (int r3 = 0; r3 < 4; ++r3) acc[4 * q + r3] = (r3 == best) ? da : 0.f;          // routed gradient dz
                }
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    Vec8<bf16> gh;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) gh.v[jj] = (bf16)acc[8 * st + jj];
                    G[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[st].v, gh.v, G[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (nxt < N * To) pre.put(patch, F);
        if (SLAB) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) gx[j][q] = gxn[j][q];
        }
        __syncthreads();
    }
    // G[j][r]: row (tap) = (r & 3) + 8 (r >> 2) + 4 half, column = channel 32 j + l31
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = 32 * j + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (t < 9) red[wv][c][t] = G[j][r];
        }
        const float db = dbeta[j] + __shfl_xor(dbeta[j], 32, 64), dg = dgamma[j] + __shfl_xor(dgamma[j], 32, 64);
        if (half == 0) { red[wv][c][9] = db; red[wv][c][10] = dg; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 11; i += 256) {
        const int c = i / 11, k = i - c * 11;
        atomicAdd(&accg[(blockIdx.x & 31) * 704 + i], red[0][c][k] + red[1][c][k] + red[2][c][k] + red[3][c][k]);
    }
}

// 704 threads: thread i folds entry i of the 32 replicas (fp64, replica order - coalesced across the workgroup; one thread per channel
// walking 11 x 32 strided loads took 12 us at the serial end of the step), then one thread per (channel, tap) finishes dW
__global__ __launch_bounds__(704) void conv1_bwd_finalize_kernel(const float* __restrict__ acc, const double* __restrict__ mom,
                                          const float* __restrict__ w, const float* __restrict__ bias,
                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, double count, float* dW, float* dbias,
                                          float* dgamma, float* dbeta, const float* __restrict__ gstat, float* lstat) {
    __shared__ double s11[704];                                   // [channel][9 tap sums, dbeta, dgamma]
    const int i = threadIdx.x;
    if (i < 704) {
        double t = 0.0;
        for (int r = 0; r < 32; ++r) t += (double)acc[r * 704 + i];
        s11[i] = t;
    }
    __syncthreads();
    if (lstat) {                                                  // sums only: this rank's (dbeta, dgamma)
        if (i < 64) { lstat[i] = (float)s11[i * 11 + 9]; lstat[64 + i] = (float)s11[i * 11 + 10]; }
        return;
    }
    if (i < 576) {
        const int c = i / 9, t = i - c * 9;
        // SyncBatchNorm: the two means of the BatchNorm backward are over the GLOBAL batch (gstat = all-reduced sums, count global);
        // the parameter gradients and the tap sums stay this rank's own
        const double db = gstat ? (double)gstat[c] : s11[c * 11 + 9], dg = gstat ? (double)gstat[64 + c] : s11[c * 11 + 10];
        const double rs = rstd[c], mu = mean[c], gm = gamma[c];
        double wS2 = 0.0;
        for (int a = 0; a < 9; ++a) wS2 += (double)w[c * 9 + a] * mom[9 + (a <= t ? tri(a, t) : tri(t, a))];
        const double yhat_x = rs * (wS2 + ((double)bias[c] - mu) * mom[t]);           // sum_pos yhat_c * x_tap
        const double v = gm * rs * (s11[c * 11 + t] - db / count * mom[t] - dg / count * yhat_x);
        dW[c * 9 + t] += (float)v;
    }
    if (i < 64) {
        dgamma[i] += (float)s11[i * 11 + 10];
        dbeta[i] += (float)s11[i * 11 + 9];
    }
    (void)dbias;            // d(conv bias) is identically zero under train-mode BN
}

void launch_moments(const float* img, double* mom, int N, int F, int T, hipStream_t s) {
    // narrower chunks = more workgroups = more fp64 atomics at the end: 31 / 37 / 55 / 82 us at 128 / 64 / 32 / 16 columns (B = 512)
    const int tcw = MOM_TC;
    const int nchunk = ceil_div(T, tcw);
    const size_t lds = sizeof(float) * (F + 2) * (min(T, tcw) + 2);
    hipLaunchKernelGGL(conv1_moments_kernel, dim3(N * nchunk), dim3(256), lds, s, img, mom, N, F, T, nchunk, tcw);
}

}  // namespace

extern "C" int audiossl_conv1_stats(const float* img, int N, int F, int T, const float* w, const float* bias,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, double* mom, float* scale, float* shift,
                                    float* save_mean, float* save_rstd, void* stream) {
    ASSL_REQUIRE(img && w && bias && gamma && beta && mom && scale && shift && save_mean && save_rstd);
    ASSL_REQUIRE(N > 0 && F >= 2 && F <= 120 && T >= 2 && (long)N * F * T < 0x7FFFFFFFL - 0x1000000L);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ASSL_ZERO(mom, sizeof(double) * NMOM * MOM_REPL, s);
    const long total = (long)N * F * T;
    // every workgroup ends with 54 fp64 atomics; on ONE set of 54 addresses they serialise at the memory side -
    // MOM_REPL replicas of the accumulator, folded by the finalize kernel
    launch_moments(img, mom, N, F, T, s);
    hipLaunchKernelGGL(conv1_finalize_kernel, dim3(1), dim3(576), 0, s, mom, w, bias, gamma, beta, running_mean,
                       running_var, momentum, eps, (double)total, scale, shift, save_mean, save_rstd, MOM_REPL);
    ASSL_LAUNCH_CHECK();
}

// ---- the same in two halves, for statistics that are exchanged between ranks in between (SyncBatchNorm of the
// DeepCluster-v2 trainer, extras/decar-v2/main.py:82): conv1_moments leaves this rank's 54 tap-moment totals in mom[0..53];
// conv1_finalize turns (all-reduced) totals and the GLOBAL element count into scale / shift / mean / rstd + running buffers.
__global__ void conv1_fold_kernel(double* __restrict__ momr) {
    __shared__ double tot[NMOM];
    const int c = threadIdx.x;
    if (c < NMOM) {
        double t = 0.0;
        for (int r = 0; r < MOM_REPL; ++r) t += momr[r * NMOM + c];
        tot[c] = t;
    }
    __syncthreads();
    if (c < NMOM) momr[c] = tot[c];
}
extern "C" int audiossl_conv1_moments(const float* img, int N, int F, int T, double* mom, void* stream) {
    ASSL_REQUIRE(img && mom && N > 0 && F >= 2 && F <= 120 && T >= 2 && (long)N * F * T < 0x7FFFFFFFL - 0x1000000L);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ASSL_ZERO(mom, sizeof(double) * NMOM * MOM_REPL, s);
    launch_moments(img, mom, N, F, T, s);
    hipLaunchKernelGGL(conv1_fold_kernel, dim3(1), dim3(64), 0, s, mom);
    ASSL_LAUNCH_CHECK();
}
extern "C" int audiossl_conv1_finalize(double* mom_totals, const float* w, const float* bias, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps, double count,
                                       float* scale, float* shift, float* save_mean, float* save_rstd, void* stream) {
    ASSL_REQUIRE(mom_totals && w && bias && gamma && beta && scale && shift && save_mean && save_rstd && count > 1.0);
    hipLaunchKernelGGL(conv1_finalize_kernel, dim3(1), dim3(576), 0, static_cast<hipStream_t>(stream), mom_totals, w, bias, gamma, beta,
                       running_mean, running_var, momentum, eps, count, scale, shift, save_mean, save_rstd, 1);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_conv1_fwd(int dtype, const float* img, int N, int F, int T, const float* w, const float* bias,
                                  const float* scale, const float* shift, void* out, float* xl, void* stream) {
    ASSL_REQUIRE(img && w && bias && scale && shift && out && N > 0 && F >= 2 && F <= 126 && T >= 2 && (F % 2) == 0);
    ASSL_REQUIRE(dtype == 0 || dtype == 1 || dtype == 2);
    ASSL_REQUIRE(!xl || dtype == 1);                              // the fused layer mean exists for the MFMA kernel only
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int xs_split = 4;                                       // AUDIOSSL_CONV1_XL_PARTS in include/audiossl_hip.h
    const int grid = xl ? N * xs_split : min(N * (T / 2), 2048);
    size_t lds = sizeof(float) * (F + 2) * 4;
    if (dtype == 1) lds = sizeof(float) * (((F + 2) * 4 + 3) & ~3) + sizeof(bf16) * (F / 2) * 64;     // + the output staging tile
    if (dtype == 0)
        hipLaunchKernelGGL(conv1_fwd_kernel<float>, dim3(grid), dim3(256), lds, s, img, w, bias, scale, shift,
                           static_cast<float*>(out), N, F, T);
    else if (dtype == 2)
        hipLaunchKernelGGL(conv1_fwd_kernel<bf16>, dim3(grid), dim3(256), lds, s, img, w, bias, scale, shift,
                           static_cast<bf16*>(out), N, F, T);
    else
        hipLaunchKernelGGL(conv1_fwd_mfma_kernel, dim3(grid), dim3(256), lds, s, img, w, bias, scale, shift,
                           static_cast<bf16*>(out), xl, xs_split, N, F, T);
    ASSL_LAUNCH_CHECK();
}

// acc: 32*64*11 floats of scratch (zeroed here).  dxl may be null.  Grad outputs are accumulated (+=).
static int conv1_bwd_main(int dtype, int conv_dtype, const float* img, int N, int F, int T, const float* w, const float* bias,
                                  const float* gamma, const float* scale, const float* shift, const float* mean,
                                  const float* rstd, const double* mom, const void* dP, const void* dxl, float* acc,
                                  void* stream) {
    ASSL_REQUIRE(img && w && bias && gamma && scale && shift && mean && rstd && mom && dP && acc);
    ASSL_REQUIRE(N > 0 && F >= 2 && F <= 64 && T >= 2 && (F % 2) == 0 && (dtype == 0 || dtype == 1));
    // conv_dtype 1 (MFMA recompute): dtype is the type of dP alone (fp32 or bf16), dxl is fp32 either way
    hipStream_t s = static_cast<hipStream_t>(stream);
    ASSL_ZERO(acc, sizeof(float) * 32 * 64 * 11, s);
    static const int bwd_grid = getenv("AUDIOSSL_CONV1_BWD_GRID") ? atoi(getenv("AUDIOSSL_CONV1_BWD_GRID")) : 2048;
    const int grid = min(N * (T / 2), bwd_grid);
    const size_t lds = sizeof(float) * (F + 2) * 4;
    const float inv_To = 1.f / (float)(T / 2);
    if (conv_dtype == 1 && F == 64 && dtype == 0)
        hipLaunchKernelGGL((conv1_bwd_mfma_kernel<true, float>), dim3(grid), dim3(256), lds + 2 * 2048 * sizeof(float), s, img, w, bias, scale,
                           shift, mean, rstd, static_cast<const float*>(dP), static_cast<const float*>(dxl), inv_To, acc, N, F, T);
    else if (conv_dtype == 1 && F == 64)
        hipLaunchKernelGGL((conv1_bwd_mfma_kernel<true, bf16>), dim3(grid), dim3(256), lds + 2 * 2048 * sizeof(float), s, img, w, bias, scale,
                           shift, mean, rstd, static_cast<const bf16*>(dP), static_cast<const float*>(dxl), inv_To, acc, N, F, T);
    else if (conv_dtype == 1 && dtype == 0)
        hipLaunchKernelGGL((conv1_bwd_mfma_kernel<false, float>), dim3(grid), dim3(256), lds, s, img, w, bias, scale, shift, mean, rstd,
                           static_cast<const float*>(dP), static_cast<const float*>(dxl), inv_To, acc, N, F, T);
    else if (conv_dtype == 1)
        hipLaunchKernelGGL((conv1_bwd_mfma_kernel<false, bf16>), dim3(grid), dim3(256), lds, s, img, w, bias, scale, shift, mean, rstd,
                           static_cast<const bf16*>(dP), static_cast<const float*>(dxl), inv_To, acc, N, F, T);
    else if (dtype == 0)
        hipLaunchKernelGGL(conv1_bwd_kernel<float>, dim3(grid), dim3(256), lds, s, img, w, bias, scale, shift, mean, rstd,
                           static_cast<const float*>(dP), static_cast<const float*>(dxl), inv_To, acc, N, F, T);
    else
        hipLaunchKernelGGL(conv1_bwd_kernel<bf16>, dim3(grid), dim3(256), lds, s, img, w, bias, scale, shift, mean, rstd,
                           static_cast<const bf16*>(dP), static_cast<const bf16*>(dxl), inv_To, acc, N, F, T);
    return ASSL_OK;
}

extern "C" int audiossl_conv1_bwd(int dtype, int conv_dtype, const float* img, int N, int F, int T, const float* w, const float* bias,
                                  const float* gamma, const float* scale, const float* shift, const float* mean,
                                  const float* rstd, const double* mom, const void* dP, const void* dxl, float* acc,
                                  float* dW, float* dbias, float* dgamma, float* dbeta, void* stream) {
    ASSL_REQUIRE(dW && dgamma && dbeta);
    const int rc = conv1_bwd_main(dtype, conv_dtype, img, N, F, T, w, bias, gamma, scale, shift, mean, rstd, mom, dP, dxl, acc, stream);
    if (rc != ASSL_OK) return rc;
    hipLaunchKernelGGL(conv1_bwd_finalize_kernel, dim3(1), dim3(704), 0, static_cast<hipStream_t>(stream), acc, mom, w, bias, gamma, mean,
                       rstd, (double)N * F * T, dW, dbias, dgamma, dbeta, (const float*)nullptr, (float*)nullptr);
    ASSL_LAUNCH_CHECK();
}

// SyncBatchNorm halves: conv1_bwd_sums = the gradient kernel + this rank's (sum g, sum g xhat) per channel in lstat [2][64];
// conv1_bwd_finalize = weight / BatchNorm gradients from the rank's own sums with the GLOBAL means (gstat = all-reduced lstat).
extern "C" int audiossl_conv1_bwd_sums(int dtype, int conv_dtype, const float* img, int N, int F, int T, const float* w,
                                       const float* bias, const float* gamma, const float* scale, const float* shift,
                                       const float* mean, const float* rstd, const double* mom, const void* dP, const void* dxl,
                                       float* acc, float* lstat, void* stream) {
    ASSL_REQUIRE(lstat);
    const int rc = conv1_bwd_main(dtype, conv_dtype, img, N, F, T, w, bias, gamma, scale, shift, mean, rstd, mom, dP, dxl, acc, stream);
    if (rc != ASSL_OK) return rc;
    hipLaunchKernelGGL(conv1_bwd_finalize_kernel, dim3(1), dim3(704), 0, static_cast<hipStream_t>(stream), acc, mom, w, bias, gamma, mean,
                       rstd, 1.0, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, lstat);
    ASSL_LAUNCH_CHECK();
}
extern "C" int audiossl_conv1_bwd_finalize(const float* acc, const double* mom, const float* w, const float* bias, const float* gamma,
                                           const float* mean, const float* rstd, double count_global, const float* gstat, float* dW,
                                           float* dbias, float* dgamma, float* dbeta, void* stream) {
    ASSL_REQUIRE(acc && mom && w && bias && gamma && mean && rstd && gstat && dW && dgamma && dbeta && count_global > 1.0);
    hipLaunchKernelGGL(conv1_bwd_finalize_kernel, dim3(1), dim3(704), 0, static_cast<hipStream_t>(stream), acc, mom, w, bias, gamma, mean,
                       rstd, count_global, dW, dbias, dgamma, dbeta, gstat, (float*)nullptr);
    ASSL_LAUNCH_CHECK();
}
