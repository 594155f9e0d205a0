// MViTv2 pooling attention on gfx950: the pieces of `MultiScaleAttention` / `MultiScaleBlock`
// (extras/mast_new/mast/mvit/models/attention.py:93-302, 304-393; the encoder `models_msn.py:147` -> ASTModel(model_size='mvit')
// stacks) that the plain ViT block does not have.  The GEMMs (qkv, proj, MLP, width-changing projections) are gemm.hip,
// LayerNorm over the model width and GELU are vit.hip; here:
//   pool_fwd / pool_bwd_a / pool_bwd_b : the Q / K / V pooling - depthwise 3x3 convolution over the token grid, one filter bank
//                                        of head_dim channels shared by all heads (`attention_pool`, :12-41, mode "conv"), followed
//                                        by LayerNorm over head_dim - and the head split itself when a path is not pooled
//   attn_rel_fwd / attn_rel_bwd        : softmax(scale q k^T + rel_h + rel_w) v with the decomposed relative-position terms
//                                        (`cal_rel_pos_spatial`, :44-90: q . R_h[dh(query row, key row)] + q . R_w[...], UNSCALED q)
//                                        and the residual pooling connection (`x = x + q`, :283-288) folded in
//   tokpool_max_fwd / _bwd             : MaxPool2d(stride + 1, stride, pad) of the skip path when the query stride is > 1 (:343-350)
// Shapes of this encoder are small where these kernels work - 108 tokens per second of audio, keys pooled to 3 x 3 per head in
// MViTv2-B (adaptive kv stride), head_dim 96 - so they are plain fp32 VALU kernels organised for coalesced traffic (a wave per
// token row / four lanes per query); the MFMA work of the block is in its GEMMs.
#include "common.h"

namespace {

constexpr int MAXD = 128;                 // head_dim <= 128 (two channels per lane in the pooling kernels)

struct PoolArgs {
    const bf16* qkv;                      // [B*L][ldq] (q | k | v column blocks of width att = heads * d)
    int ldq, col0;                        // leading dimension, first column of the pooled tensor's block (which * att)
    const float* w;                       // [d][3][3] depthwise filters or null (no pooling: head split only)
    const float* gamma; const float* beta;
    float* out;                           // [B][heads][Lo][d] fp32
    float* z;                             // pre-LayerNorm conv output, same shape (kept for the backward) or null
    float* mean; float* rstd;             // [B*heads*Lo]
    int B, heads, d, H, W, Ho, Wo, sh, sw;
    float eps;
};

// grid = B * heads workgroups, 4 waves each walking the output positions; lane = channel (and channel + 64)
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs a) {
    const int bh = blockIdx.x, b = bh / a.heads, h = bh % a.heads;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < a.d, v1 = c1 < a.d;
    float w0[9], w1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        w0[t] = (a.w && v0) ? a.w[c0 * 9 + t] : 0.f;
        w1[t] = (a.w && v1) ? a.w[c1 * 9 + t] : 0.f;
    }
    const float g0 = (a.w && v0) ? a.gamma[c0] : 0.f, g1 = (a.w && v1) ? a.gamma[c1] : 0.f;
    const float e0 = (a.w && v0) ? a.beta[c0] : 0.f, e1 = (a.w && v1) ? a.beta[c1] : 0.f;
    const bf16* src = a.qkv + (long)b * a.H * a.W * a.ldq + a.col0 + h * a.d;
    const int Lo = a.Ho * a.Wo;
    for (int po = wave; po < Lo; po += 4) {
        const int ho = po / a.Wo, wo = po % a.Wo;
        float z0 = 0.f, z1 = 0.f;
        if (a.w) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hi = ho * a.sh - 1 + kh;
                if (hi < 0 || hi >= a.H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int wi = wo * a.sw - 1 + kw;
                    if (wi < 0 || wi >= a.W) continue;
                    const bf16* p = src + (long)(hi * a.W + wi) * a.ldq;
                    if (v0) z0 += w0[kh * 3 + kw] * (float)p[c0];
                    if (v1) z1 += w1[kh * 3 + kw] * (float)p[c1];
                }
            }
        } else {
            const bf16* p = src + (long)po * a.ldq;
            if (v0) z0 = (float)p[c0];
            if (v1) z1 = (float)p[c1];
        }
        const long o = ((long)bh * Lo + po) * a.d;
        if (!a.w) {
            if (v0) a.out[o + c0] = z0;
            if (v1) a.out[o + c1] = z1;
            continue;
        }
        const float mu = wave_sum(z0 + z1) / (float)a.d;
        const float d0 = v0 ? z0 - mu : 0.f, d1 = v1 ? z1 - mu : 0.f;
        const float rs = rsqrtf(wave_sum(d0 * d0 + d1 * d1) / (float)a.d + a.eps);
        if (v0) { a.out[o + c0] = d0 * rs * g0 + e0; if (a.z) a.z[o + c0] = z0; }
        if (v1) { a.out[o + c1] = d1 * rs * g1 + e1; if (a.z) a.z[o + c1] = z1; }
        if (lane == 0 && a.mean) { a.mean[(long)bh * Lo + po] = mu; a.rstd[(long)bh * Lo + po] = rs; }
    }
}

struct PoolBwdArgs {
    const bf16* qkv; int ldq, col0;
    const float* w; const float* gamma;
    const float* dout;                    // [B][heads][Lo][d] gradient of the pooled + normalised tensor
    const float* z; const float* mean; const float* rstd;
    float* dz;                            // [B][heads][Lo][d] scratch: gradient of the conv output
    float* dw; float* dgamma; float* dbeta;        // accumulated (atomics)
    bf16* dqkv;                           // [B*L][ldq]: the tensor's column block is WRITTEN (every element has one writer)
    int B, heads, d, H, W, Ho, Wo, sh, sw;
};

// (a) LayerNorm backward per output row -> dz, and the parameter gradients: dgamma, dbeta, dw[c][tap] = sum dz[c] * in(tap)[c]
__global__ __launch_bounds__(256) void pool_bwd_a_kernel(PoolBwdArgs a) {
    __shared__ float red[4][11][MAXD];
    const int bh = blockIdx.x, b = bh / a.heads, h = bh % a.heads;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < a.d, v1 = c1 < a.d;
    const float g0 = v0 ? a.gamma[c0] : 0.f, g1 = v1 ? a.gamma[c1] : 0.f;
    float dw0[9], dw1[9], dg0 = 0.f, dg1 = 0.f, db0 = 0.f, db1 = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) { dw0[t] = 0.f; dw1[t] = 0.f; }
    const bf16* src = a.qkv + (long)b * a.H * a.W * a.ldq + a.col0 + h * a.d;
    const int Lo = a.Ho * a.Wo;
    for (int po = wave; po < Lo; po += 4) {
        const int ho = po / a.Wo, wo = po % a.Wo;
        const long o = ((long)bh * Lo + po) * a.d;
        const float mu = a.mean[(long)bh * Lo + po], rs = a.rstd[(long)bh * Lo + po];
        const float dy0 = v0 ? a.dout[o + c0] : 0.f, dy1 = v1 ? a.dout[o + c1] : 0.f;
        const float x0 = v0 ? (a.z[o + c0] - mu) * rs : 0.f, x1 = v1 ? (a.z[o + c1] - mu) * rs : 0.f;
        const float q0 = dy0 * g0, q1 = dy1 * g1;
        const float s1 = wave_sum(q0 + q1) / (float)a.d, s2 = wave_sum(q0 * x0 + q1 * x1) / (float)a.d;
        const float z0 = rs * (q0 - s1 - x0 * s2), z1 = rs * (q1 - s1 - x1 * s2);
        if (v0) a.dz[o + c0] = z0;
        if (v1) a.dz[o + c1] = z1;
        dg0 += dy0 * x0; dg1 += dy1 * x1; db0 += dy0; db1 += dy1;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * a.sh - 1 + kh;
            if (hi < 0 || hi >= a.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * a.sw - 1 + kw;
                if (wi < 0 || wi >= a.W) continue;
                const bf16* p = src + (long)(hi * a.W + wi) * a.ldq;
                if (v0) dw0[kh * 3 + kw] += z0 * (float)p[c0];
                if (v1) dw1[kh * 3 + kw] += z1 * (float)p[c1];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) { red[wave][t][c0] = dw0[t]; red[wave][t][c1] = dw1[t]; }
    red[wave][9][c0] = dg0; red[wave][9][c1] = dg1; red[wave][10][c0] = db0; red[wave][10][c1] = db1;
    __syncthreads();
    for (int i = threadIdx.x; i < 11 * a.d; i += 256) {
        const int t = i / a.d, c = i % a.d;
        const float v = red[0][t][c] + red[1][t][c] + red[2][t][c] + red[3][t][c];
        if (t < 9) atomicAdd(a.dw + c * 9 + t, v);
        else if (t == 9) atomicAdd(a.dgamma + c, v);
        else atomicAdd(a.dbeta + c, v);
    }
}

// (b) gradient of the conv INPUT, gathered per input token (no atomics): din[c] = sum over the outputs whose 3x3 window covers
// the token of w[c][tap] * dz[output][c]; without pooling (w == null) the head split's transpose back, from `dout`
__global__ __launch_bounds__(256) void pool_bwd_b_kernel(PoolBwdArgs a) {
    const int bh = blockIdx.x, b = bh / a.heads, h = bh % a.heads;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = lane, c1 = lane + 64;
    const bool v0 = c0 < a.d, v1 = c1 < a.d;
    float w0[9], w1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        w0[t] = (a.w && v0) ? a.w[c0 * 9 + t] : 0.f;
        w1[t] = (a.w && v1) ? a.w[c1 * 9 + t] : 0.f;
    }
    bf16* dst = a.dqkv + (long)b * a.H * a.W * a.ldq + a.col0 + h * a.d;
    const int L = a.H * a.W, Lo = a.Ho * a.Wo;
    const float* dzb = (a.w ? a.dz : a.dout) + (long)bh * Lo * a.d;
    for (int pi = wave; pi < L; pi += 4) {
        float r0 = 0.f, r1 = 0.f;
        if (a.w) {
            const int hi = pi / a.W, wi = pi % a.W;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int th = hi + 1 - kh;
                if (th < 0 || th % a.sh) continue;
                const int ho = th / a.sh;
                if (ho >= a.Ho) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tw = wi + 1 - kw;
                    if (tw < 0 || tw % a.sw) continue;
                    const int wo = tw / a.sw;
                    if (wo >= a.Wo) continue;
                    const float* p = dzb + (long)(ho * a.Wo + wo) * a.d;
                    if (v0) r0 += w0[kh * 3 + kw] * p[c0];
                    if (v1) r1 += w1[kh * 3 + kw] * p[c1];
                }
            }
        } else {
            const float* p = dzb + (long)pi * a.d;
            if (v0) r0 = p[c0];
            if (v1) r1 = p[c1];
        }
        bf16* o = dst + (long)pi * a.ldq;
        if (v0) o[c0] = (bf16)r0;
        if (v1) o[c1] = (bf16)r1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct AttnRelArgs {
    const float* q; const float* k; const float* v;     // [B][heads][Lq | Lk][d] fp32
    const float* rh; const float* rw;                   // [nrh][d], [nrw][d] relative-position tables or null
    const int* ih; const int* iw;                       // [qh][kh], [qw][kw] row of the table for (query coord, key coord)
    bf16* out;                                          // [B*Lq][heads*d]
    float* lse;                                         // [B][heads][Lq]
    const bf16* dout;                                   // backward: [B*Lq][heads*d]
    float* dq; float* dk; float* dv;                    // backward: dq written, dk / dv accumulated (atomics; zeroed by the caller)
    float* drh; float* drw;                             // backward: accumulated
    int B, heads, d, Lq, Lk, qh, qw, kh, kw, nrh, nrw, residual;
    float scale;
};

__device__ __forceinline__ float sum4(float v) {                  // over the four lanes of a query
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}

// workgroup = 64 queries of one (clip, head); four lanes per query, DPL = d / 4 channels each.  Dynamic LDS:
//   bias_h [64][kh], bias_w [64][kw]   (q . R rows of this query's coordinates, computed once)
template <int DPL>
__global__ __launch_bounds__(256) void attn_rel_fwd_kernel(AttnRelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sbh = sm;
    float* sbw = sm + 64 * a.kh;
    const int bh = blockIdx.y, ql = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int qi = min(blockIdx.x * 64 + ql, a.Lq - 1);
    const bool live = blockIdx.x * 64 + ql < a.Lq;
    const int c0 = part * DPL;
    const float* qp = a.q + ((long)bh * a.Lq + qi) * a.d + c0;
    float q[DPL], acc[DPL];
#pragma unroll
    for (int c = 0; c < DPL; ++c) { q[c] = qp[c]; acc[c] = 0.f; }
    const int qhi = qi / a.qw, qwi = qi % a.qw;
    if (a.rh) {
        for (int r = 0; r < a.kh; ++r) {
            const float* t = a.rh + (long)a.ih[qhi * a.kh + r] * a.d + c0;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DPL; ++c) s += q[c] * t[c];
            s = sum4(s);
            if (part == 0) sbh[ql * a.kh + r] = s;
        }
        for (int r = 0; r < a.kw; ++r) {
            const float* t = a.rw + (long)a.iw[qwi * a.kw + r] * a.d + c0;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DPL; ++c) s += q[c] * t[c];
            s = sum4(s);
            if (part == 0) sbw[ql * a.kw + r] = s;
        }
    }
    __syncthreads();
    const float* kb = a.k + (long)bh * a.Lk * a.d + c0;
    const float* vb = a.v + (long)bh * a.Lk * a.d + c0;
    float m = -3.0e38f, l = 0.f;
    int jh = 0, jw = 0;
    for (int j = 0; j < a.Lk; ++j) {
        const float* kp = kb + (long)j * a.d;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DPL; ++c) s += q[c] * kp[c];
        s = sum4(s) * a.scale;
        if (a.rh) s += sbh[ql * a.kh + jh] + sbw[ql * a.kw + jw];
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);
        l = l * corr + p;
        const float* vp = vb + (long)j * a.d;
#pragma unroll
        for (int c = 0; c < DPL; ++c) acc[c] = acc[c] * corr + p * vp[c];
        m = mn;
        if (++jw == a.kw) { jw = 0; ++jh; }
    }
    if (!live) return;
    const float inv = 1.f / l;
    const int b = bh / a.heads, h = bh % a.heads;
    bf16* o = a.out + ((long)b * a.Lq + qi) * (a.heads * a.d) + h * a.d + c0;
#pragma unroll
    for (int c = 0; c < DPL; ++c) o[c] = (bf16)(acc[c] * inv + (a.residual ? q[c] : 0.f));
    if (part == 0) a.lse[(long)bh * a.Lq + qi] = m + __logf(l);
}

// Backward.  Dynamic LDS: bias_h, bias_w as above; dbias_h [64][kh], dbias_w [64][kw] (sums of dS over key columns / rows);
// dK, dV [Lk][d]; dR_h [nrh][d], dR_w [nrw][d] - per-workgroup accumulators (LDS atomics), added to memory once at the end.
// D = sum_j P_j dP_j is taken in a first sweep over the keys (equal to rowsum(dO * O_attention), and needs no copy of the
// attention output without the residual-pooling term).
template <int DPL>
__global__ __launch_bounds__(256) void attn_rel_bwd_kernel(AttnRelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sbh = sm;
    float* sbw = sbh + 64 * a.kh;
    float* sdh = sbw + 64 * a.kw;
    float* sdw = sdh + 64 * a.kh;
    float* sdk = sdw + 64 * a.kw;
    float* sdv = sdk + a.Lk * a.d;
    float* srh = sdv + a.Lk * a.d;
    float* srw = srh + a.nrh * a.d;
    const int nz = 2 * a.Lk * a.d + (a.nrh + a.nrw) * a.d;
    for (int i = threadIdx.x; i < nz; i += 256) sdk[i] = 0.f;
    const int bh = blockIdx.y, ql = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int qi = min(blockIdx.x * 64 + ql, a.Lq - 1);
    const bool live = blockIdx.x * 64 + ql < a.Lq;
    const int c0 = part * DPL;
    const int b = bh / a.heads, h = bh % a.heads;
    const float* qp = a.q + ((long)bh * a.Lq + qi) * a.d + c0;
    const bf16* dop = a.dout + ((long)b * a.Lq + qi) * (a.heads * a.d) + h * a.d + c0;
    float q[DPL], g[DPL], dq[DPL];
#pragma unroll
    for (int c = 0; c < DPL; ++c) { q[c] = qp[c]; g[c] = live ? (float)dop[c] : 0.f; dq[c] = a.residual ? g[c] : 0.f; }
    const int qhi = qi / a.qw, qwi = qi % a.qw;
    if (a.rh) {
        for (int r = 0; r < a.kh; ++r) {
            const float* t = a.rh + (long)a.ih[qhi * a.kh + r] * a.d + c0;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DPL; ++c) s += q[c] * t[c];
            s = sum4(s);
            if (part == 0) { sbh[ql * a.kh + r] = s; sdh[ql * a.kh + r] = 0.f; }
        }
        for (int r = 0; r < a.kw; ++r) {
            const float* t = a.rw + (long)a.iw[qwi * a.kw + r] * a.d + c0;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DPL; ++c) s += q[c] * t[c];
            s = sum4(s);
            if (part == 0) { sbw[ql * a.kw + r] = s; sdw[ql * a.kw + r] = 0.f; }
        }
    }
    __syncthreads();
    const float* kb = a.k + (long)bh * a.Lk * a.d + c0;
    const float* vb = a.v + (long)bh * a.Lk * a.d + c0;
    const float lse = a.lse[(long)bh * a.Lq + qi];
    float D = 0.f;
    int jh = 0, jw = 0;
    for (int j = 0; j < a.Lk; ++j) {
        const float* kp = kb + (long)j * a.d;
        const float* vp = vb + (long)j * a.d;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < DPL; ++c) { s += q[c] * kp[c]; dp += g[c] * vp[c]; }
        s = sum4(s) * a.scale;
        dp = sum4(dp);
        if (a.rh) s += sbh[ql * a.kh + jh] + sbw[ql * a.kw + jw];
        D += __expf(s - lse) * dp;
        if (++jw == a.kw) { jw = 0; ++jh; }
    }
    jh = 0; jw = 0;
    for (int j = 0; j < a.Lk; ++j) {
        const float* kp = kb + (long)j * a.d;
        const float* vp = vb + (long)j * a.d;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < DPL; ++c) { s += q[c] * kp[c]; dp += g[c] * vp[c]; }
        s = sum4(s) * a.scale;
        dp = sum4(dp);
        if (a.rh) s += sbh[ql * a.kh + jh] + sbw[ql * a.kw + jw];
        const float p = live ? __expf(s - lse) : 0.f;
        const float ds = p * (dp - D);
        const float dss = ds * a.scale;
#pragma unroll
        for (int c = 0; c < DPL; ++c) {
            dq[c] += dss * kp[c];
            atomicAdd(sdk + j * a.d + c0 + c, dss * q[c]);
            atomicAdd(sdv + j * a.d + c0 + c, p * g[c]);
        }
        if (a.rh && part == 0) { sdh[ql * a.kh + jh] += ds; sdw[ql * a.kw + jw] += ds; }
        if (++jw == a.kw) { jw = 0; ++jh; }
    }
    if (a.rh) {
        __syncthreads();                                          // the query's dbias sums are complete (written by its lane 0)
        for (int r = 0; r < a.kh; ++r) {
            const int row = a.ih[qhi * a.kh + r];
            const float db = live ? sdh[ql * a.kh + r] : 0.f;
            const float* t = a.rh + (long)row * a.d + c0;
#pragma unroll
            for (int c = 0; c < DPL; ++c) { dq[c] += db * t[c]; atomicAdd(srh + row * a.d + c0 + c, db * q[c]); }
        }
        for (int r = 0; r < a.kw; ++r) {
            const int row = a.iw[qwi * a.kw + r];
            const float db = live ? sdw[ql * a.kw + r] : 0.f;
            const float* t = a.rw + (long)row * a.d + c0;
#pragma unroll
            for (int c = 0; c < DPL; ++c) { dq[c] += db * t[c]; atomicAdd(srw + row * a.d + c0 + c, db * q[c]); }
        }
    }
    if (live) {
        float* o = a.dq + ((long)bh * a.Lq + qi) * a.d + c0;
#pragma unroll
        for (int c = 0; c < DPL; ++c) o[c] = dq[c];
    }
    __syncthreads();
    float* dkb = a.dk + (long)bh * a.Lk * a.d;
    float* dvb = a.dv + (long)bh * a.Lk * a.d;
    for (int i = threadIdx.x; i < a.Lk * a.d; i += 256) { atomicAdd(dkb + i, sdk[i]); atomicAdd(dvb + i, sdv[i]); }
    if (a.rh) {
        for (int i = threadIdx.x; i < a.nrh * a.d; i += 256) atomicAdd(a.drh + i, srh[i]);
        for (int i = threadIdx.x; i < a.nrw * a.d; i += 256) atomicAdd(a.drw + i, srw[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// MaxPool2d(kernel (kh, kw), stride (sh, sw), padding (kh / 2, kw / 2)) over the token grid of x [B][H*W][C] fp32; arg = tap index
// of the (first) maximum, as torch reports it.  Thread = (output token, 4 consecutive channels).
__global__ __launch_bounds__(256) void tokpool_max_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ arg,
                                                              int B, int H, int W, int C, int Ho, int Wo, int kh, int kw, int sh, int sw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (i >= (long)B * Ho * Wo * c4) return;
    const int c = (int)(i % c4) * 4;
    const long tok = i / c4;
    const int wo = (int)(tok % Wo), ho = (int)((tok / Wo) % Ho), b = (int)(tok / ((long)Wo * Ho));
    f32x4 best = f32x4{-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
    int at[4] = {0, 0, 0, 0};
    for (int a = 0; a < kh; ++a) {
        const int hi = ho * sh - kh / 2 + a;
        if (hi < 0 || hi >= H) continue;
        for (int e = 0; e < kw; ++e) {
            const int wi = wo * sw - kw / 2 + e;
            if (wi < 0 || wi >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((long)b * H * W + hi * W + wi) * C + c);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (v[u] > best[u]) { best[u] = v[u]; at[u] = a * kw + e; }
        }
    }
    *reinterpret_cast<f32x4*>(y + tok * C + c) = best;
    *reinterpret_cast<uint32_t*>(arg + tok * C + c) = (uint32_t)at[0] | ((uint32_t)at[1] << 8) | ((uint32_t)at[2] << 16) | ((uint32_t)at[3] << 24);
}

// gradient gathered per INPUT token: dx = sum over the outputs whose window covers the token and whose arg-max is this tap
__global__ __launch_bounds__(256) void tokpool_max_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ arg, float* __restrict__ dx,
                                                              int B, int H, int W, int C, int Ho, int Wo, int kh, int kw, int sh, int sw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (i >= (long)B * H * W * c4) return;
    const int c = (int)(i % c4) * 4;
    const long tok = i / c4;
    const int wi = (int)(tok % W), hi = (int)((tok / W) % H), b = (int)(tok / ((long)W * H));
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < kh; ++a) {
        const int th = hi + kh / 2 - a;
        if (th < 0 || th % sh) continue;
        const int ho = th / sh;
        if (ho >= Ho) continue;
        for (int e = 0; e < kw; ++e) {
            const int tw = wi + kw / 2 - e;
            if (tw < 0 || tw % sw) continue;
            const int wo = tw / sw;
            if (wo >= Wo) continue;
            const long o = ((long)b * Ho * Wo + ho * Wo + wo) * C + c;
            const uint32_t at = *reinterpret_cast<const uint32_t*>(arg + o);
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (((at >> (8 * u)) & 0xFFu) == (uint32_t)(a * kw + e)) r[u] += g[u];
        }
    }
    *reinterpret_cast<f32x4*>(dx + tok * C + c) = r;
}

}  // namespace

#define S_(stream) static_cast<hipStream_t>(stream)

static bool pool_shape_ok(int B, int heads, int d, int H, int W, int Ho, int Wo, int sh, int sw, bool pooled) {
    if (B <= 0 || heads <= 0 || d <= 0 || d > MAXD || H <= 0 || W <= 0 || sh <= 0 || sw <= 0) return false;
    if (!pooled) return Ho == H && Wo == W;
    return Ho == (H - 1) / sh + 1 && Wo == (W - 1) / sw + 1;          // floor((H + 2 - 3) / s) + 1
}

extern "C" int audiossl_mvit_pool_fwd(const void* qkv, int ldq, int col0, const float* w, const float* gamma, const float* beta, float* out,
                                      float* z, float* mean, float* rstd, int B, int heads, int d, int H, int W, int Ho, int Wo, int sh,
                                      int sw, float eps, void* stream) {
    ASSL_REQUIRE(qkv && out && ldq > 0 && col0 >= 0 && pool_shape_ok(B, heads, d, H, W, Ho, Wo, sh, sw, w != nullptr));
    ASSL_REQUIRE(!w || (gamma && beta && z && mean && rstd));
    PoolArgs a{static_cast<const bf16*>(qkv), ldq, col0, w, gamma, beta, out, z, mean, rstd, B, heads, d, H, W, Ho, Wo, sh, sw, eps};
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(B * heads), dim3(256), 0, S_(stream), a);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_mvit_pool_bwd(const void* qkv, int ldq, int col0, const float* w, const float* gamma, const float* dout, const float* z,
                                      const float* mean, const float* rstd, float* dz, float* dw, float* dgamma, float* dbeta, void* dqkv,
                                      int B, int heads, int d, int H, int W, int Ho, int Wo, int sh, int sw, void* stream) {
    ASSL_REQUIRE(qkv && dout && dqkv && ldq > 0 && col0 >= 0 && pool_shape_ok(B, heads, d, H, W, Ho, Wo, sh, sw, w != nullptr));
    ASSL_REQUIRE(!w || (gamma && z && mean && rstd && dz && dw && dgamma && dbeta));
    PoolBwdArgs a{static_cast<const bf16*>(qkv), ldq, col0, w, gamma, dout, z, mean, rstd, dz, dw, dgamma, dbeta, static_cast<bf16*>(dqkv),
                  B, heads, d, H, W, Ho, Wo, sh, sw};
    if (w) hipLaunchKernelGGL(pool_bwd_a_kernel, dim3(B * heads), dim3(256), 0, S_(stream), a);
    hipLaunchKernelGGL(pool_bwd_b_kernel, dim3(B * heads), dim3(256), 0, S_(stream), a);
    ASSL_LAUNCH_CHECK();
}

static bool attn_rel_ok(const AttnRelArgs& a) {
    if (!a.q || !a.k || !a.v || !a.lse || a.B <= 0 || a.heads <= 0 || a.Lq <= 0 || a.Lk <= 0) return false;
    if (a.d != 64 && a.d != 96 && a.d != 128) return false;
    if (a.qh * a.qw != a.Lq || a.kh * a.kw != a.Lk) return false;
    if ((a.rh != nullptr) != (a.rw != nullptr)) return false;
    if (a.rh && (!a.ih || !a.iw || a.nrh <= 0 || a.nrw <= 0)) return false;
    return (long)a.B * a.heads <= 65535;
}

extern "C" int audiossl_mvit_attn_fwd(const float* q, const float* k, const float* v, const float* rh, const float* rw, const int* ih,
                                      const int* iw, void* out, float* lse, int B, int heads, int d, int qh, int qw, int kh, int kw, int nrh,
                                      int nrw, int residual, float scale, void* stream) {
    AttnRelArgs a{q, k, v, rh, rw, ih, iw, static_cast<bf16*>(out), lse, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                  B, heads, d, qh * qw, kh * kw, qh, qw, kh, kw, nrh, nrw, residual, scale};
    ASSL_REQUIRE(out && attn_rel_ok(a));
    const size_t lds = sizeof(float) * 64 * (size_t)(kh + kw);
    ASSL_REQUIRE(lds <= 64 * 1024);
    const dim3 grid(ceil_div(a.Lq, 64), B * heads);
    if (d == 64) hipLaunchKernelGGL((attn_rel_fwd_kernel<16>), grid, dim3(256), lds, S_(stream), a);
    else if (d == 96) hipLaunchKernelGGL((attn_rel_fwd_kernel<24>), grid, dim3(256), lds, S_(stream), a);
    else hipLaunchKernelGGL((attn_rel_fwd_kernel<32>), grid, dim3(256), lds, S_(stream), a);
    ASSL_LAUNCH_CHECK();
}

template <int DPL>
static int launch_attn_rel_bwd(const AttnRelArgs& a, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_rel_bwd_kernel<DPL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_rel_bwd_kernel<DPL>), dim3(ceil_div(a.Lq, 64), a.B * a.heads), dim3(256), lds, s, a);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_mvit_attn_bwd(const float* q, const float* k, const float* v, const float* rh, const float* rw, const int* ih,
                                      const int* iw, const void* dout, const float* lse, float* dq, float* dk, float* dv, float* drh,
                                      float* drw, int B, int heads, int d, int qh, int qw, int kh, int kw, int nrh, int nrw, int residual,
                                      float scale, void* stream) {
    AttnRelArgs a{q, k, v, rh, rw, ih, iw, nullptr, const_cast<float*>(lse), static_cast<const bf16*>(dout), dq, dk, dv, drh, drw,
                  B, heads, d, qh * qw, kh * kw, qh, qw, kh, kw, rh ? nrh : 0, rh ? nrw : 0, residual, scale};
    ASSL_REQUIRE(dout && dq && dk && dv && attn_rel_ok(a) && (!rh || (drh && drw)));
    const size_t lds = sizeof(float) * (128 * (size_t)(kh + kw) + 2 * (size_t)a.Lk * d + (size_t)(a.nrh + a.nrw) * d);
    ASSL_REQUIRE(lds <= 160 * 1024);
    if (d == 64) return launch_attn_rel_bwd<16>(a, lds, S_(stream));
    if (d == 96) return launch_attn_rel_bwd<24>(a, lds, S_(stream));
    return launch_attn_rel_bwd<32>(a, lds, S_(stream));
}

extern "C" int audiossl_tokpool_max_fwd(const float* x, float* y, void* arg, int B, int H, int W, int C, int kh, int kw, int sh, int sw,
                                        void* stream) {
    ASSL_REQUIRE(x && y && arg && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && kh > 0 && kw > 0 && kh * kw <= 255 && sh > 0 && sw > 0);
    if (!ASSL_ALIGNED16(x) || !ASSL_ALIGNED16(y) || ((uintptr_t)arg & 3)) return ASSL_EALIGN;
    const int Ho = (H + 2 * (kh / 2) - kh) / sh + 1, Wo = (W + 2 * (kw / 2) - kw) / sw + 1;
    const long n = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(tokpool_max_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, S_(stream), x, y, static_cast<uint8_t*>(arg), B, H, W,
                       C, Ho, Wo, kh, kw, sh, sw);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_tokpool_max_bwd(const float* dy, const void* arg, float* dx, int B, int H, int W, int C, int kh, int kw, int sh,
                                        int sw, void* stream) {
    ASSL_REQUIRE(dy && dx && arg && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && kh > 0 && kw > 0 && kh * kw <= 255 && sh > 0 && sw > 0);
    if (!ASSL_ALIGNED16(dy) || !ASSL_ALIGNED16(dx) || ((uintptr_t)arg & 3)) return ASSL_EALIGN;
    const int Ho = (H + 2 * (kh / 2) - kh) / sh + 1, Wo = (W + 2 * (kw / 2) - kw) / sw + 1;
    const long n = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(tokpool_max_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, S_(stream), dy, static_cast<const uint8_t*>(arg), dx, B,
                       H, W, C, Ho, Wo, kh, kw, sh, sw);
    ASSL_LAUNCH_CHECK();
}
