// Multi-head self-attention for the AST / MAST encoder (BASELINE config 4: "AST-base 12x768", 12 heads x 64) on gfx950.
//   reference: the ViT attention timm builds for `ASTModel` (`extras/mast_new/mast/models/ast_work.py:70-81, 183-230`):
//   softmax(q k^T / sqrt(64)) v per head, q/k/v = column blocks of one fused Linear(768, 2304).
//
// Spectrogram patch sequences are short (1 s -> 12 x 9 = 108 tokens), so one 256-thread workgroup owns one (clip, head)
// completely: Q, K, V (<= 128 x 64 bf16 each) are staged once in LDS, the 128 x 128 score tile lives in MFMA accumulators,
// nothing but O (and the row log-sum-exp for the backward) goes back to HBM.
//
// Layout trick: the scores are computed TRANSPOSED, S^T = K Q^T, so that a lane owns one query column (col = lane & 31 of
// the 32x32 C/D map) and its keys sit in registers: the softmax reductions are in-register sums plus one exchange with
// lane ^ 32, and the probabilities are already in A-operand position for O = P V.  The key order a lane holds,
// k(j, half) = 16*step + (j & 3) + 8*(j >> 2) + 4*half, is not the hardware's natural k order - the contraction does not
// care as long as the B operand uses the same order, which is why V (and K / Q / dO in the backward) are also kept
// transposed in LDS: the matching 8 values are two aligned 8-byte reads.
//
// Longer sequences (10 s clips: 12 x 101 = 1,212 tokens) use the *_long kernels below: the same 128 x 128 tile, one
// workgroup per (clip, head, block of 128 queries - or keys, for dK / dV) that walks the other axis in blocks of 128 and
// re-stages K / V (Q / dO) per block; the score matrix still never leaves the accumulators.  The forward is two sweeps
// (row log-sum-exp first, then P V with final probabilities - no accumulator rescaling), the backward takes
// D = rowsum(dO * O) from the forward output, as flash attention does.
#include "common.h"

namespace {

constexpr int SP = 128;          // padded sequence length (hard upper bound on S)
constexpr int DH = 64;           // head dimension
constexpr int RP = DH + 8;       // [token][d] image pitch (elements): 144-byte rows, 16-byte aligned fragments
constexpr int TP = SP + 8;       // [d][token] image pitch (elements): 272-byte rows, 8-byte aligned fragments

struct AttnArgs {
    const bf16* qkv;    // [B*S][3*H*64]: q | k | v column blocks
    const bf16* dout;   // [B*S][H*64]           (backward)
    bf16* out;          // [B*S][H*64]           (forward; read by the long-sequence backward for D = rowsum(dO * O))
    bf16* dqkv;         // [B*S][3*H*64]         (backward)
    float* lse;         // [B*H][S] natural-log sum-exp of the scaled scores
    int B, S, H;
    float scale;
};

__device__ __forceinline__ Vec8<bf16> frag_row(const bf16* img, int row, int k0, int half) {
    return Vec8<bf16>::load(img + row * RP + k0 + 8 * half);
}
// the permuted-k fragment out of a transposed image: elements j=0..3 at base, j=4..7 at base + 8
__device__ __forceinline__ Vec8<bf16> frag_perm(const bf16* imgT, int d, int base) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(imgT + d * TP + base);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(imgT + d * TP + base + 8);
    Vec8<bf16> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
    return r;
}
__device__ __forceinline__ void mma(f32x16& acc, const Vec8<bf16>& a, const Vec8<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ int row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }   // C/D row of register r

// stage one [S][64] operand (a column block of qkv or dout) into a row image and/or a transposed image; rows >= S are zero
__device__ __forceinline__ void stage(const bf16* src, long ld, int S, bf16* img, bf16* imgT, int tid) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int idx = tid + 256 * v, row = idx >> 3, c8 = idx & 7;
        const Vec8<bf16> x = row < S ? Vec8<bf16>::load(src + (long)row * ld + c8 * 8) : Vec8<bf16>::zero();
        if (img) x.store(img + row * RP + c8 * 8);
        if (imgT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) imgT[(c8 * 8 + i) * TP + row] = x.v[i];
        }
    }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Qs = reinterpret_cast<bf16*>(smem);
    bf16* Ks = Qs + SP * RP;
    bf16* Vt = Ks + SP * RP;
    float* rs = reinterpret_cast<float*>(Vt + DH * TP);                  // 1 / row sum, [128]
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int S = a.S;
    const long ld = 3L * a.H * DH;
    const bf16* base = a.qkv + (long)b * S * ld + h * DH;
    stage(base, ld, S, Qs, nullptr, tid);
    stage(base + a.H * DH, ld, S, Ks, nullptr, tid);
    stage(base + 2 * a.H * DH, ld, S, nullptr, Vt, tid);
    __syncthreads();

    // S^T tile of this wave: keys (4 tiles of 32) x queries [32w, 32w+32)
    f32x16 acc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
    Vec8<bf16> fq[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) fq[ds] = frag_row(Qs, w * 32 + l31, ds * 16, half);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) mma(acc[kt], frag_row(Ks, kt * 32 + l31, ds * 16, half), fq[ds]);

    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + row_of(r, half);
            const float t = key < S ? acc[kt][r] * a.scale : -3.0e38f;
            acc[kt][r] = t;
            m = fmaxf(m, t);
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(acc[kt][r] - m);          // masked keys: exp(-3e38 - m) = 0
            acc[kt][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const int q = w * 32 + l31;
    if (half == 0) {
        rs[q] = 1.f / sum;
        if (q < S) a.lse[(long)bh * S + q] = m + __logf(sum);
    }

    // O = P V : A = P (row = query = this lane's column), B = V in the permuted key order
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            Vec8<bf16> pa;
#pragma unroll
            for (int j = 0; j < 8; ++j) pa.v[j] = (bf16)acc[kt][8 * st + j];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) mma(o[dt], pa, frag_perm(Vt, dt * 32 + l31, kt * 32 + 16 * st + 4 * half));
        }
    __syncthreads();                                          // rs[] written by every wave's lower half
    bf16* ob = a.out + (long)b * S * (a.H * DH) + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = w * 32 + row_of(r, half);
            if (qq < S) ob[(long)qq * (a.H * DH) + dt * 32 + l31] = (bf16)(o[dt][r] * rs[qq]);
        }
}

// Backward: dQ from the transposed tile (lane = query), dK / dV from the plain tile (lane = key); both recompute the
// probabilities from the saved log-sum-exp.  D[q] = sum_k P dP (= dO . O) comes out of the first pass.
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Qs = reinterpret_cast<bf16*>(smem);
    bf16* Ks = Qs + SP * RP;
    bf16* Vs = Ks + SP * RP;
    bf16* Gs = Vs + SP * RP;                 // dO
    bf16* Qt = Gs + SP * RP;
    bf16* Kt = Qt + DH * TP;
    bf16* Gt = Kt + DH * TP;
    float* Ls = reinterpret_cast<float*>(Gt + DH * TP);
    float* Ds = Ls + SP;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int S = a.S;
    const long ld = 3L * a.H * DH, ldo = (long)a.H * DH;
    const bf16* base = a.qkv + (long)b * S * ld + h * DH;
    stage(base, ld, S, Qs, Qt, tid);
    stage(base + a.H * DH, ld, S, Ks, Kt, tid);
    stage(base + 2 * a.H * DH, ld, S, Vs, nullptr, tid);
    stage(a.dout + (long)b * S * ldo + h * DH, ldo, S, Gs, Gt, tid);
    if (tid < SP) Ls[tid] = tid < S ? a.lse[(long)bh * S + tid] : 0.f;
    __syncthreads();

    f32x16 acc[4], dp[4];
    bf16* dq_out = a.dqkv + (long)b * S * ld + h * DH;
    // ------------------------------------------------------------------ pass 1: lane <-> query, dQ
    {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[kt][r] = 0.f; dp[kt][r] = 0.f; }
        Vec8<bf16> fq[4], fg[4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            fq[ds] = frag_row(Qs, w * 32 + l31, ds * 16, half);
            fg[ds] = frag_row(Gs, w * 32 + l31, ds * 16, half);
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                mma(acc[kt], frag_row(Ks, kt * 32 + l31, ds * 16, half), fq[ds]);      // S^T
                mma(dp[kt], frag_row(Vs, kt * 32 + l31, ds * 16, half), fg[ds]);       // dP^T = V dO^T
            }
        const int q = w * 32 + l31;
        const float Lq = Ls[q];
        float Dq = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + row_of(r, half);
                const float p = (key < S && q < S) ? __expf(acc[kt][r] * a.scale - Lq) : 0.f;
                acc[kt][r] = p;
                Dq += p * dp[kt][r];
            }
        Dq += __shfl_xor(Dq, 32, 64);
        if (half == 0) Ds[q] = Dq;
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                Vec8<bf16> sa;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * st + j;
                    sa.v[j] = (bf16)(acc[kt][r] * (dp[kt][r] - Dq) * a.scale);          // dS[q][key]
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) mma(dq[dt], sa, frag_perm(Kt, dt * 32 + l31, kt * 32 + 16 * st + 4 * half));
            }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = w * 32 + row_of(r, half);
                if (qq < S) dq_out[(long)qq * ld + dt * 32 + l31] = (bf16)dq[dt][r];
            }
    }
    __syncthreads();                                          // Ds[] complete
    // ------------------------------------------------------------------ pass 2: lane <-> key, dK and dV
    {
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[qt][r] = 0.f; dp[qt][r] = 0.f; }
        Vec8<bf16> fk[4], fv[4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            fk[ds] = frag_row(Ks, w * 32 + l31, ds * 16, half);
            fv[ds] = frag_row(Vs, w * 32 + l31, ds * 16, half);
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                mma(acc[qt], frag_row(Qs, qt * 32 + l31, ds * 16, half), fk[ds]);      // S
                mma(dp[qt], frag_row(Gs, qt * 32 + l31, ds * 16, half), fv[ds]);       // dP = dO V^T
            }
        const int key = w * 32 + l31;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + row_of(r, half);
                const float p = (key < S && q < S) ? __expf(acc[qt][r] * a.scale - Ls[q]) : 0.f;
                acc[qt][r] = p;
                dp[qt][r] = p * (dp[qt][r] - Ds[q]) * a.scale;                          // dS[q][key]
            }
        f32x16 dv[2], dk[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                Vec8<bf16> pa, sa;
#pragma unroll
                for (int j = 0; j < 8; ++j) { pa.v[j] = (bf16)acc[qt][8 * st + j]; sa.v[j] = (bf16)dp[qt][8 * st + j]; }
                const int qb = qt * 32 + 16 * st + 4 * half;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    mma(dv[dt], pa, frag_perm(Gt, dt * 32 + l31, qb));                  // dV = P^T dO
                    mma(dk[dt], sa, frag_perm(Qt, dt * 32 + l31, qb));                  // dK = dS^T Q
                }
            }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = w * 32 + row_of(r, half);
                if (kk < S) {
                    dq_out[(long)kk * ld + a.H * DH + dt * 32 + l31] = (bf16)dk[dt][r];
                    dq_out[(long)kk * ld + 2 * a.H * DH + dt * 32 + l31] = (bf16)dv[dt][r];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------- sequences longer than one tile
// S^T tile (keys x this wave's 32 queries) of one staged key block
__device__ __forceinline__ void score_tile(f32x16 (&acc)[4], const bf16* Ks, const Vec8<bf16> (&fq)[4], int l31, int half) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) mma(acc[kt], frag_row(Ks, kt * 32 + l31, ds * 16, half), fq[ds]);
    }
}

__global__ __launch_bounds__(256) void attn_fwd_long_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Qs = reinterpret_cast<bf16*>(smem);
    bf16* Ks = Qs + SP * RP;
    bf16* Vt = Ks + SP * RP;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H, q0 = blockIdx.x * SP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int S = a.S, nkb = (S + SP - 1) / SP;
    const long ld = 3L * a.H * DH;
    const bf16* base = a.qkv + (long)b * S * ld + h * DH;
    stage(base + (long)q0 * ld, ld, S - q0, Qs, nullptr, tid);
    __syncthreads();
    Vec8<bf16> fq[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) fq[ds] = frag_row(Qs, w * 32 + l31, ds * 16, half);
    f32x16 acc[4];
    // sweep 1: running maximum and sum of exponentials of this lane's query over all keys
    float m = -3.0e38f, sum = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        stage(base + a.H * DH + (long)kb * SP * ld, ld, S - kb * SP, Ks, nullptr, tid);
        __syncthreads();
        score_tile(acc, Ks, fq, l31, half);
        float bm = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * SP + kt * 32 + row_of(r, half);
                const float t = key < S ? acc[kt][r] * a.scale : -3.0e38f;
                acc[kt][r] = t;
                bm = fmaxf(bm, t);
            }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        float part = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part += __expf(acc[kt][r] - mn);
        sum = sum * __expf(m - mn) + part;
        m = mn;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float L = m + __logf(sum);
    const int q = q0 + w * 32 + l31;
    if (half == 0 && q < S) a.lse[(long)bh * S + q] = L;
    // sweep 2: O = sum over key blocks of exp(s - L) V
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        stage(base + a.H * DH + (long)kb * SP * ld, ld, S - kb * SP, Ks, nullptr, tid);
        stage(base + 2 * a.H * DH + (long)kb * SP * ld, ld, S - kb * SP, nullptr, Vt, tid);
        __syncthreads();
        score_tile(acc, Ks, fq, l31, half);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                Vec8<bf16> pa;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * st + j, key = kb * SP + kt * 32 + row_of(r, half);
                    pa.v[j] = (bf16)(key < S ? __expf(acc[kt][r] * a.scale - L) : 0.f);
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) mma(o[dt], pa, frag_perm(Vt, dt * 32 + l31, kt * 32 + 16 * st + 4 * half));
            }
    }
    bf16* ob = a.out + (long)b * S * (a.H * DH) + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + w * 32 + row_of(r, half);
            if (qq < S) ob[(long)qq * (a.H * DH) + dt * 32 + l31] = (bf16)o[dt][r];
        }
}

// D[q] = sum_d dO[q][d] O[q][d] for the 128 queries of a staged dO block (row image Gs); O read from HBM
__device__ __forceinline__ void delta_block(const bf16* Gs, const bf16* o, long ldo, int nvalid, float* Ds, int tid) {
    const int q = tid >> 1, part = tid & 1;
    float acc = 0.f;
    if (q < nvalid) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const Vec8<bf16> x = Vec8<bf16>::load(o + (long)q * ldo + part * 32 + v * 8);
            const Vec8<bf16> g = Vec8<bf16>::load(Gs + q * RP + part * 32 + v * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += (float)x.v[i] * (float)g.v[i];
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    if (part == 0) Ds[q] = acc;
}

// dQ of one block of 128 queries: lane <-> query, walks the key blocks
__global__ __launch_bounds__(256) void attn_bwd_dq_long_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Qs = reinterpret_cast<bf16*>(smem);
    bf16* Gs = Qs + SP * RP;
    bf16* Ks = Gs + SP * RP;
    bf16* Vs = Ks + SP * RP;
    bf16* Kt = Vs + SP * RP;
    float* Ds = reinterpret_cast<float*>(Kt + DH * TP);
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H, q0 = blockIdx.x * SP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int S = a.S, nkb = (S + SP - 1) / SP;
    const long ld = 3L * a.H * DH, ldo = (long)a.H * DH;
    const bf16* base = a.qkv + (long)b * S * ld + h * DH;
    stage(base + (long)q0 * ld, ld, S - q0, Qs, nullptr, tid);
    stage(a.dout + ((long)b * S + q0) * ldo + h * DH, ldo, S - q0, Gs, nullptr, tid);
    __syncthreads();
    delta_block(Gs, a.out + ((long)b * S + q0) * ldo + h * DH, ldo, S - q0, Ds, tid);
    Vec8<bf16> fq[4], fg[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        fq[ds] = frag_row(Qs, w * 32 + l31, ds * 16, half);
        fg[ds] = frag_row(Gs, w * 32 + l31, ds * 16, half);
    }
    __syncthreads();
    const int q = q0 + w * 32 + l31;
    const float Lq = q < S ? a.lse[(long)bh * S + q] : 0.f;
    const float Dq = Ds[w * 32 + l31];
    f32x16 acc[4], dp[4], dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        stage(base + a.H * DH + (long)kb * SP * ld, ld, S - kb * SP, Ks, Kt, tid);
        stage(base + 2 * a.H * DH + (long)kb * SP * ld, ld, S - kb * SP, Vs, nullptr, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[kt][r] = 0.f; dp[kt][r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                mma(acc[kt], frag_row(Ks, kt * 32 + l31, ds * 16, half), fq[ds]);      // S^T
                mma(dp[kt], frag_row(Vs, kt * 32 + l31, ds * 16, half), fg[ds]);       // dP^T = V dO^T
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                Vec8<bf16> sa;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * st + j, key = kb * SP + kt * 32 + row_of(r, half);
                    const float p = (key < S && q < S) ? __expf(acc[kt][r] * a.scale - Lq) : 0.f;
                    sa.v[j] = (bf16)(p * (dp[kt][r] - Dq) * a.scale);                   // dS[q][key]
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) mma(dq[dt], sa, frag_perm(Kt, dt * 32 + l31, kt * 32 + 16 * st + 4 * half));
            }
    }
    bf16* dq_out = a.dqkv + (long)b * S * ld + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + w * 32 + row_of(r, half);
            if (qq < S) dq_out[(long)qq * ld + dt * 32 + l31] = (bf16)dq[dt][r];
        }
}

// dK and dV of one block of 128 keys: lane <-> key, walks the query blocks
__global__ __launch_bounds__(256) void attn_bwd_dkv_long_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Ks = reinterpret_cast<bf16*>(smem);
    bf16* Vs = Ks + SP * RP;
    bf16* Qs = Vs + SP * RP;
    bf16* Gs = Qs + SP * RP;
    bf16* Qt = Gs + SP * RP;
    bf16* Gt = Qt + DH * TP;
    float* Ls = reinterpret_cast<float*>(Gt + DH * TP);
    float* Ds = Ls + SP;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H, k0 = blockIdx.x * SP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int S = a.S, nqb = (S + SP - 1) / SP;
    const long ld = 3L * a.H * DH, ldo = (long)a.H * DH;
    const bf16* base = a.qkv + (long)b * S * ld + h * DH;
    stage(base + a.H * DH + (long)k0 * ld, ld, S - k0, Ks, nullptr, tid);
    stage(base + 2 * a.H * DH + (long)k0 * ld, ld, S - k0, Vs, nullptr, tid);
    __syncthreads();
    Vec8<bf16> fk[4], fv[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        fk[ds] = frag_row(Ks, w * 32 + l31, ds * 16, half);
        fv[ds] = frag_row(Vs, w * 32 + l31, ds * 16, half);
    }
    const int key = k0 + w * 32 + l31;
    f32x16 acc[4], dp[4], dv[2], dk[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dv[dt][r] = 0.f; dk[dt][r] = 0.f; }
    for (int qb = 0; qb < nqb; ++qb) {
        const int q0 = qb * SP;
        __syncthreads();
        stage(base + (long)q0 * ld, ld, S - q0, Qs, Qt, tid);
        stage(a.dout + ((long)b * S + q0) * ldo + h * DH, ldo, S - q0, Gs, Gt, tid);
        if (tid < SP) Ls[tid] = q0 + tid < S ? a.lse[(long)bh * S + q0 + tid] : 0.f;
        __syncthreads();
        delta_block(Gs, a.out + ((long)b * S + q0) * ldo + h * DH, ldo, S - q0, Ds, tid);
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[qt][r] = 0.f; dp[qt][r] = 0.f; }
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                mma(acc[qt], frag_row(Qs, qt * 32 + l31, ds * 16, half), fk[ds]);      // S
                mma(dp[qt], frag_row(Gs, qt * 32 + l31, ds * 16, half), fv[ds]);       // dP = dO V^T
            }
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                Vec8<bf16> pa, sa;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * st + j, ql = qt * 32 + row_of(r, half);
                    const float p = (key < S && q0 + ql < S) ? __expf(acc[qt][r] * a.scale - Ls[ql]) : 0.f;
                    pa.v[j] = (bf16)p;
                    sa.v[j] = (bf16)(p * (dp[qt][r] - Ds[ql]) * a.scale);               // dS[q][key]
                }
                const int qbase = qt * 32 + 16 * st + 4 * half;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    mma(dv[dt], pa, frag_perm(Gt, dt * 32 + l31, qbase));               // dV = P^T dO
                    mma(dk[dt], sa, frag_perm(Qt, dt * 32 + l31, qbase));               // dK = dS^T Q
                }
            }
    }
    bf16* dq_out = a.dqkv + (long)b * S * ld + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + w * 32 + row_of(r, half);
            if (kk < S) {
                dq_out[(long)kk * ld + a.H * DH + dt * 32 + l31] = (bf16)dk[dt][r];
                dq_out[(long)kk * ld + 2 * a.H * DH + dt * 32 + l31] = (bf16)dv[dt][r];
            }
        }
}

constexpr size_t FWD_LDS = sizeof(bf16) * (2 * SP * RP + DH * TP) + sizeof(float) * SP;
constexpr size_t BWD_LDS = sizeof(bf16) * (4 * SP * RP + 3 * DH * TP) + sizeof(float) * 2 * SP;
constexpr size_t DQ_LDS = sizeof(bf16) * (4 * SP * RP + DH * TP) + sizeof(float) * SP;
constexpr size_t DKV_LDS = sizeof(bf16) * (4 * SP * RP + 2 * DH * TP) + sizeof(float) * 2 * SP;

template <typename K>
bool allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace

extern "C" int audiossl_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, float scale, void* stream) {
    ASSL_REQUIRE(qkv && out && lse && B > 0 && S > 0 && H > 0 && (long)B * H <= 65535);
    if (!ASSL_ALIGNED16(qkv) || !ASSL_ALIGNED16(out)) return ASSL_EALIGN;
    static const bool ok = allow_lds(&attn_fwd_kernel, FWD_LDS) && allow_lds(&attn_fwd_long_kernel, FWD_LDS);
    if (!ok) return ASSL_ELAUNCH;
    AttnArgs a{static_cast<const bf16*>(qkv), nullptr, static_cast<bf16*>(out), nullptr, lse, B, S, H, scale};
    if (S <= SP) hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(256), FWD_LDS, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(attn_fwd_long_kernel, dim3(ceil_div(S, SP), B * H), dim3(256), FWD_LDS, static_cast<hipStream_t>(stream), a);
    ASSL_LAUNCH_CHECK();
}

extern "C" int audiossl_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S, int H,
                                 float scale, void* stream) {
    ASSL_REQUIRE(qkv && dout && lse && dqkv && B > 0 && S > 0 && H > 0 && (long)B * H <= 65535);
    ASSL_REQUIRE(S <= SP || out);                       // the multi-block backward needs the forward output
    if (!ASSL_ALIGNED16(qkv) || !ASSL_ALIGNED16(dout) || !ASSL_ALIGNED16(dqkv) || (out && !ASSL_ALIGNED16(out))) return ASSL_EALIGN;
    static const bool ok = allow_lds(&attn_bwd_kernel, BWD_LDS) && allow_lds(&attn_bwd_dq_long_kernel, DQ_LDS) &&
                           allow_lds(&attn_bwd_dkv_long_kernel, DKV_LDS);
    if (!ok) return ASSL_ELAUNCH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    AttnArgs a{static_cast<const bf16*>(qkv), static_cast<const bf16*>(dout), const_cast<bf16*>(static_cast<const bf16*>(out)),
               static_cast<bf16*>(dqkv), const_cast<float*>(lse), B, S, H, scale};
    if (S <= SP) {
        hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H), dim3(256), BWD_LDS, s, a);
    } else {
        const dim3 grid(ceil_div(S, SP), B * H);
        hipLaunchKernelGGL(attn_bwd_dq_long_kernel, grid, dim3(256), DQ_LDS, s, a);
        hipLaunchKernelGGL(attn_bwd_dkv_long_kernel, grid, dim3(256), DKV_LDS, s, a);
    }
    ASSL_LAUNCH_CHECK();
}
