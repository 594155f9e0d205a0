// MFMA GEMM for gfx950 (CDNA4): C[M,N] (+)= alpha * op(A)[M,K] * op(B)[K,N] with fused epilogue.
//
// One 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile; each wave a 64x64 sub-tile =
// 2x2 MFMA 32x32 accumulators.  K is walked in steps of 32 through double-buffered LDS with register
// prefetch (global loads of step k+1 are issued before the MFMAs of step k; their LDS write lands after).
//   T = bf16 : v_mfma_f32_32x32x16_bf16  (fp32 accumulate)
//   T = float: v_mfma_f32_32x32x2_f32    (exact fp32 fma chain) - the validation / high-precision path
// Operand storage ("row" = the M index of A or the N index of B):
//   TRANS=false : [row][K]  K contiguous  -> LDS image [row][K], fragments by one 16/32-byte read
//   TRANS=true  : [K][row]  row contiguous -> LDS image [K][row], fragments by 8 strided reads
// so  NT (Linear fwd: X*W^T) = <false,false>, NN (dX = dY*W) = <false,true>, TN (dW = dY^T*X) = <true,true>.
// Fragment K-order: lane half h owns k in [8h, 8h+8) of every 16-wide k-step for both operands; for
// bf16 that is the hardware map of 32x32x16, for f32 the 8 k's are fed to 8 successive 32x32x2 MFMAs
// (any bijection k->(step,half) is valid as long as A and B agree).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
// K-step per launch: bf16 64 when the grid fills the chip several times over (73 KB of LDS -> 2 workgroups / CU), 128 when
// there is at most one workgroup per CU anyway (the loop is then latency-bound per step: fewer, fatter steps win;
// measured 40 -> 26 us on 1024x2048x2048).  fp32 (validation path): 32.

struct GemmArgs {
    const void* A; const void* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha;
    const float* bias;        // [N] fp32 or null
    int relu;
    const uint8_t* keep;      // [M][ldk] keep-mask (1 = keep) or null
    long ldk;
    float keep_scale;
    const void* gate;         // T [M][ldg]: out = gate > 0 ? out : 0
    long ldg;
    int out_f32;              // C is float even when T is bf16
    int atomic;               // 1: C (float) += via atomicAdd (split-K / concurrent accumulation); 2: C += by plain
                              // load-add-store (one launch at a time owns C: no split-K, no concurrent writers)
    int ksplit;               // gridDim.z
    const float* resid;       // fp32 [M][ldr] added to the result (out-of-place residual connection) or null
    long ldr;
    unsigned a_bytes, b_bytes; // extents of A and B for the buffer descriptors (hardware bounds check)
    int vec_epi;              // every epilogue operand allows 8-column vectors (set by epi_vectorisable)
    int xcd_remap;            // grid is a multiple of 8 workgroups: contiguous tile runs per XCD (set by the launchers)
    // row-softmax epilogues of the MoCo InfoNCE head (audiossl_moco_logits): the [B][K] logits never reach memory
    int lse_mode;             // 1: part[row][col0/64] = (max, sum exp) of alpha*acc over the wave's 64 columns
                              // 2: C (bf16) = exp(alpha*acc - lse[row]) * gscale
    float* part;              // mode 1: [M][nslot][2] fp32
    const float* lse;         // mode 2: [M]
    float gscale;
    int nslot;                // ceil(N / 64)
    // Barlow-twins epilogue (audiossl_gemm_multi_barlow): c = alpha*acc is the cross-correlation; the launch stores
    // dc = bl_dscale * (c - I) (bf16) and adds bl_coef * sum (c - I)^2 into one of 32 replicas of the loss - c itself never reaches memory
    int bl_mode;
    float* bl_loss;           // [32] fp32 replicas (workgroup i adds into replica i & 31; the caller sums them)
    float bl_coef, bl_dscale;
    // SGD epilogue (audiossl_gemm_multi_sgd): the fp32 result is a weight gradient that is consumed on the spot - parameter sgd_p and
    // momentum sgd_m ([M][ldc] fp32, indexed like C) are updated in place, the bf16 copy of the new parameter goes to sgd_s (nullable);
    // nothing is written to C
    float* sgd_p; float* sgd_m; bf16* sgd_s;
    const float* sgd_gs_dev;  // optional device scalar multiplied into the gradient scale
    float sgd_lr, sgd_mu, sgd_wd, sgd_gs;
    // dropout drawn in the epilogue (audiossl_gemm_dropout): element (row, col) is kept iff dropout_keep(seed', row * N + col) - the very
    // mask audiossl_dropout_mask would have written for an [M][N] tensor - and scaled by keep_scale; no mask tensor exists
    int drop_on;
    unsigned int drop_thr;
    unsigned long long drop_seed;
    const long long* drop_counter;
};

// kernel names as rocprofv3 prints them (demangled, except for instantiations on __bf16, which its demangler leaves mangled)
template <typename T> const char* type_code();
template <> const char* type_code<bf16>() { return "DF16b"; }
template <> const char* type_code<float>() { return "f"; }
static const char* note_name(char (&buf)[96], const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return buf;
}
#define TF(b) ((b) ? "true" : "false")

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    __device__ __forceinline__ static void run(f32x16& acc, const Vec8<bf16>& a, const Vec8<bf16>& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ __forceinline__ static void run(f32x16& acc, const Vec8<float>& a, const Vec8<float>& b) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[s], b.lo[s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[s], b.hi[s], acc, 0, 0, 0);
    }
};

// Epilogue through LDS.  The accumulators of a wave (64 x 64, C/D map of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) are parked in a wave-private fp32 tile, then written out row by row with
// lane = column: every optional term (bias, ReLU, dropout mask, gate, residual) is a coalesced load in a short runtime loop.
// The fully unrolled per-element version this replaces was ~35 KB of straight-line code executed once per workgroup - the
// instruction fetch alone cost ~14 us per launch (a one-tile, one-k-step GEMM took 16.7 us; tools/gemm_floor.py).
constexpr int EPI_PITCH = 65;                                    // floats; odd pitch: conflict-free row reads
constexpr size_t EPI_LDS = sizeof(float) * 4 * 64 * EPI_PITCH;   // 66,560 B for four waves (8-wave workgroups: twice that)

// Accumulators -> the wave's fp32 tile [32 * MI][64] in LDS.  32x32 MFMA: C/D map col = lane & 31, row = (r & 3) + 8 * (r >> 2) +
// 4 * (lane >> 5); 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r.
template <int MI>
__device__ __forceinline__ void park(const f32x16 (&acc)[MI][2], float* ct, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 65 + j * 32 + l31] = acc[i][j][r];
}
template <int MB>
__device__ __forceinline__ void park(const f32x4 (&acc)[MB][4], float* ct, int lane) {
    const int q = lane >> 4, c = lane & 15;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ct[(i * 16 + 4 * q + r) * 65 + j * 16 + c] = acc[i][j][r];
}

// MI = 32-row MFMA blocks per wave in M (wave tile = 32*MI x 64)
template <typename T, int MI, typename ACC>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& g, ACC& acc, char* smem, int row0, int col0, int lane,
                                             int wave) {
    __syncthreads();                                             // every wave is done with the staging buffers
    float* ct = reinterpret_cast<float*>(smem) + wave * (32 * MI) * EPI_PITCH;
    park(acc, ct, lane);
    const int col = col0 + lane;
    if (col >= g.N) return;
    const float bias = (g.bias && blockIdx.z == 0) ? g.bias[col] : 0.f;      // split-K: the first split adds the bias
    const float floor_ = g.relu ? 0.f : -3.4e38f;
    const T* gate = static_cast<const T*>(g.gate);
    const int rows = min(32 * MI, g.M - row0);
    // 8 rows per trip: the LDS reads and the optional global loads of a trip are independent, so their latencies overlap
    // (one row per trip serialised ~150 cycles of LDS + store issue per row: 4 us of a 14 us single-tile launch)
    for (int r0 = 0; r0 < rows; r0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ct[min(r0 + u, 32 * MI - 1) * EPI_PITCH + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long row = row0 + min(r0 + u, rows - 1);
            float x = fmaxf(g.alpha * v[u] + bias, floor_);
            if (g.keep) x = g.keep[row * g.ldk + col] ? x * g.keep_scale : 0.f;
            if (g.drop_on) x = dropout_keep(dropout_seed(g.drop_seed, g.drop_counter), row * g.N + col, g.drop_thr) ? x * g.keep_scale : 0.f;
            if (gate) x = to_f32(gate[row * g.ldg + col]) > 0.f ? x : 0.f;
            if (g.resid) x += g.resid[row * g.ldr + col];
            v[u] = x;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (r0 + u >= rows) break;
            const long o = (long)(row0 + r0 + u) * g.ldc + col;
            if (g.sgd_p) {
                float pj = g.sgd_p[o], b = g.sgd_m[o];
                sgd_step(pj, v[u], b, g.sgd_lr, g.sgd_mu, g.sgd_wd, g.sgd_gs_dev ? g.sgd_gs * g.sgd_gs_dev[0] : g.sgd_gs, false);
                g.sgd_p[o] = pj; g.sgd_m[o] = b;
                if (g.sgd_s) g.sgd_s[o] = (bf16)pj;
                continue;
            }
            if (g.atomic == 1)  atomicAdd(static_cast<float*>(g.C) + o, v[u]);
            else if (g.atomic)  static_cast<float*>(g.C)[o] += v[u];
            else if (g.out_f32) static_cast<float*>(g.C)[o] = v[u];
            else                static_cast<T*>(g.C)[o] = from_f32<T>(v[u]);
        }
    }
}

// The same with 8 consecutive columns per lane (8 rows x 64 columns per wave trip): 16-byte stores of bf16 results (two for
// fp32), 8-byte keep-mask loads, 16-byte gate / residual loads.  The one-column-per-lane form above issues a 128-byte
// store per row and wave - on the 6144 x 2048 Linear layers of the encoder the epilogue then took longer than the k-loop
// (70 us at K = 512 against 113 us at K = 2048).  bf16 operands, non-atomic results, N % 8 == 0 and aligned operands only.
template <int MI, typename ACC>
__device__ __forceinline__ void epilogue_vec(const GemmArgs& g, ACC& acc, char* smem, int row0, int col0, int lane,
                                             int wave) {
    __syncthreads();
    float* ct = reinterpret_cast<float*>(smem) + wave * (32 * MI) * EPI_PITCH;
    park(acc, ct, lane);
    const int c8 = (lane & 7) * 8, rr = lane >> 3;
    const int col = col0 + c8;
    if (col >= g.N) return;
    float bias[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) bias[u] = 0.f;
    if (g.bias && blockIdx.z == 0) {
        const Vec8<float> b = Vec8<float>::load(g.bias + col);
#pragma unroll
        for (int u = 0; u < 8; ++u) bias[u] = b.get(u);
    }
    const float floor_ = g.relu ? 0.f : -3.4e38f;
    const bf16* gate = static_cast<const bf16*>(g.gate);
    float bl_acc = 0.f;
#pragma unroll 2
    for (int r0 = 0; r0 < 32 * MI; r0 += 8) {
        const int rl = r0 + rr;
        const long row = row0 + rl;
        if (row >= g.M) continue;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = fmaxf(g.alpha * ct[rl * EPI_PITCH + c8 + u] + bias[u], floor_);
        if (g.bl_mode) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = v[u] - (row == col + u ? 1.f : 0.f);
                bl_acc += d * d;
                v[u] = g.bl_dscale * d;
            }
        }
        if (g.keep) {
            const unsigned long long k8 = *reinterpret_cast<const unsigned long long*>(g.keep + row * g.ldk + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ((k8 >> (8 * u)) & 0xFFull) ? v[u] * g.keep_scale : 0.f;
        }
        if (g.drop_on) {
            const unsigned long long sd = dropout_seed(g.drop_seed, g.drop_counter);
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dropout_keep(sd, row * g.N + col + u, g.drop_thr) ? v[u] * g.keep_scale : 0.f;
        }
        if (gate) {
            const Vec8<bf16> gt = Vec8<bf16>::load(gate + row * g.ldg + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gt.get(u) > 0.f ? v[u] : 0.f;
        }
        if (g.resid) {
            const Vec8<float> rs = Vec8<float>::load(g.resid + row * g.ldr + col);
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] += rs.get(u);
        }
        const long o = row * g.ldc + col;
        if (g.sgd_p) {
            const float gs = g.sgd_gs_dev ? g.sgd_gs * g.sgd_gs_dev[0] : g.sgd_gs;
            Vec8<float> pv = Vec8<float>::load(g.sgd_p + o), mv = Vec8<float>::load(g.sgd_m + o);
            Vec8<bf16> sv;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float pj = pv.get(u), b = mv.get(u);
                sgd_step(pj, v[u], b, g.sgd_lr, g.sgd_mu, g.sgd_wd, gs, false);
                pv.set(u, pj); mv.set(u, b); sv.set(u, pj);
            }
            pv.store(g.sgd_p + o);
            mv.store(g.sgd_m + o);
            if (g.sgd_s) sv.store(g.sgd_s + o);
            continue;
        }
        if (g.out_f32) {
            Vec8<float> out;
            if (g.atomic) {                                   // exclusive accumulation (atomic == 2)
                const Vec8<float> c0 = Vec8<float>::load(static_cast<const float*>(g.C) + o);
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] += c0.get(u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) out.set(u, v[u]);
            out.store(static_cast<float*>(g.C) + o);
        } else {
            Vec8<bf16> out;
#pragma unroll
            for (int u = 0; u < 8; ++u) out.set(u, v[u]);
            out.store(static_cast<bf16*>(g.C) + o);
        }
    }
    if (g.bl_mode) {                                          // N % 64 == 0 here: no lane left the function early
        bl_acc = wave_sum(bl_acc);
        if (lane == 0) atomicAdd(g.bl_loss + (blockIdx.x & 31), g.bl_coef * bl_acc);
    }
}

// Row-softmax epilogues (MoCo InfoNCE, `delores_m/upstream_expert.py:250-264`): logits = alpha * acc are reduced / transformed
// in the tile and never stored.  Same parking of the accumulators as above; a lane owns 8 consecutive columns of one of 8 rows.
template <int MI, typename ACC>
__device__ __forceinline__ void epilogue_softmax(const GemmArgs& g, ACC& acc, char* smem, int row0, int col0, int lane,
                                                 int wave) {
    __syncthreads();
    float* ct = reinterpret_cast<float*>(smem) + wave * (32 * MI) * EPI_PITCH;
    park(acc, ct, lane);
    const int c8 = (lane & 7) * 8, rr = lane >> 3;
    const int col = col0 + c8;
    if (col0 >= g.N) return;                                   // the whole wave tile is outside (wave-uniform)
#pragma unroll 2
    for (int r0 = 0; r0 < 32 * MI; r0 += 8) {
        const int rl = r0 + rr;
        const long row = row0 + rl;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (col + u < g.N) ? g.alpha * ct[rl * EPI_PITCH + c8 + u] : -3.0e38f;
        if (g.lse_mode == 1) {
            float m = v[0];
#pragma unroll
            for (int u = 1; u < 8; ++u) m = fmaxf(m, v[u]);
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) sm += __expf(v[u] - m);
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {                   // the 8 lanes of a row are consecutive
                const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(sm, o, 64);
                const float mm = fmaxf(m, m2);
                sm = sm * __expf(m - mm) + s2 * __expf(m2 - mm);
                m = mm;
            }
            if ((lane & 7) == 0 && row < g.M) {
                float* o = g.part + (row * g.nslot + (col0 >> 6)) * 2;
                o[0] = m; o[1] = sm;
            }
        } else if (row < g.M && col < g.N) {
            const float l = g.lse[row];
            Vec8<bf16> out;
#pragma unroll
            for (int u = 0; u < 8; ++u) out.set(u, __expf(v[u] - l) * g.gscale);
            out.store(static_cast<bf16*>(g.C) + row * g.ldc + col);
        }
    }
}

// MI = 32-row blocks of the wave tile; ACC = f32x16 [MI][2] (32x32 MFMA) or f32x4 [2 * MI][4] (16x16 MFMA)
template <typename T, int MI, typename ACC>
__device__ __forceinline__ void epilogue(const GemmArgs& g, ACC& acc, char* smem, int row0, int col0, int lane, int wave) {
    if constexpr (sizeof(T) == 2) {
        if (g.lse_mode) { epilogue_softmax<MI>(g, acc, smem, row0, col0, lane, wave); return; }
        if (g.vec_epi) { epilogue_vec<MI>(g, acc, smem, row0, col0, lane, wave); return; }
    }
    epilogue_lds<T, MI>(g, acc, smem, row0, col0, lane, wave);
}

// One operand's staging: 128 rows x 32 k per step, two Vec8 per thread.
template <typename T, bool TRANS, int BK_, int ROWS = 128, int NTH = 256>
struct Stage {
    static constexpr int BK = BK_;
    static constexpr int NT_PITCH = BK + 8;          // elements; [row][k] image, conflict-free 16-byte row reads
    static constexpr int TR_PITCH = ROWS + 32;       // elements; [k][row] image: 4 consecutive k-rows fall on distinct 32-byte bank groups
    static constexpr int VR = ROWS / 8;              // vectors per k-row of the [k][row] image
    static constexpr int NV = ROWS * BK / 8 / NTH;   // 8-element vectors per thread and step
    static constexpr int VPR = BK / 8;               // vectors per row of the [row][k] image
    Vec8<T> r[NV];
    // Branch-free staging loads: a raw buffer load per 16 bytes, out-of-range vectors get an offset past the
    // descriptor's extent and come back as zeros (no exec-masked branches -> the compiler keeps counted vmcnt waits).
    // rows = extent of the row dimension (M or N); kend = exclusive K bound of this split
    __device__ __forceinline__ static Vec8<bf16> bload16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
        Vec8<bf16> r;
        r.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        return r;
    }
    __device__ __forceinline__ static Vec8<float> bload32(__amdgpu_buffer_rsrc_t rs, unsigned off) {
        Vec8<float> r;
        r.lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        r.hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, 0));
        return r;
    }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, long ld, int row0, int rows, int k0, int kend) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = t + i * NTH;
            int row, k;
            long idx;
            if (!TRANS) { row = row0 + v / VPR; k = k0 + (v % VPR) * 8; idx = (long)row * ld + k; }
            else        { k = k0 + v / VR; row = row0 + (v % VR) * 8; idx = (long)k * ld + row; }
            const unsigned off = (row < rows && k < kend) ? (unsigned)(idx * (long)sizeof(T)) : 0xFFFFFFE0u;
            r[i] = select_load(rs, off, (T*)nullptr);
        }
    }
    __device__ __forceinline__ static Vec8<bf16> select_load(__amdgpu_buffer_rsrc_t rs, unsigned off, bf16*) { return bload16(rs, off); }
    __device__ __forceinline__ static Vec8<float> select_load(__amdgpu_buffer_rsrc_t rs, unsigned off, float*) { return bload32(rs, off); }
    __device__ __forceinline__ void put(T* lds) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = t + i * NTH;
            if (!TRANS) r[i].store(lds + (v / VPR) * NT_PITCH + (v % VPR) * 8);
            else        r[i].store(lds + (v / VR) * TR_PITCH + (v % VR) * 8);
        }
    }
    // fragment of 32 rows starting at `row`, k-step `kk` (0 or 16) for this lane
    __device__ __forceinline__ static Vec8<T> frag(const T* lds, int row, int kk, int lane) {
        const int r_ = row + (lane & 31), kb = kk + 8 * (lane >> 5);
        if (!TRANS) return Vec8<T>::load(lds + r_ * NT_PITCH + kb);
        return frag_trans(lds, row, kb, r_, lane);
    }
    // [k][row] image -> 8 consecutive k of one row.  bf16: two ds_read_b64_tr_b16 (each 16-lane group transposes a
    // 4 (k) x 16 (row) block: lane 4q+p supplies the address of k-row q, columns 4p..4p+3, and receives column i);
    // fp32: 8 scalar reads (validation path).
    __device__ __forceinline__ static Vec8<float> frag_trans(const float* lds, int, int kb, int r_, int) {
        Vec8<float> f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.set(j, lds[(kb + j) * TR_PITCH + r_]);
        return f;
    }
    __device__ __forceinline__ static Vec8<bf16> frag_trans(const bf16* lds, int row, int kb, int, int lane) {
        typedef __attribute__((address_space(3))) bf16x4* lds4_t;
        const int i = lane & 15, q = i >> 2, p = i & 3;
        const bf16* a = lds + (kb + q) * TR_PITCH + row + 16 * ((lane >> 4) & 1) + 4 * p;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)a);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(a + 4 * TR_PITCH));
        Vec8<bf16> f;
        f.v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    }
    static constexpr int LDS_ELEMS = TRANS ? BK * TR_PITCH : ROWS * NT_PITCH;
};

// NW = 4 waves (2 x 2): 128 x 128 tile (MI = 2) or 64 x 128 (MI = 1: twice the workgroups for grids that would leave CUs idle).
// NW = 8 waves (4 x 2): 256 x 128 tile - twice the MFMA work per byte staged (the 128 x 128 tile needs as many L1->LDS cycles
// per k-step as MFMA cycles) and two waves per SIMD, so one wave's MFMAs cover the other's LDS reads and staging stores.
template <typename T, bool TA, bool TB, int BK, int MI, int NW = 4, bool SPLIT_EPI = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, char* smem, int wg) {
    constexpr int BMv = 16 * MI * NW;
    using SA = Stage<T, TA, BK, BMv, 64 * NW>;
    using SB = Stage<T, TB, BK, BN, 64 * NW>;
    T* const ldsA0 = reinterpret_cast<T*>(smem);                 // two A buffers, then two B buffers
    T* const ldsB0 = ldsA0 + 2 * SA::LDS_ELEMS;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int bm = (wg / tiles_n) * BMv, bn = (wg % tiles_n) * BN;
    // split-K range, in whole BK steps
    const int ksteps = (g.K + BK - 1) / BK;
    const int per = (ksteps + g.ksplit - 1) / g.ksplit;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);
    if (ks0 >= ks1) return;
    const int kend = min(g.K, ks1 * BK);

    const __amdgpu_buffer_rsrc_t A = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t B = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, g.b_bytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32 * MI, wn = (wave & 1) * 64;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Double-buffered LDS + one register stage: the loads of step k+1 are issued before the MFMAs of step k and land in
    // LDS after them (counted vmcnt, no branches).  A second register stage (prefetch distance 2) was measured slower:
    // hipcc aliases the staging and fragment registers and drains vmcnt at the loop back-edge (gpurun_out/gemm_d2.log).
    SA sa; SB sb;
    sa.load(A, g.lda, bm, g.M, ks0 * BK, kend);
    sb.load(B, g.ldb, bn, g.N, ks0 * BK, kend);
    sa.put(ldsA0); sb.put(ldsB0);
    __syncthreads();

    int cur = 0;
    for (int ks = ks0; ks < ks1; ++ks) {
        const bool more = ks + 1 < ks1;
        if (more) {
            sa.load(A, g.lda, bm, g.M, (ks + 1) * BK, kend);
            sb.load(B, g.ldb, bn, g.N, (ks + 1) * BK, kend);
        }
        const T* la = ldsA0 + cur * SA::LDS_ELEMS;
        const T* lb = ldsB0 + cur * SB::LDS_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            Vec8<T> fa[MI], fb[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = SA::frag(la, wm + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = SB::frag(lb, wn + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
        }
        if (more) { sa.put(ldsA0 + (cur ^ 1) * SA::LDS_ELEMS); sb.put(ldsB0 + (cur ^ 1) * SB::LDS_ELEMS); }
        __syncthreads();
        cur ^= 1;
    }

    if constexpr (SPLIT_EPI && MI == 2) {                    // half-height epilogue tile: 33 KB of LDS instead of 66.5 KB
        epilogue<T, 1>(g, *reinterpret_cast<f32x16 (*)[1][2]>(&acc[0]), smem, bm + wm, bn + wn, lane, wave);
        epilogue<T, 1>(g, *reinterpret_cast<f32x16 (*)[1][2]>(&acc[1]), smem, bm + wm + 32, bn + wn, lane, wave);
    } else {
        epilogue<T, MI>(g, acc, smem, bm + wm, bn + wn, lane, wave);
    }
}

template <typename T, bool TA, bool TB, int BK, int MI, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Workgroups are dealt to the 8 XCDs round-robin (id mod 8), each with its own 4 MB L2.  Give every XCD a CONTIGUOUS run
    // of tiles (whole tile rows: one A row panel is then fetched into one L2 instead of all eight) - operand re-reads
    // that miss L2 are served by the Infinity Cache at about half the L2 rate, which bounded the 6144-row GEMMs.
    gemm_body<T, TA, TB, BK, MI, NW>(g, smem, g.xcd_remap ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x);
}

// 256 x 256 output tile, 512 threads = 8 waves (2 x 4), each wave 128 x 64 = 4 x 2 accumulators: per 64-deep k-step the
// tile needs 1,024 L1->LDS cycles against 2,048 MFMA cycles (the 128 x 128 tile: 512 / 512) and 6 fragment reads per 8
// MFMAs.  For the 6144-row layers (24 x 8 = 192 tiles: one round on 256 CUs).  bf16 only.
template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs g) {
    constexpr int BK = 64, T256 = 256;
    using SA = Stage<bf16, TA, BK, T256, 512>;
    using SB = Stage<bf16, TB, BK, T256, 512>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const ldsA0 = reinterpret_cast<bf16*>(smem);
    bf16* const ldsB0 = ldsA0 + 2 * SA::LDS_ELEMS;
    const int tiles_n = (g.N + T256 - 1) / T256;
    int wg = blockIdx.x;
    if (g.xcd_remap) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int bm = (wg / tiles_n) * T256, bn = (wg % tiles_n) * T256;
    const int ksteps = (g.K + BK - 1) / BK;
    const int per = (ksteps + g.ksplit - 1) / g.ksplit;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);
    if (ks0 >= ks1) return;
    const int kend = min(g.K, ks1 * BK);
    const __amdgpu_buffer_rsrc_t A = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t B = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, g.b_bytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    SA sa; SB sb;
    sa.load(A, g.lda, bm, g.M, ks0 * BK, kend);
    sb.load(B, g.ldb, bn, g.N, ks0 * BK, kend);
    sa.put(ldsA0); sb.put(ldsB0);
    __syncthreads();
    int cur = 0;
    for (int ks = ks0; ks < ks1; ++ks) {
        const bool more = ks + 1 < ks1;
        if (more) {
            sa.load(A, g.lda, bm, g.M, (ks + 1) * BK, kend);
            sb.load(B, g.ldb, bn, g.N, (ks + 1) * BK, kend);
        }
        const bf16* la = ldsA0 + cur * SA::LDS_ELEMS;
        const bf16* lb = ldsB0 + cur * SB::LDS_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            Vec8<bf16> fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = SA::frag(la, wm + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = SB::frag(lb, wn + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<bf16>::run(acc[i][j], fa[i], fb[j]);
        }
        if (more) { sa.put(ldsA0 + (cur ^ 1) * SA::LDS_ELEMS); sb.put(ldsB0 + (cur ^ 1) * SB::LDS_ELEMS); }
        __syncthreads();
        cur ^= 1;
    }
    // the wave-private epilogue tile holds 64 rows: two passes over the wave's 128
    epilogue<bf16, 2>(g, *reinterpret_cast<f32x16 (*)[2][2]>(&acc[0]), smem, bm + wm, bn + wn, lane, wave);
    epilogue<bf16, 2>(g, *reinterpret_cast<f32x16 (*)[2][2]>(&acc[2]), smem, bm + wm + 64, bn + wn, lane, wave);
}

template <bool TA, bool TB>
int launch256(const GemmArgs& g, hipStream_t s) {
    constexpr size_t stage = sizeof(bf16) * 2 * (Stage<bf16, TA, 64, 256, 512>::LDS_ELEMS + Stage<bf16, TB, 64, 256, 512>::LDS_ELEMS);
    constexpr size_t epi = EPI_LDS * 2;
    constexpr size_t lds = stage > epi ? stage : epi;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<TA, TB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, 256) * ceil_div(g.N, 256);
    GemmArgs ga = g;
    ga.xcd_remap = tiles % 8 == 0 && tiles >= 64;
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm256_kernel<%s, %s>", TF(TA), TF(TB)); }
    hipLaunchKernelGGL((gemm256_kernel<TA, TB>), dim3(tiles, 1, g.ksplit), dim3(512), lds, s, ga);
    ASSL_LAUNCH_CHECK();
}

int dispatch256(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch256<false, false>(g, s);
    if (!ta && tb) return launch256<false, true>(g, s);
    if (ta && tb) return launch256<true, true>(g, s);
    return launch256<true, false>(g, s);
}

// Several independent problems of one kind in ONE launch (blockIdx.y = problem): the three Barlow heads of delores_m run
// the same chain of GEMMs on different operands; issued as separate launches on separate streams they were serialised by
// the hardware-queue mapping of the graph executor (1.6 ms of a 3.3 ms step), and each launch filled half the chip at best.
constexpr int MAX_MULTI = 4;
// tile_end: running totals of the problems' tile counts for the tile shape of the kernel that is launched (set by multi_grid)
struct GemmMulti { GemmArgs p[MAX_MULTI]; int tile_end[MAX_MULTI]; int total; };

// One-dimensional grid over the tiles of ALL problems, padded to a multiple of 8 workgroups.  Workgroup ids go to the XCDs round-
// robin, so id -> (id & 7) * (grid / 8) + (id >> 3) hands every XCD a CONTIGUOUS run of the concatenated tile list: whole tile
// rows of ONE problem (its A panels and its B operand meet in one L2) instead of a slice of every tile row of every problem -
// the multi-problem launches of round 2 re-fetched their operands 4.2 x (86.6 MB per launch against 21 MB, PMC), and the grid was
// sized for the widest problem (idle workgroups for the narrower ones).  -> problem index (-1: padding), tile index in `wg`.
__device__ __forceinline__ int multi_locate(const GemmMulti& gm, int& wg) {
    const int v = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (v >= gm.total) return -1;
    int p = 0;
    while (v >= gm.tile_end[p]) ++p;
    wg = v - (p ? gm.tile_end[p - 1] : 0);
    return p;
}
// host side: fills tile_end / total for BMt x BNt tiles, returns the grid size
static int multi_grid(GemmMulti& gm, int count, int BMt, int BNt) {
    int tot = 0;
    for (int i = 0; i < MAX_MULTI; ++i) {
        if (i < count) tot += ceil_div(gm.p[i].M, BMt) * ceil_div(gm.p[i].N, BNt);
        gm.tile_end[i] = tot;
    }
    gm.total = tot;
    return (tot + 7) / 8 * 8;
}

template <typename T, bool TA, bool TB, int BK, int MI, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_multi_kernel(GemmMulti gm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int wg;
    const int p = multi_locate(gm, wg);
    if (p < 0) return;
    gemm_body<T, TA, TB, BK, MI, NW>(gm.p[p], smem, wg);
}

template <typename T, bool TA, bool TB, int BK, int MI, int NW>
constexpr size_t gemm_lds() {
    constexpr size_t stage = sizeof(T) * 2 * (Stage<T, TA, BK, 16 * MI * NW, 64 * NW>::LDS_ELEMS + Stage<T, TB, BK, BN, 64 * NW>::LDS_ELEMS);
    constexpr size_t epi = EPI_LDS * (NW / 4);
    return stage > epi ? stage : epi;
}

template <typename T, bool TA, bool TB, int BK, int MI, int NW = 4>
int launch_multi(const GemmMulti& gm_, int count, int ksplit, hipStream_t s) {
    GemmMulti gm = gm_;
    const int grid = multi_grid(gm, count, 16 * MI * NW, BN);
    const size_t lds = gemm_lds<T, TA, TB, BK, MI, NW>();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_multi_kernel<T, TA, TB, BK, MI, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_multi_kernelI%sLb%dELb%dELi%dELi%dELi%dE", type_code<T>(), (int)TA, (int)TB, BK, MI, NW); }
    hipLaunchKernelGGL((gemm_multi_kernel<T, TA, TB, BK, MI, NW>), dim3(grid, 1, ksplit), dim3(64 * NW), lds, s, gm);
    ASSL_LAUNCH_CHECK();
}

template <typename T, bool TA, bool TB, int BK, int MI, int NW = 4>
int launch(const GemmArgs& g, hipStream_t s) {
    const size_t lds = gemm_lds<T, TA, TB, BK, MI, NW>();
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, TA, TB, BK, MI, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, 16 * MI * NW) * ceil_div(g.N, BN);
    static const bool remap = getenv("AUDIOSSL_GEMM_XCD") ? atoi(getenv("AUDIOSSL_GEMM_XCD")) != 0 : true;
    GemmArgs ga = g;
    ga.xcd_remap = remap && tiles % 8 == 0 && tiles >= 64;
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_kernelI%sLb%dELb%dELi%dELi%dELi%dE", type_code<T>(), (int)TA, (int)TB, BK, MI, NW); }
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, BK, MI, NW>), dim3(tiles, 1, g.ksplit), dim3(64 * NW), lds, s, ga);
    ASSL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 with direct-to-LDS staging (`global_load_lds_dwordx4`) and an NST-deep ring: the pipelined variant.  The
// register-staged kernel above keeps ONE 64-deep k-step in flight per workgroup, so a k-step costs a full global-load
// latency (~1 us against 0.1-0.2 us of MFMA work on the projector shapes of the step); here up to NST-1 stages are in
// flight across a single raw barrier per step, waited for with counted s_waitcnt vmcnt(N) (never 0 in steady state).
//   * an LDS-DMA instruction writes wave-uniform base + lane * 16 B, so a stage image is the plain tile with no padding;
//     bank conflicts of the fragment reads are removed by XOR swizzles applied to the SOURCE address of the DMA and to
//     the ds_read address:
//       [row][64 k] image (K-contiguous operand, 128-byte rows): 16-byte chunk ^= (row >> 1) & 7
//       [64 k][R rows] image (row-contiguous operand, read with ds_read_b64_tr_b16): 64-byte group ^= k & 3 (R = 128)
//         or ^= (k >> 1) & 1 (R = 64), so the four k-rows one transposing read touches fall on four distinct bank groups;
//   * rows past M / N are clamped to the last row / last 8-row chunk (their results are never stored); K % 64 == 0.
constexpr int GBK = 64;

__device__ __forceinline__ void glds16(const bf16* src, char* dst) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// One operand of the ring: R rows x 64 k per stage.
template <bool TRANS, int R>
struct RingOp {
    static constexpr int BYTES = R * GBK * 2;
    static constexpr int NI = BYTES / 4096;                  // DMA instructions per wave and stage (1 KiB each, 4 waves)
    static constexpr int NB = R / 64;                        // 32-row fragment blocks per wave
    const bf16* p[NI];
    long kstride;
    int foff[NB], fsw[NB];
    __device__ __forceinline__ void init(const void* base_, long ld, int r0, int ext, int k0, int wave, int lane, int wr) {
        const bf16* base = static_cast<const bf16*>(base_);
        if (!TRANS) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rl = wave * (R / 4) + i * 8 + (lane >> 3);
                const int sc = (lane & 7) ^ ((rl >> 1) & 7);
                p[i] = base + (long)min(r0 + rl, ext - 1) * ld + sc * 8 + k0;
            }
            kstride = GBK;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int r = wr + b * 32 + (lane & 31);
                foff[b] = r * 128; fsw[b] = (r >> 1) & 7;
            }
        } else {
            constexpr int CPR = R / 8;                       // 16-byte chunks per k-row
            constexpr int KPI = 64 / CPR;                    // k-rows per DMA instruction
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int kl = wave * 16 + i * KPI + lane / CPR, pc = lane % CPR;
                const int sw = R == 128 ? (kl & 3) : ((kl >> 1) & 1);
                const int roff = ((((pc >> 2) ^ sw) << 2) + (pc & 3)) * 8;
                p[i] = base + (long)(k0 + kl) * ld + min(r0 + roff, ext - 8);
            }
            kstride = (long)GBK * ld;
            const int q = (lane & 15) >> 2, pp = lane & 3, h4 = (lane >> 4) & 1, half = lane >> 5;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int grp = ((wr + b * 32) >> 5) ^ (R == 128 ? q : (q >> 1));
                foff[b] = (8 * half + q) * (R * 2) + grp * 64 + 32 * h4 + 8 * pp; fsw[b] = 0;
            }
        }
    }
    __device__ __forceinline__ void issue(long st, char* dst) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) glds16(p[i] + st * kstride, dst + i * 1024);
    }
    // fragment of block b, 16-wide k-step kk (0..3) of the stage image at `img`
    __device__ __forceinline__ Vec8<bf16> frag(const char* img, int b, int kk, int half) const {
        if (!TRANS) return Vec8<bf16>::load(reinterpret_cast<const bf16*>(img + foff[b] + (((kk * 2 + half) ^ fsw[b]) << 4)));
        typedef __attribute__((address_space(3))) bf16x4* lds4_t;
        const char* a = img + foff[b] + kk * 16 * (R * 2);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)a);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(a + 4 * (R * 2)));
        Vec8<bf16> f;
        f.v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    }
};

template <bool TA, bool TB, int MI, int NST>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmArgs g) {
    constexpr int RA = 64 * MI;
    using OA = RingOp<TA, RA>;
    using OB = RingOp<TB, 128>;
    constexpr int STAGE = OA::BYTES + OB::BYTES;
    constexpr int NPS = OA::NI + OB::NI;                     // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tiles_n = (g.N + BN - 1) / BN;
    // blocks that share an XCD (ids congruent mod 8) get consecutive tiles: they share A row panels / B column panels in L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int bm = (wg / tiles_n) * RA, bn = (wg % tiles_n) * BN;
    const int ksteps = g.K / GBK;
    const int per = (ksteps + g.ksplit - 1) / g.ksplit;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);
    if (ks0 >= ks1) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int wm = (wave >> 1) * 32 * MI, wn = (wave & 1) * 64;

    OA oa; OB ob;
    oa.init(g.A, g.lda, bm, g.M, ks0 * GBK, wave, lane, wm);
    ob.init(g.B, g.ldb, bn, g.N, ks0 * GBK, wave, lane, wn);
    char* const wa = smem + wave * (OA::BYTES / 4);          // this wave's slice of the A image / the B image of slot 0
    char* const wb = smem + OA::BYTES + wave * (OB::BYTES / 4);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = ks1 - ks0;
    int islot = 0;                                           // ring slot the next issued stage goes to
    auto issue = [&](int st) {
        oa.issue(st, wa + islot * STAGE);
        ob.issue(st, wb + islot * STAGE);
        islot = islot + 1 == NST ? 0 : islot + 1;
    };
    for (int st = 0; st < NST - 1 && st < nk; ++st) issue(st);

    int cslot = 0;
    for (int it = 0; it < nk; ++it) {
        const int ahead = min(nk - 1 - it, NST - 2);
        if (ahead == 0) wait_vm<0>();                        // all but the `ahead` youngest stages have landed
        else if (ahead == 1) wait_vm<NPS>();
        else if (ahead == 2) wait_vm<2 * NPS>();
        else if (ahead == 3) wait_vm<3 * NPS>();
        else wait_vm<4 * NPS>();
        __builtin_amdgcn_s_barrier();                        // stage `it` visible to every wave; slot of stage it-1 free again
        if (it + NST - 1 < nk) issue(it + NST - 1);
        const char* sa = smem + cslot * STAGE;
        const char* sb = sa + OA::BYTES;
        cslot = cslot + 1 == NST ? 0 : cslot + 1;
        // fragment reads run one k-step ahead of the MFMAs that consume them
        Vec8<bf16> fa[2][MI], fb[2][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[0][i] = oa.frag(sa, i, 0, half);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = ob.frag(sb, j, 0, half);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk < 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[nxt][i] = oa.frag(sa, i, kk + 1, half);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[nxt][j] = ob.frag(sb, j, kk + 1, half);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<bf16>::run(acc[i][j], fa[cur][i], fb[cur][j]);
        }
    }

    epilogue<bf16, MI>(g, acc, smem, bm + wm, bn + wn, lane, wave);
}

template <bool TA, bool TB, int MI, int NST>
int launch_ring(const GemmArgs& g, hipStream_t s) {
    constexpr size_t stage = RingOp<TA, 64 * MI>::BYTES + RingOp<TB, 128>::BYTES;
    const size_t lds = max((size_t)NST * stage, EPI_LDS);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<TA, TB, MI, NST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, 64 * MI) * ceil_div(g.N, BN);
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_ring_kernel<%s, %s, %d, %d>", TF(TA), TF(TB), MI, NST); }
    hipLaunchKernelGGL((gemm_ring_kernel<TA, TB, MI, NST>), dim3(tiles, 1, g.ksplit), dim3(256), lds, s, g);
    ASSL_LAUNCH_CHECK();
}

template <int MI, int NST>
int dispatch_ring(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_ring<false, false, MI, NST>(g, s);
    if (!ta && tb) return launch_ring<false, true, MI, NST>(g, s);
    if (ta && tb) return launch_ring<true, true, MI, NST>(g, s);
    return launch_ring<true, false, MI, NST>(g, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 output tile, 8 waves (2 x 4, wave tile 128 x 64), K-tile 64, eight phases per pair of K-tiles: the
// hand-scheduled main loop.  What bounded the kernels above was the loop SHAPE (one fat k-step: issue loads, read fragments,
// MFMA, wait for everything, barrier): 31 % MFMA-busy on 6144 x 2048 x 2048, waves parked at s_waitcnt / the barrier.
// Here
//   * a K-tile is staged as FOUR 16 KB parts - A(r0), B(c0), B(c1), A(r1): the 64-row halves of every wave's 128 rows and
//     the 32-column halves of every wave's 64 columns - by direct-to-LDS DMA, two 1 KB instructions per wave and part;
//   * a phase = one quadrant (64 x 32 x K 64 = 8 MFMA 32x32x16) of every wave's tile; it issues ONE part of a later K-tile,
//     reads only the fragments that are new for its quadrant (A(r0) + B(c0), then B(c1), then A(r1), then nothing), runs
//     its MFMAs under s_setprio, waits with a COUNTED vmcnt that leaves five parts (80 KB) in flight, and meets the other
//     waves at ONE raw s_barrier;
//   * a part is re-staged no earlier than the phase after the barrier that followed its last fragment read (WAR), and is read
//     no earlier than the phase after the barrier that followed the wait that retired it (RAW): with the issue order
//     A(r0) B(c0) B(c1) A(r1) per K-tile and the consumption order the same, "everything but the five youngest parts has
//     landed" is exactly what each phase needs, so the wait is the same vmcnt(10) in every steady-state phase.
// Two buffers x four parts = 128 KB of LDS; accumulators 128 VGPRs + 64 of fragments.  Operand layouts as the ring kernel
// (K-contiguous: [row][64 k] image, 16-byte chunk ^= (row >> 1) & 7; row-contiguous: [64 k][128 rows] image read with
// ds_read_b64_tr_b16, 64-byte group ^= k & 3), the swizzle applied to the DMA's source address.  K % 64 == 0.
// GROUP = rows of ONE wave row (column) inside a part, STRIDE = tile rows between consecutive wave rows: part-row p of part X is
// tile row (p / GROUP) * STRIDE + X * GROUP + p % GROUP.  256 x 256 kernel: A <64, 128>, B <32, 64>; 256 x 128 kernel: A <32, 64>, B <64, 64>.
template <bool TRANS, int GROUP, int STRIDE>
struct PartOp {
    static constexpr int NBLK = GROUP / 32;                  // 32-row fragment blocks of this wave inside a part
    // DMA = buffer_load_dwordx4 ... lds: 32-bit per-lane byte offsets (two instructions per part), everything that is uniform -
    // tile origin, K-tile, part - in the scalar offset; rows / columns past the matrix read zeros or other valid elements of the
    // buffer (the descriptor bounds the access), their results are never stored
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[2];
    int kstride, part_delta, base_off;                       // bytes; wave-uniform
    int foff[NBLK], fsw[NBLK];
    __device__ __forceinline__ void init(const void* base_, unsigned bytes, long ld, int r0, int k0, int wave, int lane, int wsel) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_), 0, bytes, 0x00020000);
        const int w0 = wsel * GROUP;                         // this wave's first part-row
        if (!TRANS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = wave * 16 + i * 8 + (lane >> 3);
                const int sc = (lane & 7) ^ ((p >> 1) & 7);
                const int row = (p / GROUP) * STRIDE + (p % GROUP);
                voff[i] = (int)(((long)row * ld + sc * 8) * 2);
            }
            kstride = GBK * 2;
            part_delta = (int)(GROUP * ld * 2);
            base_off = (int)(((long)r0 * ld + k0) * 2);
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const int p = w0 + b * 32 + (lane & 31);
                foff[b] = p * 128; fsw[b] = (p >> 1) & 7;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kl = wave * 8 + i * 4 + (lane >> 4), pc = lane & 15;
                const int roff = ((((pc >> 2) ^ (kl & 3)) << 2) + (pc & 3)) * 8;
                const int row = (roff / GROUP) * STRIDE + (roff % GROUP);
                voff[i] = (int)(((long)kl * ld + row) * 2);
            }
            kstride = (int)(GBK * ld * 2);
            part_delta = GROUP * 2;
            base_off = (int)(((long)k0 * ld + r0) * 2);
            const int q = (lane & 15) >> 2, pp = lane & 3, h4 = (lane >> 4) & 1, half = lane >> 5;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const int grp = ((w0 + b * 32) >> 5) ^ q;
                foff[b] = (8 * half + q) * 256 + grp * 64 + 32 * h4 + 8 * pp; fsw[b] = 0;
            }
        }
    }
    __device__ __forceinline__ void issue(int part, int st, char* dst, bool live) const {
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int so = live ? base_off + st * kstride + part * part_delta : 0x7FFFFF00;        // past the buffer: zeros, no traffic
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, voff[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + 1024), 16, voff[1], so, 0, 0);
    }
    __device__ __forceinline__ Vec8<bf16> frag(const char* img, int b, int kk, int half) const {
        if (!TRANS) return Vec8<bf16>::load(reinterpret_cast<const bf16*>(img + foff[b] + (((kk * 2 + half) ^ fsw[b]) << 4)));
        // The transposing reads go through inline asm: behind an LDS-DMA in flight hipcc waits vmcnt(0) before a
        // ds_read_b64_tr_b16 BUILTIN (it treats the DMA as a store the read may alias), which drained the pipeline in every
        // phase.  The caller's s_waitcnt lgkmcnt(0) + sched_barrier after the section's barrier covers these reads.
        typedef __attribute__((address_space(3))) const char* lds_t;
        const unsigned a = (unsigned)(size_t)(lds_t)(img + foff[b] + kk * 16 * 256);
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a) : "memory");
        Vec8<bf16> f;
        f.v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    }
};

constexpr int P8_PART = 16384, P8_BUF = 4 * P8_PART;         // slot order inside a buffer: A(r0) B(c0) B(c1) A(r1)

// Workgroup ids are dealt to the 8 XCDs round-robin (id mod 8): give every XCD a CONTIGUOUS run of tiles, so that the tiles of a
// row (one A panel) and neighbouring rows meet in one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_contiguous(int id, int nwg) {
    const int xcd = id & 7, q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
}

template <bool TA, bool TB, int DBG = 0>
__device__ __forceinline__ void p8_body(const GemmArgs& g, char* smem, int wg) {
    const int tiles_n = (g.N + 255) / 256;
    const int bm = (wg / tiles_n) * 256, bn = (wg % tiles_n) * 256;
    const int ksteps = g.K / GBK;
    const int per = (ksteps + g.ksplit - 1) / g.ksplit;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);
    if (ks0 >= ks1) return;
    const int nk = ks1 - ks0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;

    PartOp<TA, 64, 128> oa; PartOp<TB, 32, 64> ob;
    oa.init(g.A, g.a_bytes, g.lda, bm, ks0 * GBK, wave, lane, wr);
    ob.init(g.B, g.b_bytes, g.ldb, bn, ks0 * GBK, wave, lane, wc);
    char* const wdst = smem + wave * 2048;                   // this wave's 2 KB slice of every part image

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // slot: 0 A(r0), 1 B(c0), 2 B(c1), 3 A(r1).  A part of a K-tile past the end is still "issued" (branch-free, the counted
    // waits stay uniform) but with a scalar offset past the buffer: the descriptor turns it into a no-traffic load of zeros
    auto issue = [&](int slot, int t, int buf) {
        char* d = wdst + buf * P8_BUF + slot * P8_PART;
        const bool live = t < nk && !((DBG & 2) && t >= 2);
        if (slot == 0) oa.issue(0, t, d, live); else if (slot == 3) oa.issue(1, t, d, live);
        else ob.issue(slot - 1, t, d, live);
    };
    // A phase has two sections, each closed by a raw s_barrier: LOAD (issue one part, read this quadrant's new fragments, wait
    // until everything but the five youngest parts has landed and the fragment reads have returned) and MFMA.  The two waves
    // of a SIMD (wave rows wr = 0 / 1) run ONE SECTION APART - wr = 1 passes an extra barrier up front, wr = 0 one at the end -
    // so while one of them issues MFMAs the other issues its DMA and LDS reads instead of both doing the same thing at once
    // (in lockstep the MFMA pipe idled during every load section: 67.7 us on 6144 x 2048 x 2048 against 45 us of MFMA sections alone).
    // RAW / WAR as above: a part read in phase p + 1 was retired by every wave's wait in its LOAD section of phase p, and the
    // later of those sections is followed by a barrier before the earlier reader starts; a part is re-staged at least one phase
    // after its last read, i.e. after a barrier that follows the later group's (completed: lgkmcnt(0)) reads.
    auto load_end = [&]() {
        wait_vm<8>();                                        // all but the four youngest parts have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this section's fragment reads (they returned while waiting at the barrier)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync = [&]() {                                      // end of the MFMA section
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    Vec8<bf16> fa[2][4], fb0[4], fb1[4];
    if (DBG & 4) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { fa[0][kk] = Vec8<bf16>::zero(); fa[1][kk] = Vec8<bf16>::zero(); fb0[kk] = Vec8<bf16>::zero(); fb1[kk] = Vec8<bf16>::zero(); }
    }
    auto loadA = [&](const char* img) {
        if ((DBG & 4) && nk > 0) return;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fa[b][kk] = oa.frag(img, b, kk, half);
    };
    auto loadB = [&](const char* img, Vec8<bf16> (&fb)[4]) {
        if ((DBG & 4) && nk > 0) return;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb[kk] = ob.frag(img, 0, kk, half);
    };
    auto quad = [&](int X, int Y, const Vec8<bf16> (&fb)[4]) {
        if (DBG & 1) return;
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int b = 0; b < 2; ++b) Mma<bf16>::run(acc[2 * X + b][Y], fa[b][kk], fb[kk]);
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(0);
    };

    // Issue order = consumption order (per K-tile A(r0) B(c0) B(c1) A(r1)), one part per phase, each part re-staged TWO phases
    // after its last fragment read (those reads are only known complete after the barrier that closes their LOAD section) and five
    // to six phases before its first: "all but the four youngest parts landed" (vmcnt(8)) is what every phase needs.
    // prologue: the first K-tile whole, two parts of the second
    issue(0, 0, 0); issue(1, 0, 0); issue(2, 0, 0); issue(3, 0, 0);
    issue(0, 1, 1); issue(1, 1, 1);
    wait_vm<8>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // the second wave of every SIMD: one section behind
    const char* const E = smem;
    const char* const O = smem + P8_BUF;
    for (int t = 0; t < nk; t += 2) {
        // ---- K-tile t (buffer E)
        loadA(E); loadB(E + P8_PART, fb0);
        issue(2, t + 1, 1);                                  // B(c1) of the odd tile of this pair
        load_end();
        quad(0, 0, fb0);
        sync();
        loadB(E + 2 * P8_PART, fb1);
        issue(3, t + 1, 1);                                  // A(r1) of the odd tile
        load_end();
        quad(0, 1, fb1);
        sync();
        loadA(E + 3 * P8_PART);
        issue(0, t + 2, 0);
        load_end();
        quad(1, 1, fb1);
        sync();
        issue(1, t + 2, 0);
        load_end();
        quad(1, 0, fb0);
        sync();
        // ---- K-tile t + 1 (buffer O); past the end its parts are zeros: the MFMAs add nothing
        loadA(O); loadB(O + P8_PART, fb0);
        issue(2, t + 2, 0);
        load_end();
        quad(0, 0, fb0);
        sync();
        loadB(O + 2 * P8_PART, fb1);
        issue(3, t + 2, 0);
        load_end();
        quad(0, 1, fb1);
        sync();
        loadA(O + 3 * P8_PART);
        issue(0, t + 3, 1);
        load_end();
        quad(1, 1, fb1);
        sync();
        issue(1, t + 3, 1);
        load_end();
        quad(1, 0, fb0);
        sync();
    }
    if (wr == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    wait_vm<0>();
    epilogue<bf16, 2>(g, *reinterpret_cast<f32x16 (*)[2][2]>(&acc[0]), smem, bm + wr * 128, bn + wc * 64, lane, wave);
    epilogue<bf16, 2>(g, *reinterpret_cast<f32x16 (*)[2][2]>(&acc[2]), smem, bm + wr * 128 + 64, bn + wc * 64, lane, wave);
}

template <bool TA, bool TB, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    p8_body<TA, TB, DBG>(g, smem, xcd_contiguous(blockIdx.x, gridDim.x));
}

template <bool TA, bool TB, int DBG = 0>
int launch_p8(const GemmArgs& g, hipStream_t s) {
    const size_t lds = max((size_t)2 * P8_BUF, EPI_LDS * 2);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_kernel<TA, TB, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, 256) * ceil_div(g.N, 256);
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_p8_kernel<%s, %s, %d>", TF(TA), TF(TB), DBG); }
    hipLaunchKernelGGL((gemm_p8_kernel<TA, TB, DBG>), dim3(tiles, 1, g.ksplit), dim3(512), lds, s, g);
    ASSL_LAUNCH_CHECK();
}
template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm_p8_multi_kernel(GemmMulti gm) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    int wg;
    const int p = multi_locate(gm, wg);
    if (p < 0) return;
    p8_body<TA, TB, 0>(gm.p[p], smem, wg);
}
// the same kernel under its own name when the epilogue applies the optimiser's update (audiossl_gemm_multi_sgd): a launch that moves
// 16 + 2 bytes per result element on top of the GEMM shows up as what it is in a profile
template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm_p8_multi_sgd_kernel(GemmMulti gm) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    int wg;
    const int p = multi_locate(gm, wg);
    if (p < 0) return;
    p8_body<TA, TB, 0>(gm.p[p], smem, wg);
}
template <bool TA, bool TB>
int launch_p8_multi(const GemmMulti& gm_, int count, int ksplit, hipStream_t s) {
    GemmMulti gm = gm_;
    const int grid = multi_grid(gm, count, 256, 256);
    const size_t lds = max((size_t)2 * P8_BUF, EPI_LDS * 2);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_multi_kernel<TA, TB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_multi_sgd_kernel<TA, TB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    if (gm.p[0].sgd_p) {
        { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_p8_multi_sgd_kernel<%s, %s>", TF(TA), TF(TB)); }
        hipLaunchKernelGGL((gemm_p8_multi_sgd_kernel<TA, TB>), dim3(grid, 1, ksplit), dim3(512), lds, s, gm);
        ASSL_LAUNCH_CHECK();
    }
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_p8_multi_kernel<%s, %s>", TF(TA), TF(TB)); }
    hipLaunchKernelGGL((gemm_p8_multi_kernel<TA, TB>), dim3(grid, 1, ksplit), dim3(512), lds, s, gm);
    ASSL_LAUNCH_CHECK();
}
int dispatch_p8_multi(const GemmMulti& gm, int count, int ksplit, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_p8_multi<false, false>(gm, count, ksplit, s);
    if (!ta && tb) return launch_p8_multi<false, true>(gm, count, ksplit, s);
    if (ta && tb) return launch_p8_multi<true, true>(gm, count, ksplit, s);
    return launch_p8_multi<true, false>(gm, count, ksplit, s);
}

int dispatch_p8(const GemmArgs& g, int ta, int tb, hipStream_t s) {
#ifdef AUDIOSSL_ABLATE
    static const int dbg = getenv("AUDIOSSL_GEMM_P8_DBG") ? atoi(getenv("AUDIOSSL_GEMM_P8_DBG")) : 0;   // ablation (wrong results)
#else
    constexpr int dbg = 0;            // the ablation instantiations exist only in -DAUDIOSSL_ABLATE builds (python audio-ssl_amd/build.py --ablate)
#endif
    if (!ta && !tb && dbg) {
        switch (dbg) {
            case 1: return launch_p8<false, false, 1>(g, s);      // no MFMA
            case 2: return launch_p8<false, false, 2>(g, s);      // no DMA after the first two K-tiles
            case 4: return launch_p8<false, false, 4>(g, s);      // no fragment reads
            case 6: return launch_p8<false, false, 6>(g, s);      // MFMA + barriers only
            case 8: return launch_p8<false, false, 8>(g, s);      // no s_setprio
            default: break;
        }
    }
    if (!ta && !tb) return launch_p8<false, false>(g, s);
    if (!ta && tb) return launch_p8<false, true>(g, s);
    if (ta && tb) return launch_p8<true, true>(g, s);
    return launch_p8<true, false>(g, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 128 output tile, 8 waves (4 x 2, wave tile 32 x 64): the same loop for the M = 512 problems (the heads' dzn / last data
// gradient launches: 96 tiles of 256 x 128, 192 of 128 x 128) and single problems with a narrow output.  A K-tile is TWO 16 KB parts
// (A: every wave's 32 rows, B: every wave's 64 columns) and ONE phase (8 MFMA 32x32x16 per wave); K-tile t + 3 is staged in
// phase t into a ring of five buffers (160 KB: the buffer it replaces was last read in phase t - 2, two phases back as the WAR rule
// asks), and the wait at the end of phase t's LOAD section retires K-tile t + 1 (issued two phases earlier): all but the two
// youngest K-tiles = vmcnt(8).  More LDS read bytes per MFMA than the 256-row kernels (12 fragment reads per 8 MFMAs).
constexpr int P5_PART = 16384, P5_BUF = 2 * P5_PART, P5_NBUF = 5, P5_AHEAD = 3;

template <bool TA, bool TB, int SCHED>
__device__ __forceinline__ void p5_body(const GemmArgs& g, char* smem, int wg) {
    const int tiles_n = (g.N + 127) / 128;
    const int bm = (wg / tiles_n) * 128, bn = (wg % tiles_n) * 128;
    const int ksteps = g.K / GBK;
    const int per = (ksteps + g.ksplit - 1) / g.ksplit;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);
    if (ks0 >= ks1) return;
    const int nk = ks1 - ks0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    PartOp<TA, 32, 32> oa; PartOp<TB, 64, 64> ob;
    oa.init(g.A, g.a_bytes, g.lda, bm, ks0 * GBK, wave, lane, wr);
    ob.init(g.B, g.b_bytes, g.ldb, bn, ks0 * GBK, wave, lane, wc);
    char* const wdst = smem + wave * 2048;

    f32x16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    auto issue = [&](int t, int boff) {
        const bool live = t < nk;
        oa.issue(0, t, wdst + boff, live);
        ob.issue(0, t, wdst + boff + P5_PART, live);
    };
    Vec8<bf16> fa[4], fb[2][4];
    issue(0, 0); issue(1, P5_BUF); issue(2, 2 * P5_BUF);
    wait_vm<8>();                                            // K-tile 0 has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave >= 4) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    int cb = 0, ib = P5_AHEAD * P5_BUF;
    for (int t = 0; t < nk; ++t) {
        const char* const img = smem + cb;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fa[kk] = oa.frag(img, 0, kk, half);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fb[j][kk] = ob.frag(img + P5_PART, j, kk, half);
        if (SCHED == 0) { issue(t + P5_AHEAD, ib); wait_vm<8>(); }
        else wait_vm<4>();                                   // K-tile t + 3 is issued inside the MFMA section below
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) Mma<bf16>::run(acc[0][j], fa[kk], fb[j][kk]);
            if (SCHED == 1 && kk == 0) oa.issue(0, t + P5_AHEAD, wdst + ib, t + P5_AHEAD < nk);
            if (SCHED == 1 && kk == 2) ob.issue(0, t + P5_AHEAD, wdst + ib + P5_PART, t + P5_AHEAD < nk);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        cb = cb + P5_BUF == P5_NBUF * P5_BUF ? 0 : cb + P5_BUF;
        ib = ib + P5_BUF == P5_NBUF * P5_BUF ? 0 : ib + P5_BUF;
    }
    if (wave < 4) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    wait_vm<0>();
    epilogue<bf16, 1>(g, acc, smem, bm + wr * 32, bn + wc * 64, lane, wave);
}

// KIND 5 = the 128 x 128 tiles of p5_body (DMA issued inside the MFMA section: 25.1 -> 22.4 us on the M = 512 launches).  The
// 256 x 128 tiles that used to be KIND 6 run in the software-pipelined form below (SpBody, shape 6).
template <bool TA, bool TB, int KIND>
__device__ __forceinline__ void pk_body(const GemmArgs& g, char* smem, int wg) {
    static_assert(KIND == 5, "the sectioned loop remains for the 128 x 128 tiles only");
    p5_body<TA, TB, 1>(g, smem, wg);
}

template <bool TA, bool TB, int KIND>
__global__ __launch_bounds__(512) void gemm_pk_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    pk_body<TA, TB, KIND>(g, smem, xcd_contiguous(blockIdx.x, gridDim.x));
}
template <bool TA, bool TB, int KIND>
__global__ __launch_bounds__(512) void gemm_pk_multi_kernel(GemmMulti gm) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    int wg;
    const int p = multi_locate(gm, wg);
    if (p < 0) return;
    pk_body<TA, TB, KIND>(gm.p[p], smem, wg);
}
template <int KIND> constexpr size_t pk_lds() {
    return (size_t)P5_NBUF * P5_BUF;
}
template <int KIND> constexpr int pk_rows() { return 128; }

template <bool TA, bool TB, int KIND>
int launch_pk(const GemmArgs& g, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pk_kernel<TA, TB, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pk_lds<KIND>()) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, pk_rows<KIND>()) * ceil_div(g.N, 128);
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_pk_kernel<%s, %s, %d>", TF(TA), TF(TB), KIND); }
    hipLaunchKernelGGL((gemm_pk_kernel<TA, TB, KIND>), dim3(tiles, 1, g.ksplit), dim3(512), pk_lds<KIND>(), s, g);
    ASSL_LAUNCH_CHECK();
}
template <bool TA, bool TB, int KIND>
int launch_pk_multi(const GemmMulti& gm_, int count, int ksplit, hipStream_t s) {
    GemmMulti gm = gm_;
    const int grid = multi_grid(gm, count, pk_rows<KIND>(), 128);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pk_multi_kernel<TA, TB, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pk_lds<KIND>()) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_pk_multi_kernel<%s, %s, %d>", TF(TA), TF(TB), KIND); }
    hipLaunchKernelGGL((gemm_pk_multi_kernel<TA, TB, KIND>), dim3(grid, 1, ksplit), dim3(512), pk_lds<KIND>(), s, gm);
    ASSL_LAUNCH_CHECK();
}
template <int KIND>
int dispatch_pk(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_pk<false, false, KIND>(g, s);
    if (!ta && tb) return launch_pk<false, true, KIND>(g, s);
    if (ta && tb) return launch_pk<true, true, KIND>(g, s);
    return launch_pk<true, false, KIND>(g, s);
}
template <int KIND>
int dispatch_pk_multi(const GemmMulti& gm, int count, int ksplit, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_pk_multi<false, false, KIND>(gm, count, ksplit, s);
    if (!ta && tb) return launch_pk_multi<false, true, KIND>(gm, count, ksplit, s);
    if (ta && tb) return launch_pk_multi<true, true, KIND>(gm, count, ksplit, s);
    return launch_pk_multi<true, false, KIND>(gm, count, ksplit, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// The SOFTWARE-PIPELINED family: the hand-scheduled tiles above without LOAD / MFMA sections and without the wave-group stagger.
// A step = one 16-deep slice of the contraction: the wave's 2 MI MFMAs (MI x 2 blocks of 32 x 32) on one register set of MI + 2
// fragments, while the fragments of the NEXT step are read into the other set - two reads in each gap between MFMAs, where the LDS
// array serves them at no cost to the MFMA pipe - and the DMA instructions that stage a later slice follow in the remaining gaps.
// A stage = BK / 16 steps = one [rows][BK] slice of both operands, kept as 128-row sub-images in a ring of NB stages.  ONE
// s_barrier per stage, at the top of its last step, behind the wait that retired that step's own fragments (the wave's last reads
// of the stage):
//   * after it every wave is done reading stage h -> its ring slot takes stage h + NB (WAR), issued in this step and the following
//     BK / 16 - 1 steps;
//   * before it every wave waited for all of its DMA but the NB - 2 youngest stages -> stage h + 1, which the reads issued from
//     here on address, has landed in every wave's slice of the sub-images (RAW).
// Measured on the three-head launches (M = 1,024, K = 2,048, 256 x 128 tiles): 35.1 -> 31.8 us against the sectioned loop.
// Sub-image layouts: K-contiguous operand: [128 rows][BK] with the 16-byte chunk c of row r stored at c ^ swz(r), swz = (r >> 2) & 3
// for BK = 32 (64-byte rows) and (r >> 1) & 7 for BK = 64: the 16 lanes the LDS serves together (ds_read_b128) hit all 64 banks
// once; row-contiguous operand: [BK k][128 rows], 64-byte group ^= k & 3, read with ds_read_b64_tr_b16 (as PartOp above).
template <bool TRANS, int BK>
struct SliceOp {
    static constexpr int NI = BK / 32;                       // DMA instructions per wave and sub-image (1 KB each)
    static constexpr int SUBB = 128 * BK * 2;                // bytes of a sub-image
    static constexpr int CH = BK / 8;                        // 16-byte chunks per row (K-contiguous form)
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[NI];
    int kstride, sub_delta, base_off;                        // bytes; wave-uniform
    __device__ __forceinline__ static int swz(int r) { return BK == 32 ? (r >> 2) & 3 : (r >> 1) & 7; }
    __device__ __forceinline__ void init(const void* base_, unsigned bytes, long ld, int r0, int k0, int wave, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_), 0, bytes, 0x00020000);
        if (!TRANS) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int p = wave * 16 + i * (64 / CH) + lane / CH;          // row of the sub-image this lane fills
                const int sc = (lane % CH) ^ swz(p);                          // the logical chunk stored at position lane % CH
                voff[i] = (int)(((long)p * ld + sc * 8) * 2);
            }
            kstride = BK * 2;
            sub_delta = (int)(128 * ld * 2);
            base_off = (int)(((long)r0 * ld + k0) * 2);
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int kl = wave * (BK / 8) + i * 4 + (lane >> 4), pc = lane & 15;
                const int roff = ((((pc >> 2) ^ (kl & 3)) << 2) + (pc & 3)) * 8;
                voff[i] = (int)(((long)kl * ld + roff) * 2);
            }
            kstride = (int)(BK * ld * 2);
            sub_delta = 256;
            base_off = (int)(((long)k0 * ld + r0) * 2);
        }
    }
    // DMA instruction i of sub-image `sub` of stage st; dst = this wave's slice (wave * 1024 * NI) of that sub-image
    __device__ __forceinline__ void issue(int sub, int i, int st, char* dst, bool live) const {
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int so = live ? base_off + st * kstride + sub * sub_delta : 0x7FFFFF00;          // past the buffer: zeros, no traffic
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + i * 1024), 16, voff[i], so, 0, 0);
    }
    // byte offset (inside a stage's run of this operand's sub-images) of the step-0 fragment of the 32-row block at tile row row0
    __device__ __forceinline__ static int frag_off(int row0, int lane) {
        if (!TRANS) {
            const int local = (row0 & 127) + (lane & 31), half = lane >> 5;
            return (row0 >> 7) * SUBB + local * (BK * 2) + ((half ^ swz(local)) << 4);
        }
        const int gb = row0 >> 5, q = (lane & 15) >> 2, pp = lane & 3, h4 = (lane >> 4) & 1, half = lane >> 5;
        return (gb >> 2) * SUBB + (8 * half + q) * 256 + (((gb & 3) ^ q) << 6) + 32 * h4 + 8 * pp;
    }
    __device__ __forceinline__ static Vec8<bf16> frag(const char* stage, int off, int kk) {
        if (!TRANS) return Vec8<bf16>::load(reinterpret_cast<const bf16*>(stage + (off ^ (kk << 5))));
        typedef __attribute__((address_space(3))) const char* lds_t;                            // inline asm: see PartOp::frag
        const unsigned a = (unsigned)(size_t)(lds_t)(stage + off + kk * 4096);
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a) : "memory");
        Vec8<bf16> f;
        f.v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return f;
    }
};

// SHAPE 8: 256 x 256 (waves 2 x 4, wave tile 128 x 64), BK 32, ring of 5 x 32 KB; SHAPE 6: 256 x 128 (4 x 2, 64 x 64), BK 64, ring of
// 3 x 48 KB; SHAPE 5: 128 x 128 (4 x 2, 32 x 64), BK 64, ring of 5 x 32 KB
template <int SHAPE> struct SpCfg;
template <> struct SpCfg<8> { static constexpr int WR = 2, WC = 4, MI = 4, BK = 32, NB = 5; };
template <> struct SpCfg<6> { static constexpr int WR = 4, WC = 2, MI = 2, BK = 64, NB = 3; };
template <> struct SpCfg<5> { static constexpr int WR = 4, WC = 2, MI = 1, BK = 64, NB = 5; };
template <int SHAPE> struct SpDim {
    using C = SpCfg<SHAPE>;
    static constexpr int RA = C::WR * C::MI * 32, RB = C::WC * 64;
    static constexpr int SA = RA / 128, SBN = RB / 128, NSUB = SA + SBN;
    static constexpr int SUBB = 128 * C::BK * 2, STG = NSUB * SUBB;
    static constexpr int PER = NSUB * (C::BK / 32);          // DMA instructions per wave and stage
    static constexpr int KS = C::BK / 16;                    // steps per stage
    static constexpr int DPS = (PER + KS - 1) / KS;          // DMA instructions per step
    static constexpr size_t LDS = (size_t)C::NB * STG > EPI_LDS * 2 ? (size_t)C::NB * STG : EPI_LDS * 2;
};

// (a class with a static member: as a function template the host pass of hipcc rejected every instantiation after the first)
template <bool TA, bool TB, int SHAPE>
struct SpBody {
__device__ __forceinline__ static void run(const GemmArgs& g, char* smem, int wg) {
    using C = SpCfg<SHAPE>; using D = SpDim<SHAPE>;
    constexpr int MI = C::MI, BK = C::BK, NB = C::NB, KS = D::KS, NI = BK / 32, NM = 2 * MI, NR = MI + 2;
    static_assert(C::WR * C::WC == 8 && D::RA % 128 == 0 && D::RB % 128 == 0 && KS % 2 == 0, "shape");
    constexpr int first_dma_gap = NM - 1 > (NR + 1) / 2 ? (NR + 1) / 2 : NM - 2;     // the gap after the last pair of fragment reads
    const int tiles_n = (g.N + D::RB - 1) / D::RB;
    const int bm = (wg / tiles_n) * D::RA, bn = (wg % tiles_n) * D::RB;
    const int k64 = g.K / GBK;
    const int per = (k64 + g.ksplit - 1) / g.ksplit;
    const int ks0 = blockIdx.z * per, ks1 = min(k64, ks0 + per);
    if (ks0 >= ks1) return;
    const int nk = (ks1 - ks0) * (GBK / BK);                 // stages
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / C::WC, wc = wave % C::WC;

    SliceOp<TA, BK> oa; SliceOp<TB, BK> ob;
    oa.init(g.A, g.a_bytes, g.lda, bm, ks0 * GBK, wave, lane);
    ob.init(g.B, g.b_bytes, g.ldb, bn, ks0 * GBK, wave, lane);
    char* const wdst = smem + wave * (1024 * NI);            // this wave's slice of every sub-image

    int foff[NR];                                            // step-0 fragment offsets inside a stage: MI row blocks, 2 column blocks
#pragma unroll
    for (int i = 0; i < MI; ++i) foff[i] = oa.frag_off(wr * MI * 32 + i * 32, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) foff[MI + j] = D::SA * D::SUBB + ob.frag_off(wc * 64 + j * 32, lane);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA instruction n (0 .. PER - 1) of stage st into the ring slot at byte offset sbase
    auto dma = [&](int n, int st, int sbase) {
        const int sub = n / NI, i = n % NI;
        const bool live = st < nk;
        if (sub < D::SA) oa.issue(sub, i, st, wdst + sbase + sub * D::SUBB, live);
        else ob.issue(sub - D::SA, i, st, wdst + sbase + sub * D::SUBB, live);
    };
    Vec8<bf16> fr[2][NR];                                    // [register set][fragment]
    auto read = [&](int set, int n, const char* stage, int kk) {
        if (n < MI) fr[set][n] = oa.frag(stage, foff[n], kk);
        else fr[set][n] = ob.frag(stage, foff[n], kk);
    };

    // prologue: stages 0 .. NB - 2 whole and the first step's share of stage NB - 1
#pragma unroll
    for (int h = 0; h < NB - 1; ++h)
#pragma unroll
        for (int n = 0; n < D::PER; ++n) dma(n, h, h * D::STG);
#pragma unroll
    for (int n = 0; n < D::DPS && n < D::PER; ++n) dma(n, NB - 1, (NB - 1) * D::STG);
    wait_vm<(NB - 2) * D::PER + (D::DPS < D::PER ? D::DPS : D::PER)>();       // stage 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int n = 0; n < NR; ++n) read(0, n, smem, 0);

    int cb = 0, pb = (NB - 1) * D::STG;                      // ring slots of stage h and of stage h - 1
    for (int h = 0; h < nk; ++h) {
        const int nb = cb + D::STG == NB * D::STG ? 0 : cb + D::STG;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            const bool last = kk == KS - 1;
            const char* const src = smem + (last ? nb : cb);
            const int kn = last ? 0 : kk + 1;
            // DMA share of this step: the slot freed by the most recent barrier takes stage (that stage) + NB
            const int c = last ? 0 : kk + 1, dst_stage = last ? h + NB : h - 1 + NB, dst_slot = last ? cb : pb;
            // the transposing reads are inline asm the compiler does not track; the K-contiguous ones get its own counted waits
            if (TA || TB || last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (last) {
                wait_vm<(NB - 2) * D::PER>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                Mma<bf16>::run(acc[m >> 1][m & 1], fr[cur][m >> 1], fr[cur][MI + (m & 1)]);
                __builtin_amdgcn_sched_barrier(0);
                if (m < NM - 1) {
                    // gap m: two fragment reads, then (behind the reads) the step's DMA instructions, all of them by the last gap
#pragma unroll
                    for (int n = 2 * m; n < 2 * m + 2 && n < NR; ++n) read(nxt, n, src, kn);
#pragma unroll
                    for (int d = 0; d < D::DPS; ++d) {
                        constexpr int gaps = NM - 1 - first_dma_gap, per_gap = (D::DPS + gaps - 1) / gaps;
                        const int gap = first_dma_gap + d / per_gap;
                        if (gap == m && c * D::DPS + d < D::PER) dma(c * D::DPS + d, dst_stage, dst_slot);
                    }
                    if (NM == 2) {                           // one gap only: the third fragment too
#pragma unroll
                        for (int n = 2; n < NR; ++n) read(nxt, n, src, kn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        pb = cb; cb = nb;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (MI == 4) {
        epilogue<bf16, 2>(g, *reinterpret_cast<f32x16 (*)[2][2]>(&acc[0]), smem, bm + wr * 128, bn + wc * 64, lane, wave);
        epilogue<bf16, 2>(g, *reinterpret_cast<f32x16 (*)[2][2]>(&acc[2]), smem, bm + wr * 128 + 64, bn + wc * 64, lane, wave);
    } else epilogue<bf16, MI>(g, acc, smem, bm + wr * MI * 32, bn + wc * 64, lane, wave);
}
};

template <bool TA, bool TB, int SHAPE>
__global__ __launch_bounds__(512) void gemm_sp_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    SpBody<TA, TB, SHAPE>::run(g, smem, xcd_contiguous(blockIdx.x, gridDim.x));
}
template <bool TA, bool TB, int SHAPE>
__global__ __launch_bounds__(512) void gemm_sp_multi_kernel(GemmMulti gm) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    int wg;
    const int p = multi_locate(gm, wg);
    if (p < 0) return;
    SpBody<TA, TB, SHAPE>::run(gm.p[p], smem, wg);
}
template <bool TA, bool TB, int SHAPE>
int launch_sp(const GemmArgs& g, hipStream_t s) {
    using D = SpDim<SHAPE>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sp_kernel<TA, TB, SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)D::LDS) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, D::RA) * ceil_div(g.N, D::RB);
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_sp_kernel<%s, %s, %d>", TF(TA), TF(TB), SHAPE); }
    hipLaunchKernelGGL((gemm_sp_kernel<TA, TB, SHAPE>), dim3(tiles, 1, g.ksplit), dim3(512), D::LDS, s, g);
    ASSL_LAUNCH_CHECK();
}
template <bool TA, bool TB, int SHAPE>
__global__ __launch_bounds__(512) void gemm_sp_multi_sgd_kernel(GemmMulti gm) {      // see gemm_p8_multi_sgd_kernel
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    int wg;
    const int p = multi_locate(gm, wg);
    if (p < 0) return;
    SpBody<TA, TB, SHAPE>::run(gm.p[p], smem, wg);
}
template <bool TA, bool TB, int SHAPE>
int launch_sp_multi(const GemmMulti& gm_, int count, int ksplit, hipStream_t s) {
    using D = SpDim<SHAPE>;
    GemmMulti gm = gm_;
    const int grid = multi_grid(gm, count, D::RA, D::RB);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sp_multi_kernel<TA, TB, SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)D::LDS) != hipSuccess) return ASSL_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sp_multi_sgd_kernel<TA, TB, SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)D::LDS) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    if (gm.p[0].sgd_p) {
        { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_sp_multi_sgd_kernel<%s, %s, %d>", TF(TA), TF(TB), SHAPE); }
        hipLaunchKernelGGL((gemm_sp_multi_sgd_kernel<TA, TB, SHAPE>), dim3(grid, 1, ksplit), dim3(512), D::LDS, s, gm);
        ASSL_LAUNCH_CHECK();
    }
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_sp_multi_kernel<%s, %s, %d>", TF(TA), TF(TB), SHAPE); }
    hipLaunchKernelGGL((gemm_sp_multi_kernel<TA, TB, SHAPE>), dim3(grid, 1, ksplit), dim3(512), D::LDS, s, gm);
    ASSL_LAUNCH_CHECK();
}
template <int SHAPE>
int dispatch_sp(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_sp<false, false, SHAPE>(g, s);
    if (!ta && tb) return launch_sp<false, true, SHAPE>(g, s);
    if (ta && tb) return launch_sp<true, true, SHAPE>(g, s);
    return launch_sp<true, false, SHAPE>(g, s);
}
template <int SHAPE>
int dispatch_sp_multi(const GemmMulti& gm, int count, int ksplit, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_sp_multi<false, false, SHAPE>(gm, count, ksplit, s);
    if (!ta && tb) return launch_sp_multi<false, true, SHAPE>(gm, count, ksplit, s);
    if (ta && tb) return launch_sp_multi<true, true, SHAPE>(gm, count, ksplit, s);
    return launch_sp_multi<true, false, SHAPE>(gm, count, ksplit, s);
}
// Shape 6 always runs in this form.  AUDIOSSL_GEMM_SP: bit mask that ALSO moves the 256 x 256 (bit 8) / 128 x 128 (bit 2) tiles to it.
// Measured (tools/heads_gemm_bench.py, tools/gemm_shapes.py; us, sectioned -> pipelined):
//   256 x 128: three-head M = 1,024 launches NT 35.1 -> 31.3 (822 TF/s), first layer 33.4 -> 30.0, NN 36.5 -> 35.3; single TN
//              2048 x 2048 x 1024 26.2 -> 23.6, x 512 19.4 -> 17.7                                              -> default
//   256 x 256: 6144 x 2048 x 2048 NT 63.0 -> 65.2, NN 64.2 -> 64.4, three-head TN 39.1 -> 38.4 (BK 32, ring of five; BK 64 with two
//              buffers and the whole stage issued in the last step: 64.7 / 65.1 / 41.2): the same within the box-to-box spread - the
//              sectioned kernel stays, these tiles are MFMA-paced either way
//   128 x 128: M = 512 launches 21.9 -> 22.6, 23.3 -> 24.2, 20.4 -> 21.4: two MFMAs per step leave one gap for three reads and a DMA
//              instruction - the sectioned kernel stays
//   BK 32 stages for the 256 x 128 / 128 x 128 tiles (rings of 6 / 10): 38.3 vs 31.5 and 25.4 vs 22.5 - twice the barriers
int sp_mask() {
    static const int m = getenv("AUDIOSSL_GEMM_SP") ? atoi(getenv("AUDIOSSL_GEMM_SP")) : 0;
    return m;
}

// BK = 32 with a register budget for THREE workgroups per CU (41 KB of LDS each; the epilogue runs on half-height tiles so
// that it fits the same 41 KB); AUDIOSSL_GEMM_BK32=0 disables, =2 also uses it for transposed-A problems.
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 3) void gemm_bk32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<bf16, TA, TB, 32, 2, 4, true>(g, smem, g.xcd_remap ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x);
}
template <bool TA, bool TB>
int launch_bk32(const GemmArgs& g, hipStream_t s) {
    constexpr size_t stage = sizeof(bf16) * 2 * (Stage<bf16, TA, 32, 128, 256>::LDS_ELEMS + Stage<bf16, TB, 32, BN, 256>::LDS_ELEMS);
    const size_t lds = stage > EPI_LDS / 2 ? stage : EPI_LDS / 2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bk32_kernel<TA, TB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, 128) * ceil_div(g.N, BN);
    GemmArgs ga = g;
    ga.xcd_remap = tiles % 8 == 0 && tiles >= 64;
    { static char nm[96]; g_assl_last_kernel = note_name(nm, "gemm_bk32_kernel<%s, %s>", TF(TA), TF(TB)); }
    hipLaunchKernelGGL((gemm_bk32_kernel<TA, TB>), dim3(tiles, 1, g.ksplit), dim3(256), lds, s, ga);
    ASSL_LAUNCH_CHECK();
}
int dispatch_bk32(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch_bk32<false, false>(g, s);
    if (!ta && tb) return launch_bk32<false, true>(g, s);
    if (ta && tb) return launch_bk32<true, true>(g, s);
    return launch_bk32<true, false>(g, s);
}

template <typename T, int BK, int MI, int NW = 4>
int dispatch(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch<T, false, false, BK, MI, NW>(g, s);
    if (!ta && tb) return launch<T, false, true, BK, MI, NW>(g, s);
    if (ta && tb) return launch<T, true, true, BK, MI, NW>(g, s);
    return launch<T, true, false, BK, MI, NW>(g, s);
}

// the 8-columns-per-lane epilogue needs whole, aligned vectors of every operand it touches
int epi_vectorisable(const GemmArgs& g, int dtype) {
    static const bool on = getenv("AUDIOSSL_GEMM_VEC_EPI") ? atoi(getenv("AUDIOSSL_GEMM_VEC_EPI")) != 0 : true;
    if (!on || dtype != 1 || g.atomic == 1 || g.N % 8) return 0;
    const size_t csz = g.out_f32 ? 4 : 2;
    if (((size_t)g.C * 1) % 16 || (g.ldc * csz) % 16) return 0;
    if (g.bias && (size_t)g.bias % 16) return 0;
    if (g.keep && ((size_t)g.keep % 8 || g.ldk % 8)) return 0;
    if (g.gate && ((size_t)g.gate % 16 || g.ldg % 8)) return 0;
    if (g.resid && ((size_t)g.resid % 16 || g.ldr % 4)) return 0;
    return 1;
}

}  // namespace

static int dispatch_multi(const GemmMulti& gm, int count, int M, int N, int kmin, int ksplit, int trans_a, int trans_b, hipStream_t s,
                          const int* K, const int* Nv) {
    const long blocks = (long)ceil_div(M, BM) * ceil_div(N, BN) * ksplit * count;
    const bool small = blocks <= 128 && M > 64;
#define MULTI(BK_, MI_, NW_)                                                                              \
    do {                                                                                                  \
        if (!trans_a && !trans_b) return launch_multi<bf16, false, false, BK_, MI_, NW_>(gm, count, ksplit, s); \
        if (!trans_a && trans_b) return launch_multi<bf16, false, true, BK_, MI_, NW_>(gm, count, ksplit, s);   \
        if (trans_a && trans_b) return launch_multi<bf16, true, true, BK_, MI_, NW_>(gm, count, ksplit, s);     \
        return launch_multi<bf16, true, false, BK_, MI_, NW_>(gm, count, ksplit, s);                            \
    } while (0)
    static const int p8 = getenv("AUDIOSSL_GEMM_P8") ? atoi(getenv("AUDIOSSL_GEMM_P8")) : -1;
    static const int p6 = getenv("AUDIOSSL_GEMM_P6") ? atoi(getenv("AUDIOSSL_GEMM_P6")) : -1;
    bool hs_ok = M >= 128 && N >= 128 && (!trans_a || M % 8 == 0);           // the hand-scheduled kernels: K % 64, whole 16-byte rows
    long t5 = 0, t6 = 0, t8 = 0;
    for (int i = 0; i < count; ++i) {
        hs_ok = hs_ok && K[i] % GBK == 0 && (!trans_b || Nv[i] % 8 == 0);
        t5 += (long)ceil_div(M, 128) * ceil_div(Nv[i], 128);
        t6 += (long)ceil_div(M, 256) * ceil_div(Nv[i], 128);
        t8 += (long)ceil_div(M, 256) * ceil_div(Nv[i], 256);
    }
    t5 *= ksplit; t6 *= ksplit; t8 *= ksplit;
    // the 256 x 128 kernel: problems whose 256 x 256 tiles would leave more than half of the CUs idle while the 256 x 128 tiles
    // fit the chip in one round (the M = 1,024 layers and data gradients of the three heads: 192 workgroups); the 128 x 128 kernel
    // where even those are too few (M = 512: 96 -> 192 workgroups)
    if (p6 != 0 && hs_ok && M >= 256 && (p6 == 1 || (t6 >= 128 && t6 <= 288 && t8 < 128 && kmin >= 512)))
    {
        return dispatch_sp_multi<6>(gm, count, ksplit, trans_a, trans_b, s);
    }
    if (p6 != 0 && hs_ok && (p6 == 5 || (p6 != 1 && t5 >= 96 && t5 <= 288 && t6 < 128 && kmin >= 512)))
    {
        if (sp_mask() & 2) return dispatch_sp_multi<5>(gm, count, ksplit, trans_a, trans_b, s);
        return dispatch_pk_multi<5>(gm, count, ksplit, trans_a, trans_b, s);
    }
    // multi-problem launches: measured wins for the transposed-A weight gradients of the three heads (2048 x 2048 x 1024:
    // 50.9 -> 43.9 us, x 512: 31.3 -> 29.0 us)
    if (p8 != 0 && p8 != 2 && hs_ok && M >= 256 && N >= 256 && (p8 == 1 || (trans_a && t8 >= 128 && kmin >= 512)))
    {
        if (sp_mask() & 8) return dispatch_sp_multi<8>(gm, count, ksplit, trans_a, trans_b, s);
        return dispatch_p8_multi(gm, count, ksplit, trans_a, trans_b, s);
    }
    static const int w8 = getenv("AUDIOSSL_GEMM_W8") ? atoi(getenv("AUDIOSSL_GEMM_W8")) : 0;
    if (w8 && M >= 256) MULTI(64, 2, 8);
    if (small) MULTI(64, 1, 4);
    if (blocks <= 256 && kmin >= 512) MULTI(128, 2, 4);
    MULTI(64, 2, 4);
#undef MULTI
}


extern "C" int audiossl_gemm_multi(int count, int trans_a, int trans_b, int M, const int* Nv, const int* K, float alpha,
                                   const void* const* A, const long* lda, const void* const* B, const long* ldb,
                                   void* const* C, const long* ldcv, int out_f32, int atomic, int ksplit, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAX_MULTI && Nv && K && A && B && C && lda && ldb && ldcv && M > 0 && ksplit >= 1);
    int N = 0;                                               // widest problem: sizes the grid, narrower ones leave blocks idle
    for (int i = 0; i < count; ++i) { ASSL_REQUIRE(Nv[i] > 0); N = max(N, Nv[i]); }
    ASSL_REQUIRE(atomic >= 0 && atomic <= 2);
    ASSL_REQUIRE(!atomic || out_f32);
    ASSL_REQUIRE(ksplit == 1 || atomic == 1);
    GemmMulti gm;
    int kmin = 1 << 30;
    for (int i = 0; i < count; ++i) {
        ASSL_REQUIRE(A[i] && B[i] && C[i] && K[i] > 0);
        ASSL_REQUIRE((trans_a ? M : K[i]) % 8 == 0 && (trans_b ? Nv[i] : K[i]) % 8 == 0);
        if (!ASSL_ALIGNED16(A[i]) || !ASSL_ALIGNED16(B[i]) || lda[i] % 8 || ldb[i] % 8) return ASSL_EALIGN;
        const long a_ext = (trans_a ? ((long)(K[i] - 1) * lda[i] + M) : ((long)(M - 1) * lda[i] + K[i])) * 2;
        const long b_ext = (trans_b ? ((long)(K[i] - 1) * ldb[i] + Nv[i]) : ((long)(Nv[i] - 1) * ldb[i] + K[i])) * 2;
        ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);
        gm.p[i] = GemmArgs{A[i], B[i], C[i], M, Nv[i], K[i], lda[i], ldb[i], ldcv[i], alpha, nullptr, 0, nullptr, 0, 1.f, nullptr, 0,
                           out_f32, atomic, ksplit, nullptr, 0, (unsigned)a_ext, (unsigned)b_ext, 0};
        gm.p[i].vec_epi = epi_vectorisable(gm.p[i], 1);
        kmin = min(kmin, K[i]);
    }
    return dispatch_multi(gm, count, M, N, kmin, ksplit, trans_a, trans_b, static_cast<hipStream_t>(stream), K, Nv);
}

// Weight gradients that are applied where they are produced: G_i = alpha * op(A_i) op(B_i) is the gradient of parameter P_i ([M][N_i]
// fp32, leading dimension ldp[i]); the epilogue runs torch.optim.SGD's update on P_i and its momentum buffer Mom_i in place (sgd_step,
// common.h - the same arithmetic as audiossl_sgd_momentum, bit for bit) and writes the bf16 copy of the new parameter to Shadow_i
// (array or entries nullable).  G is never stored: per step that saves the 4-byte write and the optimiser's 4-byte read of every element
// of the projector weights (264 MB at DeLoRes-M's three heads).  One K split only; the momentum buffers must exist (not the first step).
extern "C" int audiossl_gemm_multi_sgd(int count, int trans_a, int trans_b, int M, const int* Nv, const int* K, float alpha,
                                       const void* const* A, const long* lda, const void* const* B, const long* ldb,
                                       float* const* P, float* const* Mom, void* const* Shadow, const long* ldp, float lr, float momentum,
                                       float weight_decay, float grad_scale, const float* grad_scale_dev, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAX_MULTI && Nv && K && A && B && P && Mom && lda && ldb && ldp && M > 0);
    int N = 0;
    for (int i = 0; i < count; ++i) { ASSL_REQUIRE(Nv[i] > 0); N = max(N, Nv[i]); }
    GemmMulti gm;
    int kmin = 1 << 30;
    for (int i = 0; i < count; ++i) {
        ASSL_REQUIRE(A[i] && B[i] && P[i] && Mom[i] && K[i] > 0);
        ASSL_REQUIRE((trans_a ? M : K[i]) % 8 == 0 && (trans_b ? Nv[i] : K[i]) % 8 == 0);
        if (!ASSL_ALIGNED16(A[i]) || !ASSL_ALIGNED16(B[i]) || lda[i] % 8 || ldb[i] % 8) return ASSL_EALIGN;
        if (!ASSL_ALIGNED16(P[i]) || !ASSL_ALIGNED16(Mom[i]) || ldp[i] % 8 || Nv[i] % 8) return ASSL_EALIGN;
        void* sh = Shadow ? Shadow[i] : nullptr;
        if (sh && !ASSL_ALIGNED16(sh)) return ASSL_EALIGN;
        const long a_ext = (trans_a ? ((long)(K[i] - 1) * lda[i] + M) : ((long)(M - 1) * lda[i] + K[i])) * 2;
        const long b_ext = (trans_b ? ((long)(K[i] - 1) * ldb[i] + Nv[i]) : ((long)(Nv[i] - 1) * ldb[i] + K[i])) * 2;
        ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);
        gm.p[i] = GemmArgs{A[i], B[i], P[i], M, Nv[i], K[i], lda[i], ldb[i], ldp[i], alpha, nullptr, 0, nullptr, 0, 1.f, nullptr, 0,
                           1, 0, 1, nullptr, 0, (unsigned)a_ext, (unsigned)b_ext, 0};
        gm.p[i].vec_epi = epi_vectorisable(gm.p[i], 1);
        gm.p[i].sgd_p = P[i]; gm.p[i].sgd_m = Mom[i]; gm.p[i].sgd_s = static_cast<bf16*>(sh); gm.p[i].sgd_gs_dev = grad_scale_dev;
        gm.p[i].sgd_lr = lr; gm.p[i].sgd_mu = momentum; gm.p[i].sgd_wd = weight_decay; gm.p[i].sgd_gs = grad_scale;
        kmin = min(kmin, K[i]);
    }
    return dispatch_multi(gm, count, M, N, kmin, 1, trans_a, trans_b, static_cast<hipStream_t>(stream), K, Nv);
}

// The Barlow-twins cross-correlation of several heads, c_h = alpha * A_h^T B_h (A_h, B_h: [K_h][M] / [K_h][N] row-major, i.e. the
// two normalised views), with the loss and its gradient folded into the epilogue (`delores_s/upstream_expert.py:118-131`):
// dc_h = dscale_h * (c_h - I) in bf16, loss_rep_h[0..31] += coef_h * sum (c_h - I)^2 (32 replicas, the caller sums them; zeroed by
// the caller).  c_h is never stored (50 MB written + read per launch at D = 2048, three heads).
extern "C" int audiossl_gemm_multi_barlow(int count, int D, const int* K, float alpha, const void* const* A, const long* lda,
                                          const void* const* B, const long* ldb, void* const* dc, const float* coef, const float* dscale,
                                          float* const* loss_rep, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAX_MULTI && K && A && B && dc && lda && ldb && coef && dscale && loss_rep && D > 0 && D % 64 == 0);
    GemmMulti gm;
    int kmin = 1 << 30;
    int Nv[MAX_MULTI];
    for (int i = 0; i < count; ++i) {
        ASSL_REQUIRE(A[i] && B[i] && dc[i] && loss_rep[i] && K[i] > 0 && K[i] % 8 == 0);
        if (!ASSL_ALIGNED16(A[i]) || !ASSL_ALIGNED16(B[i]) || !ASSL_ALIGNED16(dc[i]) || lda[i] % 8 || ldb[i] % 8) return ASSL_EALIGN;
        const long a_ext = ((long)(K[i] - 1) * lda[i] + D) * 2, b_ext = ((long)(K[i] - 1) * ldb[i] + D) * 2;
        ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);
        gm.p[i] = GemmArgs{A[i], B[i], dc[i], D, D, K[i], lda[i], ldb[i], D, alpha, nullptr, 0, nullptr, 0, 1.f, nullptr, 0,
                           0, 0, 1, nullptr, 0, (unsigned)a_ext, (unsigned)b_ext, 0};
        gm.p[i].vec_epi = epi_vectorisable(gm.p[i], 1);
        ASSL_REQUIRE(gm.p[i].vec_epi);
        gm.p[i].bl_mode = 1; gm.p[i].bl_loss = loss_rep[i]; gm.p[i].bl_coef = coef[i]; gm.p[i].bl_dscale = dscale[i];
        kmin = min(kmin, K[i]);
        Nv[i] = D;
    }
    return dispatch_multi(gm, count, D, D, kmin, 1, 1, 1, static_cast<hipStream_t>(stream), K, Nv);
}

// kernel choice for bf16 operands (measured rules, see the comments inside)
static int run_bf16(const GemmArgs& g, int trans_a, int trans_b, hipStream_t s) {
    const int M = g.M, N = g.N, K = g.K, ksplit = g.ksplit;
    const long blocks = (long)ceil_div(M, BM) * ceil_div(N, BN) * ksplit;
    // AUDIOSSL_GEMM_RING = 10 * MI + NST forces one ring variant (tools/gemm_shapes.py sweeps them); 0 disables the ring
    static const int ring = getenv("AUDIOSSL_GEMM_RING") ? atoi(getenv("AUDIOSSL_GEMM_RING")) : -1;
    const bool ring_ok = K % GBK == 0 && M >= 8 && N >= 8;
    static const int w8 = getenv("AUDIOSSL_GEMM_W8") ? atoi(getenv("AUDIOSSL_GEMM_W8")) : 0;
    if (w8 && M >= 256) return dispatch<bf16, 64, 2, 8>(g, trans_a, trans_b, s);
    // the hand-scheduled 256 x 256 kernel: AUDIOSSL_GEMM_P8 = 1 forces it wherever it is legal, 0 disables it
    static const int p8 = getenv("AUDIOSSL_GEMM_P8") ? atoi(getenv("AUDIOSSL_GEMM_P8")) : -1;
    const bool p8_ok = K % GBK == 0 && M >= 8 && N >= 8 && (!trans_a || M % 8 == 0) && (!trans_b || N % 8 == 0);
    // transposed-A problems whose split-K launch fills the chip with 256 x 128 tiles: the software-pipelined kernel with the caller's
    // split beats the 256 x 256 kernel with the split raised to fill the chip (the encoder's fc.3 weight gradient, 2048 x 2048 x 6144
    // split in two: 97.4 -> 77.6 us, 664 TFLOP/s; AUDIOSSL_GEMM_P6=0 disables)
    static const int p6_first = getenv("AUDIOSSL_GEMM_P6") ? atoi(getenv("AUDIOSSL_GEMM_P6")) : -1;
    if (p6_first != 0 && p8 != 1 && p8_ok && trans_a && g.atomic == 1 && !g.lse_mode && M >= 256 && N >= 128) {
        const long t6s = (long)ceil_div(M, 256) * ceil_div(N, 128) * ksplit;
        if (t6s >= 192 && t6s <= 288 && K / ksplit >= 2048) return dispatch_sp<6>(g, trans_a, trans_b, s);
    }
    if (p8 != 0 && p8_ok && M >= 256 && N >= 256) {
        // measured (tools/gemm_shapes.py, gemm_one.py): it wins once the 256 x 256 tiles occupy at least half of the CUs and every
        // workgroup walks >= 16 K-tiles (6144 x 2048 x 2048 NT 83.9 -> 63.0 us with the bias / ReLU / dropout epilogue, NN 71.0 ->
        // 60.4; 4096^3 214 -> 127 us, 8192^3 1.30 PFLOP/s); with few K-tiles its one-workgroup-per-CU epilogue is exposed
        // (6144 x 2048 x 512: 30.8 vs 27.2 us) and on small grids the 128-row kernels spread over more CUs.
        const long t256 = (long)ceil_div(M, 256) * ceil_div(N, 256);
        GemmArgs ga = g;
        if (g.atomic == 1 && t256 * ksplit < 192) {                  // accumulating outputs: split K until the chip is full
            int ks = ksplit;
            while (t256 * ks * 2 <= 256 && K / (ks * 2) >= 1024) ks *= 2;
            ga.ksplit = ks;
        }
        const bool layout_ok = p8 != 2 || (!trans_a && !trans_b);            // AUDIOSSL_GEMM_P8=2: K-contiguous operands only
        // ... and with 12 K-tiles once the grid is several rounds deep (the transformer blocks' K = 768 layers at 27,648 rows:
        // 27648 x 2304 x 768 NT 188.6 -> 140.5 us, x 3072 235.3 -> 181.2, x 768 63.9 -> 53.2 against the K-step-32 kernel)
        const bool deep = t256 * ga.ksplit >= 512 && K / ga.ksplit >= 768;
        if (p8 == 1 || (layout_ok && t256 * ga.ksplit >= 128 && (K / ga.ksplit >= 1024 || deep)))
            return sp_mask() & 8 ? dispatch_sp<8>(ga, trans_a, trans_b, s) : dispatch_p8(ga, trans_a, trans_b, s);
    }
    // the hand-scheduled 256 x 128 kernel (written for the multi-problem launches of the projector heads, see dispatch_multi):
    // single problems whose 256 x 128 tiles fill the chip once while the 256 x 256 tiles would leave half of it idle
    static const int p6 = getenv("AUDIOSSL_GEMM_P6") ? atoi(getenv("AUDIOSSL_GEMM_P6")) : -1;
    if (p6 != 0 && p8_ok && !g.lse_mode && M >= 128 && N >= 128) {
        const long t5 = (long)ceil_div(M, 128) * ceil_div(N, 128) * ksplit, t6 = (long)ceil_div(M, 256) * ceil_div(N, 128) * ksplit,
                   t8 = (long)ceil_div(M, 256) * ceil_div(N, 256) * ksplit;
        if (M >= 256 && (p6 == 1 || (t6 >= 128 && t6 <= 288 && t8 < 128 && K / ksplit >= 512)))
            return dispatch_sp<6>(g, trans_a, trans_b, s);
        // measured (tools/gemm_shapes.py): NN 6144 x 512 x 2048 34.1 -> 25.8 us, TN 2048 x 512 x 6144 / 3 41.9 -> 32.2 us; with short K
        // loops (2048 x 2048 x 512 TN: 19.3 vs 17.9 us) and for K-contiguous operands on <= 128 tiles (the ring kernel: 20.2 vs 21.7 us)
        // the older kernels stay
        if (p6 == 5 || (p6 != 1 && t5 >= 96 && t5 <= 288 && t6 < 128 && K / ksplit >= 1024 && (trans_a || trans_b || t5 > 128)))
            return sp_mask() & 2 ? dispatch_sp<5>(g, trans_a, trans_b, s) : dispatch_pk<5>(g, trans_a, trans_b, s);
    }
    // grids of >= 2 workgroups per CU with a K-contiguous A operand: K-step 32 and THREE co-resident workgroups per CU (41 KB
    // of LDS each, 126 VGPRs) - 12 waves per CU hide the staged-load and barrier waits better than two workgroups at K-step
    // 64 (6144x2048x2048 NT 80.9 -> 71.5 us, NN 90.2 -> 76.8 us; with a transposed A operand it measured slower)
    static const int bk32 = getenv("AUDIOSSL_GEMM_BK32") ? atoi(getenv("AUDIOSSL_GEMM_BK32")) : 1;
    if (bk32 && (!trans_a || bk32 == 2) && (long)ceil_div(M, BM) * ceil_div(N, BN) * ksplit >= 512)
        return dispatch_bk32(g, trans_a, trans_b, s);
    static const int t256 = getenv("AUDIOSSL_GEMM_T256") ? atoi(getenv("AUDIOSSL_GEMM_T256")) : 0;
    if (t256 && M >= 1024 && N >= 512) return dispatch256(g, trans_a, trans_b, s);
    if (ring_ok && ring > 0) {
        switch (ring) {
            case 12: return dispatch_ring<1, 2>(g, trans_a, trans_b, s);
            case 13: return dispatch_ring<1, 3>(g, trans_a, trans_b, s);
            case 14: return dispatch_ring<1, 4>(g, trans_a, trans_b, s);
            case 16: return dispatch_ring<1, 6>(g, trans_a, trans_b, s);
            case 22: return dispatch_ring<2, 2>(g, trans_a, trans_b, s);
            case 23: return dispatch_ring<2, 3>(g, trans_a, trans_b, s);
            case 24: return dispatch_ring<2, 4>(g, trans_a, trans_b, s);
            default: break;
        }
    }
    // measured (tools/gemm_shapes.py): the ring wins for K-contiguous operands (NT) on grids of at most two waves of
    // workgroups - 64-row tiles up to 128 full tiles, 128-row tiles up to 512; with a transposed operand or on larger grids
    // two co-resident workgroups of the register-staged kernel overlap each other better
    if (ring_ok && ring < 0 && !trans_a && !trans_b) {
        if (blocks <= 128 && M > 64) return dispatch_ring<1, 4>(g, 0, 0, s);
        if (blocks > 128 && blocks <= 512) return dispatch_ring<2, 4>(g, 0, 0, s);
    }
    static const bool small_tiles = getenv("AUDIOSSL_GEMM_SMALL") ? atoi(getenv("AUDIOSSL_GEMM_SMALL")) != 0 : true;
    // these shapes are latency-bound per workgroup (~0.5 us per 64-deep k-step whatever the tile): when 128 x 128 tiles
    // would occupy at most half of the 256 CUs, halve the tile in M and run twice as many workgroups
    if (small_tiles && blocks <= 128 && M > 64) return dispatch<bf16, 64, 1>(g, trans_a, trans_b, s);
    return blocks <= 256 && K >= 512 ? dispatch<bf16, 128, 2>(g, trans_a, trans_b, s) : dispatch<bf16, 64, 2>(g, trans_a, trans_b, s);
}

// dtype: 0 = fp32 operands (exact f32 MFMA), 1 = bf16 operands.  See include/audiossl_hip.h.
extern "C" int audiossl_gemm(int dtype, int trans_a, int trans_b, int M, int N, int K, float alpha,
                             const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                             const float* bias, int relu, const uint8_t* keep, long ldk, float keep_scale,
                             const void* gate, long ldg, int out_f32, int atomic, int ksplit, const float* resid, long ldr,
                             void* stream) {
    ASSL_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ksplit >= 1);
    ASSL_REQUIRE(dtype == 0 || dtype == 1);
    ASSL_REQUIRE(atomic >= 0 && atomic <= 2);
    ASSL_REQUIRE(!atomic || out_f32 || dtype == 0);
    ASSL_REQUIRE(ksplit == 1 || atomic == 1);
    ASSL_REQUIRE(ksplit == 1 || (!relu && !keep && !gate));             // non-linear epilogue terms need the whole sum (the bias is added by split 0)
    ASSL_REQUIRE(!resid || (ksplit == 1 && !atomic));
    // vector (8-element) dimension of each operand must be a multiple of 8 and its rows 16-byte aligned
    ASSL_REQUIRE((trans_a ? M : K) % 8 == 0 && (trans_b ? N : K) % 8 == 0);
    if (!ASSL_ALIGNED16(A) || !ASSL_ALIGNED16(B) || lda % 8 || ldb % 8) return ASSL_EALIGN;
    const long esz = dtype == 0 ? 4 : 2;
    const long a_ext = (trans_a ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * esz;
    const long b_ext = (trans_b ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K)) * esz;
    ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);            // 32-bit buffer offsets
    GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc, alpha, bias, relu, keep, ldk, keep_scale, gate, ldg,
               (dtype == 0) ? 1 : out_f32, atomic, ksplit, resid, ldr, (unsigned)a_ext, (unsigned)b_ext, 0};
    g.vec_epi = epi_vectorisable(g, dtype);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == 0) return dispatch<float, 32, 2>(g, trans_a, trans_b, s);
    return run_bf16(g, trans_a, trans_b, s);
}


// Linear -> ReLU -> Dropout of the encoder's first fully connected layer (`src/encoder/audiontt.py:62-66`) with the keep mask drawn in
// the epilogue: C = dropout(relu(alpha * op(A) op(B) + bias)), element (row, col) kept iff dropout_keep(seed', row * N + col) with
// seed' = (seed + *counter) mod 2^48 - bit for bit the mask audiossl_dropout_mask(keep [M][N], seed, p, counter) writes, which this
// launch neither needs written (12.6 MB at B = 512) nor read back.  The backward needs no mask: it gates on the stored output.
extern "C" int audiossl_gemm_dropout(int trans_a, int trans_b, int M, int N, int K, float alpha, const void* A, long lda,
                                     const void* B, long ldb, void* C, long ldc, const float* bias, int relu,
                                     unsigned long long seed, float p, const long long* counter, float keep_scale, void* stream) {
    ASSL_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && p >= 0.f && p < 1.f);
    ASSL_REQUIRE((trans_a ? M : K) % 8 == 0 && (trans_b ? N : K) % 8 == 0);
    if (!ASSL_ALIGNED16(A) || !ASSL_ALIGNED16(B) || lda % 8 || ldb % 8) return ASSL_EALIGN;
    const long a_ext = (trans_a ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 2;
    const long b_ext = (trans_b ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K)) * 2;
    ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);
    GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc, alpha, bias, relu, nullptr, 0, keep_scale, nullptr, 0,
               0, 0, 1, nullptr, 0, (unsigned)a_ext, (unsigned)b_ext, 0};
    g.vec_epi = epi_vectorisable(g, 1);
    g.drop_on = 1;
    g.drop_thr = (unsigned int)fminf(p * 4294967296.f, 4294967295.f);
    g.drop_seed = seed;
    g.drop_counter = counter;
    return run_bf16(g, trans_a, trans_b, static_cast<hipStream_t>(stream));
}

// MoCo InfoNCE logits  qn [B][dim] * queue [dim][K] / T  with the row soft-max folded into the GEMM epilogue
// (`src/upstream/delores_m/upstream_expert.py:250-264`): mode 1 writes per-row (max, sum exp) partials of every 64-column
// slab (merged by audiossl_moco_lse_merge), mode 2 recomputes the logits and stores P = exp(l - lse) * gscale in bf16, the
// operand of the dq GEMM.  The [B][K] fp32 logits (134 MB at B = 512, K = 65,536) are never written.
extern "C" int audiossl_moco_logits(int mode, const void* qn, const void* queue, int B, int K, int dim, float inv_t, float* part,
                                    const float* lse, float gscale, void* P, void* stream) {
    ASSL_REQUIRE(qn && queue && B > 0 && K > 0 && dim > 0 && (mode == 1 || mode == 2));
    ASSL_REQUIRE(mode == 1 ? part != nullptr : (lse != nullptr && P != nullptr));
    ASSL_REQUIRE(dim % 8 == 0 && K % 8 == 0);
    if (!ASSL_ALIGNED16(qn) || !ASSL_ALIGNED16(queue) || (P && !ASSL_ALIGNED16(P))) return ASSL_EALIGN;
    const long a_ext = ((long)(B - 1) * dim + dim) * 2, b_ext = ((long)(dim - 1) * K + K) * 2;
    ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);
    GemmArgs g{qn, queue, P, B, K, dim, dim, K, K, inv_t, nullptr, 0, nullptr, 0, 1.f, nullptr, 0,
               0, 0, 1, nullptr, 0, (unsigned)a_ext, (unsigned)b_ext, 0, 0, mode, part, lse, gscale, ceil_div(K, 64)};
    return run_bf16(g, 0, 1, static_cast<hipStream_t>(stream));
}
