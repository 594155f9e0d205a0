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
    int atomic;               // C (float) += via atomicAdd (split-K / multi-source accumulation)
    int ksplit;               // gridDim.z
    const float* resid;       // fp32 [M][ldr] added to the result (out-of-place residual connection) or null
    long ldr;
    unsigned a_bytes, b_bytes; // extents of A and B for the buffer descriptors (hardware bounds check)
};

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
constexpr size_t EPI_LDS = sizeof(float) * 4 * 64 * EPI_PITCH;   // 66,560 B, fits inside every variant's staging LDS

// MI = 32-row MFMA blocks per wave in M (wave tile = 32*MI x 64)
template <typename T, int MI>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& g, f32x16 (&acc)[MI][2], char* smem, int row0, int col0, int lane,
                                             int wave) {
    __syncthreads();                                             // every wave is done with the staging buffers
    float* ct = reinterpret_cast<float*>(smem) + wave * 64 * EPI_PITCH;
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ct[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * EPI_PITCH + j * 32 + l31] = acc[i][j][r];
    const int col = col0 + lane;
    if (col >= g.N) return;
    const float bias = g.bias ? g.bias[col] : 0.f;
    const float floor_ = g.relu ? 0.f : -3.4e38f;
    const T* gate = static_cast<const T*>(g.gate);
    const int rows = min(32 * MI, g.M - row0);
    // 8 rows per trip: the LDS reads and the optional global loads of a trip are independent, so their latencies overlap
    // (one row per trip serialised ~150 cycles of LDS + store issue per row: 4 us of a 14 us single-tile launch)
    for (int r0 = 0; r0 < rows; r0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ct[min(r0 + u, 63) * EPI_PITCH + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long row = row0 + min(r0 + u, rows - 1);
            float x = fmaxf(g.alpha * v[u] + bias, floor_);
            if (g.keep) x = g.keep[row * g.ldk + col] ? x * g.keep_scale : 0.f;
            if (gate) x = to_f32(gate[row * g.ldg + col]) > 0.f ? x : 0.f;
            if (g.resid) x += g.resid[row * g.ldr + col];
            v[u] = x;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (r0 + u >= rows) break;
            const long o = (long)(row0 + r0 + u) * g.ldc + col;
            if (g.atomic)       atomicAdd(static_cast<float*>(g.C) + o, v[u]);
            else if (g.out_f32) static_cast<float*>(g.C)[o] = v[u];
            else                static_cast<T*>(g.C)[o] = from_f32<T>(v[u]);
        }
    }
}

// One operand's staging: 128 rows x 32 k per step, two Vec8 per thread.
template <typename T, bool TRANS, int BK_, int ROWS = 128>
struct Stage {
    static constexpr int BK = BK_;
    static constexpr int NT_PITCH = BK + 8;          // elements; [row][k] image, conflict-free 16-byte row reads
    static constexpr int TR_PITCH = ROWS + 32;       // elements; [k][row] image: 4 consecutive k-rows fall on distinct 32-byte bank groups
    static constexpr int VR = ROWS / 8;              // vectors per k-row of the [k][row] image
    static constexpr int NV = ROWS * BK / 8 / 256;   // 8-element vectors per thread and step
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
            const int v = t + i * 256;
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
            const int v = t + i * 256;
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

template <typename T, bool TA, bool TB, int BK, int MI>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, char* smem) {
    constexpr int BMv = 64 * MI;                                 // 128 x 128 tile (MI = 2) or 64 x 128 (MI = 1: twice the
    using SA = Stage<T, TA, BK, BMv>;                            // workgroups for grids that would leave CUs idle)
    using SB = Stage<T, TB, BK, BN>;
    T* const ldsA0 = reinterpret_cast<T*>(smem);                 // two A buffers, then two B buffers
    T* const ldsB0 = ldsA0 + 2 * SA::LDS_ELEMS;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int bm = (blockIdx.x / tiles_n) * BMv, bn = (blockIdx.x % tiles_n) * BN;
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

    epilogue_lds<T, MI>(g, acc, smem, bm + wm, bn + wn, lane, wave);
}

template <typename T, bool TA, bool TB, int BK, int MI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<T, TA, TB, BK, MI>(g, smem);
}

// Several independent problems of one kind in ONE launch (blockIdx.y = problem): the three Barlow heads of delores_m run
// the same chain of GEMMs on different operands; issued as separate launches on separate streams they were serialised by
// the hardware-queue mapping of the graph executor (1.6 ms of a 3.3 ms step), and each launch filled half the chip at best.
constexpr int MAX_MULTI = 4;
struct GemmMulti { GemmArgs p[MAX_MULTI]; };

template <typename T, bool TA, bool TB, int BK, int MI>
__global__ __launch_bounds__(256) void gemm_multi_kernel(GemmMulti gm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmArgs& g = gm.p[blockIdx.y];
    const int tiles = ((g.M + 64 * MI - 1) / (64 * MI)) * ((g.N + BN - 1) / BN);
    if ((int)blockIdx.x >= tiles) return;
    gemm_body<T, TA, TB, BK, MI>(g, smem);
}

template <typename T, bool TA, bool TB, int BK, int MI>
int launch_multi(const GemmMulti& gm, int count, int max_tiles, int ksplit, hipStream_t s) {
    const size_t lds = max(sizeof(T) * 2 * (Stage<T, TA, BK, 64 * MI>::LDS_ELEMS + Stage<T, TB, BK, BN>::LDS_ELEMS), EPI_LDS);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_multi_kernel<T, TA, TB, BK, MI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_multi_kernel<T, TA, TB, BK, MI>), dim3(max_tiles, count, ksplit), dim3(256), lds, s, gm);
    ASSL_LAUNCH_CHECK();
}

template <typename T, bool TA, bool TB, int BK, int MI>
int launch(const GemmArgs& g, hipStream_t s) {
    const size_t lds = max(sizeof(T) * 2 * (Stage<T, TA, BK, 64 * MI>::LDS_ELEMS + Stage<T, TB, BK, BN>::LDS_ELEMS), EPI_LDS);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, TA, TB, BK, MI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, 64 * MI) * ceil_div(g.N, BN);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TB, BK, MI>), dim3(tiles, 1, g.ksplit), dim3(256), lds, s, g);
    ASSL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------------
// NT bf16 with direct-to-LDS staging (`global_load_lds_dwordx4`) and a 4-deep ring: the pipelined variant for the shapes
// where one register stage cannot hide the load latency (M = 512..6144 Linear layers of the step: ~1.4 us per 64-deep
// k-step against 0.2 us of MFMA work) and for the large transformer GEMMs.
//   * an LDS-DMA instruction writes wave-uniform base + lane * 16 B, so the stage image is the plain [128 rows][64 k]
//     tile (128-byte rows, no padding); bank conflicts of the fragment reads are removed by an XOR swizzle of the 16-byte
//     chunk index, chunk ^= (row >> 1) & 7, applied to the SOURCE address of the DMA and to the ds_read address;
//   * up to three stages stay in flight across the (raw) barrier: counted s_waitcnt vmcnt(N), never 0 in steady state;
//   * rows past M / N are clamped to the last row (their results are never stored); K must be a multiple of 64.
constexpr int GBK = 64;
constexpr int GSTAGES = 4;
constexpr int GOP_BYTES = 128 * GBK * 2;                 // one operand of one stage: 16 KB
constexpr int GSTAGE_BYTES = 2 * GOP_BYTES;

__device__ __forceinline__ void glds16(const bf16* src, char* dst) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

__global__ __launch_bounds__(256) void gemm_nt_glds_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tiles_n = (g.N + BN - 1) / BN;
    // blocks that share an XCD (ids congruent mod 8) get consecutive tiles: they share A row panels / B column panels in L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int bm = (wg / tiles_n) * BM, bn = (wg % tiles_n) * BN;
    const int ksteps = g.K / GBK;
    const int per = (ksteps + g.ksplit - 1) / g.ksplit;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);
    if (ks0 >= ks1) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    const bf16* Ap[4];
    const bf16* Bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = wave * 32 + i * 8 + (lane >> 3);
        const int sc = (lane & 7) ^ ((rl >> 1) & 7);
        Ap[i] = static_cast<const bf16*>(g.A) + (long)min(bm + rl, g.M - 1) * g.lda + sc * 8 + (long)ks0 * GBK;
        Bp[i] = static_cast<const bf16*>(g.B) + (long)min(bn + rl, g.N - 1) * g.ldb + sc * 8 + (long)ks0 * GBK;
    }
    char* const wbase = smem + (wave * 32) * 128;        // this wave's first row inside an operand image

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = ks1 - ks0;
    auto issue = [&](int st) {
        char* d = wbase + (st % GSTAGES) * GSTAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(Ap[i] + (long)st * GBK, d + i * 8 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(Bp[i] + (long)st * GBK, d + GOP_BYTES + i * 8 * 128);
    };
    for (int st = 0; st < GSTAGES - 1 && st < nk; ++st) issue(st);

    // fragment addressing: row r, logical 16-byte chunk cc -> physical chunk cc ^ ((r >> 1) & 7)
    const int half = lane >> 5, l31 = lane & 31;
    int offA[2], offB[2], swA[2], swB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm + i * 32 + l31, rb = wn + i * 32 + l31;
        offA[i] = ra * 128; swA[i] = (ra >> 1) & 7;
        offB[i] = GOP_BYTES + rb * 128; swB[i] = (rb >> 1) & 7;
    }
    for (int it = 0; it < nk; ++it) {
        const int ahead = min(nk - 1 - it, GSTAGES - 2);
        if (ahead >= 2)      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // stage `it` visible to every wave; slot of stage it-1 free again
        if (it + GSTAGES - 1 < nk) issue(it + GSTAGES - 1);
        const char* sb = smem + (it % GSTAGES) * GSTAGE_BYTES;
        // fragment reads run one k-step ahead of the MFMAs that consume them (one wave per SIMD: nobody else hides the
        // ~120-cycle LDS latency)
        Vec8<bf16> fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[0][i] = Vec8<bf16>::load(reinterpret_cast<const bf16*>(sb + offA[i] + ((half ^ swA[i]) << 4)));
            fb[0][i] = Vec8<bf16>::load(reinterpret_cast<const bf16*>(sb + offB[i] + ((half ^ swB[i]) << 4)));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk < 3) {
                const int cc = (kk + 1) * 2 + half;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[nxt][i] = Vec8<bf16>::load(reinterpret_cast<const bf16*>(sb + offA[i] + ((cc ^ swA[i]) << 4)));
                    fb[nxt][i] = Vec8<bf16>::load(reinterpret_cast<const bf16*>(sb + offB[i] + ((cc ^ swB[i]) << 4)));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<bf16>::run(acc[i][j], fa[cur][i], fb[cur][j]);
        }
    }

    epilogue_lds<bf16, 2>(g, acc, smem, bm + wm, bn + wn, lane, wave);
}

int launch_nt_glds(const GemmArgs& g, hipStream_t s) {
    constexpr size_t lds = (size_t)GSTAGES * GSTAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_glds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return ASSL_ELAUNCH;
        attr_set = true;
    }
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.N, BN);
    hipLaunchKernelGGL(gemm_nt_glds_kernel, dim3(tiles, 1, g.ksplit), dim3(256), lds, s, g);
    ASSL_LAUNCH_CHECK();
}

template <typename T, int BK, int MI>
int dispatch(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && !tb) return launch<T, false, false, BK, MI>(g, s);
    if (!ta && tb) return launch<T, false, true, BK, MI>(g, s);
    if (ta && tb) return launch<T, true, true, BK, MI>(g, s);
    return launch<T, true, false, BK, MI>(g, s);
}

}  // namespace

extern "C" int audiossl_gemm_multi(int count, int trans_a, int trans_b, int M, int N, const int* K, float alpha,
                                   const void* const* A, const long* lda, const void* const* B, const long* ldb,
                                   void* const* C, long ldc, int out_f32, int atomic, int ksplit, void* stream) {
    ASSL_REQUIRE(count >= 1 && count <= MAX_MULTI && K && A && B && C && lda && ldb && M > 0 && N > 0 && ksplit >= 1);
    ASSL_REQUIRE(!atomic || out_f32);
    ASSL_REQUIRE(ksplit == 1 || atomic);
    GemmMulti gm;
    int kmin = 1 << 30;
    for (int i = 0; i < count; ++i) {
        ASSL_REQUIRE(A[i] && B[i] && C[i] && K[i] > 0);
        ASSL_REQUIRE((trans_a ? M : K[i]) % 8 == 0 && (trans_b ? N : K[i]) % 8 == 0);
        if (!ASSL_ALIGNED16(A[i]) || !ASSL_ALIGNED16(B[i]) || lda[i] % 8 || ldb[i] % 8) return ASSL_EALIGN;
        const long a_ext = (trans_a ? ((long)(K[i] - 1) * lda[i] + M) : ((long)(M - 1) * lda[i] + K[i])) * 2;
        const long b_ext = (trans_b ? ((long)(K[i] - 1) * ldb[i] + N) : ((long)(N - 1) * ldb[i] + K[i])) * 2;
        ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);
        gm.p[i] = GemmArgs{A[i], B[i], C[i], M, N, K[i], lda[i], ldb[i], ldc, alpha, nullptr, 0, nullptr, 0, 1.f, nullptr, 0,
                           out_f32, atomic, ksplit, nullptr, 0, (unsigned)a_ext, (unsigned)b_ext};
        kmin = min(kmin, K[i]);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long blocks = (long)ceil_div(M, BM) * ceil_div(N, BN) * ksplit * count;
    const bool small = blocks <= 128 && M > 64;
    const int max_tiles = ceil_div(M, small ? 64 : BM) * ceil_div(N, BN);
#define MULTI(BK_, MI_)                                                                                   \
    do {                                                                                                  \
        if (!trans_a && !trans_b) return launch_multi<bf16, false, false, BK_, MI_>(gm, count, max_tiles, ksplit, s); \
        if (!trans_a && trans_b) return launch_multi<bf16, false, true, BK_, MI_>(gm, count, max_tiles, ksplit, s);   \
        if (trans_a && trans_b) return launch_multi<bf16, true, true, BK_, MI_>(gm, count, max_tiles, ksplit, s);     \
        return launch_multi<bf16, true, false, BK_, MI_>(gm, count, max_tiles, ksplit, s);                            \
    } while (0)
    if (small) MULTI(64, 1);
    if (blocks <= 256 && kmin >= 512) MULTI(128, 2);
    MULTI(64, 2);
#undef MULTI
}

// dtype: 0 = fp32 operands (exact f32 MFMA), 1 = bf16 operands.  See include/audiossl_hip.h.
extern "C" int audiossl_gemm(int dtype, int trans_a, int trans_b, int M, int N, int K, float alpha,
                             const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                             const float* bias, int relu, const uint8_t* keep, long ldk, float keep_scale,
                             const void* gate, long ldg, int out_f32, int atomic, int ksplit, const float* resid, long ldr,
                             void* stream) {
    ASSL_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ksplit >= 1);
    ASSL_REQUIRE(dtype == 0 || dtype == 1);
    ASSL_REQUIRE(!atomic || out_f32 || dtype == 0);
    ASSL_REQUIRE(ksplit == 1 || atomic);
    ASSL_REQUIRE(!resid || (ksplit == 1 && !atomic));
    // vector (8-element) dimension of each operand must be a multiple of 8 and its rows 16-byte aligned
    ASSL_REQUIRE((trans_a ? M : K) % 8 == 0 && (trans_b ? N : K) % 8 == 0);
    if (!ASSL_ALIGNED16(A) || !ASSL_ALIGNED16(B) || lda % 8 || ldb % 8) return ASSL_EALIGN;
    const long esz = dtype == 0 ? 4 : 2;
    const long a_ext = (trans_a ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * esz;
    const long b_ext = (trans_b ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K)) * esz;
    ASSL_REQUIRE(a_ext < 0xFFFFFF00L && b_ext < 0xFFFFFF00L);            // 32-bit buffer offsets
    GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc, alpha, bias, relu, keep, ldk, keep_scale, gate, ldg,
               (dtype == 0) ? 1 : out_f32, atomic, ksplit, resid, ldr, (unsigned)a_ext, (unsigned)b_ext};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == 0) return dispatch<float, 32, 2>(g, trans_a, trans_b, s);
    static const bool use_glds = getenv("AUDIOSSL_GEMM_GLDS") ? atoi(getenv("AUDIOSSL_GEMM_GLDS")) != 0 : true;
    const long blocks = (long)ceil_div(M, BM) * ceil_div(N, BN) * ksplit;
    // the 4-deep direct-to-LDS ring (one workgroup per CU) wins for one to two waves of workgroups (below that the 64-row
    // tiles further down are faster still); beyond
    // that two co-resident workgroups of the register-staged kernel overlap each other better (tools/gemm_floor.py)
    if (use_glds && !trans_a && !trans_b && K % GBK == 0 && blocks > 128 && blocks <= 512) return launch_nt_glds(g, s);
    static const bool small_tiles = getenv("AUDIOSSL_GEMM_SMALL") ? atoi(getenv("AUDIOSSL_GEMM_SMALL")) != 0 : true;
    // these shapes are latency-bound per workgroup (~0.5 us per 64-deep k-step whatever the tile): when 128 x 128 tiles
    // would occupy at most half of the 256 CUs, halve the tile in M and run twice as many workgroups
    if (small_tiles && blocks <= 128 && M > 64) return dispatch<bf16, 64, 1>(g, trans_a, trans_b, s);
    return blocks <= 256 && K >= 512 ? dispatch<bf16, 128, 2>(g, trans_a, trans_b, s) : dispatch<bf16, 64, 2>(g, trans_a, trans_b, s);
}
