// K7: 3x3 / 64->64 channel convolution of AudioNTT2020Task6's blocks 2 and 3 (`src/encoder/audiontt.py:52-60`) as
// implicit GEMM on the bf16 MFMA pipe - forward, data gradient (same kernel, flipped/transposed weights) and weight
// gradient - with no im2col buffer.
//
// Layout: activations [N][T][F][64] bf16 (pixel = 128-byte channel vector); rows of F pixels are enumerated globally
// (g = n*T + t) so tiles never waste work at image ends; image borders in T are handled by predicates, borders in F by
// zero columns of the LDS halo tile.
//
// conv3x3_kernel (fwd / dgrad):  Y[pix][co] = sum_{tap,ci} X[pix+tap][ci] * W[co][tap*64+ci]
//   one persistent 256-thread workgroup per CU: the whole 64x576 weight matrix (74 KB) stays in LDS, each tile is 256
//   pixels (8 rows x 32 or 16 rows x 16) whose (rows+2) x (F+2) halo is prefetched into registers during the MFMAs of
//   the previous tile; a wave owns 64 pixels x 64 channels = 2x2 accumulators of v_mfma_f32_32x32x16_bf16, 36 k-steps
//   (9 taps x 4 channel chunks), 1 ds_read_b128 per MFMA.  Optional epilogue: + bias, per-channel sum / sum of squares
//   of the fp32 accumulators (BatchNorm batch statistics, fp64 atomics once per workgroup).
// conv3x3_wgrad_kernel:  dW[co][tap*64+ci] += sum_pix dY[pix][co] * X[pix+tap][ci]
//   k = pixels, so both operands are read from their row-major LDS tiles with ds_read_b64_tr_b16 (4 pixels x 16
//   channels transposed per 16-lane group); the 64x576 fp32 result lives in registers (9 accumulators per wave) across
//   all tiles of the workgroup and is added to global memory once (fp32 atomics, 128-byte runs).
// Roofline (B=512, block 2): 60.4 GFLOP per pass; algorithmic HBM traffic 105 MB in + 105 MB out.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int CH = 64, KTOT = 576;
constexpr int WPITCH = KTOT + 8;        // elements: 1168-byte weight rows, conflict-free ds_read_b128 over 16 rows
constexpr int PIXP = 72;                // elements: 144-byte pixel pitch, conflict-free ds_read_b128 over 16 pixels
constexpr int PIXW = 96;                // elements: 192-byte pixel pitch for the tr-read tiles of the wgrad kernel

typedef __attribute__((address_space(3))) bf16x4* lds4_t;

struct ConvArgs {
    const bf16* X; const bf16* W; const float* bias; bf16* Y; double* sum; double* sumsq;
    int Ti, rows_total, tiles, out_f32;
};

template <int FI, int PITCH, int NTH = 256>
struct Halo {
    static constexpr int TT = 256 / FI;                         // t-rows per tile
    static constexpr int COLS = FI + 2;
    static constexpr int ELEMS = (TT + 2) * COLS * PITCH;
    static constexpr int NV = ((TT + 2) * FI * 8 + NTH - 1) / NTH;   // 16-byte vectors per thread
    static constexpr int NVEC = (TT + 2) * FI * 8;
    Vec8<bf16> r[NV];
    __device__ __forceinline__ void load(const bf16* __restrict__ X, int g0, int rows_total) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + i * NTH;
            const int row = v / (FI * 8), rem = v % (FI * 8);
            const int gr = g0 - 1 + row;
            r[i] = (v < NVEC && gr >= 0 && gr < rows_total) ? Vec8<bf16>::load(X + ((long)gr * FI) * CH + rem * 8) : Vec8<bf16>::zero();
        }
    }
    __device__ __forceinline__ void put(bf16* lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + i * NTH;
            const int row = v / (FI * 8), rem = v % (FI * 8);
            if (v < NVEC) r[i].store(lds + (row * COLS + (rem >> 3) + 1) * PITCH + (rem & 7) * 8);
        }
    }
    __device__ __forceinline__ static void zero_border(bf16* lds) {       // columns 0 and FI+1 of every row
        for (int i = threadIdx.x; i < (TT + 2) * 2 * 8; i += NTH) {
            const int row = i / 16, side = (i >> 3) & 1, c8 = i & 7;
            Vec8<bf16>::zero().store(lds + (row * COLS + side * (FI + 1)) * PITCH + c8 * 8);
        }
    }
};

// NW = 8: 512 threads, one 32-pixel row-tile per wave: two waves per SIMD, so one wave's MFMAs cover the other's LDS reads
// (with 4 waves the tile took ~15 us against 1.9 us of MFMA work).
template <int FI, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv3x3_kernel(ConvArgs a) {
    using H = Halo<FI, PIXP, 64 * NW>;
    constexpr int RT = 8 / NW;                                    // 32-pixel row-tiles per wave
    constexpr int TT = H::TT, COLS = H::COLS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const wl = reinterpret_cast<bf16*>(smem);               // [64][WPITCH]
    bf16* const hl = wl + CH * WPITCH;                            // halo tile
    float* const red = reinterpret_cast<float*>(hl);              // reused after the loop: [4][64][2]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    for (int v = threadIdx.x; v < CH * KTOT / 8; v += 64 * NW) {
        const int row = v / (KTOT / 8), c = v % (KTOT / 8);
        Vec8<bf16>::load(a.W + row * KTOT + c * 8).store(wl + row * WPITCH + c * 8);
    }
    H::zero_border(hl);

    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
    H halo;
    int tile = blockIdx.x;
    if (tile < a.tiles) { halo.load(a.X, tile * TT, a.rows_total); halo.put(hl); }
    __syncthreads();

    for (; tile < a.tiles; tile += gridDim.x) {
        const int g0 = tile * TT;
        const int next = tile + gridDim.x;
        if (next < a.tiles) halo.load(a.X, next * TT, a.rows_total);

        // per-lane geometry of the two 32-pixel row-tiles this wave owns
        int tl[RT], fcol[RT];
        bool ok[RT][3];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int rt = RT * wave + i;
            tl[i] = FI == 32 ? rt : 2 * rt + (m >> 4);
            fcol[i] = FI == 32 ? m : (m & 15);
            const int g = g0 + tl[i];
            const int ti = g % a.Ti;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) ok[i][dt] = g < a.rows_total && ti + dt - 1 >= 0 && ti + dt - 1 < a.Ti;
        }
        f32x16 acc[RT][2];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dt = tap % 3, df = tap / 3;                 // tap = kh*3 + kw: kh walks mel (f), kw walks time (t)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                Vec8<bf16> fa[RT], fb[2];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const bf16* p = hl + ((tl[i] + dt) * COLS + fcol[i] + df) * PIXP + cc * 16 + 8 * h;
                    fa[i] = ok[i][dt] ? Vec8<bf16>::load(p) : Vec8<bf16>::zero();
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = Vec8<bf16>::load(wl + (j * 32 + m) * WPITCH + tap * 64 + cc * 16 + 8 * h);
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].v, fb[j].v, acc[i][j], 0, 0, 0);
            }
        }
        // epilogue: C/D map col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*h (pixel of the row-tile)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int rt = RT * wave + i;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = j * 32 + m;
                const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int t_local = FI == 32 ? rt : 2 * rt + (row >> 4);
                    const int f = FI == 32 ? row : (row & 15);
                    const int g = g0 + t_local;
                    if (g < a.rows_total) {
                        const float v = acc[i][j][r] + bias;
                        ssum[j] += v;
                        ssq[j] += v * v;
                        if (a.out_f32) reinterpret_cast<float*>(a.Y)[((long)g * FI + f) * CH + co] = v;
                        else a.Y[((long)g * FI + f) * CH + co] = (bf16)v;
                    }
                }
            }
        }
        __syncthreads();                       // every wave is done reading this halo tile
        if (next < a.tiles) halo.put(hl);
        __syncthreads();
    }
    if (a.sum) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float s = ssum[j] + __shfl_xor(ssum[j], 32, 64), q = ssq[j] + __shfl_xor(ssq[j], 32, 64);
            if (h == 0) { red[(wave * 64 + j * 32 + m) * 2] = s; red[(wave * 64 + j * 32 + m) * 2 + 1] = q; }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += (double)red[(w * 64 + c) * 2 + k];
            atomicAdd(k == 0 ? &a.sum[c] : &a.sumsq[c], t);
        }
    }
}

struct WgradArgs {
    const bf16* dY; const bf16* X; float* dW; int Ti, rows_total, tiles;
};

// KG = 2: 8 waves; waves 4-7 take the second half of each tile's pixels (the contraction dimension) with their own 9
// accumulators - two waves per SIMD overlap LDS reads and MFMAs; both halves end in the same fp32 atomics.
template <int FI, int KG>
__global__ __launch_bounds__(256 * KG, 1) void conv3x3_wgrad_kernel(WgradArgs a) {
    using H = Halo<FI, PIXW, 256 * KG>;
    constexpr int TT = H::TT, COLS = H::COLS;
    constexpr int NTH = 256 * KG;
    constexpr int NVY = 256 * 8 / NTH;                            // 16-byte vectors of the dY tile per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const yl = reinterpret_cast<bf16*>(smem);               // dY tile [256 pixels][PIXW]
    bf16* const hl = yl + 256 * PIXW;                             // X halo tile

    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, grp = threadIdx.x >> 8;
    const int h = lane >> 5, half = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int cot = wave & 1;                                     // channel-out tile of this wave
    const int n0 = (wave >> 1) * 9;                               // first of its 9 (tap, ci-half) column tiles
    H::zero_border(hl);

    f32x16 acc[9];
#pragma unroll
    for (int n = 0; n < 9; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    H halo;
    Vec8<bf16> ry[NVY];
    auto load_y = [&](int g0) {
#pragma unroll
        for (int i = 0; i < NVY; ++i) {
            const int v = threadIdx.x + i * NTH;                 // pixel = v>>3, chunk = v&7
            const long pix = (long)g0 * FI + (v >> 3);
            ry[i] = pix < (long)a.rows_total * FI ? Vec8<bf16>::load(a.dY + pix * CH + (v & 7) * 8) : Vec8<bf16>::zero();
        }
    };
    auto put_y = [&]() {
#pragma unroll
        for (int i = 0; i < NVY; ++i) {
            const int v = threadIdx.x + i * NTH;
            ry[i].store(yl + (v >> 3) * PIXW + (v & 7) * 8);
        }
    };
    int tile = blockIdx.x;
    if (tile < a.tiles) { halo.load(a.X, tile * TT, a.rows_total); load_y(tile * TT); halo.put(hl); put_y(); }
    __syncthreads();

    for (; tile < a.tiles; tile += gridDim.x) {
        const int g0 = tile * TT;
        const int next = tile + gridDim.x;
        if (next < a.tiles) { halo.load(a.X, next * TT, a.rows_total); load_y(next * TT); }

#pragma unroll 1
        for (int ks = grp * (16 / KG); ks < (grp + 1) * (16 / KG); ++ks) {   // 16 pixels per k-step, all inside one t-row
            const int t_local = FI == 32 ? (ks >> 1) : ks;
            const int f0 = FI == 32 ? (ks & 1) * 16 : 0;
            const int g = g0 + t_local;
            if (g >= a.rows_total) break;                         // wave-uniform: the remaining rows are padding
            const int ti = g % a.Ti;
            // A = dY^T: rows = co, k = pixel.  lane gets co = cot*32 + (lane&31), pixels 8h + {0..7}
            const bf16* ay = yl + (ks * 16 + 8 * h + q) * PIXW + cot * 32 + 16 * half + 4 * p;
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)ay);
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(ay + 4 * PIXW));
            const bf16x8 fa = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                const int nt = n0 + n, tap = nt >> 1, cih = nt & 1;
                const int dt = tap % 3, df = tap / 3;
                if (ti + dt - 1 < 0 || ti + dt - 1 >= a.Ti) continue;       // wave-uniform: tap leaves the image in t
                const bf16* bx = hl + ((t_local + dt) * COLS + f0 + df + 8 * h + q) * PIXW + cih * 32 + 16 * half + 4 * p;
                const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)bx);
                const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(bx + 4 * PIXW));
                const bf16x8 fb = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[n], 0, 0, 0);
            }
        }
        __syncthreads();
        if (next < a.tiles) { halo.put(hl); put_y(); }
        __syncthreads();
    }
    // dW[co][tap*64 + ci] += acc: col = lane&31 -> ci, row -> co
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        const int nt = n0 + n, tap = nt >> 1, cih = nt & 1;
        const int col = tap * 64 + cih * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            atomicAdd(&a.dW[co * KTOT + col], acc[n][r]);
        }
    }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
           hipSuccess ? 0 : -1;
}

}  // namespace

// X, W, Y bf16.  W = packed [64][576] (audiossl_pack_conv_w: Wf for the forward, Wd for the data gradient).
// bias / sum / sumsq may be NULL; sum and sumsq (fp64 [64]) are zeroed here.  Fi must be 32 or 16.
// out_f32: Y is float (used for the data gradient that feeds a BatchNorm backward).
extern "C" int audiossl_conv3x3_fwd(const void* X, const void* W, const float* bias, void* Y, int out_f32, double* sum,
                                    double* sumsq, int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(X && W && Y && N > 0 && Ti > 0 && (Fi == 32 || Fi == 16) && (!sum == !sumsq));
    if (!ASSL_ALIGNED16(X) || !ASSL_ALIGNED16(W) || !ASSL_ALIGNED16(Y)) return ASSL_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sum) {
        if (sumsq == sum + 64) {
            ASSL_ZERO(sum, sizeof(double) * 128, s);
        } else {
            ASSL_ZERO(sum, sizeof(double) * 64, s);
            ASSL_ZERO(sumsq, sizeof(double) * 64, s);
        }
    }
    const int rows = N * Ti, TT = 256 / Fi, tiles = (rows + TT - 1) / TT;
    ConvArgs a{static_cast<const bf16*>(X), static_cast<const bf16*>(W), bias, static_cast<bf16*>(Y), sum, sumsq, Ti, rows, tiles, out_f32};
    const int grid = tiles < 256 ? tiles : 256;
    static const int nw = getenv("AUDIOSSL_CONV_WAVES") ? atoi(getenv("AUDIOSSL_CONV_WAVES")) : 8;
    static bool attr[4] = {false, false, false, false};
#define CONV_LAUNCH(FI_, NW_, SLOT)                                                                       \
    do {                                                                                                  \
        const size_t lds = sizeof(bf16) * (CH * WPITCH + Halo<FI_, PIXP>::ELEMS);                         \
        if (!attr[SLOT]) { if (set_lds(conv3x3_kernel<FI_, NW_>, lds)) return ASSL_ELAUNCH; attr[SLOT] = true; } \
        hipLaunchKernelGGL((conv3x3_kernel<FI_, NW_>), dim3(grid), dim3(64 * NW_), lds, s, a);           \
    } while (0)
    if (Fi == 32) { if (nw == 8) CONV_LAUNCH(32, 8, 0); else CONV_LAUNCH(32, 4, 1); }
    else          { if (nw == 8) CONV_LAUNCH(16, 8, 2); else CONV_LAUNCH(16, 4, 3); }
#undef CONV_LAUNCH
    ASSL_LAUNCH_CHECK();
}

// dWp fp32 [64][576] += dY^T * im2col(X)   (caller zeroes dWp; unpack with audiossl_unpack_conv_dw)
extern "C" int audiossl_conv3x3_wgrad(const void* dY, const void* X, float* dWp, int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(dY && X && dWp && N > 0 && Ti > 0 && (Fi == 32 || Fi == 16));
    if (!ASSL_ALIGNED16(X) || !ASSL_ALIGNED16(dY)) return ASSL_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = N * Ti, TT = 256 / Fi, tiles = (rows + TT - 1) / TT;
    WgradArgs a{static_cast<const bf16*>(dY), static_cast<const bf16*>(X), dWp, Ti, rows, tiles};
    const int grid = tiles < 256 ? tiles : 256;
    // measured: the pixel split pays on the 32-pixel-wide layer (304 -> 275 us) and costs on the 16-wide one (112 -> 119 us)
    static const int kg_env = getenv("AUDIOSSL_WGRAD_KG") ? atoi(getenv("AUDIOSSL_WGRAD_KG")) : 0;
    const int kg = kg_env ? kg_env : (Fi == 32 ? 2 : 1);
    static bool attr[4] = {false, false, false, false};
#define WGRAD_LAUNCH(FI_, KG_, SLOT)                                                                      \
    do {                                                                                                  \
        const size_t lds = sizeof(bf16) * (256 * PIXW + Halo<FI_, PIXW>::ELEMS);                          \
        if (!attr[SLOT]) { if (set_lds(conv3x3_wgrad_kernel<FI_, KG_>, lds)) return ASSL_ELAUNCH; attr[SLOT] = true; } \
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<FI_, KG_>), dim3(grid), dim3(256 * KG_), lds, s, a);    \
    } while (0)
    if (Fi == 32) { if (kg == 2) WGRAD_LAUNCH(32, 2, 0); else WGRAD_LAUNCH(32, 1, 1); }
    else          { if (kg == 2) WGRAD_LAUNCH(16, 2, 2); else WGRAD_LAUNCH(16, 1, 3); }
#undef WGRAD_LAUNCH
    ASSL_LAUNCH_CHECK();
}
