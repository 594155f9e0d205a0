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
    int Ti, rows_total, tiles, out_f32, dbg, stat_rep;
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

    // The MFMAs compute Y^T (A = weights: rows = output channels, B = pixels: columns), so a lane ends up with ONE pixel and
    // 16 channels per accumulator in runs of four consecutive ones (C/D map: column = lane & 31 = pixel of the row-tile,
    // row = (r & 3) + 8 * (r >> 2) + 4 * h = channel): the epilogue stores 8 bytes (bf16) or 16 bytes (fp32) per lane and
    // run.  With channels across lanes it issued one 2-byte store per element - 256 wave-stores of 128 bytes per tile, which
    // took ~5x longer than the tile's MFMAs.  Per-channel statistics are kept per lane and folded across lanes once.
    float ssum[2][16], ssq[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
    f32x4 bias4[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bias4[j][q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + j * 32 + 8 * q + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
    H halo;
    int tile = blockIdx.x;
    if (tile < a.tiles) { halo.load(a.X, tile * TT, a.rows_total); halo.put(hl); }
    __syncthreads();

    for (; tile < a.tiles; tile += gridDim.x) {
        const int g0 = tile * TT;
        const int next = tile + gridDim.x;
        if (next < a.tiles && !(a.dbg & 2)) halo.load(a.X, next * TT, a.rows_total);

        // per-lane geometry of the 32-pixel row-tiles this wave owns
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

        if (!(a.dbg & 4))
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j].v, fa[i].v, acc[i][j], 0, 0, 0);
            }
        }
        // epilogue: this lane's pixel, channels j * 32 + 8 * q + 4 * h + (0..3) in accumulator registers 4q..4q+3
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int g = g0 + tl[i];
            if (g < a.rows_total && !(a.dbg & 1)) {
                const long o = ((long)g * FI + fcol[i]) * CH;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[i][j][4 * q + e] + bias4[j][q][e];
                            ssum[j][4 * q + e] += v[e];
                            ssq[j][4 * q + e] += v[e] * v[e];
                        }
                        const int co = j * 32 + 8 * q + 4 * h;
                        if (a.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.Y) + o + co) = v;
                        else *reinterpret_cast<bf16x4*>(a.Y + o + co) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    }
            }
        }
        __syncthreads();                       // every wave is done reading this halo tile
        if (next < a.tiles) halo.put(hl);
        __syncthreads();
    }
    if (a.sum) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = ssum[j][r], q = ssq[j][r];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }   // over the 32 pixels
                if (m == 0) {
                    const int co = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    red[(wave * 64 + co) * 2] = s; red[(wave * 64 + co) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += (double)red[(w * 64 + c) * 2 + k];
            const int rep = (blockIdx.x % a.stat_rep) * 128;     // replicated accumulators: same-address atomics serialise
            atomicAdd(k == 0 ? &a.sum[rep + c] : &a.sumsq[rep + c], t);
        }
    }
}

// ---- weights-stationary variant (the default): the 64 x 576 weight matrix lives in REGISTERS (72 MFMA A-fragments = 288
// VGPRs per wave, loaded once per persistent workgroup), so the k-loop reads only pixel fragments from LDS: 2 ds_read_b128
// per 4 MFMAs instead of 3 per 2 (the kernel above was LDS-read bound: 48 us of k-loop against 24 us of MFMA time at
// B = 512, tools/conv_fwd_ablate.py).  4 waves x (64 pixels x 64 channels); 256-pixel tiles as above.
// The halo tile is double-buffered and filled by LDS-DMA (buffer_load ... lds, no staging registers): pixel pitch 128 bytes,
// 16-byte chunk c of the pixel in halo column `col` stored at chunk position c ^ ((col >> 1) & 7) (16 consecutive columns then
// cover all 64 banks once: conflict-free ds_read_b128) - the swizzle is applied to the
// DMA's source address, rows outside [0, rows_total) come back as zeros from the buffer bounds check.  One raw s_barrier
// per tile; the epilogue's global stores are issued after the wait for the next tile's DMA and drain under the next k-loop.
// Epilogue: the MFMA result layout (lane = pixel, 4 consecutive channels per register group) would store 8-byte pieces - 64
// separate write requests per instruction, which made the stores the second largest cost of the kernel above (37 us at
// B = 512).  Each wave transposes its 32-pixel row-tiles through a private, swizzled LDS staging area instead and writes whole
// pixels: 1 KB contiguous per store instruction.
template <int FI>
struct WsGeom {
    static constexpr int TT = 256 / FI, COLS = FI + 2, ROWS = TT + 2, SEGS = FI / 8;
    static constexpr int UNITS = ROWS * SEGS;                    // DMA instructions per tile (8 pixels = 1 KB each)
    static constexpr int UPW = (UNITS + 3) / 4;                  // per wave
    static constexpr int BYTES = ROWS * COLS * 128;
};

template <int FI, bool STATS, bool F32OUT, int PF = 2>
__global__ __launch_bounds__(256, 1) void conv3x3_ws_kernel(ConvArgs a) {
    using G = WsGeom<FI>;
    constexpr int TT = G::TT, COLS = G::COLS;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const biasl = reinterpret_cast<float*>(smem + 2 * G::BYTES);          // [64]
    float* const red = biasl + 64;                                              // [4][64][2]
    constexpr int OPIX = F32OUT ? 256 : 128;                                    // bytes per staged output pixel
    char* const stg = reinterpret_cast<char*>(red + 4 * 64 * 2) + (threadIdx.x >> 6) * 64 * OPIX;   // this wave's 2 x 32-pixel staging tiles
    char* const zblk = reinterpret_cast<char*>(red + 4 * 64 * 2) + 4 * 64 * OPIX;                   // 512 bytes of zeros
    if (threadIdx.x < 32) *reinterpret_cast<f32x4*>(zblk + threadIdx.x * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    long* const rts = (!STATS && (a.dbg & 16) && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
                          ? reinterpret_cast<long*>(a.sum) + 64 + (blockIdx.x ? 8 : 0) : nullptr;
    if (rts) rts[0] = __builtin_amdgcn_s_memrealtime();

    // weights: A fragments, row = output channel j * 32 + m, k = ks * 16 + 8 h .. + 7.  Staged through LDS (the halo buffers
    // are still free): read straight from global memory every lane fetches its own 1,152-byte-strided row - 4,608 line
    // requests per wave, 8 us before the first tile could start.
    {
        bf16* const wl = reinterpret_cast<bf16*>(smem);           // [64][WPITCH]
        Vec8<bf16> tmp[CH * KTOT / 8 / 256];                       // 18 loads in flight, then 18 LDS stores
#pragma unroll
        for (int k = 0; k < CH * KTOT / 8 / 256; ++k) tmp[k] = Vec8<bf16>::load(a.W + (threadIdx.x + 256 * k) * 8);
#pragma unroll
        for (int k = 0; k < CH * KTOT / 8 / 256; ++k) {
            const int v = threadIdx.x + 256 * k, row = v / (KTOT / 8), c = v % (KTOT / 8);
            tmp[k].store(wl + row * WPITCH + c * 8);
        }
        __syncthreads();
    }
    Vec8<bf16> wr[2][36];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 36; ++ks)
            wr[j][ks] = Vec8<bf16>::load(reinterpret_cast<const bf16*>(smem) + (j * 32 + m) * WPITCH + ks * 16 + 8 * h);
    // pin the homes of the 288 weight registers: channels 0-31 in VGPRs, 32-63 in AGPRs next to the accumulators.  Left to
    // itself the allocator spreads them over both files and copies four registers into a VGPR temporary before every MFMA,
    // each copy waiting for the MFMAs in flight to release that temporary (k-loop 50 cycles per MFMA instead of 32).
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) {
        asm volatile("" : "+v"(wr[0][ks].v));
        asm volatile("" : "+a"(wr[1][ks].v));
    }

    __syncthreads();                                                            // every wave holds its weights: the buffers are free
    // zero the border columns of both buffers once (the DMA only ever writes interior columns)
    for (int i = threadIdx.x; i < 2 * G::ROWS * 2 * 8; i += 256) {
        const int b = i / (G::ROWS * 16), r = (i / 16) % G::ROWS, side = (i >> 3) & 1, c8 = i & 7;
        *reinterpret_cast<f32x4*>(smem + b * G::BYTES + (r * COLS + side * (FI + 1)) * 128 + c8 * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (threadIdx.x < 64) biasl[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;

    if (rts) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); rts[3] = __builtin_amdgcn_s_memrealtime(); }
    // DMA addressing: lane -> (pixel of the 8-pixel segment, chunk position); source chunk = position ^ (halo column & 7)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.X), 0, (unsigned)((long)a.rows_total * FI * 128), 0x00020000);
    const int pl = lane >> 3, cp = lane & 7;
    const int key0 = ((1 + pl) >> 1) & 7;                                       // swizzle key of halo column 1 + 8 seg + pl, seg even
    const int voff0 = pl * 128 + ((cp ^ key0) << 4), voff1 = pl * 128 + ((cp ^ key0 ^ 4) << 4);
    auto dma_unit = [&](int g0, char* buf, int i) {
        {
            const int u = wave + 4 * i;                                         // wave-uniform
            if (u < G::UNITS) {
                const int row = u / G::SEGS, seg = u % G::SEGS;
                const int gr = g0 - 1 + row;
                const int so = (gr >= 0 && gr < a.rows_total) ? (gr * FI + 8 * seg) * 128 : 0x7FFFFF00;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(buf + (row * COLS + 1 + 8 * seg) * 128), 16, (seg & 1) ? voff1 : voff0, so, 0, 0);
            }
        }
    };
    auto dma_tile = [&](int g0, char* buf) {
#pragma unroll
        for (int i = 0; i < G::UPW; ++i) dma_unit(g0, buf, i);
    };

    // per-lane geometry of this wave's two 32-pixel row-tiles
    int tl[2];
    const int fcol = FI == 32 ? m : (m & 15);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = 2 * wave + i;
        tl[i] = FI == 32 ? rt : 2 * rt + (m >> 4);
    }
    int hs[3];                                                                  // ((h ^ swizzle key of column fcol + df) << 4
#pragma unroll
    for (int df = 0; df < 3; ++df) hs[df] = (h ^ (((fcol + df) >> 1) & 7)) << 4;

    // batch statistics are taken from the values as they are stored (after the transpose a lane always sees the same NS
    // channels: 2 * NS accumulators instead of 64 - the weights leave no room for more)
    constexpr int NS = F32OUT ? 4 : 8;
    float ssum[NS], ssq[NS];
    if (STATS) {
#pragma unroll
        for (int e = 0; e < NS; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    }

    int tile = blockIdx.x, cur = 0;
    if (tile < a.tiles) dma_tile(tile * TT, smem);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    long* const ts = (!STATS && (a.dbg & 16) && blockIdx.x == 0 && threadIdx.x == 0) ? reinterpret_cast<long*>(a.sum) : nullptr;
    int tsi = 0;
    if (rts) rts[1] = __builtin_amdgcn_s_memrealtime();
    for (; tile < a.tiles; tile += gridDim.x, cur ^= 1) {
        if (ts) ts[tsi++] = __builtin_amdgcn_s_memtime();
        const int g0 = tile * TT;
        const int next = tile + gridDim.x;
        if (next < a.tiles && !(a.dbg & 2)) dma_tile(next * TT, smem + (cur ^ 1) * G::BYTES);
        const char* const hl = smem + cur * G::BYTES;

        // rows above / below the image (the global row enumeration runs across images) must read as zeros: such a
        // (row-tile, dt) gets the address of a block of zeros instead of a select on every fragment
        int bdt[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = g0 + tl[i];
            const int ti = g % a.Ti;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const bool ok = g < a.rows_total && ti + dt - 1 >= 0 && ti + dt - 1 < a.Ti;
                bdt[i][dt] = ok ? (int)(size_t)((lptr_t)const_cast<char*>(hl)) + ((tl[i] + dt) * COLS + fcol) * 128
                                : (int)(size_t)((lptr_t)zblk);
            }
        }
        f32x16 acc[2][2];                                         // start from the bias: channel j * 32 + 8 q + 4 h + e in register 4 q + e
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(biasl + j * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[0][j][4 * q + e] = b4[e]; acc[1][j][4 * q + e] = b4[e]; }
            }

        // pixel fragments are fetched PF k-steps ahead of the MFMAs that consume them (one wave per SIMD: nothing else hides
        // the LDS latency); k-step ks = tap * 4 + cc, tap = kh*3 + kw: kh walks mel (f), kw walks time (t)
        Vec8<bf16> fa[PF + 1][2];
        auto fetch = [&](int ks, Vec8<bf16> (&f)[2]) {
            const int tap = ks >> 2, cc = ks & 3, dt = tap % 3, df = tap / 3;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                f[i] = Vec8<bf16>::load(reinterpret_cast<const bf16*>((const char*)(lptr_t)(size_t)(bdt[i][dt] + df * 128 + (hs[df] ^ (cc * 32)))));
        };
        if (!(a.dbg & 4)) {
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) fetch(ks, fa[ks]);
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) {
            if (ks + PF < 36) fetch(ks + PF, fa[(ks + PF) % (PF + 1)]);
            __builtin_amdgcn_sched_barrier(0);                    // keep the reads ahead: the scheduler otherwise sinks them to their use
            Vec8<bf16> (&f)[2] = fa[ks % (PF + 1)];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[j][ks].v, f[i].v, acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (ts) ts[tsi++] = __builtin_amdgcn_s_memtime();
        // the next tile's halo has had the whole k-loop to land; the stores below then drain under the next k-loop
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ts) ts[tsi++] = __builtin_amdgcn_s_memtime();
        // epilogue: this lane's pixel, channels j * 32 + 8 * q + 4 * h + (0..3) in accumulator registers 4q..4q+3 -> staging
        if (!(a.dbg & 1)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                char* const st = stg + i * 32 * OPIX;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int slot = j * 8 + 2 * q + h;       // 16 slots of 4 channels per pixel
                        const f32x4 v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        if (F32OUT) *reinterpret_cast<f32x4*>(st + m * 256 + ((slot ^ (m & 15)) << 4)) = v;
                        else *reinterpret_cast<bf16x4*>(st + m * 128 + ((slot ^ ((m >> 1) & 15)) << 3)) =
                                 bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // whole pixels back out of the staging tiles: 1 KB contiguous per store instruction
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* const st = stg + i * 32 * OPIX;
                const int grow = g0 + (2 * wave + i) * (32 / FI); // the row-tile's 32 consecutive output pixels start at row grow
                const int prow_ok = a.rows_total - grow;          //   (FI = 16: two rows, the second may be past the end)
                char* const yrow = reinterpret_cast<char*>(a.Y) + (long)grow * FI * 64 * (F32OUT ? 4 : 2);
                if (F32OUT) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int p = it * 4 + (lane >> 4), c = lane & 15;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(st + p * 256 + ((c ^ (p & 15)) << 4));
                        if ((p >> (FI == 32 ? 5 : 4)) < prow_ok) {
                            *reinterpret_cast<f32x4*>(yrow + p * 256 + c * 16) = v;
                            if (STATS) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int p = it * 8 + (lane >> 3), c = lane & 7, k = (p >> 1) & 15;
                        f32x4 v = *reinterpret_cast<const f32x4*>(st + p * 128 + ((c ^ (k >> 1)) << 4));
                        if (k & 1) v = f32x4{v[2], v[3], v[0], v[1]};                  // the two 8-byte slots of the chunk sit swapped
                        if ((p >> (FI == 32 ? 5 : 4)) < prow_ok) {
                            *reinterpret_cast<f32x4*>(yrow + p * 128 + c * 16) = v;
                            if (STATS) {
                                const bf16x8 hv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const float x = (float)hv[e];
                                    ssum[e] += x; ssq[e] += x * x;
                                }
                            }
                        }
                    }
                }
            }
        }
        if (ts) ts[tsi++] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave is done reading hb[cur]; every wave's share of hb[cur ^ 1] has landed
    }
    if (rts) rts[2] = __builtin_amdgcn_s_memrealtime();
    if (STATS) {
        // lane l holds channels (l & (64 / NS - 1)) * NS + e; fold the lanes that share them
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            float s = ssum[e], q = ssq[e];
#pragma unroll
            for (int o = 64 / NS; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            if (lane < 64 / NS) {
                const int co = lane * NS + e;
                red[(wave * 64 + co) * 2] = s; red[(wave * 64 + co) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += (double)red[(w * 64 + c) * 2 + k];
            const int rep = (blockIdx.x % a.stat_rep) * 128;     // replicated accumulators: same-address atomics serialise
            atomicAdd(k == 0 ? &a.sum[rep + c] : &a.sumsq[rep + c], t);
        }
    }
}

// accumulator n (0..8) of wave w (0..3 inside its pixel half) -> (channel-out tile, (tap, ci-half) column tile): n < 8 walks
// the wave's four column tiles 4w..4w+3 with both channel-out tiles (each X fragment feeds two MFMAs), n = 8 is one of the four
// left-over (column tile 16 / 17, channel-out tile) pairs.  7 fragments per 9 MFMAs instead of 10.
__host__ __device__ __forceinline__ int wg_cot(int w, int n) { return n < 8 ? (n & 1) : (w & 1); }
__host__ __device__ __forceinline__ int wg_nt(int w, int n) { return n < 8 ? 4 * w + (n >> 1) : 16 + (w >> 1); }

struct WgradArgs {
    const bf16* dY; const bf16* X; float* dW; int Ti, rows_total, tiles;
    float* partial;           // [gridDim.x * KG][64 * 576] per-(workgroup, pixel-half) results, or null: fp32 atomics into dW
    int dbg;                  // AUDIOSSL_CONV_DBG: 2 no loads of the next tile, 4 no k-loop (tools/conv_bench.py)
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
        if (next < a.tiles && !(a.dbg & 2)) { halo.load(a.X, next * TT, a.rows_total); load_y(next * TT); }

#pragma unroll 1
        for (int ks = grp * (16 / KG); ks < (grp + 1) * (16 / KG) && !(a.dbg & 4); ++ks) {   // 16 pixels per k-step, all inside one t-row
            const int t_local = FI == 32 ? (ks >> 1) : ks;
            const int f0 = FI == 32 ? (ks & 1) * 16 : 0;
            const int g = g0 + t_local;
            if (g >= a.rows_total) break;                         // wave-uniform: the remaining rows are padding
            const int ti = g % a.Ti;
            const bool okt[3] = {ti - 1 >= 0, true, ti + 1 < a.Ti};          // wave-uniform: tap row inside the image in t
            // A = dY^T: rows = co, k = pixel.  lane gets co = cot*32 + (lane&31), pixels 8h + {0..7}; both channel-out tiles
            bf16x8 fa[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const bf16* ay = yl + (ks * 16 + 8 * h + q) * PIXW + c2 * 32 + 16 * half + 4 * p;
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)ay);
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(ay + 4 * PIXW));
                fa[c2] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            }
            bf16x8 fb;
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                if (n == 8 || (n & 1) == 0) {                     // a new column tile: its X fragment (shared by the pair n, n + 1)
                    const int nt = wg_nt(wave, n), tap = nt >> 1, cih = nt & 1;
                    const int dt = tap % 3, df = tap / 3;
                    // no branch around a tap that leaves the image (the halo row exists in LDS, it belongs to the neighbouring
                    // image): the fragment is zeroed instead, so the transposing reads of a k-step can all be in flight
                    const bf16* bx = hl + ((t_local + dt) * COLS + f0 + df + 8 * h + q) * PIXW + cih * 32 + 16 * half + 4 * p;
                    const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)bx);
                    const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(bx + 4 * PIXW));
                    const bf16 z = (bf16)0.f;
                    fb = okt[dt] ? bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]} : bf16x8{z, z, z, z, z, z, z, z};
                }
                const int c2 = wg_cot(wave, n);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2 ? fa[1] : fa[0], fb, acc[n], 0, 0, 0);
            }
        }
        __syncthreads();
        if (next < a.tiles) { halo.put(hl); put_y(); }
        __syncthreads();
    }
    // dW[co][tap*64 + ci] += acc: col = lane&31 -> ci, row -> co.  With a workspace every (workgroup, pixel-half) stores its
    // 64 x 576 result with plain stores and wgrad_reduce_kernel folds them: 256 workgroups x 2 halves adding into the same
    // 36,864 addresses with device-scope atomics took longer than the MFMAs of the whole pass.
    if (a.partial) {
        // workspace layout = accumulator layout, element ((wave * 9 + n) * 4 + r / 4) * 64 + lane holds registers 4(r/4)..+3:
        // 16-byte stores, 1 KB contiguous per wave-instruction; wgrad_reduce_kernel undoes the map.  The two pixel halves of
        // the workgroup (KG = 2) are added through LDS first (the tile buffers are free now): one 147 KB result per workgroup
        // instead of two - half the partial traffic and half the work of the fold
        if (KG == 2) {
            f32x4* xch = reinterpret_cast<f32x4*>(smem);
            __syncthreads();
            if (grp == 1) {
#pragma unroll
                for (int n = 0; n < 9; ++n)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        xch[((wave * 9 + n) * 4 + r4) * 64 + lane] = f32x4{acc[n][4 * r4], acc[n][4 * r4 + 1], acc[n][4 * r4 + 2], acc[n][4 * r4 + 3]};
            }
            __syncthreads();
            if (grp == 1) return;
            f32x4* part = reinterpret_cast<f32x4*>(a.partial + (long)blockIdx.x * (CH * KTOT));
#pragma unroll
            for (int n = 0; n < 9; ++n)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 o = xch[((wave * 9 + n) * 4 + r4) * 64 + lane];
                    part[((wave * 9 + n) * 4 + r4) * 64 + lane] = f32x4{acc[n][4 * r4] + o[0], acc[n][4 * r4 + 1] + o[1], acc[n][4 * r4 + 2] + o[2], acc[n][4 * r4 + 3] + o[3]};
                }
            return;
        }
        f32x4* part = reinterpret_cast<f32x4*>(a.partial + (long)blockIdx.x * (CH * KTOT));
#pragma unroll
        for (int n = 0; n < 9; ++n)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                part[((wave * 9 + n) * 4 + r4) * 64 + lane] = f32x4{acc[n][4 * r4], acc[n][4 * r4 + 1], acc[n][4 * r4 + 2], acc[n][4 * r4 + 3]};
        return;
    }
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        const int nt = wg_nt(wave, n), tap = nt >> 1, cih = nt & 1;
        const int col = tap * 64 + cih * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wg_cot(wave, n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            atomicAdd(&a.dW[co * KTOT + col], acc[n][r]);
        }
    }
}

// dW[i] += sum over parts; blockIdx.y takes every gridDim.y-th part (a handful of atomics per address remain)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ dW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= CH * KTOT) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = blockIdx.y;
    const int step = gridDim.y;
    for (; w + 3 * step < nparts; w += 4 * step) {
        s0 += partial[(long)w * (CH * KTOT) + i];
        s1 += partial[(long)(w + step) * (CH * KTOT) + i];
        s2 += partial[(long)(w + 2 * step) * (CH * KTOT) + i];
        s3 += partial[(long)(w + 3 * step) * (CH * KTOT) + i];
    }
    for (; w < nparts; w += step) s0 += partial[(long)w * (CH * KTOT) + i];
    // i = (((wave * 9 + n) * 4 + r4) * 64 + lane) * 4 + e  ->  dW[co][tap * 64 + ci]
    const int e = i & 3, lane = (i >> 2) & 63, r4 = (i >> 8) & 3, wn = i >> 10;
    const int wave = wn / 9, n = wn - wave * 9;
    const int nt = wg_nt(wave, n), tap = nt >> 1, cih = nt & 1;
    const int col = tap * 64 + cih * 32 + (lane & 31);
    const int co = wg_cot(wave, n) * 32 + e + 8 * r4 + 4 * (lane >> 5);
    atomicAdd(&dW[co * KTOT + col], (s0 + s1) + (s2 + s3));
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
           hipSuccess ? 0 : -1;
}

}  // namespace

// X, W, Y bf16.  W = packed [64][576] (audiossl_pack_conv_w: Wf for the forward, Wd for the data gradient).
// bias / sum / sumsq may be NULL; sum and sumsq (fp64 [64]) are zeroed here.  Fi must be 32 or 16.
// stat_replicas > 1: sum points to [stat_replicas][128] doubles (replica r: sums at r*128, sums of squares at r*128 + 64, sumsq ==
// sum + 64); workgroup i adds into replica i % stat_replicas and the consumer folds them (bn_relu_pool_train_fwd).
// out_f32: Y is float (used for the data gradient that feeds a BatchNorm backward).
extern "C" int audiossl_conv3x3_fwd(const void* X, const void* W, const float* bias, void* Y, int out_f32, double* sum,
                                    double* sumsq, int stat_replicas, int N, int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(X && W && Y && N > 0 && Ti > 0 && (Fi == 32 || Fi == 16) && (!sum == !sumsq));
    ASSL_REQUIRE(stat_replicas >= 1 && stat_replicas <= 64 && (stat_replicas == 1 || !sum || sumsq == sum + 64));
    if (!ASSL_ALIGNED16(X) || !ASSL_ALIGNED16(W) || !ASSL_ALIGNED16(Y)) return ASSL_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // AUDIOSSL_CONV_DBG (diagnostics, tools/conv_fwd_ablate.py / conv_ts.py): 1 no epilogue, 2 no halo DMA, 4 no k-loop, 16 cycle stamps
    // - honoured only by -DAUDIOSSL_ABLATE builds (python audio-ssl_amd/build.py --ablate): the shipped library cannot be switched
    //   into a mode that produces wrong results
#ifdef AUDIOSSL_ABLATE
    static const int conv_dbg = getenv("AUDIOSSL_CONV_DBG") ? atoi(getenv("AUDIOSSL_CONV_DBG")) : 0;
#else
    constexpr int conv_dbg = 0;
#endif
    if (sum && !(conv_dbg & 16)) {
        if (sumsq == sum + 64) {
            ASSL_ZERO(sum, sizeof(double) * 128 * stat_replicas, s);
        } else {
            ASSL_ZERO(sum, sizeof(double) * 64, s);
            ASSL_ZERO(sumsq, sizeof(double) * 64, s);
        }
    }
    const int rows = N * Ti, TT = 256 / Fi, tiles = (rows + TT - 1) / TT;
    ConvArgs a{static_cast<const bf16*>(X), static_cast<const bf16*>(W), bias, static_cast<bf16*>(Y), sum, sumsq, Ti, rows, tiles, out_f32, conv_dbg, stat_replicas};
    const int grid = tiles < 256 ? tiles : 256;
    // default: weights-stationary kernel (AUDIOSSL_CONV_WS=0 selects the LDS-weights kernel above)
    static const int ws = getenv("AUDIOSSL_CONV_WS") ? atoi(getenv("AUDIOSSL_CONV_WS")) : 1;
    if (ws && (long)rows * Fi * 128 < 0x7FFFFF00L) {
        static bool wattr[8] = {false, false, false, false, false, false, false, false};
#define WS_LAUNCH(FI_, ST_, F32_, SLOT)                                                                    \
    do {                                                                                                  \
        const size_t lds = 2 * WsGeom<FI_>::BYTES + sizeof(float) * (64 + 4 * 64 * 2) + 4 * 64 * (F32_ ? 256 : 128) + 512; \
        if (!wattr[SLOT]) { if (set_lds(conv3x3_ws_kernel<FI_, ST_, F32_>, lds)) return ASSL_ELAUNCH; wattr[SLOT] = true; } \
        hipLaunchKernelGGL((conv3x3_ws_kernel<FI_, ST_, F32_>), dim3(grid), dim3(256), lds, s, a);       \
    } while (0)
        if (Fi == 32) {
            if (sum && !(conv_dbg & 16)) { if (out_f32) WS_LAUNCH(32, true, true, 0); else WS_LAUNCH(32, true, false, 1); }
            else     { if (out_f32) WS_LAUNCH(32, false, true, 2); else WS_LAUNCH(32, false, false, 3); }
        } else {
            if (sum) { if (out_f32) WS_LAUNCH(16, true, true, 4); else WS_LAUNCH(16, true, false, 5); }
            else     { if (out_f32) WS_LAUNCH(16, false, true, 6); else WS_LAUNCH(16, false, false, 7); }
        }
#undef WS_LAUNCH
        ASSL_LAUNCH_CHECK();
    }
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
extern "C" int audiossl_conv3x3_wgrad(const void* dY, const void* X, float* dWp, float* workspace, long workspace_floats, int N,
                                      int Ti, int Fi, void* stream) {
    ASSL_REQUIRE(dY && X && dWp && N > 0 && Ti > 0 && (Fi == 32 || Fi == 16));
    if (!ASSL_ALIGNED16(X) || !ASSL_ALIGNED16(dY)) return ASSL_EALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = N * Ti, TT = 256 / Fi, tiles = (rows + TT - 1) / TT;
    const int grid = tiles < 256 ? tiles : 256;
    // workspace (optional): 2 * 256 * 64 * 576 floats cover every launch shape; without it the results go through atomics
    const bool two_stage = workspace != nullptr && workspace_floats >= (long)grid * CH * KTOT;
#ifdef AUDIOSSL_ABLATE
    static const int wg_dbg = getenv("AUDIOSSL_CONV_DBG") ? atoi(getenv("AUDIOSSL_CONV_DBG")) : 0;
#else
    constexpr int wg_dbg = 0;
#endif
    WgradArgs a{static_cast<const bf16*>(dY), static_cast<const bf16*>(X), dWp, Ti, rows, tiles, two_stage ? workspace : nullptr, wg_dbg};
    // measured in isolation (tools/conv_bench.py, B = 512): two-stage with the pixel split 122 / 46 us (32- / 16-wide layer),
    // without it 140 / 51 us; atomics 147 / 67 us without the split and 155 / 91 us with it (twice the atomics)
    static const int kg_env = getenv("AUDIOSSL_WGRAD_KG") ? atoi(getenv("AUDIOSSL_WGRAD_KG")) : 0;
    const int kg = kg_env ? kg_env : (two_stage ? 2 : 1);
    static bool attr[4] = {false, false, false, false};
#define WGRAD_LAUNCH(FI_, KG_, SLOT)                                                                      \
    do {                                                                                                  \
        const size_t lds = max(sizeof(bf16) * (256 * PIXW + Halo<FI_, PIXW>::ELEMS), (size_t)(KG_ == 2 ? 4 * 9 * 16 * 64 * 4 : 0)); \
        if (!attr[SLOT]) { if (set_lds(conv3x3_wgrad_kernel<FI_, KG_>, lds)) return ASSL_ELAUNCH; attr[SLOT] = true; } \
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<FI_, KG_>), dim3(grid), dim3(256 * KG_), lds, s, a);    \
    } while (0)
    if (Fi == 32) { if (kg == 2) WGRAD_LAUNCH(32, 2, 0); else WGRAD_LAUNCH(32, 1, 1); }
    else          { if (kg == 2) WGRAD_LAUNCH(16, 2, 2); else WGRAD_LAUNCH(16, 1, 3); }
#undef WGRAD_LAUNCH
    if (two_stage)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(CH * KTOT / 256, 8), dim3(256), 0, s, workspace, grid, dWp);
    ASSL_LAUNCH_CHECK();
}
